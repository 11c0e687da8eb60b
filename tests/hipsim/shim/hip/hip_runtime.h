// hipsim -- TEST INFRASTRUCTURE ONLY (never shipped, never loaded by the package).
//
// A lockstep SIMT interpreter for the CPU-only CI container: the product's
// .hip sources are compiled UNCHANGED by the host clang++ against this header
// (it shadows <hip/hip_runtime.h> on the include path) into
// tests/hipsim/build/liblxo_sim.so.  Every GPU thread of a workgroup becomes a
// fiber; __syncthreads(), wave shuffles and MFMA builtins are rendezvous points
// that exchange operands between the 64 lane-fibers of a wave.  This checks
// kernel LOGIC (indexing, tiling, barriers, reductions, the C-ABI plumbing)
// without a GPU; it says nothing about speed, and the MFMA lane layouts it
// assumes are themselves verified on real gfx950 by tests/test_gpu_*.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define LXO_HIPSIM 1      // lets a source opt out of what a one-workgroup-at-a-time interpreter cannot run (csrc/xdec.hip)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hipsim {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
extern char g_dyn_lds[163840];      // dynamic LDS of the workgroup being interpreted
void syncthreads();
void wave_sync();
unsigned lane();                 // linear thread id % 64
char* slot(unsigned lane);       // 256-byte exchange slot of a lane of the current wave
unsigned long long live_mask();  // lanes of the current wave that have not exited

template <class T> inline T exchange(T v, unsigned src) {
    static_assert(sizeof(T) <= 256, "slot too small");
    memcpy(slot(lane()), &v, sizeof(T));
    wave_sync();
    T r; memcpy(&r, slot(src & 63u), sizeof(T));
    wave_sync();
    return r;
}
inline float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
}  // namespace hipsim

#define HIP_DYNAMIC_SHARED(type, var) static type* var = reinterpret_cast<type*>(hipsim::g_dyn_lds);
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

#define threadIdx hipsim::g_threadIdx
#define blockIdx hipsim::g_blockIdx
#define blockDim hipsim::g_blockDim
#define gridDim hipsim::g_gridDim
static const int warpSize = 64;
using std::min; using std::max;

inline void __syncthreads() { hipsim::syncthreads(); }

template <class T> inline T __shfl(T v, int src, int width = 64) {
    unsigned l = hipsim::lane();
    unsigned base = l & ~(unsigned)(width - 1);
    return hipsim::exchange(v, base + ((unsigned)src & (unsigned)(width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    unsigned l = hipsim::lane();
    (void)width;
    return hipsim::exchange(v, l ^ (unsigned)mask);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    unsigned l = hipsim::lane();
    unsigned src = ((l & (unsigned)(width - 1)) + d < (unsigned)width) ? l + d : l;
    return hipsim::exchange(v, src);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    unsigned l = hipsim::lane();
    unsigned src = ((l & (unsigned)(width - 1)) >= d) ? l - d : l;
    return hipsim::exchange(v, src);
}
inline unsigned long long __ballot(int pred) {
    unsigned l = hipsim::lane();
    int p = pred ? 1 : 0;
    memcpy(hipsim::slot(l), &p, 4);
    hipsim::wave_sync();
    unsigned long long m = 0, live = hipsim::live_mask();
    for (unsigned i = 0; i < 64; ++i) {
        int q; memcpy(&q, hipsim::slot(i), 4);
        if (((live >> i) & 1ull) && q) m |= 1ull << i;
    }
    hipsim::wave_sync();
    return m;
}
inline int __any(int p) { return __ballot(p) != 0ull; }
inline int __all(int p) { return __ballot(!p) == 0ull; }
template <class T> inline T __builtin_amdgcn_readfirstlane_sim(T v) {
    unsigned long long live = hipsim::live_mask();
    unsigned first = (unsigned)__builtin_ctzll(live);
    return hipsim::exchange(v, first);
}
#define __builtin_amdgcn_readfirstlane(v) __builtin_amdgcn_readfirstlane_sim(v)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_getreg(n) (0u)
#define __builtin_amdgcn_sched_barrier(n) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipsim::wave_sync()
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_s_barrier() hipsim::syncthreads()
#define __builtin_amdgcn_s_sleep(n) ((void)0)
// a counted wait is executed by the whole wave: rendezvous its live lanes so that LDS-DMA pieces issued by
// other lanes of the SAME wave have been copied before this lane reads them (the hardware lands a wave's DMA as a unit)
#define __builtin_amdgcn_s_waitcnt(n) hipsim::wave_sync()


// ---- buffer resources + raw buffer loads (csrc: the double-buffered attention streams).  base + soffset + voffset; a request that
// reaches beyond num_records returns zeros (the hardware's range check) ----
struct __amdgpu_buffer_rsrc_t { const char* base; long long bytes; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_records, int) {
    __amdgpu_buffer_rsrc_t r = {static_cast<const char*>(p), (long long)(unsigned)num_records};
    return r;
}
template <class T> inline T hipsim_buffer_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    T v; memset(&v, 0, sizeof(T));
    const long long o = (long long)(unsigned)voff + (long long)(unsigned)soff;
    if (o + (long long)sizeof(T) <= r.bytes) memcpy(&v, r.base + o, sizeof(T));
    return v;
}
typedef unsigned hipsim_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipsim_u32x2 __attribute__((ext_vector_type(2)));
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) hipsim_buffer_load<hipsim_u32x4>(r, voff, soff)
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) hipsim_buffer_load<hipsim_u32x2>(r, voff, soff)
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) hipsim_buffer_load<unsigned>(r, voff, soff)

// ---- atomics (single host thread: plain read-modify-write) ----
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
inline void __threadfence() {}

// ---- math ----
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __builtin_amdgcn_rcpf_sim(float a) { return 1.0f / a; }
#define __builtin_amdgcn_rcpf(a) __builtin_amdgcn_rcpf_sim(a)
inline float __builtin_amdgcn_exp2f_sim(float a) { return exp2f(a); }
#define __builtin_amdgcn_exp2f(a) __builtin_amdgcn_exp2f_sim(a)
inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }

// ---- MFMA emulation (lane layouts: cdna_hip_programming.md section 3) ----
namespace hipsim {
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf16;
typedef __attribute__((ext_vector_type(8))) unsigned short v8u16;
typedef __attribute__((ext_vector_type(16))) float v16f;
typedef __attribute__((ext_vector_type(4))) float v4f;
struct AB16 { v8u16 a, b; };
struct ABf { float a, b; };

// D[i][j] += sum_k A[i][k] B[k][j];  A[i][k] lives in lane i + 32*(k/8), elem k%8
inline v16f mfma_32x32x16_bf16(v8bf16 a, v8bf16 b, v16f c) {
    AB16 me; me.a = __builtin_bit_cast(v8u16, a); me.b = __builtin_bit_cast(v8u16, b);
    unsigned l = lane();
    memcpy(slot(l), &me, sizeof(me));
    wave_sync();
    unsigned col = l & 31;
    for (int r = 0; r < 16; ++r) {
        unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            AB16 sa, sb;
            memcpy(&sa, slot(row + 32 * (k >> 3)), sizeof(sa));
            memcpy(&sb, slot(col + 32 * (k >> 3)), sizeof(sb));
            acc += bf2f(sa.a[k & 7]) * bf2f(sb.b[k & 7]);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// 16x16x32: A[i][k] in lane i + 16*(k/8), elem k%8; D col = lane&15, row = (lane>>4)*4 + r
inline v4f mfma_16x16x32_bf16(v8bf16 a, v8bf16 b, v4f c) {
    AB16 me; me.a = __builtin_bit_cast(v8u16, a); me.b = __builtin_bit_cast(v8u16, b);
    unsigned l = lane();
    memcpy(slot(l), &me, sizeof(me));
    wave_sync();
    unsigned col = l & 15;
    for (int r = 0; r < 4; ++r) {
        unsigned row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            AB16 sa, sb;
            memcpy(&sa, slot(row + 16 * (k >> 3)), sizeof(sa));
            memcpy(&sb, slot(col + 16 * (k >> 3)), sizeof(sb));
            acc += bf2f(sa.a[k & 7]) * bf2f(sb.b[k & 7]);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// f32 32x32x2: A[i][k] in lane i + 32*k; B[k][j] in lane j + 32*k
inline v16f mfma_32x32x2_f32(float a, float b, v16f c) {
    ABf me{a, b};
    unsigned l = lane();
    memcpy(slot(l), &me, sizeof(me));
    wave_sync();
    unsigned col = l & 31;
    for (int r = 0; r < 16; ++r) {
        unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            ABf sa, sb;
            memcpy(&sa, slot(row + 32 * k), sizeof(sa));
            memcpy(&sb, slot(col + 32 * k), sizeof(sb));
            acc = fmaf(sa.a, sb.b, acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// f32 16x16x4: A[i][k] in lane i + 16*k; B[k][j] in lane j + 16*k
inline v4f mfma_16x16x4_f32(float a, float b, v4f c) {
    ABf me{a, b};
    unsigned l = lane();
    memcpy(slot(l), &me, sizeof(me));
    wave_sync();
    unsigned col = l & 15;
    for (int r = 0; r < 4; ++r) {
        unsigned row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            ABf sa, sb;
            memcpy(&sa, slot(row + 16 * k), sizeof(sa));
            memcpy(&sb, slot(col + 16 * k), sizeof(sb));
            acc = fmaf(sa.a, sb.b, acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
}  // namespace hipsim
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipsim::mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipsim::mfma_16x16x32_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipsim::mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipsim::mfma_16x16x4_f32(a, b, c)

// ds_read_b64_tr_b16 (semantics measured on gfx950): within each group of 16 lanes let chunk[q] be the
// four b16 at lane q's address; lane i receives result[j] = chunk[4*j + (i >> 2)][i & 3], j = 0..3.
typedef __attribute__((ext_vector_type(4))) short hipsim_v4s;
template <class Lp> inline hipsim_v4s hipsim_ds_read_tr16_b64(Lp p) {
    unsigned l = hipsim::lane();
    uintptr_t a = (uintptr_t)p;
    memcpy(hipsim::slot(l), &a, sizeof(a));
    hipsim::wave_sync();
    hipsim_v4s r;
    const unsigned g = l & ~15u, i = l & 15u;
    for (int j = 0; j < 4; ++j) {
        uintptr_t qa; memcpy(&qa, hipsim::slot(g + 4 * j + (i >> 2)), sizeof(qa));
        short v; memcpy(&v, reinterpret_cast<const char*>(qa) + 2 * (i & 3), 2);
        r[j] = v;
    }
    hipsim::wave_sync();
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipsim_ds_read_tr16_b64(p)
inline unsigned hipsim_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31)); }
#define __builtin_amdgcn_alignbit(hi, lo, sh) hipsim_alignbit(hi, lo, sh)

// v_permlane32_swap_b32 vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src; returns {new vdst, new src}
typedef __attribute__((ext_vector_type(2))) unsigned hipsim_v2u;
inline hipsim_v2u hipsim_permlane32_swap(unsigned vdst, unsigned src) {
    const unsigned l = hipsim::lane();
    const unsigned other_src = hipsim::exchange(src, l ^ 32u);      // what the partner lane holds in src
    const unsigned other_dst = hipsim::exchange(vdst, l ^ 32u);     // what the partner lane holds in vdst
    hipsim_v2u r;
    if (l < 32) { r[0] = vdst; r[1] = other_dst; }
    else { r[0] = other_src; r[1] = src; }
    return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipsim_permlane32_swap(a, b)

// LDS-DMA: every lane copies `size` bytes from its own global pointer to (wave-uniform LDS base + lane*size)
template <class G, class Lp> inline void hipsim_global_load_lds(G g, Lp l, unsigned size, int off, unsigned) {
    memcpy(reinterpret_cast<char*>((uintptr_t)l) + off + hipsim::lane() * size, reinterpret_cast<const void*>((uintptr_t)g), size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipsim_global_load_lds(g, l, size, off, aux)
// csrc/lxo_common.h: the inline-asm LDS-DMA
#define LXO_GLDS16_HIDDEN(gsrc, lds_base, byte_off) hipsim_global_load_lds((const void*)(gsrc), (char*)(lds_base) + (byte_off), 16, 0, 0)
#define LXO_GLDS16_SADDR(voff, sbase, lds_base, m0base, byte_off) hipsim_global_load_lds((const void*)((const char*)(sbase) + (voff)), (char*)(lds_base) + (byte_off), 16, 0, 0)
// the buffer-resource form: base + soff + voff, zeros when soff + voff is out of range
struct lxo_rsrc_t { const char* base; unsigned nbytes; };
inline lxo_rsrc_t lxo_make_rsrc(const void* base, unsigned nbytes) { return lxo_rsrc_t{(const char*)base, nbytes}; }
inline void hipsim_buffer_load_lds16(unsigned voff, lxo_rsrc_t r, unsigned soff, char* dst) {
    const unsigned long long off = (unsigned long long)voff + soff;
    char* d = dst + hipsim::lane() * 16;
    if (off + 16 > r.nbytes) memset(d, 0, 16); else memcpy(d, r.base + off, 16);
}
#define LXO_BLDS16(voff, rsrc, soff, lds_base, m0base, byte_off) hipsim_buffer_load_lds16((voff), (rsrc), (soff), (char*)(lds_base) + (byte_off))

// ---- host API subset used by the C-ABI layer ----
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }      // 2 "CUs": the persistent conv grid (4 workgroups) then walks several tiles
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipsim"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipsim::launch(dim3(grid), dim3(block), [=]() { kern(__VA_ARGS__); })
