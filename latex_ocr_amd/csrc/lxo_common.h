// Device-side common definitions for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere; MFMA lane layouts per the CDNA4 ISA:
//   32x32x16 bf16 : A[i=l&31][k=(l>>5)*8+e]  B[k=(l>>5)*8+e][j=l&31]
//   32x32x2  f32  : A[i=l&31][k=l>>5]        B[k=l>>5][j=l&31]
//   C/D (both)    : col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

#define LXO_DEV __device__ __forceinline__

LXO_DEV float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
// round-to-nearest-even; NaN stays NaN (quiet)
// (v_cvt_pk_bf16_f32 on gfx950)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
LXO_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
LXO_DEV unsigned pack_bf2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <class T> struct is_bf16 { static constexpr bool value = false; };
template <> struct is_bf16<bf16_t> { static constexpr bool value = true; };

LXO_DEV float to_f32(float v) { return v; }
LXO_DEV float to_f32(bf16_t v) { return bf2f(v); }
template <class T> LXO_DEV T from_f32(float v);
template <> LXO_DEV float from_f32<float>(float v) { return v; }
template <> LXO_DEV bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }

// 8 consecutive elements -> float[8] (pointer must be 16-byte aligned for bf16,
// 16-byte aligned for float)
LXO_DEV void load8(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
LXO_DEV void load8(const bf16_t* p, float (&v)[8]) {
    u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(a[i] << 16);
        v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u);
    }
}
LXO_DEV void store8(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
LXO_DEV void store8(bf16_t* p, const float (&v)[8]) {
    u32x4 a = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    *reinterpret_cast<u32x4*>(p) = a;
}

LXO_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
LXO_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
LXO_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS-DMA that hipcc's s_waitcnt bookkeeping does not see: global -> LDS, 16 bytes per lane, destination = wave-uniform LDS
// byte address (lds_base + byte_off) + 16 * lane.  With the builtin form hipcc places `s_waitcnt vmcnt(0)` in front of the next
// ds_read_b64_tr_b16 (it cannot tell the two LDS stages apart), which drains the prefetch of the NEXT pixel block before the
// current one is computed: the double buffering never overlapped anything.  The caller counts completion itself
// (s_waitcnt vmcnt(N), then a barrier, then the reads).  M0 is compiler-reserved: saved, set and restored inside the statement
// (cdna_hip_programming.md section 5.7).  tests/hipsim pre-defines the macro with its interpreter hook.
#ifndef LXO_GLDS16_HIDDEN
#define LXO_GLDS16_HIDDEN(gsrc, lds_base, byte_off) do { unsigned keep_m0_; \
    const unsigned dst_m0_ = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_base) + (unsigned)(byte_off)); \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_m0_) : "v"(gsrc), "s"(dst_m0_) : "memory"); } while (0)
#endif

// The same with a SCALAR base and a 32-bit per-lane byte offset (`global_load_lds_dwordx4 voff, s[base]`): when the base is wave-uniform
// (a weight tile) the request costs no VALU address arithmetic at all.  m0base = readfirstlane of the LDS array's address (taken once);
// (lds_base, byte_off) name the same destination for the tests/hipsim build.
#ifndef LXO_GLDS16_SADDR
#define LXO_GLDS16_SADDR(voff, sbase, lds_base, m0base, byte_off) do { unsigned keep_m0_; \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_m0_) : "v"(voff), "s"(sbase), "s"((m0base) + (unsigned)(byte_off)) : "memory"); } while (0)
#endif

// LDS-DMA through a BUFFER RESOURCE (`buffer_load_dwordx4 voff, s[rsrc], soff offen lds`): the address is base + soff (scalar) + voff
// (32 bits per lane), and a lane whose soff + voff lies at or beyond num_records writes ZEROS into its LDS slot (measured:
// tools/blds_probe.hip) -- zero padding costs one v_cndmask of the offset instead of a 64-bit select against a zero line.  The
// request is also cheaper to issue than the 64-bit-vaddr form (tools/issue_probe.hip: ~29 vs ~56 cycles of the matrix pipe).
#define LXO_BLDS_OOB 0x80000000u          // an offset that is out of range of every buffer below 2 GB
#ifndef LXO_BLDS16
typedef __attribute__((ext_vector_type(4))) int lxo_rsrc_t;
static __device__ __forceinline__ lxo_rsrc_t lxo_make_rsrc(const void* base, unsigned nbytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    lxo_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));   // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)nbytes);
    r[3] = 0x00020000;
    return r;
}
#define LXO_BLDS16(voff, rsrc, soff, lds_base, m0base, byte_off) do { unsigned keep_m0_; \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_m0_) : "v"(voff), "s"(rsrc), "s"(soff), "s"((m0base) + (unsigned)(byte_off)) : "memory"); } while (0)
#endif

// dtype codes of the C ABI (include/lxo.h)
#ifndef LXO_F32
#define LXO_F32 0
#define LXO_BF16 1
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
