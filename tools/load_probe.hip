// Measurement aid: how fast can the four waves of a workgroup REQUEST and RECEIVE ~100 KB (the operand fetch of a fused recurrent-step
// kernel: 26 one-KB loads per wave, rows of 128 B) in three forms -- global_load_dwordx4 with 64-bit per-lane addresses (what
// rstep.hip does), buffer_load_dwordx4 with a scalar base + 32-bit lane offsets, and buffer_load_dwordx4 ... lds (LDS-DMA)?
// 256 workgroups (one per CU); every workgroup reads the SAME 64 KB "A" region (L2-resident) and its own 32 KB "W" region.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) int rsrc_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int NL = 26;
__device__ __forceinline__ rsrc_t mk(const void* b, unsigned n) {
    const unsigned long long a = (unsigned long long)b; rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a); r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFF)); r[2] = (int)n; r[3] = 0x00020000; return r;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* A, const char* W, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4][NL * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave w: loads 0..17 from A (rows of 128 B: lane -> row lane >> 3, chunk lane & 7), 18..25 from this workgroup's W
    const char* a0 = A + wave * 16384 + (lane >> 3) * 2048 + (lane & 7) * 16;      // row pitch 2 KB
    const char* w0 = W + (size_t)blockIdx.x * 32768 + wave * 8192 + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    u32x4 v[NL];
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 18; ++j) v[j] = *reinterpret_cast<const u32x4*>(a0 + (j >> 1) * 16384 * 0 + (j & 1) * 128 + (j >> 1) * 256);
#pragma unroll
        for (int j = 18; j < NL; ++j) v[j] = *reinterpret_cast<const u32x4*>(w0 + (j - 18) * 1024);
    } else {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 1 << 20, 0x00020000), rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 1 << 30, 0x00020000);
        const unsigned va = (unsigned)(a0 - A), vw = (unsigned)(wave * 8192 + lane * 16);
        const unsigned sw = __builtin_amdgcn_readfirstlane(blockIdx.x * 32768u);
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 18; ++j) v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, va, (j & 1) * 128 + (j >> 1) * 256, 0));
#pragma unroll
            for (int j = 18; j < NL; ++j) v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vw, sw + (j - 18) * 1024, 0));
        } else {
#pragma unroll
            for (int j = 0; j < 18; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(uintptr_t)(lds[wave] + j * 1024), 16, va, (j & 1) * 128 + (j >> 1) * 256, 0, 0);
#pragma unroll
            for (int j = 18; j < NL; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(uintptr_t)(lds[wave] + j * 1024), 16, vw, sw + (j - 18) * 1024, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    unsigned acc = 0;
    if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc += v[j][0] ^ v[j][3];
    } else acc = *reinterpret_cast<unsigned*>(lds[wave] + lane * 4);
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}
template <int MODE> void run(const char* name, const char* A, const char* W, unsigned long long* out, unsigned* sink) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, A, W, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 4 * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b;
    for (size_t i = 0; i < h.size(); i += 2) { a.push_back(h[i]); b.push_back(h[i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-44s requests issued: median %5llu  p90 %5llu | all data back: median %5llu  p90 %5llu cycles\n", name, a[a.size() / 2], a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() * 9 / 10]);
}
int main() {
    char *A, *W; unsigned long long* out; unsigned* sink;
    hipMalloc(&A, 1 << 20); hipMalloc(&W, 256 * 32768); hipMalloc(&out, 256 * 4 * 2 * 8); hipMalloc(&sink, 4);
    hipMemset(A, 1, 1 << 20); hipMemset(W, 2, 256 * 32768);
    run<0>("global_load_dwordx4, 64-bit lane address", A, W, out, sink);
    run<1>("buffer_load_dwordx4, scalar base + offset", A, W, out, sink);
    run<2>("buffer_load_dwordx4 ... lds (LDS-DMA)", A, W, out, sink);
    return 0;
}
