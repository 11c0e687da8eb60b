// Measurement aid (not part of the library): what does a wave pay, in cycles per v_mfma_f32_32x32x16_bf16, for the non-matrix
// instructions of an implicit-GEMM K loop?  One or two waves per SIMD run 8 independent accumulators back to back; a variant
// inserts LDS fragment reads and / or one VMEM request per 8 MFMAs (the ratio of conv_halo2wg_kernel) in different encodings.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/build/issue_probe && tools/build/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ char lds[];
enum { V_NONE = 0, V_DMA64 = 1, V_DMASADDR = 2, V_GLOAD = 3, V_BUFLDS = 4 };

template <int LDSREADS, int VM>
__global__ __launch_bounds__(256, 2) void probe(const char* __restrict__ g, unsigned long long* out, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 a[8], b = {1, 2, 3, 4};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = u32x4{(unsigned)i, 2, 3, 4};
    const char* gp = g + (size_t)(blockIdx.x * 4 + wave) * 65536 + lane * 16;
    unsigned ldsaddr = (unsigned)(uintptr_t)lds + wave * 8192 + lane * 16;
    u32x4 keep = {0, 0, 0, 0};
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + 40960 + wave * 1024);
    unsigned voff = lane * 16;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (LDSREADS) b = *reinterpret_cast<const u32x4*>(lds + ((ldsaddr + 4096) & 0x7fff));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a[i]), acc[i], 0, 0, 0);
            if (LDSREADS) a[i] = *reinterpret_cast<const u32x4*>(lds + ((ldsaddr + i * 512 + (it & 1) * 32) & 0x7fff));
            if (i == 0) {
                const char* src = gp + (it & 31) * 1024;
                if (VM == V_DMA64) {
                    unsigned k; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(k) : "v"(src), "s"(m0v) : "memory");
                } else if (VM == V_DMASADDR) {
                    const char* sb = g + (size_t)(blockIdx.x * 4 + wave) * 65536 + (it & 31) * 1024;
                    unsigned long long sbu = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sb) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)sb >> 32)) << 32);
                    unsigned k; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(k) : "v"(voff), "s"(sbu), "s"(m0v) : "memory");
                } else if (VM == V_GLOAD) {
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(keep) : "v"(src) : "memory");      // landing register stays live to the end
                }
                if (VM != V_NONE && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.f) sink[0] = s + keep[0];
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int L, int V>
static void run(const char* name, const char* g, unsigned long long* out, float* sink, int grid) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<L, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<L, V>), dim3(grid), dim3(256), 65536, 0, g, out, iters, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s grid %4d (%d wave/SIMD): cycles per MFMA  median %6.1f  min %6.1f  max %6.1f\n", name, grid, grid / 256,
           h[h.size() / 2] / (8.0 * iters), h[0] / (8.0 * iters), h.back() / (8.0 * iters));
}

int main() {
    char* g; unsigned long long* out; float* sink;
    hipMalloc(&g, (size_t)512 * 4 * 65536 + 65536); hipMemset(g, 1, (size_t)512 * 4 * 65536 + 65536);
    hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 64);
    for (int grid : {256, 512}) {
        run<0, V_NONE>("MFMA only", g, out, sink, grid);
        run<1, V_NONE>("+ 9 ds_read_b128 per 8 MFMA", g, out, sink, grid);
        run<0, V_DMA64>("+ 1 global_load_lds (64-bit vaddr) per 8", g, out, sink, grid);
        run<0, V_DMASADDR>("+ 1 global_load_lds (saddr + voff) per 8", g, out, sink, grid);
        run<0, V_GLOAD>("+ 1 global_load_dwordx4 per 8", g, out, sink, grid);
        run<1, V_DMA64>("+ ds_reads + global_load_lds (64-bit)", g, out, sink, grid);
        run<1, V_DMASADDR>("+ ds_reads + global_load_lds (saddr)", g, out, sink, grid);
        run<1, V_GLOAD>("+ ds_reads + global_load_dwordx4", g, out, sink, grid);
    }
    return 0;
}
