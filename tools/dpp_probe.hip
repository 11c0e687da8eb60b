// Wave-wide sum without the LDS crossbar: DPP quad_perm / row_half_mirror / row_mirror for the four steps inside a row of 16 lanes,
// v_permlane16_swap / v_permlane32_swap (gfx950) across rows.  Checks the result against the __shfl_xor butterfly (ds_bpermute) and
// times both (dependent chains of 64 reductions per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ float dpp_f(float v, int) { return v; }
template <int CTRL> __device__ __forceinline__ float dppx(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dppx<0xB1>(v);                  // quad_perm [1,0,3,2]
    v += dppx<0x4E>(v);                  // quad_perm [2,3,0,1]
    v += dppx<0x141>(v);                 // row_half_mirror
    v += dppx<0x140>(v);                 // row_mirror
    { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ void check(const float* in, float* out_a, float* out_b) {
    const float v = in[blockIdx.x * 64 + threadIdx.x];
    out_a[blockIdx.x * 64 + threadIdx.x] = wave_sum_dpp(v);
    out_b[blockIdx.x * 64 + threadIdx.x] = wave_sum_shfl(v);
}
template <int MODE> __global__ void timeit(float* out, int n) {
    float v = threadIdx.x * 0.001f, acc = 0.f;
    for (int i = 0; i < n; ++i) { const float s = MODE ? wave_sum_dpp(v) : wave_sum_shfl(v); acc += s; v = v * 0.5f + s * 1e-9f; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const int nb = 64;
    float *in, *a, *b; hipMalloc(&in, nb * 64 * 4); hipMalloc(&a, nb * 64 * 4); hipMalloc(&b, nb * 64 * 4);
    float* h = (float*)malloc(nb * 64 * 4); for (int i = 0; i < nb * 64; ++i) h[i] = (float)(rand() % 2001 - 1000) / 64.f;
    hipMemcpy(in, h, nb * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(nb), dim3(64), 0, 0, in, a, b);
    float* ha = (float*)malloc(nb * 64 * 4); float* hb = (float*)malloc(nb * 64 * 4);
    hipMemcpy(ha, a, nb * 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, nb * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int w = 0; w < nb; ++w) { double ref = 0; for (int l = 0; l < 64; ++l) ref += h[w * 64 + l];
        for (int l = 0; l < 64; ++l) { double d = fabs(ha[w * 64 + l] - ref); if (d > worst) worst = d; if (d > 1e-3 * (1 + fabs(ref))) ++bad; if (ha[w*64+l] != ha[w*64]) ++bad; } }
    printf("dpp wave sum: %d bad of %d (every lane must hold the full sum), worst abs diff to the f64 sum %.2e\n", bad, nb * 64, worst);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode) hipLaunchKernelGGL(timeit<1>, dim3(256), dim3(512), 0, 0, a, 4096); else hipLaunchKernelGGL(timeit<0>, dim3(256), dim3(512), 0, 0, a, 4096);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f ns per dependent wave reduction (8 waves per CU)\n", mode ? "dpp + permlane swap" : "shfl_xor (ds_bpermute)", ms * 1e6 / 4096);
    }
    return bad != 0;
}
