// Dense C[M][N] = A[M][K] * B[N][K]^T for bf16 operands (both K-contiguous), gfx950: LDS-DMA staging.
//
// gemm_nt_kernel (gemm.hip) stages its tiles global -> registers -> LDS and tops out at 240-290 TFLOP/s on the decoder's
// batched projections (att_img = img W, logits, d_o).  Here the 128-row x 64-k tiles of both operands go straight to LDS by
// LDS-DMA, two stages of 32 KB, the DMA of K-step k+1 in flight under the MFMAs of K-step k; a tile row is 128 bytes = eight
// 16-byte chunks, stored XOR-swizzled by ((row >> 1) & 7) so that the 16-byte fragment reads of 32 consecutive rows spread
// over all banks (the weight-stage layout of conv_halo2wg_kernel).  128 x 128 output tile, 4 waves as 2 x 2, each
// 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 tiles.  Product + optional per-column bias only: the fused epilogues (activation,
// addend, masks) stay with gemm_nt_kernel.
#include "gemm.h"
#include "api_util.h"

namespace {

constexpr int NTILE = 128 * 128;                     // bytes of one operand tile: 128 rows x 64 k x 2
constexpr int NSTAGE = 2 * NTILE;

}  // namespace

HIP_DYNAMIC_SHARED(char, lxo_ntdma_lds)

namespace {

template <typename OT>
__global__ __launch_bounds__(256) void gemm_nt_dma_kernel(GemmNT p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(p.Bp);

    // DMA geometry: thread -> (row (tid >> 3) + 32 j, LDS slot tid & 7); slot s of row r holds the GLOBAL chunk s ^ ((r >> 1) & 7).
    // Buffer resources: the per-lane byte offset (row, chunk) is fixed, the K-step is the request's scalar offset, rows beyond M / N
    // carry an out-of-range offset and read as zeros -- no address arithmetic per request (see conv_wgrad.hip).
    const unsigned m0b = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lxo_ntdma_lds);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const lxo_rsrc_t ra = lxo_make_rsrc(A, (unsigned)((long long)p.M * p.lda * 2));
    const lxo_rsrc_t rb = lxo_make_rsrc(B, (unsigned)((long long)p.N * p.ldb * 2));
    unsigned voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (tid >> 3) + 32 * j, gch = ((tid & 7) ^ ((r >> 1) & 7)) << 3;
        voa[j] = m0 + r < p.M ? (unsigned)(((m0 + r) * p.lda + gch) * 2) : LXO_BLDS_OOB;
        vob[j] = n0 + r < p.N ? (unsigned)(((n0 + r) * p.ldb + gch) * 2) : LXO_BLDS_OOB;
    }
    auto issue = [&](int k0, int stage) {
        const unsigned so = (unsigned)(k0 * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            LXO_BLDS16(voa[j], ra, so, lxo_ntdma_lds, m0b, stage * NSTAGE + wave_u * 1024 + 4096 * j);
            LXO_BLDS16(vob[j], rb, so, lxo_ntdma_lds, m0b, stage * NSTAGE + NTILE + wave_u * 1024 + 4096 * j);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // fragment reads: row (lane & 31) of a 32-row sub-tile, 16-byte chunk kc = 2 ks + (lane >> 5) of the 64-k row
    const int khalf = lane >> 5;
    int rowa[2], rowb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { rowa[t] = wm * 64 + t * 32 + (lane & 31); rowb[t] = wn * 64 + t * 32 + (lane & 31); }

    issue(0, 0);
    for (int k0 = 0, it = 0; k0 < p.K; k0 += 64, ++it) {        // K % 64 == 0 (checked by the launcher)
        const int stage = it & 1;
        __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) + lgkmcnt(0): every ds_read of the stage the NEXT DMA overwrites has retired before any wave passes the barrier (gfx950 barriers carry no implicit wait); only this K-step's DMA is outstanding here
        __builtin_amdgcn_s_barrier();
        if (k0 + 64 < p.K) issue(k0 + 64, stage ^ 1);
        const char* sa = lxo_ntdma_lds + stage * NSTAGE;
        const char* sb = sa + NTILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = ks * 2 + khalf;
            u32x4 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const u32x4*>(sa + rowa[t] * 128 + ((kc ^ ((rowa[t] >> 1) & 7)) << 4));
                bf[t] = *reinterpret_cast<const u32x4*>(sb + rowb[t] * 128 + ((kc ^ ((rowb[t] >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bf[b]), acc[a][b], 0, 0, 0);
        }
    }
    // acc[a][b][e]: row m = m0 + wm*64 + a*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column n = n0 + wn*64 + b*32 + (lane & 31)
    OT* __restrict__ C = reinterpret_cast<OT*>(p.C);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + (lane & 31);
        const float bias = p.bias ? p.bias[n < p.N ? n : 0] : 0.f;      // per output column (the LSTM bias of the x-part)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) C[(long long)m * p.ldc + n] = from_f32<OT>(acc[a][b][e] + bias);
            }
    }
}

}  // namespace

// product (+ bias), bf16 A and B (16-byte aligned, pitches multiples of 8), K % 64 == 0; c_f32 selects the output type
int lxo_launch_gemm_nt_dma(const GemmNT& p, int c_f32, hipStream_t s) {
    if (p.conv || p.addend || p.relu_ref || p.out_pre || p.colsum || p.accumulate || p.act || p.alpha != 1.f) return -2;
    if (p.K % 64 || p.lda % 8 || p.ldb % 8 || (((uintptr_t)p.A | (uintptr_t)p.Bp) & 15)) return -2;
    if ((long long)p.M * p.lda * 2 >= (1LL << 31) || (long long)p.N * p.ldb * 2 >= (1LL << 31)) return -2;   // 32-bit buffer offsets
    {   // per device, not per process (see conv_igemm.hip attr_needed)
        static bool done[64] = {};
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (!known || !done[dev]) {
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_dma_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NSTAGE));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_dma_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NSTAGE));
            if (known) done[dev] = true;
        }
    }
    dim3 grid((p.N + 127) / 128, (p.M + 127) / 128);
    if (c_f32) hipLaunchKernelGGL((gemm_nt_dma_kernel<float>), grid, dim3(256), 2 * NSTAGE, s, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<bf16_t>), grid, dim3(256), 2 * NSTAGE, s, p);
    return (int)hipGetLastError();
}
