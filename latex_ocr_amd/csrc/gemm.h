// Host-visible parameter blocks and launchers of the two MFMA GEMM families.
#pragma once
#include "lxo_common.h"

// C[M,N] = epilogue( A[M,K] * Bp[N,K]^T )   ("NT": both operands K-contiguous)
// A is either a dense row-major matrix (lda) or the implicit im2col view of an
// NHWC tensor for a 3x3 stride-1 convolution (row m = (b,oy,ox), k = (kh,kw,ci)).
struct GemmNT {
    const void* A; const void* Bp; void* C;
    int M, N, K;               // K % 32 == 0
    int lda, ldb, ldc;         // elements; lda, ldb % 8 == 0
    // implicit-im2col geometry (conv != 0)
    int conv, H, W, Cin, Ho, Wo, pad;   // input H x W x Cin, output grid Ho x Wo
    // epilogue: v = alpha*acc + bias[n]; v = act(v); [out_pre = v]; v += addend[m % addend_rows][n];
    //           v *= (relu_ref[m][n] > 0); colsum[n] += v; C = (accumulate ? C : 0) + v
    const float* bias;
    const float* addend; int addend_rows;
    int act;                   // 0 none, 1 relu, 2 tanh
    const void* relu_ref; int ldr;
    void* out_pre;
    int accumulate;
    float* colsum;
    // deterministic mode: colsum_part != null = no float atomics for the column sums -- conv_halo2wg_kernel stores its tile's sums to slot
    // [tile index along M][N] of colsum_part and the launcher adds the slots in order into colsum; kernels without the slot form run
    // without the fused sum and an ordered column-sum pass over C follows (lxo_launch_gemm_nt)
    float* colsum_part; size_t colsum_part_floats;
    float alpha;
    unsigned long long* dbg;   // measurement aid (null = off): conv_halo2wg_kernel stamps its phases here, [workgroup][64] (tools/conv_stamps.py)
    // fused max pool (conv_halo2wg_kernel, plain epilogue only): pool_out[b][ceil(Ho/ph)][ceil(Wo/pw)][N] = max over the ph x pw window of
    // the activated tile, pool_mask (one byte per pooled element) = first-max position (qy * pw + qx) | 4 if the maximum is > 0;
    // C may be null (the full-resolution activation is then never written: nothing else reads it)
    void* pool_out; unsigned char* pool_mask; int pool_h, pool_w;
    int diag;                  // -DLXO_DIAG builds only (LXO_CONV_DIAG, wrong results by design): 1 weights always from slice 0 / tap 0, 2 patch always slice 0, 4 no output stores; ignored by the shipped library
};

// C[I,J] (+)= sum_m A[m,I] * B[m,J]      ("TN": reduction over rows)
// A is dense [M][lda] or the implicit im2col view (I = 9*Cin); output is f32,
// reduced across `nsplit` row ranges with atomics.  Batched over `nbatch`.
struct GemmTN {
    const void* A; const void* B; float* C;
    int M, I, J;
    int lda, ldb, ldc;
    int conv, H, W, Cin, Ho, Wo, pad;
    int nsplit, nbatch;
    long long strideA, strideB, strideC;
    int atomic;                // 1: atomicAdd into C (C pre-initialised); 0: plain store (nsplit must be 1)
    unsigned long long* dbg;   // measurement aid (null = off): conv_wgrad_kernel stamps its pixel blocks here, [workgroup][64] (tools/conv_stamps.py)
    // deterministic mode (bf16 kernels; the f32 kernels always take ONE row range per tile): det_slab != null = no float atomics --
    // conv_wgrad_kernel stores its partial tiles to the slab and an ordered pass adds them; the dense kernels take one row range per tile
    float* det_slab; size_t det_floats;
};

// dt: LXO_F32 / LXO_BF16 = compute type (type of Bp / conv tensors);
// a_f32 / c_f32: the dense A operand / the output is float even when dt == bf16.
// small != 0 selects the 64x64 tile (M <= 64 step GEMMs).
int lxo_launch_gemm_nt(int dt, int a_f32, int c_f32, int small, const GemmNT& p, hipStream_t s);
int lxo_launch_gemm_tn(int dt, int a_f32, int b_f32, const GemmTN& p, hipStream_t s);

// split-K partial products: slab[ks][M][ldc] = A[:, ks*128:(ks+1)*128] * Bp[:, same]^T, ks < K/128 (A float)
int lxo_launch_gemm_slab(int dt, const GemmNT& p, float* slab, long long slab_stride, hipStream_t s);

// bf16 3x3 implicit-GEMM convolution on halo tiles (conv_igemm.hip)
int lxo_launch_conv_igemm(const GemmNT& p, hipStream_t s);

// bf16 dense NT GEMM (plain product) with LDS-DMA staging (gemm_nt_dma.hip); -2 if the call does not qualify
int lxo_launch_gemm_nt_dma(const GemmNT& p, int c_f32, hipStream_t s);

// bf16 dense TN GEMM with row-major LDS tiles and transposing LDS reads (gemm_tn_tr.hip); -2 if the operands do not qualify
int lxo_launch_gemm_tn_tr(const GemmTN& p, hipStream_t s);

// bf16 3x3 weight gradient with tap reuse and transposing LDS reads (conv_wgrad.hip)
int lxo_launch_conv_wgrad(const GemmTN& p, hipStream_t s);
