// Launchers of the non-GEMM encoder kernels and of the weight-packing kernels.
#pragma once
#include "lxo_common.h"
#include "decoder_kernels.h"
// one weight-pack job: kind 0 transpose (dst[n][coff+k] = src[k][n], k < Kpad zero padded), kind 1 padded row copy
// (K rows, N real / Kpad padded columns), kind 2 conv dgrad layout (K = Cin, N = Cout)
struct PackJob { int kind, K, N, lds, ldd, coff, Kpad, first_block, nblocks; long long src; long long dst; };
struct PackTable { int n; PackJob job[64]; };
int lxo_k_conv1_pool_fwd(int dt, const uint8_t* img, const float* w, const float* b, void* out, int B, int H, int W, hipStream_t s);
// det (f32 parity mode): the workgroups' partial dW / db go to slots of the scratch and are added in order (no float atomics)
int lxo_k_conv1_pool_bwd(int dt, const uint8_t* img, const float* w, const float* b, const void* dout, float* dw, float* db, int B, int H, int W, DetScratch det, hipStream_t s);
int lxo_k_maxpool_fwd(int dt, const void* in, void* out, int B, int H, int W, int C, int ph, int pw, hipStream_t s);
// bf16 mode, layers whose forward ran the fused conv + pool epilogue: routes by the one-byte-per-element mask instead of the activation
int lxo_k_maxpool_mask_bwd(const unsigned char* mask, const void* dp, void* dy, float* db, int B, int H, int W, int C, int ph, int pw, DetScratch det, hipStream_t s);      // det.p: the bias sums through ordered per-workgroup slots
int lxo_k_maxpool_relu_bwd(int dt, const void* y, const void* dp, void* dy, float* db, int B, int H, int W, int C, int ph, int pw, hipStream_t s);
int lxo_k_mask_convert(int dt, const float* d, const void* ref, void* out, float* db, long long rows, int C, hipStream_t s);
int lxo_k_timing_signal(float* pos, int Hp, int Wp, int C, hipStream_t s);
int lxo_k_pack_transpose(int dt, const float* src, void* dst, int K, int N, int lds, int ldd, int coff, int Kpad, hipStream_t s);
int lxo_k_pack_copy(int dt, const float* src, void* dst, int R, int Ccols, int lds, int ldd, int Cpad, hipStream_t s);
int lxo_k_pack_conv_dgrad(int dt, const float* w, void* dst, int Cin, int Cout, hipStream_t s);
int lxo_k_pack_batch(int dt, const PackTable& tab, int total_blocks, const float* prm, void* wpk, hipStream_t s);
// "cnn" encoder: (2,4) stride-2 SAME conv of encoder.py:54-56 as im2col + dense GEMM; its input gradient through the ReLU of conv5
int lxo_k_im2col_s2(int dt, const void* in, void* cols, int B, int H, int W, int Ho, int Wo, int C, hipStream_t s);
int lxo_k_col2im_s2_relu(int dt, const void* dcols, const void* yref, void* dy, float* db, int B, int H, int W, int Ho, int Wo, int C, hipStream_t s);
int lxo_k_colsum_ct(int dt, const void* a, float* out, long long M, int N, hipStream_t s);
