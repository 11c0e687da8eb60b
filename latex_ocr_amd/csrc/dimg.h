// Gradient w.r.t. the encoder output, in one batched GEMM (dimg.hip).
#pragma once
#include "lxo_common.h"

// per sample b:  d[r][c] = sum_t alpha[t][b][r] * dctx[t][b][c]            (context path, attention_mechanism.py:69-74)
//                        + sum_e datt[b][r][e] * W[c][e]                    (d_att_img through the img projection, :38-44)
//                        + dmean[b][c] / R                                   (initial-state mean, :84-96)
// out (f32 [B*R][C]) = d, or, when y6 is given: dy6 (bf16) = d * (y6 > 0) and db[c] += column sums (conv6 bias gradient).
struct DimgArgs {
    const float* alpha; long long ld_alpha; int Rp;      // [T][B][Rp]
    const float* dctx; long long ld_dctx; int HC;        // row t of sample b at dctx + t * ld_dctx + b * HC
    const bf16_t* datt;                                   // [B][R][E]
    const bf16_t* W; int ldw;                             // [C][E]
    const float* dmean;                                   // [B][C]
    float* out;
    const bf16_t* y6; bf16_t* dy6; float* db;
    float* db_part; size_t db_part_floats;               // deterministic mode: per-workgroup slots of the bias sums (added in order into db by the launcher)
    int T, B, R, C, E;
};
int lxo_launch_dimg_fused(const DimgArgs& p, hipStream_t st);
