// Parameter block and launcher of the fused recurrent-step GEMMs (rstep.hip).
#pragma once
#include "lxo_common.h"
#include "decoder_kernels.h"

enum RStepEpi {
    RS_PLAIN = 0,     // out[m][n] = acc                                         (att_h = h~ W, [d_h~|d_ctx] = g o_W^T)
    RS_TANH_O = 1,    // out = dropout(tanh(acc))                                (o_t, attention_cell.py:82-83)
    RS_LSTM_FWD = 2,  // z = zx + acc -> gates, c, h, h~                         (TF LSTMCell, attention_cell.py:70-72)
    RS_LSTM_BWD = 3,  // d_h = (dhm + acc) * mask + carry_h -> d_z, d_c          (LSTM backward)
    RS_CARRY = 4,     // acc = d_z K^T: g_{t-1} = (dolog + acc[:, :O]) * tanh' ; carry_h = acc[:, O:]
};

// C[M x N] = A[M x K] * W[N x K]^T, both K-contiguous; the epilogue named by `epi` consumes the tile.
struct RStep {
    const void* A; int lda;        // bf16 mirror or float (converted on load)
    const void* W; int ldw;        // compute dtype, [N][K]
    int M, N, K, epi;
    int U, O;                      // decoder sizes the epilogues index with
    float* out; int ldo;           // PLAIN/TANH_O: result; LSTM_FWD: h (f32 record); LSTM_BWD: d_z [M][4U]; CARRY: g_{t-1} [M][O] (first: raw carries [M][ldo])
    bf16_t* outb; int ldob;        // bf16 mirror of `out` (null in the f32 mode)
    float* out2; bf16_t* out2b;    // LSTM_FWD: h~ (same pitches as out / outb); CARRY: carry_h [M][U]
    // LSTM_FWD
    const float* zx; const float* c_prev; float* gates; float* c_out;
    const int* a_par; int a_k;     // beam decode (nullable): row m of A and of c_prev is row (m / a_k) * a_k + a_par[m] -- the parent hypothesis' state read in place
                                   // (beam_search_decoder_cell.py:176-178) instead of a launch that re-orders the rows first
    const int* zx_idx; int zx_vocab, zx_row;   // decode: row m of the x-part = zx[clamp(zx_idx[m], 0, zx_vocab - 1)] (zx = per-token table), or zx[zx_row] for
                                               // every m when zx_idx is null and zx_row >= 0 (start token); training: zx_idx null, zx_row < 0 -> zx[m]
    // LSTM_BWD (c_prev shared with LSTM_FWD)
    const float* dhm; int lddhm; const float* carry_h; const float* gates_in; const float* c_cur; float* dcc; int carry_rows;
    // PLAIN / TANH_O, off the recurrent loops (initial states and their gradients): v = acc + bias[n] (+ the old out when accumulate)
    const float* bias; int accumulate;
    // CARRY
    const float* dolog; const float* o_prev; int ldoprev; int first;
    Drop dr;
    unsigned long long* dbg;       // measurement aid: per-workgroup phase timestamps [grid.x][8] (null = off; tools/rstep_stamps.py)
};

int lxo_launch_rstep(int dt, int a_bf16, const RStep& p, hipStream_t st);
// measurement aid: the next launches whose epilogue is `epi` stamp their phases into buf (null = off); per host thread
#include "lxo_debug.h"      // lxo_rstep_debug
// rows x cols of an f32 matrix -> bf16 copy (record mirrors of the initial state / after beam re-ordering)
int lxo_k_mirror(const float* src, int lds, void* dst, int ldd, int rows, int cols, hipStream_t st);
