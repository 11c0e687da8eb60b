// Launchers of the non-GEMM decoder / loss / optimizer / decode kernels.
#pragma once
#include "lxo_common.h"
// split-K partial products of gemm_slab_kernel: value(row, col) = sum_{s<n} p[s*stride + row*ld + col]
struct Slabs { const float* p; int n; long long stride; int ld; };
// tf.nn.dropout(x, keep) of attention_cell.py:72,83 with a counter-based mask: element (step t, batch row
// row0 + r, column c) of stream `which` is kept iff hash24(seed, which, (t*rows_total + row0 + r)*width + c) < thr;
// kept values are scaled by inv_keep.  thr == 0 disables (keep_prob 1, decode).
struct Drop { unsigned thr; float inv_keep; unsigned seed; int t, row0, rows_total; };
// f32 parity mode ("det"): reductions whose partial sums would meet in one address by float atomics store the partials to slots of
// this scratch (ws region "det_part") instead and lxo_k_det_reduce adds the slots in ascending order, so that two runs of the
// same step agree bit for bit.  p == nullptr (bf16 mode): atomics.
struct DetScratch { float* p; size_t floats; };
int lxo_k_det_reduce(const float* part, int nslot, long long stride, int N, float* out, hipStream_t st);    // out[n] += sum_q part[q * stride + n]
int lxo_k_rowmean(int dt, const void* img, float* mean, int B, int R, int C, hipStream_t st);
int lxo_k_embed_gather(int dt, const float* table, const float* start, const int* formula, void* out, int B, int T, int D, int Dp, int V, hipStream_t st);
int lxo_k_embed_rows(int dt, const float* table, const float* start, const int* ids, void* out, int n, int D, int Dp, int V, hipStream_t st);
int lxo_k_embed_table(int dt, const float* table, const float* start, void* out, int V, int D, int Dp, hipStream_t st);
int lxo_k_lstm_fwd(const float* z, Slabs zs, const float* c_prev, float* gates, float* c_out, float* h_out, float* ht_out, int ldh, Drop dr, int B, int U, hipStream_t st);
int lxo_k_lstm_bwd(const float* gates, const float* c_prev, const float* c_cur, Slabs s1, Slabs s3, Slabs s4, int off4,
                   float* dcc, float* dz, Drop dr, int carry_rows, int B, int U, hipStream_t st);
// gb (nullable): bf16 copy of g [rows][ldg] (A operand of the fused [d_h~|d_ctx] GEMM)
int lxo_k_tanh_bwd(const float* a, int lda, Slabs carry, const float* o, int ldo, float* g, int ldg, void* gb, int ldgb, Drop dr, int carry_rows, int rows, int cols, hipStream_t st);
int lxo_k_slab_reduce(Slabs sl, float* o, int ldo, int rows, int cols, hipStream_t st);
int lxo_k_tanh_finalize(Slabs sl, float* o, int ldo, Drop dr, int rows, int cols, hipStream_t st);
int lxo_k_attn_fwd(int dt, const void* att_img, const void* img, const float* att_h, Slabs ahs, float* att_h_out, const float* beta, float* alpha, float* part,
                   float* ctx, int ldctx, void* ctxb, int ldcb, int nv, int R, int Rp, int E, int C, int beam, int nch, int rev, hipStream_t st, const void* att_exp = nullptr);   // ctxb (nullable): bf16 copy of ctx, pitch ldcb; att_exp (nullable, bf16 beam decode): e^{2 att_img}
int lxo_k_att_exp(const void* att_img, void* att_exp, long long n, hipStream_t st);       // bf16: att_exp = e^{2 att_img}, n elements (n % 8 == 0)
int lxo_k_attn_bwd(int dt, const void* att_img, const void* att_exp, const void* img, const float* att_h, const float* beta, const float* alpha,
                   Slabs dcs, int dcoff, float* dctx_out, int lddc, const float* ctx, int ldctx, float* de, float* datth,
                   int nv, int R, int Rp, int E, int C, int nch, int rev, hipStream_t st);
int lxo_k_datt_img(int dt, const void* att_img, const float* att_h, const float* beta, const float* de, void* dout, float* dbeta,
                   int T, int B, int R, int Rp, int E, DetScratch det, hipStream_t st);
int lxo_k_add_mean_grad(float* dimg, const float* dmean, int B, int R, int C, hipStream_t st);
// ntok_dev (nullable): device scalar holding the global token count; when set the kernel uses 1 / *ntok_dev instead of inv_ntok
int lxo_k_ce_loss(int dt, const float* logits, const int* formula, const int* lengths, void* dlogits, float* loss_acc, float inv_ntok,
                  const float* ntok_dev, const unsigned* chain_err, int B, int T, int V, int Vp, DetScratch det, hipStream_t st);      // chain_err (nullable): error word of the persistent decoder chain; non-zero poisons the loss (NaN)
int lxo_k_colsum_det(const void* a, int bf16, long long lda, float* out, long long M, int N, DetScratch det, hipStream_t st);      // ordered column sums of an f32 / bf16 matrix (no atomics)
int lxo_k_colsum(const float* a, long long lda, float* out, long long M, int N, DetScratch det, hipStream_t st);
int lxo_k_embed_scatter(const float* demb, const int* formula, float* dtable, float* dstart, int B, int T, int D, int V, int det, hipStream_t st);
int lxo_k_init_bwd(const float* dcc, Slabs dxh, const float* c0, const float* rec0, int ldr, float* dpre, int B, int U, int O, hipStream_t st);
int lxo_k_argmax(const float* logits, int Vp, int V, int n, int id_end, int* ids_step, int* ids_out, int max_steps, int step,
                 int* finished, int* n_unfinished, hipStream_t st);
int lxo_k_beam_step(float* logits, int Vp, int V, int nimg, int k, int id_end, int time, float div_gamma, float div_prob, int div_seed,
                    float* scratch, float* logp, int* finished,
                    int* ids_step, int* parents_step, int* ids_out, int* par_out, int max_steps, int* n_unfinished, hipStream_t st);
int lxo_k_beam_gather(float* rec, int ldr, int XH, float* cs, int U, const int* parents, int k, float* tmp_rec, float* tmp_cs, int n, void* recb, int ldrb, hipStream_t st);      // recb (nullable): bf16 mirror of the re-ordered [o | h] rows
int lxo_k_tile_rows(const float* src, int lds, float* dst, int ldd, int n, int k, int cols, hipStream_t st);
int lxo_k_global_norm_scale(long long n, const float* g, float clip, float* sumsq_tmp, float* out, hipStream_t st);
int lxo_k_chain_guard(const unsigned* err_fwd, const unsigned* err_bwd, const float* probe, float* scale, int have_scale, unsigned* status, hipStream_t st);
int lxo_k_chain_poison(const unsigned* err_fwd, const unsigned* err_bwd, float* probe, hipStream_t st);
int lxo_k_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2, float eps, const float* scale, hipStream_t st);
int lxo_k_simple_opt(float* p, const float* g, float* slot, long long n, float lr, int mode, const float* scale, hipStream_t st);
