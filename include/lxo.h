/* C ABI of the MI355X-native image->LaTeX hot path (liblxo.so).
 *
 * The reference (LinXueyuanStdio/LaTeX_OCR) has no FFI: its numeric boundary is
 * `sess.run` inside model/img2seq.py (:169, :236, :263).  These entry points
 * are what a binding for that boundary calls; each cites the reference code it
 * replaces.  All pointers are DEVICE pointers owned by the caller, all calls
 * are asynchronous on the given hipStream_t (passed as void*), none allocates.
 * Return value: 0 = OK, negative = error (see lxo_last_error()).
 */
#ifndef LXO_H
#define LXO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* liblxo.so is built with -fvisibility=hidden: what this header (and include/lxo_debug.h, the measurement hooks) declares is the
 * library's whole dynamic symbol table (tests/test_abi.py compares `nm -D` with the two headers) */
#pragma GCC visibility push(default)

#define LXO_F32 0
#define LXO_BF16 1

const char* lxo_last_error(void);
/* ABI version of this header.  Bumped whenever lxo_shape grows or an entry point changes (round 4: 4 -- lxo_comm_*,
 * lxo_allreduce_bucket, LXO_GNORM_FLOATS behind lxo_global_norm_scale's scale_out; round 5: 5 -- lxo_shape.deterministic,
 * lxo_chain_guard, lxo_decode_state_get / _set, lxo_decode_cell_step; a NaN *scale_dev drops an optimizer step; round 6: 6 --
 * lxo_beam_decode_attn, lxo_shape.live_B; the lxo_decoder_train_*_active calls are gone: they ran the launch-per-step kernels and were slower than the persistent chains computing every padded step).  A binding must check
 * lxo_version() == LXO_ABI_VERSION and lxo_shape_size() == sizeof(its own lxo_shape) before the first call: a caller built
 * against an older header passes a shorter struct and the library would read past its end. */
#define LXO_ABI_VERSION 6
int lxo_version(void);
int lxo_shape_size(void);

/* ---- building blocks (exposed for parity tests and for bindings that want
 *      the encoder / projections alone) ---- */

/* C[M,N] = act(alpha * A[M,K] * Bp[N,K]^T + bias) ; Bp is the K-contiguous
 * ("transposed", TF [in,out] -> [out,in]) weight copy in the compute dtype.
 * dt: compute dtype; a_f32/c_f32: A / C are float even when dt == LXO_BF16.
 * Replaces tf.layers.dense / tf.matmul call sites (attention_mechanism.py:43,79;
 * attention_cell.py:82-84). */
int lxo_gemm_nt(int dt, int a_f32, int c_f32, int small, const void* A, const void* Bp, void* C,
                int M, int N, int K, int lda, int ldb, int ldc, const float* bias, int act,
                float alpha, int accumulate, void* stream);

/* C[I,J] += sum_m A[m,I] * B[m,J] (f32 output, atomics across nsplit row ranges). */
int lxo_gemm_tn(int dt, int a_f32, int b_f32, const void* A, const void* B, float* C,
                int M, int I, int J, int lda, int ldb, int ldc, int nsplit, int atomic, void* stream);

/* split-K partial products of a skinny GEMM (the recurrent steps): slab[ks][M][ldc] = A[:, ks*128:+128] *
 * Bp[:, same]^T for ks < K/128, A float [M][lda], Bp compute dtype [N][ldb]; the consumer adds the slabs. */
int lxo_gemm_slab(int dt, const void* A, const void* Bp, float* slab, int M, int N, int K, int lda, int ldb, int ldc,
                  long long slab_stride, void* stream);

/* 3x3 stride-1 convolution as an implicit GEMM on the MFMA units (the tf.layers.conv2d call
 * sites model/encoder.py:37-59): out[B,Ho,Wo,Cout] = act(conv(in[B,H,W,Cin]) + bias), NHWC,
 * in/out/wpk in the compute dtype, wpk = [Cout][9*Cin] (tap-major, channel-minor).
 * pad 1 + Ho=H,Wo=W is SAME; pad 0 + Ho=H-2,Wo=W-2 is VALID; pad 2 + Ho=H+2,Wo=W+2 is the
 * dgrad of a VALID layer.  Cin % 64 == 0 (bf16) / % 32 (f32). */
int lxo_conv3x3(int dt, const void* in, const void* wpk, const float* bias, void* out, int B, int H, int W,
                int Cin, int Ho, int Wo, int Cout, int pad, int relu, void* stream);

/* lxo_conv3x3 with the rest of the fused epilogue the encoder uses: out_pre (optional) receives
 * act(conv + bias) before `addend` (f32 [addend_rows][Cout], row = pixel index % addend_rows: the timing signal of
 * positional.py:10-65 fused into conv6) is added; relu_ref (optional, same shape as out) zeroes out where the
 * reference activation is <= 0 (the ReLU backward of the producing layer when this call is a dgrad) and
 * colsum (optional, f32 [Cout], accumulated) receives the column sums of the result (that layer's bias gradient). */
int lxo_conv3x3_ex(int dt, const void* in, const void* wpk, const float* bias, void* out, int B, int H, int W,
                   int Cin, int Ho, int Wo, int Cout, int pad, int relu, const float* addend, int addend_rows,
                   void* out_pre, const void* relu_ref, float* colsum, void* stream);

/* weight gradient of lxo_conv3x3: dw[9*Cin][Cout] (f32, HWIO flattened) += sum_pixels in (x) dout.
 * What TF autodiff derives for the tf.layers.conv2d kernels (model/img2seq.py:119-123). */
int lxo_conv3x3_wgrad(int dt, const void* in, const void* dout, float* dw, int B, int H, int W,
                      int Cin, int Ho, int Wo, int Cout, int pad, void* stream);

/* AttentionMechanism.context (model/components/attention_mechanism.py:46-94) for nv decoder rows:
 * alpha = softmax_r(sum_k beta_k tanh(att_img[r,k] + att_h[k])), ctx = sum_r alpha_r img[r,:].
 * att_img [nimg,R,E], img [nimg,R,C] compute dtype; row v uses image v / beam; alpha f32 [nv,Rp],
 * Rp = (R+7)/8*8; ctx f32 [nv, ldctx]; part = scratch of nv*32*(C+2) floats. */
int lxo_attention_fwd(int dt, const void* att_img, const void* img, const float* att_h, const float* beta,
                      float* alpha, float* part, float* ctx, int ldctx, int nv, int R, int E, int C, int beam,
                      void* stream);

/* Measurement aid (off by default, per host thread): with timing enabled every implicit-GEMM conv launch (forward,
 * data gradient, weight gradient) and every attention launch of the lxo_encoder_* / lxo_decoder_train_* calls is
 * bracketed by HIP events on the stream it is launched on, so that bench.py's roofline comes from kernel durations
 * INSIDE a real training step.  lxo_timing_enable(1) clears the records; lxo_timing_get synchronises record i and returns
 * its family ("conv_fwd", "conv_dgrad", "conv_wgrad", "attn_fwd", "attn_bwd"), name, algorithmic work (FLOPs or bytes)
 * and duration in milliseconds. */
int lxo_timing_enable(int on);
int lxo_timing_count(void);
int lxo_timing_get(int i, const char** family, const char** name, double* work, float* ms);

/* ---- the hot path proper ---- */

/* Shape of one call.  H, W are the batch-max image extents after white padding
 * (model/utils/image.py:27-44); T is the padded formula width Lmax+1
 * (model/utils/text.py:157-162); V is Vocab.n_tok (text.py:16); C/E/U/O/D are the
 * encoder width (512) and attn_cell_config {dim_e, num_units, dim_o, dim_embeddings}
 * (configs/model.json:6-12). */
typedef struct lxo_shape {
    int B, H, W, T, V;
    int C, E, U, O, D;
    int dtype;      /* LXO_F32 (parity mode) or LXO_BF16 (bf16 storage, f32 accumulate) */
    int beam;       /* decode only: beam width the workspace is sized for (>= 1) */
    int max_steps;  /* decode only: step capacity (max_length_formula + 2) */
    /* training only: tf.nn.dropout keep probability on h and o (attention_cell.py:72,83; config.dropout,
     * fed at img2seq.py:166) and the seed of this step's counter-based masks; 0 or >= 1 disables */
    float keep_prob;
    int dropout_seed;
    /* beam decode only: add_div_penalty (beam_search_decoder_cell.py:258-287; configs/model.json:15-16).
     * div_gamma 0 or 1, or div_prob 0, disables; div_seed keys the Bernoulli(div_prob) draws */
    float div_gamma, div_prob;
    int div_seed;
    /* encoder variants of configs/model.json: encoder_cnn 0 = "vanilla" (pools after conv4 / conv5,
     * encoder.py:46-52), 1 = "cnn" (no pools, a (2,4) stride-2 SAME conv after conv5, encoder.py:54-56);
     * no_positional != 0 = positional_embeddings false (encoder.py:60-65 skipped) */
    int encoder_cnn;
    int no_positional;
    /* decoder step decomposition: 0 (default) = automatic: the teacher-forced recurrence of lxo_decoder_train_fwd as ONE persistent launch
     * of 8 XCD-local chains (csrc/xdec.hip) where the shape qualifies (bf16, the shipped widths U = O = C = 512, E = 256, B in
     * {8, 16, 32, 64}, an MI355X), else the fused step kernels; 2 = always the fused step kernels (full-K workgroups with the cell's
     * point-wise stages in the GEMM epilogues, csrc/rstep.hip: 9 dependent launches per training step pair); 1 = the split-K slab
     * GEMMs + separate point-wise kernels of round 1 (13 launches; also what the side-stream interleave uses) */
    int step_kernels;
    /* optional row encoder between the CNN and the decoder (north_star's "row-BiLSTM encoder"; ABSENT from the reference, whose
     * encoder.py:4 imports GRUCell / LSTMCell and never uses them -- off by default, outside the parity contract): 1 = every
     * row of the H' x W' feature map is run through a bidirectional TF-style LSTMCell (C/2 units per direction, zero initial
     * state) along W'; the concatenated outputs replace the features the attention reads.  Needs C in {256, 512}. */
    int encoder_rnn;
    /* bf16 mode only (the f32 parity mode is always run-to-run reproducible): != 0 = every reduction of the training step runs in a
     * fixed order -- the float atomics of the bf16 epilogues (conv weight gradients, the deferred all-step weight gradients, bias
     * sums, d_beta, the embedding scatter, the loss statistics) go through ordered partial slots (ws region "det_part") -- so that
     * two runs of one binary on the same inputs give bit-identical losses, gradients and weights.  Costs a few per cent of a step
     * (bench.py: secondary.deterministic_bf16); off by default.  SURVEY.md Appendix D step 8. */
    int deterministic;
    /* training only: 0 (or B) = every batch row is a sample.  0 < live_B < B: rows live_B .. B - 1 are DEAD PADDING ROWS -- a batch the
     * persistent chains do not take (the reference trains at 3, buckets and evaluates at 20: configs/training.json:6, data_generator.py:41,
     * evaluate_txt.py:42) filled up to a chain batch of 8 / 16 / 32 / 64.  The caller gives them any valid token ids and formula
     * length 0 (so no token of theirs is inside the loss mask, img2seq.py:68-71: their d(logits) and every gradient contribution is an
     * exact zero, n_words does not see them); `img` needs only live_B images.  The encoder computes the live images only (the dead
     * rows' features are zeros), the decoder steps all B rows.  Ignored with encoder_rnn. */
    int live_B;
} lxo_shape;

/* flat f32 parameter / gradient / Adam-slot buffers: variable inventory in TF
 * checkpoint order.  id in [0, lxo_param_num()). */
int lxo_param_num(void);
const char* lxo_param_name(int id);
/* the TF variable name of slot id for this shape's encoder variant (the tf.layers.conv2d scopes are numbered in
 * creation order, so "cnn" renames the last two convs); slots with count 0 in lxo_param_info are absent */
const char* lxo_param_name_for(const lxo_shape* s, int id);
long long lxo_param_total(const lxo_shape* s);
int lxo_param_info(const lxo_shape* s, int id, long long* offset, long long* count);

size_t lxo_wpack_bytes(const lxo_shape* s);
size_t lxo_workspace_bytes(const lxo_shape* s);
/* offset/size of a named workspace region (tests and bindings read activations,
 * logits, loss statistics and attention weights through this) */
int lxo_ws_region(const lxo_shape* s, const char* name, size_t* offset, size_t* bytes);
/* element type of a workspace region for this shape: LXO_F32, LXO_BF16, LXO_I32 or LXO_U8 (negative: unknown name).
 * Matters for "d_img", the hand-over between lxo_decoder_train_bwd and lxo_encoder_bwd: f32 mode = the f32 gradient w.r.t. the
 * encoder output; bf16 mode = d_y6, that gradient with conv6's ReLU mask already applied, stored as bf16 (and conv6's bias
 * gradient already accumulated) -- a caller that drives lxo_encoder_bwd on its own must write the region in THAT form. */
#define LXO_I32 2
#define LXO_U8 3
int lxo_ws_region_dtype(const lxo_shape* s, const char* name);

/* refresh the compute-dtype GEMM operand copies from the f32 master parameters
 * (call after every optimizer step / checkpoint load) */
int lxo_pack_weights(const lxo_shape* s, const float* params, void* wpack, void* stream);

/* Encoder.__call__ (model/encoder.py:17-68) + add_timing_signal_nd
 * (model/components/positional.py:10-65): u8 [B,H,W,1] -> ws region "img" [B,R,C] */
int lxo_encoder_fwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                    const uint8_t* img, void* stream);
/* backward of conv layers last_layer..first_layer (6..1), accumulating into grads;
 * layer 6 consumes ws region "d_img" (f32 mode: the f32 gradient w.r.t. the encoder output; bf16 mode: d_y6, i.e. that
 * gradient with conv6's ReLU mask applied, in bf16 -- the decoder's last GEMM applies the mask and sums conv6's bias
 * gradient in its epilogue, see lxo_decoder_train_bwd).  Split so a data-parallel caller can
 * all-reduce finished buckets while earlier layers still run. */
int lxo_encoder_bwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                    const uint8_t* img, float* grads, int last_layer, int first_layer, void* stream);
/* The same range in ONE call, for a data-parallel caller that does not want the compute stream to stop at every bucket:
 * ready_events[l] (l = 1..6; a table of 7 hipEvent_t created by the caller, NULL entries skipped) is recorded as soon as layer l's
 * weight and bias gradients are final -- on the weight-gradient side stream when one is bound (lxo_set_encoder_side_stream), else on
 * `stream` -- so a communication stream can wait for a layer's event and reduce its bucket while the earlier layers, and the
 * weight gradients beside them, still run.  Returns like lxo_encoder_bwd: `stream` ordered after the whole range. */
int lxo_encoder_bwd_ready(const lxo_shape* s, const float* params, const void* wpack, void* ws, const uint8_t* img, float* grads,
                          int last_layer, int first_layer, void* const* ready_events, void* stream);
/* The whole backward pass of a training step in one call: lxo_decoder_train_bwd followed by lxo_encoder_bwd for layers 6..1 (the two
 * halves of what one sess.run(train_op) differentiates, img2seq.py:169), with the weight-gradient side stream -- when one is bound --
 * joined ONCE, at the end, instead of once per call: the decoder's deferred weight gradients then also run beside conv6's data
 * gradient.  ready_events (optional): as for lxo_encoder_bwd_ready, plus entry 0 = every decoder gradient final (incl. y_W_o and the
 * chain-failure probe element lxo_chain_guard reads).  `stream` is ordered after everything when the call returns. */
int lxo_train_bwd(const lxo_shape* s, const float* params, const void* wpack, void* ws, const int32_t* formula, const uint8_t* img,
                  float* grads, void* const* ready_events, void* stream);

/* Optional second HIP stream for the calling host thread (NULL disables).  When set, the recurrent
 * loops of lxo_decoder_train_fwd / _bwd run the two halves of the batch on `stream` and on this side
 * stream (fork/join by events), so the launch- and latency-bound step kernels of one half overlap the
 * other half's.  The only per-thread state the library keeps. */
int lxo_set_side_stream(void* stream);
/* Optional second HIP stream for the weight gradients of the calling host thread (NULL disables; bf16 mode only,
 * ignored while lxo_timing_enable(1) records).  lxo_encoder_bwd runs the conv weight-gradient
 * kernels on it, beside the data-gradient / pool-backward chain of the main stream (three rotating gradient buffers);
 * lxo_decoder_train_bwd runs its deferred all-step weight gradients (dense dW GEMMs, LSTM bias, embeddings, initial-state
 * parameters, dW_att_img) on it, beside the d_att_img -> d_img path.  Fork / join by events inside each call: every call
 * returns with `stream` ordered after all of its work on the side stream, so gradients are final in stream order as without
 * it, and no kernel of the side stream is in flight when a later call launches a persistent chain. */
int lxo_set_encoder_side_stream(void* stream);

/* Decoder.__call__ training branch (model/decoder.py:41-57): AttentionMechanism
 * set-up (attention_mechanism.py:19-43,124-153), T steps of AttentionCell.step
 * (attention_cell.py:58-89) under teacher forcing, logits for every step. */
int lxo_decoder_train_fwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                          const int32_t* formula, void* stream);
/* loss of model/img2seq.py:68-75 and d(loss)/d(logits).  inv_ntok = 1 / (number of
 * unmasked tokens in the GLOBAL batch).  ws region "loss" = {sum CE, token count}. */
int lxo_ce_loss_fwd_bwd(const lxo_shape* s, void* ws, const int32_t* formula,
                        const int32_t* lengths, float inv_ntok, void* stream);
/* The same with the GLOBAL token count read from device memory (*ntok_dev, a float): under data parallelism the count is
 * the sum over ranks of host-known integers, all-reduced on a side stream while the forward runs, so no host
 * synchronisation sits between forward and backward (img2seq.py:69-71 takes the mean over all tokens of the batch). */
int lxo_ce_loss_fwd_bwd_dev(const lxo_shape* s, void* ws, const int32_t* formula,
                            const int32_t* lengths, const float* ntok_dev, void* stream);
/* BPTT through the decoder (what TF autodiff does for img2seq.py:119-123);
 * accumulates decoder gradients into grads and leaves d(enc) in ws region "d_img" (bf16 mode: already masked by conv6's
 * ReLU and converted, with conv6's bias gradient added to grads; f32 mode: the plain f32 gradient). */
int lxo_decoder_train_bwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                          const int32_t* formula, float* grads, void* stream);
/* The same in two parts so that a data-parallel caller can start reducing gradients before the recurrence has run:
 * parts bit 0 = d_o from the logits for every step + the y_W_o gradient (final after this part);
 * parts bit 1 = BPTT, every other decoder gradient and d(enc). */
int lxo_decoder_train_bwd_part(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                               const int32_t* formula, float* grads, int parts, void* stream);

/* Health of the persistent decoder chains (lxo_shape.step_kernels == 0, bf16: lxo_decoder_train_fwd / _bwd run the recurrence as ONE
 * launch of 8 XCD-local chains that rely on the hardware placing 32 workgroups of the grid on every XCD).  A chain that does not
 * assemble within 200 ms gives up (no hang) and leaves a non-zero ERROR WORD; the call's outputs are then invalid.  The words live in
 * ws region "xdec_sync": int32 index LXO_XDEC_ERR_WORD of block 0 (forward chain) and of block 1 (backward chain), a block being
 * LXO_XDEC_BLOCK_BYTES bytes; both are cleared by every lxo_decoder_train_fwd / _bwd call, whether a chain runs in it or not.
 * The forward chain's word also turns the loss of lxo_ce_loss_fwd_bwd* into NaN.
 * lxo_chain_guard folds both words into the optimizer's scale ON THE DEVICE (no host synchronisation): scale_io[0] becomes NaN when
 * either word is set, else stays (have_scale != 0: the value lxo_global_norm_scale left) or becomes 1 (have_scale == 0);
 * lxo_adam_step / lxo_optimizer_step DROP the step (touch nothing) when *scale_dev is NaN.  Data parallel: lxo_decoder_train_bwd[_part]
 * turns the LAST element of `grads` (y_W_o's last) into NaN when a chain of this rank failed, the gradient all-reduce spreads it, and
 * lxo_chain_guard (grads != NULL) also drops the step when it reads a NaN there -- every rank drops the same steps, the replicas stay
 * identical.  status_out (device, 3 words, may be NULL) receives {forward word, backward word, 1 if the step is dropped} so that a
 * caller can copy them to the host asynchronously, switch to lxo_shape.step_kernels = 2 (the launch-per-step kernels) when one of its
 * own words is set, and take back the time step of a dropped Adam update.  Call it behind lxo_decoder_train_bwd and the gradient
 * exchange, on the stream the optimizer runs on.
 * Reference: the step this protects is model/img2seq.py:169 (one sess.run = forward, loss, gradients, update). */
#define LXO_XDEC_BLOCK_BYTES (4096 + (384 << 10))
#define LXO_XDEC_ERR_WORD 512
int lxo_chain_guard(const lxo_shape* s, void* ws, const float* grads, float* scale_io, int have_scale, uint32_t* status_out, void* stream);

/* tf.clip_by_global_norm scale (img2seq.py:119-121): scale_out[0] = clip / max(||g||, clip),
 * scale_out[1] = ||g|| ; device memory, no host sync.  clip <= 0 -> scale 1.  scale_out must hold LXO_GNORM_FLOATS floats:
 * the rest is scratch for one partial sum of squares per workgroup, which are added in workgroup order (no float atomics:
 * the norm is the same in every run). */
#define LXO_GNORM_FLOATS (2 + 1024)
int lxo_global_norm_scale(long long n, const float* grads, float clip, float* scale_out, void* stream);
/* tf.train.AdamOptimizer update (img2seq.py:101), TF epsilon placement:
 * m,v updated; theta -= lr_t * m / (sqrt(v) + eps), grads pre-multiplied by *scale_dev if given.
 * A NaN *scale_dev (lxo_chain_guard) drops the step: params, m and v are left untouched. */
int lxo_adam_step(long long n, float* params, const float* grads, float* m, float* v,
                  float lr_t, float beta1, float beta2, float eps, const float* scale_dev, void* stream);

/* the non-Adam branches of add_optimizer (img2seq.py:102-107), TF-1.12 defaults:
 * method 1 GradientDescentOptimizer, 2 AdagradOptimizer (slot = accumulator, caller initialises to 0.1),
 * 3 RMSPropOptimizer (slot = rms, caller initialises to 1; decay 0.9, momentum 0, epsilon 1e-10). */
int lxo_optimizer_step(int method, long long n, float* params, const float* grads, float* slot, float lr,
                       const float* scale_dev, void* stream);

/* dynamic_decode + GreedyDecoderCell (dynamic_decode.py:17-74, greedy_decoder_cell.py:40-66):
 * runs on the encoder output already in ws; ids_out int32 [B, max_steps] (device),
 * *steps_out = number of steps executed (<= max_iter + 1).  Host-synchronising.  Columns >= *steps_out of ids_out are
 * unspecified (the loop runs ahead of the host's all-finished check).  Where the shape qualifies (bf16, step_kernels 0,
 * U = O = C = 512, E = 256, B in {8, 16, 32, 64}, V <= 512, MI355X) the steps run as a persistent chain, up to 16 per
 * launch (csrc/xdec.hip: xdec_dec_kernel; LXO_XDEC_DEC=0 disables); a chain that fails to assemble is detected by the
 * call, which then repeats the decode on the launch-per-step kernels. */
int lxo_greedy_decode(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                      int id_end, int max_iter, int32_t* ids_out, int* steps_out, void* stream);
/* lxo_greedy_decode that also exports the attention weights of every step: alpha_out f32
 * [max_steps][B][Rp] (device), Rp = (R+7)/8*8, row r = region (y*W' + x) -- the data the reference taps with
 * its tf.py_func hook (attention_mechanism.py:96-105) for visualize_attention.py. */
int lxo_greedy_decode_attn(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                           int id_end, int max_iter, int32_t* ids_out, float* alpha_out, int* steps_out, void* stream);
/* The same loops ONE STEP AT A TIME: the calls behind the reference's decoder-cell protocol (dynamic_decode.py:35-36,43-44:
 * decoder_cell.initialize() / .step(time, state, inputs, finished); greedy_decoder_cell.py:46-66,
 * beam_search_decoder_cell.py:113-187).  latex_ocr_amd/model/components/ wraps them in cell objects with the reference's
 * method names.  The cell state (c, h, o, running log-probs, finished flags, the ids fed back) stays in ws.
 *   lxo_decode_begin: initialize() -- attention set-up and initial states for s->beam hypotheses per image (1 = greedy).
 *   lxo_decode_step:  step(time) -- writes column `time` of ids_out int32 [B, max_steps(, beam)] (and parents_out, beam only;
 *     may be NULL), copies the per-row finished flags (int32 [B * beam], 0 / 1, after this step, before the loop's own
 *     `time >= maximum_iterations` OR) to finished_host and the number of unfinished rows to *unfinished_host (either may be
 *     NULL; with unfinished_host the call synchronises the stream). */
int lxo_decode_begin(const lxo_shape* s, const float* params, const void* wpack, void* ws, void* stream);
int lxo_decode_step(const lxo_shape* s, const float* params, const void* wpack, void* ws, int id_end, int time,
                    int32_t* ids_out, int32_t* parents_out, int32_t* finished_host, int* unfinished_host, void* stream);
/* The AttentionState of the step-wise decode (attention_cell.py:8: AttentionState(cell_state = LSTMStateTuple(c, h), o)) as data:
 * the state that lxo_decode_step(time) / lxo_decode_cell_step(time) steps FROM (after lxo_decode_begin: time 0 = the initial state
 * of attention_cell.py:51-56; after a step at `time`: time + 1).  c, h f32 [B * beam][U], o f32 [B * beam][O], DEVICE pointers, any
 * may be NULL (skipped).  _set also takes ids_prev int32 [B * beam] = the tokens whose embeddings are the step's input
 * (greedy_decoder_cell.py:61: embedding_lookup(E, new_ids)); NULL leaves them.  With these a caller can run AttentionCell.step from
 * a state of its choosing, as the reference's cell allows (attention_cell.py:58: step(embedding, attn_cell_state)). */
int lxo_decode_state_get(const lxo_shape* s, void* ws, int time, float* c, float* h, float* o, void* stream);
int lxo_decode_state_set(const lxo_shape* s, void* ws, int time, const float* c, const float* h, const float* o,
                         const int32_t* ids_prev, void* stream);
/* AttentionCell.step alone (attention_cell.py:58-89): LSTM cell, attention context, o and logits projections for every decoder row,
 * from the state of `time` to the state of time + 1; logits f32 [B * beam][Vp] in ws region "dec_logits" (Vp = V rounded up to 32).
 * start_token != 0: the input is the learnt start token (greedy_decoder_cell.py:40-43), else the embeddings of the ids last set /
 * produced.  No arg-max, no finished flags, no beam bookkeeping: those belong to the decoder cells (lxo_decode_step). */
int lxo_decode_cell_step(const lxo_shape* s, const float* params, const void* wpack, void* ws, int time, int start_token, void* stream);
/* dynamic_decode + BeamSearchDecoderCell (beam_search_decoder_cell.py:98-250),
 * reference-faithful finalize (parents not followed): ids_out int32 [B, max_steps, beam],
 * parents_out same shape (may be NULL). */
int lxo_beam_decode(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                    int id_end, int max_iter, int32_t* ids_out, int32_t* parents_out,
                    int* steps_out, void* stream);
/* lxo_beam_decode that also exports the attention weights: alpha_out f32 [max_steps][B * beam][Rp] (device), Rp = (R+7)/8*8 --
 * entry [t][b * beam + j] = the weights decoder row j of image b attended with AT step t, i.e. of hypothesis slot j as it stood
 * BEFORE step t's top-k re-ordering: the rows the reference's tf.py_func tap sees on the merged batch x beam tensor when
 * config.decoding == "beam_search" (attention_mechanism.py:59-65,96-121; the shipped configs/model.json:13-14 decodes with beam 2).
 * The token ids_out[b][t][i] was read off row parents_out[b][t][i] of that step, so the map under which token i of step t was emitted is
 * alpha_out[t][b * beam + parents_out[b][t][i]] (time 0: every slot descends from row 0, whose k copies are identical). */
int lxo_beam_decode_attn(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                         int id_end, int max_iter, int32_t* ids_out, int32_t* parents_out, float* alpha_out,
                         int* steps_out, void* stream);

/* ---- data parallel: one process per GPU, RCCL over xGMI -------------------------------------
 * The reference trains on one device (one sess.run per step, model/img2seq.py:169).  Samples are independent through encoder,
 * decoder and loss, so a data-parallel binding needs exactly two exchanges per step, both sums over ranks:
 *   (1) the number of unmasked tokens -- the loss is the mean over ALL tokens of the GLOBAL batch (img2seq.py:69-71); the summed
 *       count is what lxo_ce_loss_fwd_bwd_dev reads from device memory;
 *   (2) the gradients, in buckets as lxo_decoder_train_bwd_part / lxo_encoder_bwd(l, l) finalise them (y_W_o, the rest of the
 *       decoder, conv6 .. conv3, conv2 + conv1), on a side stream so that they overlap the rest of the backward.
 * RCCL (librccl.so) is bound with dlopen at the first lxo_comm_* call; nothing else in the library needs it.
 * Bootstrap: rank 0 calls lxo_comm_unique_id and ships the LXO_COMM_ID_BYTES bytes to the other ranks by any host channel
 * (a file, a TCP store, MPI); every rank then calls lxo_comm_init with ITS device current (hipSetDevice). */
#define LXO_COMM_ID_BYTES 128
int lxo_comm_unique_id(void* id_out /* host, LXO_COMM_ID_BYTES */);
int lxo_comm_init(const void* unique_id, int rank, int world, void** comm_out);
/* rank / world as the communicator reports them (ncclCommUserRank / ncclCommCount): what bench.py prints as rccl_ranks_seen */
int lxo_comm_info(void* comm, int* rank, int* world);
/* In-place sum over ranks of `count` elements (LXO_F32, LXO_BF16 or LXO_I32) at device pointer ptr, enqueued on side_stream
 * (a hipStream_t).  ready_event (a hipEvent_t, may be NULL): recorded by the caller on its compute stream behind the kernels
 * that finalised the bucket; the side stream waits for it before the collective.  A caller that orders buckets on the host
 * (hipEventSynchronize, then this call) passes NULL.  The caller makes its compute stream wait for side_stream before the
 * optimizer step.  Every rank must issue the same sequence of calls on a communicator. */
int lxo_allreduce_bucket(void* comm, void* ptr, long long count, int dtype, void* side_stream, void* ready_event);
int lxo_comm_destroy(void* comm);
const char* lxo_comm_last_error(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
