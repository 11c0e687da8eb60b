// extern "C" boundary of liblxo.so (see include/lxo.h).
#include "lxo.h"
#include "gemm.h"
#include "impl.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[256] = "";
static int fail(int code, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s (code %d)", what, code);
    return code < 0 ? code : -code;
}
#define CHECK_LAUNCH(expr, what) do { int rc_ = (expr); if (rc_ != 0) return fail(rc_, what); } while (0)

extern "C" const char* lxo_last_error(void) { return g_err; }
extern "C" int lxo_version(void) { return LXO_ABI_VERSION; }
extern "C" int lxo_shape_size(void) { return (int)sizeof(lxo_shape); }

extern "C" int lxo_gemm_nt(int dt, int a_f32, int c_f32, int small, const void* A, const void* Bp, void* C,
                           int M, int N, int K, int lda, int ldb, int ldc, const float* bias, int act,
                           float alpha, int accumulate, void* stream) {
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.Bp = Bp; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.act = act; p.alpha = alpha; p.accumulate = accumulate; p.addend_rows = 1;
    CHECK_LAUNCH(lxo_launch_gemm_nt(dt, a_f32, c_f32, small, p, (hipStream_t)stream), "lxo_gemm_nt");
    return 0;
}

extern "C" int lxo_gemm_tn(int dt, int a_f32, int b_f32, const void* A, const void* B, float* C,
                           int M, int I, int J, int lda, int ldb, int ldc, int nsplit, int atomic, void* stream) {
    GemmTN p; memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.M = M; p.I = I; p.J = J; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.nsplit = nsplit; p.nbatch = 1; p.atomic = atomic;
    CHECK_LAUNCH(lxo_launch_gemm_tn(dt, a_f32, b_f32, p, (hipStream_t)stream), "lxo_gemm_tn");
    return 0;
}

// ---------------------------------------------------------------- plan ----
#define MAKE_PLAN(P, s) if (!(s)) return fail(-1, "null shape"); Plan P(*(s)); \
    { char m_[200]; int v_ = P.validate(m_, sizeof(m_)); if (v_) { snprintf(g_err, sizeof(g_err), "%s", m_); return v_; } }

extern "C" int lxo_param_num(void) { return P_COUNT; }
extern "C" const char* lxo_param_name_for(const lxo_shape* s, int id) { return lxo_param_name_mode(id, s ? s->encoder_cnn : 0); }
extern "C" long long lxo_param_total(const lxo_shape* s) { if (!s) return -1; Plan P(*s); return P.ptotal; }
extern "C" int lxo_param_info(const lxo_shape* s, int id, long long* offset, long long* count) {
    if (!s || id < 0 || id >= P_COUNT) return fail(-1, "lxo_param_info");
    Plan P(*s);
    if (offset) *offset = P.poff[id];
    if (count) *count = P.pcount[id];
    return 0;
}
extern "C" size_t lxo_wpack_bytes(const lxo_shape* s) { if (!s) return 0; Plan P(*s); return P.ktotal; }
extern "C" size_t lxo_workspace_bytes(const lxo_shape* s) { if (!s) return 0; Plan P(*s); return P.wtotal; }
extern "C" int lxo_ws_region(const lxo_shape* s, const char* name, size_t* offset, size_t* bytes) {
    if (!s || !name) return fail(-1, "lxo_ws_region");
    Plan P(*s);
    for (int i = 0; i < W_COUNT; ++i)
        if (strcmp(name, lxo_ws_name(i)) == 0) {
            if (offset) *offset = P.woff[i];
            if (bytes) *bytes = P.wbytes[i];
            return 0;
        }
    return fail(-1, "unknown workspace region");
}
extern "C" int lxo_ws_region_dtype(const lxo_shape* s, const char* name) {
    if (!s || !name) return fail(-1, "lxo_ws_region_dtype");
    const bool bf = s->dtype == LXO_BF16;
    static const char* const kCompute[] = {"p1", "y2", "p2", "y3", "y4", "p4", "y5", "p5", "y6", "img", "att_img", "emb_in", "dlogits",
                                           "d_att_img", "g0", "g1", "g2", "cols", "dec_emb", "dec_txe", "rxt"};
    static const char* const kBf16[] = {"recb", "gb", "dzb", "rhb", "rdzb", "att_exp", "datth_b"};
    static const char* const kI32[] = {"dec_ids", "dec_flags", "beam_par"};
    static const char* const kU8[] = {"m2", "m4", "m5"};
    for (const char* n : kCompute) if (strcmp(name, n) == 0) return bf ? LXO_BF16 : LXO_F32;
    for (const char* n : kBf16) if (strcmp(name, n) == 0) return LXO_BF16;
    for (const char* n : kI32) if (strcmp(name, n) == 0) return LXO_I32;
    for (const char* n : kU8) if (strcmp(name, n) == 0) return LXO_U8;
    // "d_img": the hand-over from the decoder backward to the encoder backward.  f32 mode: the f32 gradient w.r.t. the encoder
    // output; bf16 mode: d_y6 (that gradient with conv6's ReLU mask applied, bf16; conv6's bias gradient is already in grads)
    // -- and only where the decoder's last GEMM applies the mask (Plan::dimg_masked: bf16, E % 32 == 0, no row encoder); with the row
    // encoder, or an E the fused GEMM does not take, the region is the plain f32 gradient in bf16 mode too
    if (strcmp(name, "d_img") == 0) { Plan P(*s); return P.dimg_masked() ? LXO_BF16 : LXO_F32; }
    for (int i = 0; i < W_COUNT; ++i) if (strcmp(name, lxo_ws_name(i)) == 0) return LXO_F32;
    return fail(-1, "unknown workspace region");
}
extern "C" int lxo_pack_weights(const lxo_shape* s, const float* params, void* wpack, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_pack_weights(P, params, wpack, (hipStream_t)stream), "lxo_pack_weights");
    return 0;
}
extern "C" int lxo_encoder_fwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                               const uint8_t* img, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_encoder_fwd(P, params, wpack, ws, img, (hipStream_t)stream), "lxo_encoder_fwd");
    return 0;
}
extern "C" int lxo_encoder_bwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                               const uint8_t* img, float* grads, int last_layer, int first_layer, void* stream) {
    MAKE_PLAN(P, s);
    if (last_layer > 6 || first_layer < 1 || last_layer < first_layer) return fail(-1, "lxo_encoder_bwd: layer range");
    CHECK_LAUNCH(lxo_impl_encoder_bwd(P, params, wpack, ws, img, grads, last_layer, first_layer, (hipStream_t)stream), "lxo_encoder_bwd");
    return 0;
}
extern "C" int lxo_encoder_bwd_ready(const lxo_shape* s, const float* params, const void* wpack, void* ws, const uint8_t* img, float* grads,
                                     int last_layer, int first_layer, void* const* ready_events, void* stream) {
    MAKE_PLAN(P, s);
    if (last_layer > 6 || first_layer < 1 || last_layer < first_layer) return fail(-1, "lxo_encoder_bwd_ready: layer range");
    if (!ready_events) return fail(-1, "lxo_encoder_bwd_ready: null event table");
    CHECK_LAUNCH(lxo_impl_encoder_bwd(P, params, wpack, ws, img, grads, last_layer, first_layer, (hipStream_t)stream, ready_events), "lxo_encoder_bwd_ready");
    return 0;
}

#include "decoder_kernels.h"
extern "C" int lxo_decoder_train_fwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                                     const int32_t* formula, void* stream) {
    MAKE_PLAN(P, s);
    if (s->T <= 0) return fail(-1, "T must be positive");
    CHECK_LAUNCH(lxo_impl_decoder_train_fwd(P, params, wpack, ws, formula, (hipStream_t)stream), "lxo_decoder_train_fwd");
    return 0;
}
extern "C" int lxo_ce_loss_fwd_bwd(const lxo_shape* s, void* ws, const int32_t* formula, const int32_t* lengths,
                                   float inv_ntok, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_ce_loss(P, ws, formula, lengths, inv_ntok, nullptr, (hipStream_t)stream), "lxo_ce_loss_fwd_bwd");
    return 0;
}
extern "C" int lxo_ce_loss_fwd_bwd_dev(const lxo_shape* s, void* ws, const int32_t* formula, const int32_t* lengths,
                                       const float* ntok_dev, void* stream) {
    MAKE_PLAN(P, s);
    if (!ntok_dev) return fail(-1, "lxo_ce_loss_fwd_bwd_dev: null token count");
    CHECK_LAUNCH(lxo_impl_ce_loss(P, ws, formula, lengths, 0.f, ntok_dev, (hipStream_t)stream), "lxo_ce_loss_fwd_bwd_dev");
    return 0;
}
extern "C" int lxo_decoder_train_bwd(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                                     const int32_t* formula, float* grads, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_decoder_train_bwd(P, params, wpack, ws, formula, grads, 3, (hipStream_t)stream), "lxo_decoder_train_bwd");
    return 0;
}
extern "C" int lxo_train_bwd(const lxo_shape* s, const float* params, const void* wpack, void* ws, const int32_t* formula, const uint8_t* img,
                            float* grads, void* const* ready_events, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_decoder_train_bwd(P, params, wpack, ws, formula, grads, 3, (hipStream_t)stream, true, ready_events ? ready_events[0] : nullptr),
                 "lxo_train_bwd (decoder)");
    CHECK_LAUNCH(lxo_impl_encoder_bwd(P, params, wpack, ws, img, grads, 6, 1, (hipStream_t)stream, ready_events), "lxo_train_bwd (encoder)");
    return 0;
}
extern "C" int lxo_decoder_train_bwd_part(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                                          const int32_t* formula, float* grads, int parts, void* stream) {
    MAKE_PLAN(P, s);
    if (parts < 1 || parts > 3) return fail(-1, "lxo_decoder_train_bwd_part: parts must be 1, 2 or 3");
    CHECK_LAUNCH(lxo_impl_decoder_train_bwd(P, params, wpack, ws, formula, grads, parts, (hipStream_t)stream), "lxo_decoder_train_bwd_part");
    return 0;
}
extern "C" int lxo_global_norm_scale(long long n, const float* grads, float clip, float* scale_out, void* stream) {
    // scale_out: LXO_GNORM_FLOATS floats = {scale, norm, up to 1024 per-workgroup partial sums (added in order: no atomics)}
    CHECK_LAUNCH(lxo_k_global_norm_scale(n, grads, clip, scale_out + 2, scale_out, (hipStream_t)stream), "lxo_global_norm_scale");
    return 0;
}
extern "C" int lxo_adam_step(long long n, float* params, const float* grads, float* m, float* v,
                             float lr_t, float beta1, float beta2, float eps, const float* scale_dev, void* stream) {
    CHECK_LAUNCH(lxo_k_adam(params, grads, m, v, n, lr_t, beta1, beta2, eps, scale_dev, (hipStream_t)stream), "lxo_adam_step");
    return 0;
}
extern "C" int lxo_greedy_decode(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                                 int id_end, int max_iter, int32_t* ids_out, int* steps_out, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_greedy_decode(P, params, wpack, ws, id_end, max_iter, ids_out, nullptr, steps_out, (hipStream_t)stream), "lxo_greedy_decode");
    return 0;
}
extern "C" int lxo_greedy_decode_attn(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                                      int id_end, int max_iter, int32_t* ids_out, float* alpha_out, int* steps_out, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_greedy_decode(P, params, wpack, ws, id_end, max_iter, ids_out, alpha_out, steps_out, (hipStream_t)stream), "lxo_greedy_decode_attn");
    return 0;
}
extern "C" int lxo_decode_begin(const lxo_shape* s, const float* params, const void* wpack, void* ws, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_decode_begin(P, params, wpack, ws, (hipStream_t)stream), "lxo_decode_begin");
    return 0;
}
extern "C" int lxo_decode_step(const lxo_shape* s, const float* params, const void* wpack, void* ws, int id_end, int time,
                               int32_t* ids_out, int32_t* parents_out, int32_t* finished_host, int* unfinished_host, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_decode_step(P, params, wpack, ws, id_end, time, ids_out, parents_out, finished_host, unfinished_host, (hipStream_t)stream),
                 "lxo_decode_step");
    return 0;
}
extern "C" int lxo_beam_decode(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                               int id_end, int max_iter, int32_t* ids_out, int32_t* parents_out, int* steps_out, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_beam_decode(P, params, wpack, ws, id_end, max_iter, ids_out, parents_out, nullptr, steps_out, (hipStream_t)stream), "lxo_beam_decode");
    return 0;
}
extern "C" int lxo_beam_decode_attn(const lxo_shape* s, const float* params, const void* wpack, void* ws,
                                    int id_end, int max_iter, int32_t* ids_out, int32_t* parents_out, float* alpha_out, int* steps_out, void* stream) {
    MAKE_PLAN(P, s);
    if (!alpha_out) return fail(-1, "lxo_beam_decode_attn: null alpha_out");
    CHECK_LAUNCH(lxo_impl_beam_decode(P, params, wpack, ws, id_end, max_iter, ids_out, parents_out, alpha_out, steps_out, (hipStream_t)stream), "lxo_beam_decode_attn");
    return 0;
}

extern "C" int lxo_conv3x3(int dt, const void* in, const void* wpk, const float* bias, void* out, int B, int H, int W,
                           int Cin, int Ho, int Wo, int Cout, int pad, int relu, void* stream) {
    GemmNT g; memset(&g, 0, sizeof(g));
    g.A = in; g.Bp = wpk; g.C = out; g.conv = 1; g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.pad = pad;
    g.M = B * Ho * Wo; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.ldb = 9 * Cin; g.ldc = Cout;
    g.bias = bias; g.act = relu ? 1 : 0; g.alpha = 1.f; g.addend_rows = 1;
    CHECK_LAUNCH(lxo_launch_gemm_nt(dt, 0, 0, 0, g, (hipStream_t)stream), "lxo_conv3x3");
    return 0;
}
extern "C" int lxo_conv3x3_ex(int dt, const void* in, const void* wpk, const float* bias, void* out, int B, int H, int W,
                              int Cin, int Ho, int Wo, int Cout, int pad, int relu, const float* addend, int addend_rows,
                              void* out_pre, const void* relu_ref, float* colsum, void* stream) {
    GemmNT g; memset(&g, 0, sizeof(g));
    g.A = in; g.Bp = wpk; g.C = out; g.conv = 1; g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.pad = pad;
    g.M = B * Ho * Wo; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.ldb = 9 * Cin; g.ldc = Cout;
    g.bias = bias; g.act = relu ? 1 : 0; g.alpha = 1.f;
    g.addend = addend; g.addend_rows = addend_rows > 0 ? addend_rows : 1; g.out_pre = out_pre;
    g.relu_ref = relu_ref; g.ldr = Cout; g.colsum = colsum;
    CHECK_LAUNCH(lxo_launch_gemm_nt(dt, 0, 0, 0, g, (hipStream_t)stream), "lxo_conv3x3_ex");
    return 0;
}
extern "C" int lxo_attention_fwd(int dt, const void* att_img, const void* img, const float* att_h, const float* beta,
                                 float* alpha, float* part, float* ctx, int ldctx, int nv, int R, int E, int C, int beam,
                                 void* stream) {
    int nch = (512 + nv - 1) / nv; if (nch > 16) nch = 16;
    const int by_rows = R / 32 > 0 ? R / 32 : 1; if (nch > by_rows) nch = by_rows;
    const int need = (R + 1023) / 1024; if (nch < need) nch = need;
    Slabs none = {nullptr, 0, 0, 0};
    CHECK_LAUNCH(lxo_k_attn_fwd(dt, att_img, img, att_h, none, nullptr, beta, alpha, part, ctx, ldctx, nullptr, 0, nv, R, (R + 7) / 8 * 8, E, C, beam < 1 ? 1 : beam,
                                nch, 0, (hipStream_t)stream), "lxo_attention_fwd");
    return 0;
}

// test / micro-benchmark entry: split-K slab GEMM (see gemm.hip)
extern "C" int lxo_gemm_slab(int dt, const void* A, const void* Bp, float* slab, int M, int N, int K, int lda, int ldb, int ldc,
                             long long slab_stride, void* stream) {
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.Bp = Bp; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = 1.f; p.addend_rows = 1;
    CHECK_LAUNCH(lxo_launch_gemm_slab(dt, p, slab, slab_stride, (hipStream_t)stream), "lxo_gemm_slab");
    return 0;
}

extern "C" int lxo_conv3x3_wgrad(int dt, const void* in, const void* dout, float* dw, int B, int H, int W,
                                 int Cin, int Ho, int Wo, int Cout, int pad, void* stream) {
    GemmTN g; memset(&g, 0, sizeof(g));
    g.A = in; g.B = dout; g.C = dw; g.conv = 1; g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.pad = pad;
    g.M = B * Ho * Wo; g.I = 9 * Cin; g.J = Cout; g.lda = Cin; g.ldb = Cout; g.ldc = Cout;
    const int tiles = cdiv(g.I, 128) * cdiv(g.J, 128);
    int ns = cdiv(1024, tiles); const int maxs = g.M / 256 > 0 ? g.M / 256 : 1; if (ns > maxs) ns = maxs;
    g.nsplit = ns < 1 ? 1 : ns; g.nbatch = 1; g.atomic = 1;
    CHECK_LAUNCH(lxo_launch_gemm_tn(dt, 0, 0, g, (hipStream_t)stream), "lxo_conv3x3_wgrad");
    return 0;
}

extern "C" int lxo_chain_guard(const lxo_shape* s, void* ws, const float* grads, float* scale_io, int have_scale, uint32_t* status_out, void* stream) {
    MAKE_PLAN(P, s);
    if (!ws || !scale_io) return fail(-1, "lxo_chain_guard: null workspace / scale");
    CHECK_LAUNCH(lxo_impl_chain_guard(P, ws, grads, scale_io, have_scale, status_out, (hipStream_t)stream), "lxo_chain_guard");
    return 0;
}
extern "C" int lxo_decode_state_get(const lxo_shape* s, void* ws, int time, float* c, float* h, float* o, void* stream) {
    MAKE_PLAN(P, s);
    if (time < 0) return fail(-5, "lxo_decode_state_get: time < 0");
    CHECK_LAUNCH(lxo_impl_decode_state_get(P, ws, time, c, h, o, (hipStream_t)stream), "lxo_decode_state_get");
    return 0;
}
extern "C" int lxo_decode_state_set(const lxo_shape* s, void* ws, int time, const float* c, const float* h, const float* o,
                                    const int32_t* ids_prev, void* stream) {
    MAKE_PLAN(P, s);
    if (time < 0) return fail(-5, "lxo_decode_state_set: time < 0");
    CHECK_LAUNCH(lxo_impl_decode_state_set(P, ws, time, c, h, o, ids_prev, (hipStream_t)stream), "lxo_decode_state_set");
    return 0;
}
extern "C" int lxo_decode_cell_step(const lxo_shape* s, const float* params, const void* wpack, void* ws, int time, int start_token, void* stream) {
    MAKE_PLAN(P, s);
    CHECK_LAUNCH(lxo_impl_decode_cell_step(P, params, wpack, ws, time, start_token, (hipStream_t)stream), "lxo_decode_cell_step");
    return 0;
}

extern "C" int lxo_set_side_stream(void* stream) {
    CHECK_LAUNCH(lxo_impl_set_side_stream((hipStream_t)stream), "lxo_set_side_stream");
    return 0;
}
extern "C" int lxo_set_encoder_side_stream(void* stream) {
    CHECK_LAUNCH(lxo_impl_set_encoder_side_stream((hipStream_t)stream), "lxo_set_encoder_side_stream");
    return 0;
}

extern "C" int lxo_optimizer_step(int method, long long n, float* params, const float* grads, float* slot, float lr,
                                  const float* scale_dev, void* stream) {
    CHECK_LAUNCH(lxo_k_simple_opt(params, grads, slot, n, lr, method, scale_dev, (hipStream_t)stream), "lxo_optimizer_step");
    return 0;
}
