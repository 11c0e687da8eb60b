// Host orchestration of the decoder: attention set-up, the T-step recurrent loop
// (forward and BPTT), loss, deferred weight-gradient GEMMs, greedy and beam decode.
// Reference graph: model/decoder.py:41-72, model/components/attention_mechanism.py,
// model/components/attention_cell.py:58-89, model/img2seq.py:68-75.
#include "plan.h"
#include "gemm.h"
#include "decoder_kernels.h"
#include "rstep.h"
#include "xdec.h"
#include "dimg.h"
#include "api_util.h"
#include "timing.h"

namespace {
// dense C = act(A * Bp^T + bias) on the compute dtype `dt`
int nt(const Plan& P, bool a_f32, bool c_f32, bool small, const void* A, int lda, const void* Bp, int ldb, void* C, int ldc,
       int M, int N, int K, const float* bias, int act, bool accumulate, hipStream_t st) {
    GemmNT g; memset(&g, 0, sizeof(g));
    g.A = A; g.Bp = Bp; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.bias = bias; g.act = act; g.alpha = 1.f; g.accumulate = accumulate ? 1 : 0; g.addend_rows = 1;
    if (P.s.dtype == LXO_F32) { a_f32 = true; c_f32 = true; }
    return lxo_launch_gemm_nt(P.s.dtype, a_f32, c_f32, small, g, st);
}
// C[I][J] += A^T B over M rows
int tn(const Plan& P, bool a_f32, bool b_f32, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
       int M, int I, int J, hipStream_t st, DetScratch det = DetScratch{nullptr, 0}) {
    GemmTN g; memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.M = M; g.I = I; g.J = J; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    const int tiles = cdiv(I, 128) * cdiv(J, 128);
    int ns = cdiv(512, tiles);
    const int maxs = M / 64 > 0 ? M / 64 : 1;
    if (ns > maxs) ns = maxs;
    g.nsplit = ns < 1 ? 1 : ns; g.nbatch = 1; g.atomic = 1;
    // bf16 deterministic mode: partial tiles to the slab, added in range order (or, without a slab, one row range per output tile)
    if (P.bf && P.det()) { if (det.p) { g.det_slab = det.p; g.det_floats = det.floats; } else g.nsplit = 1; }
    if (P.s.dtype == LXO_F32) { a_f32 = true; b_f32 = true; }
    return lxo_launch_gemm_tn(P.s.dtype, a_f32, b_f32, g, st);
}
// split-K partial products slab[ks] = A[:, ks*128:+128] * Bp^T  (one memory round trip; consumers add the slabs)
int slab(const Plan& P, const float* A, int lda, const void* Bp, int ldb, float* out, int M, int N, int K, hipStream_t st) {
    GemmNT g; memset(&g, 0, sizeof(g));
    g.A = A; g.Bp = Bp; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N; g.alpha = 1.f; g.addend_rows = 1;
    return lxo_launch_gemm_slab(P.s.dtype, g, out, (long long)M * N, st);
}
Slabs view(const float* p, int K, int M, int N) { Slabs s = {p, K / 128, (long long)M * N, N}; return s; }
const Slabs kNoSlabs = {nullptr, 0, 0, 0};
}  // namespace

// the attention kernels walk their row blocks in alternating directions from step to step (L2 reuse; LXO_ATT_ALT=0: always forward)
static bool att_alternate() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LXO_ATT_ALT"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v == 1;
}
// att_img projection + initial states (attention_mechanism.py:19-43, 124-153; attention_cell.py:51-56)
// nv = number of decoder rows (B for training/greedy, B*beam for beam search; rows v use image v / beam).
static bool fused_steps(const Plan& P);
// out[M][N] (+)= act(A[M][K] W[N][K]^T + bias) for the few-row GEMMs outside the loops (initial states, their gradients) on the
// fused step kernels (16-row tiles, one workgroup per 16 columns: 128 workgroups where gemm_skinny_kernel has 16)
static int rs_dense(const Plan& P, const float* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K,
                    const float* bias, bool tanh_act, bool accumulate, hipStream_t st) {
    RStep a; memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K; a.U = P.s.U; a.O = P.s.O; a.zx_row = -1;
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.out = out; a.ldo = ldo; a.bias = bias; a.accumulate = accumulate ? 1 : 0;
    a.epi = tanh_act ? RS_TANH_O : RS_PLAIN;              // dropout descriptor zero: tanh only
    a.dr.inv_keep = 1.f;
    return lxo_launch_rstep(P.s.dtype, 0, a, st);
}

static int attention_prepare(const Plan& P, const float* prm, const void* wp, void* ws, int beam, hipStream_t st) {
    const int B = P.s.B, C = P.s.C, E = P.s.E, U = P.s.U, O = P.s.O;
    RC(nt(P, false, false, false, P.ws<void>(ws, W_IMG), C, P.pk(wp, K_ATT_IMG_T), C, P.ws<void>(ws, W_ATT_IMG), E,
          B * P.R, E, C, nullptr, 0, false, st));
    RC(lxo_k_rowmean(P.s.dtype, P.ws<void>(ws, W_IMG), P.ws<float>(ws, W_MEAN), B, P.R, C, st));
    float* rec0 = P.ws<float>(ws, W_REC);
    float* cs0 = P.ws<float>(ws, W_CS);
    const char* wt = (const char*)P.pk(wp, K_INIT_T);
    float* mean = P.ws<float>(ws, W_MEAN);
    if (beam <= 1 && fused_steps(P)) {
        RC(rs_dense(P, mean, C, wt, C, cs0, U, B, U, C, prm + P.poff[P_BC0], true, false, st));
        RC(rs_dense(P, mean, C, wt + (size_t)U * C * P.esz, C, rec0 + O, P.REC, B, U, C, prm + P.poff[P_BH0], true, false, st));
        RC(rs_dense(P, mean, C, wt + (size_t)2 * U * C * P.esz, C, rec0, P.REC, B, O, C, prm + P.poff[P_BO0], true, false, st));
    } else if (beam <= 1) {
        RC(nt(P, true, true, true, mean, C, wt, C, cs0, U, B, U, C, prm + P.poff[P_BC0], 2, false, st));
        RC(nt(P, true, true, true, mean, C, wt + (size_t)U * C * P.esz, C, rec0 + O, P.REC, B, U, C, prm + P.poff[P_BH0], 2, false, st));
        RC(nt(P, true, true, true, mean, C, wt + (size_t)2 * U * C * P.esz, C, rec0, P.REC, B, O, C, prm + P.poff[P_BO0], 2, false, st));
    } else {
        // compute once per image into the beam scratch, then tile over the beam (beam_search_decoder_cell.py:98-109)
        float* tmp = P.ws<float>(ws, W_BEAM_TMP);
        float* tc = tmp; float* th = tmp + (size_t)B * U; float* to = th + (size_t)B * U;
        RC(nt(P, true, true, true, mean, C, wt, C, tc, U, B, U, C, prm + P.poff[P_BC0], 2, false, st));
        RC(nt(P, true, true, true, mean, C, wt + (size_t)U * C * P.esz, C, th, U, B, U, C, prm + P.poff[P_BH0], 2, false, st));
        RC(nt(P, true, true, true, mean, C, wt + (size_t)2 * U * C * P.esz, C, to, O, B, O, C, prm + P.poff[P_BO0], 2, false, st));
        RC(lxo_k_tile_rows(tc, U, cs0, U, B * beam, beam, U, st));
        RC(lxo_k_tile_rows(th, U, rec0 + O, P.REC, B * beam, beam, U, st));
        RC(lxo_k_tile_rows(to, O, rec0, P.REC, B * beam, beam, O, st));
    }
    return 0;
}

// Side stream for the half-batch interleave of the recurrent loop (set per host thread through
// lxo_set_side_stream; null = single stream).  The recurrence is a chain of ~13 short, latency-bound
// launches per step pair; the two halves of the batch are independent, so running them on two HIP
// streams lets one half's launch/latency gaps be filled by the other half's kernels.
static thread_local hipStream_t g_side = nullptr;
static thread_local hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;
int lxo_impl_set_side_stream(hipStream_t s) {
    g_side = s;
    if (s && !g_ev_fork) {
        HIPRC(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming));
        HIPRC(hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming));
    }
    return 0;
}
static int fork_side(hipStream_t st) { HIPRC(hipEventRecord(g_ev_fork, st)); HIPRC(hipStreamWaitEvent(g_side, g_ev_fork, 0)); return 0; }
static int join_side(hipStream_t st) { HIPRC(hipEventRecord(g_ev_join, g_side)); HIPRC(hipStreamWaitEvent(st, g_ev_join, 0)); return 0; }

// One AttentionCell.step (attention_cell.py:58-89) for rows [r0, r0+nr) of nv decoder rows.  The step
// pointers address row 0; zx_t must already hold emb_t * K[0:D] + b; rec_prev/cs_prev = state t-1
// (o final), rec_cur/cs_cur receive state t.  Every GEMM is a split-K slab GEMM; the kernel that
// consumes a product adds its slabs.
static int cell_step(const Plan& P, const float* prm, const void* wp, void* ws, int r0, int nr, int beam,
                     const float* zx_t, const float* rec_prev, const float* cs_prev, float* rec_cur, float* cs_cur,
                     float* gates_t, float* atth_t, float* alpha_t, Drop dr, hipStream_t st) {
    const int C = P.s.C, E = P.s.E, U = P.s.U, O = P.s.O;
    const size_t r = (size_t)r0;
    float* s1 = P.ws<float>(ws, W_S_K1) + r * (P.XH / 128) * 4 * U;
    float* s2 = P.ws<float>(ws, W_S_K2) + r * (U / 128) * E;
    float* s4 = P.ws<float>(ws, W_S_K4) + r * (P.HC / 128) * O;
    zx_t += r * 4 * U; rec_prev += r * P.REC; cs_prev += r * U; rec_cur += r * P.REC; cs_cur += r * U;
    if (gates_t) gates_t += r * 4 * U;
    atth_t += r * E; alpha_t += r * P.Rp;
    const char* att_img = (const char*)P.ws<void>(ws, W_ATT_IMG) + (r / beam) * P.R * E * P.esz;
    const char* img = (const char*)P.ws<void>(ws, W_IMG) + (r / beam) * P.R * C * P.esz;
    float* part = P.ws<float>(ws, W_APART) + r * 32 * (C + 2);
    // z = zx + [o_prev, h_prev] K[D:]            (attention_cell.py:70-71)
    RC(slab(P, rec_prev, P.REC, P.pk(wp, K_LSTM_RT), P.ldRT, s1, nr, 4 * U, P.XH, st));
    RC(lxo_k_lstm_fwd(zx_t, view(s1, P.XH, nr, 4 * U), cs_prev, gates_t, cs_cur, rec_cur + O, rec_cur + P.OFF_HT, P.REC, dr, nr, U, st));
    // att_h = h~ W                                (attention_mechanism.py:79)
    RC(slab(P, rec_cur + P.OFF_HT, P.REC, P.pk(wp, K_ATT_H_T), P.ldAHT, s2, nr, E, U, st));
    RC(lxo_k_attn_fwd(P.s.dtype, att_img, img, nullptr, view(s2, U, nr, E), atth_t,
                      prm + P.poff[P_BETA], alpha_t, part, rec_cur + P.OFF_CTX, P.REC, nullptr, 0, nr, P.R, P.Rp, E, C, beam,
                      P.attn_chunks(nr), 0, st));
    // o = tanh([h, ctx] [o_W_h; o_W_c])           (attention_cell.py:82)
    RC(slab(P, rec_cur + P.OFF_HT, P.REC, P.pk(wp, K_OW_T), P.ldOWT, s4, nr, O, P.HC, st));
    RC(lxo_k_tanh_finalize(view(s4, P.HC, nr, O), rec_cur, P.REC, dr, nr, O, st));
    return 0;
}

// The same step on the fused full-K kernels (rstep.hip): 5 dependent launches instead of 7, no split-K slabs.
// bf16 mode reads the GEMM A operands from the bf16 mirror of the record (recb_*), which every producer writes next
// to its f32 value; in the f32 parity mode the mirrors are null and A is the f32 record itself.
static int cell_step_fused(const Plan& P, const float* prm, const void* wp, void* ws, int nr, int beam,
                           const float* zx_t, const float* rec_prev, const float* cs_prev, float* rec_cur, float* cs_cur,
                           const bf16_t* recb_prev, bf16_t* recb_cur,
                           float* gates_t, float* atth_t, float* alpha_t, Drop dr, hipStream_t st,
                           const int* zx_idx = nullptr, int zx_row = -1, const int* a_par = nullptr) {
    // zx_t: training = the step's rows of emb K[0:D] + b; decode = the per-token table (zx_idx picks the row of each decoder row,
    // zx_row >= 0 = one row for all: the start token)
    const int C = P.s.C, E = P.s.E, U = P.s.U, O = P.s.O;
    const bool bf = P.bf;
    const char* att_img = (const char*)P.ws<void>(ws, W_ATT_IMG);
    const char* img = (const char*)P.ws<void>(ws, W_IMG);
    float* part = P.ws<float>(ws, W_APART);
    RStep a; memset(&a, 0, sizeof(a));
    a.M = nr; a.U = U; a.O = O; a.dr = dr; a.zx_row = -1;
    // z = zx + [o_prev, h_prev] K[D:] -> gates, c, h, h~        (attention_cell.py:70-72)
    RStep k1 = a;
    k1.A = bf ? (const void*)recb_prev : (const void*)rec_prev; k1.lda = bf ? P.RECB : P.REC;
    k1.W = P.pk(wp, K_LSTM_RT); k1.ldw = P.ldRT; k1.N = 4 * U; k1.K = P.XH; k1.epi = RS_LSTM_FWD;
    k1.zx = zx_t; k1.c_prev = cs_prev; k1.gates = gates_t; k1.c_out = cs_cur;
    k1.zx_idx = zx_idx; k1.zx_vocab = P.s.V; k1.zx_row = zx_row;
    k1.a_par = a_par; k1.a_k = beam;                         // beam decode: the previous state is read through the parents (no re-ordering launch)
    k1.out = rec_cur + O; k1.out2 = rec_cur + P.OFF_HT; k1.ldo = P.REC;
    if (bf) { k1.outb = recb_cur + O; k1.out2b = recb_cur + P.OFF_HT; k1.ldob = P.RECB; }
    RC(lxo_launch_rstep(P.s.dtype, bf, k1, st));
    // att_h = h~ W                                                (attention_mechanism.py:79)
    RStep k2 = a;
    k2.A = bf ? (const void*)(recb_cur + P.OFF_HT) : (const void*)(rec_cur + P.OFF_HT); k2.lda = bf ? P.RECB : P.REC;
    k2.W = P.pk(wp, K_ATT_H_T); k2.ldw = P.ldAHT; k2.N = E; k2.K = U; k2.epi = RS_PLAIN;
    k2.out = atth_t; k2.ldo = E;
    RC(lxo_launch_rstep(P.s.dtype, bf, k2, st));
    {
    LxoTimed tm("attn_fwd", "part+combine", (double)nr * P.R * (E + C) * P.esz, st);
    RC(lxo_k_attn_fwd(P.s.dtype, att_img, img, atth_t, kNoSlabs, nullptr,
                      prm + P.poff[P_BETA], alpha_t, part, rec_cur + P.OFF_CTX, P.REC, bf ? recb_cur + P.OFF_CTX : nullptr, P.RECB, nr, P.R, P.Rp, E, C, beam,
                      P.attn_chunks(nr), att_alternate() ? (dr.t & 1) : 0, st,
                      (!gates_t && P.att_exp()) ? P.ws<void>(ws, W_ATT_EXP) : nullptr));      // decode (no gates kept), bf16: the E-domain copy the decode set-up wrote
    }
    // o = dropout(tanh([h~, ctx] [o_W_h; o_W_c]))                  (attention_cell.py:82-83)
    RStep k4 = a;
    k4.A = bf ? (const void*)(recb_cur + P.OFF_HT) : (const void*)(rec_cur + P.OFF_HT); k4.lda = bf ? P.RECB : P.REC;
    k4.W = P.pk(wp, K_OW_T); k4.ldw = P.ldOWT; k4.N = O; k4.K = P.HC; k4.epi = RS_TANH_O;
    k4.out = rec_cur; k4.ldo = P.REC;
    if (bf) { k4.outb = recb_cur; k4.ldob = P.RECB; }
    RC(lxo_launch_rstep(P.s.dtype, bf, k4, st));
    return 0;
}
// The fused step kernels split a contraction over the 4 waves in chunks of 32 / 64 k whose count must be a power of two up to 16
// (rstep.hip: launch_nch): true for every contraction of the shipped sizes (and of any U, O, E, C in {128, 256, 512} that are
// equal); a mixed shape such as U = 128, C = 256 (K = 384) runs on round 1's split-K step kernels instead.
static bool rstep_k_ok(int K, bool bf) {
    if (K % 128) return false;
    const int kq = K / 4;
    auto pow2_16 = [](int n) { return n >= 1 && n <= 16 && (n & (n - 1)) == 0; };
    if (bf && kq % 64 == 0 && pow2_16(kq / 64)) return true;
    return kq % 32 == 0 && pow2_16(kq / 32);
}
static bool fused_steps(const Plan& P) {
    if (P.s.step_kernels == 1) return false;
    const int ks[7] = {P.XH, P.s.U, P.HC, P.s.O, P.s.E, 4 * P.s.U, P.s.C};     // C: the initial-state projections run on the step kernels too
    for (int k : ks) if (!rstep_k_ok(k, P.bf) || !rstep_k_ok(k, false)) return false;     // the f32-operand launches (A converted on load) use 32-k chunks
    return true;
}
// bf16 mirror of the [o | h] columns of `rows` records (initial state; beam re-ordering)
static int mirror_oh(const Plan& P, void* ws, size_t slot_rows, int rows, hipStream_t st) {
    if (!P.bf) return 0;
    return lxo_k_mirror(P.ws<float>(ws, W_REC) + slot_rows * P.REC, P.REC, P.ws<bf16_t>(ws, W_RECB) + slot_rows * P.RECB, P.RECB, rows, P.XH, st);
}

// All B rows run all T steps, padded ones included, as the reference does (decoder.py:57: dynamic_rnn without sequence_length; the padded
// steps are masked in the loss, img2seq.py:68-71).
int lxo_impl_decoder_train_fwd(const Plan& P, const float* prm, const void* wp, void* ws, const int* formula, hipStream_t st) {
    const int B = P.s.B, T = P.s.T, U = P.s.U, O = P.s.O, E = P.s.E, D = P.s.D, V = P.s.V, C = P.s.C;
    RC(attention_prepare(P, prm, wp, ws, 1, st));
    if (P.att_exp()) RC(lxo_k_att_exp(P.ws<void>(ws, W_ATT_IMG), P.ws<void>(ws, W_ATT_EXP), (long long)B * P.R * E, st));     // E_x = e^{2 att_img}: what the recurrence's attention kernels read
    RC(lxo_k_embed_gather(P.s.dtype, prm + P.poff[P_EMB], prm + P.poff[P_START], formula, P.ws<void>(ws, W_EMB_IN), B, T, D, P.Dp, V, st));
    float* zx = P.ws<float>(ws, W_ZX);
    RC(nt(P, false, true, false, P.ws<void>(ws, W_EMB_IN), P.Dp, P.pk(wp, K_LSTM_XT), P.Dp, zx, 4 * U, T * B, 4 * U, P.Dp,
          prm + P.poff[P_LSTM_B], 0, false, st));
    float* rec = P.ws<float>(ws, W_REC); float* cs = P.ws<float>(ws, W_CS);
    const bool dual = g_side != nullptr && B >= 2 && (B % 2) == 0;
    const int nh = dual ? 2 : 1;
    const bool fused = fused_steps(P) && !dual;
    bool chain_done = false;
    if (fused && P.bf && P.s.step_kernels == 0) {
        // the whole recurrence in one launch: 8 XCD-local chains of B / 8 samples (xdec.hip); -2 = the shape does not qualify
        RC(mirror_oh(P, ws, 0, B, st));
        XDecFwd x; memset(&x, 0, sizeof(x));
        x.Wrt = (const bf16_t*)P.pk(wp, K_LSTM_RT); x.ldrt = P.ldRT;
        x.Wah = (const bf16_t*)P.pk(wp, K_ATT_H_T); x.ldah = P.ldAHT;
        x.Wow = (const bf16_t*)P.pk(wp, K_OW_T); x.ldow = P.ldOWT;
        x.beta = prm + P.poff[P_BETA];
        x.att_img = P.ws<bf16_t>(ws, W_ATT_IMG); x.img = P.ws<bf16_t>(ws, W_IMG);
        x.att_exp = P.att_exp() ? P.ws<bf16_t>(ws, W_ATT_EXP) : nullptr;
        x.zx = zx; x.rec = rec; x.recb = P.ws<bf16_t>(ws, W_RECB); x.cs = cs;
        x.gates = P.ws<float>(ws, W_GATES); x.atth = P.ws<float>(ws, W_ATTH); x.alpha = P.ws<float>(ws, W_ALPHA);
        x.part = P.ws<float>(ws, W_APART); x.sync = P.ws<unsigned>(ws, W_XSYNC);
        x.T = T; x.B = B; x.R = P.R; x.Rp = P.Rp; x.REC = P.REC; x.RECB = P.RECB;
        x.dr = P.drop(0, 0);
        LxoTimed tm("xdec_fwd", "chain", (double)T * B * P.R * (E + C) * P.esz, st);
        const int rc = lxo_launch_xdec_fwd(x, U, O, C, E, st);
        if (rc == 0) chain_done = true;
        else if (rc != -2) return rc < 0 ? rc : -rc;
    }
    if (P.bf && !chain_done) HIPRC(hipMemsetAsync(P.ws<unsigned>(ws, W_XSYNC), 0, kXDecSyncBytes, st));      // no chain in this call: clear its tickets and its error word (the loss kernel and lxo_chain_guard read it; a reused workspace may hold another shape's bytes here)
    if (fused && !chain_done) {
        bf16_t* recb = P.ws<bf16_t>(ws, W_RECB);
        RC(mirror_oh(P, ws, 0, B, st));
        for (int t = 0; t < T; ++t) {
            const int nr = B;
            if (nr <= 0) break;
            RC(cell_step_fused(P, prm, wp, ws, nr, 1, zx + (size_t)t * B * 4 * U,
                               rec + (size_t)t * B * P.REC, cs + (size_t)t * B * U,
                               rec + (size_t)(t + 1) * B * P.REC, cs + (size_t)(t + 1) * B * U,
                               recb + (size_t)t * B * P.RECB, recb + (size_t)(t + 1) * B * P.RECB,
                               P.ws<float>(ws, W_GATES) + (size_t)t * B * 4 * U,
                               P.ws<float>(ws, W_ATTH) + (size_t)t * B * E,
                               P.ws<float>(ws, W_ALPHA) + (size_t)t * B * P.Rp, P.drop(t, 0), st));
        }
    }
    if (dual) RC(fork_side(st));
    for (int t = 0; t < T && !fused; ++t)
        for (int h = 0; h < nh; ++h) {
            const int hb = B / nh;
            if (hb <= 0) continue;
            RC(cell_step(P, prm, wp, ws, h * hb, hb, 1, zx + (size_t)t * B * 4 * U,
                         rec + (size_t)t * B * P.REC, cs + (size_t)t * B * U,
                         rec + (size_t)(t + 1) * B * P.REC, cs + (size_t)(t + 1) * B * U,
                         P.ws<float>(ws, W_GATES) + (size_t)t * B * 4 * U,
                         P.ws<float>(ws, W_ATTH) + (size_t)t * B * E,
                         P.ws<float>(ws, W_ALPHA) + (size_t)t * B * P.Rp, P.drop(t, h * hb), h ? g_side : st));
        }
    if (dual) RC(join_side(st));
    // logits_t = o_t y_W_o for every step at once  (attention_cell.py:84)
    if (P.bf && fused)      // A = the bf16 mirror of o_t the step kernels wrote next to the f32 record
        RC(nt(P, false, true, false, P.ws<bf16_t>(ws, W_RECB) + (size_t)B * P.RECB, P.RECB, P.pk(wp, K_YWO_T), O, P.ws<float>(ws, W_LOGITS), P.Vp,
              T * B, V, O, nullptr, 0, false, st));
    else
        RC(nt(P, true, true, false, rec + (size_t)B * P.REC, P.REC, P.pk(wp, K_YWO_T), O, P.ws<float>(ws, W_LOGITS), P.Vp,
              T * B, V, O, nullptr, 0, false, st));
    return 0;
}

int lxo_impl_ce_loss(const Plan& P, void* ws, const int* formula, const int* lengths, float inv_ntok, const float* ntok_dev, hipStream_t st) {
    HIPRC(hipMemsetAsync(P.ws<float>(ws, W_LOSS), 0, 64, st));
    RC(lxo_k_ce_loss(P.s.dtype, P.ws<float>(ws, W_LOGITS), formula, lengths, P.ws<void>(ws, W_DLOGITS), P.ws<float>(ws, W_LOSS),
                     inv_ntok, ntok_dev, P.bf ? P.ws<unsigned>(ws, W_XSYNC) + 8 * 64 : nullptr, P.s.B, P.s.T, P.s.V, P.Vp,
                     DetScratch{P.ws<float>(ws, W_DET), P.wbytes[W_DET] / 4}, st));      // every mode: per-workgroup partial statistics added in order (25 us; the atomic form measured 53)
    return 0;
}

// The weight-gradient side stream of the encoder backward (model_encoder.hip: lxo_set_encoder_side_stream) also takes the decoder's
// deferred all-step weight gradients (round 5): behind the backward chain the critical path is init states -> d_att_img (bound by the
// transcendental rate) -> d_img -> conv6; the four dense dW GEMMs, d_z's column sum, the embedding gradient and dW_att_img feed nothing
// downstream and run beside it.  The call joins the two streams before it returns (its caller reduces / applies the gradients).
hipStream_t lxo_impl_encoder_side_stream();
static thread_local hipEvent_t g_dbw_fork = nullptr, g_dbw_fork2 = nullptr, g_dbw_join = nullptr, g_dbw_init = nullptr;

int lxo_impl_decoder_train_bwd(const Plan& P, const float* prm, const void* wp, void* ws, const int* formula, float* grads,
                               int parts, hipStream_t st, bool defer_join, void* ready) {
    const int B = P.s.B, T = P.s.T, C = P.s.C, E = P.s.E, U = P.s.U, O = P.s.O, D = P.s.D, V = P.s.V;
    const int TB = T * B;
    auto gw = [&](int pid) { return grads + P.poff[pid]; };
    float* rec = P.ws<float>(ws, W_REC); float* cs = P.ws<float>(ws, W_CS);
    float* dolog = P.ws<float>(ws, W_DOLOG); float* gall = P.ws<float>(ws, W_G); float* dhc = P.ws<float>(ws, W_DHC);
    float* de = P.ws<float>(ws, W_DE); float* datth = P.ws<float>(ws, W_DATTH); float* dz = P.ws<float>(ws, W_DZ);
    float* dcc = P.ws<float>(ws, W_DCC);
    float* gates = P.ws<float>(ws, W_GATES); float* atth = P.ws<float>(ws, W_ATTH); float* alpha = P.ws<float>(ws, W_ALPHA);
    const void* dlog = P.ws<void>(ws, W_DLOGITS);
    const DetScratch det = P.det_scratch(ws);      // f32 parity mode: ordered reductions (no float atomics); null in bf16 mode

    // the same conditions as in lxo_impl_decoder_train_fwd: the fused step kernels ran (and left the bf16 mirrors of the record)
    const bool dual = g_side != nullptr && B >= 2 && (B % 2) == 0;
    const bool fused = fused_steps(P) && !dual;
    // d_o (from logits) for every step, and dy_W_o
    if (parts & 1) {
        RC(nt(P, false, true, false, dlog, P.Vp, P.pk(wp, K_YWO), P.Vp, dolog, O, TB, O, P.Vp, nullptr, 0, false, st));
        if (P.bf && fused) RC(tn(P, false, false, P.ws<bf16_t>(ws, W_RECB) + (size_t)B * P.RECB, P.RECB, dlog, P.Vp, gw(P_YWO), V, TB, O, V, st, det));
        else RC(tn(P, true, false, rec + (size_t)B * P.REC, P.REC, dlog, P.Vp, gw(P_YWO), V, TB, O, V, st));
    }
    // a failed forward chain poisons the last gradient element (y_W_o's, final after this part): under data parallelism its all-reduce
    // carries the failure to every rank (lxo_chain_guard)
    if ((parts & 1) && P.bf) RC(lxo_k_chain_poison(P.ws<unsigned>(ws, W_XSYNC) + 8 * 64, nullptr, grads + P.ptotal - 1, st));
    if (!(parts & 2)) return 0;

    // the launch-per-step kernels accumulate d_c in place and d_att_h with atomics; the backward chain writes both with plain stores
    const bool want_chain = fused && P.bf && P.s.step_kernels == 0;
    bool bwd_chain = false;                              // the backward chain ran (and left the bf16 mirror of d_att_h)
    auto zero_acc = [&]() -> int {
        HIPRC(hipMemsetAsync(dcc, 0, (size_t)B * U * 4, st));
        HIPRC(hipMemsetAsync(datth, 0, (size_t)TB * E * 4, st));
        return 0;
    };
    if (!want_chain) RC(zero_acc());
    // no backward chain in this call: clear block 1's tickets and error word (lxo_chain_guard / Engine.chain_status read them; the workspace
    // is reused across shapes and the region offsets move with the shape, so stale bytes there would read as a broken chain)
    if (P.bf && !want_chain) HIPRC(hipMemsetAsync(P.ws<unsigned>(ws, W_XSYNC) + kXDecBlockBytes / 4, 0, kXDecSyncBytes, st));
    const int nh = dual ? 2 : 1;
    float* dxh = P.ws<float>(ws, W_DXH);
    if (fused) {
        // 4 dependent launches per step: [d_h~|d_ctx] GEMM, attention backward, d_att_h GEMM + LSTM backward,
        // d_z K^T GEMM + the tanh' of step t-1.  All operands are final values (no split-K slabs).
        const bool bf = P.bf;
        bf16_t* gb = P.ws<bf16_t>(ws, W_GB); bf16_t* dzb = P.ws<bf16_t>(ws, W_DZB);
        float* carry_h = P.ws<float>(ws, W_CARRYH);
        const char* att_img = (const char*)P.ws<void>(ws, W_ATT_IMG);
        const char* img = (const char*)P.ws<void>(ws, W_IMG);
        const int t_last = T - 1, n_last = B;
        // g_{t_last} = d_o(logits) * tanh'   (no carry yet)
        RC(lxo_k_tanh_bwd(dolog + (size_t)t_last * B * O, O, kNoSlabs, rec + (size_t)(t_last + 1) * B * P.REC, P.REC,
                          gall + (size_t)t_last * B * O, O, bf ? gb + (size_t)t_last * B * P.GBP : nullptr, P.GBP, P.drop(t_last, 0), 0, n_last, O, st));
        bool chain_done = false;
        if (want_chain) {
            // the whole recurrence in one launch (xdec.hip, the backward chain); -2 = the shape does not qualify
            XDecBwd x; memset(&x, 0, sizeof(x));
            x.Wow = (const bf16_t*)P.pk(wp, K_OW); x.ldow = P.ldOW;
            x.Wah = (const bf16_t*)P.pk(wp, K_ATT_H); x.ldah = P.ldAH;
            x.Wk = (const bf16_t*)P.pk(wp, K_LSTM) + (size_t)D * P.ldK; x.ldk = P.ldK;
            x.beta = prm + P.poff[P_BETA];
            x.att_img = (const bf16_t*)att_img; x.img = (const bf16_t*)img;
            x.att_exp = P.att_exp() ? P.ws<bf16_t>(ws, W_ATT_EXP) : nullptr;
            x.rec = rec; x.REC = P.REC; x.cs = cs; x.gates = gates; x.atth = atth; x.alpha = alpha; x.Rp = P.Rp;
            x.dolog = dolog; x.gall = gall; x.gb = gb; x.GBP = P.GBP; x.dhc = dhc; x.de = de; x.datth = datth; x.datthb = P.ws<bf16_t>(ws, W_DATTHB);
            x.dz = dz; x.dzb = dzb; x.DZBP = P.DZBP; x.carry_h = carry_h; x.dcc = dcc; x.dxh = dxh;
            x.part = P.ws<float>(ws, W_APART);                       // the forward chain's chunk partials are dead by now
            x.sync = P.ws<unsigned>(ws, W_XSYNC) + kXDecBlockBytes / 4;      // its own block
            x.T = T; x.B = B; x.R = P.R;
            x.dr = P.drop(0, 0);
            LxoTimed tm("xdec_bwd", "chain", (double)T * B * P.R * (E + C) * P.esz, st);
            const int rc = lxo_launch_xdec_bwd(x, U, O, C, E, st);
            if (rc == 0) chain_done = bwd_chain = true;
            else if (rc != -2) return rc < 0 ? rc : -rc;
            else { RC(zero_acc()); HIPRC(hipMemsetAsync(x.sync, 0, kXDecSyncBytes, st)); }      // the shape does not qualify: no tickets, no error
        }
        RStep a; memset(&a, 0, sizeof(a));
        a.U = U; a.O = O; a.zx_row = -1;
        for (int t = t_last; t >= 0 && !chain_done; --t) {
            const int nr = B;
            const int nchb = P.det() ? 1 : P.attn_chunks(nr);      // parity mode: one chunk per sample = one writer per d_att_h element
            a.M = nr;
            const float* rec_cur = rec + (size_t)(t + 1) * B * P.REC;
            float* g_t = gall + (size_t)t * B * O;
            float* dhc_t = dhc + (size_t)t * B * P.HC;
            float* datth_t = datth + (size_t)t * B * E;
            float* dz_t = dz + (size_t)t * B * 4 * U;
            // [d_h~ | d_ctx] = g [o_W_h; o_W_c]^T
            RStep b1 = a;
            b1.A = bf ? (const void*)(gb + (size_t)t * B * P.GBP) : (const void*)g_t; b1.lda = bf ? P.GBP : O;
            b1.W = P.pk(wp, K_OW); b1.ldw = P.ldOW; b1.N = P.HC; b1.K = O; b1.epi = RS_PLAIN;
            b1.out = dhc_t; b1.ldo = P.HC;
            RC(lxo_launch_rstep(P.s.dtype, bf, b1, st));
            const Slabs dc1 = {dhc_t, 1, 0, P.HC};
            {
            LxoTimed tm("attn_bwd", "part", (double)nr * P.R * (E + C) * P.esz, st);
            RC(lxo_k_attn_bwd(P.s.dtype, att_img, P.att_exp() ? P.ws<void>(ws, W_ATT_EXP) : nullptr, img, atth + (size_t)t * B * E, prm + P.poff[P_BETA],
                              alpha + (size_t)t * B * P.Rp, dc1, U, nullptr, P.HC, rec_cur + P.OFF_CTX, P.REC,
                              de + (size_t)t * B * P.Rp, datth_t, nr, P.R, P.Rp, E, C, nchb, att_alternate() ? (t & 1) : 0, st));
            }
            // d_h = (d_h~(o projection) + d_att_h W_att_h^T) * mask + carry -> d_z, d_c
            RStep b3 = a;
            b3.A = datth_t; b3.lda = E;                                  // f32 (atomically accumulated), converted on load
            b3.W = P.pk(wp, K_ATT_H); b3.ldw = P.ldAH; b3.N = U; b3.K = E; b3.epi = RS_LSTM_BWD;
            b3.dhm = dhc_t; b3.lddhm = P.HC; b3.carry_h = carry_h; b3.carry_rows = (t == t_last) ? 0 : B;
            b3.gates_in = gates + (size_t)t * B * 4 * U; b3.c_prev = cs + (size_t)t * B * U; b3.c_cur = cs + (size_t)(t + 1) * B * U;
            b3.dcc = dcc; b3.out = dz_t; b3.outb = bf ? dzb + (size_t)t * B * P.DZBP : nullptr; b3.ldob = P.DZBP; b3.dr = P.drop(t, 0);
            RC(lxo_launch_rstep(P.s.dtype, 0, b3, st));
            // [d_o carry | d_h carry] = d_z K[D:]^T ; g_{t-1} = (d_o(logits) + d_o carry) * tanh'
            RStep b4 = a;
            b4.A = bf ? (const void*)(dzb + (size_t)t * B * P.DZBP) : (const void*)dz_t; b4.lda = bf ? P.DZBP : 4 * U;
            b4.W = (const char*)P.pk(wp, K_LSTM) + (size_t)D * P.ldK * P.esz; b4.ldw = P.ldK; b4.N = P.XH; b4.K = 4 * U; b4.epi = RS_CARRY;
            if (t == 0) { b4.first = 1; b4.out = dxh; b4.ldo = P.XH; }
            else {
                b4.M = B;
                b4.out = gall + (size_t)(t - 1) * B * O; b4.outb = bf ? gb + (size_t)(t - 1) * B * P.GBP : nullptr; b4.ldob = P.GBP; b4.out2 = carry_h;
                b4.dolog = dolog + (size_t)(t - 1) * B * O; b4.o_prev = rec + (size_t)t * B * P.REC; b4.ldoprev = P.REC;
                b4.dr = P.drop(t - 1, 0);
            }
            RC(lxo_launch_rstep(P.s.dtype, bf, b4, st));
        }
    }
    if (dual) RC(fork_side(st));
    for (int t = T - 1; t >= 0 && !fused; --t) {
        for (int h = 0; h < nh; ++h) {
            const int hb = B / nh;
            if (hb <= 0) continue;
            // rows that also ran step t+1 receive its carries; the others end here (their later steps were skipped)
            const int crows = (t == T - 1) ? 0 : hb;
            const int nchb = P.det() ? 1 : P.attn_chunks(hb);
            hipStream_t sh = h ? g_side : st;
            const size_t r0 = (size_t)h * hb;
            float* sb1 = P.ws<float>(ws, W_S_B1) + r0 * (O / 128) * P.HC;
            float* sb3 = P.ws<float>(ws, W_S_B3) + r0 * (E / 128) * U;
            float* sb4 = P.ws<float>(ws, W_S_B4) + r0 * (4 * U / 128) * P.XH;
            const float* rec_cur = rec + ((size_t)(t + 1) * B + r0) * P.REC;
            float* g_t = gall + ((size_t)t * B + r0) * O;
            float* dhc_t = dhc + ((size_t)t * B + r0) * P.HC;
            const char* att_img = (const char*)P.ws<void>(ws, W_ATT_IMG) + r0 * P.R * E * P.esz;
            const char* img = (const char*)P.ws<void>(ws, W_IMG) + r0 * P.R * C * P.esz;
            // carry [d_o | d_h] from step t+1 = the B4 slabs of the previous iteration (none at t = T-1)
            const Slabs carry = (crows <= 0) ? kNoSlabs : view(sb4, 4 * U, crows, P.XH);
            // g = (d_o_logits + d_o_carry) * (1 - o^2)
            const Drop dr = P.drop(t, (int)r0);
            RC(lxo_k_tanh_bwd(dolog + ((size_t)t * B + r0) * O, O, carry, rec_cur, P.REC, g_t, O, nullptr, 0, dr, crows, hb, O, sh));
            // [d_h~ | d_ctx] = g [o_W_h; o_W_c]^T
            RC(slab(P, g_t, O, P.pk(wp, K_OW), P.ldOW, sb1, hb, P.HC, O, sh));
            RC(lxo_k_attn_bwd(P.s.dtype, att_img, P.att_exp() ? (const char*)P.ws<void>(ws, W_ATT_EXP) + r0 * P.R * E * P.esz : nullptr, img, atth + ((size_t)t * B + r0) * E, prm + P.poff[P_BETA],
                              alpha + ((size_t)t * B + r0) * P.Rp, view(sb1, O, hb, P.HC), U, dhc_t + U, P.HC, rec_cur + P.OFF_CTX, P.REC,
                              de + ((size_t)t * B + r0) * P.Rp, datth + ((size_t)t * B + r0) * E, hb, P.R, P.Rp, E, C, nchb, 0, sh));
            // d_h += d_att_h W_att_h^T
            RC(slab(P, datth + ((size_t)t * B + r0) * E, E, P.pk(wp, K_ATT_H), P.ldAH, sb3, hb, U, E, sh));
            RC(lxo_k_lstm_bwd(gates + ((size_t)t * B + r0) * 4 * U, cs + ((size_t)t * B + r0) * U, cs + ((size_t)(t + 1) * B + r0) * U,
                              view(sb1, O, hb, P.HC), view(sb3, E, hb, U), carry, O, dcc + r0 * U, dz + ((size_t)t * B + r0) * 4 * U, dr, crows, hb, U, sh));
            // [d_o carry | d_h carry] = d_z K[D:]^T
            RC(slab(P, dz + ((size_t)t * B + r0) * 4 * U, 4 * U, (const char*)P.pk(wp, K_LSTM) + (size_t)D * P.ldK * P.esz, P.ldK,
                    sb4, hb, P.XH, 4 * U, sh));
        }
    }
    if (dual) RC(join_side(st));
    // the final carries of the two halves are separate slab sets; gather them into one [B][XH] buffer
    for (int h = 0; h < nh && !fused; ++h) {
        const int hb = B / nh;
        const size_t r0 = (size_t)h * hb;
        float* sb4 = P.ws<float>(ws, W_S_B4) + r0 * (4 * U / 128) * P.XH;
        RC(lxo_k_slab_reduce(view(sb4, 4 * U, hb, P.XH), dxh + r0 * P.XH, P.XH, hb, P.XH, st));
    }
    // ---- deferred weight gradients over all steps ----
    hipStream_t side = (P.bf && fused && !lxo_timer_on()) ? lxo_impl_encoder_side_stream() : nullptr;
    const DetScratch det_s = side ? P.det_scratch_side(ws) : det;      // deterministic mode: the side stream's own half of the ordered-partials scratch
    if (side) {
        if (!g_dbw_fork) {
            HIPRC(hipEventCreateWithFlags(&g_dbw_fork, hipEventDisableTiming));
            HIPRC(hipEventCreateWithFlags(&g_dbw_fork2, hipEventDisableTiming));
            HIPRC(hipEventCreateWithFlags(&g_dbw_join, hipEventDisableTiming));
            HIPRC(hipEventCreateWithFlags(&g_dbw_init, hipEventDisableTiming));
        }
        HIPRC(hipEventRecord(g_dbw_fork, st));
        HIPRC(hipStreamWaitEvent(side, g_dbw_fork, 0));
    }
    hipStream_t sd = side ? side : st;
    // ---- initial states (first on the side stream: d_img below needs d_mean; d_mean before the initial projections' own weight gradients) ----
    float* dpre = P.ws<float>(ws, W_DPRE0); float* mean = P.ws<float>(ws, W_MEAN); float* dmean = P.ws<float>(ws, W_DMEAN);
    const int W3 = 2 * U + O;
    { const Slabs one = {dxh, 1, 0, P.XH}; RC(lxo_k_init_bwd(dcc, one, cs, rec, P.REC, dpre, B, U, O, sd)); }
    const char* wi = (const char*)P.pk(wp, K_INIT);
    if (fused_steps(P)) {
        RC(rs_dense(P, dpre, W3, wi, U, dmean, C, B, C, U, nullptr, false, false, sd));
        RC(rs_dense(P, dpre + U, W3, wi + (size_t)C * U * P.esz, U, dmean, C, B, C, U, nullptr, false, true, sd));
        RC(rs_dense(P, dpre + 2 * U, W3, wi + (size_t)2 * C * U * P.esz, O, dmean, C, B, C, O, nullptr, false, true, sd));
    } else {
    RC(nt(P, true, true, true, dpre, W3, wi, U, dmean, C, B, C, U, nullptr, 0, false, sd));
    RC(nt(P, true, true, true, dpre + U, W3, wi + (size_t)C * U * P.esz, U, dmean, C, B, C, U, nullptr, 0, true, sd));
    RC(nt(P, true, true, true, dpre + 2 * U, W3, wi + (size_t)2 * C * U * P.esz, O, dmean, C, B, C, O, nullptr, 0, true, sd));
    }
    if (side) HIPRC(hipEventRecord(g_dbw_init, sd));      // d_mean is complete: the compute stream's d_img waits for THIS, not for the six small launches below
    RC(tn(P, true, true, mean, C, dpre, W3, gw(P_WC0), U, B, C, U, sd));
    RC(tn(P, true, true, mean, C, dpre + U, W3, gw(P_WH0), U, B, C, U, sd));
    RC(tn(P, true, true, mean, C, dpre + 2 * U, W3, gw(P_WO0), O, B, C, O, sd));
    RC(lxo_k_colsum(dpre, W3, gw(P_BC0), B, U, det_s, sd));
    RC(lxo_k_colsum(dpre + U, W3, gw(P_BH0), B, U, det_s, sd));
    RC(lxo_k_colsum(dpre + 2 * U, W3, gw(P_BO0), B, O, det_s, sd));
    if (P.bf && fused) {
        // the bf16 mirrors the step kernels left (record, g_t, d_z_t) are the operands: half the bytes of the f32 originals, and
        // these reductions over T*B rows are bound by operand re-reads (every 128 x 128 tile walks all rows of both operands)
        const bf16_t* recb = P.ws<bf16_t>(ws, W_RECB);
        const bf16_t* gb = P.ws<bf16_t>(ws, W_GB); const bf16_t* dzb = P.ws<bf16_t>(ws, W_DZB);
        RC(tn(P, false, false, recb + (size_t)B * P.RECB + P.OFF_HT, P.RECB, gb, P.GBP, gw(P_OWH), O, TB, P.HC, O, sd, det_s));      // d[o_W_h; o_W_c]
        if (bwd_chain) RC(tn(P, false, false, recb + (size_t)B * P.RECB + P.OFF_HT, P.RECB, P.ws<bf16_t>(ws, W_DATTHB), E, gw(P_ATT_H), E, TB, U, E, sd, det_s));      // dW_att_h
        else RC(tn(P, false, true, recb + (size_t)B * P.RECB + P.OFF_HT, P.RECB, datth, E, gw(P_ATT_H), E, TB, U, E, sd));
        RC(tn(P, false, false, P.ws<void>(ws, W_EMB_IN), P.Dp, dzb, P.DZBP, gw(P_LSTM_K), 4 * U, TB, D, 4 * U, sd, det_s));           // dK rows 0..D
        RC(tn(P, false, false, recb, P.RECB, dzb, P.DZBP, gw(P_LSTM_K) + (size_t)D * 4 * U, 4 * U, TB, P.XH, 4 * U, sd, det_s));     // dK rows D..
    } else {
    RC(tn(P, true, true, rec + (size_t)B * P.REC + P.OFF_HT, P.REC, gall, O, gw(P_OWH), O, TB, P.HC, O, st));      // d[o_W_h; o_W_c]
    RC(tn(P, true, true, rec + (size_t)B * P.REC + P.OFF_HT, P.REC, datth, E, gw(P_ATT_H), E, TB, U, E, st));      // dW_att_h
    RC(tn(P, false, true, P.ws<void>(ws, W_EMB_IN), P.Dp, dz, 4 * U, gw(P_LSTM_K), 4 * U, TB, D, 4 * U, st)); // dK rows 0..D
    RC(tn(P, true, true, rec, P.REC, dz, 4 * U, gw(P_LSTM_K) + (size_t)D * 4 * U, 4 * U, TB, P.XH, 4 * U, st)); // dK rows D..
    }
    RC(lxo_k_colsum(dz, 4 * U, gw(P_LSTM_B), TB, 4 * U, det_s, sd));
    // embeddings
    float* demb = P.ws<float>(ws, W_DEMB);
    if (fused) {   // d_emb = d_z K[0:D]^T over all T*B rows: a tall GEMM on the step kernel (505 workgroups; the bf16 mirror of d_z halves its bytes)
        RStep e; memset(&e, 0, sizeof(e));
        e.M = TB; e.N = D; e.K = 4 * U; e.U = U; e.O = O; e.zx_row = -1; e.epi = RS_PLAIN; e.dr.inv_keep = 1.f;
        e.A = P.bf ? (const void*)P.ws<bf16_t>(ws, W_DZB) : (const void*)dz; e.lda = P.bf ? P.DZBP : 4 * U;
        e.W = P.pk(wp, K_LSTM); e.ldw = P.ldK; e.out = demb; e.ldo = D;
        RC(lxo_launch_rstep(P.s.dtype, P.bf, e, sd));
    } else
    RC(nt(P, true, true, true, dz, 4 * U, P.pk(wp, K_LSTM), P.ldK, demb, D, TB, D, 4 * U, nullptr, 0, false, st));
    RC(lxo_k_embed_scatter(demb, formula, gw(P_EMB), gw(P_START), B, T, D, V, P.det() ? 1 : 0, sd));
    // ---- d_img = sum_t alpha_t (x) d_ctx_t  (batched over samples)  + d_mean / R + d_att_img W_att_img^T ----
    float* dimg = P.ws<float>(ws, W_DIMG);
    if (P.dimg_masked()) {
        // one batched GEMM contracts both products, adds the mean gradient and applies conv6's ReLU mask + bias-gradient sum
        // in its epilogue (dimg.hip): region "d_img" receives d_y6 in the compute dtype, lxo_encoder_bwd skips its mask pass
        RC(lxo_k_datt_img(P.s.dtype, P.ws<void>(ws, W_ATT_IMG), atth, prm + P.poff[P_BETA], de, P.ws<void>(ws, W_DATTIMG), gw(P_BETA),
                          T, B, P.R, P.Rp, E, det, st));
        if (side) { HIPRC(hipEventRecord(g_dbw_fork2, st)); HIPRC(hipStreamWaitEvent(st, g_dbw_init, 0)); }      // d_mean comes from the side stream
        DimgArgs a; memset(&a, 0, sizeof(a));
        a.alpha = alpha; a.ld_alpha = (long long)B * P.Rp; a.Rp = P.Rp;
        a.dctx = dhc + U; a.ld_dctx = (long long)B * P.HC; a.HC = P.HC;
        a.datt = P.ws<bf16_t>(ws, W_DATTIMG); a.W = (const bf16_t*)P.pk(wp, K_ATT_IMG); a.ldw = E;
        a.dmean = dmean; a.T = T; a.B = B; a.R = P.R; a.C = C; a.E = E;
        a.y6 = P.ws<bf16_t>(ws, W_Y6); a.dy6 = P.ws<bf16_t>(ws, W_DIMG); a.db = gw(P_CONV6_B);
        if (P.det()) { a.db_part = det.p; a.db_part_floats = det.floats; }      // deterministic mode: per-workgroup slots, added in order
        RC(lxo_launch_dimg_fused(a, st));
    } else {
        {
            GemmTN g; memset(&g, 0, sizeof(g));
            g.A = alpha; g.B = dhc + U; g.C = dimg; g.M = T; g.I = P.R; g.J = C;
            g.lda = B * P.Rp; g.ldb = B * P.HC; g.ldc = C;
            g.nsplit = 1; g.nbatch = B; g.strideA = P.Rp; g.strideB = P.HC; g.strideC = (long long)P.R * C; g.atomic = 0;
            RC(lxo_launch_gemm_tn(P.s.dtype, 1, 1, g, st));
        }
        if (side) HIPRC(hipStreamWaitEvent(st, g_dbw_init, 0));      // d_mean comes from the side stream (init_bwd / rs_dense above): before its first reader
        RC(lxo_k_add_mean_grad(dimg, dmean, B, P.R, C, st));
        RC(lxo_k_datt_img(P.s.dtype, P.ws<void>(ws, W_ATT_IMG), atth, prm + P.poff[P_BETA], de, P.ws<void>(ws, W_DATTIMG), gw(P_BETA),
                          T, B, P.R, P.Rp, E, det, st));
        if (side) HIPRC(hipEventRecord(g_dbw_fork2, st));
        RC(nt(P, false, true, false, P.ws<void>(ws, W_DATTIMG), E, P.pk(wp, K_ATT_IMG), E, dimg, C, B * P.R, C, E, nullptr, 0, true, st));
    }
    if (side) HIPRC(hipStreamWaitEvent(side, g_dbw_fork2, 0));      // dW_att_img needs d_att_img, nothing needs dW_att_img
    RC(tn(P, false, false, P.ws<void>(ws, W_IMG), C, P.ws<void>(ws, W_DATTIMG), E, gw(P_ATT_IMG), E, B * P.R, C, E, sd, det_s));
    // lxo_train_bwd (defer_join): the side stream's work is NOT joined here -- lxo_impl_encoder_bwd follows on the same two streams and
    // joins them once, at its end; the deferred gradients then also run beside conv6's data gradient instead of holding the main stream
    if (side && !defer_join) {
        HIPRC(hipEventRecord(g_dbw_join, side));
        HIPRC(hipStreamWaitEvent(st, g_dbw_join, 0));
    }
    // the backward chain's error word -> the probe element (as for the forward chain above; y_W_o's bucket is reduced behind this call
    // where the backward chain runs, Engine.backward)
    if (bwd_chain) RC(lxo_k_chain_poison(nullptr, P.ws<unsigned>(ws, W_XSYNC) + kXDecBlockBytes / 4 + 8 * 64, grads + P.ptotal - 1, st));
    // `ready`: every decoder gradient (and the probe element) is final -- recorded on the side stream behind both streams' work when
    // one is in use, so that a communication stream can wait for it without the main stream stopping
    if (ready) {
        if (side) {
            HIPRC(hipEventRecord(g_dbw_fork, st));
            HIPRC(hipStreamWaitEvent(side, g_dbw_fork, 0));
            HIPRC(hipEventRecord((hipEvent_t)ready, side));
        } else HIPRC(hipEventRecord((hipEvent_t)ready, st));
    }
    return 0;
}

// ------------------------------------------------------------------ decode ----
// x-part of the LSTM pre-activation for every possible input token, once per decode call: row v = embedding_table[v] K[0:D] + b,
// row V = the start token's.  The step kernels then pick rows by the previous ids (no per-step gather + GEMM launches).
static int decode_token_table(const Plan& P, const float* prm, const void* wp, void* ws, hipStream_t st) {
    const int U = P.s.U, D = P.s.D, V = P.s.V;
    RC(lxo_k_embed_table(P.s.dtype, prm + P.poff[P_EMB], prm + P.poff[P_START], P.ws<void>(ws, W_DEC_TXE), V, D, P.Dp, st));
    RC(nt(P, false, true, false, P.ws<void>(ws, W_DEC_TXE), P.Dp, P.pk(wp, K_LSTM_XT), P.Dp, P.ws<float>(ws, W_DEC_TX), 4 * U, V + 1, 4 * U, P.Dp,
          prm + P.poff[P_LSTM_B], 0, false, st));
    return 0;
}
static int decode_common_step(const Plan& P, const float* prm, const void* wp, void* ws, int nv, int beam, int cur, const int* ids_prev, hipStream_t st, const int* a_par = nullptr) {
    const int U = P.s.U, O = P.s.O, D = P.s.D, V = P.s.V;
    float* rec = P.ws<float>(ws, W_REC); float* cs = P.ws<float>(ws, W_CS);
    const int prev = cur ^ 1;
    if (fused_steps(P)) {
        bf16_t* recb = P.ws<bf16_t>(ws, W_RECB);
        RC(cell_step_fused(P, prm, wp, ws, nv, beam, P.ws<float>(ws, W_DEC_TX), rec + (size_t)prev * nv * P.REC, cs + (size_t)prev * nv * U,
                           rec + (size_t)cur * nv * P.REC, cs + (size_t)cur * nv * U,
                           recb + (size_t)prev * nv * P.RECB, recb + (size_t)cur * nv * P.RECB, nullptr,
                           P.ws<float>(ws, W_ATTH), P.ws<float>(ws, W_ALPHA), Drop{0u, 1.f, 0u, cur, 0, 0}, st,      // t = the record slot: only its parity is used (attention direction)
                           ids_prev, ids_prev ? -1 : V, ids_prev ? a_par : nullptr));
    } else {
        // next input embedding (start token at time 0), its LSTM x-part, then the cell step
        float* zx = P.ws<float>(ws, W_DEC_ZX);
        RC(lxo_k_embed_rows(P.s.dtype, prm + P.poff[P_EMB], prm + P.poff[P_START], ids_prev, P.ws<void>(ws, W_DEC_EMB), nv, D, P.Dp, V, st));
        RC(nt(P, false, true, false, P.ws<void>(ws, W_DEC_EMB), P.Dp, P.pk(wp, K_LSTM_XT), P.Dp, zx, 4 * U, nv, 4 * U, P.Dp,
              prm + P.poff[P_LSTM_B], 0, false, st));
        RC(cell_step(P, prm, wp, ws, 0, nv, beam, zx, rec + (size_t)prev * nv * P.REC, cs + (size_t)prev * nv * U,
                     rec + (size_t)cur * nv * P.REC, cs + (size_t)cur * nv * U, nullptr,
                     P.ws<float>(ws, W_ATTH), P.ws<float>(ws, W_ALPHA), Drop{0u, 1.f, 0u, 0, 0, 0}, st));
    }
    // logits = o y_W_o (attention_cell.py:84).  On the step kernel where it runs: one workgroup per 16 vocabulary columns and 16 / 64 rows
    // (128 .. 160 workgroups); the dense-GEMM tiles gave 8 (greedy, 64 rows) or 12 (beam 5, 320 rows) workgroups -- 19 us of a beam step
    if (fused_steps(P) && V % 4 == 0) {
        RStep e; memset(&e, 0, sizeof(e));
        e.M = nv; e.N = V; e.K = O; e.U = U; e.O = O; e.zx_row = -1; e.epi = RS_PLAIN; e.dr.inv_keep = 1.f;
        e.A = P.bf ? (const void*)(P.ws<bf16_t>(ws, W_RECB) + (size_t)cur * nv * P.RECB) : (const void*)(rec + (size_t)cur * nv * P.REC);
        e.lda = P.bf ? P.RECB : P.REC;
        e.W = P.pk(wp, K_YWO_T); e.ldw = O; e.out = P.ws<float>(ws, W_DEC_LOGITS); e.ldo = P.Vp;
        const int rc = lxo_launch_rstep(P.s.dtype, P.bf, e, st);
        if (rc != -2) return rc < 0 ? rc : -rc;
    }
    RC(nt(P, true, true, nv <= 64, rec + (size_t)cur * nv * P.REC, P.REC, P.pk(wp, K_YWO_T), O, P.ws<float>(ws, W_DEC_LOGITS), P.Vp,
          nv, V, O, nullptr, 0, false, st));
    return 0;
}

// Host side of dynamic_decode's `while not all(finished)` (dynamic_decode.py:38-61): steps are enqueued in chunks of CHUNK;
// each chunk ends with an asynchronous copy of its per-step "rows still unfinished" counters into pinned host memory and
// an event.  The host enqueues chunk c + 1 BEFORE it waits for chunk c's event, so the stream never drains while the
// host looks at the flags (the round-1 loop synchronised the stream every 8 steps); if chunk c turns out to contain the
// last step, chunk c + 1 ran speculatively (its ids land beyond `steps` columns, which no caller reads).
namespace {
struct DecodePoll { int* host; hipEvent_t ev[2]; bool ok; };
thread_local DecodePoll g_poll = {nullptr, {nullptr, nullptr}, false};
int poll_init() {
    if (g_poll.ok) return 0;
    HIPRC(hipHostMalloc((void**)&g_poll.host, 2 * 64 * sizeof(int), 0));      // one-time 512-byte pinned buffer per host thread
    HIPRC(hipEventCreateWithFlags(&g_poll.ev[0], hipEventDisableTiming));
    HIPRC(hipEventCreateWithFlags(&g_poll.ev[1], hipEventDisableTiming));
    g_poll.ok = true;
    return 0;
}
}  // namespace
template <typename StepFn>
static int decode_loop(int max_iter, int* flags, hipStream_t st, int* steps_out, StepFn step) {
    RC(poll_init());
    const int CHUNK = 8;
    int enq = 0;                 // steps enqueued so far
    int nchunks = 0;             // chunks enqueued
    int steps = -1;              // final step count once known
    auto enqueue_chunk = [&]() -> int {
        const int slot = nchunks & 1;
        int* dflags = flags + slot * 32;                      // device counters of this chunk (flags[0..63]: two slots of 32)
        HIPRC(hipMemsetAsync(dflags, 0, CHUNK * sizeof(int), st));
        int n = 0;
        for (; n < CHUNK && enq <= max_iter; ++n, ++enq) RC(step(enq, dflags + n));
        HIPRC(hipMemcpyAsync(g_poll.host + slot * 64, dflags, CHUNK * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPRC(hipEventRecord(g_poll.ev[slot], st));
        g_poll.host[slot * 64 + 32] = n;                      // steps in this chunk
        g_poll.host[slot * 64 + 33] = enq - n;                // first step of this chunk
        ++nchunks;
        return 0;
    };
    RC(enqueue_chunk());
    int checked = 0;
    while (steps < 0) {
        if (enq <= max_iter) RC(enqueue_chunk());             // speculative: keeps the stream busy while the host polls
        const int slot = checked & 1;
        HIPRC(hipEventSynchronize(g_poll.ev[slot]));
        const int n = g_poll.host[slot * 64 + 32], first = g_poll.host[slot * 64 + 33];
        for (int c = 0; c < n; ++c) {
            // dynamic_decode.py:38-51: stop after the first step that leaves nothing unfinished, or after step max_iter
            if (g_poll.host[slot * 64 + c] == 0 || first + c >= max_iter) { steps = first + c + 1; break; }
        }
        ++checked;
        if (steps < 0 && checked == nchunks && enq > max_iter) steps = enq;      // cannot happen (the last step hits the bound); belt and braces
    }
    if (nchunks > checked) HIPRC(hipEventSynchronize(g_poll.ev[(nchunks - 1) & 1]));   // the speculative chunk must not outlive the call's buffers
    if (steps_out) *steps_out = steps;
    return 0;
}

// The same host loop at CHUNK granularity: one call of `chunk(first_step, n_steps, device counters)` enqueues n_steps steps (the persistent
// greedy-decode chain runs a chunk as ONE launch).
template <typename ChunkFn>
static int decode_loop_chunked(int max_iter, int CHUNK, int* flags, hipStream_t st, int* steps_out, ChunkFn chunk) {
    RC(poll_init());
    if (CHUNK > 32) CHUNK = 32;
    int enq = 0, nchunks = 0, steps = -1;
    auto enqueue_chunk = [&]() -> int {
        const int slot = nchunks & 1;
        int* dflags = flags + slot * 32;
        HIPRC(hipMemsetAsync(dflags, 0, 32 * sizeof(int), st));
        int n = max_iter + 1 - enq; if (n > CHUNK) n = CHUNK;
        RC(chunk(enq, n, dflags));
        enq += n;
        HIPRC(hipMemcpyAsync(g_poll.host + slot * 64, dflags, 32 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPRC(hipEventRecord(g_poll.ev[slot], st));
        g_poll.host[slot * 64 + 32] = n;
        g_poll.host[slot * 64 + 33] = enq - n;
        ++nchunks;
        return 0;
    };
    RC(enqueue_chunk());
    int checked = 0;
    while (steps < 0) {
        if (enq <= max_iter) RC(enqueue_chunk());
        const int slot = checked & 1;
        HIPRC(hipEventSynchronize(g_poll.ev[slot]));
        const int n = g_poll.host[slot * 64 + 32], first = g_poll.host[slot * 64 + 33];
        for (int c = 0; c < n; ++c)
            if ((g_poll.host[slot * 64 + c] & 0xffff) == 0 || first + c >= max_iter) { steps = first + c + 1; break; }      // low 16 bits: unfinished rows (high bits: chains that reported)
        ++checked;
        if (steps < 0 && checked == nchunks && enq > max_iter) steps = enq;
    }
    if (nchunks > checked) HIPRC(hipEventSynchronize(g_poll.ev[(nchunks - 1) & 1]));
    if (steps_out) *steps_out = steps;
    return 0;
}

int lxo_impl_greedy_decode(const Plan& P, const float* prm, const void* wp, void* ws, int id_end, int max_iter,
                           int* ids_out, float* alpha_out, int* steps_out, hipStream_t st) {
    const int B = P.s.B, ms = P.s.max_steps;
    if (ms < max_iter + 1) return -5;
    RC(attention_prepare(P, prm, wp, ws, 1, st));
    if (P.att_exp()) RC(lxo_k_att_exp(P.ws<void>(ws, W_ATT_IMG), P.ws<void>(ws, W_ATT_EXP), (long long)B * P.R * P.s.E, st));      // bf16: E_x = e^{2 att_img}, once per call
    if (fused_steps(P)) RC(mirror_oh(P, ws, 0, B, st));
    int* flags = P.ws<int>(ws, W_DEC_FLAGS);          // [0..63]: per-step unfinished counters ; [64..]: finished[B]
    int* finished = flags + 64;
    int* ids_step = P.ws<int>(ws, W_DEC_IDS);
    HIPRC(hipMemsetAsync(flags, 0, 256 + (size_t)B * 4, st));
    // rec/cs slot 0 holds the initial state; slots alternate
    if (fused_steps(P)) RC(decode_token_table(P, prm, wp, ws, st));
    if (fused_steps(P) && P.bf && P.att_exp() && !alpha_out && P.s.step_kernels == 0) {
        // the persistent greedy-decode chain (xdec.hip: xdec_dec_kernel): 16 steps per launch; -2 = the shape does not qualify
        XDecDec x; memset(&x, 0, sizeof(x));
        x.Wrt = (const bf16_t*)P.pk(wp, K_LSTM_RT); x.ldrt = P.ldRT;
        x.Wah = (const bf16_t*)P.pk(wp, K_ATT_H_T); x.ldah = P.ldAHT;
        x.Wow = (const bf16_t*)P.pk(wp, K_OW_T); x.ldow = P.ldOWT;
        x.Wyo = (const bf16_t*)P.pk(wp, K_YWO_T); x.ldyo = P.s.O;
        x.beta = prm + P.poff[P_BETA];
        x.att_exp = P.ws<bf16_t>(ws, W_ATT_EXP); x.img = P.ws<bf16_t>(ws, W_IMG);
        x.tx = P.ws<float>(ws, W_DEC_TX);
        x.rec = P.ws<float>(ws, W_REC); x.recb = P.ws<bf16_t>(ws, W_RECB); x.cs = P.ws<float>(ws, W_CS);
        x.part = P.ws<float>(ws, W_APART); x.sync = P.ws<unsigned>(ws, W_XSYNC);
        x.ids_step = ids_step; x.ids_out = ids_out; x.finished = finished;
        x.B = B; x.R = P.R; x.REC = P.REC; x.RECB = P.RECB; x.V = P.s.V; x.id_end = id_end; x.max_steps = ms;
        x.t0 = 0; x.nsteps = 1; x.unfinished = flags;
        x.stop = ids_step + B;                            // one word behind the fed-back ids (region "dec_ids" holds B x max_steps ints)
        HIPRC(hipMemsetAsync(x.stop, 0, sizeof(int), st));
        HIPRC(hipMemsetAsync(P.ws<unsigned>(ws, W_XSYNC) + 8 * 64, 0, sizeof(unsigned), st));      // the error word: once per decode (the launcher leaves it alone)
        int chunk_steps = 16;                             // steps per launch (LXO_XDEC_DEC_CHUNK: 1 .. 16; read per call so that a test can vary it)
        { const char* e = getenv("LXO_XDEC_DEC_CHUNK"); if (e && atoi(e) > 0 && atoi(e) < 16) chunk_steps = atoi(e); }
        bool took = true;
        const int rc = decode_loop_chunked(max_iter, chunk_steps, flags, st, steps_out, [&](int first, int n, int* unfinished) -> int {
            XDecDec y = x; y.t0 = first; y.nsteps = n; y.unfinished = unfinished;
            const int r = lxo_launch_xdec_dec(y, P.s.U, P.s.O, P.s.C, P.s.E, st);
            if (r == -2 && first == 0) { took = false; return -2; }
            return r;
        });
        if (took) {
            if (rc) return rc;
            // the chain's error word (a hand-over that timed out: the ids are garbage).  The call has synchronised with the stream already.
            unsigned errw = 0;
            HIPRC(hipMemcpyAsync(&errw, P.ws<unsigned>(ws, W_XSYNC) + 8 * 64, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIPRC(hipStreamSynchronize(st));
            if (errw == 0) return 0;
            // fall back to the launch-per-step kernels: the initial state, the finished flags and the token table are rebuilt first
            RC(attention_prepare(P, prm, wp, ws, 1, st));
            if (fused_steps(P)) RC(mirror_oh(P, ws, 0, B, st));
        } else {
            HIPRC(hipStreamSynchronize(st));              // (nothing was enqueued by the refused first chunk but its counter memset)
            HIPRC(hipMemsetAsync(P.ws<unsigned>(ws, W_XSYNC), 0, 8 * 64 * 4, st));      // no chain in this call: no tickets (Engine.chain_status reads them)
        }
        HIPRC(hipMemsetAsync(flags, 0, 256 + (size_t)B * 4, st));
    } else if (P.bf) HIPRC(hipMemsetAsync(P.ws<unsigned>(ws, W_XSYNC), 0, 8 * 64 * 4, st));
    RC(decode_loop(max_iter, flags, st, steps_out, [&](int time, int* unfinished) -> int {
        const int cur = (time + 1) & 1;
        RC(decode_common_step(P, prm, wp, ws, B, 1, cur, time == 0 ? nullptr : ids_step, st));
        if (alpha_out)      // attention weights of this step (what attention_mechanism.py:96-105 hands to its py_func hook)
            HIPRC(hipMemcpyAsync(alpha_out + (size_t)time * B * P.Rp, P.ws<float>(ws, W_ALPHA), (size_t)B * P.Rp * 4, hipMemcpyDeviceToDevice, st));
        RC(lxo_k_argmax(P.ws<float>(ws, W_DEC_LOGITS), P.Vp, P.s.V, B, id_end, ids_step, ids_out, ms, time, finished, unfinished, st));
        return 0;
    }));
    return 0;
}

// lxo_chain_guard: scale[0] = NaN when a chain of this step left an error word, else the clip scale / 1 (decoder_kernels.hip)
int lxo_impl_chain_guard(const Plan& P, void* ws, const float* grads, float* scale, int have_scale, unsigned* status, hipStream_t st) {
    const unsigned* ef = P.bf ? P.ws<unsigned>(ws, W_XSYNC) + 8 * 64 : nullptr;
    const unsigned* eb = P.bf ? P.ws<unsigned>(ws, W_XSYNC) + kXDecBlockBytes / 4 + 8 * 64 : nullptr;
    return lxo_k_chain_guard(ef, eb, grads ? grads + P.ptotal - 1 : nullptr, scale, have_scale, status, st);
}

// ---- AttentionState of the step-wise decode (attention_cell.py:8: cell_state = LSTMStateTuple(c, h), o): the state lxo_decode_step(time) /
// lxo_decode_cell_step(time) steps FROM lives in record slot time & 1 ----
static int state_rows(const Plan& P) { return P.s.B * (P.s.beam > 1 ? P.s.beam : 1); }
int lxo_impl_decode_state_get(const Plan& P, void* ws, int time, float* c, float* h, float* o, hipStream_t st) {
    const int nv = state_rows(P), U = P.s.U, O = P.s.O, slot = time & 1;
    const float* rec = P.ws<float>(ws, W_REC) + (size_t)slot * nv * P.REC;
    const float* cs = P.ws<float>(ws, W_CS) + (size_t)slot * nv * U;
    if (c) HIPRC(hipMemcpyAsync(c, cs, (size_t)nv * U * 4, hipMemcpyDeviceToDevice, st));
    if (h) HIPRC(hipMemcpy2DAsync(h, (size_t)U * 4, rec + O, (size_t)P.REC * 4, (size_t)U * 4, nv, hipMemcpyDeviceToDevice, st));
    if (o) HIPRC(hipMemcpy2DAsync(o, (size_t)O * 4, rec, (size_t)P.REC * 4, (size_t)O * 4, nv, hipMemcpyDeviceToDevice, st));
    return 0;
}
int lxo_impl_decode_state_set(const Plan& P, void* ws, int time, const float* c, const float* h, const float* o, const int* ids_prev, hipStream_t st) {
    const int nv = state_rows(P), U = P.s.U, O = P.s.O, slot = time & 1;
    float* rec = P.ws<float>(ws, W_REC) + (size_t)slot * nv * P.REC;
    float* cs = P.ws<float>(ws, W_CS) + (size_t)slot * nv * U;
    if (c) HIPRC(hipMemcpyAsync(cs, c, (size_t)nv * U * 4, hipMemcpyDeviceToDevice, st));
    if (h) HIPRC(hipMemcpy2DAsync(rec + O, (size_t)P.REC * 4, h, (size_t)U * 4, (size_t)U * 4, nv, hipMemcpyDeviceToDevice, st));
    if (o) HIPRC(hipMemcpy2DAsync(rec, (size_t)P.REC * 4, o, (size_t)O * 4, (size_t)O * 4, nv, hipMemcpyDeviceToDevice, st));
    if ((h || o) && fused_steps(P)) RC(mirror_oh(P, ws, (size_t)slot * nv, nv, st));      // the step GEMMs read the bf16 mirror of [o | h]
    if (ids_prev) HIPRC(hipMemcpyAsync(P.ws<int>(ws, W_DEC_IDS), ids_prev, (size_t)nv * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}
// AttentionCell.step alone (attention_cell.py:58-89): state slot time & 1 -> slot (time + 1) & 1, logits in ws region "dec_logits";
// no arg-max, no finished flags (those belong to the decoder cells: lxo_decode_step)
int lxo_impl_decode_cell_step(const Plan& P, const float* prm, const void* wp, void* ws, int time, int start_token, hipStream_t st) {
    const int k = P.s.beam > 1 ? P.s.beam : 1;
    if (time < 0) return -5;
    return decode_common_step(P, prm, wp, ws, state_rows(P), k, (time + 1) & 1, start_token ? nullptr : P.ws<int>(ws, W_DEC_IDS), st);
}

// ---- the decode loop one step at a time: what the reference's cell protocol (dynamic_decode.py:34-61: initialize / step /
// finalize) is bound to.  State (c, h, o, running log-probs, finished flags, previous ids) stays in the workspace. ----
int lxo_impl_decode_begin(const Plan& P, const float* prm, const void* wp, void* ws, hipStream_t st) {
    const int B = P.s.B, k = P.s.beam > 1 ? P.s.beam : 1, nv = B * k;
    if (P.s.max_steps < 1 || k > 16) return -5;
    RC(attention_prepare(P, prm, wp, ws, k, st));
    if (P.att_exp()) RC(lxo_k_att_exp(P.ws<void>(ws, W_ATT_IMG), P.ws<void>(ws, W_ATT_EXP), (long long)B * P.R * P.s.E, st));      // bf16 decode: E_x = e^{2 att_img}, once per call
    if (fused_steps(P)) RC(mirror_oh(P, ws, 0, nv, st));
    HIPRC(hipMemsetAsync(P.ws<int>(ws, W_DEC_FLAGS), 0, 256 + (size_t)nv * 4, st));
    if (k > 1) HIPRC(hipMemsetAsync(P.ws<float>(ws, W_BEAM_LP), 0, (size_t)nv * 4, st));
    if (fused_steps(P)) RC(decode_token_table(P, prm, wp, ws, st));
    return 0;
}

int lxo_impl_decode_step(const Plan& P, const float* prm, const void* wp, void* ws, int id_end, int time,
                         int* ids_out, int* parents_out, int* finished_out, int* unfinished_host, hipStream_t st) {
    const int B = P.s.B, k = P.s.beam > 1 ? P.s.beam : 1, nv = B * k, ms = P.s.max_steps, U = P.s.U;
    if (time < 0 || time >= ms) return -5;
    int* flags = P.ws<int>(ws, W_DEC_FLAGS);
    int* finished = flags + 64;
    int* ids_step = P.ws<int>(ws, W_DEC_IDS);
    HIPRC(hipMemsetAsync(flags, 0, sizeof(int), st));
    const int cur = (time + 1) & 1;
    RC(decode_common_step(P, prm, wp, ws, nv, k, cur, time == 0 ? nullptr : ids_step, st));
    if (k == 1) {
        RC(lxo_k_argmax(P.ws<float>(ws, W_DEC_LOGITS), P.Vp, P.s.V, B, id_end, ids_step, ids_out, ms, time, finished, flags, st));
    } else {
        int* par_step = P.ws<int>(ws, W_BEAM_PAR);
        float* tmp = P.ws<float>(ws, W_BEAM_TMP);
        float* rec = P.ws<float>(ws, W_REC); float* cs = P.ws<float>(ws, W_CS);
        RC(lxo_k_beam_step(P.ws<float>(ws, W_DEC_LOGITS), P.Vp, P.s.V, B, k, id_end, time, P.s.div_gamma, P.s.div_prob, P.s.div_seed, tmp,
                           P.ws<float>(ws, W_BEAM_LP), finished, ids_step, par_step, ids_out, parents_out, ms, flags, st));
        RC(lxo_k_beam_gather(rec + (size_t)cur * nv * P.REC, P.REC, P.XH, cs + (size_t)cur * nv * U, U, par_step, k,
                             tmp, tmp + (size_t)nv * P.XH, nv,
                             (fused_steps(P) && P.bf) ? P.ws<bf16_t>(ws, W_RECB) + (size_t)cur * nv * P.RECB : nullptr, P.RECB, st));      // + the bf16 mirror of the re-ordered [o | h] rows
    }
    if (finished_out) HIPRC(hipMemcpyAsync(finished_out, finished, (size_t)nv * 4, hipMemcpyDeviceToHost, st));
    if (unfinished_host) {
        HIPRC(hipMemcpyAsync(unfinished_host, flags, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPRC(hipStreamSynchronize(st));
    }
    return 0;
}

int lxo_impl_beam_decode(const Plan& P, const float* prm, const void* wp, void* ws, int id_end, int max_iter,
                         int* ids_out, int* parents_out, float* alpha_out, int* steps_out, hipStream_t st) {
    const int B = P.s.B, k = P.s.beam, ms = P.s.max_steps, nv = B * k, U = P.s.U;
    if (ms < max_iter + 1 || k < 1 || k > 16) return -5;
    RC(attention_prepare(P, prm, wp, ws, k, st));
    if (P.att_exp()) RC(lxo_k_att_exp(P.ws<void>(ws, W_ATT_IMG), P.ws<void>(ws, W_ATT_EXP), (long long)B * P.R * P.s.E, st));      // bf16: E_x = e^{2 att_img}, once per call
    if (fused_steps(P)) RC(mirror_oh(P, ws, 0, nv, st));
    int* flags = P.ws<int>(ws, W_DEC_FLAGS);
    int* finished = flags + 64;
    int* ids_step = P.ws<int>(ws, W_DEC_IDS);
    int* par_step = P.ws<int>(ws, W_BEAM_PAR);
    float* logp = P.ws<float>(ws, W_BEAM_LP);
    float* tmp = P.ws<float>(ws, W_BEAM_TMP);
    HIPRC(hipMemsetAsync(flags, 0, 256 + (size_t)nv * 4, st));
    HIPRC(hipMemsetAsync(logp, 0, (size_t)nv * 4, st));
    float* rec = P.ws<float>(ws, W_REC); float* cs = P.ws<float>(ws, W_CS);
    if (fused_steps(P)) RC(decode_token_table(P, prm, wp, ws, st));
    // the state of a step's rows is that of their PARENT hypotheses (beam_search_decoder_cell.py:176-178).  With the fused step kernels the next LSTM launch
    // reads its [o | h] and c rows through the parents in place; the launch that re-ordered the rows (beam_permute_kernel, 5.3 us + its gap) is only left for
    // the split-K step kernels and for lxo_decode_step, whose callers may look at the state between steps (LXO_BEAM_INDIRECT=0: always re-order; A/B)
    static int indirect_on = -1;
    if (indirect_on < 0) { const char* e = getenv("LXO_BEAM_INDIRECT"); indirect_on = (e && e[0] == '0') ? 0 : 1; }
    const bool indirect = indirect_on && fused_steps(P);
    RC(decode_loop(max_iter, flags, st, steps_out, [&](int time, int* unfinished) -> int {
        const int cur = (time + 1) & 1;
        RC(decode_common_step(P, prm, wp, ws, nv, k, cur, time == 0 ? nullptr : ids_step, st, indirect ? par_step : nullptr));
        if (alpha_out)      // the attention weights of this step's B x k decoder rows, as they ran (row b * k + j = hypothesis slot j BEFORE this step's
                            // re-ordering: what the reference's py_func tap sees on the merged batch x beam rows, attention_mechanism.py:59-65,96-105)
            HIPRC(hipMemcpyAsync(alpha_out + (size_t)time * nv * P.Rp, P.ws<float>(ws, W_ALPHA), (size_t)nv * P.Rp * 4, hipMemcpyDeviceToDevice, st));
        RC(lxo_k_beam_step(P.ws<float>(ws, W_DEC_LOGITS), P.Vp, P.s.V, B, k, id_end, time, P.s.div_gamma, P.s.div_prob, P.s.div_seed, tmp,
                           logp, finished, ids_step, par_step,
                           ids_out, parents_out, ms, unfinished, st));
        if (!indirect)
        RC(lxo_k_beam_gather(rec + (size_t)cur * nv * P.REC, P.REC, P.XH, cs + (size_t)cur * nv * U, U, par_step, k,
                             tmp, tmp + (size_t)nv * P.XH, nv,
                             (fused_steps(P) && P.bf) ? P.ws<bf16_t>(ws, W_RECB) + (size_t)cur * nv * P.RECB : nullptr, P.RECB, st));    // the re-ordered [o | h] rows (+ their bf16 mirror) feed the next LSTM GEMM
        return 0;
    }));
    return 0;
}
