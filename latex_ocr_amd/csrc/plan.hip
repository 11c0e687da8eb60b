// Host-side plan arithmetic (see plan.h).
#include "plan.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static const char* kParamNames[P_COUNT] = {
    "Encoder/convolutional_encoder/conv2d/kernel", "Encoder/convolutional_encoder/conv2d/bias",
    "Encoder/convolutional_encoder/conv2d_1/kernel", "Encoder/convolutional_encoder/conv2d_1/bias",
    "Encoder/convolutional_encoder/conv2d_2/kernel", "Encoder/convolutional_encoder/conv2d_2/bias",
    "Encoder/convolutional_encoder/conv2d_3/kernel", "Encoder/convolutional_encoder/conv2d_3/bias",
    "Encoder/convolutional_encoder/conv2d_4/kernel", "Encoder/convolutional_encoder/conv2d_4/bias",
    "Encoder/convolutional_encoder/conv2d_strided/kernel", "Encoder/convolutional_encoder/conv2d_strided/bias",
    "Encoder/convolutional_encoder/conv2d_5/kernel", "Encoder/convolutional_encoder/conv2d_5/bias",
    "Encoder/row_encoder/bidirectional_rnn/fw/lstm_cell/kernel", "Encoder/row_encoder/bidirectional_rnn/fw/lstm_cell/bias",
    "Encoder/row_encoder/bidirectional_rnn/bw/lstm_cell/kernel", "Encoder/row_encoder/bidirectional_rnn/bw/lstm_cell/bias",
    "Decoder/embedding_table", "Decoder/start_token",
    "Decoder/AttentionCell/att_img/kernel",
    "Decoder/AttentionCell/att_mechanism/W_c_0", "Decoder/AttentionCell/att_mechanism/b_c_0",
    "Decoder/AttentionCell/att_mechanism/W_h_0", "Decoder/AttentionCell/att_mechanism/b_h_0",
    "Decoder/AttentionCell/att_mechanism/W_o_0", "Decoder/AttentionCell/att_mechanism/b_o_0",
    "Decoder/AttentionCell/rnn/lstm_cell/kernel", "Decoder/AttentionCell/rnn/lstm_cell/bias",
    "Decoder/AttentionCell/rnn/att_mechanism/dense/kernel", "Decoder/AttentionCell/rnn/att_mechanism/att_beta",
    "Decoder/AttentionCell/rnn/o_W_h", "Decoder/AttentionCell/rnn/o_W_c", "Decoder/AttentionCell/rnn/y_W_o",
};
const char* lxo_param_name(int id) { return (id >= 0 && id < P_COUNT) ? kParamNames[id] : ""; }
// TF numbers the tf.layers.conv2d scopes in creation order: with encoder_cnn == "cnn" the strided conv is
// conv2d_5 and the last 3x3 conv conv2d_6 (encoder.py:54-59)
const char* lxo_param_name_mode(int id, int encoder_cnn) {
    if (encoder_cnn) {
        switch (id) {
        case P_CONVS_W: return "Encoder/convolutional_encoder/conv2d_5/kernel";
        case P_CONVS_B: return "Encoder/convolutional_encoder/conv2d_5/bias";
        case P_CONV6_W: return "Encoder/convolutional_encoder/conv2d_6/kernel";
        case P_CONV6_B: return "Encoder/convolutional_encoder/conv2d_6/bias";
        default: break;
        }
    }
    return lxo_param_name(id);
}

static const char* kWsNames[W_COUNT] = {
    "p1", "y2", "p2", "y3", "y4", "p4", "y5", "p5", "y6", "img", "pos",
    "att_img", "att_part", "mean", "emb_in", "zx", "rec", "cs", "gates", "att_h", "alpha", "logits",
    "dlogits", "loss", "do_log", "g", "dhc", "de", "datth", "dz", "dxh", "dcc", "d_img", "d_att_img",
    "d_emb", "dpre0", "dmean", "g0", "g1", "g2", "gnorm",
    "s_k1", "s_k2", "s_k4", "s_b1", "s_b3", "s_b4",
    "recb", "gb", "dzb", "carry_h", "dec_tx", "dec_txe",
    "dec_ids", "dec_flags", "dec_emb", "dec_zx", "dec_logits", "beam_lp", "beam_par", "beam_tmp",
    "cols",
    "rxt", "rzx", "rg", "rc", "rh", "rhb", "rdz", "rdzb", "rdh", "rdcc", "rzero",
    "m2", "m4", "m5",
    "att_exp", "xdec_sync", "det_part", "datth_b",
};
const char* lxo_ws_name(int id) { return (id >= 0 && id < W_COUNT) ? kWsNames[id] : ""; }

static inline int cd2(int n) { return (n + 1) / 2; }
static inline size_t al256(size_t n) { return (n + 255) / 256 * 256; }

Plan::Plan(const lxo_shape& sh, int /*unused*/) : s(sh) {
    bf = s.dtype == LXO_BF16;
    esz = bf ? 2 : 4;
    const int B = s.B, C = s.C, E = s.E, U = s.U, O = s.O, D = s.D, V = s.V, T = s.T > 0 ? s.T : 1;
    H1 = cd2(s.H); W1 = cd2(s.W);          // after conv1's 2x2 pool      (encoder.py:34)
    H2 = cd2(H1); W2 = cd2(W1);            // after conv2's 2x2 pool      (encoder.py:39)
    cnn = s.encoder_cnn != 0;
    rnn = s.encoder_rnn != 0;
    Be = (s.live_B > 0 && s.live_B < s.B && !rnn) ? s.live_B : s.B;
    Ur = C / 2;
    H4 = cnn ? H2 : cd2(H2);               // after conv4's (2,1) pool    (encoder.py:47; "cnn": no pool)
    W5 = cd2(W2);                          // after conv5's (1,2) pool    (encoder.py:52) / the stride-2 conv (:54-56)
    H6 = cnn ? cd2(H2) : H4;               // rows conv6 reads
    Hp = H6 - 2; Wp = W5 - 2;              // conv6 VALID                 (encoder.py:59)
    R = Hp > 0 && Wp > 0 ? Hp * Wp : 0;
    const int ci[6] = {1, 64, 128, 256, 256, C}, co[6] = {64, 128, 256, 256, C, C};
    const int wid[6] = {P_CONV1_W, P_CONV2_W, P_CONV3_W, P_CONV4_W, P_CONV5_W, P_CONV6_W};
    for (int i = 0; i < 6; ++i) { convCin[i] = ci[i]; convCout[i] = co[i]; convW[i] = wid[i]; convB[i] = wid[i] + 1; }
    Vp = (V + 31) / 32 * 32;
    Dp = (D + 63) / 64 * 64;              // embedding rows padded to the 64-element K-step of the LDS-DMA GEMM (zero columns)
    Rp = (R + 7) / 8 * 8;
    XH = O + U; HC = U + C; OFF_HT = O + U; OFF_CTX = O + 2 * U; REC = O + 2 * U + C;
    const int WPAD = 128;
    ldRT = XH + WPAD; ldAHT = U + WPAD; ldOWT = HC + WPAD; ldOW = O + WPAD; ldAH = E + WPAD; ldK = 4 * U + WPAD;
    RECB = REC + WPAD; GBP = O + WPAD; DZBP = 4 * U + WPAD;

    long long cnt[P_COUNT];
    for (int i = 0; i < 6; ++i) { cnt[convW[i]] = 9LL * ci[i] * co[i]; cnt[convB[i]] = co[i]; }
    cnt[P_CONVS_W] = cnn ? 8LL * C * C : 0; cnt[P_CONVS_B] = cnn ? C : 0;
    cnt[P_ROWF_K] = cnt[P_ROWB_K] = rnn ? (long long)(C + Ur) * 4 * Ur : 0; cnt[P_ROWF_B] = cnt[P_ROWB_B] = rnn ? 4LL * Ur : 0;
    cnt[P_EMB] = (long long)V * D; cnt[P_START] = D; cnt[P_ATT_IMG] = (long long)C * E;
    cnt[P_WC0] = (long long)C * U; cnt[P_BC0] = U; cnt[P_WH0] = (long long)C * U; cnt[P_BH0] = U;
    cnt[P_WO0] = (long long)C * O; cnt[P_BO0] = O;
    cnt[P_LSTM_K] = (long long)(D + O + U) * 4 * U; cnt[P_LSTM_B] = 4LL * U;
    cnt[P_ATT_H] = (long long)U * E; cnt[P_BETA] = E;
    cnt[P_OWH] = (long long)U * O; cnt[P_OWC] = (long long)C * O; cnt[P_YWO] = (long long)O * V;
    ptotal = 0;
    for (int i = 0; i < P_COUNT; ++i) { poff[i] = ptotal; pcount[i] = cnt[i]; ptotal += cnt[i]; }

    size_t kb[K_COUNT];
    for (int l = 1; l < 6; ++l) {
        kb[K_CONV2_F + l - 1] = (size_t)co[l] * 9 * ci[l] * esz;
        kb[K_CONV2_D + l - 1] = (size_t)ci[l] * 9 * co[l] * esz;
    }
    kb[K_ATT_IMG_T] = (size_t)E * C * esz; kb[K_ATT_IMG] = (size_t)C * E * esz;
    kb[K_INIT_T] = (size_t)(2 * U + O) * C * esz; kb[K_INIT] = (size_t)C * (2 * U + O) * esz;
    kb[K_LSTM_XT] = (size_t)4 * U * Dp * esz; kb[K_LSTM_RT] = (size_t)4 * U * ldRT * esz;
    kb[K_LSTM] = (size_t)(D + O + U) * ldK * esz;
    kb[K_ATT_H_T] = (size_t)E * ldAHT * esz; kb[K_ATT_H] = (size_t)U * ldAH * esz;
    kb[K_OW_T] = (size_t)O * ldOWT * esz; kb[K_OW] = (size_t)HC * ldOW * esz;
    kb[K_YWO_T] = (size_t)V * O * esz; kb[K_YWO] = (size_t)O * Vp * esz;
    kb[K_CONVS_F] = cnn ? (size_t)8 * C * C * esz : 0; kb[K_CONVS_D] = kb[K_CONVS_F];
    for (int d = 0; d < 2; ++d) {
        kb[K_ROWX_T + d] = rnn ? (size_t)4 * Ur * C * esz : 0; kb[K_ROWX + d] = kb[K_ROWX_T + d];
        kb[K_ROWH_T + d] = rnn ? (size_t)4 * Ur * Ur * esz : 0; kb[K_ROWH + d] = kb[K_ROWH_T + d];
    }
    ktotal = 0;
    for (int i = 0; i < K_COUNT; ++i) { koff[i] = ktotal; ktotal += al256(kb[i] + 64); }

    const size_t f4 = 4;
    size_t wb[W_COUNT];
    memset(wb, 0, sizeof(wb));
    const size_t BL = (size_t)B;
    wb[W_P1] = BL * H1 * W1 * 64 * esz;
    wb[W_Y2] = BL * H1 * W1 * 128 * esz;  wb[W_P2] = BL * H2 * W2 * 128 * esz;
    wb[W_Y3] = BL * H2 * W2 * 256 * esz;  wb[W_Y4] = BL * H2 * W2 * 256 * esz;
    wb[W_P4] = cnn ? 0 : BL * H4 * W2 * 256 * esz;
    wb[W_Y5] = BL * H4 * W2 * C * esz;    wb[W_P5] = BL * H6 * W5 * C * esz;
    wb[W_COLS] = cnn ? BL * H6 * W5 * 8 * C * esz : 0;
    wb[W_Y6] = BL * R * C * esz;          wb[W_IMG] = BL * R * C * esz;
    wb[W_POS] = (size_t)R * C * f4;
    if (rnn) {
        const size_t TM = (size_t)B * (Hp > 0 ? Hp : 0) * (Wp > 0 ? Wp : 0), Mr = (size_t)B * (Hp > 0 ? Hp : 0);
        wb[W_RXT] = TM * C * esz;
        wb[W_RZX] = 2 * TM * 4 * Ur * f4;             // also holds d_X (TM x C f32) in the backward pass
        wb[W_RG] = 2 * TM * 4 * Ur * f4; wb[W_RC] = 2 * TM * Ur * f4; wb[W_RH] = 2 * TM * Ur * f4;
        wb[W_RHB] = bf ? 2 * TM * Ur * 2 : 0;
        wb[W_RDZ] = 2 * TM * 4 * Ur * f4; wb[W_RDZB] = bf ? 2 * TM * 4 * Ur * 2 : 0;
        wb[W_RDH] = TM * C * f4; wb[W_RDCC] = 2 * Mr * Ur * f4; wb[W_RZERO] = Mr * 4 * Ur * f4;
    }
    wb[W_M2] = bf ? BL * H2 * W2 * 128 : 0;
    wb[W_M4] = (bf && !cnn) ? BL * H4 * W2 * 256 : 0;
    wb[W_M5] = (bf && !cnn) ? BL * H6 * W5 * C : 0;
    wb[W_ATT_EXP] = bf ? BL * R * E * esz : 0;
    wb[W_XSYNC] = bf ? 2 * ((size_t)4096 + (384u << 10)) : 8192;     // one block per chain, forward then backward (xdec.h: kXDecBlockBytes)
    wb[W_DET] = 8192;                    // every mode: the 512 x 2 per-workgroup partial loss statistics (lxo_impl_ce_loss: ordered AND faster than 2048 workgroups' atomics on two words)
    if (!bf || s.deterministic) {        // the largest user (f32): d_beta, one E-vector per (sample, 16 regions) workgroup; column sums use at most 1024 row blocks x 4U columns
        size_t need = BL * (size_t)((R + 15) / 16) * E * f4;
        const size_t floor_ = (size_t)1024 * (4 * U > 2048 ? 4 * U : 2048) * f4;
        wb[W_DET] = need > floor_ ? need : floor_;
        // bf16 deterministic mode: conv_wgrad_kernel stores one (64 ci x 128 co x 9 taps) f32 partial per (pixel range, tile) workgroup
        // -- at most 264 of them (256 rounded up to whole XCD rounds) -- and an ordered pass adds them (conv_wgrad.hip)
        const size_t wg = (size_t)264 * 9 * 64 * 128 * f4;
        if (bf && wb[W_DET] < wg) wb[W_DET] = wg;
        if (bf) wb[W_DET] = 2 * ((wb[W_DET] + 255) / 256 * 256);      // one half per stream (Plan::det_scratch / det_scratch_side)
    }
    const int nb = s.beam > 1 ? s.beam : 1;
    const size_t BK_ = BL * nb;                      // decoder rows (beam-expanded for decode)
    const size_t TB = (size_t)T * B;
    wb[W_DATTHB] = bf ? TB * E * 2 : 0;
    wb[W_ATT_IMG] = BL * R * E * esz;
    wb[W_APART] = BK_ * 32 * (C + 16) * f4;        // chunk partials of the attention forward: at most 32 chunks per row, [max, sum, context] (the persistent chain pads a partial to C + 4 floats, or -- polled hand-over form, xdec.hip: PLW -- to C / 2 + 8 words of 8 bytes)
    wb[W_MEAN] = BL * C * f4;
    wb[W_EMB_IN] = TB * Dp * esz;
    wb[W_ZX] = TB * 4 * U * f4;
    wb[W_REC] = (size_t)(T + 1) * B * REC * f4;
    wb[W_CS] = (size_t)(T + 1) * B * U * f4;
    wb[W_GATES] = TB * 4 * U * f4;
    wb[W_ATTH] = TB * E * f4;
    wb[W_ALPHA] = TB * Rp * f4;
    wb[W_LOGITS] = TB * Vp * f4;
    wb[W_DLOGITS] = TB * Vp * esz;
    wb[W_LOSS] = 64;
    wb[W_DOLOG] = TB * O * f4;
    wb[W_G] = TB * O * f4;
    wb[W_DHC] = TB * HC * f4;
    wb[W_DE] = TB * Rp * f4;
    wb[W_DATTH] = TB * E * f4;
    wb[W_DZ] = TB * 4 * U * f4;
    wb[W_DXH] = BL * XH * f4;
    wb[W_DCC] = BL * U * f4;
    wb[W_DIMG] = BL * R * C * f4;
    wb[W_DATTIMG] = BL * R * E * esz;
    wb[W_DEMB] = TB * D * f4;
    wb[W_DPRE0] = BL * (2 * U + O) * f4;
    wb[W_DMEAN] = BL * C * f4;
    // encoder-backward ping-pong scratch: the largest activation gradient is d_y2
    size_t gmax = wb[W_Y2];
    const size_t cand[] = {wb[W_P1], wb[W_P2], wb[W_Y3], wb[W_Y4], wb[W_P4], wb[W_Y5], wb[W_P5], wb[W_Y6]};
    for (size_t c : cand) if (c > gmax) gmax = c;
    wb[W_G0] = gmax; wb[W_G1] = gmax; wb[W_G2] = gmax;
    wb[W_GNORM] = 64;
    wb[W_S_K1] = (size_t)(XH / 128) * BK_ * 4 * U * f4;
    wb[W_S_K2] = (size_t)(U / 128) * BK_ * E * f4;
    wb[W_S_K4] = (size_t)(HC / 128) * BK_ * O * f4;
    wb[W_S_B1] = (size_t)(O / 128) * BL * HC * f4;
    wb[W_S_B3] = (size_t)(E / 128) * BL * U * f4;
    wb[W_S_B4] = (size_t)(4 * U / 128) * BL * XH * f4;
    wb[W_RECB] = bf ? (size_t)(T + 1) * B * RECB * 2 : 0;
    wb[W_GB] = bf ? TB * GBP * 2 : 0;                // every step's g_t / d_z_t: the deferred weight-gradient GEMMs read the mirrors
    wb[W_DZB] = bf ? TB * DZBP * 2 : 0;
    wb[W_CARRYH] = BL * U * f4;
    const int ms = s.max_steps > 0 ? s.max_steps : 0;
    if (ms > 0) {
        wb[W_DEC_IDS] = BK_ * ms * 4 + 64;                // + the greedy chain's stop word, one int behind the B fed-back ids (model_decoder.hip: x.stop), whatever max_steps
        wb[W_DEC_FLAGS] = 256 + BK_ * 4;
        wb[W_DEC_EMB] = BK_ * Dp * esz;
        wb[W_DEC_TX] = (size_t)(V + 1) * 4 * U * f4;       // row v = embedding_table[v] K[0:D] + b, row V = start_token K[0:D] + b
        wb[W_DEC_TXE] = (size_t)(V + 1) * Dp * esz;
        wb[W_DEC_ZX] = BK_ * 4 * U * f4;
        wb[W_DEC_LOGITS] = BK_ * Vp * f4;
        wb[W_BEAM_LP] = BK_ * 2 * f4;
        wb[W_BEAM_PAR] = BK_ * ms * 4;
        wb[W_BEAM_TMP] = 3 * (BK_ * REC + BK_ * U) * f4 / 3 + BK_ * 64;
        if (wb[W_BEAM_TMP] < BK_ * Vp * f4) wb[W_BEAM_TMP] = BK_ * Vp * f4;   // also the penalised-score scratch of add_div_penalty
        // decode reuses rec / cs / att_h / alpha with (T = 2 ping-pong) rows per beam
        const size_t recd = 2 * BK_ * REC * f4, csd = 2 * BK_ * U * f4;
        if (wb[W_REC] < recd) wb[W_REC] = recd;
        if (bf && wb[W_RECB] < 2 * BK_ * RECB * 2) wb[W_RECB] = 2 * BK_ * RECB * 2;
        if (wb[W_CS] < csd) wb[W_CS] = csd;
        if (wb[W_GATES] < BK_ * 4 * U * f4) wb[W_GATES] = BK_ * 4 * U * f4;
        if (wb[W_ATTH] < BK_ * E * f4) wb[W_ATTH] = BK_ * E * f4;
        if (wb[W_ALPHA] < BK_ * Rp * f4) wb[W_ALPHA] = BK_ * Rp * f4;
        if (wb[W_ZX] < BK_ * 4 * U * f4) wb[W_ZX] = BK_ * 4 * U * f4;
    }
    wtotal = 0;
    for (int i = 0; i < W_COUNT; ++i) { wbytes[i] = wb[i]; woff[i] = wtotal; wtotal += al256(wb[i] + 256); }
}

int Plan::attn_chunks(int nv) const {
    static int target = -1;                        // tuning knob: workgroups wanted for the attention stream
    if (target < 0) { const char* e = getenv("LXO_ATT_WGS"); target = (e && atoi(e) > 0) ? atoi(e) : 512; }
    int nch = (target + nv - 1) / nv;              // ~2 workgroups per CU by default
    if (nch > 16) nch = 16;
    const int by_rows = R / 32 > 0 ? R / 32 : 1;   // keep >= 32 rows per chunk
    if (nch > by_rows) nch = by_rows;
    const int need = (R + 1023) / 1024;            // <= 1024 rows per chunk (LDS)
    if (nch < need) nch = need;
    return nch < 1 ? 1 : nch;
}

int Plan::validate(char* msg, size_t n) const {
#define BAD(cond, text) if (cond) { snprintf(msg, n, "lxo_shape invalid: %s", text); return -10; }
    BAD(s.B <= 0 || s.H <= 0 || s.W <= 0, "B/H/W must be positive");
    BAD(Hp <= 0 || Wp <= 0, "image too small: need ceil(H/8) >= 3 and ceil(W/8) >= 3");
    BAD(s.V < 4, "V < 4");
    BAD(s.C != 512 && s.C != 256 && s.C != 128, "C must be 128/256/512");
    BAD(s.E % 128 || s.U % 128 || s.O % 128 || s.E <= 0 || s.U <= 0 || s.O <= 0, "E, U, O must be positive multiples of 128");
    BAD(s.D % 8 || s.D <= 0, "D must be a positive multiple of 8");
    BAD(s.dtype != LXO_F32 && s.dtype != LXO_BF16, "dtype");
    BAD(s.E > 1024 || s.C > 512, "E <= 1024, C <= 512");
    BAD(R > 16384, "more than 16384 regions");
    BAD(s.encoder_rnn && s.C != 512 && s.C != 256, "encoder_rnn needs C in {256, 512} (C/2 units per direction, multiples of 128)");
#undef BAD
    return 0;
}

bool Plan::att_exp() const {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LXO_ATT_EXP"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return bf && v == 1 && s.E <= 256 && (long long)s.B * R * s.E % 8 == 0;
}
// tf.nn.dropout(., config.dropout) masks of one decoder step (attention_cell.py:72,83)
bool Plan::pool_fused() const {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LXO_POOL_FUSED"); v = (e && atoi(e) == 0) ? 0 : 1; }
    static int halo = -1;                                 // the A/B switches that select the older conv kernels (no fused pool there)
    if (halo < 0) { const char* b2 = getenv("LXO_CONV_2WG"); halo = (b2 && b2[0] == '0') ? 0 : 1; }
    return bf && v == 1 && halo == 1 && s.C % 128 == 0;
}
Drop Plan::drop(int t, int row0) const {
    Drop d = {0u, 1.f, (unsigned)s.dropout_seed, t, row0, s.B};
    if (s.keep_prob > 0.f && s.keep_prob < 1.f) {
        d.thr = (unsigned)(s.keep_prob * 16777216.0f);
        if (d.thr == 0u) d.thr = 1u;
        d.inv_keep = 1.f / s.keep_prob;
    }
    return d;
}
