// Shape-derived plan of the hot path: encoder geometry, offsets of every variable
// in the flat f32 parameter/gradient buffers (TF variable order of SURVEY.md
// Appendix B), offsets of the packed compute-dtype weight copies, and the carve-up
// of the caller-provided workspace.  Pure host arithmetic; no allocation.
#pragma once
#include "lxo.h"
#include "lxo_common.h"
#include "decoder_kernels.h"

enum ParamId {
    P_CONV1_W, P_CONV1_B, P_CONV2_W, P_CONV2_B, P_CONV3_W, P_CONV3_B, P_CONV4_W, P_CONV4_B,
    P_CONV5_W, P_CONV5_B,
    P_CONVS_W, P_CONVS_B,          // encoder_cnn == "cnn": the (2,4) stride-2 conv of encoder.py:54-56 (count 0 otherwise)
    P_CONV6_W, P_CONV6_B,
    P_ROWF_K, P_ROWF_B, P_ROWB_K, P_ROWB_B,   // optional row-BiLSTM encoder (lxo_shape.encoder_rnn; count 0 otherwise): TF LSTMCell kernel [(C + C/2)][4 C/2] + bias, forward / backward direction
    P_EMB, P_START, P_ATT_IMG, P_WC0, P_BC0, P_WH0, P_BH0, P_WO0, P_BO0,
    P_LSTM_K, P_LSTM_B, P_ATT_H, P_BETA, P_OWH, P_OWC, P_YWO, P_COUNT
};

enum PackId {
    K_CONV2_F, K_CONV3_F, K_CONV4_F, K_CONV5_F, K_CONV6_F,      // [Cout][9*Cin]
    K_CONV2_D, K_CONV3_D, K_CONV4_D, K_CONV5_D, K_CONV6_D,      // [Cin][9*Cout] flipped
    K_ATT_IMG_T,   // [E][C]
    K_ATT_IMG,     // [C][E]
    K_INIT_T,      // [3][U][C]   (c, h, o) ; o block is [O][C]
    K_INIT,        // [3][C][U]
    K_LSTM_XT,     // [4U][Dp]
    K_LSTM_RT,     // [4U][O+U]
    K_LSTM,        // [D+O+U][4U]
    K_ATT_H_T,     // [E][U]
    K_ATT_H,       // [U][E]
    K_OW_T,        // [O][U+C]
    K_OW,          // [U+C][O]
    K_YWO_T,       // [V][O]
    K_YWO,         // [O][Vp]
    K_CONVS_F,     // [C][8C]   strided conv, tap-major rows (cnn encoder only)
    K_CONVS_D,     // [8C][C]
    K_ROWX_T, K_ROWX_T1,   // row encoder (fw, bw): x-part of the LSTM kernel as [4Ur][C]      (ZX = X Kx)
    K_ROWH_T, K_ROWH_T1,   //   h-part as [4Ur][Ur]                                            (step GEMM h_prev Kh)
    K_ROWH, K_ROWH1,       //   h-part as stored [Ur][4Ur]                                     (carry GEMM d_z Kh^T)
    K_ROWX, K_ROWX1,       //   x-part as stored [C][4Ur]                                      (d_X = d_Z Kx^T)
    K_COUNT
};

enum WsId {
    W_P1, W_Y2, W_P2, W_Y3, W_Y4, W_P4, W_Y5, W_P5, W_Y6, W_IMG, W_POS,
    W_ATT_IMG, W_APART, W_MEAN, W_EMB_IN, W_ZX, W_REC, W_CS, W_GATES, W_ATTH, W_ALPHA, W_LOGITS,
    W_DLOGITS, W_LOSS, W_DOLOG, W_G, W_DHC, W_DE, W_DATTH, W_DZ, W_DXH, W_DCC, W_DIMG, W_DATTIMG,
    W_DEMB, W_DPRE0, W_DMEAN, W_G0, W_G1, W_G2, W_GNORM,
    W_S_K1, W_S_K2, W_S_K4, W_S_B1, W_S_B3, W_S_B4,   // split-K slabs of the recurrent GEMMs (step_kernels == 1 only)
    W_RECB, W_GB, W_DZB, W_CARRYH, W_DEC_TX, W_DEC_TXE,                    // fused step kernels: bf16 mirrors of rec / g_t / d_z_t, carried d_h; decode: x-part of the LSTM pre-activation per token (+ its input rows)
    // decode-only
    W_DEC_IDS, W_DEC_FLAGS, W_DEC_EMB, W_DEC_ZX, W_DEC_LOGITS, W_BEAM_LP, W_BEAM_PAR, W_BEAM_TMP,
    W_COLS,        // cnn encoder only: im2col of the strided conv [B*H6*W5][8C], reused for its column gradient
    // row encoder only (time-major = [W'][B*H'][.]): input copy, x-part pre-activations (reused for d_X), gates, cell states, outputs (+ bf16 mirror),
    // d_z (+ mirror), upstream gradient, carried d_c, a zero block
    W_RXT, W_RZX, W_RG, W_RC, W_RH, W_RHB, W_RDZ, W_RDZB, W_RDH, W_RDCC, W_RZERO,
    W_M2, W_M4, W_M5,   // bf16 mode: pool masks of conv2 / conv4 / conv5 (one byte per pooled element; fused conv + pool epilogue)
    W_ATT_EXP,          // bf16 mode: e^{2 att_img} (bf16), what the E-domain attention kernels of the training step read (att_exp)
    W_XSYNC,            // persistent decoder chain (xdec.hip): per-XCD flag lines, tickets, error word
    W_DET,              // f32 parity mode: slots of per-workgroup partial sums for the ordered (atomic-free) reductions (DetScratch)
    W_DATTHB,           // bf16 mirror of d_att_h [T][B][E], left by the backward chain: operand of the deferred dW_att_h product
    W_COUNT
};

struct Plan {
    lxo_shape s;
    bool bf;                 // compute dtype is bf16
    size_t esz;              // bytes per compute-dtype element
    // encoder geometry
    int H1, W1, H2, W2, H4, W5, H6, Hp, Wp, R;   // conv5 runs at H4 x W2, conv6 reads H6 x W5 (vanilla: H6 == H4)
    bool cnn;                                    // encoder_cnn == "cnn": no pools after conv4/conv5, (2,4)/2 conv instead
    bool rnn;                                    // encoder_rnn: row-BiLSTM between conv6 and the decoder
    int Be;                                      // images the ENCODER computes: lxo_shape.live_B (the rows behind them are dead padding rows of a filled-up batch), else B
    int Ur;                                      //   units per direction (C / 2)
    int convCin[6], convCout[6], convW[6], convB[6];   // per 3x3 layer: channels and ParamIds
    int Vp, Dp, Rp, XH, HC, REC;   // padded V / D / R (row pitches), O+U, U+C, record width O+2U+C
    int OFF_HT, OFF_CTX;           // record = [o | h | h~ | ctx]: [o|h] feeds the LSTM, [h~|ctx] the attention and o projection
    // Row pitches of the operands the step GEMMs read with many rows in flight.  A pitch that is a multiple of 4 KB puts
    // the same 128-byte piece of EVERY row on one L2 channel (256-byte interleave over 16 channels): in-kernel stamps
    // showed the fetch phase of a step GEMM at half the rate of a contiguous read.  +128 elements walks the channels.
    int ldRT, ldAHT, ldOWT, ldOW, ldAH, ldK;     // packed recurrent weights K_LSTM_RT, K_ATT_H_T, K_OW_T, K_OW, K_ATT_H, K_LSTM
    int RECB, GBP, DZBP;                         // bf16 mirrors: record, g_t, d_z_t
    Drop drop(int t, int row0) const;   // dropout descriptor of decoder step t for rows row0.. (off when keep_prob is 0 or >= 1)
    // flat parameter buffer
    long long poff[P_COUNT], pcount[P_COUNT], ptotal;
    // packed weights (bytes)
    size_t koff[K_COUNT], ktotal;
    // workspace (bytes)
    size_t woff[W_COUNT], wbytes[W_COUNT], wtotal;

    explicit Plan(const lxo_shape& sh, int beam = 1);
    int validate(char* msg, size_t n) const;
    // chunks the R regions are split into for the attention stream, for nv decoder rows
    int attn_chunks(int nv) const;

    // bf16 mode: the decoder's d_img GEMM (dimg.hip) applies conv6's ReLU mask and bias-gradient sum in its epilogue and
    // leaves d_y6 (compute dtype) in ws region "d_img"; otherwise the region holds the f32 gradient w.r.t. the encoder output
    // bf16 mode: conv2 / conv4 / conv5 pool in their own epilogue and leave a routing mask; the full-resolution activations
    // y2 / y4 / y5 are then never written (LXO_POOL_FUSED=0 restores the separate pool kernels for A/B runs)
    bool pool_fused() const;
    bool dimg_masked() const { return bf && s.E % 32 == 0 && !rnn; }
    // f32 mode is the PARITY mode: every reduction runs in a fixed order (no float atomics), so a step is reproducible bit for bit
    // (SURVEY.md Appendix D step 8); bf16 mode keeps the atomic epilogues
    // lxo_shape.deterministic extends that to the bf16 mode (opt-in: ordered partial slots instead of the atomic epilogues)
    bool det() const { return !bf || s.deterministic != 0; }
    // bf16 training: the attention kernels of the recurrence read E_x = e^{2 att_img} (region att_exp) and form tanh from one reciprocal
    // per element; LXO_ATT_EXP=0 keeps the x form (A/B)
    bool att_exp() const;
    // bf16 deterministic mode: the region is two halves -- the main stream's ordered slots / slabs and those of the weight-gradient side stream
    // (model_encoder.hip, model_decoder.hip: two streams must not share one scratch)
    DetScratch det_scratch(void* base) const { DetScratch d = {nullptr, 0}; if (det()) { d.p = ws<float>(base, W_DET); d.floats = wbytes[W_DET] / 4 / (bf ? 2 : 1); } return d; }
    DetScratch det_scratch_side(void* base) const { DetScratch d = {nullptr, 0}; if (det() && bf) { d.floats = wbytes[W_DET] / 8; d.p = ws<float>(base, W_DET) + d.floats; } return d; }   // with the row encoder "d_img" is the gradient w.r.t. ITS output: plain f32
    template <class T> T* ws(void* base, WsId id) const { return reinterpret_cast<T*>(static_cast<char*>(base) + woff[id]); }
    const void* pk(const void* base, PackId id) const { return static_cast<const char*>(base) + koff[id]; }
    void* pk(void* base, PackId id) const { return static_cast<char*>(base) + koff[id]; }
};

const char* lxo_param_name(int id);
const char* lxo_param_name_mode(int id, int encoder_cnn);
