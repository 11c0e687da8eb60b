// Persistent, XCD-local decoder chain (xdec.hip): the T teacher-forced steps of AttentionCell.step (attention_cell.py:58-89) in ONE
// launch -- 8 independent chains (one per XCD, B/8 samples each), workgroups of a chain hand over through their XCD's L2 and
// meet at a flag-line barrier that costs 0.44 us (tools/xcd_barrier_probe.hip) where a dependent kernel boundary costs 1.5 us.
#pragma once
#include "lxo_common.h"
#include "decoder_kernels.h"
#include "lxo_debug.h"

struct XDecFwd {
    // recurrent weights, bf16, K-contiguous ([out][in]) with the row pitches of the plan; each workgroup keeps its column slice of all
    // three in REGISTERS for the whole launch
    const bf16_t* Wrt; int ldrt;      // K_LSTM_RT [4U][XH]: z = [o_prev | h_prev] Wrt^T
    const bf16_t* Wah; int ldah;      // K_ATT_H_T [E][U]
    const bf16_t* Wow; int ldow;      // K_OW_T    [O][U + C]
    const float* beta;                // [E]
    const bf16_t* att_img;            // [B][R][E]
    const bf16_t* att_exp;            // [B][R][E] e^{2 att_img} (nullable): when given, the scores are formed in the E domain (one reciprocal per element)
    const bf16_t* img;                // [B][R][C]
    const float* zx;                  // [T][B][4U]: emb_t K[0:D] + b
    float* rec; bf16_t* recb;         // [(T + 1)][B][REC] f32 record [o | h | h~ | ctx] and its bf16 mirror (pitch RECB); slot 0 = initial state
    float* cs;                        // [(T + 1)][B][U]
    float* gates;                     // [T][B][4U] activated gates i, j, f, o
    float* atth;                      // [T][B][E]
    float* alpha;                     // [T][B][Rp]
    float* part;                      // [B][nq][C + 2] chunk partials (max, sum, unnormalised context)
    unsigned* sync;                   // this chain's block of ws region "xdec_sync" (kXDecBlockBytes, see below; zeroed by the launcher)
    int T, B, R, Rp, REC, RECB;
    Drop dr;                          // dropout of h and o (thr == 0: off); dr.t is set per step
    unsigned long long* dbg;          // measurement aid (null = off): [256 workgroups][T][16] 100 MHz timestamps at the phase boundaries (tools/xdec_stamps.py)
};

// 0 = launched; -2 = the shape does not qualify (the caller runs the launch-per-step chain instead)
int lxo_launch_xdec_fwd(const XDecFwd& p, int U, int O, int C, int E, hipStream_t st);

// The same chain for GREEDY DECODE (dynamic_decode.py:17-74 over greedy_decoder_cell.py:53-66): steps t0 .. t0 + nsteps - 1 of the decode in
// one launch.  What differs from the teacher-forced chain: the x-part of the LSTM pre-activation of a row is the row of the per-token table
// its previous arg-max picks (row V: the start token), and every step ends with logits = o y_W_o, ids = arg-max -- computed at the START of
// the next step's LSTM phase from the very o fragments that phase polls (the chain's last step is followed by one more such boundary).
// The arg-max crosses the 32 workgroups of a chain as hand-over words {max, step tag << 16 | index}.  State lives in two alternating record
// slots (slot t & 1 holds the state step t starts from), as in the launch-per-step decode.
struct XDecDec {
    const bf16_t* Wrt; int ldrt; const bf16_t* Wah; int ldah; const bf16_t* Wow; int ldow;      // as in XDecFwd
    const bf16_t* Wyo; int ldyo;      // K_YWO_T [V][O]
    const float* beta;
    const bf16_t* att_exp;            // [B][R][E] e^{2 att_img}
    const bf16_t* img;                // [B][R][C]
    const float* tx;                  // [V + 1][4U] f32: emb(id) K[0:D] + b per input token; row V = the start token
    float* rec; bf16_t* recb;         // [2][B][REC] / [2][B][RECB]
    float* cs;                        // [2][B][U]
    float* part;                      // [B][nq][C + 4] chunk partials
    unsigned* sync;                   // block 0 of ws region "xdec_sync" (zeroed by the launcher); the arg-max words live in block 1's hand-over area
    int* ids_step;                    // [B] in: the ids fed at step t0 (t0 > 0); out: the ids of the launch's last step
    int* ids_out;                     // [B][max_steps]
    int* finished;                    // [B] (0 / 1, sticky)
    int* unfinished;                  // [nsteps <= 16] zeroed by the caller, one word per step of this launch: low 16 bits = rows still unfinished after
                                      // the step (summed over the chains), high bits = chains that have reported it (8 = all)
    int* stop;                        // one word, zero at the start of the decode: set once every row of every chain has finished -- the chains of
                                      // this launch stop within three steps of that point, later (speculative) launches return at once
    int B, R, REC, RECB, V, id_end, t0, nsteps, max_steps;
};
int lxo_launch_xdec_dec(const XDecDec& p, int U, int O, int C, int E, hipStream_t st);

// The same chain for BPTT (attention_cell.py:58-89 backwards, steps T-1 .. 0): per step [d_h~ | d_ctx] = g o_W^T, the attention stream
// (d_e, d_att_h), d_h -> LSTM cell backward (d_z, d_c), the carries [d_o | d_h] = d_z K[D:]^T and g_{t-1}.  Everything a step hands to
// the deferred all-step weight-gradient GEMMs (g, d_z and their bf16 mirrors, [d_h~ | d_ctx], d_e, d_att_h) is left in the same arrays
// the launch-per-step chain fills.
struct XDecBwd {
    const bf16_t* Wow; int ldow;      // K_OW    [U + C][O]   ([out][in] of the backward product)
    const bf16_t* Wah; int ldah;      // K_ATT_H [U][E]
    const bf16_t* Wk; int ldk;        // K_LSTM rows D.. : [O + U][4U]
    const float* beta;                // [E]
    const bf16_t* att_img; const bf16_t* att_exp; const bf16_t* img;     // as in XDecFwd
    const float* rec; int REC;        // forward record [(T + 1)][B][REC]
    const float* cs; const float* gates; const float* atth; const float* alpha; int Rp;
    const float* dolog;               // [T][B][O] d_o from the logits
    float* gall; bf16_t* gb; int GBP; // g_t [T][B][O] + mirror; slot T-1 holds g_{T-1} on entry
    float* dhc;                       // [T][B][U + C]
    float* de;                        // [T][B][Rp]
    float* datth;                     // [T][B][E]
    bf16_t* datthb;                   // [T][B][E] bf16 mirror of it (nullable)
    float* dz; bf16_t* dzb; int DZBP; // [T][B][4U] + mirror
    float* carry_h;                   // [B][U] scratch
    float* dcc;                       // [B][U] out: d_c of the initial state
    float* dxh;                       // [B][O + U] out: the raw carries of step 0
    float* part;                      // [B][nq][E] d_att_h chunk partials
    unsigned* sync;                   // as in XDecFwd (its own block)
    int T, B, R;
    Drop dr;
    unsigned long long* dbg;          // as in XDecFwd
};
int lxo_launch_xdec_bwd(const XDecBwd& p, int U, int O, int C, int E, hipStream_t st);
// measurement aid (include/lxo_debug.h: lxo_xdec_debug / lxo_xdec_debug_bwd): the next launches of this host thread stamp their phases into a buffer
// One block per chain in ws region "xdec_sync": [sync words, 4096 B: per-XCD flag line + ticket line, error word at [512]]
// [hand-over words, 384 KB: 8-byte {value, tag} pairs for up to 64 samples]; the launcher zeroes the whole block (tags of an earlier
// launch must not pass for this one's).  Forward chain: block 0, backward chain: block 1.
constexpr size_t kXDecSyncBytes = 4096, kXDecLLBytes = 384u << 10, kXDecBlockBytes = kXDecSyncBytes + kXDecLLBytes;
constexpr size_t kLLFwdHt = 0, kLLFwdAh = 128u << 10, kLLFwdO = 256u << 10;      // h~ [B][256 pairs], att_h [B][256], o [B][256 pairs]
constexpr size_t kLLBwdGb = 0, kLLBwdDctx = 128u << 10;                          // g [B][256 pairs], d_ctx [B][512]
