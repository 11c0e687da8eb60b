// Optional row encoder between the CNN and the decoder (lxo_shape.encoder_rnn): every row of the H' x W' feature map runs
// through a bidirectional LSTM along W' (north_star's "row-BiLSTM encoder"; the arrangement of Deng et al., "Image-to-Markup
// Generation with Coarse-to-Fine Attention", which the reference's decoder.py:16 cites).  It is ABSENT from the reference --
// encoder.py:4 imports GRUCell / LSTMCell and never uses them -- so it is off by default and outside the parity contract; the
// cell arithmetic is the TF-1.12 LSTMCell the decoder uses (gate order i, j, f, o; forget_bias 1.0; zero initial state), C/2
// units per direction, outputs concatenated [forward | backward] in place of the features the attention reads.
//
// Everything is time-major here ([W'][B*H'][.]): one transposing copy in, one out.  The x-part of all pre-activations is ONE
// GEMM per direction over B*H'*W' rows; a time step is ONE fused step kernel per direction (csrc/rstep.hip: RS_LSTM_FWD = step
// GEMM h_prev Kh + gates + cell in the epilogue; RS_LSTM_BWD = carry GEMM d_z_next Kh^T + the cell's backward in the epilogue);
// weight gradients are deferred GEMMs over all steps (gemm_tn), d_X = d_Z Kx^T one GEMM per direction.
#include "impl.h"
#include "rstep.h"
#include "gemm.h"
#include "api_util.h"

namespace {

// rows of 16-byte pieces: TO_TM: dst row (w*M + m) <- src row (m*T + w); else the reverse
template <bool TO_TM>
__global__ __launch_bounds__(256) void rows_permute_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int M, int T, int pieces) {
    const long long total = (long long)M * T * pieces;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / pieces;                       // destination row
        const int pc = (int)(i - row * pieces);
        long long srow;
        if (TO_TM) { const int w = (int)(row / M), m = (int)(row - (long long)w * M); srow = (long long)m * T + w; }
        else { const int m = (int)(row / T), w = (int)(row - (long long)m * T); srow = (long long)w * M + m; }
        dst[row * pieces + pc] = src[srow * pieces + pc];
    }
}

// img[(m*T + w)][d*U + u] = h_d[(w*M + m)][u]
template <typename CT>
__global__ __launch_bounds__(256) void rowenc_assemble_kernel(const float* __restrict__ hf, const float* __restrict__ hb, CT* __restrict__ img, int M, int T, int U) {
    const long long total = (long long)M * T * (2 * U / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / (2 * U / 4);                  // (m, w) order
        const int c = (int)(i - row * (2 * U / 4)) * 4;
        const int m = (int)(row / T), w = (int)(row - (long long)m * T);
        const float* src = (c < U ? hf + ((long long)w * M + m) * U + c : hb + ((long long)w * M + m) * U + (c - U));
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        CT* dst = img + row * 2 * U + c;
        if constexpr (sizeof(CT) == 2) { u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}; *reinterpret_cast<u32x2*>(dst) = pk; }
        else *reinterpret_cast<f32x4*>(dst) = v;
    }
}

int permute(bool to_tm, const void* src, void* dst, int M, int T, size_t row_bytes, hipStream_t st) {
    if (row_bytes % 16) return -2;
    const int pieces = (int)(row_bytes / 16);
    long long blocks = ((long long)M * T * pieces + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (to_tm) hipLaunchKernelGGL((rows_permute_kernel<true>), dim3((int)blocks), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, M, T, pieces);
    else hipLaunchKernelGGL((rows_permute_kernel<false>), dim3((int)blocks), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, M, T, pieces);
    return (int)hipGetLastError();
}

int gemm_nt_plain(const Plan& P, bool a_f32, const void* A, int lda, const void* Bp, int ldb, float* C, int ldc, int M, int N, int K,
                  const float* bias, bool accumulate, hipStream_t st) {
    GemmNT g; memset(&g, 0, sizeof(g));
    g.A = A; g.Bp = Bp; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.bias = bias; g.alpha = 1.f; g.accumulate = accumulate ? 1 : 0; g.addend_rows = 1;
    const bool f32 = P.s.dtype == LXO_F32;
    return lxo_launch_gemm_nt(P.s.dtype, f32 || a_f32, 1, 0, g, st);
}
int gemm_tn_acc(const Plan& P, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int I, int J, hipStream_t st) {
    GemmTN g; memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.M = M; g.I = I; g.J = J; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    const int tiles = cdiv(I, 128) * cdiv(J, 128);
    int ns = cdiv(512, tiles);
    const int maxs = M / 64 > 0 ? M / 64 : 1;
    if (ns > maxs) ns = maxs;
    g.nsplit = ns < 1 ? 1 : ns; g.nbatch = 1; g.atomic = 1;
    if (P.bf && P.det()) g.nsplit = 1;                    // bf16 deterministic mode: one row range per output tile (these products are small)
    const bool f32 = P.s.dtype == LXO_F32;
    return lxo_launch_gemm_tn(P.s.dtype, f32, f32, g, st);     // bf16 mode: both operands are bf16 (X^T / h mirrors, d_z mirrors)
}

struct RowGeo { int M, T, U, C; size_t TM; };
RowGeo geo(const Plan& P) { RowGeo g; g.M = P.s.B * P.Hp; g.T = P.Wp; g.U = P.Ur; g.C = P.s.C; g.TM = (size_t)g.M * g.T; return g; }

}  // namespace

int lxo_impl_rowenc_fwd(const Plan& P, const float* prm, const void* wp, void* ws, hipStream_t st) {
    const RowGeo G = geo(P);
    const int M = G.M, T = G.T, U = G.U, C = G.C;
    const bool bf = P.bf;
    char* img = P.ws<char>(ws, W_IMG);
    char* xt = P.ws<char>(ws, W_RXT);
    float* zx = P.ws<float>(ws, W_RZX); float* rg = P.ws<float>(ws, W_RG); float* rc = P.ws<float>(ws, W_RC);
    float* rh = P.ws<float>(ws, W_RH); bf16_t* rhb = P.ws<bf16_t>(ws, W_RHB);
    float* zero = P.ws<float>(ws, W_RZERO);
    RC(permute(true, img, xt, M, T, (size_t)C * P.esz, st));
    HIPRC(hipMemsetAsync(zero, 0, (size_t)M * 4 * U * 4, st));
    for (int d = 0; d < 2; ++d)          // x-part of every pre-activation, with the LSTM bias: [T*M][C] x [C][4U]
        RC(gemm_nt_plain(P, false, xt, C, P.pk(wp, (PackId)(K_ROWX_T + d)), C, zx + (size_t)d * G.TM * 4 * U, 4 * U, (int)G.TM, 4 * U, C,
                         prm + P.poff[P_ROWF_B + 2 * d], false, st));
    const Drop nodrop = {0u, 1.f, 0u, 0, 0, M};
    for (int step = 0; step < T; ++step) {
        for (int d = 0; d < 2; ++d) {
            const int w = d == 0 ? step : T - 1 - step, pw = d == 0 ? w - 1 : w + 1;     // the position the state comes from
            const bool has = pw >= 0 && pw < T;
            const size_t dz4 = (size_t)d * G.TM * 4 * U, du = (size_t)d * G.TM * U;
            RStep k; memset(&k, 0, sizeof(k));
            k.M = M; k.U = U; k.O = U; k.dr = nodrop; k.zx_row = -1; k.epi = RS_LSTM_FWD;
            const size_t prev = du + (size_t)(has ? pw : 0) * M * U, cur = du + (size_t)w * M * U;
            if (has) k.A = bf ? (const void*)(rhb + prev) : (const void*)(rh + prev);
            else k.A = zero;                                                            // zero initial state (all-zero bits in either dtype)
            k.lda = U; k.W = P.pk(wp, (PackId)(K_ROWH_T + d)); k.ldw = U; k.N = 4 * U; k.K = U;
            k.zx = zx + dz4 + (size_t)w * M * 4 * U; k.c_prev = has ? rc + prev : zero;
            k.gates = rg + dz4 + (size_t)w * M * 4 * U; k.c_out = rc + cur;
            k.out = rh + cur; k.out2 = rh + cur; k.ldo = U;
            if (bf) { k.outb = rhb + cur; k.out2b = rhb + cur; k.ldob = U; }
            RC(lxo_launch_rstep(P.s.dtype, bf, k, st));
        }
    }
    const long long total = (long long)G.TM * (2 * U / 4);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (bf) hipLaunchKernelGGL((rowenc_assemble_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, rh, rh + G.TM * U, (bf16_t*)img, M, T, U);
    else hipLaunchKernelGGL((rowenc_assemble_kernel<float>), dim3(blocks), dim3(256), 0, st, rh, rh + G.TM * U, (float*)img, M, T, U);
    return (int)hipGetLastError();
}

// d_img (f32, gradient w.r.t. the row encoder's OUTPUT, (b, h, w) row order) -> d_img (gradient w.r.t. its input), plus the
// four parameter gradients accumulated into grads
int lxo_impl_rowenc_bwd(const Plan& P, const float* prm, const void* wp, void* ws, float* grads, hipStream_t st) {
    (void)prm;
    const RowGeo G = geo(P);
    const int M = G.M, T = G.T, U = G.U, C = G.C;
    const bool bf = P.bf;
    float* dimg = P.ws<float>(ws, W_DIMG);
    float* dh = P.ws<float>(ws, W_RDH);
    const char* xt = P.ws<char>(ws, W_RXT);
    float* rg = P.ws<float>(ws, W_RG); float* rc = P.ws<float>(ws, W_RC);
    float* rh = P.ws<float>(ws, W_RH); bf16_t* rhb = P.ws<bf16_t>(ws, W_RHB);
    float* dz = P.ws<float>(ws, W_RDZ); bf16_t* dzb = P.ws<bf16_t>(ws, W_RDZB);
    float* dcc = P.ws<float>(ws, W_RDCC); float* zero = P.ws<float>(ws, W_RZERO);
    float* dxt = P.ws<float>(ws, W_RZX);                                                // the x-part pre-activations are dead by now
    RC(permute(true, dimg, dh, M, T, (size_t)C * 4, st));
    HIPRC(hipMemsetAsync(dcc, 0, (size_t)2 * M * U * 4, st));
    HIPRC(hipMemsetAsync(zero, 0, (size_t)M * 4 * U * 4, st));
    const Drop nodrop = {0u, 1.f, 0u, 0, 0, M};
    for (int step = T - 1; step >= 0; --step) {
        for (int d = 0; d < 2; ++d) {
            const int w = d == 0 ? step : T - 1 - step;
            const int pw = d == 0 ? w - 1 : w + 1, nw = d == 0 ? w + 1 : w - 1;          // predecessor / successor in the direction's time order
            const bool hasp = pw >= 0 && pw < T, hasn = nw >= 0 && nw < T;
            const size_t dz4 = (size_t)d * G.TM * 4 * U, du = (size_t)d * G.TM * U;
            RStep k; memset(&k, 0, sizeof(k));
            k.M = M; k.U = U; k.O = U; k.dr = nodrop; k.zx_row = -1; k.epi = RS_LSTM_BWD;
            // acc = d_z(successor) Kh^T: the d_h this step receives through the recurrence
            const size_t nxt = dz4 + (size_t)(hasn ? nw : 0) * M * 4 * U;
            if (hasn) k.A = bf ? (const void*)(dzb + nxt) : (const void*)(dz + nxt);
            else k.A = zero;
            k.lda = 4 * U; k.W = P.pk(wp, (PackId)(K_ROWH + d)); k.ldw = 4 * U; k.N = U; k.K = 4 * U;
            k.dhm = dh + (size_t)w * M * C + (size_t)d * U; k.lddhm = C;                 // this direction's half of the upstream gradient
            k.carry_h = zero; k.carry_rows = 0;
            k.gates_in = rg + dz4 + (size_t)w * M * 4 * U;
            k.c_cur = rc + du + (size_t)w * M * U; k.c_prev = hasp ? rc + du + (size_t)pw * M * U : zero;
            k.dcc = dcc + (size_t)d * M * U;
            k.out = dz + dz4 + (size_t)w * M * 4 * U;
            if (bf) { k.outb = dzb + dz4 + (size_t)w * M * 4 * U; k.ldob = 4 * U; }
            RC(lxo_launch_rstep(P.s.dtype, bf, k, st));
        }
    }
    for (int d = 0; d < 2; ++d) {
        const size_t dz4 = (size_t)d * G.TM * 4 * U, du = (size_t)d * G.TM * U;
        float* gK = grads + P.poff[P_ROWF_K + 2 * d];
        RC(lxo_k_colsum(dz + dz4, 4 * U, grads + P.poff[P_ROWF_B + 2 * d], (long long)G.TM, 4 * U, P.det_scratch(ws), st));
        const void* dzop = bf ? (const void*)(dzb + dz4) : (const void*)(dz + dz4);
        const size_t opsz = bf ? 2 : 4;
        // d_Kx = X^T d_Z over all positions
        RC(gemm_tn_acc(P, xt, C, dzop, 4 * U, gK, 4 * U, (int)G.TM, C, 4 * U, st));
        // d_Kh = sum_w h(pred(w))^T d_z(w): forward direction pairs h blocks 0..T-2 with d_z blocks 1..T-1, backward the other way round
        if (T > 1) {
            const size_t blk_h = (size_t)M * U, blk_z = (size_t)M * 4 * U;
            const char* hop = bf ? (const char*)(rhb + du) : (const char*)(rh + du);
            const char* a = hop + (d == 0 ? 0 : blk_h * opsz);
            const char* b2 = (const char*)dzop + (d == 0 ? blk_z * opsz : 0);
            RC(gemm_tn_acc(P, a, U, b2, 4 * U, gK + (size_t)C * 4 * U, 4 * U, (T - 1) * M, U, 4 * U, st));
        }
        // d_X (time-major) = d_Z Kx^T, both directions summed
        RC(gemm_nt_plain(P, false, dzop, 4 * U, P.pk(wp, (PackId)(K_ROWX + d)), 4 * U, dxt, C, (int)G.TM, C, 4 * U, nullptr, d == 1, st));
    }
    RC(permute(false, dxt, dimg, M, T, (size_t)C * 4, st));
    return 0;
}
