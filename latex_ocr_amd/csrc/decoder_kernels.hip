// Decoder kernels that are not GEMMs.  One workgroup of 4 waves per sample for the
// attention stream (attention_mechanism.py:46-94): scores, softmax over the R regions
// and the context sum in ONE launch, att_img and img each read exactly once.
#include "decoder_kernels.h"
#include "api_util.h"
#include "drop.h"
#include <stdlib.h>

namespace {


// 4 consecutive k of one row as floats
LXO_DEV void load4(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
LXO_DEV void load4(const bf16_t* p, float (&v)[4]) {
    u32x2 a = *reinterpret_cast<const u32x2*>(p);
    v[0] = __uint_as_float(a[0] << 16); v[1] = __uint_as_float(a[0] & 0xffff0000u);
    v[2] = __uint_as_float(a[1] << 16); v[3] = __uint_as_float(a[1] & 0xffff0000u);
}
LXO_DEV void store4(float* p, const float (&v)[4]) { f32x4 a = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = a; }
LXO_DEV void store4(bf16_t* p, const float (&v)[4]) { u32x2 a = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}; *reinterpret_cast<u32x2*>(p) = a; }

// sum of the n split-K partial products written by gemm_slab_kernel: value(row, col) = sum_s p[s*stride + row*ld + col]
LXO_DEV float slab_sum(const Slabs& sl, long long row, int col) {
    float v = 0.f;
    const float* q = sl.p + row * sl.ld + col;
    for (int s = 0; s < sl.n; ++s) v += q[(long long)s * sl.stride];
    return v;
}

// 4 consecutive columns (col % 4 == 0, ld % 4 == 0)
LXO_DEV f32x4 slab_sum4(const Slabs& sl, long long row, int col) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* q = sl.p + row * sl.ld + col;
    for (int s0 = 0; s0 < sl.n; s0 += 8) {              // 8 slab loads in flight at a time
        f32x4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            t[j] = (s0 + j < sl.n) ? *reinterpret_cast<const f32x4*>(q + (long long)(s0 + j) * sl.stride) : z;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    return v;
}

// mean over regions: img [B][R][C] -> mean [B][C]   (attention_mechanism.py:148)
// grid (C / 64, B): a workgroup owns 64 channels of one image (8 lanes x 8 channels) and sums 32 rows at a time, so
// that B * C / 64 workgroups stream the feature map instead of B (deterministic: no atomics in the forward path)
template <typename CT>
__global__ __launch_bounds__(256) void rowmean_kernel(const CT* __restrict__ img, float* __restrict__ mean, int R, int C) {
    __shared__ float red[32][64 + 1];
    const int b = blockIdx.y, c0 = blockIdx.x * 64 + (threadIdx.x & 7) * 8, rg = threadIdx.x >> 3;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int r = rg; r < R; r += 32) {
        float v[8];
        load8(img + ((long long)b * R + r) * C + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][(threadIdx.x & 7) * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][threadIdx.x];
        mean[(long long)b * C + blockIdx.x * 64 + threadIdx.x] = t / (float)R;
    }
}

// teacher-forcing inputs (decoder.py:75-95): row t*B+b = t ? table[formula[b][t-1]] : start_token
template <typename CT>
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const float* __restrict__ start,
                                                          const int* __restrict__ formula, CT* __restrict__ out,
                                                          int B, int T, int D, int Dp, int V) {
    const long long total = (long long)T * B * Dp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % Dp);
        const long long row = i / Dp;
        const int b = (int)(row % B), t = (int)(row / B);
        float v = 0.f;
        if (d < D) {
            if (t == 0) v = start[d];
            else {
                int id = formula[(long long)b * T + t - 1];
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);
                v = table[(long long)id * D + d];
            }
        }
        out[i] = from_f32<CT>(v);
    }
}
// rows gathered by explicit ids (decode): out[v] = ids ? table[ids[v]] : start
template <typename CT>
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ table, const float* __restrict__ start,
                                                        const int* __restrict__ ids, CT* __restrict__ out,
                                                        int n, int D, int Dp, int V) {
    const long long total = (long long)n * Dp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % Dp);
        const int r = (int)(i / Dp);
        float v = 0.f;
        if (d < D) {
            if (!ids) v = start[d];
            else {
                int id = ids[r];
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);
                v = table[(long long)id * D + d];
            }
        }
        out[i] = from_f32<CT>(v);
    }
}

// decode: every possible next input once -- row v < V = embedding_table[v], row V = start_token (decoder.py:75-95, greedy_decoder_cell.py:40-43,59)
template <typename CT>
__global__ __launch_bounds__(256) void embed_table_kernel(const float* __restrict__ table, const float* __restrict__ start, CT* __restrict__ out,
                                                         int V, int D, int Dp) {
    const long long total = (long long)(V + 1) * Dp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % Dp);
        const int r = (int)(i / Dp);
        float v = 0.f;
        if (d < D) v = r < V ? table[(long long)r * D + d] : start[d];
        out[i] = from_f32<CT>(v);
    }
}

// TF-1.12 LSTMCell, gate order i,j,f,o, forget_bias 1.0 (attention_cell.py:71).  A group of 4 threads owns 4 units of
// one row: thread q sums gate q's pre-activations for the 4 units (z + the K1 slabs: one round trip of 16-byte loads),
// activates them and hands them over through LDS; then thread q finishes unit u + q.  B*U threads (128 workgroups at
// B = 64) instead of B*U/4: the step kernels are latency-bound, so the wider launch is the faster one.
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ z, Slabs zs, const float* __restrict__ c_prev,
                                                      float* __restrict__ gates, float* __restrict__ c_out,
                                                      float* __restrict__ h_out, float* __restrict__ ht_out, int ldh, Drop dr,
                                                      int B, int U) {
    __shared__ float act[64][4][4 + 1];
    const int total = B * (U >> 2);                       // unit groups
    const int q = threadIdx.x & 3, ugl = threadIdx.x >> 2;
    for (int base = blockIdx.x * 64; base < total; base += gridDim.x * 64) {
        const int ug = base + ugl;
        const bool ok = ug < total;
        const int b = ok ? ug / (U >> 2) : 0, u = ok ? (ug - b * (U >> 2)) << 2 : 0;
        float cp = 0.f;
        if (ok) {
            cp = c_prev[(long long)b * U + u + q];
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + (long long)b * 4 * U + q * U + u) + slab_sum4(zs, b, q * U + u);
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                a[e] = (q == 1) ? tanhf(zz[e]) : sigmoidf_(q == 2 ? zz[e] + 1.0f : zz[e]);
            if (gates) *reinterpret_cast<f32x4*>(gates + (long long)b * 4 * U + q * U + u) = a;
#pragma unroll
            for (int e = 0; e < 4; ++e) act[ugl][q][e] = a[e];
        }
        __syncthreads();
        if (ok) {
            const float gi = act[ugl][0][q], gj = act[ugl][1][q], gf = act[ugl][2][q], go = act[ugl][3][q];
            const float c = gf * cp + gi * gj;
            const float h = go * tanhf(c);
            c_out[(long long)b * U + u + q] = c;
            h_out[(long long)b * ldh + u + q] = h;
            // h~ = dropout(h): what attention and the o projection read; the LSTM carries the un-dropped h (attention_cell.py:71-72)
            ht_out[(long long)b * ldh + u + q] = h * drop_scale(dr, 1u, b, u + q, U);
        }
        __syncthreads();
    }
}

// slabs q, q+4, q+8, ... of a split-K product (4 consecutive columns)
LXO_DEV f32x4 slab_part4(const Slabs& sl, long long row, int col, int q) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* p = sl.p + row * sl.ld + col;
    for (int s0 = q; s0 < sl.n; s0 += 32) {               // 8 loads in flight per round (one round up to 32 slabs)
        f32x4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            t[j] = (s0 + 4 * j < sl.n) ? *reinterpret_cast<const f32x4*>(p + (long long)(s0 + 4 * j) * sl.stride) : z;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    return v;
}

// Backward of the LSTM cell.  As in lstm_fwd, 4 threads share a group of 4 units: thread q adds every fourth slab of
// the three split-K products that make up d_h (B1: o projection, B3: attention, B4: the next step's LSTM carry), the
// partial sums meet in LDS, and thread q finishes unit u + q with scalar accesses.  B*U threads, <= 8 loads each.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                      const float* __restrict__ c_cur, Slabs s1, Slabs s3, Slabs s4, int off4,
                                                      float* __restrict__ dcc, float* __restrict__ dz, Drop dr, int carry_rows,
                                                      int B, int U) {
    __shared__ float ex[64][4][8 + 1];
    const int total = B * (U >> 2);
    const int q = threadIdx.x & 3, ugl = threadIdx.x >> 2;
    for (int base = blockIdx.x * 64; base < total; base += gridDim.x * 64) {
        const int ug = base + ugl;
        const bool ok = ug < total;
        const int b = ok ? ug / (U >> 2) : 0, u = ok ? (ug - b * (U >> 2)) << 2 : 0;
        float gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f, cc = 0.f, cp = 0.f, dci = 0.f;
        if (ok) {
            const float* gr = gates + (long long)b * 4 * U + u + q;
            gi = gr[0]; gj = gr[U]; gf = gr[2 * U]; go = gr[3 * U];
            cc = c_cur[(long long)b * U + u + q]; cp = c_prev[(long long)b * U + u + q]; dci = dcc[(long long)b * U + u + q];
            const f32x4 pm = slab_part4(s1, b, u, q) + slab_part4(s3, b, u, q);          // d_h~ (o projection + attention)
            f32x4 pc = {0.f, 0.f, 0.f, 0.f};
            if (b < carry_rows) pc = slab_part4(s4, b, off4 + u, q);                    // d_h carried by the next step's LSTM
#pragma unroll
            for (int e = 0; e < 4; ++e) { ex[ugl][q][e] = pm[e]; ex[ugl][q][4 + e] = pc[e]; }
        }
        __syncthreads();
        if (ok) {
            const float dhm = ex[ugl][0][q] + ex[ugl][1][q] + ex[ugl][2][q] + ex[ugl][3][q];
            const float dhc = ex[ugl][0][4 + q] + ex[ugl][1][4 + q] + ex[ugl][2][4 + q] + ex[ugl][3][4 + q];
            const float dh = dhm * drop_scale(dr, 1u, b, u + q, U) + dhc;
            const float tc = tanhf(cc);
            const float dc = dci + dh * go * (1.f - tc * tc);
            float* dzr = dz + (long long)b * 4 * U + u + q;
            dzr[0] = dc * gj * gi * (1.f - gi);
            dzr[U] = dc * gi * (1.f - gj * gj);
            dzr[2 * U] = dc * cp * gf * (1.f - gf);
            dzr[3 * U] = dh * tc * go * (1.f - go);
            dcc[(long long)b * U + u + q] = dc * gf;
        }
        __syncthreads();
    }
}

// g = (d_o from the logits + d_o carry) * (1 - o^2)      (backward through o = tanh(.), attention_cell.py:82)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ a, int lda, Slabs carry,
                                                      const float* __restrict__ o, int ldo, float* __restrict__ g, int ldg,
                                                      bf16_t* __restrict__ gb, int ldgb, Drop dr, int carry_rows, int rows, int cols) {
    const int total = rows * (cols >> 2);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / (cols >> 2), c = (i - r * (cols >> 2)) << 2;
        const f32x4 ov = *reinterpret_cast<const f32x4*>(o + (long long)r * ldo + c);
        f32x4 d = *reinterpret_cast<const f32x4*>(a + (long long)r * lda + c);
        if (r < carry_rows) d += slab_sum4(carry, r, c);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // rec holds the dropped o = tanh * m/keep; where m = 1 the pre-dropout tanh is o * keep
            const float sc = drop_scale(dr, 2u, r, c + e, cols);
            const float th = (dr.thr == 0u) ? ov[e] : ov[e] / dr.inv_keep;
            out[e] = d[e] * sc * (1.f - th * th);
        }
        *reinterpret_cast<f32x4*>(g + (long long)r * ldg + c) = out;
        if (gb) { u32x2 pk = {pack_bf2(out[0], out[1]), pack_bf2(out[2], out[3])}; *reinterpret_cast<u32x2*>(gb + (long long)r * ldgb + c) = pk; }
    }
}
// o = tanh(sum of the K4 slabs) -> rec      (attention_cell.py:82)
__global__ __launch_bounds__(256) void tanh_finalize_kernel(Slabs sl, float* __restrict__ o, int ldo, Drop dr, int rows, int cols) {
    const int total = rows * (cols >> 2);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / (cols >> 2), c = (i - r * (cols >> 2)) << 2;
        const f32x4 v = slab_sum4(sl, r, c);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = tanhf(v[e]) * drop_scale(dr, 2u, r, c + e, cols);
        *reinterpret_cast<f32x4*>(o + (long long)r * ldo + c) = out;
    }
}

// ---- attention stream ----
// The R regions of a sample are split into NCH chunks, one workgroup (8 waves) per
// (chunk, sample), so that B*NCH >= ~2 workgroups per CU and every wave keeps 4 rows of
// both streams in flight (the stream is latency-bound otherwise: one 512-B row per wave per
// round trip measured 5 GB/s per workgroup).  Each chunk produces flash-style partials
// (max, sum, unnormalised context); attn_fwd_combine normalises.
constexpr int ATT_W = 8;        // waves per workgroup

template <typename CT> LXO_DEV float tanh_ct(float x);
template <> LXO_DEV float tanh_ct<float>(float x) { return tanhf(x); }
// bf16 mode: 1 - 2/(e^{2x}+1) on v_exp_f32 / v_rcp_f32 (abs error ~1e-7, far below bf16 resolution)
template <> LXO_DEV float tanh_ct<bf16_t>(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// Forward: ONE pass over both streams with a per-wave online softmax (running max / sum / context),
// so the att_img row and the img row of 8 regions are all in flight together; the raw scores go to
// `alpha` and attn_fwd_combine turns them into normalised weights.
template <typename CT, int KCT, int ATT_U>
__global__ __launch_bounds__(512) void attn_fwd_part_kernel(const CT* __restrict__ att_img, const CT* __restrict__ img,
                                                           const float* __restrict__ att_h, Slabs ahs, float* __restrict__ att_h_out,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ alpha, float* __restrict__ part,
                                                           int R, int Rp, int E, int C, int beam, int nch, int rows_per, int rev) {
    __shared__ float redc[ATT_W][512];
    __shared__ float red[2 * ATT_W];
    const int ch = blockIdx.x, v = blockIdx.y, bi = v / beam;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = ch * rows_per;
    const int n = min(R, r0 + rows_per) - r0;          // may be <= 0 for a trailing chunk
    const CT* ai = att_img + ((long long)bi * R + r0) * E;
    const CT* im = img + ((long long)bi * R + r0) * C;
    float* pout = part + ((long long)v * nch + ch) * (C + 2);
    constexpr int KC = KCT;
    float ah[KCT][4], bt[KCT][4];
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
        const int k0 = kc * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ah[kc][j] = 0.f; bt[kc][j] = 0.f; }
        if (kc < KC && k0 < E) {
            const f32x4 a4 = ahs.n > 0 ? slab_sum4(ahs, v, k0) : *reinterpret_cast<const f32x4*>(att_h + (long long)v * E + k0);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ah[kc][j] = a4[j]; bt[kc][j] = b4[j]; }
            if (ahs.n > 0 && att_h_out && ch == 0 && wave == 0) *reinterpret_cast<f32x4*>(att_h_out + (long long)v * E + k0) = a4;
        }
    }
    const int c0 = lane * 8;
    const bool cok = c0 < C;
    float m = -3.0e38f, l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // Every load of the loop is UNCONDITIONAL (row and channel indices clamped into the chunk, the contribution of a clamped
    // row or lane masked afterwards): with `if (r < n) load` hipcc branched around each load and put `s_waitcnt vmcnt(0)` behind
    // it, so the 8 row loads of an iteration were 8 serial memory round trips instead of 8 requests in flight.
    const int c0l = cok ? c0 : 0;                            // lanes beyond C re-read channel 0 (their accumulators are never stored)
    // rev: walk the chunk's row blocks last to first.  The streams are re-read by every step of the recurrence with the same
    // workgroup -> XCD mapping, 10.6 MB per XCD through a 4 MB L2: read in the same order every time, LRU keeps nothing; with
    // the direction alternating from step to step, the blocks a launch read last are the ones the next launch reads first.
    const int nit = n > wave ? (n - wave + ATT_W * ATT_U - 1) / (ATT_W * ATT_U) : 0;
    for (int it = 0; it < nit; ++it) {
        const int base = wave + ATT_W * ATT_U * (rev ? nit - 1 - it : it);
        float xi[ATT_U][8], pt[ATT_U];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {                 // issue the img rows first: they are consumed last
            const int r = min(base + ATT_W * u, n - 1);
            load8(im + (long long)r * C + c0l, xi[u]);
            pt[u] = 0.f;
        }
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
            const bool kok = kc < KC && k0 < E;
            const int k0l = kok ? k0 : 0;
            float x[ATT_U][4];
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) load4(ai + (long long)min(base + ATT_W * u, n - 1) * E + k0l, x[u]);
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) a = fmaf(tanh_ct<CT>(x[u][j] + ah[kc][j]), bt[kc][j], a);
                pt[u] += kok ? a : 0.f;                     // bt is zero for masked lanes anyway; keep NaN-free
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) pt[u] += __shfl_xor(pt[u], o);
        }
        float mn = m;
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) if (base + ATT_W * u < n) mn = fmaxf(mn, pt[u]);
        const float sc = expf(m - mn);                    // wave-uniform rescale of the running sums
        l *= sc;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= sc;
        m = mn;
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int r = base + ATT_W * u;
            if (r < n) {
                const float pw = expf(pt[u] - m);
                l += pw;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pw, xi[u][e], acc[e]);
                if (lane == 0) alpha[(long long)v * Rp + r0 + r] = pt[u];       // raw score
            }
        }
    }
    // merge the 8 waves
    if (lane == 0) red[wave] = m;
    __syncthreads();
    float mc = red[0];
#pragma unroll
    for (int w = 1; w < ATT_W; ++w) mc = fmaxf(mc, red[w]);
    const float sw = (l > 0.f) ? expf(m - mc) : 0.f;
    if (lane == 0) red[ATT_W + wave] = l * sw;
    if (cok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) redc[wave][c0 + e] = acc[e] * sw;
    }
    __syncthreads();
    if (tid == 0) {
        float lt = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_W; ++w) lt += red[ATT_W + w];
        pout[0] = mc; pout[1] = lt;
    }
    for (int c = tid; c < C; c += 512) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_W; ++w) t += redc[w][c];
        pout[2 + c] = t;
    }
}


// The same pass for BEAM SEARCH (beam_search_decoder_cell.py:98-109 tiles the image features over the beam; here they are not tiled):
// the NBM hypotheses of an image read the SAME att_img / img rows, so one workgroup = (chunk, IMAGE) loads every row once and runs the
// scores / online softmax / context of all NBM decoder rows v = image * NBM + b on it.  attn_fwd_part_kernel with its (chunk, decoder row)
// grid streamed the image's rows once per hypothesis: 427 MB per beam-5 step at B = 64 instead of 85 MB, 67 us of a 161 us step
// (profiles/r05_beam_kernels_before.csv).  The arithmetic of a (row, hypothesis) pair is the per-row kernel's, in the same order (same
// butterfly, same online-softmax updates), so the partials -- and with them the ids of the f32 parity mode -- are bit-identical to the
// per-row kernel's for the same chunking.
// wave-wide sum without the LDS crossbar (v_add_f32_dpp inside a row of 16 lanes, v_permlane16 / 32_swap across rows: 69 ns per
// dependent 64-lane reduction against 197 through ds_bpermute, tools/dpp_probe.hip): the forms of csrc/xdec.hip, for the bf16 instantiation
#ifdef LXO_HIPSIM
// (tests/hipsim has no DPP / permlane-swap model: the interpreter build sums through the shuffles it does model)
LXO_DEV float bm_wave_sum_dpp(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
#else
template <int CTRL> LXO_DEV float bm_dpp_add(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
LXO_DEV float bm_wave_sum_dpp(float v) {
    v = bm_dpp_add<0xB1>(v); v = bm_dpp_add<0x4E>(v); v = bm_dpp_add<0x141>(v); v = bm_dpp_add<0x140>(v);
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
#endif
// EXPD (bf16 decode): att_img holds E_x = e^{2 att_img} (ws region att_exp, written once per decode call); with E_a = e^{2 att_h} and
// r = 1 / (1 + E_x E_a), tanh = 1 - 2 r and the score is sum_k beta_k - 2 sum_k beta_k r_k: the constant is the same for every region of a
// hypothesis, the softmax does not see it, it is dropped -- ONE transcendental per element instead of two (as in the training chains);
// the softmax exponentials on v_exp_f32, the 64-lane sums on DPP.  ~400 -> ~200 issue cycles per (row, hypothesis): the kernel is bound
// by that arithmetic (5 hypotheses per loaded row), not by the stream.
// OCC = waves per SIMD the register allocation must admit: 4 = two 8-wave workgroups per CU (the grid of 512 workgroups in ONE round, and
// twice the waves to cover the loads) -- the 5-hypothesis instantiation gets there with 2 rows per wave in flight instead of 4 (118 VGPRs;
// with 4 rows it needs 142 and forcing it into 128 spills 16 dwords)
template <typename CT, int ATT_U, int NBM, bool EXPD, int OCC>
__global__ __launch_bounds__(512, OCC) void attn_fwd_part_beam_kernel(const CT* __restrict__ att_img, const CT* __restrict__ img,
                                                                const float* __restrict__ att_h, const float* __restrict__ beta,
                                                                float* __restrict__ alpha, float* __restrict__ part,
                                                                int R, int Rp, int E, int C, int nch, int rows_per, int rev) {
    __shared__ float redc[ATT_W][512];
    __shared__ float red[2 * ATT_W];
    const int ch = blockIdx.x, bi = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = ch * rows_per;
    const int n = min(R, r0 + rows_per) - r0;          // may be <= 0 for a trailing chunk
    const CT* ai = att_img + ((long long)bi * R + r0) * E;
    const CT* im = img + ((long long)bi * R + r0) * C;
    const int k0 = lane * 4;
    const bool kok = k0 < E;
    float ah[NBM][4], bt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bt[j] = 0.f;
    if (kok) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + k0); for (int j = 0; j < 4; ++j) bt[j] = EXPD ? -2.f * b4[j] : b4[j]; }
#pragma unroll
    for (int b = 0; b < NBM; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ah[b][j] = 0.f;
        if (kok) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(att_h + (long long)(bi * NBM + b) * E + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) ah[b][j] = EXPD ? __builtin_amdgcn_exp2f(fminf(fmaxf(a4[j] * 2.8853900817779268f, -60.f), 60.f)) : a4[j];
        }
    }
    const int c0 = lane * 8;
    const bool cok = c0 < C;
    const int c0l = cok ? c0 : 0, k0l = kok ? k0 : 0;
    float m[NBM], l[NBM], acc[NBM][8];
#pragma unroll
    for (int b = 0; b < NBM; ++b) {
        m[b] = -3.0e38f; l[b] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
    }
    const int nit = n > wave ? (n - wave + ATT_W * ATT_U - 1) / (ATT_W * ATT_U) : 0;
    for (int it = 0; it < nit; ++it) {
        const int base = wave + ATT_W * ATT_U * (rev ? nit - 1 - it : it);
        float xi[ATT_U][8], x[ATT_U][4];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {                 // unconditional, clamped (see attn_fwd_part_kernel)
            const int r = min(base + ATT_W * u, n - 1);
            load8(im + (long long)r * C + c0l, xi[u]);
            load4(ai + (long long)r * E + k0l, x[u]);
        }
#pragma unroll
        for (int b = 0; b < NBM; ++b) {
            float pt[ATT_U];
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) {
                float a = 0.f;
                if constexpr (EXPD) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a = fmaf(__builtin_amdgcn_rcpf(fmaf(x[u][j], ah[b][j], 1.f)), bt[j], a);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a = fmaf(tanh_ct<CT>(x[u][j] + ah[b][j]), bt[j], a);
                }
                pt[u] = kok ? a : 0.f;
            }
            if constexpr (EXPD) {
#pragma unroll
                for (int u = 0; u < ATT_U; ++u) pt[u] = bm_wave_sum_dpp(pt[u]);
            } else {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
                    for (int u = 0; u < ATT_U; ++u) pt[u] += __shfl_xor(pt[u], o);
                }
            }
            float mn = m[b];
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) if (base + ATT_W * u < n) mn = fmaxf(mn, pt[u]);
            if (mn != m[b]) {                                    // wave-uniform (the scores are wave sums): a running maximum that stands still rescales by exp(0) = 1 exactly -- skipped
                const float sc = EXPD ? __expf(m[b] - mn) : expf(m[b] - mn);
                l[b] *= sc;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[b][e] *= sc;
                m[b] = mn;
            }
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) {
                const int r = base + ATT_W * u;
                if (r < n) {
                    const float pw = EXPD ? __expf(pt[u] - mn) : expf(pt[u] - mn);
                    l[b] += pw;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[b][e] = fmaf(pw, xi[u][e], acc[b][e]);
                    if (lane == 0) alpha[(long long)(bi * NBM + b) * Rp + r0 + r] = pt[u];       // raw score (EXPD: up to the constant sum of beta, which the softmax does not see)
                }
            }
        }
    }
    // merge the 8 waves, one hypothesis after the other (the LDS tile is shared)
#pragma unroll
    for (int b = 0; b < NBM; ++b) {
        float* pout = part + ((long long)(bi * NBM + b) * nch + ch) * (C + 2);
        if (b) __syncthreads();
        if (lane == 0) red[wave] = m[b];
        __syncthreads();
        float mc = red[0];
#pragma unroll
        for (int w = 1; w < ATT_W; ++w) mc = fmaxf(mc, red[w]);
        const float sw = (l[b] > 0.f) ? (EXPD ? __expf(m[b] - mc) : expf(m[b] - mc)) : 0.f;
        if (lane == 0) red[ATT_W + wave] = l[b] * sw;
        if (cok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) redc[wave][c0 + e] = acc[b][e] * sw;
        }
        __syncthreads();
        if (tid == 0) {
            float lt = 0.f;
#pragma unroll
            for (int w = 0; w < ATT_W; ++w) lt += red[ATT_W + w];
            pout[0] = mc; pout[1] = lt;
        }
        for (int c = tid; c < C; c += 512) {
            float tt = 0.f;
#pragma unroll
            for (int w = 0; w < ATT_W; ++w) tt += redc[w][c];
            pout[2 + c] = tt;
        }
    }
}

// merge the chunk partials: alpha = exp(e - m) / l ; ctx = sum_c ctx_c exp(m_c - m) / l.
// grid (4, nv): every workgroup recomputes the (tiny) chunk scales in registers, then handles a
// quarter of the channels and a quarter of the regions -- no serial phase, no barrier.
__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(const float* __restrict__ part, float* __restrict__ alpha,
                                                              float* __restrict__ ctx, int ldctx, bf16_t* __restrict__ ctxb, int ldcb, int R, int Rp, int C,
                                                              int nch, int rows_per) {
    const int v = blockIdx.y, qd = blockIdx.x, tid = threadIdx.x;
    const float* pv = part + (long long)v * nch * (C + 2);
    // everything this thread needs is requested up front (chunk statistics, its channel's partial contexts, its raw
    // scores) so that the kernel costs ONE memory round trip, not one for the statistics and a second for the data
    const int cq = (C + 3) / 4, rq = (R + 3) / 4;
    const int c = qd * cq + tid, r = qd * rq + tid;           // cq, rq <= 256 for C <= 1024, R <= 1024 (else the loops below)
    const bool c_ok = tid < cq && c < C, r_ok = tid < rq && r < R;
    const bool fast = nch <= 8 && cq <= 256 && rq <= 256;
    if (fast) {
        // every operand is requested up front by UNCONDITIONAL loads (chunk index clamped, the clamped copies masked by a
        // zero sum): conditional loads made hipcc wait behind each of them, one memory round trip per chunk statistic
        const int cl = c_ok ? c : 0, rl = r_ok ? r : 0;
        float pc8[8], m8[8], l8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long o = (long long)min(k, nch - 1) * (C + 2);
            m8[k] = pv[o]; l8[k] = pv[o + 1]; pc8[k] = pv[o + 2 + cl];
        }
        const float sraw = alpha[(long long)v * Rp + rl];
        float m = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (k >= nch) l8[k] = 0.f; if (l8[k] > 0.f) m = fmaxf(m, m8[k]); }
        float l = 0.f, t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float w = l8[k] > 0.f ? expf(m8[k] - m) : 0.f; l += l8[k] * w; t = fmaf(pc8[k], w, t); }
        const float inv = 1.0f / l;
        if (c_ok) {
            ctx[(long long)v * ldctx + c] = t * inv;
            if (ctxb) ctxb[(long long)v * ldcb + c] = f2bf(t * inv);      // bf16 mirror: A operand of the fused o projection
        }
        if (r_ok) alpha[(long long)v * Rp + r] = expf(sraw - m) * inv;
        return;
    }
    float mc[32], lc[32];
    float m = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        mc[k] = k < nch ? pv[(long long)k * (C + 2)] : -3.0e38f;
        lc[k] = k < nch ? pv[(long long)k * (C + 2) + 1] : 0.f;
        if (lc[k] > 0.f) m = fmaxf(m, mc[k]);
    }
    float l = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { mc[k] = lc[k] > 0.f ? expf(mc[k] - m) : 0.f; l += lc[k] * mc[k]; }
    const float inv = 1.0f / l;
    for (int cc = qd * cq + tid; cc < min(C, (qd + 1) * cq); cc += 256) {
        float t = 0.f;
        for (int k = 0; k < nch; ++k) t = fmaf(pv[(long long)k * (C + 2) + 2 + cc], mc[k], t);
        ctx[(long long)v * ldctx + cc] = t * inv;
        if (ctxb) ctxb[(long long)v * ldcb + cc] = f2bf(t * inv);
    }
    for (int rr = qd * rq + tid; rr < min(R, (qd + 1) * rq); rr += 256)
        alpha[(long long)v * Rp + rr] = expf(alpha[(long long)v * Rp + rr] - m) * inv;
}

// ---- forward attention WITHOUT a merge step (round 3 experiment, opt-in: LXO_ATT_SPLIT=1; measured SLOWER than part + combine:
// 11.1 + 12.2 us against 16.6 + 4.9 us per decoder step -- the scores kernel moves 28 MB in one burst per workgroup and pays the
// launch ramp, the att_h / beta round trip and the tail without anything to overlap them with, i.e. the fixed costs twice) ----
// The part / combine pair splits a sample's regions over workgroups, so the softmax needs a second launch that merges the
// chunk partials (4.9 us per decoder step for 1 MB of data).  Here the two streams are split instead: (1) the scores
// e[v][r] = sum_k beta_k tanh(att_img[r][k] + att_h[k]) need only `att_img` (a third of the bytes) and no merge at all;
// (2) the context is split over CHANNEL slices of 64: every workgroup of a sample recomputes the softmax statistics from the
// <= 1024 raw scores (868 exps) and accumulates its 64 channels of sum_r alpha_r img[r][:] -- no cross-workgroup step.
// Rows are fetched in 16-byte pieces (scores: 32 lanes per 512-byte row, two rows per instruction; context: 8 lanes per
// 128-byte row segment, eight rows per instruction), ATT_U instructions in flight per wave.
template <typename CT, int ATT_U>
__global__ __launch_bounds__(512) void attn_scores_kernel(const CT* __restrict__ att_img, const float* __restrict__ att_h,
                                                         const float* __restrict__ beta, float* __restrict__ raw,
                                                         int R, int Rp, int E, int beam, int rows_per) {
    constexpr int EPL = 16 / (int)sizeof(CT);                  // elements per lane and load: 8 (bf16) / 4 (f32)
    const int ch = blockIdx.x, v = blockIdx.y, bi = v / beam;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lpr = (E + EPL - 1) / EPL;                       // lanes per row (E <= 64 * EPL: one instruction covers >= 1 row)
    const int rpi = 64 / lpr;                                  // rows per load instruction
    const int rl = lane / lpr, kl = (lane - rl * lpr) * EPL;   // this lane's row within the instruction, first k
    const int r0 = ch * rows_per;
    const int n = min(R, r0 + rows_per) - r0;
    if (n <= 0) return;
    const CT* ai = att_img + ((long long)bi * R + r0) * E;
    const bool kok = rl < rpi && kl < E;
    const int kc = kok ? kl : 0;
    float ah[EPL], bt[EPL];
    {
        const float* ap = att_h + (long long)v * E + kc;
        const float* bp = beta + kc;
#pragma unroll
        for (int j = 0; j < EPL; j += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + j), b4 = *reinterpret_cast<const f32x4*>(bp + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ah[j + e] = a4[e]; bt[j + e] = kok ? b4[e] : 0.f; }
        }
    }
    const int rows_it = ATT_W * ATT_U * rpi;                    // rows a workgroup covers per iteration
    for (int base = 0; base < n; base += rows_it) {
        float x[ATT_U][EPL];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {                      // unconditional (clamped) loads: all ATT_U requests in flight together
            const int r = min(base + (u * ATT_W + wave) * rpi + rl, n - 1);
            if constexpr (EPL == 8) load8(ai + (long long)r * E + kc, x[u]);
            else load4(ai + (long long)r * E + kc, x[u]);
        }
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < EPL; ++j) a = fmaf(tanh_ct<CT>(x[u][j] + ah[j]), bt[j], a);
            for (int o = 1; o < lpr; o <<= 1) a += __shfl_xor(a, o);      // lpr is a power of two (E = 256: 32 / 64 lanes)
            const int r = base + (u * ATT_W + wave) * rpi + rl;
            if (kok && kl == 0 && r < n) raw[(long long)v * Rp + r0 + r] = a;
        }
    }
}

template <typename CT, int ATT_U>
__global__ __launch_bounds__(512) void attn_ctx_kernel(const CT* __restrict__ img, const float* __restrict__ raw, float* __restrict__ alpha,
                                                      float* __restrict__ ctx, int ldctx, bf16_t* __restrict__ ctxb, int ldcb,
                                                      int R, int Rp, int C, int beam) {
    constexpr int EPL = 16 / (int)sizeof(CT);                  // channels per lane: 8 (bf16) / 4 (f32)
    constexpr int LPR = 64 / EPL;                              // lanes per 64-channel row segment: 8 / 16
    constexpr int RPI = 64 / LPR;                              // rows per load instruction: 8 / 4
    __shared__ float sc[1024];
    __shared__ float red[ATT_W];
    __shared__ float accs[ATT_W][64];
    const int sl = blockIdx.x, ns = gridDim.x, v = blockIdx.y, bi = v / beam;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // softmax statistics of the whole sample (every slice recomputes them: R <= 1024 scores)
    const float* rv = raw + (long long)v * Rp;
    const float e0 = rv[min(tid, R - 1)], e1 = rv[min(tid + 512, R - 1)];
    float m = fmaxf(tid < R ? e0 : -3.0e38f, tid + 512 < R ? e1 : -3.0e38f);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < ATT_W; ++w) m = fmaxf(m, red[w]);
    const float p0 = tid < R ? expf(e0 - m) : 0.f, p1 = tid + 512 < R ? expf(e1 - m) : 0.f;
    float l = wave_sum(p0 + p1);
    __syncthreads();
    if (lane == 0) red[wave] = l;
    __syncthreads();
    l = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_W; ++w) l += red[w];
    const float inv = 1.0f / l;
    if (tid < R) sc[tid] = p0 * inv;
    if (tid + 512 < R) sc[tid + 512] = p1 * inv;
    // the normalised weights (kept for the backward pass / the visualisation export): slice s writes its share of the rows
    const int rq = (R + ns - 1) / ns;
    if (tid < R && tid / rq == sl) alpha[(long long)v * Rp + tid] = p0 * inv;
    if (tid + 512 < R && (tid + 512) / rq == sl) alpha[(long long)v * Rp + tid + 512] = p1 * inv;
    __syncthreads();
    // context of this slice's 64 channels
    const int rl = lane / LPR, cl = (lane - rl * LPR) * EPL;
    const CT* im = img + (long long)bi * R * C + sl * 64 + cl;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    constexpr int ROWS_IT = ATT_W * ATT_U * RPI;
    for (int base = 0; base < R; base += ROWS_IT) {
        float x[ATT_U][EPL];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int r = min(base + (u * ATT_W + wave) * RPI + rl, R - 1);
            if constexpr (EPL == 8) load8(im + (long long)r * C, x[u]);
            else load4(im + (long long)r * C, x[u]);
        }
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int r = base + (u * ATT_W + wave) * RPI + rl;
            const float w = r < R ? sc[r] : 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(w, x[u][e], acc[e]);
        }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += __shfl_xor(acc[e], o);
    }
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) accs[wave][cl + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_W; ++w) t += accs[w][tid];
        const int c = sl * 64 + tid;
        ctx[(long long)v * ldctx + c] = t;
        if (ctxb) ctxb[(long long)v * ldcb + c] = f2bf(t);
    }
}

// ---- attention backward (per step): d_e and d_att_h in one pass; d_img / d_att_img are deferred ----
// EXPD (bf16 mode): `att_img` holds E_x = e^{2x} (ws region "att_exp") instead of x.  With E_a = e^{2 att_h} formed once per launch,
// r = 1 / (1 + E_x E_a) gives tanh(x + a) = 1 - 2r and 1 - tanh^2 = 4 r (1 - r): ONE transcendental (the reciprocal) and three plain
// operations per element where the x form needs two transcendentals and six -- this kernel is bound by its vector arithmetic.
template <typename CT, int KCT, int ATT_U, bool EXPD = false>
__global__ __launch_bounds__(512) void attn_bwd_part_kernel(const CT* __restrict__ att_img, const CT* __restrict__ img,
                                                           const float* __restrict__ att_h, const float* __restrict__ beta,
                                                           const float* __restrict__ alpha, Slabs dcs, int dcoff, float* __restrict__ dctx_out, int lddc,
                                                           const float* __restrict__ ctx, int ldctx,
                                                           float* __restrict__ de, float* __restrict__ datth,
                                                           int R, int Rp, int E, int C, int rows_per, int rev) {
    __shared__ float rede[ATT_W][1024];
    __shared__ float red[ATT_W];
    const int ch = blockIdx.x, v = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = ch * rows_per;
    const int n = min(R, r0 + rows_per) - r0;
    if (n <= 0) return;                                  // block-uniform
    const CT* ai = att_img + ((long long)v * R + r0) * E;
    const CT* im = img + ((long long)v * R + r0) * C;
    constexpr int KC = KCT;
    const int c0 = lane * 8;
    const bool cok = c0 < C;
    const int c0l = cok ? c0 : 0;                            // unconditional loads, as in the forward kernel
    float s = 0.f, dc[8];
    float ah[KCT][4], acc[KCT][4];
    if (dcs.n == 1 && C <= 512) {
        // d_ctx arrives as final values (fused step kernels): ONE round of loads -- this lane's 8 channels of d_ctx and ctx,
        // its slice of att_h -- and every wave forms s = <ctx, d_ctx> = sum_r alpha_r d_alpha_r by itself (its 64 lanes cover all
        // C channels): no slab loop, no LDS reduction, no barrier, instead of four dependent memory round trips
        const float* dp = dcs.p + (long long)v * dcs.ld + dcoff + c0l;
        const float* cp = ctx + (long long)v * ldctx + c0l;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(cp), x1 = *reinterpret_cast<const f32x4*>(cp + 4);
        f32x4 a4[KCT];
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
            a4[kc] = *reinterpret_cast<const f32x4*>(att_h + (long long)v * E + ((kc < KC && k0 < E) ? k0 : 0));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { dc[e] = cok ? d0[e] : 0.f; dc[4 + e] = cok ? d1[e] : 0.f; }
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(x0[e], dc[e], fmaf(x1[e], dc[4 + e], s));
        s = wave_sum(s);
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
            const bool kok = kc < KC && k0 < E;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ah[kc][j] = kok ? a4[kc][j] : 0.f; acc[kc][j] = 0.f; }
        }
        if (ch == 0 && dctx_out) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (cok) dctx_out[(long long)v * lddc + c0 + e] = dc[e];
        }
    } else {
    // s = <ctx, d_ctx> = sum_r alpha_r d_alpha_r
    for (int c = tid; c < C; c += 512) {
        const float d = slab_sum(dcs, v, dcoff + c);
        if (ch == 0 && dctx_out) dctx_out[(long long)v * lddc + c] = d;          // summed d_ctx, kept for the deferred d_img GEMM
        s = fmaf(ctx[(long long)v * ldctx + c], d, s);
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_W; ++w) s += red[w];
#pragma unroll
    for (int e = 0; e < 8; ++e) dc[e] = 0.f;
    if (cok) {
        const f32x4 d0 = slab_sum4(dcs, v, dcoff + c0), d1 = slab_sum4(dcs, v, dcoff + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dc[e] = d0[e]; dc[4 + e] = d1[e]; }
    }
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
        const int k0 = kc * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ah[kc][j] = 0.f; acc[kc][j] = 0.f; }
        if (kc < KC && k0 < E) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(att_h + (long long)v * E + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) ah[kc][j] = a4[j];
        }
    }
    }
    if constexpr (EXPD) {
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc)
#pragma unroll
            for (int j = 0; j < 4; ++j) ah[kc][j] = __builtin_amdgcn_exp2f(fminf(fmaxf(ah[kc][j] * 2.8853900817779268f, -60.f), 60.f));     // masked lanes hold 0 -> 1: harmless (their accumulators are not stored)
    }
    const int nit = n > wave ? (n - wave + ATT_W * ATT_U - 1) / (ATT_W * ATT_U) : 0;
    for (int it = 0; it < nit; ++it) {
        const int base = wave + ATT_W * ATT_U * (rev ? nit - 1 - it : it);       // see attn_fwd_part_kernel
        float xi[ATT_U][8], pt[ATT_U], al[ATT_U];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int r = base + ATT_W * u, rc = min(r, n - 1);
            load8(im + (long long)rc * C + c0l, xi[u]);
            const float a = alpha[(long long)v * Rp + r0 + rc];
            al[u] = r < n ? a : 0.f;                        // a clamped row contributes nothing: d = al * (...) = 0
        }
        float x[KCT][ATT_U][4];
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
            const int k0l = (kc < KC && k0 < E) ? k0 : 0;
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) load4(ai + (long long)min(base + ATT_W * u, n - 1) * E + k0l, x[kc][u]);
        }
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(xi[u][e], dc[e], a);
            pt[u] = a;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < ATT_U; ++u) pt[u] += __shfl_xor(pt[u], o);
        }
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int r = base + ATT_W * u;
            const float d = al[u] * (pt[u] - s);                     // softmax backward (0 for r >= n)
            if (lane == 0 && r < n) de[(long long)v * Rp + r0 + r] = d;
#pragma unroll
            for (int kc = 0; kc < KCT; ++kc) {
                const int k0 = kc * 256 + lane * 4;
                if (kc < KC && k0 < E) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (EXPD) {
                            const float rr = __builtin_amdgcn_rcpf(fmaf(x[kc][u][j], ah[kc][j], 1.f));
                            acc[kc][j] = fmaf(4.f * d, fmaf(-rr, rr, rr), acc[kc][j]);
                        } else {
                        const float tau = tanh_ct<CT>(x[kc][u][j] + ah[kc][j]);
                        acc[kc][j] = fmaf(d, 1.f - tau * tau, acc[kc][j]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
        const int k0 = kc * 256 + lane * 4;
        if (kc < KC && k0 < E) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) rede[wave][k0 + j] = acc[kc][j] * b4[j];
        }
    }
    __syncthreads();
    for (int k = tid; k < E; k += 512) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_W; ++w) t += rede[w][k];
        atomicAdd(&datth[(long long)v * E + k], t);
    }
}

// The same for the shipped widths in bf16 mode (E = 256, C = 512, d_ctx as final values): the chunk is walked in blocks of 8 waves x ATT_U
// rows with TWO blocks in flight -- block i + 1 is requested before block i is computed (the kernel above issues a block, waits for
// all of it, computes, issues the next: one memory round trip per block with nothing under it: 16.2 us for 85 MB where the forward
// chain's stream phase moves the same bytes in 11.5).  Loads are buffer loads whose row offset is SCALAR (the row is wave-uniform) and
// whose per-lane offset is one loop-invariant register; rows beyond the chunk are clamped and masked.
typedef __amdgpu_buffer_rsrc_t ab_rsrc_t;
template <int ATT_U>
LXO_DEV void attb_load(u32x4 (&xi)[ATT_U], u32x2 (&xa)[ATT_U], float (&al)[ATT_U], ab_rsrc_t rim, ab_rsrc_t rai, ab_rsrc_t ral, int base, int n, int lane) {
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
        const int r = max(min(base + ATT_W * u, n - 1), 0);
        xi[u] = __builtin_amdgcn_raw_buffer_load_b128(rim, lane * 16, r * 1024, 0);
        xa[u] = __builtin_amdgcn_raw_buffer_load_b64(rai, lane * 8, r * 512, 0);
        al[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ral, 0, r * 4, 0));
    }
}
template <int ATT_U>
LXO_DEV void attb_block(const u32x4 (&xi)[ATT_U], const u32x2 (&xa)[ATT_U], const float (&al)[ATT_U], int base, int n, const float (&dc)[8],
                        const float (&ah)[4], float s, float (&acc)[4], float* de_row, int lane) {
    float pt[ATT_U];
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a = fmaf(__uint_as_float(xi[u][e] << 16), dc[2 * e], a);
            a = fmaf(__uint_as_float(xi[u][e] & 0xffff0000u), dc[2 * e + 1], a);
        }
        pt[u] = a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) pt[u] += __shfl_xor(pt[u], o);
    }
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
        const int r = base + ATT_W * u;
        if (r >= n) continue;                                           // wave-uniform: a clamped row costs its loads, not its arithmetic
        const float d = al[u] * (pt[u] - s);                            // softmax backward
        if (lane == 0) de_row[r] = d;
        const float x0 = __uint_as_float(xa[u][0] << 16), x1 = __uint_as_float(xa[u][0] & 0xffff0000u);
        const float x2 = __uint_as_float(xa[u][1] << 16), x3 = __uint_as_float(xa[u][1] & 0xffff0000u);
        const float t0 = tanh_ct<bf16_t>(x0 + ah[0]), t1 = tanh_ct<bf16_t>(x1 + ah[1]), t2 = tanh_ct<bf16_t>(x2 + ah[2]), t3 = tanh_ct<bf16_t>(x3 + ah[3]);
        acc[0] = fmaf(d, 1.f - t0 * t0, acc[0]); acc[1] = fmaf(d, 1.f - t1 * t1, acc[1]);
        acc[2] = fmaf(d, 1.f - t2 * t2, acc[2]); acc[3] = fmaf(d, 1.f - t3 * t3, acc[3]);
    }
}
template <int ATT_U>
__global__ __launch_bounds__(512) void attn_bwd_stream_kernel(const bf16_t* __restrict__ att_img, const bf16_t* __restrict__ img,
                                                             const float* __restrict__ att_h, const float* __restrict__ beta,
                                                             const float* __restrict__ alpha, const float* __restrict__ dctx, int lddc,
                                                             const float* __restrict__ ctx, int ldctx,
                                                             float* __restrict__ de, float* __restrict__ datth,
                                                             int R, int Rp, int rows_per, int rev) {
    constexpr int E = 256, C = 512;
    __shared__ float rede[ATT_W][E];
    const int ch = blockIdx.x, v = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = ch * rows_per;
    const int n = min(R, r0 + rows_per) - r0;
    if (n <= 0) return;                                  // block-uniform
    const ab_rsrc_t rim = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(img + ((long long)v * R + r0) * C), 0, n * C * 2, 0x00020000);
    const ab_rsrc_t rai = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(att_img + ((long long)v * R + r0) * E), 0, n * E * 2, 0x00020000);
    const ab_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(alpha + (long long)v * Rp + r0), 0, n * 4, 0x00020000);
    const int nblk2 = ((n + ATT_W * ATT_U - 1) / (ATT_W * ATT_U) + 1) & ~1;          // blocks, rounded up to pairs (an extra block is all clamped rows)
#define ABASE(i) (wave + ATT_W * ATT_U * (rev ? nblk2 - 1 - (i) : (i)))
    u32x4 xiA[ATT_U], xiB[ATT_U]; u32x2 xaA[ATT_U], xaB[ATT_U]; float alA[ATT_U], alB[ATT_U];
    attb_load<ATT_U>(xiA, xaA, alA, rim, rai, ral, ABASE(0), n, lane);               // the first two blocks are on their way before anything else is read
    attb_load<ATT_U>(xiB, xaB, alB, rim, rai, ral, ABASE(1), n, lane);
    // this lane's 8 channels of d_ctx and ctx, its 4 columns of att_h; every wave forms s = <ctx, d_ctx> = sum_r alpha_r d_alpha_r by itself
    const int c0 = lane * 8;
    const float* dp = dctx + (long long)v * lddc + c0;
    const float* cp = ctx + (long long)v * ldctx + c0;
    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(cp), x1 = *reinterpret_cast<const f32x4*>(cp + 4);
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(att_h + (long long)v * E + lane * 4);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + lane * 4);
    float dc[8], ah[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { dc[e] = d0[e]; dc[4 + e] = d1[e]; ah[e] = a4[e]; }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s = fmaf(x0[e], dc[e], fmaf(x1[e], dc[4 + e], s));
    s = wave_sum(s);
    float* de_row = de + (long long)v * Rp + r0;
    for (int it = 0; it + 2 < nblk2; it += 2) {
        attb_block<ATT_U>(xiA, xaA, alA, ABASE(it), n, dc, ah, s, acc, de_row, lane);
        __builtin_amdgcn_sched_barrier(0);
        attb_load<ATT_U>(xiA, xaA, alA, rim, rai, ral, ABASE(it + 2), n, lane);
        __builtin_amdgcn_sched_barrier(0);
        attb_block<ATT_U>(xiB, xaB, alB, ABASE(it + 1), n, dc, ah, s, acc, de_row, lane);
        __builtin_amdgcn_sched_barrier(0);
        attb_load<ATT_U>(xiB, xaB, alB, rim, rai, ral, ABASE(it + 3), n, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the last pair: nothing left to request
    attb_block<ATT_U>(xiA, xaA, alA, ABASE(nblk2 - 2), n, dc, ah, s, acc, de_row, lane);
    attb_block<ATT_U>(xiB, xaB, alB, ABASE(nblk2 - 1), n, dc, ah, s, acc, de_row, lane);
#undef ABASE
#pragma unroll
    for (int j = 0; j < 4; ++j) rede[wave][lane * 4 + j] = acc[j] * b4[j];
    __syncthreads();
    if (tid < E) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_W; ++w) t += rede[w][tid];
        atomicAdd(&datth[(long long)v * E + tid], t);
    }
}

// E_x = e^{2x} of the projected features (bf16 -> bf16), once per batch: what the E-domain attention kernels read instead of x.
// Exponent clamped to 2^+-60 so that E_x E_a stays finite (|x| > 20.8 is saturated tanh anyway).
__global__ __launch_bounds__(256) void att_exp_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float v[8];
        load8(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_exp2f(fminf(fmaxf(v[e] * 2.8853900817779268f, -60.f), 60.f));
        store8(out + i * 8, v);
    }
}

// deferred: d_att_img[b][r][k] = beta_k sum_t de[t][b][r] (1 - tau^2),  d_beta_k += sum de * tau
template <typename CT, int KCT>
__global__ __launch_bounds__(256) void datt_img_kernel(const CT* __restrict__ att_img, const float* __restrict__ att_h,
                                                      const float* __restrict__ beta, const float* __restrict__ de,
                                                      CT* __restrict__ dout, float* __restrict__ dbeta, float* __restrict__ dbeta_part,
                                                      int T, int B, int R, int Rp, int E) {
    __shared__ float redb[4][1024];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KC = (E + 255) >> 8;
    float bt[KCT][4], db[KCT][4];
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
        const int k0 = kc * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bt[kc][j] = (kc < KC && k0 + j < E) ? beta[k0 + j] : 0.f; db[kc][j] = 0.f; }
    }
    // a wave owns 4 regions and walks the T steps ONCE for all of them: the att_h row of a step is loaded once per
    // 4 regions (re-reading it per region made the kernel L2-bandwidth-bound: B*R*T KB = 5.6 GB per launch)
    const int rbase = blockIdx.x * 16 + wave * 4;
    float x[4][KCT][4], acc[4][KCT][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[rr][kc][j] = 0.f; acc[rr][kc][j] = 0.f; }
            if (rbase + rr < R && kc < KC && k0 < E) load4(att_img + ((long long)b * R + rbase + rr) * E + k0, x[rr][kc]);
            if constexpr (is_bf16<CT>::value) {
                // e^{2(x + a)} = e^{2x} e^{2a}: the factor of the region element is formed ONCE here, the factor of the step's att_h
                // once per step for all four regions -- per element and step that leaves one transcendental (the reciprocal) instead
                // of two.  Exponents clamped to +-120 (2^+-120: products never meet inf * 0; |x| > 41 is 1 - tanh^2 < 1e-35 anyway)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[rr][kc][j] = __builtin_amdgcn_exp2f(fminf(fmaxf(x[rr][kc][j] * 2.8853900817779268f, -120.f), 120.f));
            }
        }
    float sumd = 0.f;                                     // bf16 path: sum of d over the steps and the wave's regions
    if (rbase < R)
    for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
            if (kc < KC && k0 < E) {
                float a[4][4], d[4][4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {          // 4 time steps of loads in flight: unconditional (step clamped, d zeroed), or hipcc waits behind each
                    const int t = t0 + tt, tc = min(t, T - 1);
                    load4(att_h + ((long long)tc * B + b) * E + k0, a[tt]);
                    f32x4 dq = *reinterpret_cast<const f32x4*>(de + ((long long)tc * B + b) * Rp + rbase);      // rbase % 4 == 0, Rp % 8 == 0: in bounds
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) d[tt][rr] = (t < T && rbase + rr < R) ? dq[rr] : 0.f;
                    if constexpr (is_bf16<CT>::value) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[tt][j] = __builtin_amdgcn_exp2f(fminf(fmaxf(a[tt][j] * 2.8853900817779268f, -120.f), 120.f));
                    }
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    // d_e is wave-uniform and exactly 0 for every padded (sample, step) pair (the loss mask zeroes the whole
                    // gradient flow of those steps) -- all four regions belong to the same sample, so they vanish together
                    if (d[tt][0] == 0.f && d[tt][1] == 0.f && d[tt][2] == 0.f && d[tt][3] == 0.f) continue;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float dd = d[tt][rr];
                        if constexpr (is_bf16<CT>::value) {
                            // with r = 1 / (1 + e^{2s}):  tau = 1 - 2r,  1 - tau^2 = 4 r (1 - r).  The kernel is VALU-bound
                            // (1.4 G tanh per launch), so everything around the two transcendentals is pared down: x and
                            // att_h arrive as e^{2x} / e^{2a} (x once, att_h once per step for all four regions), the
                            // element pairs run on the packed-f32 ALU (v_pk_add / v_pk_fma), and
                            // d_beta = sum d - 2 sum d r keeps only the r-part per element
                            const f32x2 d4 = {4.f * dd, 4.f * dd}, d2 = {dd, dd}, one = {1.f, 1.f};
                            sumd += dd;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x2 xs = {x[rr][kc][2 * h], x[rr][kc][2 * h + 1]}, as = {a[tt][2 * h], a[tt][2 * h + 1]};
                                const f32x2 w = __builtin_elementwise_fma(xs, as, one);          // 1 + e^{2x} e^{2a}
                                const f32x2 r = {__builtin_amdgcn_rcpf(w[0]), __builtin_amdgcn_rcpf(w[1])};
                                const f32x2 q = __builtin_elementwise_fma(-r, r, r);
                                f32x2 ac = {acc[rr][kc][2 * h], acc[rr][kc][2 * h + 1]}, dv = {db[kc][2 * h], db[kc][2 * h + 1]};
                                ac = __builtin_elementwise_fma(d4, q, ac);
                                dv = __builtin_elementwise_fma(d2, r, dv);
                                acc[rr][kc][2 * h] = ac[0]; acc[rr][kc][2 * h + 1] = ac[1];
                                db[kc][2 * h] = dv[0]; db[kc][2 * h + 1] = dv[1];
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float tau = tanh_ct<CT>(x[rr][kc][j] + a[tt][j]);
                                acc[rr][kc][j] = fmaf(dd, 1.f - tau * tau, acc[rr][kc][j]);
                                db[kc][j] = fmaf(dd, tau, db[kc][j]);
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
            const int k0 = kc * 256 + lane * 4;
            if (rbase + rr < R && kc < KC && k0 < E) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[rr][kc][j] * bt[kc][j];
                store4(dout + ((long long)b * R + rbase + rr) * E + k0, o);
            }
        }
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
        const int k0 = kc * 256 + lane * 4;
        if (kc < KC && k0 < E) {
#pragma unroll
            for (int j = 0; j < 4; ++j) redb[wave][k0 + j] = is_bf16<CT>::value ? sumd - 2.f * db[kc][j] : db[kc][j];
        }
    }
    __syncthreads();
    if (dbeta_part) {       // f32 parity mode: this workgroup's slot of the scratch; lxo_k_det_reduce adds the slots in order
        float* slot = dbeta_part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * E;
        for (int k = tid; k < E; k += 256) slot[k] = redb[0][k] + redb[1][k] + redb[2][k] + redb[3][k];
        return;
    }
    for (int k = tid; k < E; k += 256) atomicAdd(&dbeta[k], redb[0][k] + redb[1][k] + redb[2][k] + redb[3][k]);
}

// d_img[b][r][c] += dmean[b][c] / R      (backward of the region mean)
__global__ __launch_bounds__(256) void add_mean_grad_kernel(float* __restrict__ dimg, const float* __restrict__ dmean, int B, int R, int C) {
    const long long total = (long long)B * R * C;
    const float inv = 1.0f / (float)R;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long long)R * C));
        dimg[i] += dmean[(long long)b * C + c] * inv;
    }
}

// loss of img2seq.py:68-75 + gradient; one wave per (t, b) row, rows strided over the grid; the two loss statistics
// are summed per workgroup first (one atomic pair per workgroup instead of one per token: the tokens all hit the same
// two addresses)
template <typename CT>
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, const int* __restrict__ formula,
                                                     const int* __restrict__ lengths, CT* __restrict__ dlogits,
                                                     float* __restrict__ loss_acc, float* __restrict__ loss_part, float inv_ntok, const float* __restrict__ ntok_dev,
                                                     const unsigned* __restrict__ chain_err, int B, int T, int V, int Vp) {
    __shared__ float red[8];
    // the persistent decoder chain (xdec.hip) flags a barrier that timed out: its logits are then garbage -- make the loss say so (NaN)
    // instead of training on them silently
    if (chain_err && blockIdx.x == 0 && threadIdx.x == 0 && chain_err[0] != 0u) atomicAdd(&loss_acc[0], __uint_as_float(0x7fc00000u));
    if (ntok_dev) inv_ntok = 1.0f / ntok_dev[0];      // data parallel: the global token count arrives by all-reduce, never through the host
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ce_sum = 0.f, n_sum = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < T * B; row += gridDim.x * 4) {
        const int t = row / B, b = row - t * B;
        const float* lg = logits + (long long)row * Vp;
        CT* dl = dlogits + (long long)row * Vp;
        const bool valid = t < lengths[b];
        int tgt = formula[(long long)b * T + t];
        tgt = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
        float m = -3.0e38f;
        for (int j = lane; j < V; j += 64) m = fmaxf(m, lg[j]);
        m = wave_max(m);
        float l = 0.f;
        for (int j = lane; j < V; j += 64) l += expf(lg[j] - m);
        l = wave_sum(l);
        const float lse = m + logf(l);
        const float scale = valid ? inv_ntok : 0.f;
        for (int j = lane; j < Vp; j += 64) {
            float g = 0.f;
            if (j < V) g = (expf(lg[j] - lse) - (j == tgt ? 1.f : 0.f)) * scale;
            dl[j] = from_f32<CT>(g);
        }
        if (valid) { ce_sum += lse - lg[tgt]; n_sum += 1.0f; }      // wave-uniform values
    }
    if (lane == 0) { red[wave] = ce_sum; red[4 + wave] = n_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float c = red[0] + red[1] + red[2] + red[3], n = red[4] + red[5] + red[6] + red[7];
        if (loss_part) { loss_part[2 * blockIdx.x] = c; loss_part[2 * blockIdx.x + 1] = n; }     // f32 parity mode: summed in workgroup order by lxo_k_det_reduce
        else if (n > 0.f) { atomicAdd(&loss_acc[0], c); atomicAdd(&loss_acc[1], n); }
    }
}

// The same with the row held in registers (Vp <= 64 * KV): ONE pass over the logits -- 16-byte loads, a lane owns 4 consecutive columns per
// quarter of KV -- instead of three passes of 4-byte loads, and every row has its own wave from the start (the three-pass kernel: 31 us for
// 13 MB at the benchmark shape, a chain of dependent passes per row; this one 8).  bf16 mode: v_exp_f32 (1 ulp); f32 parity mode: expf.
template <typename CT, int KV>
__global__ __launch_bounds__(256) void ce_loss_rows_kernel(const float* __restrict__ logits, const int* __restrict__ formula,
                                                          const int* __restrict__ lengths, CT* __restrict__ dlogits,
                                                          float* __restrict__ loss_acc, float* __restrict__ loss_part, float inv_ntok, const float* __restrict__ ntok_dev,
                                                          const unsigned* __restrict__ chain_err, int B, int T, int V, int Vp) {
    __shared__ float red[8];
    if (chain_err && blockIdx.x == 0 && threadIdx.x == 0 && chain_err[0] != 0u) atomicAdd(&loss_acc[0], __uint_as_float(0x7fc00000u));
    if (ntok_dev) inv_ntok = 1.0f / ntok_dev[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ce_sum = 0.f, n_sum = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < T * B; row += gridDim.x * 4) {
        const int t = row / B, b = row - t * B;
        const float* lg = logits + (long long)row * Vp;
        CT* dl = dlogits + (long long)row * Vp;
        const bool valid = t < lengths[b];
        int tgt = formula[(long long)b * T + t];
        tgt = tgt < 0 ? 0 : (tgt >= V ? V - 1 : tgt);
        const float xt = lg[tgt];
        float x[KV];
#pragma unroll
        for (int q = 0; q < KV / 4; ++q) {
            const int j0 = 4 * (lane + 64 * q);
            const f32x4 v = *reinterpret_cast<const f32x4*>(lg + (j0 < Vp ? j0 : 0));          // unconditional (clamped) load
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * q + e] = (j0 + e < V) ? v[e] : -3.0e38f;
        }
        float m = x[0];
#pragma unroll
        for (int e = 1; e < KV; ++e) m = fmaxf(m, x[e]);
        m = wave_max(m);
        float l = 0.f;
#pragma unroll
        for (int e = 0; e < KV; ++e) l += is_bf16<CT>::value ? __expf(x[e] - m) : expf(x[e] - m);
        l = wave_sum(l);
        const float lse = m + logf(l);
        const float scale = valid ? inv_ntok : 0.f;
#pragma unroll
        for (int q = 0; q < KV / 4; ++q) {
            const int j0 = 4 * (lane + 64 * q);
            if (j0 >= Vp) continue;
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                const float pr = is_bf16<CT>::value ? __expf(x[4 * q + e] - lse) : expf(x[4 * q + e] - lse);
                g[e] = j < V ? (pr - (j == tgt ? 1.f : 0.f)) * scale : 0.f;
            }
            if constexpr (is_bf16<CT>::value) { const u32x2 pk = {pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3])}; *reinterpret_cast<u32x2*>(dl + j0) = pk; }
            else { const f32x4 gv = {g[0], g[1], g[2], g[3]}; *reinterpret_cast<f32x4*>(dl + j0) = gv; }
        }
        if (valid) { ce_sum += lse - xt; n_sum += 1.0f; }
    }
    if (lane == 0) { red[wave] = ce_sum; red[4 + wave] = n_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float c = red[0] + red[1] + red[2] + red[3], n = red[4] + red[5] + red[6] + red[7];
        if (loss_part) { loss_part[2 * blockIdx.x] = c; loss_part[2 * blockIdx.x + 1] = n; }
        else if (n > 0.f) { atomicAdd(&loss_acc[0], c); atomicAdd(&loss_acc[1], n); }
    }
}

// out[n] += sum_m a[m][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int lda, float* __restrict__ out, int M, int N, int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    if (n >= N) return;
    float s = 0.f;
    for (int m = m0; m < m1; ++m) s += a[(long long)m * lda + n];
    atomicAdd(&out[n], s);
}

// the same for N % 4 == 0, lda % 4 == 0, 16-byte aligned a: a thread owns 4 columns and every fourth row of the block's range, eight 16-byte
// loads in flight (the scalar kernel above walked its 64 rows one load at a time: 16 us for a 64 x 512 matrix)
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ a, int lda, float* __restrict__ out, int M, int N, int rows_per_block) {
    __shared__ f32x4 red[4][64];
    const int cq = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + cq) * 4;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    const int nl = n < N ? n : 0;                              // clamped: loads stay unconditional
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int mb = m0 + ry; mb < m1; mb += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int m = min(mb + 4 * u, m1 - 1); v[u] = *reinterpret_cast<const f32x4*>(a + (long long)m * lda + nl); }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (mb + 4 * u < m1) s += v[u];
    }
    red[ry][cq] = s;
    __syncthreads();
    if (ry == 0 && n < N) {
        const f32x4 t = red[0][cq] + red[1][cq] + red[2][cq] + red[3][cq];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&out[n + e], t[e]);
    }
}


// ---- f32 parity mode: reductions without float atomics (the order of every sum is fixed, so two runs agree bit for bit) ----
// part[rb][n] = sum of rows [rb * rows_per, ...) of column n, rows in ascending order
template <typename AT>
__global__ __launch_bounds__(256) void colsum_part_kernel(const AT* __restrict__ a, long long lda, float* __restrict__ part, long long M, int N, int rows_per) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const long long m0 = (long long)blockIdx.y * rows_per, m1 = m0 + rows_per < M ? m0 + rows_per : M;
    if (n >= N) return;
    float s = 0.f;
    for (long long mb = m0; mb < m1; mb += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long long m = mb + u < m1 ? mb + u : m1 - 1; v[u] = to_f32(a[m * lda + n]); }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (mb + u < m1) s += v[u];
    }
    part[(long long)blockIdx.y * N + n] = s;
}
// out[n] += sum of part[q][n] over the slots q, in a FIXED order: the slots are cut into 16 contiguous groups, a thread adds one group
// left to right (eight loads in flight), thread 0 of a column adds the 16 group sums left to right.  The association is a function of
// (nslot) only, so two runs agree bit for bit -- and 2048 slots cost 16 batches of loads instead of a 2048-long chain of dependent
// load -> add round trips on 128 threads (round 5: that chain was 1.5 ms of the bf16 deterministic mode's step).
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ part, int nslot, long long stride, int N, float* __restrict__ out) {
    __shared__ float gs[16][16];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    const int per = (nslot + 15) / 16;
    const int q0 = g * per, q1 = min(nslot, q0 + per);
    float s = 0.f;
    if (n < N) {
        for (int q = q0; q < q1; q += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)min(q + u, q1 - 1) * stride + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (q + u < q1) s += v[u];
        }
    }
    gs[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float tsum = gs[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) tsum += gs[k][c];
        out[n] += tsum;
    }
}

// d_emb rows -> embedding_table / start_token gradients (decoder.py:90-93 backward)
__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ demb, const int* __restrict__ formula,
                                                           float* __restrict__ dtable, float* __restrict__ dstart,
                                                           int B, int T, int D, int V) {
    const long long total = (long long)T * B * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D);
        const long long row = i / D;
        const int b = (int)(row % B), t = (int)(row / B);
        const float g = demb[i];
        if (t == 0) atomicAdd(&dstart[d], g);
        else {
            int id = formula[(long long)b * T + t - 1];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            atomicAdd(&dtable[(long long)id * D + d], g);
        }
    }
}

// The same without global atomics and robust to skewed token frequencies: workgroup `id` (V of them, + one for the start token) scans
// the T*B token slots in chunks of 4096, compacts the slots that hold ITS token into an LDS list, then 256 / D rows of d_emb at a time
// are summed column-wise (coalesced row reads) and the total is added to the table row once.  The atomic kernel above spent 46 us on
// 0.5 M global atomics (uniform tokens; a frequent token serialises on its row); this one reads the token ids V times out of L2.
__global__ __launch_bounds__(256) void embed_scatter_rows_kernel(const float* __restrict__ demb, const int* __restrict__ formula,
                                                                float* __restrict__ dtable, float* __restrict__ dstart,
                                                                int B, int T, int D, int V, int det) {
    constexpr int CH = 2048;
    __shared__ int list[CH];
    __shared__ float part[256];
    __shared__ int cnt;
    __shared__ int wcnt[4];
    const int id = blockIdx.x;                                  // V = the start token (step 0 of every sample)
    const int tid = threadIdx.x;
    const int rpw = 256 / D, r = tid / D, c = tid - r * D;      // rows summed in parallel, this thread's (row lane, column)
    float sum = 0.f;
    const int total = T * B;
    // blockIdx.y: a sixteenth of the slots each (a frequent token -- the PAD id of a padded batch holds a third of all slots -- would
    // otherwise make one workgroup the whole kernel); the parts meet in D atomics per workgroup
    const int per = (total + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s0 = blockIdx.y * per, s1 = min(total, s0 + per);
    for (int base = s0; base < s1; base += CH) {
        if (tid == 0) cnt = 0;
        __syncthreads();
        if (det) {
            // f32 parity mode: the list is compacted in slot order (ballot + prefix counts), so the rows are summed in the same order every run
            for (int jb = base; jb < min(s1, base + CH); jb += 256) {      // block-uniform trip count
                const int j = jb + tid;
                int row = -1;
                if (j < min(s1, base + CH)) {
                    const int b = j / T, tt = j - b * T;
                    if (id == V) { if (j < B) row = j; }
                    else if (tt + 1 < T) { int tok = formula[j]; tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok); if (tok == id) row = (tt + 1) * B + b; }
                }
                const unsigned long long m = __ballot(row >= 0);
                const int lane = tid & 63, wv = tid >> 6;
                if (lane == 0) wcnt[wv] = __builtin_popcountll(m);
                __syncthreads();
                int off = cnt;
                for (int w = 0; w < wv; ++w) off += wcnt[w];
                if (row >= 0) list[off + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = row;
                __syncthreads();
                if (tid == 0) cnt += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
                __syncthreads();
            }
        } else
        for (int j = base + tid; j < min(s1, base + CH); j += 256) {
            // slot j of `formula` ([B][T], read in its own order: coalesced) = sample b, position tt: the input of step tt + 1 (the
            // last position feeds no step); the start token feeds step 0 of every sample: rows 0 .. B-1 of d_emb
            const int b = j / T, tt = j - b * T;
            if (id == V) { if (j < B) list[atomicAdd(&cnt, 1)] = j; continue; }
            if (tt + 1 >= T) continue;
            int tok = formula[j]; tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
            if (tok == id) list[atomicAdd(&cnt, 1)] = (tt + 1) * B + b;
        }
        __syncthreads();
        const int n = cnt;
        if (r < rpw) {
            for (int k = r; k < n; k += 4 * rpw) {              // four rows per thread in flight
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int kk = min(k + u * rpw, n - 1); v[u] = demb[(long long)list[kk] * D + c]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (k + u * rpw < n) sum += v[u];
            }
        }
        __syncthreads();
    }
    part[tid] = (r < rpw) ? sum : 0.f;
    __syncthreads();
    if (tid < D) {
        float t = 0.f;
        for (int q = 0; q < rpw; ++q) t += part[q * D + tid];
        float* dst = id == V ? dstart : dtable + (long long)id * D;
        if (t != 0.f) atomicAdd(&dst[tid], t);
    }
}

// dpre = d_s0 * (1 - s0^2) for s in (c, h, o) -> [B][U+U+O]   (attention_mechanism.py:151 backward)
__global__ __launch_bounds__(256) void init_bwd_kernel(const float* __restrict__ dcc, Slabs dxh,
                                                      const float* __restrict__ c0, const float* __restrict__ rec0, int ldr,
                                                      float* __restrict__ dpre, int B, int U, int O) {
    const int W = 2 * U + O;
    const int total = B * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int b = i / W, k = i - b * W;
        float d, s;
        if (k < U) { d = dcc[(long long)b * U + k]; s = c0[(long long)b * U + k]; }
        else if (k < 2 * U) { d = slab_sum(dxh, b, O + (k - U)); s = rec0[(long long)b * ldr + O + (k - U)]; }
        else { d = slab_sum(dxh, b, k - 2 * U); s = rec0[(long long)b * ldr + (k - 2 * U)]; }
        dpre[i] = d * (1.f - s * s);
    }
}

// ---- decode ----
// greedy_decoder_cell.py:58-64: id = argmax (first max), finished |= id == END; one wave per row
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int Vp, int V, int n, int id_end,
                                                    int* __restrict__ ids_step, int* __restrict__ ids_out, int max_steps, int step,
                                                    int* __restrict__ finished, int* __restrict__ n_unfinished) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* lg = logits + (long long)row * Vp;
    float best = -3.0e38f; int bi = 0x7fffffff;
    for (int j = lane; j < V; j += 64) { const float x = lg[j]; if (x > best) { best = x; bi = j; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        if (bi >= V) bi = 0;
        ids_step[row] = bi;
        ids_out[(long long)row * max_steps + step] = bi;
        const int f = finished[row] | (bi == id_end ? 1 : 0);
        finished[row] = f;
        if (!f) atomicAdd(n_unfinished, 1);
    }
}

// One block per image: beam_search_decoder_cell.py:146-187.
//  log_softmax, mask finished beams (0 at END, f32 lowest elsewhere), add running log-probs,
//  top-k over k*V (beam 0 only at time 0), ids = idx % V, parents = idx / V, gather finished.
// add_div_penalty (beam_search_decoder_cell.py:258-287, Li et al. 2016): score += log(div_gamma) * rank * bernoulli(div_prob),
// rank = position of the entry in the descending sort of its hypothesis' V scores (ties: lower id first, as
// tf.nn.top_k orders them).  Bernoulli draws: the counter hash of drop_scale on (time, image, beam, id).
struct DivPen { float log_gamma; unsigned thr; unsigned seed; float* scratch; };   // log_gamma == 0 or thr == 0: off

__global__ __launch_bounds__(256) void beam_step_kernel(float* __restrict__ logits, int Vp, int V, int k, int id_end, int time, DivPen dp,
                                                       float* __restrict__ logp, int* __restrict__ finished,
                                                       int* __restrict__ ids_step, int* __restrict__ parents_step,
                                                       int* __restrict__ ids_out, int* __restrict__ par_out, int max_steps,
                                                       int* __restrict__ n_unfinished) {
    __shared__ float lse[16];
    __shared__ float cand_v[16 * 4]; __shared__ int cand_i[16 * 4];
    __shared__ float sel_v[16]; __shared__ int sel_i[16];
    __shared__ int fin_old[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float FMIN = -3.40282347e38f;
    // log-sum-exp per beam (one wave per beam, round-robin)
    for (int j = wave; j < k; j += 4) {
        const float* lg = logits + ((long long)b * k + j) * Vp;
        float m = -3.0e38f;
        for (int c = lane; c < V; c += 64) m = fmaxf(m, lg[c]);
        m = wave_max(m);
        float l = 0.f;
        for (int c = lane; c < V; c += 64) l += expf(lg[c] - m);
        l = wave_sum(l);
        if (lane == 0) lse[j] = m + logf(l);
    }
    if (tid < k) fin_old[tid] = finished[b * k + tid];
    __syncthreads();
    const int nb = time > 0 ? k : 1;
    const int total = nb * V;
    const bool div = dp.log_gamma != 0.f && dp.thr != 0u;
    float* pen = dp.scratch + (long long)b * k * Vp;
    if (div) {
        float* row0 = logits + (long long)b * k * Vp;
        for (int i = tid; i < k * V; i += 256) {            // scores of every hypothesis, in place of its logits
            const int j = i / V, c = i - j * V;
            float sl = row0[j * Vp + c] - lse[j];
            const float f = fin_old[j] ? 1.f : 0.f;
            sl = (1.f - f) * sl + f * (c == id_end ? 0.f : FMIN);
            row0[j * Vp + c] = logp[b * k + j] + sl;
        }
        __syncthreads();
        for (int i = tid; i < k * V; i += 256) {
            const int j = i / V, c = i - j * V;
            const float* row = row0 + j * Vp;
            const float v = row[c];
            int rank = 0;
            for (int q = 0; q < V; ++q) { const float w = row[q]; rank += (w > v || (w == v && q < c)) ? 1 : 0; }
            const Drop dd = {dp.thr, 1.f, dp.seed, time, b * k + j, (int)gridDim.x * k};
            pen[j * Vp + c] = v + dp.log_gamma * (float)rank * drop_scale(dd, 3u, 0, c, V);
        }
        __syncthreads();
    }
    for (int sel = 0; sel < k; ++sel) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < total; i += 256) {
            const int j = i / V, c = i - j * V;
            bool taken = false;
            for (int q = 0; q < sel; ++q) taken |= (sel_i[q] == i);
            if (taken) continue;
            float val;
            if (div) val = pen[j * Vp + c];
            else {
                float sl = logits[((long long)b * k + j) * Vp + c] - lse[j];
                const float f = fin_old[j] ? 1.f : 0.f;
                sl = (1.f - f) * sl + f * (c == id_end ? 0.f : FMIN);
                val = logp[b * k + j] + sl;
            }
            if (val > best || (val == best && i < bi)) { best = val; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { cand_v[wave] = best; cand_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float bv = cand_v[0]; int bx = cand_i[0];
            for (int w = 1; w < 4; ++w)
                if (cand_v[w] > bv || (cand_v[w] == bv && cand_i[w] < bx)) { bv = cand_v[w]; bx = cand_i[w]; }
            sel_v[sel] = bv; sel_i[sel] = bx;
        }
        __syncthreads();
    }
    if (tid < k) {
        const int idx = sel_i[tid];
        const int id = idx % V, par = idx / V;
        const int f = fin_old[par] | (id == id_end ? 1 : 0);
        ids_step[b * k + tid] = id;
        parents_step[b * k + tid] = par;
        ids_out[((long long)b * max_steps + time) * k + tid] = id;
        if (par_out) par_out[((long long)b * max_steps + time) * k + tid] = par;
        logp[b * k + tid] = sel_v[tid];
        finished[b * k + tid] = f;
        if (!f) atomicAdd(n_unfinished, 1);
    }
}

// The same step with every candidate score held in REGISTERS (k * V <= BS_TH * BS_NPT = 4096, k <= 64 / BS_NW, no diversity penalty): beam_step_kernel re-reads and
// re-forms all k * V scores (an integer division each) for every one of its k selections and makes two passes over a row for its
// log-sum-exp -- 31 us of a 161 us beam-5 step at B = 64, on 64 workgroups.  Here a lane loads its share of a row ONCE (max, then the
// exponentials, from registers), a thread forms its <= BS_NPT scores ONCE, and a selection is a register scan + the block-wide arg-max.
// Same expressions, same summation order inside a wave, same tie rule (lower flat index first): the ids and parents are the slow kernel's.
constexpr int BS_TH = 512, BS_NW = BS_TH / 64, BS_NPT = 4096 / BS_TH;      // 8 waves x 8 candidates per thread (it was 4 x 16: a selection round scans a thread's candidates, k rounds per step)
// (value, index) arg-max over the 64 lanes of a wave -- value descending, index ascending -- every lane ends with the winner.  The four steps inside a
// row of 16 lanes are DPP moves, the two across rows permlane swaps (as bm_wave_sum_dpp: no LDS crossbar round trips).
#ifdef LXO_HIPSIM
LXO_DEV void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o); const int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
#else
template <int CTRL> LXO_DEV void amax_dpp(float& v, int& i) {
    const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xf, 0xf, true);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
LXO_DEV void wave_argmax(float& v, int& i) {
    amax_dpp<0xB1>(v, i); amax_dpp<0x4E>(v, i); amax_dpp<0x141>(v, i); amax_dpp<0x140>(v, i);
    {
        const auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const auto ri = __builtin_amdgcn_permlane16_swap((unsigned)i, (unsigned)i, false, false);
        const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]); const int i0 = (int)ri[0], i1 = (int)ri[1];
        const bool first = v0 > v1 || (v0 == v1 && i0 < i1);
        v = first ? v0 : v1; i = first ? i0 : i1;
    }
    {
        const auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const auto ri = __builtin_amdgcn_permlane32_swap((unsigned)i, (unsigned)i, false, false);
        const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]); const int i0 = (int)ri[0], i1 = (int)ri[1];
        const bool first = v0 > v1 || (v0 == v1 && i0 < i1);
        v = first ? v0 : v1; i = first ? i0 : i1;
    }
}
#endif
__global__ __launch_bounds__(BS_TH) void beam_step_fast_kernel(const float* __restrict__ logits, int Vp, int V, int k, int id_end, int time,
                                                            float* __restrict__ logp, int* __restrict__ finished,
                                                            int* __restrict__ ids_step, int* __restrict__ parents_step,
                                                            int* __restrict__ ids_out, int* __restrict__ par_out, int max_steps,
                                                            int* __restrict__ n_unfinished) {
    __shared__ float lse[16];
    __shared__ float sel_v[16]; __shared__ int sel_i[16];
    __shared__ int fin_old[16];
    __shared__ float lp_old[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float FMIN = -3.40282347e38f;
    __shared__ float wc_v[BS_NW * 16]; __shared__ int wc_i[BS_NW * 16];      // the waves' k best each
    const int nb = time > 0 ? k : 1;
    const int total = nb * V;
    float raw[BS_NPT];                                         // raw logits of this thread's candidates (unconditional, clamped: requested before anything is waited for)
    {
        int jq = tid / V, cq = tid - jq * V;
#pragma unroll
        for (int u = 0; u < BS_NPT; ++u) {
            const int j = min(jq, k - 1), c = cq;
            cq += BS_TH;
            while (cq >= V) { cq -= V; ++jq; }
            raw[u] = logits[((long long)b * k + j) * Vp + c];
        }
    }
    if (V <= 64 * 8 && k <= 8) {
        // log-sum-exp per hypothesis: a wave takes hypothesis `wave` (+ BS_NW q: every row requested before any is reduced), the row in registers
        // for its two passes
        constexpr int RQ = 8 / BS_NW > 0 ? 8 / BS_NW : 1;
        float x[RQ][8];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int j = min(wave + BS_NW * q, k - 1);
            const float* lg = logits + ((long long)b * k + j) * Vp;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int c = lane + 64 * u; x[q][u] = lg[c < V ? c : V - 1]; }
        }
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int j = wave + BS_NW * q;
            float m = -3.0e38f;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (lane + 64 * u < V) m = fmaxf(m, x[q][u]);
            m = wave_max(m);
            float l = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (lane + 64 * u < V) l += expf(x[q][u] - m);
            l = wave_sum(l);
            if (lane == 0 && j < k) lse[j] = m + logf(l);
        }
    } else
    for (int j = wave; j < k; j += BS_NW) {                   // the general form: one wave per hypothesis, round-robin
        const float* lg = logits + ((long long)b * k + j) * Vp;
        float m = -3.0e38f;
        if (V <= 64 * 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int c = lane + 64 * u; x[u] = lg[c < V ? c : V - 1]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (lane + 64 * u < V) m = fmaxf(m, x[u]);
            m = wave_max(m);
            float l = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) if (lane + 64 * u < V) l += expf(x[u] - m);
            l = wave_sum(l);
            if (lane == 0) lse[j] = m + logf(l);
        } else {
            for (int c = lane; c < V; c += 64) m = fmaxf(m, lg[c]);
            m = wave_max(m);
            float l = 0.f;
            for (int c = lane; c < V; c += 64) l += expf(lg[c] - m);
            l = wave_sum(l);
            if (lane == 0) lse[j] = m + logf(l);
        }
    }
    if (tid < k) { fin_old[tid] = finished[b * k + tid]; lp_old[tid] = logp[b * k + tid]; }
    __syncthreads();
    // candidates of this thread: i = tid + BS_TH u -> (hypothesis, id) walked instead of divided; their raw logits were requested in front of the
    // log-sum-exp pass (raw[]: the second read of the rows no longer waits behind the first)
    float val[BS_NPT];
    {
        int jq = tid / V, cq = tid - jq * V;
#pragma unroll
        for (int u = 0; u < BS_NPT; ++u) {
            const int i = tid + BS_TH * u;
            val[u] = -INFINITY;
            const int j = jq, c = cq;
            cq += BS_TH;
            while (cq >= V) { cq -= V; ++jq; }
            if (i < total) {
                float sl = raw[u] - lse[j];
                const float f = fin_old[j] ? 1.f : 0.f;
                sl = (1.f - f) * sl + f * (c == id_end ? 0.f : FMIN);
                val[u] = lp_old[j] + sl;
            }
        }
    }
    // top-k in two stages, ONE workgroup barrier between them (it was two per selection): every wave selects the k best of ITS candidates by itself
    // -- the k best of the block are among them -- then wave 0 selects the k best of the BS_NW k survivors, one per lane.  Order everywhere:
    // value descending, flat index ascending (tf.nn.top_k over the flattened [k * V] scores: the lower index of equal values first).
    {
        unsigned taken = 0u;                                   // bit u: this thread's candidate u has been selected
        for (int sel = 0; sel < k; ++sel) {
            float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < BS_NPT; ++u) {
                const int i = tid + BS_TH * u;
                if (i < total && !((taken >> u) & 1u) && (val[u] > best || (val[u] == best && i < bi))) { best = val[u]; bi = i; }
            }
            wave_argmax(best, bi);
            if (lane == 0) { wc_v[wave * 16 + sel] = best; wc_i[wave * 16 + sel] = bi; }
#pragma unroll
            for (int u = 0; u < BS_NPT; ++u) if (tid + BS_TH * u == bi) taken |= 1u << u;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int w = lane / k, e = lane - w * k;              // survivor e of wave w (BS_NW k <= 64 lanes: the launcher's condition)
        float cv = -INFINITY; int ci = 0x7fffffff;
        if (w < BS_NW) { cv = wc_v[w * 16 + e]; ci = wc_i[w * 16 + e]; }
        for (int sel = 0; sel < k; ++sel) {
            float bv = cv; int bx = ci;
            wave_argmax(bv, bx);
            if (lane == 0) { sel_v[sel] = bv; sel_i[sel] = bx; }
            if (ci == bx) { cv = -INFINITY; ci = 0x7fffffff; }
        }
    }
    __syncthreads();
    if (tid < k) {
        const int idx = sel_i[tid];
        const int id = idx % V, par = idx / V;
        const int f = fin_old[par] | (id == id_end ? 1 : 0);
        ids_step[b * k + tid] = id;
        parents_step[b * k + tid] = par;
        ids_out[((long long)b * max_steps + time) * k + tid] = id;
        if (par_out) par_out[((long long)b * max_steps + time) * k + tid] = par;
        logp[b * k + tid] = sel_v[tid];
        finished[b * k + tid] = f;
        if (!f) atomicAdd(n_unfinished, 1);
    }
}

// new[v] = old[b*k + parents[v]] for the carried state (o | h of rec, and c)   (beam_search_decoder_cell.py:176-178)
__global__ __launch_bounds__(256) void beam_gather_kernel(const float* __restrict__ rec, int ldr, int XH, const float* __restrict__ cs, int U,
                                                         const int* __restrict__ parents, int k, float* __restrict__ tmp_rec, float* __restrict__ tmp_cs, int n) {
    const int W = XH + U;
    const long long total = (long long)n * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / W), c = (int)(i - (long long)v * W);
        const int src = (v / k) * k + parents[v];
        if (c < XH) tmp_rec[(long long)v * XH + c] = rec[(long long)src * ldr + c];
        else tmp_cs[(long long)v * U + (c - XH)] = cs[(long long)src * U + (c - XH)];
    }
}
__global__ __launch_bounds__(256) void beam_scatter_kernel(float* __restrict__ rec, int ldr, int XH, float* __restrict__ cs, int U,
                                                          const float* __restrict__ tmp_rec, const float* __restrict__ tmp_cs, int n,
                                                          bf16_t* __restrict__ recb, int ldrb) {
    // recb (nullable): the bf16 mirror of the re-ordered [o | h] rows, written in the same pass (it was a launch of its own: mirror_kernel)
    const int W = XH + U;
    const long long total = (long long)n * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / W), c = (int)(i - (long long)v * W);
        if (c < XH) {
            const float x = tmp_rec[(long long)v * XH + c];
            rec[(long long)v * ldr + c] = x;
            if (recb) recb[(long long)v * ldrb + c] = f2bf(x);
        }
        else cs[(long long)v * U + (c - XH)] = tmp_cs[(long long)v * U + (c - XH)];
    }
}
// The same re-ordering in ONE launch and in place: the permutation moves whole ROWS inside an image's k rows, so column c of the state is
// permuted independently of every other column -- a thread takes one column of one image, reads its k values through `parents`, then
// writes them back in order (+ the bf16 mirror of [o | h]).  No scratch copy, no barrier; grid = images x ceil((XH + U) / 256) workgroups.
// (The gather / scatter pair above went through a scratch copy: two launches of ~5 us each in a 110 us beam step.)
constexpr int BP_KMAX = 16;
__global__ __launch_bounds__(256) void beam_permute_kernel(float* __restrict__ rec, int ldr, int XH, float* __restrict__ cs, int U,
                                                          const int* __restrict__ parents, int k, bf16_t* __restrict__ recb, int ldrb) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= XH + U) return;
    float v[BP_KMAX];
#pragma unroll
    for (int r = 0; r < BP_KMAX; ++r) {
        v[r] = 0.f;
        if (r < k) {
            const int src = b * k + parents[b * k + r];
            v[r] = c < XH ? rec[(long long)src * ldr + c] : cs[(long long)src * U + (c - XH)];
        }
    }
#pragma unroll
    for (int r = 0; r < BP_KMAX; ++r) {
        if (r < k) {
            const int dst = b * k + r;
            if (c < XH) { rec[(long long)dst * ldr + c] = v[r]; if (recb) recb[(long long)dst * ldrb + c] = f2bf(v[r]); }
            else cs[(long long)dst * U + (c - XH)] = v[r];
        }
    }
}
// row v of dst = row v / k of src (tile the initial state over the beam, beam_search_decoder_cell.py:98-109)
__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int n, int k, int cols) {
    const long long total = (long long)n * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / cols), c = (int)(i - (long long)v * cols);
        dst[(long long)v * ldd + c] = src[(long long)(v / k) * lds + c];
    }
}

// ---- optimizer ----
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s = fmaf(g[i], g[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];      // one partial per workgroup: summed in order below (no atomics: every run gives the same norm)
}
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, int nparts, float clip, float* __restrict__ out) {
    float ss = 0.f;
    for (int i = 0; i < nparts; ++i) ss += sumsq[i];
    const float gn = sqrtf(ss);
    out[1] = gn;
    out[0] = clip > 0.f ? clip / fmaxf(gn, clip) : 1.0f;
}
// lxo_chain_guard: fold the error words of the two persistent decoder chains (xdec.hip) into the optimizer's scale.  scale[0] = NaN when
// either chain of this step did not assemble (its outputs are garbage) or when the probe element of the gradients is NaN (another rank's
// chain failed: chain_poison_kernel below put a NaN into that element before the all-reduce), else the clip scale already there
// (have_scale) or 1.  status = {forward word, backward word, step dropped}.
__global__ void chain_guard_kernel(const unsigned* __restrict__ err_fwd, const unsigned* __restrict__ err_bwd, const float* __restrict__ probe,
                                   float* __restrict__ scale, int have_scale, unsigned* __restrict__ status) {
    const unsigned ef = err_fwd ? err_fwd[0] : 0u, eb = err_bwd ? err_bwd[0] : 0u;
    const float pv = probe ? probe[0] : 0.f;
    const bool drop = (ef | eb) != 0u || pv != pv;
    if (status) { status[0] = ef; status[1] = eb; status[2] = drop ? 1u : 0u; }
    scale[0] = drop ? __uint_as_float(0x7fc00000u) : (have_scale ? scale[0] : 1.0f);
}
// behind the decoder backward: a failed chain turns one gradient element into NaN, so that under data parallelism the gradient all-reduce
// carries the failure to every rank and all of them drop the step (lxo_chain_guard reads the element back)
__global__ void chain_poison_kernel(const unsigned* __restrict__ err_fwd, const unsigned* __restrict__ err_bwd, float* __restrict__ probe) {
    if ((err_fwd && err_fwd[0]) || (err_bwd && err_bwd[0])) probe[0] = __uint_as_float(0x7fc00000u);
}
// TF AdamOptimizer: theta -= lr_t * m / (sqrt(v) + eps), lr_t = lr sqrt(1-b2^t)/(1-b1^t) computed by the host
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long long n, float lr_t, float b1, float b2, float eps, const float* __restrict__ scale) {
    const float sc = scale ? scale[0] : 1.0f;
    if (sc != sc) return;                 // a NaN scale = a poisoned step (lxo_chain_guard: a decoder chain did not assemble): dropped, no slot is touched
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * sc;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// the other optimizers of img2seq.py:102-107 with TF-1.12 defaults:
//  mode 1 GradientDescent : theta -= lr g
//  mode 2 Adagrad         : acc (init 0.1) += g^2 ; theta -= lr g / sqrt(acc)
//  mode 3 RMSProp         : ms (init 1) = 0.9 ms + 0.1 g^2 ; theta -= lr g / sqrt(ms + 1e-10)
__global__ __launch_bounds__(256) void simple_opt_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ slot,
                                                        long long n, float lr, int mode, const float* __restrict__ scale) {
    const float sc = scale ? scale[0] : 1.0f;
    if (sc != sc) return;                 // as in adam_kernel: a poisoned step is dropped
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * sc;
        if (mode == 1) p[i] -= lr * gi;
        else if (mode == 2) { const float a = slot[i] + gi * gi; slot[i] = a; p[i] -= lr * gi / sqrtf(a); }
        else { const float ms = 0.9f * slot[i] + 0.1f * gi * gi; slot[i] = ms; p[i] -= lr * gi / sqrtf(ms + 1e-10f); }
    }
}

inline int grid1(long long items, int per_block = 256, int cap = 2048) {
    long long g = (items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
}  // namespace

#define LAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, __VA_ARGS__)
#define BYCT(dt, kern, grid, ...) do { if ((dt) == LXO_BF16) LAUNCH((kern<bf16_t>), grid, __VA_ARGS__); else LAUNCH((kern<float>), grid, __VA_ARGS__); } while (0)
#define DONE return (int)hipGetLastError()

int lxo_k_rowmean(int dt, const void* img, float* mean, int B, int R, int C, hipStream_t st) {
    if (C % 64) return -2;
    if (dt == LXO_BF16) LAUNCH((rowmean_kernel<bf16_t>), dim3(C / 64, B), (const bf16_t*)img, mean, R, C);
    else LAUNCH((rowmean_kernel<float>), dim3(C / 64, B), (const float*)img, mean, R, C);
    DONE;
}
int lxo_k_embed_gather(int dt, const float* table, const float* start, const int* formula, void* out, int B, int T, int D, int Dp, int V, hipStream_t st) {
    const int g = grid1((long long)T * B * Dp);
    if (dt == LXO_BF16) LAUNCH((embed_gather_kernel<bf16_t>), g, table, start, formula, (bf16_t*)out, B, T, D, Dp, V);
    else LAUNCH((embed_gather_kernel<float>), g, table, start, formula, (float*)out, B, T, D, Dp, V);
    DONE;
}
int lxo_k_embed_rows(int dt, const float* table, const float* start, const int* ids, void* out, int n, int D, int Dp, int V, hipStream_t st) {
    const int g = grid1((long long)n * Dp);
    if (dt == LXO_BF16) LAUNCH((embed_rows_kernel<bf16_t>), g, table, start, ids, (bf16_t*)out, n, D, Dp, V);
    else LAUNCH((embed_rows_kernel<float>), g, table, start, ids, (float*)out, n, D, Dp, V);
    DONE;
}
int lxo_k_embed_table(int dt, const float* table, const float* start, void* out, int V, int D, int Dp, hipStream_t st) {
    const int g = grid1((long long)(V + 1) * Dp);
    if (dt == LXO_BF16) LAUNCH((embed_table_kernel<bf16_t>), g, table, start, (bf16_t*)out, V, D, Dp);
    else LAUNCH((embed_table_kernel<float>), g, table, start, (float*)out, V, D, Dp);
    DONE;
}
int lxo_k_lstm_fwd(const float* z, Slabs zs, const float* c_prev, float* gates, float* c_out, float* h_out, float* ht_out, int ldh, Drop dr, int B, int U, hipStream_t st) {
    LAUNCH(lstm_fwd_kernel, grid1((long long)B * U), z, zs, c_prev, gates, c_out, h_out, ht_out, ldh, dr, B, U);
    DONE;
}
int lxo_k_lstm_bwd(const float* gates, const float* c_prev, const float* c_cur, Slabs s1, Slabs s3, Slabs s4, int off4,
                   float* dcc, float* dz, Drop dr, int carry_rows, int B, int U, hipStream_t st) {
    LAUNCH(lstm_bwd_kernel, grid1((long long)B * U), gates, c_prev, c_cur, s1, s3, s4, off4, dcc, dz, dr, carry_rows, B, U);
    DONE;
}
int lxo_k_tanh_bwd(const float* a, int lda, Slabs carry, const float* o, int ldo, float* g, int ldg, void* gb, int ldgb, Drop dr, int carry_rows, int rows, int cols, hipStream_t st) {
    LAUNCH(tanh_bwd_kernel, grid1((long long)rows * cols / 4), a, lda, carry, o, ldo, g, ldg, (bf16_t*)gb, ldgb, dr, carry_rows, rows, cols);
    DONE;
}
__global__ __launch_bounds__(256) static void slab_reduce_kernel(Slabs sl, float* __restrict__ o, int ldo, int rows, int cols) {
    const int total = rows * (cols >> 2);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / (cols >> 2), c = (i - r * (cols >> 2)) << 2;
        *reinterpret_cast<f32x4*>(o + (long long)r * ldo + c) = slab_sum4(sl, r, c);
    }
}
int lxo_k_slab_reduce(Slabs sl, float* o, int ldo, int rows, int cols, hipStream_t st) {
    LAUNCH(slab_reduce_kernel, grid1((long long)rows * cols / 4), sl, o, ldo, rows, cols);
    DONE;
}
int lxo_k_tanh_finalize(Slabs sl, float* o, int ldo, Drop dr, int rows, int cols, hipStream_t st) {
    LAUNCH(tanh_finalize_kernel, grid1((long long)rows * cols / 4), sl, o, ldo, dr, rows, cols);
    DONE;
}
static int att_u(int rows_per) {
    static int forced = -1;
    // rows of both streams a wave keeps in flight per iteration (8 waves per workgroup): 8, or 7 when 56 rows per iteration need no more
    // iterations than 64 would -- the benchmark's chunk of 109 rows is 2 x 56 instead of 64 + 45 (forward pair 23.3 -> 22.3 us per step).
    // LXO_ATT_U=7 / 8 forces one instantiation.
    if (forced < 0) { const char* e = getenv("LXO_ATT_U"); forced = e ? atoi(e) : 0; }
    if (forced == 7 || forced == 8) return forced;
    return (rows_per + 55) / 56 <= (rows_per + 63) / 64 ? 7 : 8;
}
int lxo_k_attn_fwd(int dt, const void* att_img, const void* img, const float* att_h, Slabs ahs, float* att_h_out, const float* beta, float* alpha, float* part,
                   float* ctx, int ldctx, void* ctxb, int ldcb, int nv, int R, int Rp, int E, int C, int beam, int nch, int rev, hipStream_t st, const void* att_exp) {
    if (E > 1024 || C > 512 || nch < 1 || nch > 32) return -2;
    const int rows_per = cdiv(R, nch);
    dim3 grid(nch, nv);
    static int split = -1;      // LXO_ATT_SPLIT=1: scores + softmax-context, no merge launch (A/B, measured slower); default: part + combine
    if (split < 0) { const char* e = getenv("LXO_ATT_SPLIT"); split = (e && e[0] == '1') ? 1 : 0; }
    const int epl = dt == LXO_BF16 ? 8 : 4;
    if (split && ahs.n == 0 && att_h && R <= 1024 && C % 64 == 0 && E % epl == 0 && E <= 64 * epl && ((E / epl) & (E / epl - 1)) == 0) {
        // raw scores travel through the `part` scratch (>= nv * Rp floats); alpha receives the normalised weights
        if (dt == LXO_BF16) {
            hipLaunchKernelGGL((attn_scores_kernel<bf16_t, 8>), grid, dim3(512), 0, st, (const bf16_t*)att_img, att_h, beta, part, R, Rp, E, beam, rows_per);
            hipLaunchKernelGGL((attn_ctx_kernel<bf16_t, 8>), dim3(C / 64, nv), dim3(512), 0, st, (const bf16_t*)img, part, alpha, ctx, ldctx, (bf16_t*)ctxb, ldcb, R, Rp, C, beam);
        } else {
            hipLaunchKernelGGL((attn_scores_kernel<float, 8>), grid, dim3(512), 0, st, (const float*)att_img, att_h, beta, part, R, Rp, E, beam, rows_per);
            hipLaunchKernelGGL((attn_ctx_kernel<float, 8>), dim3(C / 64, nv), dim3(512), 0, st, (const float*)img, part, alpha, ctx, ldctx, (bf16_t*)ctxb, ldcb, R, Rp, C, beam);
        }
        DONE;
    }
    // beam search: one pass over an image's rows for all its hypotheses (attn_fwd_part_beam_kernel); the chunk count is chosen per IMAGE
    static int beam_shared = -1;                          // LXO_ATT_BEAM_SHARED=0: the per-row kernel for beam search too (A/B)
    if (beam_shared < 0) { const char* e = getenv("LXO_ATT_BEAM_SHARED"); beam_shared = (e && e[0] == '0') ? 0 : 1; }
    // (beam == 1 with att_exp: greedy decode in bf16 -- the same kernel with one hypothesis per image, for its E-domain arithmetic)
    if (beam_shared && (beam == 2 || beam == 3 || beam == 5 || (beam == 1 && att_exp && dt == LXO_BF16)) && ahs.n == 0 && att_h && E <= 256 && C <= 512 && nv % beam == 0) {
        const int nimg = nv / beam;
        int nb = cdiv(512, nimg);
        if (nb > 16) nb = 16;
        const int by_rows = R / 32 > 0 ? R / 32 : 1;
        if (nb > by_rows) nb = by_rows;
        const int need = cdiv(R, 1024);
        if (nb < need) nb = need;
        const int rpb = cdiv(R, nb);
        const dim3 gb(nb, nimg);
#define ABM(CT_, U_, NB_, X_, OCC_, AI_) hipLaunchKernelGGL((attn_fwd_part_beam_kernel<CT_, U_, NB_, X_, OCC_>), gb, dim3(512), 0, st, (const CT_*)(AI_), (const CT_*)img, att_h, beta, alpha, part, R, Rp, E, C, nb, rpb, rev)
        if (dt == LXO_BF16 && att_exp) { if (beam == 1) ABM(bf16_t, 4, 1, true, 4, att_exp); else if (beam == 2) ABM(bf16_t, 4, 2, true, 4, att_exp); else if (beam == 3) ABM(bf16_t, 4, 3, true, 4, att_exp); else ABM(bf16_t, 2, 5, true, 4, att_exp); }
        else if (dt == LXO_BF16) { if (beam == 2) ABM(bf16_t, 4, 2, false, 2, att_img); else if (beam == 3) ABM(bf16_t, 4, 3, false, 2, att_img); else ABM(bf16_t, 4, 5, false, 2, att_img); }
        else { if (beam == 2) ABM(float, 4, 2, false, 2, att_img); else if (beam == 3) ABM(float, 4, 3, false, 2, att_img); else ABM(float, 4, 5, false, 2, att_img); }
#undef ABM
        hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3(4, nv), dim3(256), 0, st, part, alpha, ctx, ldctx, (bf16_t*)ctxb, ldcb, R, Rp, C, nb, rpb);
        DONE;
    }
#define AF_ARGS att_h, ahs, att_h_out, beta, alpha, part, R, Rp, E, C, beam, nch, rows_per, rev
    if (dt == LXO_BF16) {
        if (E <= 256) { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_fwd_part_kernel<bf16_t, 1, 8>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AF_ARGS); else hipLaunchKernelGGL((attn_fwd_part_kernel<bf16_t, 1, 7>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AF_ARGS); }
        else { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_fwd_part_kernel<bf16_t, 4, 8>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AF_ARGS); else hipLaunchKernelGGL((attn_fwd_part_kernel<bf16_t, 4, 7>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AF_ARGS); }
    } else {
        if (E <= 256) { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_fwd_part_kernel<float, 1, 8>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AF_ARGS); else hipLaunchKernelGGL((attn_fwd_part_kernel<float, 1, 7>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AF_ARGS); }
        else { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_fwd_part_kernel<float, 4, 8>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AF_ARGS); else hipLaunchKernelGGL((attn_fwd_part_kernel<float, 4, 7>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AF_ARGS); }
    }
#undef AF_ARGS
    hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3(4, nv), dim3(256), 0, st, part, alpha, ctx, ldctx, (bf16_t*)ctxb, ldcb, R, Rp, C, nch, rows_per);
    DONE;
}
// datth must be zero on entry (chunks accumulate with atomics; nch == 1 -- the f32 parity mode -- has one writer per element)
int lxo_k_att_exp(const void* att_img, void* att_exp, long long n, hipStream_t st) {
    if (n % 8) return -2;
    LAUNCH(att_exp_kernel, grid1(n / 8, 256, 4096), (const bf16_t*)att_img, (bf16_t*)att_exp, n / 8);
    DONE;
}
// att_exp (nullable, bf16 mode only): e^{2 att_img}; when given the kernel works in the E domain (one transcendental per element)
int lxo_k_attn_bwd(int dt, const void* att_img, const void* att_exp, const void* img, const float* att_h, const float* beta, const float* alpha,
                   Slabs dcs, int dcoff, float* dctx_out, int lddc, const float* ctx, int ldctx, float* de, float* datth,
                   int nv, int R, int Rp, int E, int C, int nch, int rev, hipStream_t st) {
    if (E > 1024 || C > 512 || nch < 1 || nch > 32) return -2;
    const int rows_per = cdiv(R, nch);
    dim3 grid(nch, nv);
    // LXO_ATT_BWD2=1: the variant with two row blocks in flight.  Measured SLOWER (decoder backward 3.42 vs 3.33 ms): the kernel is bound by
    // its vector arithmetic (~380 issue cycles per row and wave, 10 us per launch), not by exposed latency -- with two workgroups per CU the
    // other waves already cover the loads -- and 32-row blocks pad a 109-row chunk to 128.  Kept for A/B runs.
    static int stream2 = -1;
    if (stream2 < 0) { const char* e = getenv("LXO_ATT_BWD2"); stream2 = (e && e[0] == '1') ? 1 : 0; }
    if (stream2 && dt == LXO_BF16 && E == 256 && C == 512 && dcs.n == 1 && lddc % 4 == 0 && ldctx % 4 == 0 && (long long)rows_per * 1024 < (1LL << 31)) {
        const float* dctx = dcs.p + dcoff;
        if (dctx_out) return -2;                           // (the fused path never asks for the summed copy)
        hipLaunchKernelGGL((attn_bwd_stream_kernel<4>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, att_h, beta, alpha, dctx, dcs.ld,
                           ctx, ldctx, de, datth, R, Rp, rows_per, rev);
        DONE;
    }
#define AB_ARGS att_h, beta, alpha, dcs, dcoff, dctx_out, lddc, ctx, ldctx, de, datth, R, Rp, E, C, rows_per, rev
    if (dt == LXO_BF16 && att_exp && E <= 256) {
        if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_bwd_part_kernel<bf16_t, 1, 8, true>), grid, dim3(512), 0, st, (const bf16_t*)att_exp, (const bf16_t*)img, AB_ARGS);
        else hipLaunchKernelGGL((attn_bwd_part_kernel<bf16_t, 1, 7, true>), grid, dim3(512), 0, st, (const bf16_t*)att_exp, (const bf16_t*)img, AB_ARGS);
    } else if (dt == LXO_BF16) {
        if (E <= 256) { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_bwd_part_kernel<bf16_t, 1, 8>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AB_ARGS); else hipLaunchKernelGGL((attn_bwd_part_kernel<bf16_t, 1, 7>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AB_ARGS); }
        else { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_bwd_part_kernel<bf16_t, 4, 8>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AB_ARGS); else hipLaunchKernelGGL((attn_bwd_part_kernel<bf16_t, 4, 7>), grid, dim3(512), 0, st, (const bf16_t*)att_img, (const bf16_t*)img, AB_ARGS); }
    } else {
        if (E <= 256) { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_bwd_part_kernel<float, 1, 8>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AB_ARGS); else hipLaunchKernelGGL((attn_bwd_part_kernel<float, 1, 7>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AB_ARGS); }
        else { if (att_u(rows_per) == 8) hipLaunchKernelGGL((attn_bwd_part_kernel<float, 4, 8>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AB_ARGS); else hipLaunchKernelGGL((attn_bwd_part_kernel<float, 4, 7>), grid, dim3(512), 0, st, (const float*)att_img, (const float*)img, AB_ARGS); }
    }
#undef AB_ARGS
    DONE;
}
int lxo_k_det_reduce(const float* part, int nslot, long long stride, int N, float* out, hipStream_t st) {
    if (nslot <= 0 || N <= 0) return 0;
    hipLaunchKernelGGL(det_reduce_kernel, dim3(cdiv(N, 16)), dim3(256), 0, st, part, nslot, stride, N, out);      // 16 columns x 16 slot groups per workgroup
    DONE;
}
int lxo_k_datt_img(int dt, const void* att_img, const float* att_h, const float* beta, const float* de, void* dout, float* dbeta,
                   int T, int B, int R, int Rp, int E, DetScratch det, hipStream_t st) {
    dim3 grid(cdiv(R, 16), B);
    if (E > 1024) return -2;
    float* part = nullptr;
    if (det.p) { if ((size_t)grid.x * grid.y * E > det.floats) return -6; part = det.p; }
    if (dt == LXO_BF16) {
        if (E <= 256) hipLaunchKernelGGL((datt_img_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)att_img, att_h, beta, de, (bf16_t*)dout, dbeta, part, T, B, R, Rp, E);
        else hipLaunchKernelGGL((datt_img_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)att_img, att_h, beta, de, (bf16_t*)dout, dbeta, part, T, B, R, Rp, E);
    } else {
        if (E <= 256) hipLaunchKernelGGL((datt_img_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)att_img, att_h, beta, de, (float*)dout, dbeta, part, T, B, R, Rp, E);
        else hipLaunchKernelGGL((datt_img_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)att_img, att_h, beta, de, (float*)dout, dbeta, part, T, B, R, Rp, E);
    }
    if (part) return lxo_k_det_reduce(part, (int)(grid.x * grid.y), E, E, dbeta, st);
    DONE;
}
int lxo_k_add_mean_grad(float* dimg, const float* dmean, int B, int R, int C, hipStream_t st) {
    LAUNCH(add_mean_grad_kernel, grid1((long long)B * R * C), dimg, dmean, B, R, C);
    DONE;
}
int lxo_k_ce_loss(int dt, const float* logits, const int* formula, const int* lengths, void* dlogits, float* loss_acc, float inv_ntok,
                  const float* ntok_dev, const unsigned* chain_err, int B, int T, int V, int Vp, DetScratch det, hipStream_t st) {
    int g = cdiv(T * B, 4);
    float* part = (det.p && det.floats >= 1024) ? det.p : nullptr;      // loss_acc is zero on entry (lxo_impl_ce_loss)
    if (Vp % 4 == 0 && Vp <= 1024 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0) {
        // a wave per row, the row in registers (one pass); the f32 parity mode keeps at most 512 workgroups (its ordered partial sums)
        if (g > (part ? 512 : 2048)) g = part ? 512 : 2048;
#define CE_ROWS(CT_, KV_) LAUNCH((ce_loss_rows_kernel<CT_, KV_>), g, logits, formula, lengths, (CT_*)dlogits, loss_acc, part, inv_ntok, ntok_dev, chain_err, B, T, V, Vp)
        if (dt == LXO_BF16) { if (Vp <= 256) CE_ROWS(bf16_t, 4); else if (Vp <= 512) CE_ROWS(bf16_t, 8); else CE_ROWS(bf16_t, 16); }
        else { if (Vp <= 256) CE_ROWS(float, 4); else if (Vp <= 512) CE_ROWS(float, 8); else CE_ROWS(float, 16); }
#undef CE_ROWS
        if (part) return lxo_k_det_reduce(part, g, 2, 2, loss_acc, st);
        DONE;
    }
    if (g > 512) g = 512;
    if (dt == LXO_BF16) LAUNCH((ce_loss_kernel<bf16_t>), g, logits, formula, lengths, (bf16_t*)dlogits, loss_acc, part, inv_ntok, ntok_dev, chain_err, B, T, V, Vp);
    else LAUNCH((ce_loss_kernel<float>), g, logits, formula, lengths, (float*)dlogits, loss_acc, part, inv_ntok, ntok_dev, chain_err, B, T, V, Vp);
    if (part) return lxo_k_det_reduce(part, g, 2, 2, loss_acc, st);
    DONE;
}
// ordered column sums (no atomics): per-row-block partial sums into the scratch, then the blocks in order; a = f32 or (bf16 != 0) bf16
int lxo_k_colsum_det(const void* a, int bf16, long long lda, float* out, long long M, int N, DetScratch det, hipStream_t st) {
    if (!det.p) return -6;
    if (M <= 0 || N <= 0) return 0;
    long long maxslots = (long long)(det.floats / (size_t)N);
    if (maxslots < 1) return -6;
    if (maxslots > 1024) maxslots = 1024;
    long long rows_per = (M + maxslots - 1) / maxslots;
    if (rows_per < 64) rows_per = 64;
    const int nrb = (int)((M + rows_per - 1) / rows_per);
    if (bf16) hipLaunchKernelGGL((colsum_part_kernel<bf16_t>), dim3(cdiv(N, 256), nrb), dim3(256), 0, st, (const bf16_t*)a, lda, det.p, M, N, (int)rows_per);
    else hipLaunchKernelGGL((colsum_part_kernel<float>), dim3(cdiv(N, 256), nrb), dim3(256), 0, st, (const float*)a, lda, det.p, M, N, (int)rows_per);
    return lxo_k_det_reduce(det.p, nrb, N, N, out, st);
}
int lxo_k_colsum(const float* a, long long lda, float* out, long long M, int N, DetScratch det, hipStream_t st) {
    if (det.p) return lxo_k_colsum_det(a, 0, lda, out, M, N, det, st);      // parity / deterministic mode: two launches, no atomics
    const int rpb = 64;
    if (N % 4 == 0 && lda % 4 == 0 && ((uintptr_t)a & 15) == 0 && M > 0)
        hipLaunchKernelGGL(colsum4_kernel, dim3(cdiv(N, 256), cdiv((int)M, rpb)), dim3(256), 0, st, a, (int)lda, out, (int)M, N, rpb);
    else
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 256), cdiv((int)M, rpb)), dim3(256), 0, st, a, (int)lda, out, (int)M, N, rpb);
    DONE;
}
// det != 0 (f32 parity mode): one workgroup per token id, its rows listed and summed in slot order (no atomics meet in a table row)
int lxo_k_embed_scatter(const float* demb, const int* formula, float* dtable, float* dstart, int B, int T, int D, int V, int det, hipStream_t st) {
    if (D <= 256) { hipLaunchKernelGGL(embed_scatter_rows_kernel, dim3(V + 1, det ? 1 : 16), dim3(256), 0, st, demb, formula, dtable, dstart, B, T, D, V, det); DONE; }
    if (det) return -2;
    LAUNCH(embed_scatter_kernel, grid1((long long)T * B * D), demb, formula, dtable, dstart, B, T, D, V);
    DONE;
}
int lxo_k_init_bwd(const float* dcc, Slabs dxh, const float* c0, const float* rec0, int ldr, float* dpre, int B, int U, int O, hipStream_t st) {
    LAUNCH(init_bwd_kernel, grid1((long long)B * (2 * U + O)), dcc, dxh, c0, rec0, ldr, dpre, B, U, O);
    DONE;
}
int lxo_k_argmax(const float* logits, int Vp, int V, int n, int id_end, int* ids_step, int* ids_out, int max_steps, int step,
                 int* finished, int* n_unfinished, hipStream_t st) {
    LAUNCH(argmax_kernel, cdiv(n, 4), logits, Vp, V, n, id_end, ids_step, ids_out, max_steps, step, finished, n_unfinished);
    DONE;
}
int lxo_k_beam_step(float* logits, int Vp, int V, int nimg, int k, int id_end, int time, float div_gamma, float div_prob, int div_seed,
                    float* scratch, float* logp, int* finished,
                    int* ids_step, int* parents_step, int* ids_out, int* par_out, int max_steps, int* n_unfinished, hipStream_t st) {
    if (k > 16) return -2;
    DivPen dp = {0.f, 0u, (unsigned)div_seed, scratch};
    if (div_gamma > 0.f && div_gamma != 1.f && div_prob > 0.f) {      // the reference returns early for gamma == 1 or prob == 0
        dp.log_gamma = logf(div_gamma);
        dp.thr = div_prob >= 1.f ? 16777216u : (unsigned)(div_prob * 16777216.0f);
    }
    static int fast = -1;                                      // LXO_BEAM_FAST=0: the general kernel always (A/B)
    if (fast < 0) { const char* e = getenv("LXO_BEAM_FAST"); fast = (e && e[0] == '0') ? 0 : 1; }
    if (fast && dp.log_gamma == 0.f && (long long)k * V <= BS_TH * BS_NPT && k * BS_NW <= 64)
        hipLaunchKernelGGL(beam_step_fast_kernel, dim3(nimg), dim3(BS_TH), 0, st, logits, Vp, V, k, id_end, time, logp, finished, ids_step, parents_step, ids_out, par_out, max_steps, n_unfinished);
    else
    LAUNCH(beam_step_kernel, nimg, logits, Vp, V, k, id_end, time, dp, logp, finished, ids_step, parents_step, ids_out, par_out, max_steps, n_unfinished);
    DONE;
}
int lxo_k_beam_gather(float* rec, int ldr, int XH, float* cs, int U, const int* parents, int k, float* tmp_rec, float* tmp_cs, int n, void* recb, int ldrb, hipStream_t st) {
    if (k >= 1 && k <= BP_KMAX && n % k == 0) {
        LAUNCH(beam_permute_kernel, dim3(n / k, cdiv(XH + U, 256)), rec, ldr, XH, cs, U, parents, k, (bf16_t*)recb, ldrb);
        DONE;
    }
    LAUNCH(beam_gather_kernel, grid1((long long)n * (XH + U)), rec, ldr, XH, cs, U, parents, k, tmp_rec, tmp_cs, n);
    LAUNCH(beam_scatter_kernel, grid1((long long)n * (XH + U)), rec, ldr, XH, cs, U, tmp_rec, tmp_cs, n, (bf16_t*)recb, ldrb);
    DONE;
}
int lxo_k_tile_rows(const float* src, int lds, float* dst, int ldd, int n, int k, int cols, hipStream_t st) {
    LAUNCH(tile_rows_kernel, grid1((long long)n * cols), src, lds, dst, ldd, n, k, cols);
    DONE;
}
// sumsq_tmp: 1024 floats (one partial per workgroup, summed in order: the norm is the same in every run)
int lxo_k_global_norm_scale(long long n, const float* g, float clip, float* sumsq_tmp, float* out, hipStream_t st) {
    const int nparts = grid1(n, 256 * 8, 1024);
    LAUNCH(sumsq_kernel, nparts, g, n, sumsq_tmp);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, st, sumsq_tmp, nparts, clip, out);
    DONE;
}
int lxo_k_simple_opt(float* p, const float* g, float* slot, long long n, float lr, int mode, const float* scale, hipStream_t st) {
    if (mode < 1 || mode > 3 || (mode > 1 && !slot)) return -2;
    LAUNCH(simple_opt_kernel, grid1(n, 256 * 4, 4096), p, g, slot, n, lr, mode, scale);
    DONE;
}
int lxo_k_chain_guard(const unsigned* err_fwd, const unsigned* err_bwd, const float* probe, float* scale, int have_scale, unsigned* status, hipStream_t st) {
    hipLaunchKernelGGL(chain_guard_kernel, dim3(1), dim3(1), 0, st, err_fwd, err_bwd, probe, scale, have_scale, status);
    DONE;
}
int lxo_k_chain_poison(const unsigned* err_fwd, const unsigned* err_bwd, float* probe, hipStream_t st) {
    hipLaunchKernelGGL(chain_poison_kernel, dim3(1), dim3(1), 0, st, err_fwd, err_bwd, probe);
    DONE;
}
int lxo_k_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2, float eps, const float* scale, hipStream_t st) {
    LAUNCH(adam_kernel, grid1(n, 256 * 4, 4096), p, g, m, v, n, lr_t, b1, b2, eps, scale);
    DONE;
}
