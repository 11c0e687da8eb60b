// MFMA GEMM families for gfx950.
//
//  gemm_nt_kernel : C = epi(A * Bp^T), A dense or implicit-im2col (3x3 conv fwd
//                   and dgrad), 4 waves, each wave a grid of 32x32 MFMA tiles,
//                   global -> registers -> LDS staging with the next K-tile's
//                   loads in flight under the current tile's MFMAs.
//  gemm_tn_kernel : C += A^T * B reduced over rows (conv wgrad, every dense
//                   weight gradient, the deferred d_img outer-product sum).
//                   The reduction index is the strided one in memory, so rows
//                   are paired and written to LDS as packed bf16x2 dwords, which
//                   makes the MFMA operand reads contiguous without a per-element
//                   scatter.
//
// bf16 mode uses v_mfma_f32_32x32x16_bf16 (f32 accumulate); f32 mode (parity
// mode) uses v_mfma_f32_32x32x2_f32, which is an exact fmaf chain.
#include "gemm.h"
#include "decoder_kernels.h"      // lxo_k_colsum_det (deterministic column sums)
#include <stdlib.h>

namespace {

template <typename CT> struct StageReg;
template <> struct StageReg<bf16_t> { u32x4 v; };
template <> struct StageReg<float> { f32x4 lo, hi; };

// The load itself is UNCONDITIONAL (callers pass a valid address -- offset 0 of the operand -- for rows outside the
// matrix) and the value is zeroed afterwards: `ok ? load : 0` made hipcc branch around every load and wait vmcnt(0) behind
// it (260 such waits in one gemm_nt instantiation), which serialised the staging loads of a K-step.
template <typename CT, typename TA>
LXO_DEV StageReg<CT> stage_load(const TA* p, bool ok) {
    StageReg<CT> r;
    if constexpr (is_bf16<CT>::value) {
        if constexpr (is_bf16<TA>::value) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            const u32x4 v = *reinterpret_cast<const u32x4*>(p);
            r.v = ok ? v : z;
        } else {
            float v[8];
            load8(p, v);
            u32x4 t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
            const u32x4 z = {0u, 0u, 0u, 0u};
            r.v = ok ? t : z;
        }
    } else {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
        r.lo = ok ? lo : z;
        r.hi = ok ? hi : z;
    }
    return r;
}

template <typename CT>
LXO_DEV void stage_store_rowmajor(CT* lds, const StageReg<CT>& r) {
    if constexpr (is_bf16<CT>::value) {
        *reinterpret_cast<u32x4*>(lds) = r.v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lds[e] = r.lo[e]; lds[4 + e] = r.hi[e]; }
    }
}

LXO_DEV f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------ NT ----
template <typename CT, bool CONV, typename TA, typename OT, int BM, int BN, int BK>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p) {
    constexpr bool BF = is_bf16<CT>::value;
    constexpr int PITCH = BF ? BK + 8 : BK + 1;
    constexpr int TM = BM / 64, TN = BN / 64;     // 32x32 tiles per wave (2x2 waves)
    constexpr int CPR = BK / 8;                   // 8-element chunks per tile row
    constexpr int RPP = 256 / CPR;                // tile rows staged per pass of the 256 threads
    constexpr int AC = BM / RPP, BC = BN / RPP;   // chunks per thread
    __shared__ __attribute__((aligned(16))) CT As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) CT Bs[BN * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const TA* __restrict__ A = reinterpret_cast<const TA*>(p.A);
    const CT* __restrict__ Bp = reinterpret_cast<const CT*>(p.Bp);

    const int srow = tid / CPR, skc = (tid % CPR) * 8;
    // per-thread row descriptors
    bool a_ok[AC]; long long a_base[AC]; int a_oy[AC], a_ox[AC];
#pragma unroll
    for (int j = 0; j < AC; ++j) {
        const int m = m0 + srow + RPP * j;
        a_ok[j] = m < p.M;
        if constexpr (CONV) {
            const int mm = a_ok[j] ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_oy[j] = oy - p.pad; a_ox[j] = ox - p.pad;
            a_base[j] = (long long)b * p.H * p.W;
        } else {
            a_base[j] = (long long)m * p.lda;
            a_oy[j] = a_ox[j] = 0;
        }
    }
    bool b_ok[BC]; long long b_base[BC];
#pragma unroll
    for (int j = 0; j < BC; ++j) {
        const int n = n0 + srow + RPP * j;
        b_ok[j] = n < p.N;
        b_base[j] = (long long)n * p.ldb;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StageReg<CT> ra[AC], rb[BC];
    auto gload = [&](int k0) {
        int kh = 0, kw = 0, ci0 = k0;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;
            ci0 = k0 - tap * p.Cin;
            kh = tap / 3; kw = tap - 3 * kh;
        }
#pragma unroll
        for (int j = 0; j < AC; ++j) {
            if constexpr (CONV) {
                const int iy = a_oy[j] + kh, ix = a_ox[j] + kw;
                const bool ok = a_ok[j] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const long long off = ok ? (a_base[j] + (long long)iy * p.W + ix) * p.Cin + ci0 + skc : 0;
                ra[j] = stage_load<CT, TA>(A + off, ok);
            } else {
                const long long off = a_ok[j] ? a_base[j] + k0 + skc : 0;
                ra[j] = stage_load<CT, TA>(A + off, a_ok[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const long long off = b_ok[j] ? b_base[j] + k0 + skc : 0;
            rb[j] = stage_load<CT, CT>(Bp + off, b_ok[j]);
        }
    };

    gload(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {   // K % BK == 0 (checked by the launcher)
#pragma unroll
        for (int j = 0; j < AC; ++j) stage_store_rowmajor<CT>(&As[(srow + RPP * j) * PITCH + skc], ra[j]);
#pragma unroll
        for (int j = 0; j < BC; ++j) stage_store_rowmajor<CT>(&Bs[(srow + RPP * j) * PITCH + skc], rb[j]);
        __syncthreads();
        if (k0 + BK < p.K) gload(k0 + BK);
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 af[TM], bfr[TN];
                const int kof = ks * 16 + (lane >> 5) * 8;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const u32x4*>(&As[(wm * (BM / 2) + i * 32 + (lane & 31)) * PITCH + kof]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j] = *reinterpret_cast<const u32x4*>(&Bs[(wn * (BN / 2) + j * 32 + (lane & 31)) * PITCH + kof]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(af[i], bfr[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[TM], bfr[TN];
                const int kof = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(wm * (BM / 2) + i * 32 + (lane & 31)) * PITCH + kof];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = Bs[(wn * (BN / 2) + j * 32 + (lane & 31)) * PITCH + kof];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----
    OT* __restrict__ C = reinterpret_cast<OT*>(p.C);
    OT* __restrict__ Cpre = reinterpret_cast<OT*>(p.out_pre);
    const CT* __restrict__ ref = reinterpret_cast<const CT*>(p.relu_ref);
    if (!Cpre && !p.addend && !ref && !p.accumulate && !p.colsum && p.act != 2) {
        // plain product (+ bias, + ReLU): nothing is loaded inside the element loop, so the stores issue back to back
        const float floor_v = p.act == 1 ? 0.f : -3.0e38f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            const bool n_ok = n < p.N;
            const float bias = p.bias ? p.bias[n_ok ? n : 0] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (n_ok && m < p.M) C[(long long)m * p.ldc + n] = from_f32<OT>(fmaxf(p.alpha * acc[i][j][r] + bias, floor_v));
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        const bool n_ok = n < p.N;
        const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
        float csum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (!(n_ok && m < p.M)) continue;
                float v = p.alpha * acc[i][j][r] + bias;
                if (p.act == 1) v = fmaxf(v, 0.f);
                else if (p.act == 2) v = tanhf(v);
                const long long o = (long long)m * p.ldc + n;
                if (Cpre) Cpre[o] = from_f32<OT>(v);
                if (p.addend) v += p.addend[(long long)(m % p.addend_rows) * p.N + n];
                if (ref) v = (to_f32(ref[(long long)m * p.ldr + n]) > 0.f) ? v : 0.f;
                csum += v;
                if (p.accumulate) v += to_f32(C[o]);
                C[o] = from_f32<OT>(v);
            }
        }
        if (p.colsum) {
            csum += __shfl_xor(csum, 32);
            if (lane < 32 && n_ok) atomicAdd(&p.colsum[n], csum);
        }
    }
}

// ------------------------------------------------------------------ TN ----
struct PixIt {                       // running (b, oy, ox) of an im2col row
    int b, oy, ox;
    LXO_DEV void init(int m, int Ho, int Wo) {
        const int hw = Ho * Wo;
        b = m / hw; const int rem = m - b * hw;
        oy = rem / Wo; ox = rem - oy * Wo;
    }
    LXO_DEV void advance(int d, int Ho, int Wo) {
        ox += d;
        while (ox >= Wo) { ox -= Wo; ++oy; }
        while (oy >= Ho) { oy -= Ho; ++b; }
    }
};

// pair two rows of 8 values into 8 dwords {row0[e], row1[e]} (bf16x2), e = 0..7
LXO_DEV void pack_rows(const u32x4& r0, const u32x4& r1, unsigned (&out)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = (r0[i] & 0xffffu) | (r1[i] << 16);
        out[2 * i + 1] = (r0[i] >> 16) | (r1[i] & 0xffff0000u);
    }
}
LXO_DEV void pack_rows(const float (&r0)[8], const float (&r1)[8], unsigned (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = pack_bf2(r0[e], r1[e]);
}

template <typename CT, bool CONV, typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
    constexpr bool BF = is_bf16<CT>::value;
    constexpr int BI = 128, BJ = 128, BR = BF ? 64 : 32;      // reduction rows per LDS tile (bf16: 64 -- half the barriers per row)
    constexpr int PITCH = BF ? BR + 4 : 33;
    __shared__ __attribute__((aligned(16))) CT As[BI * PITCH];
    __shared__ __attribute__((aligned(16))) CT Bs[BJ * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.y * BI, j0 = blockIdx.x * BJ;
    const int batch = blockIdx.z / p.nsplit, split = blockIdx.z - batch * p.nsplit;
    const int per = ((p.M + p.nsplit - 1) / p.nsplit + BR - 1) / BR * BR;
    const int mbeg = split * per;
    const int mend = min(p.M, mbeg + per);
    const TA* __restrict__ A = reinterpret_cast<const TA*>(p.A) + (long long)batch * p.strideA;
    const TB* __restrict__ B = reinterpret_cast<const TB*>(p.B) + (long long)batch * p.strideB;
    float* __restrict__ C = p.C + (long long)batch * p.strideC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging geometry
    //  bf16: two (row pair, 8-column chunk) pieces of A and of B per thread, written as packed bf16x2 dwords
    //  f32 : two (row, 8-column chunk) of A and of B per thread
    constexpr int NQ = 2;
    constexpr int NR = BF ? 2 : 1;
    int s_r[NQ], s_c[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if constexpr (BF) { s_r[q] = 2 * (lane & 15) + 32 * q; s_c[q] = (lane >> 4) + 4 * wave; }
        else { const int c = tid + 256 * q; s_r[q] = c & 31; s_c[q] = c >> 5; }
    }
    int c_kh[NQ], c_kw[NQ], c_ci[NQ];
    bool ai_ok[NQ], bj_ok[NQ];
    PixIt pix[NQ][NR];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = i0 + s_c[q] * 8;
        ai_ok[q] = i < p.I;
        bj_ok[q] = (j0 + s_c[q] * 8) < p.J;
        c_kh[q] = c_kw[q] = c_ci[q] = 0;
        if constexpr (CONV) {
            const int ii = ai_ok[q] ? i : 0;
            const int tap = ii / p.Cin;
            c_ci[q] = ii - tap * p.Cin;
            c_kh[q] = tap / 3; c_kw[q] = tap - 3 * c_kh[q];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) pix[q][rr].init(mbeg + s_r[q] + rr, p.Ho, p.Wo);
        }
    }

    // staged registers: packed pairs (bf16) or raw rows (f32)
    unsigned pa[NQ][8], pb[NQ][8];
    float fa[NQ][8], fb[NQ][8];
    auto gload = [&](int mb) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const TA* ap[NR]; bool aok[NR]; const TB* bq[NR]; bool bok[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int m = mb + s_r[q] + rr;
                aok[rr] = ai_ok[q] && m < mend;
                ap[rr] = A;
                if constexpr (CONV) {
                    const int iy = pix[q][rr].oy - p.pad + c_kh[q], ix = pix[q][rr].ox - p.pad + c_kw[q];
                    aok[rr] = aok[rr] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    if (aok[rr]) ap[rr] = A + (((long long)pix[q][rr].b * p.H + iy) * p.W + ix) * p.Cin + c_ci[q];
                    pix[q][rr].advance(BR, p.Ho, p.Wo);
                } else {
                    if (aok[rr]) ap[rr] = A + (long long)m * p.lda + i0 + s_c[q] * 8;
                }
                bok[rr] = bj_ok[q] && m < mend;
                bq[rr] = bok[rr] ? B + (long long)m * p.ldb + j0 + s_c[q] * 8 : B;
            }
            if constexpr (BF) {
                if constexpr (is_bf16<TA>::value) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    const u32x4 l0 = *reinterpret_cast<const u32x4*>(ap[0]), l1 = *reinterpret_cast<const u32x4*>(ap[1]);   // unconditional: see stage_load
                    const u32x4 r0 = aok[0] ? l0 : z, r1 = aok[1] ? l1 : z;
                    pack_rows(r0, r1, pa[q]);
                } else {
                    float r0[8], r1[8];
                    load8(ap[0], r0); load8(ap[1], r1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { r0[e] = aok[0] ? r0[e] : 0.f; r1[e] = aok[1] ? r1[e] : 0.f; }
                    pack_rows(r0, r1, pa[q]);
                }
                if constexpr (is_bf16<TB>::value) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    const u32x4 l0 = *reinterpret_cast<const u32x4*>(bq[0]), l1 = *reinterpret_cast<const u32x4*>(bq[1]);
                    const u32x4 r0 = bok[0] ? l0 : z, r1 = bok[1] ? l1 : z;
                    pack_rows(r0, r1, pb[q]);
                } else {
                    float r0[8], r1[8];
                    load8(bq[0], r0); load8(bq[1], r1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { r0[e] = bok[0] ? r0[e] : 0.f; r1[e] = bok[1] ? r1[e] : 0.f; }
                    pack_rows(r0, r1, pb[q]);
                }
            } else {
                load8(reinterpret_cast<const float*>(ap[0]), fa[q]);
                load8(reinterpret_cast<const float*>(bq[0]), fb[q]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { fa[q][e] = aok[0] ? fa[q][e] : 0.f; fb[q][e] = bok[0] ? fb[q][e] : 0.f; }
            }
        }
    };

    if (mbeg < mend) gload(mbeg);
    for (int mb = mbeg; mb < mend; mb += BR) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (BF) {
                    *reinterpret_cast<unsigned*>(&As[(s_c[q] * 8 + e) * PITCH + s_r[q]]) = pa[q][e];
                    *reinterpret_cast<unsigned*>(&Bs[(s_c[q] * 8 + e) * PITCH + s_r[q]]) = pb[q][e];
                } else {
                    As[(s_c[q] * 8 + e) * PITCH + s_r[q]] = fa[q][e];
                    Bs[(s_c[q] * 8 + e) * PITCH + s_r[q]] = fb[q][e];
                }
            }
        }
        __syncthreads();
        if (mb + BR < mend) gload(mb + BR);
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < BR / 16; ++ks) {
                u32x4 af[2], bfr[2];
                const int kof = ks * 16 + (lane >> 5) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const CT* pa_ = &As[(wi * 64 + i * 32 + (lane & 31)) * PITCH + kof];
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(pa_), hi = *reinterpret_cast<const u32x2*>(pa_ + 4);
                    af[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    const CT* pb_ = &Bs[(wj * 64 + i * 32 + (lane & 31)) * PITCH + kof];
                    const u32x2 lo2 = *reinterpret_cast<const u32x2*>(pb_), hi2 = *reinterpret_cast<const u32x2*>(pb_ + 4);
                    bfr[i] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(af[i], bfr[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                float af[2], bfr[2];
                const int kof = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = As[(wi * 64 + i * 32 + (lane & 31)) * PITCH + kof];
                    bfr[i] = Bs[(wj * 64 + i * 32 + (lane & 31)) * PITCH + kof];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int jj = j0 + wj * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ii = i0 + wi * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (ii < p.I && jj < p.J) {
                    float* dst = &C[(long long)ii * p.ldc + jj];
                    if (p.atomic) atomicAdd(dst, acc[i][j][r]);
                    else *dst = acc[i][j][r];
                }
            }
    }
}

// ------------------------------------------------------------- skinny ----
// C[M<=64 per block row][N] = act(alpha * A * Bp^T + bias) for the per-step recurrent GEMMs
// (M = batch).  These are latency-bound: one workgroup per 32 output columns, the four
// waves split K four ways, every operand fragment is loaded straight from global memory in
// MFMA layout with 8 k-steps of loads in flight, and the four partial tiles meet in LDS.
template <typename CT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmNT p) {
    constexpr bool BF = is_bf16<CT>::value;
    __shared__ float red[4][64][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
    const int kq = p.K >> 2, kb = wave * kq;
    const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
    const CT* __restrict__ Bp = reinterpret_cast<const CT*>(p.Bp);
    const int row0 = m0 + (lane & 31), row1 = row0 + 32, col = n0 + (lane & 31);
    const bool ok0 = row0 < p.M, ok1 = row1 < p.M, okb = col < p.N;
    const float* a0 = A + (long long)(ok0 ? row0 : 0) * p.lda + kb;
    const float* a1 = A + (long long)(ok1 ? row1 : 0) * p.lda + kb;
    const CT* bp = Bp + (long long)(okb ? col : 0) * p.ldb + kb;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
        const int h8 = (lane >> 5) * 8;
        for (int k = 0; k < kq; k += 128) {
            f32x4 x0[8][2], x1[8][2]; u32x4 bb[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int kk = k + s * 16 + h8;
                const bool in = kk < kq;
                x0[s][0] = (ok0 && in) ? *reinterpret_cast<const f32x4*>(a0 + kk) : z4;
                x0[s][1] = (ok0 && in) ? *reinterpret_cast<const f32x4*>(a0 + kk + 4) : z4;
                x1[s][0] = (ok1 && in) ? *reinterpret_cast<const f32x4*>(a1 + kk) : z4;
                x1[s][1] = (ok1 && in) ? *reinterpret_cast<const f32x4*>(a1 + kk + 4) : z4;
                const u32x4 zb = {0u, 0u, 0u, 0u};
                bb[s] = (okb && in) ? *reinterpret_cast<const u32x4*>(bp + kk) : zb;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 f0 = {pack_bf2(x0[s][0][0], x0[s][0][1]), pack_bf2(x0[s][0][2], x0[s][0][3]),
                                  pack_bf2(x0[s][1][0], x0[s][1][1]), pack_bf2(x0[s][1][2], x0[s][1][3])};
                const u32x4 f1 = {pack_bf2(x1[s][0][0], x1[s][0][1]), pack_bf2(x1[s][0][2], x1[s][0][3]),
                                  pack_bf2(x1[s][1][0], x1[s][1][1]), pack_bf2(x1[s][1][2], x1[s][1][3])};
                acc0 = mfma_bf16(f0, bb[s], acc0);
                acc1 = mfma_bf16(f1, bb[s], acc1);
            }
        }
    } else {
        // k permutation: half h of the wave owns k = 8j + 4h + e, used at k-step (j, e); A and B agree
        const int h4 = (lane >> 5) * 4;
        for (int k = 0; k < kq; k += 64) {
            f32x4 x0[8], x1[8], bb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = k + j * 8 + h4;
                const bool in = kk < kq;
                x0[j] = (ok0 && in) ? *reinterpret_cast<const f32x4*>(a0 + kk) : z4;
                x1[j] = (ok1 && in) ? *reinterpret_cast<const f32x4*>(a1 + kk) : z4;
                bb[j] = (okb && in) ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(bp) + kk) : z4;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[j][e], bb[j][e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[j][e], bb[j][e], acc1, 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[wave][rr][lane & 31] = acc0[r];
        red[wave][rr + 32][lane & 31] = acc1[r];
    }
    __syncthreads();
    const int orow = tid >> 2, oc0 = (tid & 3) * 8;
    const int m = m0 + orow;
    if (m < p.M) {
        float* __restrict__ C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = n0 + oc0 + e;
            if (n >= p.N) continue;
            float v = red[0][orow][oc0 + e] + red[1][orow][oc0 + e] + red[2][orow][oc0 + e] + red[3][orow][oc0 + e];
            v = p.alpha * v + (p.bias ? p.bias[n] : 0.f);
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = tanhf(v);
            const long long o = (long long)m * p.ldc + n;
            if (p.accumulate) v += C[o];
            C[o] = v;
        }
    }
}

// --------------------------------------------------------------- slab ----
// Split-K variant for the recurrent steps: grid (N/64, K/128, M/64).  Every workgroup
// issues ALL its loads at once (A 64x128 f32 coalesced, B 64x128 compute dtype), converts
// through LDS and writes one partial 64x64 tile to slab[ks] (plain stores, no atomics);
// the consumer kernel adds the KS slabs.  One memory round trip per GEMM.
template <typename CT>
__global__ __launch_bounds__(256) void gemm_slab_kernel(GemmNT p, float* __restrict__ slab, long long slab_stride) {
    constexpr bool BF = is_bf16<CT>::value;
    constexpr int KQ = 128;
    constexpr int PITCH = BF ? KQ + 8 : KQ + 1;
    __shared__ __attribute__((aligned(16))) CT As[64 * PITCH];
    __shared__ __attribute__((aligned(16))) CT Bs[64 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y, m0 = blockIdx.z * 64;
    const int kb = ks * KQ;
    const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
    const CT* __restrict__ Bp = reinterpret_cast<const CT*>(p.Bp);
    // A: 64 rows x 128 floats = 2048 float4 -> 8 per thread (a wave reads 2 full rows per instruction)
    f32x4 ra[8];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
        const int m = m0 + row;
        ra[j] = m < p.M ? *reinterpret_cast<const f32x4*>(A + (long long)m * p.lda + kb + c4) : z4;
    }
    if constexpr (BF) {
        u32x4 rb[4];
        const u32x4 zb = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, row = idx >> 4, c8 = (idx & 15) * 8;
            const int n = n0 + row;
            rb[j] = n < p.N ? *reinterpret_cast<const u32x4*>(Bp + (long long)n * p.ldb + kb + c8) : zb;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
            u32x2 v = {pack_bf2(ra[j][0], ra[j][1]), pack_bf2(ra[j][2], ra[j][3])};
            *reinterpret_cast<u32x2*>(&As[row * PITCH + c4]) = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, row = idx >> 4, c8 = (idx & 15) * 8;
            *reinterpret_cast<u32x4*>(&Bs[row * PITCH + c8]) = rb[j];
        }
    } else {
        f32x4 rb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
            const int n = n0 + row;
            rb[j] = n < p.N ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Bp) + (long long)n * p.ldb + kb + c4) : z4;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { As[row * PITCH + c4 + e] = ra[j][e]; Bs[row * PITCH + c4 + e] = rb[j][e]; }
        }
    }
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (BF) {
#pragma unroll
        for (int s8 = 0; s8 < KQ / 16; ++s8) {
            const int kof = s8 * 16 + (lane >> 5) * 8;
            const u32x4 af = *reinterpret_cast<const u32x4*>(&As[(wm * 32 + (lane & 31)) * PITCH + kof]);
            const u32x4 bfr = *reinterpret_cast<const u32x4*>(&Bs[(wn * 32 + (lane & 31)) * PITCH + kof]);
            acc = mfma_bf16(af, bfr, acc);
        }
    } else {
#pragma unroll 8
        for (int s2 = 0; s2 < KQ / 2; ++s2) {
            const int kof = s2 * 2 + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(wm * 32 + (lane & 31)) * PITCH + kof],
                                                       Bs[(wn * 32 + (lane & 31)) * PITCH + kof], acc, 0, 0, 0);
        }
    }
    float* __restrict__ out = slab + (long long)ks * slab_stride;
    const int n = n0 + wn * 32 + (lane & 31);
    if (n < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < p.M) out[(long long)m * p.ldc + n] = acc[r];
        }
    }
}

template <typename CT>
int launch_skinny(const GemmNT& p, hipStream_t s) {
    dim3 grid(cdiv(p.N, 32), cdiv(p.M, 64));
    hipLaunchKernelGGL((gemm_skinny_kernel<CT>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

template <typename CT, bool CONV, typename TA, typename OT, int BM, int BN, int BK = 32>
int launch_nt(const GemmNT& p, hipStream_t s) {
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM));
    hipLaunchKernelGGL((gemm_nt_kernel<CT, CONV, TA, OT, BM, BN, BK>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
}
template <typename CT, bool CONV, typename TA, typename TB>
int launch_tn(const GemmTN& p, hipStream_t s) {
    dim3 grid(cdiv(p.J, 128), cdiv(p.I, 128), p.nbatch * p.nsplit);
    hipLaunchKernelGGL((gemm_tn_kernel<CT, CONV, TA, TB>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

}  // namespace

int lxo_launch_gemm_nt(int dt, int a_f32, int c_f32, int small, const GemmNT& p0, hipStream_t s) {
    if (p0.colsum_part) {
        // deterministic column sums: the bf16 two-workgroup conv kernel has a slot form; everything else runs without the fused sum and the
        // columns of the stored result are summed through ordered row-block slots (the same scratch)
        if (dt == LXO_BF16 && p0.conv && !a_f32 && !c_f32 && p0.Cin % 64 == 0 && p0.Cin % 32 == 0 && p0.K % 32 == 0 && p0.K > 0 && p0.M > 0 && p0.N > 0) {
            const int rc = lxo_launch_conv_igemm(p0, s);
            if (rc != -7) return rc;
        }
        GemmNT q = p0; q.colsum = nullptr; q.colsum_part = nullptr;
        if (int rc = lxo_launch_gemm_nt(dt, a_f32, c_f32, small, q, s)) return rc;
        const DetScratch d = {p0.colsum_part, p0.colsum_part_floats};
        return lxo_k_colsum_det(p0.C, (dt == LXO_BF16 && !c_f32) ? 1 : 0, p0.ldc, p0.colsum, p0.M, p0.N, d, s);
    }
    const GemmNT& p = p0;
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.K % 32 != 0 || p.K <= 0) return -2;
    if (p.conv && p.Cin % 32) return -2;
    const bool plain = !p.conv && !p.addend && !p.relu_ref && !p.out_pre && !p.colsum;
    if (small && plain && p.K % 256 == 0 && (dt == LXO_F32 || (a_f32 && c_f32)) && p.lda % 4 == 0)
        return dt == LXO_F32 ? launch_skinny<float>(p, s) : launch_skinny<bf16_t>(p, s);
    if (dt == LXO_F32) {
        if (p.conv) return launch_nt<float, true, float, float, 128, 128>(p, s);
        if (small) return launch_nt<float, false, float, float, 64, 64>(p, s);
        return launch_nt<float, false, float, float, 128, 128>(p, s);
    }
    if (p.conv) {
        if (a_f32 || c_f32) return -3;
        if (p.Cin % 64 == 0) return lxo_launch_conv_igemm(p, s);
        return launch_nt<bf16_t, true, bf16_t, bf16_t, 128, 128>(p, s);
    }
    if (small) {
        if (!(a_f32 && c_f32)) return -3;
        return launch_nt<bf16_t, false, float, float, 64, 64>(p, s);
    }
    const bool k64 = p.K % 64 == 0;
    if (!a_f32 && k64 && plain && !p.accumulate && p.act == 0 && p.alpha == 1.f && p.lda % 8 == 0 && p.ldb % 8 == 0 &&
        (((uintptr_t)p.A | (uintptr_t)p.Bp) & 15) == 0) {
        static int use_dma = -1;                           // LXO_GEMM_NT_DMA=0: the register-staged kernel for every dense NT GEMM (A/B)
        if (use_dma < 0) { const char* e = getenv("LXO_GEMM_NT_DMA"); use_dma = (e && e[0] == '0') ? 0 : 1; }
        if (use_dma) { const int rc = lxo_launch_gemm_nt_dma(p, c_f32, s); if (rc != -2) return rc; }   // -2: does not qualify (e.g. 2 GB operands): the register-staged kernel
    }
    if (!a_f32 && !c_f32) return k64 ? launch_nt<bf16_t, false, bf16_t, bf16_t, 128, 128, 64>(p, s) : launch_nt<bf16_t, false, bf16_t, bf16_t, 128, 128>(p, s);
    if (!a_f32 && c_f32) return k64 ? launch_nt<bf16_t, false, bf16_t, float, 128, 128, 64>(p, s) : launch_nt<bf16_t, false, bf16_t, float, 128, 128>(p, s);
    if (a_f32 && c_f32) return k64 ? launch_nt<bf16_t, false, float, float, 128, 128, 64>(p, s) : launch_nt<bf16_t, false, float, float, 128, 128>(p, s);
    return -3;
}

int lxo_launch_gemm_tn(int dt, int a_f32, int b_f32, const GemmTN& p, hipStream_t s) {
    if (p.I <= 0 || p.J <= 0 || p.M <= 0) return 0;
    if (!p.atomic && p.nsplit != 1) return -2;
    if (dt == LXO_F32) {
        // f32 is the parity mode: ONE row range per output tile, so no two workgroups add into the same element and the sum over the rows
        // runs in the same order in every run (bit-reproducible gradients); the bf16 kernels keep their split ranges + f32 atomics
        GemmTN q = p; q.nsplit = 1;
        if (q.conv) return launch_tn<float, true, float, float>(q, s);
        return launch_tn<float, false, float, float>(q, s);
    }
    if (p.det_slab && !p.conv) {
        // deterministic bf16 mode, dense product: the transposing-read kernel keeps its row ranges and stores one partial tile per (range,
        // tile) to the slab, added in range order (gemm_tn_tr.hip); operands it does not take: ONE row range per output tile
        if (!a_f32 && !b_f32 && p.atomic && p.nbatch == 1 && p.lda % 8 == 0 && p.ldb % 8 == 0 && (p.I + 7) / 8 * 8 <= p.lda && (p.J + 7) / 8 * 8 <= p.ldb &&
            (((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0) { const int rc = lxo_launch_gemm_tn_tr(p, s); if (rc != -2) return rc; }
        GemmTN q = p; q.nsplit = 1; q.det_slab = nullptr;
        return lxo_launch_gemm_tn(dt, a_f32, b_f32, q, s);
    }
    if (p.conv) {
        if (a_f32 || b_f32) return -3;
        static int use_halo = -1;
        if (use_halo < 0) { const char* e = getenv("LXO_WGRAD_HALO"); use_halo = (e && e[0] == '0') ? 0 : 1; }
        if (use_halo && p.Cin % 64 == 0 && p.J % 8 == 0 && p.atomic && p.nbatch == 1) { const int rc = lxo_launch_conv_wgrad(p, s); if (rc != -2) return rc; }
        if (p.det_slab) { GemmTN q = p; q.nsplit = 1; return launch_tn<bf16_t, true, bf16_t, bf16_t>(q, s); }
        return launch_tn<bf16_t, true, bf16_t, bf16_t>(p, s);
    }
    if (!a_f32 && !b_f32) {
        static int use_tr = -1;                            // LXO_GEMM_TN_TR=0: the packing kernel for every dense TN GEMM (A/B)
        if (use_tr < 0) { const char* e = getenv("LXO_GEMM_TN_TR"); use_tr = (e && e[0] == '0') ? 0 : 1; }
        if (use_tr && p.atomic && p.nbatch == 1 && p.lda % 8 == 0 && p.ldb % 8 == 0 && (p.I + 7) / 8 * 8 <= p.lda && (p.J + 7) / 8 * 8 <= p.ldb &&
            (((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0) { const int rc = lxo_launch_gemm_tn_tr(p, s); if (rc != -2) return rc; }
        return launch_tn<bf16_t, false, bf16_t, bf16_t>(p, s);
    }
    if (a_f32 && !b_f32) return launch_tn<bf16_t, false, float, bf16_t>(p, s);
    if (a_f32 && b_f32) return launch_tn<bf16_t, false, float, float>(p, s);
    return launch_tn<bf16_t, false, bf16_t, float>(p, s);
}

int lxo_launch_gemm_slab(int dt, const GemmNT& p, float* slab, long long slab_stride, hipStream_t s) {
    if (p.K % 128 != 0 || p.lda % 4 != 0) return -2;
    dim3 grid(cdiv(p.N, 64), p.K / 128, cdiv(p.M, 64));
    if (dt == LXO_F32) hipLaunchKernelGGL((gemm_slab_kernel<float>), grid, dim3(256), 0, s, p, slab, slab_stride);
    else hipLaunchKernelGGL((gemm_slab_kernel<bf16_t>), grid, dim3(256), 0, s, p, slab, slab_stride);
    return (int)hipGetLastError();
}
