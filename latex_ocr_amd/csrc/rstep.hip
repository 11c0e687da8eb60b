// Recurrent-step GEMMs with the cell's point-wise stage fused into the epilogue.
//
// One AttentionCell.step (attention_cell.py:58-89) is a chain of DEPENDENT launches; measured on MI355X
// (tools/launch_probe.hip, tools/loop_probe.py) the chain is device-bound: 1.53 us per kernel boundary plus
// ~3 us per body, and every split-K GEMM leaves 1-4 MB of f32 partial products for its consumer to re-read.
// The kernels here take the other decomposition: a workgroup owns 16 OUTPUT COLUMNS for a tile of 64 / 32 / 16 rows over the
// FULL contraction, so the product is final inside the workgroup and the consumer stage (LSTM gates, tanh, LSTM
// backward, the carried tanh') runs in the epilogue: one launch instead of two, no slabs.
//
//   C[MT x 16] = A[MT x K] * W[16 x K]^T          4 waves, each contracts a quarter of K for all MT rows
//   v_mfma_f32_16x16x32_bf16 (v_mfma_f32_16x16x4_f32 in the f32 parity mode), operands loaded straight into
//   MFMA fragment layout (16 B per lane; the k permutation inside a 32-wide step is free because A and W
//   use the same one), ALL loads of a chunk in flight at once (one memory round trip), partial tiles of the
//   four waves summed through 17 KB of LDS, then 256 threads run the epilogue on (row, 4 columns).
//
// A is re-read by every workgroup (<= 128 KB bf16, L2-resident); W is read exactly once per launch.
#include "rstep.h"
#include "drop.h"
#include "api_util.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float v4f;

LXO_DEV v4f mfma16_bf16(u32x4 a, u32x4 b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// The cell's transcendentals.  f32 (parity) mode: the library functions.  bf16 mode: v_exp_f32 / v_rcp_f32 forms (abs error ~1e-7, far
// below what the bf16 mirrors of h / o keep): in-kernel stamps put 1.8-2.2 k cycles of a 5-6 k cycle step kernel into the epilogue,
// most of it the five library calls per LSTM element.  Saturation is exact (e^{2x} -> inf gives 1, -> 0 gives -1 / 0).
template <bool FAST> LXO_DEV float tanh_e(float x) {
    if constexpr (FAST) return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.f);
    else return tanhf(x);
}
template <bool FAST> LXO_DEV float sigm_e(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    else return sigmoidf_(x);
}

// 8 consecutive k of one row as a bf16 MFMA fragment
LXO_DEV u32x4 frag8(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
LXO_DEV u32x4 frag8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    u32x4 r = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
    return r;
}

LXO_DEV void st4(float* p, const float (&v)[4]) { f32x4 a = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = a; }
LXO_DEV void st4b(bf16_t* p, const float (&v)[4]) { u32x2 a = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}; *reinterpret_cast<u32x2*>(p) = a; }
LXO_DEV void ld4(const float* p, float (&v)[4]) { const f32x4 a = *reinterpret_cast<const f32x4*>(p); v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }

// AT: element type of A in memory (bf16 mirror, or float converted while staging); WT: compute dtype (bf16 / float = parity mode)
// MT: rows per workgroup (64, 32 or 16: the launcher takes the smallest that still gives every CU at most one workgroup); KC: k elements of this wave's share staged per chunk; NCH: chunks (compile time, so that
// every s_waitcnt is counted: staging chunk c waits for ITS loads only while the later chunks are still landing);
// NBUF: chunks whose loads are in flight in registers (chunks beyond that are re-issued into freed registers).
// Every wave stages ITS OWN quarter of the contraction for all MT rows through its own LDS region: global loads are
// row-contiguous 128-byte pieces (fragment-shaped global loads of 16 rows x 64 B measured 2x slower end to end:
// texture-addresser bound), the MFMA fragments come out of LDS (pitch = KC + 16 bytes: an odd number of 16-byte
// bank groups).  No workgroup barrier in the main loop: a wave only reads LDS it wrote itself, and the LDS pipe
// is in order per wave.  The epilogue's own operands are requested before the main loop so that their memory round
// trip overlaps the GEMM.  In-kernel stamps (tools/rstep_stamps.py): a CU accepts one 1-KB wave load per ~24 cycles, so
// the fetch of A (re-read by every workgroup) sets the kernel's length -- hence MT = 32 / 16 when that fills more CUs.
template <typename AT, typename WT, int EPI, int MT, int KC, int NCH, int NBUF>
__global__ __launch_bounds__(256) void rstep_kernel(const void* kA, const void* kW, int k_lda, int k_ldw, int k_M, int k_N, int k_K, int k_U, const int* k_apar, int k_ak, RStep p) {
    // The leading scalars repeat p.A, p.W, p.lda, p.ldw, p.M, p.N, p.K, p.U: scalar kernel arguments are PRELOADED into SGPRs at wave
    // launch (-mllvm -amdgpu-kernarg-preload-count, Makefile), a by-value struct is not -- so the operand addresses and the first
    // loads do not wait for the scalar loads of the 280-byte argument block (a memory round trip at the head of every launch of
    // the recurrence); the rest of `p` (epilogue operands, outputs) arrives while those loads are in flight.
    constexpr bool BF = is_bf16<WT>::value;
    constexpr int RB = MT / 16;                                              // 16-row MFMA blocks per wave
    constexpr int PITCH = KC + (BF ? 8 : 4);
    constexpr int APL = 16 / (int)sizeof(AT), WPL = 16 / (int)sizeof(WT);   // elements per 16-byte load
    constexpr int PRA = KC / APL, PRW = KC / WPL;                            // 16-byte pieces per tile row
    constexpr int NA = (MT * PRA) / 64, NW = (16 * PRW) / 64;                // loads per lane per chunk (A: MT rows, W: 16 rows)
    constexpr int KSTEP = BF ? 32 : 16, NS = KC / KSTEP;
    static_assert(NW >= 1 && NA >= 1, "chunk too small");
    __shared__ __attribute__((aligned(16))) WT lds[4][(MT + 16) * PITCH];
    __shared__ float red[4][MT][16 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * MT;
    const int kq = k_K >> 2;                                  // this wave's share of the contraction (= KC * NCH)
    WT* As = lds[wave];
    WT* Ws = lds[wave] + MT * PITCH;
    const AT* __restrict__ Ag = reinterpret_cast<const AT*>(kA) + wave * kq;
    const WT* __restrict__ Wg = reinterpret_cast<const WT*>(kW) + wave * kq;
    // per-lane source rows of the staging loads
    int aoff[NA], woff[NW];                                   // element offsets inside one step's operands: well below 2^31
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int i = j * 64 + lane, row = i / PRA, pc = i - row * PRA;
        int m = m0 + row;
        if (m >= k_M) m = k_M - 1;
        if (k_apar) m = (m / k_ak) * k_ak + k_apar[m];       // beam decode: the parent hypothesis' row (one dependent load at the head instead of a re-ordering launch)
        aoff[j] = m * k_lda + pc * APL;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = j * 64 + lane, row = i / PRW, pc = i - row * PRW;
        int nb;                                               // output column of tile row `row`
        if constexpr (EPI == RS_LSTM_FWD) nb = (row >> 2) * k_U + blockIdx.x * 4 + (row & 3);   // 4 units x gates i,j,f,o
        else nb = blockIdx.x * 16 + row;
        if (nb >= k_N) nb = k_N - 1;
        woff[j] = nb * k_ldw + pc * WPL;
    }
    u32x4 ra[NBUF][NA], rw[NBUF][NW];
#define ISSUE(buf, c) do { const int kb_ = (c) * KC; \
        _Pragma("unroll") for (int j = 0; j < NW; ++j) rw[buf][j] = *reinterpret_cast<const u32x4*>(Wg + woff[j] + kb_); \
        _Pragma("unroll") for (int j = 0; j < NA; ++j) ra[buf][j] = *reinterpret_cast<const u32x4*>(Ag + aoff[j] + kb_); } while (0)
#define STAMP(i) do { if (p.dbg && tid == 0) p.dbg[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#pragma unroll
    for (int b = 0; b < NBUF; ++b) if (b < NCH) ISSUE(b, b);
    STAMP(0);                                                 // p.dbg is part of the argument block: nothing of it is touched before the loads are out

    // ---- epilogue operands, requested now ----
    constexpr int ETH = MT * 4;                               // threads that run the epilogue: (row, 4 columns) each
    const int row = (tid >> 2) % MT, part = tid & 3;
    const int m = m0 + row;
    const bool mok = m < p.M && tid < ETH;
    const int mc = m < p.M ? m : p.M - 1;
    const int n = blockIdx.x * 16 + part * 4;                 // first of this thread's 4 output columns (generic epilogues)
    float pz[4] = {0.f, 0.f, 0.f, 0.f}, pcp = 0.f;            // LSTM_FWD
    float e0[4], e1[4], e2[4], e3[4], e4[4], e5[4], e6[4], e7[4], e8[4];
    if constexpr (EPI == RS_LSTM_FWD) {
        const int U = p.U, u = blockIdx.x * 4 + part;
        long long zr = mc;                                    // row of the x-part: the step's own row, or the token's row of the decode table
        if (p.zx_idx) { int id = p.zx_idx[mc]; id = id < 0 ? 0 : (id >= p.zx_vocab ? p.zx_vocab - 1 : id); zr = id; }
        else if (p.zx_row >= 0) zr = p.zx_row;
#pragma unroll
        for (int q = 0; q < 4; ++q) pz[q] = p.zx[zr * 4 * U + q * U + u];
        pcp = p.c_prev[(long long)(k_apar ? (mc / k_ak) * k_ak + k_apar[mc] : mc) * U + u];
    } else if constexpr (EPI == RS_LSTM_BWD) {
        const int U = p.U;
        const float* gr = p.gates_in + (long long)mc * 4 * U + n;
        ld4(p.dhm + (long long)mc * p.lddhm + n, e0);
        ld4(p.carry_h + (long long)mc * U + n, e1);           // unconditional (the buffer always exists); rows without a carry are zeroed below
        if (mc >= p.carry_rows) { e1[0] = e1[1] = e1[2] = e1[3] = 0.f; }
        ld4(gr, e2); ld4(gr + U, e3); ld4(gr + 2 * U, e4); ld4(gr + 3 * U, e5);
        ld4(p.c_cur + (long long)mc * U + n, e6); ld4(p.c_prev + (long long)mc * U + n, e7); ld4(p.dcc + (long long)mc * U + n, e8);
    } else if constexpr (EPI == RS_CARRY) {
        if (!p.first && n < p.O) {
            ld4(p.dolog + (long long)mc * p.O + n, e0);
            ld4(p.o_prev + (long long)mc * p.ldoprev + n, e1);
        }
    }
    __builtin_amdgcn_sched_barrier(0);                        // everything above is in flight before anything waits
    STAMP(1);

    v4f acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        constexpr int dummy = 0; (void)dummy;
        const int b = c % NBUF;                               // compile-time after unrolling
        // registers -> this wave's LDS tile
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = j * 64 + lane, trow = i / PRW, pc = i - trow * PRW;
            *reinterpret_cast<u32x4*>(&Ws[trow * PITCH + pc * WPL]) = rw[b][j];
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int i = j * 64 + lane, trow = i / PRA, pc = i - trow * PRA;
            if constexpr (sizeof(AT) == sizeof(WT)) {
                *reinterpret_cast<u32x4*>(&As[trow * PITCH + pc * APL]) = ra[b][j];
            } else {                                          // float in memory, bf16 in LDS
                const f32x4 f = __builtin_bit_cast(f32x4, ra[b][j]);
                u32x2 v = {pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3])};
                *reinterpret_cast<u32x2*>(&As[trow * PITCH + pc * APL]) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (c == 0) STAMP(2);
        if (c + NBUF < NCH) ISSUE(b, c + NBUF);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (BF) {
                const u32x4 bw = *reinterpret_cast<const u32x4*>(&Ws[r * PITCH + s * 32 + g * 8]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[rb] = mfma16_bf16(*reinterpret_cast<const u32x4*>(&As[(rb * 16 + r) * PITCH + s * 32 + g * 8]), bw, acc[rb]);
            } else {
                // 16-wide step: lane (r, g) holds k = 4g..4g+3; MFMA e takes element e of every lane (the same k set in A and W)
                const f32x4 bw = *reinterpret_cast<const f32x4*>(&Ws[r * PITCH + s * 16 + g * 4]);
                f32x4 aw[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) aw[rb] = *reinterpret_cast<const f32x4*>(&As[(rb * 16 + r) * PITCH + s * 16 + g * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[rb][e], bw[e], acc[rb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                      // the tile is rewritten by the next chunk: keep this chunk's reads ahead of it
        if (c == 0) STAMP(3);
    }
#undef ISSUE
    STAMP(4);
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][rb * 16 + g * 4 + i][r] = acc[rb][i];
    __syncthreads();
    STAMP(5);
    if (!mok) return;

    if constexpr (EPI == RS_LSTM_FWD) {
        // TF-1.12 LSTMCell, gate order i,j,f,o, forget_bias 1.0 (attention_cell.py:71): this thread finishes unit u of row m
        const int U = p.U, u = blockIdx.x * 4 + part;
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float z = pz[q] + red[0][row][q * 4 + part] + red[1][row][q * 4 + part] + red[2][row][q * 4 + part] + red[3][row][q * 4 + part];
            a[q] = (q == 1) ? tanh_e<BF>(z) : sigm_e<BF>(q == 2 ? z + 1.0f : z);
        }
        const float c = a[2] * pcp + a[0] * a[1];
        const float h = a[3] * tanh_e<BF>(c);
        const float ht = h * drop_scale(p.dr, 1u, m, u, U);      // h~ = dropout(h): what attention and the o projection read (attention_cell.py:72)
        if (p.gates) {
#pragma unroll
            for (int q = 0; q < 4; ++q) p.gates[(long long)m * 4 * U + q * U + u] = a[q];
        }
        p.c_out[(long long)m * U + u] = c;
        p.out[(long long)m * p.ldo + u] = h;
        p.out2[(long long)m * p.ldo + u] = ht;
        if (p.outb) { p.outb[(long long)m * p.ldob + u] = f2bf(h); p.out2b[(long long)m * p.ldob + u] = f2bf(ht); }
        STAMP(6);
        return;
    } else {
        if (n >= p.N) return;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = red[0][row][part * 4 + e] + red[1][row][part * 4 + e] + red[2][row][part * 4 + e] + red[3][row][part * 4 + e];
        if constexpr (EPI == RS_PLAIN || EPI == RS_TANH_O) {
            // bias / accumulate: only the launches outside the loops set them (uniform branches, their loads wait here)
            if (p.bias) { float b4[4]; ld4(p.bias + n, b4); for (int e = 0; e < 4; ++e) v[e] += b4[e]; }
            if (p.accumulate) { float o4[4]; ld4(p.out + (long long)m * p.ldo + n, o4); for (int e = 0; e < 4; ++e) v[e] += o4[e]; }
        }
        if constexpr (EPI == RS_PLAIN) {
            st4(p.out + (long long)m * p.ldo + n, v);
            if (p.outb) st4b(p.outb + (long long)m * p.ldob + n, v);
        } else if constexpr (EPI == RS_TANH_O) {
            // o = dropout(tanh(.))      (attention_cell.py:82-83)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanh_e<BF>(v[e]) * drop_scale(p.dr, 2u, m, n + e, p.N);
            st4(p.out + (long long)m * p.ldo + n, v);
            if (p.outb) st4b(p.outb + (long long)m * p.ldob + n, v);
        } else if constexpr (EPI == RS_LSTM_BWD) {
            // backward of the LSTM cell for units n..n+3 of row m; v = d_att_h * W_att_h^T (attention's share of d_h~)
            // e0 = d_h~ from the o projection, e1 = carried d_h, e2..e5 = gates i,j,f,o, e6 = c_t, e7 = c_{t-1}, e8 = carried d_c
            const int U = p.U;
            float dzi[4], dzj[4], dzf[4], dzo[4], dcn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dh = (e0[e] + v[e]) * drop_scale(p.dr, 1u, m, n + e, U) + e1[e];
                const float tc = tanh_e<BF>(e6[e]);
                const float dc = e8[e] + dh * e5[e] * (1.f - tc * tc);
                dzi[e] = dc * e3[e] * e2[e] * (1.f - e2[e]);
                dzj[e] = dc * e2[e] * (1.f - e3[e] * e3[e]);
                dzf[e] = dc * e7[e] * e4[e] * (1.f - e4[e]);
                dzo[e] = dh * tc * e5[e] * (1.f - e5[e]);
                dcn[e] = dc * e4[e];
            }
            float* dzr = p.out + (long long)m * 4 * U + n;
            st4(dzr, dzi); st4(dzr + U, dzj); st4(dzr + 2 * U, dzf); st4(dzr + 3 * U, dzo);
            if (p.outb) {
                bf16_t* db = p.outb + (long long)m * p.ldob + n;
                st4b(db, dzi); st4b(db + U, dzj); st4b(db + 2 * U, dzf); st4b(db + 3 * U, dzo);
            }
            st4(p.dcc + (long long)m * U + n, dcn);
        } else if constexpr (EPI == RS_CARRY) {
            // v = d_z K[D:]^T: columns < O are d_o carried to step t-1, the rest d_h carried to step t-1
            const int O = p.O;
            if (p.first) { st4(p.out + (long long)m * p.ldo + n, v); STAMP(6); return; }   // t == 0: raw carries for the initial-state gradients
            if (n < O) {
                // g_{t-1} = (d_o from the logits (e0) + d_o carry) * dropout mask * (1 - tanh^2), tanh from o_{t-1} (e1)   (attention_cell.py:82-83 backward)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = drop_scale(p.dr, 2u, m, n + e, O);
                    const float th = (p.dr.thr == 0u) ? e1[e] : e1[e] / p.dr.inv_keep;   // rec holds the dropped o; where the mask is 1 tanh = o * keep
                    v[e] = (e0[e] + v[e]) * sc * (1.f - th * th);
                }
                st4(p.out + (long long)m * O + n, v);
                if (p.outb) st4b(p.outb + (long long)m * p.ldob + n, v);
            } else {
                st4(p.out2 + (long long)m * p.U + (n - O), v);
            }
        }
        STAMP(6);
    }
}

template <typename AT, typename WT, int EPI, int MT, int KC>
int launch_nch(const RStep& p, dim3 grid, hipStream_t st) {
    const int nch = (p.K / 4) / KC;
    switch (nch) {
    case 1: hipLaunchKernelGGL((rstep_kernel<AT, WT, EPI, MT, KC, 1, 1>), grid, dim3(256), 0, st, p.A, p.W, p.lda, p.ldw, p.M, p.N, p.K, p.U, p.a_par, p.a_k > 0 ? p.a_k : 1, p); break;
    case 2: hipLaunchKernelGGL((rstep_kernel<AT, WT, EPI, MT, KC, 2, 2>), grid, dim3(256), 0, st, p.A, p.W, p.lda, p.ldw, p.M, p.N, p.K, p.U, p.a_par, p.a_k > 0 ? p.a_k : 1, p); break;
    case 4: hipLaunchKernelGGL((rstep_kernel<AT, WT, EPI, MT, KC, 4, 4>), grid, dim3(256), 0, st, p.A, p.W, p.lda, p.ldw, p.M, p.N, p.K, p.U, p.a_par, p.a_k > 0 ? p.a_k : 1, p); break;
    case 8: hipLaunchKernelGGL((rstep_kernel<AT, WT, EPI, MT, KC, 8, 4>), grid, dim3(256), 0, st, p.A, p.W, p.lda, p.ldw, p.M, p.N, p.K, p.U, p.a_par, p.a_k > 0 ? p.a_k : 1, p); break;
    case 16: hipLaunchKernelGGL((rstep_kernel<AT, WT, EPI, MT, KC, 16, 4>), grid, dim3(256), 0, st, p.A, p.W, p.lda, p.ldw, p.M, p.N, p.K, p.U, p.a_par, p.a_k > 0 ? p.a_k : 1, p); break;
    default: return -2;                                     // K / (4 * KC) must be a power of two up to 16 (every shape validate() admits with U, O, E, C in {128, 256, 512})
    }
    return (int)hipGetLastError();
}
template <typename AT, typename WT, int EPI>
int launch_ch(const RStep& p, hipStream_t st) {
    constexpr bool BF = is_bf16<WT>::value;
    // chunk = 128-byte source rows (64 bf16 / 32 f32 k), up to four chunks in flight: a wave's whole share up to K = 1024 bf16
    // arrives in one burst and is consumed chunk by chunk as it lands; longer contractions re-issue into freed registers
    constexpr int K0 = (BF && sizeof(AT) == 2) ? 64 : 32;
    constexpr int KMIN = BF ? 32 : 16;                       // one MFMA k-step
    const int kq = p.K / 4;
    const int ncol = EPI == RS_LSTM_FWD ? p.U / 4 : cdiv(p.N, 16);
    // 32-row tiles when 64-row tiles would leave CUs idle (the A tile, re-read by every workgroup, is the bulk of the fetch)
    // ... and for the row counts of a beam step (B x beam = 128 .. 512 rows): 32-row tiles measured 11.0 against 12.5 us (LSTM, 320 rows), 5.25 against 5.65 (o
    // projection) -- the tall all-step GEMMs (T x B rows) keep their 64-row tiles
    const bool half = ((long long)ncol * cdiv(p.M, 64) <= 128 && p.M > 32) || (p.M > 64 && p.M <= 512);
    // ... and 16-row tiles when even 32-row tiles leave half the CUs idle (o projection, att_h, the LSTM backward: 64 workgroups):
    // a workgroup's fetch -- the length of these kernels -- shrinks with its A tile
    static int q16 = -1;
    if (q16 < 0) { const char* e = getenv("LXO_RSTEP_MT16"); q16 = (e && e[0] == '0') ? 0 : 1; }
    const bool quarter = q16 && (long long)ncol * cdiv(p.M, 32) <= 128 && p.M > 16;
    if (kq % K0 == 0) {
        if (quarter) return launch_nch<AT, WT, EPI, 16, K0>(p, dim3(ncol, cdiv(p.M, 16)), st);
        if (half) return launch_nch<AT, WT, EPI, 32, K0>(p, dim3(ncol, cdiv(p.M, 32)), st);
        return launch_nch<AT, WT, EPI, 64, K0>(p, dim3(ncol, cdiv(p.M, 64)), st);
    }
    if constexpr (K0 / 2 >= KMIN) {
        if (kq % (K0 / 2) == 0) return launch_nch<AT, WT, EPI, 64, K0 / 2>(p, dim3(ncol, cdiv(p.M, 64)), st);
    }
    return -2;
}

template <int EPI>
int launch_epi(int dt, bool a_bf16, const RStep& p, hipStream_t st) {
    if (dt == LXO_F32) return launch_ch<float, float, EPI>(p, st);
    return a_bf16 ? launch_ch<bf16_t, bf16_t, EPI>(p, st) : launch_ch<float, bf16_t, EPI>(p, st);
}

// o / h / h~ / ctx columns of a record row -> bf16 mirror (initial state, beam re-ordering)
__global__ __launch_bounds__(256) void mirror_kernel(const float* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd, int rows, int cols) {
    const int total = rows * (cols >> 2);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / (cols >> 2), c = (i - r * (cols >> 2)) << 2;
        float v[4];
        ld4(src + (long long)r * lds + c, v);
        st4b(dst + (long long)r * ldd + c, v);
    }
}

}  // namespace

static thread_local unsigned long long* g_dbg = nullptr;
static thread_local int g_dbg_epi = -1;
extern "C" int lxo_rstep_debug(unsigned long long* buf, int epi) { g_dbg = buf; g_dbg_epi = epi; return 0; }
int lxo_launch_rstep(int dt, int a_bf16, const RStep& p0, hipStream_t st) {
    RStep p = p0;
    p.dbg = (g_dbg && p.epi == g_dbg_epi) ? g_dbg : nullptr;
    if (p.M <= 0 || p.N <= 0) return 0;
    // a wave contracts K/4: whole MFMA steps (32 k bf16 / 16 k f32), 16-byte fragment loads
    if (p.K % 128 != 0 || p.lda % 8 != 0 || p.ldw % 8 != 0 || p.N % 4 != 0) return -2;
    if (p.epi == RS_LSTM_FWD && (p.U % 4 != 0 || p.N != 4 * p.U)) return -2;
    switch (p.epi) {
    case RS_PLAIN: return launch_epi<RS_PLAIN>(dt, a_bf16 != 0, p, st);
    case RS_TANH_O: return launch_epi<RS_TANH_O>(dt, a_bf16 != 0, p, st);
    case RS_LSTM_FWD: return launch_epi<RS_LSTM_FWD>(dt, a_bf16 != 0, p, st);
    case RS_LSTM_BWD: return launch_epi<RS_LSTM_BWD>(dt, a_bf16 != 0, p, st);
    case RS_CARRY: return launch_epi<RS_CARRY>(dt, a_bf16 != 0, p, st);
    default: return -2;
    }
}

int lxo_k_mirror(const float* src, int lds, void* dst, int ldd, int rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return 0;
    if (cols % 4 != 0 || lds % 4 != 0 || ldd % 4 != 0) return -2;
    int g = cdiv(rows * (cols >> 2), 256);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(mirror_kernel, dim3(g), dim3(256), 0, st, src, lds, (bf16_t*)dst, ldd, rows, cols);
    return (int)hipGetLastError();
}
