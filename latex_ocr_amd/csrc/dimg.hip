// d_img of the attention decoder as ONE batched MFMA GEMM.
//
// Round 1 built it in four passes over a 114 MB f32 tensor: a batched TN GEMM (alpha^T d_ctx), a read-modify-write for the
// mean gradient, a second GEMM accumulating d_att_img W^T into it, and the ReLU-mask / bf16 conversion of the encoder's
// backward pass: 340 us per step for 740 MB of traffic.  Here a workgroup owns a 128 x 128 tile of one sample and
// contracts BOTH products into the same accumulators: first over the time steps (reduction index strided in memory, so two
// consecutive steps are packed into bf16x2 dwords on the way to LDS, as in gemm_tn_kernel), then over the attention
// channels (K-contiguous operands, staged row-major); the mean gradient is added in the epilogue and the tile is stored
// once.  4 waves as 2 x 2, each 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 tiles; the next K-tile's loads are in flight
// under the current tile's MFMAs; every load is unconditional (clamped address, masked value).
#include "dimg.h"
#include "decoder_kernels.h"      // lxo_k_det_reduce

namespace {

typedef __attribute__((ext_vector_type(16))) float v16f;
LXO_DEV v16f mfma32(u32x4 a, u32x4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int MODE>       // 0: f32 tile; 1: ReLU-masked bf16 tile + bias-gradient column sums
__global__ __launch_bounds__(256) void dimg_fused_kernel(DimgArgs p) {
    constexpr int PITCH = 36;                                 // bf16 elements per LDS tile row (32 k + 4 pad): 72 bytes
    __shared__ __attribute__((aligned(16))) bf16_t As[128 * PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[128 * PITCH];
    __shared__ float dbs[128];
    constexpr int TP = 64 + 4;                                // f32 pitch of the epilogue's transposing tile
    __shared__ __attribute__((aligned(16))) float tile[MODE == 1 ? 4 : 1][MODE == 1 ? 32 * TP : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128, b = blockIdx.z;
    if (MODE == 1) { if (tid < 128) dbs[tid] = 0.f; }        // made visible by the main loop's barriers
    const float* __restrict__ al = p.alpha + (long long)b * p.Rp;
    const float* __restrict__ dc = p.dctx + (long long)b * p.HC;
    const bf16_t* __restrict__ da = p.datt + (long long)b * p.R * p.E;
    const bf16_t* __restrict__ W = p.W;

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- phase 1 staging: a (step pair, 8-column chunk) of alpha and of d_ctx per thread ----
    const int s_r = 2 * (lane & 15), s_c = (lane >> 4) + 4 * wave;
    const bool ai_ok = i0 + s_c * 8 < p.R, bj_ok = j0 + s_c * 8 < p.C;
    const float* a_col = al + (ai_ok ? i0 + s_c * 8 : 0);
    const float* b_col = dc + (bj_ok ? j0 + s_c * 8 : 0);
    unsigned pa[8], pb[8];
    auto gload1 = [&](int t0) {
        const int ta = t0 + s_r, tb = ta + 1;
        const bool ok0 = ta < p.T, ok1 = tb < p.T;
        const long long o0 = ok0 ? ta : 0, o1 = ok1 ? tb : 0;
        float a0[8], a1[8], b0[8], b1[8];
        load8(a_col + o0 * p.ld_alpha, a0); load8(a_col + o1 * p.ld_alpha, a1);
        load8(b_col + o0 * p.ld_dctx, b0); load8(b_col + o1 * p.ld_dctx, b1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pa[e] = pack_bf2((ok0 && ai_ok) ? a0[e] : 0.f, (ok1 && ai_ok) ? a1[e] : 0.f);
            pb[e] = pack_bf2((ok0 && bj_ok) ? b0[e] : 0.f, (ok1 && bj_ok) ? b1[e] : 0.f);
        }
    };
    // ---- phase 2 staging: two 16-byte pieces of d_att_img rows and of W rows per thread ----
    int q_row[2], q_ch[2]; bool qa_ok[2], qb_ok[2];
    const bf16_t* qa[2]; const bf16_t* qb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int id = tid + 256 * q;
        q_row[q] = id >> 2; q_ch[q] = id & 3;
        qa_ok[q] = i0 + q_row[q] < p.R; qb_ok[q] = j0 + q_row[q] < p.C;
        qa[q] = da + (long long)(qa_ok[q] ? i0 + q_row[q] : 0) * p.E + q_ch[q] * 8;
        qb[q] = W + (long long)(qb_ok[q] ? j0 + q_row[q] : 0) * p.ldw + q_ch[q] * 8;
    }
    u32x4 ra[2], rb[2];
    auto gload2 = [&](int e0) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32x4 va = *reinterpret_cast<const u32x4*>(qa[q] + e0), vb = *reinterpret_cast<const u32x4*>(qb[q] + e0);
            ra[q] = qa_ok[q] ? va : z; rb[q] = qb_ok[q] ? vb : z;
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[2], bfr[2];
            const int kof = ks * 16 + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16_t* pa_ = &As[(wi * 64 + i * 32 + (lane & 31)) * PITCH + kof];
                const u32x2 lo = *reinterpret_cast<const u32x2*>(pa_), hi = *reinterpret_cast<const u32x2*>(pa_ + 4);
                af[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                const bf16_t* pb_ = &Bs[(wj * 64 + i * 32 + (lane & 31)) * PITCH + kof];
                const u32x2 lo2 = *reinterpret_cast<const u32x2*>(pb_), hi2 = *reinterpret_cast<const u32x2*>(pb_ + 4);
                bfr[i] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
        }
    };

    const int nt1 = (p.T + 31) >> 5, nt2 = p.E >> 5;          // E % 32 == 0 (checked by the launcher)
    gload1(0);
    for (int kt = 0; kt < nt1; ++kt) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            *reinterpret_cast<unsigned*>(&As[(s_c * 8 + e) * PITCH + s_r]) = pa[e];
            *reinterpret_cast<unsigned*>(&Bs[(s_c * 8 + e) * PITCH + s_r]) = pb[e];
        }
        __syncthreads();
        if (kt + 1 < nt1) gload1((kt + 1) * 32); else gload2(0);
        mma();
        __syncthreads();
    }
    for (int kt = 0; kt < nt2; ++kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16_t* sa = &As[q_row[q] * PITCH + q_ch[q] * 8];
            bf16_t* sb = &Bs[q_row[q] * PITCH + q_ch[q] * 8];
            *reinterpret_cast<u32x2*>(sa) = u32x2{ra[q][0], ra[q][1]}; *reinterpret_cast<u32x2*>(sa + 4) = u32x2{ra[q][2], ra[q][3]};
            *reinterpret_cast<u32x2*>(sb) = u32x2{rb[q][0], rb[q][1]}; *reinterpret_cast<u32x2*>(sb + 4) = u32x2{rb[q][2], rb[q][3]};
        }
        __syncthreads();
        if (kt + 1 < nt2) gload2((kt + 1) * 32);
        mma();
        __syncthreads();
    }

    // ---- epilogue ----
    const float inv_r = 1.0f / (float)p.R;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = j0 + wj * 64 + j * 32 + (lane & 31);
            const bool c_ok = c < p.C;
            const float dm = p.dmean[(long long)b * p.C + (c_ok ? c : 0)] * inv_r;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wi * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (c_ok && row < p.R) p.out[((long long)b * p.R + row) * p.C + c] = acc[i][j][r] + dm;
                }
        }
    } else {
        // A lane holds one column of 16 rows; the masked bf16 tile leaves as 16-byte row pieces (8 lanes = one 128-byte
        // row segment, read-side of y6 the same), so each wave turns its tile through a private f32 LDS region, 32 rows at
        // a time.  No workgroup barrier: a wave only reads what it wrote.
        float* tw = tile[wave];
        const int tr = lane >> 3, tc = (lane & 7) * 8;                // read side: rows tr + 8k, columns tc .. tc + 7
        const int cbase = j0 + wj * 64 + tc;
        const bool cb_ok = cbase < p.C;                              // C % 8 == 0: the 8 columns are in or out together
        float dm[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = j0 + wj * 64 + j * 32 + (lane & 31);
            dm[j] = p.dmean[(long long)b * p.C + (c < p.C ? c : 0)] * inv_r;
        }
        float cs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = i0 + wi * 64 + i * 32;
            u32x4 ref[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                            // mask operands first: unconditional, clamped
                const int row = rbase + tr + 8 * k;
                ref[k] = *reinterpret_cast<const u32x4*>(p.y6 + ((long long)b * p.R + (row < p.R ? row : 0)) * p.C + (cb_ok ? cbase : 0));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TP + j * 32 + (lane & 31)] = acc[i][j][r] + dm[j];
            __builtin_amdgcn_wave_barrier();                         // the wave's 64 lanes exchange through tw (in-order LDS: no s_barrier)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = rbase + tr + 8 * k;
                const bool ok = cb_ok && row < p.R;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&tw[(tr + 8 * k) * TP + tc]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&tw[(tr + 8 * k) * TP + tc + 4]);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned w = ref[k][e];                    // two bf16 activations: positive iff sign clear and not zero
                    const bool p0 = (short)(w & 0xffffu) > 0, p1 = (int)w > 0 && (w >> 16) != 0;
                    v[2 * e] = (ok && p0) ? (e < 2 ? lo[2 * e] : hi[2 * e - 4]) : 0.f;
                    v[2 * e + 1] = (ok && p1) ? (e < 2 ? lo[2 * e + 1] : hi[2 * e - 3]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += v[e];
                if (ok) store8(p.dy6 + ((long long)b * p.R + row) * p.C + cbase, v);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // bias gradient: the 8 lanes of equal (lane & 7) hold the same 8 columns
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], 8); cs[e] += __shfl_xor(cs[e], 16); cs[e] += __shfl_xor(cs[e], 32);
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&dbs[wj * 64 + tc + e], cs[e]);
        }
        __syncthreads();
        // (inside the workgroup a column gets exactly two addends -- the two waves of its column half -- so the LDS sum does not depend on
        // their order; deterministic mode: the workgroup's sums go to ITS slot and the launcher adds the slots in order)
        if (tid < 128 && j0 + tid < p.C) {
            if (p.db_part) p.db_part[((long long)b * gridDim.y + blockIdx.y) * p.C + j0 + tid] = dbs[tid];
            else if (p.db) atomicAdd(&p.db[j0 + tid], dbs[tid]);
        }
    }
}

}  // namespace

int lxo_launch_dimg_fused(const DimgArgs& p, hipStream_t st) {
    if (p.E % 32 || p.C % 8 || p.Rp % 8 || p.T < 1) return -2;
    dim3 grid((p.C + 127) / 128, (p.R + 127) / 128, p.B);
    if (p.db_part && (size_t)grid.y * grid.z * p.C > p.db_part_floats) return -6;
    if (p.y6) hipLaunchKernelGGL((dimg_fused_kernel<1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dimg_fused_kernel<0>), grid, dim3(256), 0, st, p);
    if (p.y6 && p.db_part && p.db) return lxo_k_det_reduce(p.db_part, (int)(grid.y * grid.z), p.C, p.C, p.db, st);
    return (int)hipGetLastError();
}
