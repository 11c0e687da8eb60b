// Dense weight-gradient GEMM  C[I][J] += sum_m A[m][I] * B[m][J]  (bf16 operands, f32 atomics), gfx950.
//
// The reduction index m is the STRIDED one of both operands.  gemm_tn_kernel (gemm.hip) packs pairs of rows into bf16x2
// dwords in registers and scatters them into a K-major LDS tile: 32 four-byte LDS writes per thread and 64-row tile, which is
// what bounds it (230-330 TFLOP/s on the decoder's T*B-row reductions).  Here the tiles go to LDS as they lie in memory --
// [64 m][128 columns] rows of 256 bytes, by LDS-DMA, 16-byte chunks XOR-swizzled by (m & 15) -- and the MFMA operands come
// out of them with the transposing read ds_read_b64_tr_b16 (4 consecutive m of one column per lane), the way
// conv_wgrad_kernel reads its d_out block.  128 x 128 output tile, 4 waves as 2 x 2, each 64 x 64 = 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles; two LDS stages of 32 KB; the DMA of tile k+1 is in flight under the MFMAs of tile k.
// Rows beyond the range and 8-column chunks beyond I / J read as zeros (out-of-range offsets of the buffer resources).
#include "gemm.h"
#include "api_util.h"

namespace {

constexpr int TBR = 64;                              // reduction rows per LDS tile
constexpr int TTILE = TBR * 256;                     // bytes of one operand tile: 64 rows x 128 columns x 2
constexpr int TSTAGE = 2 * TTILE;                    // A tile + B tile

typedef __attribute__((ext_vector_type(4))) short v4s_t;
typedef __attribute__((address_space(3))) v4s_t* ltr_t;
LXO_DEV u32x2 tr_read(const char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(uintptr_t)p));
}

}  // namespace

HIP_DYNAMIC_SHARED(char, lxo_tntr_lds)

namespace {

__global__ __launch_bounds__(256) void gemm_tn_tr_kernel(GemmTN p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    // XCD-aware order: workgroup L runs on XCD L % 8, and the XCD L2s (4 MB each) do not share.  The kernel is bound by operand
    // re-reads (a 128 x 128 tile has 64 FLOP per staged byte), so the (split, tile) units are dealt to the XCDs in CONTIGUOUS runs:
    // the workgroups resident on one XCD work on the same reduction range and re-use each other's operand rows from its L2,
    // instead of every XCD pulling every row through the fabric (4x less fabric traffic for the 4-split LSTM-kernel gradient).
    const int tj_n = (p.J + 127) / 128, ti_n = (p.I + 127) / 128, tiles = tj_n * ti_n;
    const int units = tiles * p.nsplit, per_xcd = (units + 7) / 8;
    const int unit = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= units) return;
    const int split = unit / tiles, tile = unit - split * tiles;
    const int i0 = (tile / tj_n) * 128, j0 = (tile % tj_n) * 128;
    const int per = ((p.M + p.nsplit - 1) / p.nsplit + TBR - 1) / TBR * TBR;
    const int mbeg = split * per, mend = min(p.M, mbeg + per);
    // deterministic mode: this (split, tile) unit's 128 x 128 partial goes to ITS slot of the slab with plain stores (an empty range stores
    // zeros) and tn_slab_reduce_kernel adds the slots of a tile in split order
    float* const slab = p.det_slab ? p.det_slab + ((long long)split * tiles + tile) * (128 * 128) : nullptr;
    if (mbeg >= mend) {
        if (slab) for (int i = tid; i < 128 * 128 / 4; i += 256) reinterpret_cast<f32x4*>(slab)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(p.B);

    // DMA geometry: thread -> (row r = (tid >> 4) + 16 j, LDS slot dch = tid & 15 of the 256-byte row); slot dch of row r holds the
    // GLOBAL chunk dch ^ (r & 15) (a reader of global chunk c looks in slot c ^ (r & 15)).
    // Requests go through buffer resources that END at row `mend`: the per-lane byte offset (row r, chunk) is fixed for the whole
    // kernel, the tile's first row is the request's scalar offset, rows beyond the range are out of the buffer and chunks beyond I / J
    // carry an out-of-range offset -- both read as zeros (tools/blds_probe.hip).  The addresses used to be rebuilt per request
    // (64-bit multiplies, a select against a zero line: ~25 VALU instructions for each of the 8 requests of a tile that has only
    // 16 MFMAs per wave); the matrix pipe waits while a wave issues them.
    const int dch = tid & 15, drow = tid >> 4;
    const unsigned m0b = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lxo_tntr_lds);
    const lxo_rsrc_t ra = lxo_make_rsrc(A, (unsigned)((long long)mend * p.lda * 2));
    const lxo_rsrc_t rb = lxo_make_rsrc(B, (unsigned)((long long)mend * p.ldb * 2));
    unsigned voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = drow + 16 * j;
        const int gch = (dch ^ (r & 15)) << 3;                  // first column of the chunk inside the 128-column tile
        voa[j] = i0 + gch < p.I ? (unsigned)((r * p.lda + i0 + gch) * 2) : LXO_BLDS_OOB;
        vob[j] = j0 + gch < p.J ? (unsigned)((r * p.ldb + j0 + gch) * 2) : LXO_BLDS_OOB;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int m0, int stage) {
        const unsigned sa = __builtin_amdgcn_readfirstlane((unsigned)(m0 * p.lda * 2)), sb = __builtin_amdgcn_readfirstlane((unsigned)(m0 * p.ldb * 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            LXO_BLDS16(voa[j], ra, sa, lxo_tntr_lds, m0b, stage * TSTAGE + (wave_u * 64 + 256 * j) * 16);
            LXO_BLDS16(vob[j], rb, sb, lxo_tntr_lds, m0b, stage * TSTAGE + TTILE + (wave_u * 64 + 256 * j) * 16);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // lane geometry of the transposing reads (as in conv_wgrad_kernel): half h = lane >> 5 takes k 8h .. 8h+7 of a 16-row K-step,
    // group g = (lane >> 4) & 1 the upper 16 of the 32 columns, qj = (lane & 15) >> 2 the row inside a group of 4, qc = lane & 3
    // the 4-column piece.  Two reads (rows bl and bl + 4) give the lane its 8 k values of ONE column: the MFMA operand.
    const int h = lane >> 5, g = (lane >> 4) & 1, qj = (lane & 15) >> 2, qc = lane & 3;
    const int bl = 8 * h + qj;
    int offa[2][2], offb[2][2];                                 // [sub-tile][row bl / bl + 4], bytes inside an operand tile
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ca = wi * 64 + t * 32 + 16 * g + 4 * qc, cb = wj * 64 + t * 32 + 16 * g + 4 * qc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = bl + 4 * u;                           // K-steps start at multiples of 16 rows: (row & 15) == r
            offa[t][u] = r * 256 + (((ca >> 3) ^ (r & 15)) << 4) + (ca & 7) * 2;
            offb[t][u] = TTILE + r * 256 + (((cb >> 3) ^ (r & 15)) << 4) + (cb & 7) * 2;
        }
    }

    issue(mbeg, 0);
    for (int m0 = mbeg, it = 0; m0 < mend; m0 += TBR, ++it) {
        const int stage = it & 1;
        __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) + lgkmcnt(0): every ds_read of the stage the NEXT DMA overwrites has retired before any wave passes the barrier (gfx950 barriers carry no implicit wait); only this tile's DMA is outstanding here
        __builtin_amdgcn_s_barrier();
        if (m0 + TBR < mend) issue(m0 + TBR, stage ^ 1);
        const char* sb = lxo_tntr_lds + stage * TSTAGE;
#pragma unroll
        for (int ks = 0; ks < TBR / 16; ++ks) {
            const int cd = ks * 16 * 256;
            u32x4 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x2 a0 = tr_read(sb + offa[t][0] + cd), a1 = tr_read(sb + offa[t][1] + cd);
                const u32x2 b0 = tr_read(sb + offb[t][0] + cd), b1 = tr_read(sb + offb[t][1] + cd);
                af[t] = u32x4{a0[0], a0[1], a1[0], a1[1]};
                bf[t] = u32x4{b0[0], b0[1], b1[0], b1[1]};
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bf[b]), acc[a][b], 0, 0, 0);
        }
    }
    // acc[a][b][e]: row i = i0 + wi*64 + a*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column j = j0 + wj*64 + b*32 + (lane & 31)
    if (slab) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    slab[(wi * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 128 + wj * 64 + b * 32 + (lane & 31)] = acc[a][b][e];
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int jj = j0 + wj * 64 + b * 32 + (lane & 31);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ii = i0 + wi * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (ii < p.I && jj < p.J) atomicAdd(&p.C[(long long)ii * p.ldc + jj], acc[a][b][e]);
            }
    }
}

// deterministic mode: C[i0 + r][j0 + c] += sum over the splits s = 0 .. nsplit-1 (in that order) of slab[s][tile][r][c]
__global__ __launch_bounds__(256) void tn_slab_reduce_kernel(const float* __restrict__ slab, int nsplit, int tiles, int tj_n, int I, int J, float* __restrict__ C, int ldc) {
    const int tile = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;                 // element of the 128 x 128 tile
    const int r = i >> 7, c = i & 127;
    const int ii = (tile / tj_n) * 128 + r, jj = (tile % tj_n) * 128 + c;
    float s = 0.f;
    for (int q = 0; q < nsplit; ++q) s += slab[((long long)q * tiles + tile) * (128 * 128) + i];
    if (ii < I && jj < J) C[(long long)ii * ldc + jj] += s;
}

}  // namespace

// bf16 A and B, K-strided; lda, ldb multiples of 8 and 16-byte aligned bases; atomic accumulation into C
int lxo_launch_gemm_tn_tr(const GemmTN& p, hipStream_t s) {
    // a ragged last 8-column chunk is read whole: it must lie inside the row (padded pitch), its surplus columns are never stored
    if (p.conv || !p.atomic || p.nbatch != 1 || p.lda % 8 || p.ldb % 8 || (p.I + 7) / 8 * 8 > p.lda || (p.J + 7) / 8 * 8 > p.ldb) return -2;
    if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return -2;
    if ((long long)p.M * p.lda * 2 >= (1LL << 31) || (long long)p.M * p.ldb * 2 >= (1LL << 31)) return -2;   // 32-bit buffer offsets
    {   // per device, not per process (see conv_igemm.hip attr_needed)
        static bool done[64] = {};
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (!known || !done[dev]) {
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TSTAGE));
            if (known) done[dev] = true;
        }
    }
    const int tiles = ((p.J + 127) / 128) * ((p.I + 127) / 128);
    const int units = tiles * p.nsplit;
    if (p.det_slab && (size_t)units * 128 * 128 > p.det_floats) return -2;      // (the caller falls back to one row range per tile)
    dim3 grid(8 * ((units + 7) / 8));
    hipLaunchKernelGGL(gemm_tn_tr_kernel, grid, dim3(256), 2 * TSTAGE, s, p);
    if (p.det_slab)
        hipLaunchKernelGGL(tn_slab_reduce_kernel, dim3(128 * 128 / 256, tiles), dim3(256), 0, s, p.det_slab, p.nsplit, tiles, (p.J + 127) / 128, p.I, p.J, p.C, p.ldc);
    return (int)hipGetLastError();
}
