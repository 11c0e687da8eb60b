// 3x3 convolution (forward and data gradient) as an implicit GEMM on the bf16 MFMA units, gfx950.
//
//   conv_halo2wg_kernel (the kernel every layer of the model runs on): an 8 x 32 pixel halo tile x 128 (or 64) output channels, four
//   waves, 77 KB of LDS so that two workgroups share a CU; the (8+2) x (32+2) input patch of a 32-channel slice is staged ONCE and
//   serves all nine taps; patches by LDS-DMA through a buffer resource, private per-wave weight rings, fused epilogues (bias, ReLU,
//   max pool + routing mask, ReLU mask + bias-gradient sums, timing signal).  Described at its definition below.
//   conv_halo_kernel: the general kernel for what that one does not take (Cout % 64 != 0, tensors of 2 GB and more): a 4 x 64 pixel
//   halo tile x 128 channels, eight waves, LDS-DMA with XOR-swizzled 128-byte rows, three weight stages.
//   im2col is implicit: a K-step lies inside one (kh, kw) tap because Cin % 64 == 0, so an A row chunk is 16 contiguous bytes of
//   the NHWC input, or zeros for padding taps; the blockIdx -> tile remap keeps the tiles that share an input panel on one XCD.
//   (Rounds 1-2 also carried a 256 x 128 x 64 im2col-tile kernel and a 256-channel-wide halo kernel: superseded, removed in round 4.)
#include "gemm.h"
#include "api_util.h"
#include "decoder_kernels.h"      // lxo_k_det_reduce
#include "lxo_debug.h"
#include <stdlib.h>

namespace {

__device__ unsigned lxo_zero_line[8] = {0, 0, 0, 0, 0, 0, 0, 0};

constexpr int CBM = 256, CBN = 128, CBK = 64, CTH = 512;
constexpr int A_STAGE = CBM * CBK * 2, B_STAGE = CBN * CBK * 2, STAGE = A_STAGE + B_STAGE;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
LXO_DEV void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)g, (lptr_t)(uintptr_t)l, 16, 0, 0);
}

}  // namespace

HIP_DYNAMIC_SHARED(char, lxo_conv_lds)

namespace {

// ------------------------------------------------------------------------------------------
// Halo-tiled variant: a workgroup owns a 4 x 64 block of output pixels (x 128 channels).  For
// each 64-channel slice of the input it stages the (4+2) x (64+2) input patch ONCE and serves
// all nine taps from it (an A fragment row is just the patch pixel shifted by the tap), so the
// activation operand crosses L2->LDS 1.5x instead of 9x; only the weight tiles stream per
// K-step.  LDS: two patch buffers (7 LDS-DMA slots of 16 B per thread each = 3584 slots) and
// three weight stages.
constexpr int HTH = 4, HTW = 64, HPW = HTW + 2, HPH = HTH + 2, HPROWS = HPH * HPW;   // 396 patch pixels
constexpr int HPSLOTS = 7 * CTH;                                                       // 3584 >= 396 * 8
constexpr int HPATCH = HPSLOTS * 16, HB_STAGE = CBN * CBK * 2;                         // 57344, 16384

#ifdef LXO_DIAG      // measurement builds only (make EXTRA=-DLXO_DIAG=1): parts of the kernel's work can be switched off -- WRONG results by design
#define LXO_DIAG_BIT(b) (p.diag & (b))
#else
#define LXO_DIAG_BIT(b) false
#endif
#define LXO_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also carries a
// workgroup-scope fence, which on gfx950 is `s_waitcnt vmcnt(0)`: in the epilogue that made every pass wait for the
// acknowledgement of the previous pass's global STORES (in-kernel stamps: 32 k cycles of epilogue per tile, a quarter of a
// conv4 tile and more than half of a conv2 tile).  The epilogue's barriers only protect the f32 staging tile in LDS.
#define LXO_LDS_BARRIER() do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); } while (0)

// NJ = 32-channel blocks per wave: 2 -> 128-channel tiles, 1 -> 64-channel tiles (layers with Cout <= 64: conv2's dgrad)
template <typename OT, int NJ>
__global__ __launch_bounds__(512) void conv_halo_kernel(GemmNT p, int tiles_n, int tiles_x, int tiles_y) {
    constexpr int BN = 64 * NJ, HBS = BN * CBK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;
    const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
    const int tx_i = mt % tiles_x, ty_i = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int oy0 = ty_i * HTH, ox0 = tx_i * HTW, n0 = nt * BN;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ Bp = reinterpret_cast<const bf16_t*>(p.Bp);
    const char* zline = reinterpret_cast<const char*>(lxo_zero_line);
    char* patch0 = lxo_conv_lds;
    char* bst0 = lxo_conv_lds + 2 * HPATCH;

    // patch staging: slot s = tid + 512 j -> patch pixel prow = (tid >> 3) + 64 j, LDS chunk tid & 7
    const int sch = tid & 7;
    long long a_off[7];                      // element offset of the pixel in the input, < 0 = zero (padding / outside)
    int a_gch[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int prow = (tid >> 3) + 64 * j;
        const int py = prow / HPW, px = prow - py * HPW;
        const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
        const bool ok = prow < HPROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        a_off[j] = ok ? (((long long)b * p.H + iy) * p.W + ix) * p.Cin : -1;
        a_gch[j] = (sch ^ ((prow >> 1) & 7)) << 3;
    }
    const int srow = tid >> 3;
    const bf16_t* b_ptr[NJ]; bool b_ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + srow + 64 * j;
        b_ok[j] = n < p.N;
        b_ptr[j] = Bp + (long long)(b_ok[j] ? n : 0) * p.ldb + ((sch ^ ((srow >> 1) & 7)) << 3);
    }
    auto issue_patch = [&](int c, int buf) {          // 7 LDS-DMA per thread
        char* dst = patch0 + buf * HPATCH;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const void* src = a_off[j] >= 0 ? (const void*)(A + a_off[j] + c * CBK + a_gch[j]) : (const void*)zline;
            glds16(src, dst + (wave * 64 + 512 * j) * 16);
        }
    };
    auto issue_b = [&](int t, int stage) {             // NJ LDS-DMA per thread; K index of tile t = tap*Cin + c*64
        const int c = t / 9, tap = t - 9 * c;
        const int k0 = tap * p.Cin + c * CBK;
        char* dst = bst0 + stage * HBS;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const void* src = b_ok[j] ? (const void*)(b_ptr[j] + k0) : (const void*)zline;
            glds16(src, dst + (wave * 64 + 512 * j) * 16);
        }
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // A-fragment geometry of this lane: tile pixel r -> (ty, tx); patch row for tap (kh,kw) = (ty+kh)*66 + tx + kw
    int a_prow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = wm * 64 + i * 32 + (lane & 31);
        a_prow[i] = (rr >> 6) * HPW + (rr & 63);
    }
    int b_row[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_row[j] = wn * 32 * NJ + j * 32 + (lane & 31);

    const int nchunk = p.Cin / CBK, nk = nchunk * 9;
    issue_patch(0, 0);
    issue_b(0, 0);
    if (nk > 1) issue_b(1, 1);
    for (int t = 0; t < nk; ++t) {
        // loads issued after B(t): B(t+1) [2] and the next patch [7] when it was issued at step t-2 or t-1
        const int tm = t % 9;
        const bool patch_after = (tm == 5 || tm == 6) && (t / 9 + 1 < nchunk);
        const int nafter = ((t + 1 < nk) ? NJ : 0) + (patch_after ? 7 : 0);
        if (nafter == NJ + 7) LXO_VMCNT(NJ + 7);
        else if (nafter == 7) LXO_VMCNT(7);
        else if (nafter == NJ) LXO_VMCNT(NJ);
        else LXO_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nk) issue_b(t + 2, (t + 2) % 3);
        if (tm == 4 && t / 9 + 1 < nchunk) issue_patch(t / 9 + 1, (t / 9 + 1) & 1);
        const int c = t / 9, tap = t - 9 * c;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const char* ps = patch0 + (c & 1) * HPATCH;
        const char* bs = bst0 + (t % 3) * HBS;
        const int shift = kh * HPW + kw;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = ks * 2 + (lane >> 5);
            u32x4 af[2], bfr[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int prow = a_prow[i] + shift;
                af[i] = *reinterpret_cast<const u32x4*>(ps + prow * 128 + ((kc ^ ((prow >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bfr[j] = *reinterpret_cast<const u32x4*>(bs + b_row[j] * 128 + ((kc ^ ((b_row[j] >> 1) & 7)) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: tile pixel r -> output row ((b*Ho + oy0 + r/64) * Wo + ox0 + r%64) ----
    OT* __restrict__ C = reinterpret_cast<OT*>(p.C);
    const bool plain = !p.out_pre && !p.addend && !p.relu_ref && !p.colsum && !p.accumulate && p.ldc == p.N && (p.N & 7) == 0;
    if (plain) {
        LXO_LDS_BARRIER();
        bf16_t* ot = reinterpret_cast<bf16_t*>(lxo_conv_lds);
        constexpr int OP = BN + 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nl = wn * 32 * NJ + j * 32 + (lane & 31);
            const int n = n0 + nl;
            const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    float v = p.alpha * acc[i][j][e] + bias;
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    else if (p.act == 2) v = tanhf(v);
                    ot[ml * OP + nl] = f2bf(v);
                }
        }
        LXO_LDS_BARRIER();
#pragma unroll
        for (int it = 0; it < 4 * NJ; ++it) {
            const int idx = tid + 512 * it, row = idx / (BN / 8), c8 = (idx % (BN / 8)) * 8;
            const int oy = oy0 + (row >> 6), ox = ox0 + (row & 63), n = n0 + c8;
            if (oy < p.Ho && ox < p.Wo && n < p.N)
                *reinterpret_cast<u32x4*>(C + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.ldc + n) = *reinterpret_cast<const u32x4*>(ot + row * OP + c8);
        }
        return;
    }
    OT* __restrict__ Cpre = reinterpret_cast<OT*>(p.out_pre);
    const bf16_t* __restrict__ ref = reinterpret_cast<const bf16_t*>(p.relu_ref);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * 32 * NJ + j * 32 + (lane & 31);
        const bool n_ok = n < p.N;
        const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
        float csum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int oy = oy0 + (ml >> 6), ox = ox0 + (ml & 63);
                if (!(n_ok && oy < p.Ho && ox < p.Wo)) continue;
                const long long m = ((long long)b * p.Ho + oy) * p.Wo + ox;
                float v = p.alpha * acc[i][j][e] + bias;
                if (p.act == 1) v = fmaxf(v, 0.f);
                else if (p.act == 2) v = tanhf(v);
                const long long o = m * p.ldc + n;
                if (Cpre) Cpre[o] = from_f32<OT>(v);
                if (p.addend) v += p.addend[(m % p.addend_rows) * p.N + n];
                if (ref) v = (to_f32(ref[m * p.ldr + n]) > 0.f) ? v : 0.f;
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

// tile geometry of the two-workgroup kernel below: an 8 x 32 block of output pixels, (8+2) x (32+2) patch
constexpr int QTH = 8, QTW = 32, QPW = QTW + 2, QPH = QTH + 2, QPROWS = QPH * QPW;   // 340 patch pixels

// ------------------------------------------------------------------------------------------
// Two-workgroups-per-CU variant (the default for layers with Cout % 64 == 0; LXO_CONV_2WG=0 falls back): an 8 x 32 pixel halo
// tile x 128 (or 64) channels, FOUR waves, 72 KB of LDS, so that two workgroups share a CU and the prologue and the epilogue
// of one run under the MFMAs of the other.
//
// Round 3, step 1: NO workgroup barrier inside a channel slice.  In-kernel stamps of the round-2 loop (weight stages shared by
// the four waves: `s_waitcnt vmcnt(0)` + `s_barrier` + the next stage's DMA issue + the first fragment reads at the head of
// every K-step) showed 2.6 k cycles per K-step against 2 x 1.02 k of MFMA issue for the two waves of a SIMD: the two
// co-resident workgroups fall into lockstep (the one that lags catches up while the leader sits in its bubble), so the
// ~550-cycle head of a K-step is paid in full, 36 times per tile.  Now a wave owns one 32-CHANNEL block for ALL 8 tile rows
// (8 MFMA accumulators of 32 pixels x 32 channels): its weight operand is PRIVATE -- DMA'd by the wave itself into its own
// LDS stages, completion counted with `s_waitcnt vmcnt(n)`, no other wave involved -- and the patch is read-only for the 9
// taps of a slice.  The stream of a wave is fully software-pipelined: one pixel-fragment buffer (fragment i of the next
// sub-step is requested right behind the MFMA that consumed fragment i), the first fragments of tap t+1 are read under the
// last MFMAs of tap t.  Measured: K-steps 2.6 k -> 2.1 k cycles, but the single-buffered patch reload (barrier, 11 DMA per
// thread, landing, barrier, first fragments) then stood out at 4 k cycles per slice, 12 % of a conv4 tile.
// Step 2: 32-channel slices.  The patch of a slice is 27 KB, so TWO fit: slice c+1 lands (one 4 KB piece per tap) while
// slice c is computed, and the only barrier left is one per slice (144 MFMAs per wave) with nothing to wait for but the other
// three waves.  A wave's weight stage shrinks to 2 KB (32 channels x 32 k), three of them make a ring that keeps two taps of
// requests in flight.
// Step 3: an instruction diet.  A per-CU timeline of the stamps (tools/conv_timeline.py) showed the OLDER of the two
// workgroups of a CU running a slice in 10.0 k cycles and the younger in 16.4 k against 4.6 k of MFMA issue each: 74 % of the
// matrix pipe, and diagnostic builds without the weight DMA / the patch DMA / the pixel-fragment reads each gave a piece of
// it back (8.4 k / 9.2 k / 5.65 k with all three removed).  tools/issue_probe.hip prices the non-matrix instructions: on a
// SIMD shared by two waves every ds_read_b128 costs ~6 and every LDS-DMA request ~58 cycles of MATRIX-pipe time, and the
// per-tap address arithmetic (5 VALU per fragment offset for the XOR swizzle, an XOR per second sub-step read, 64-bit adds
// and a v_readfirstlane per DMA) came to ~5 VALU per MFMA -- VALU and MFMA share one issue port.  So:
//   * patch pixels are 80 B apart in LDS (64 B of channels + 16 B unused): 16 consecutive pixels x 80 B hit 16 distinct
//     16-byte slots of the 256-byte bank row (5 is odd), no XOR needed, and EVERY fragment address of a slice is ONE per-lane
//     register + a compile-time immediate ((row + kh) * 34 + kw) * 80 + 32 ks: zero VALU per read, no offset registers.
//     LDS-DMA writes lane-linearly, so the layout is produced on the source side: slot q = 5 pixel + chunk, every fifth
//     slot is a dummy fetch;
//   * weight requests use the scalar-base form (global_load_lds voff, s[base]): the tile base is wave-uniform SALU
//     arithmetic, the per-lane byte offsets are two loop-invariant registers, M0 = a scalar + immediate;
//   * the weight fragment addresses are two loop-invariant registers (+ immediate ring stage).
// Measured and rejected: a PERSISTENT grid (two workgroups per CU walking the tiles; the per-CU timeline shows ~3.6 k idle
// cycles between the end of a workgroup and the start of its successor on the same CU slot): conv roofline 0.506 vs 0.564.
// gfx950 counts stores in vmcnt, in order, so the first DMA wait of the next tile also waits for the acknowledgement of the
// previous tile's 64 KB of output stores (gap 8-9 k cycles, epilogue 9.6 k instead of 5.8 k), which a workgroup that simply
// ends never pays; the loop-carried state also pushed the kernel into scratch (34-58 spill instructions per tile = +65 MB
// of HBM writes per launch in the PMC pass).
// 9 ds_read_b128 per 8 MFMAs (8 pixel fragments + 1 weight fragment).  With 64-channel tiles (conv2's data gradient) the four
// waves are 2 channel blocks x 2 row halves (4 accumulators each); the two waves of a channel block each keep their own
// copy of the weights.
constexpr int WTHR = 256;
constexpr int WKC = 32;                                               // channels per slice
constexpr int WPIX = 80;                                              // LDS bytes per patch pixel: 4 chunks of 8 channels + 1 unused
constexpr int WPUNITS = QPROWS * 5;                                   // 1700 16-byte slots per patch
constexpr int WPDMA = 7, WPATCHB = WPUNITS * 16;                      // 7 LDS-DMA per thread (the last one partial): 27200 B per patch
constexpr int WWSTAGE = 32 * WKC * 2, WNST = 3;                       // a wave's weight stage: 32 channels x 32 k = 2048 B; ring of 3
constexpr int WLDS = 2 * WPATCHB + 4 * WNST * WWSTAGE;                // 78976 (the epilogues need at most 69632)

// NJ = 32-channel blocks per tile: 4 -> 128-channel tiles, 2 -> 64-channel tiles (Cout = 64: conv2's dgrad)
// EPI = which fused epilogue is compiled in: 0 bias + activation only; 1 + pre-addend copy + f32 addend (conv6 forward: timing
// signal); 2 + ReLU mask + bias-gradient column sums (conv4 data gradient); 3 everything, decided at run time.  With the
// optional operands behind run-time branches hipcc put `s_waitcnt vmcnt(0)` into every row iteration (it cannot count loads
// it may or may not have issued), so each 16-byte store waited for the previous one's acknowledgement: in-kernel stamps
// showed 6.3 k cycles per 64-row pass, 32 k per tile -- a quarter of a conv4 tile, more than half of a conv2 tile.
template <int NJ, int EPI, int PH = 1, int PW = 1>
__global__ __launch_bounds__(256, 2) void conv_halo2wg_kernel(GemmNT p, int tiles_n, int tiles_x, int tiles_y) {
    constexpr int WBN = 32 * NJ;
    constexpr int NWM = 4 / NJ, RI = QTH / NWM;                    // waves along the tile rows (1 or 2), tile rows per wave (8 or 4)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: scalar branches and SALU addressing
    const int wn = wave % NJ, wm = wave / NJ;                      // this wave's channel block / row group
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;
    const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
    const int tx_i = mt % tiles_x, ty_i = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int oy0 = ty_i * QTH, ox0 = tx_i * QTW, n0 = nt * WBN;
    const bf16_t* __restrict__ Bp = reinterpret_cast<const bf16_t*>(p.Bp);
    const unsigned wst_off = 2 * WPATCHB + wave * WNST * WWSTAGE;  // this wave's private weight ring
    const unsigned m0base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lxo_conv_lds);

    // patch slot s = tid + 256 j holds chunk s % 5 (4 = unused) of pixel s / 5; slots >= WPUNITS do not exist (last piece only).
    // The requests go through a buffer resource over the whole input tensor: the per-lane part of the address (pixel, chunk) is a
    // 32-bit offset computed once per tile, the per-slice part (image, channel slice) is the request's scalar offset, and a pixel
    // of the zero padding is an out-of-range offset (the buffer writes zeros: tools/blds_probe.hip) -- no VALU work per request,
    // and the request itself costs about half the issue time of the 64-bit-address form (tools/issue_probe.hip).
    const lxo_rsrc_t rsA = lxo_make_rsrc(p.A, (unsigned)((long long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * p.Cin * 2));
    const unsigned soffA = __builtin_amdgcn_readfirstlane((unsigned)(b * p.H * p.W * p.Cin * 2));
    unsigned a_src[WPDMA];
#pragma unroll
    for (int j = 0; j < WPDMA; ++j) {
        const int s = tid + WTHR * j;
        const int prow = s / 5, ch = s - 5 * prow;
        const int py = prow / QPW, px = prow - py * QPW;
        const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
        const bool ok = ch < 4 && prow < QPROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        a_src[j] = ok ? (unsigned)(((iy * p.W + ix) * p.Cin + (ch << 3)) * 2) : LXO_BLDS_OOB;
    }
    // weight rows n0 + wn*32 + r, r = 16 j + (lane >> 2) (1 KB = 16 rows per instruction), chunk swizzle ((r >> 2) & 3):
    // per-lane BYTE offsets from the wave-uniform tile base
    unsigned w_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * j + (lane >> 2);
        w_src[j] = (unsigned)(((n0 + wn * 32 + r) * p.ldb + (((lane & 3) ^ ((r >> 2) & 3)) << 3)) * 2);
    }
    auto issue_patch_piece = [&](int c, int buf, int j) {          // piece j (4 KB per workgroup) of slice c into patch buffer buf
        if (LXO_DIAG_BIT(2)) c = 0;
        const unsigned so = soffA + (unsigned)(c * WKC * 2);
        if (j + 1 < WPDMA) LXO_BLDS16(a_src[j], rsA, so, lxo_conv_lds, m0base, buf * WPATCHB + wave * 1024 + 4096 * j);
        else if (wave * 64 + WTHR * j < WPUNITS) {                 // the partial piece: wave 3 has no slot in it, wave 2 a part of its lanes
            if (tid + WTHR * j < WPUNITS) LXO_BLDS16(a_src[j], rsA, so, lxo_conv_lds, m0base, buf * WPATCHB + wave * 1024 + 4096 * j);
        }
    };
    auto issue_patch = [&](int c, int buf) {
#pragma unroll
        for (int j = 0; j < WPDMA; ++j) issue_patch_piece(c, buf, j);
    };
    auto issue_w = [&](int c, int tap, int stage) {                // 2 LDS-DMA per lane: tap `tap` of slice c into this wave's `stage`
        const bf16_t* sb = LXO_DIAG_BIT(1) ? Bp : Bp + tap * p.Cin + c * WKC;
#pragma unroll
        for (int j = 0; j < 2; ++j) LXO_GLDS16_SADDR(w_src[j], sb, lxo_conv_lds, m0base, wst_off + stage * WWSTAGE + 1024 * j);
    };

    const int khalf = lane >> 5;

    // per-lane LDS byte offsets: pixel fragments = a_lane (+ the patch buffer of the slice) + immediate; weight fragments of
    // sub-step ks = w_lane[ks] + immediate ring stage
    const int a_lane0 = ((wm * RI) * QPW + (lane & 31)) * WPIX + khalf * 16;
    int w_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        w_lane[ks] = (int)wst_off + (lane & 31) * 64 + (((2 * ks + khalf) ^ (((lane & 31) >> 2) & 3)) << 4);

    const int nchunk = p.Cin / WKC;
    // A tile whose last two rows lie below the image (conv6 forward: 14 output rows in two 8-row tiles) skips their MFMAs and fragment
    // reads: 12.5 % of that launch's matrix work.  Wave-uniform (scalar branch); the accumulators of those rows keep the bias and are
    // masked by every epilogue like any row >= Ho.
    const bool short_tile = __builtin_amdgcn_readfirstlane((RI == 8 && oy0 + 6 >= p.Ho) ? 1 : 0) != 0;
    // (Round 6 tried skipping the MFMAs whose pixel rows are zero padding -- the pad-2 data gradient of the VALID conv6 has 12.5 % of them -- behind a wave-uniform
    // mask, a scalar branch per MFMA: compiled into every instantiation it cost the step 1.1 % (conv fraction 0.534-0.54 against 0.55-0.566, two builds on one
    // box; a run-time switch had not shown it, the branches stay), in an instantiation of its own it bought that one launch 1 us of 190.  Not in.)
#define CSTAMP(i) do { if (p.dbg && tid == 0 && (i) < 64) p.dbg[(long long)bid * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
    CSTAMP(0);
    if (p.dbg && tid == 0) p.dbg[(long long)bid * 64 + 63] = __builtin_amdgcn_s_getreg(63492);      // HW_ID: which CU / SIMD this workgroup landed on
    issue_patch(0, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);

    // The accumulators START as the bias (alpha == 1 for every convolution): its loads are issued BEHIND the first patch /
    // weight DMA requests (in front of them the accumulator initialisation made every tile wait a memory round trip before its
    // first request), and no epilogue has to fetch per-lane bias values with the MFMA results waiting (5.6 k cycles in the stamps).
    // Layout (MFMA operands swapped, see the K loop): acc[i][e] -> tile row wm*RI + i, column lane & 31,
    // channel wn*32 + 8*(e>>2) + 4*(lane>>5) + (e&3).
    f32x16 acc[RI];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // unconditional load (a null bias reads the zero line): a load behind a branch makes hipcc wait for it on the spot
        const float* bsrc = p.bias ? p.bias + n0 + wn * 32 + 8 * k + 4 * khalf : reinterpret_cast<const float*>(lxo_zero_line);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bsrc);
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][4 * k + e] = bq[e];
    }


    u32x4 af[RI], bfr[2];
    int a_lane = a_lane0;
    auto fill = [&]() {                                             // first fragments of tap 0 of a slice (weight stage 0)
        bfr[0] = *reinterpret_cast<const u32x4*>(lxo_conv_lds + w_lane[0]);
#pragma unroll
        for (int i = 0; i < RI; ++i) af[i] = *reinterpret_cast<const u32x4*>(lxo_conv_lds + a_lane + i * QPW * WPIX);
    };

    LXO_VMCNT(0);
    __builtin_amdgcn_s_barrier();                                   // the patch is everybody's
    fill();
    for (int c = 0; c < nchunk; ++c) {
        const bool last = c + 1 == nchunk;
        CSTAMP(1 + c);
        // K-step t = 9 c + tap reads weight stage t % 3 = tap % 3
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            constexpr int NP = WPDMA;
            const int kh = tap / 3, kw = tap % 3;
            // sub-step 0: the fragments of sub-step 1 are requested behind the MFMAs that consume their registers
            bfr[1] = *reinterpret_cast<const u32x4*>(lxo_conv_lds + w_lane[1] + (tap % 3) * WWSTAGE);
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                if (RI == 8 && i >= 6 && short_tile) continue;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bfr[0]),
                                                                 __builtin_bit_cast(bf16x8_t, af[i]), acc[i], 0, 0, 0);
                af[i] = *reinterpret_cast<const u32x4*>(lxo_conv_lds + a_lane + ((i + kh) * QPW + kw) * WPIX + 32);
                __builtin_amdgcn_sched_barrier(0);
            }
            // sub-step 1: behind its first MFMA (whose operands prove that every fragment of THIS tap's stage has arrived) the
            // stage is refilled with the weights three taps ahead: the ring keeps two taps (~2 k cycles) of requests in
            // flight.  Late in the sub-step: make sure the NEXT tap's stage has landed and read its first weight fragment.
            const int t3 = tap + 3 < 9 ? tap + 3 : tap + 3 - 9;     // three taps ahead: same slice, or the next one
            const int kh1 = (tap + 1) / 3, kw1 = (tap + 1) % 3;
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                if (RI == 8 && i >= 6 && short_tile) continue;      // (the bookkeeping below sits at i == 0 and i == RI - 3)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bfr[1]),
                                                                 __builtin_bit_cast(bf16x8_t, af[i]), acc[i], 0, 0, 0);
                if (i == 0 && (tap + 3 < 9 || !last)) issue_w(tap + 3 < 9 ? c : c + 1, t3, tap % 3);
                if (i == RI - 3) {
                    // Requests newer than the next tap's weights (issued two taps ago): two taps of weights (none / one near
                    // the end of the last slice) and the patch pieces issued in the last two taps.  The patch of the NEXT
                    // slice is requested one piece per tap (taps 0..6), each right behind this wait: `vmcnt` counts in
                    // order, so a piece gets three taps to land before a wait needs it.  Wave 3 has no share of piece 6.
                    const int npa = ((tap - 2 >= 0 && tap - 2 < NP) ? 1 : 0) + ((tap - 1 >= 0 && tap - 1 < NP) ? 1 : 0);
                    const int npb = ((tap - 2 >= 0 && tap - 2 < NP - 1) ? 1 : 0) + ((tap - 1 >= 0 && tap - 1 < NP - 1) ? 1 : 0);
                    if (tap + 3 < 9 || !last) {
                        const int np = last ? 0 : (wave == 3 ? npb : npa);
                        if (np == 0) LXO_VMCNT(4); else if (np == 1) LXO_VMCNT(5); else LXO_VMCNT(6);
                    } else if (tap + 2 < 9) LXO_VMCNT(2);
                    else LXO_VMCNT(0);
                    if (tap + 1 < 9) bfr[0] = *reinterpret_cast<const u32x4*>(lxo_conv_lds + w_lane[0] + ((tap + 1) % 3) * WWSTAGE);
                    if (tap < NP && !last) issue_patch_piece(c + 1, (c + 1) & 1, tap);
                }
                if (tap + 1 < 9) af[i] = *reinterpret_cast<const u32x4*>(lxo_conv_lds + a_lane + ((i + kh1) * QPW + kw1) * WPIX);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!last) {
            // next slice: its patch (requested piece by piece during this one) and its first weight stage have landed once
            // only the two newest requests (the second and third tap's weights) are outstanding; the barrier makes the other
            // waves' patch pieces visible and tells that every wave is past the buffer that is refilled next
            LXO_VMCNT(4);
            __builtin_amdgcn_s_barrier();
            a_lane = a_lane0 + ((c + 1) & 1) * WPATCHB;
            fill();
        }
    }
    const int nk = nchunk;                                          // stamp slots: 1 + slice
    // The MFMA runs with the operand roles swapped (D = W X^T: rows = channels, columns = pixels), so
    // acc[i][e] = pixel (tile row wm*RI + i, column lane & 31), channel wn*32 + 8*(e>>2) + 4*khalf + (e&3):
    // a lane holds 4 CONSECUTIVE channels of one pixel per register quad, which is what both epilogues want.

    CSTAMP(1 + nk);
    if constexpr (EPI == 0 || EPI == 2 || EPI == 4) {
        // ---- plain epilogue (bias, ReLU, bf16): every wave packs its own 64 pixels x WBN channels into a bf16 tile in LDS
        // (8-byte writes of 4 channels), ONE barrier, then the workgroup streams the 256 pixel rows out in 16-byte pieces,
        // rows contiguous across lanes.  One pass instead of four f32 passes through a single 64-row staging tile.
        bf16_t* __restrict__ Cq = reinterpret_cast<bf16_t*>(p.C);
        constexpr int BP = WBN * 2 + 16;                                // row pitch in bytes: 16-byte aligned, an odd number of 16-byte groups
        char* bt = lxo_conv_lds;
        const float floor_q = p.act == 1 ? 0.f : -3.0e38f;
        LXO_LDS_BARRIER();                                              // every wave is past the patch and the weight stages
        CSTAMP(40);
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][4 * k + e], floor_q);
                const u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(bt + ((wm * RI + i) * 32 + (lane & 31)) * BP + (wn * 32 + 8 * k + 4 * khalf) * 2) = pk;
            }
        CSTAMP(41);
        LXO_LDS_BARRIER();
        CSTAMP(42);
        constexpr int CHQ = WBN / 8, RPQ = 256 / CHQ;                   // 16-byte pieces per pixel row, rows per sweep
        const int cq = tid % CHQ;
        bf16_t* const tile0 = Cq + (((long long)b * p.Ho + oy0) * p.Wo + ox0) * p.ldc + n0 + cq * 8;      // 64-bit once; 32-bit offsets per row
        if constexpr (EPI == 4) {
            // ---- fused max pool (encoder.py:39,47,52): the activated bf16 tile is complete in LDS; a thread takes 16-byte pieces of
            // POOLED pixels, reads the ph x pw pieces of the window, keeps the first maximum in scan order (what the separate pool
            // kernels do) and writes the pooled piece plus one mask byte per element (position | 4 if the maximum is positive).
            // The backward pass routes by the mask, so the full-resolution activation is written only if somebody asked for it.
            constexpr int ph = PH, pw = PW;                                  // compile-time window: the loops below unroll, no per-element branches
            const int Hq = (p.Ho + ph - 1) / ph, Wq = (p.Wo + pw - 1) / pw;
            constexpr int tqw = QTW / pw, npool = (QTH / ph) * tqw;          // pooled pixels of the tile: QTH % ph == QTW % pw == 0
            constexpr int NPP = npool * CHQ / 256;                           // pooled 16-byte pieces per thread
            bf16_t* __restrict__ Pq = reinterpret_cast<bf16_t*>(p.pool_out);
            unsigned char* __restrict__ Mq = p.pool_mask;
#pragma unroll
            for (int it = 0; it < NPP; ++it) {
                const int pp = tid + 256 * it;
                const int pq = pp / CHQ, cqq = pp - pq * CHQ;
                const int pty = pq / tqw, ptx = pq - pty * tqw;
                const int oyq = oy0 / ph + pty, oxq = ox0 / pw + ptx;
                u32x4 w4[ph * pw];
#pragma unroll
                for (int q = 0; q < ph * pw; ++q)
                    w4[q] = *reinterpret_cast<const u32x4*>(bt + ((pty * ph + q / pw) * QTW + ptx * pw + q % pw) * BP + cqq * 16);
                float best[8]; int bq[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { best[e] = -3.0e38f; bq[e] = 0; }
#pragma unroll
                for (int q = 0; q < ph * pw; ++q) {
                    const bool ok = oy0 + pty * ph + q / pw < p.Ho && ox0 + ptx * pw + q % pw < p.Wo;      // SAME pool ignores the padding
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned w = w4[q][e >> 1];
                        const float v = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
                        const bool up = ok && v > best[e];                   // strict: the first maximum in scan order stays
                        best[e] = up ? v : best[e]; bq[e] = up ? q : bq[e];
                    }
                }
                if (oyq < Hq && oxq < Wq) {
                    const long long o = (((long long)b * Hq + oyq) * Wq + oxq) * p.ldc + n0 + cqq * 8;
                    store8(Pq + o, best);
                    unsigned m0 = 0u, m1 = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m0 |= (unsigned)(bq[e] | (best[e] > 0.f ? 4 : 0)) << (8 * e);
                        m1 |= (unsigned)(bq[4 + e] | (best[4 + e] > 0.f ? 4 : 0)) << (8 * e);
                    }
                    *reinterpret_cast<u32x2*>(Mq + o) = u32x2{m0, m1};
                }
            }
            if (!p.C) { CSTAMP(2 + nk); return; }
        }
        u32x4 q4[256 / RPQ];
#pragma unroll
        for (int it = 0; it < 256 / RPQ; ++it) q4[it] = *reinterpret_cast<const u32x4*>(bt + (tid / CHQ + RPQ * it) * BP + cq * 16);
        if constexpr (EPI == 2) {
            // data gradient of a layer whose input was a ReLU output: d *= (reference activation > 0), bias-gradient column sums of
            // the masked tile.  The mask operand of every row is requested up front (unconditional, clamped), in the same 16-byte
            // row pieces as the stores; masking the already rounded bf16 values equals rounding the masked f32 values, and the
            // column sums are taken over what is stored.
            const bf16_t* const ref0 = reinterpret_cast<const bf16_t*>(p.relu_ref) + (((long long)b * p.Ho + oy0) * p.Wo + ox0) * p.ldr + n0 + cq * 8;
            u32x4 rf[256 / RPQ];
#pragma unroll
            for (int it = 0; it < 256 / RPQ; ++it) {
                const int row = tid / CHQ + RPQ * it, ty = row >> 5, tx = row & 31;
                const bool ok = oy0 + ty < p.Ho && ox0 + tx < p.Wo;
                rf[it] = *reinterpret_cast<const u32x4*>(ref0 + (ok ? (ty * p.Wo + tx) * p.ldr : 0));
            }
            float csum[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
            for (int it = 0; it < 256 / RPQ; ++it) {
                const int row = tid / CHQ + RPQ * it, ty = row >> 5, tx = row & 31;
                const bool ok = oy0 + ty < p.Ho && ox0 + tx < p.Wo;
                u32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned r = rf[it][d], v = q4[it][d];
                    // a bf16 activation is positive iff its sign bit is clear and it is not zero
                    const unsigned lo = (ok && (short)(r & 0xffffu) > 0) ? (v & 0xffffu) : 0u;
                    const unsigned hi = (ok && (int)r > 0 && (r >> 16) != 0u) ? (v & 0xffff0000u) : 0u;
                    o[d] = lo | hi;
                    csum[2 * d] += __uint_as_float(lo << 16);
                    csum[2 * d + 1] += __uint_as_float(hi);
                }
                if (ok) *reinterpret_cast<u32x4*>(tile0 + (ty * p.Wo + tx) * p.ldc) = o;
            }
            if (p.colsum || p.colsum_part) {
                float* cs = reinterpret_cast<float*>(lxo_conv_lds);      // [RPQ][WBN] partial sums; the bf16 tile has been read
                LXO_LDS_BARRIER();
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[(tid / CHQ) * WBN + cq * 8 + e] = csum[e];
                LXO_LDS_BARRIER();
                if (tid < WBN) {
                    float sres = 0.f;
#pragma unroll
                    for (int r = 0; r < RPQ; ++r) sres += cs[r * WBN + tid];
                    if (p.colsum_part) p.colsum_part[(long long)mt * p.N + n0 + tid] = sres;      // deterministic mode: this tile's slot
                    else atomicAdd(&p.colsum[n0 + tid], sres);
                }
            }
            CSTAMP(2 + nk);
            return;
        }
#pragma unroll
        for (int it = 0; it < 256 / RPQ; ++it) {
            const int row = tid / CHQ + RPQ * it;                       // 0..255: tile row row >> 5, column row & 31
            const int ty = row >> 5, tx = row & 31;
            if (oy0 + ty < p.Ho && ox0 + tx < p.Wo && !LXO_DIAG_BIT(4)) *reinterpret_cast<u32x4*>(tile0 + (ty * p.Wo + tx) * p.ldc) = q4[it];
        }
        CSTAMP(2 + nk);
        return;
    }
    // ---- epilogue: one wave's 64 pixels x WBN channels at a time through LDS as f32 [64][WBN + 4] ----
    bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(p.C);
    bf16_t* __restrict__ Cpre = reinterpret_cast<bf16_t*>(p.out_pre);
    const bf16_t* __restrict__ ref = reinterpret_cast<const bf16_t*>(p.relu_ref);
    float* ot = reinterpret_cast<float*>(lxo_conv_lds);
    constexpr int OP = WBN + 4, CH = 4 * NJ, RPI = 256 / CH;          // 16-byte chunks per row, rows per iteration
    const int c8 = (tid % CH) * 8, n = n0 + c8;
    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    const float act_floor = p.act == 1 ? 0.f : -3.0e38f;
    for (int pass = 0; pass < 4; ++pass) {
        LXO_LDS_BARRIER();
        CSTAMP(40 + 3 * pass);
        // the pass's two tile rows (2 pass, 2 pass + 1) live in accumulators 2 pass - wm RI + {0, 1} of the waves with wm == 2 pass / RI
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            if (pp == pass && wm == (2 * pp) / RI) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ib = (2 * pp) % RI;          // a constant once the loop is unrolled
                        const f32x4 q4 = {acc[ib + ii][4 * k], acc[ib + ii][4 * k + 1], acc[ib + ii][4 * k + 2], acc[ib + ii][4 * k + 3]};
                        *reinterpret_cast<f32x4*>(&ot[(ii * 32 + (lane & 31)) * OP + wn * 32 + 8 * k + 4 * khalf]) = q4;
                    }
            }
        }
        LXO_LDS_BARRIER();
        CSTAMP(41 + 3 * pass);
        constexpr int NIT = 64 / RPI;
        constexpr bool HAS_ADD = EPI == 1 || EPI == 3, HAS_REF = EPI == 2 || EPI == 3;
        // phase 1: every operand of the pass's rows is requested (LDS tile, addend, ReLU reference, old C) before anything is used
        f32x4 t0[NIT], t1[NIT], a0[NIT], a1[NIT], c0[NIT], c1[NIT];
        u32x4 rf[NIT];
        long long mrow[NIT];
        bool okr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = tid / CH + RPI * it;                        // 0..63 within the pass
            const int oy = oy0 + pass * 2 + (row >> 5), ox = ox0 + (row & 31);
            okr[it] = oy < p.Ho && ox < p.Wo;
            mrow[it] = okr[it] ? ((long long)b * p.Ho + oy) * p.Wo + ox : 0;
            t0[it] = *reinterpret_cast<const f32x4*>(ot + row * OP + c8); t1[it] = *reinterpret_cast<const f32x4*>(ot + row * OP + c8 + 4);
            if constexpr (HAS_ADD) {
                if (p.addend) {
                    const float* ad = p.addend + (mrow[it] % p.addend_rows) * p.N + n;
                    a0[it] = *reinterpret_cast<const f32x4*>(ad); a1[it] = *reinterpret_cast<const f32x4*>(ad + 4);
                } else { a0[it] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[it] = a0[it]; }
            }
            if constexpr (HAS_REF) {
                if (ref) rf[it] = *reinterpret_cast<const u32x4*>(ref + mrow[it] * p.ldr + n);
                else rf[it] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bf16 1.0: mask passes everything
            }
            if constexpr (EPI == 3) {
                if (p.accumulate) {
                    float cv[8];
                    load8(C + mrow[it] * p.ldc + n, cv);
                    c0[it] = f32x4{cv[0], cv[1], cv[2], cv[3]}; c1[it] = f32x4{cv[4], cv[5], cv[6], cv[7]};
                } else { c0[it] = f32x4{0.f, 0.f, 0.f, 0.f}; c1[it] = c0[it]; }
            }
        }
        // phase 2: arithmetic and the stores, back to back
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)      // ReLU as a max against a wave-uniform floor: no per-element branches (act is 0 or 1 here; the launcher routes tanh elsewhere); the bias is already inside the accumulators
                v[e] = fmaxf(e < 4 ? t0[it][e] : t1[it][e - 4], act_floor);
            if constexpr (HAS_ADD) {
                if (Cpre && okr[it]) store8(Cpre + mrow[it] * p.ldc + n, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (e < 4 ? a0[it][e] : a1[it][e - 4]);
            }
            if constexpr (HAS_REF) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w = rf[it][e >> 1];
                    const float rv = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
                    v[e] = rv > 0.f ? v[e] : 0.f;
                }
            }
            if (okr[it]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[e] += v[e];
            }
            if constexpr (EPI == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (e < 4 ? c0[it][e] : c1[it][e - 4]);
            }
            if (okr[it]) store8(C + mrow[it] * p.ldc + n, v);
        }
        CSTAMP(42 + 3 * pass);
    }
    CSTAMP(2 + nk);
    if (p.colsum || p.colsum_part) {
        LXO_LDS_BARRIER();
#pragma unroll
        for (int e = 0; e < 8; ++e) ot[(tid / CH) * OP + c8 + e] = csum[e];
        LXO_LDS_BARRIER();
        if (tid < WBN) {
            float sres = 0.f;
#pragma unroll
            for (int r = 0; r < RPI; ++r) sres += ot[r * OP + tid];
            if (p.colsum_part) p.colsum_part[(long long)mt * p.N + n0 + tid] = sres;              // deterministic mode: this tile's slot
            else atomicAdd(&p.colsum[n0 + tid], sres);
        }
    }
}

}  // namespace

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of the loaded code object: set it once per
// (kernel family, device), not once per process (a second GPU used from the same process would launch with the default limit)
static bool attr_needed(int family) {
    static bool done[4][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[family][dev]) return false;
    done[family][dev] = true;
    return true;
}

// Measurement aid: LXO_CONV_2WG=0 sends every call to the general halo kernel (A/B runs; tests/test_zz_gpu_variants.py).
static thread_local unsigned long long* g_conv_dbg = nullptr;
extern "C" int lxo_conv_debug(unsigned long long* buf) { g_conv_dbg = buf; return 0; }
int lxo_launch_conv_igemm(const GemmNT& p0, hipStream_t s) {
    GemmNT p = p0;
    p.dbg = g_conv_dbg;
#ifdef LXO_DIAG      // diagnostic builds only (make EXTRA=-DLXO_DIAG=1): LXO_CONV_DIAG removes parts of the kernel's work -- results are WRONG by design
    { static int diag = -1; if (diag < 0) { const char* e = getenv("LXO_CONV_DIAG"); diag = e ? atoi(e) : 0; } p.diag = diag; }
#else
    p.diag = 0;
#endif
    if (!p.conv || p.Cin % 64 || p.K % 64) return -2;
    static int use_2wg = -1;
    if (use_2wg < 0) { const char* e = getenv("LXO_CONV_2WG"); use_2wg = (e && e[0] == '0') ? 0 : 1; }
    // the halo kernel addresses its input through 32-bit buffer offsets (2^31 = its out-of-range marker): tensors below 2 GB only
    const bool in_32bit = (long long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * p.Cin * 2 < (1LL << 31);
    if (use_2wg && in_32bit && (p.N % 64) == 0 && p.act != 2 && p.alpha == 1.f) {
        constexpr int LDS4 = WLDS, LDS2 = WLDS;                                                    // 77824: the patch + four waves' two private weight stages
        if (attr_needed(0)) {
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 4, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<4, 4, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 4, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo2wg_kernel<2, 4, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        }
        const int B = p.M / (p.Ho * p.Wo);
        const int tiles_x = cdiv(p.Wo, QTW), tiles_y = cdiv(p.Ho, QTH);
        const bool has_add = p.addend || p.out_pre, has_ref = p.relu_ref || p.colsum || p.colsum_part;
        int epi = p.accumulate || (has_add && has_ref) ? 3 : (has_add ? 1 : (has_ref ? (p.relu_ref ? 2 : 3) : 0));      // 2 = the one-pass masked epilogue: needs the reference
        if (p.pool_out) {
            if (epi != 0 || p.N % 128 || !p.pool_mask || p.pool_h < 1 || p.pool_h > 2 || p.pool_w < 1 || p.pool_w > 2 || p.pool_h * p.pool_w == 1) return -2;
            epi = 4;
        } else if (!p.C) return -2;
        const dim3 g4(B * tiles_x * tiles_y * (p.N / 128)), g2(B * tiles_x * tiles_y * (p.N / 64));
        if (p.colsum_part && (size_t)B * tiles_x * tiles_y * p.N > p.colsum_part_floats) return -6;
        // A launch of 128-channel tiles that cannot give every CU a workgroup (the reference's own batches: 3 .. 20 images of 50 x 120 .. 100 x 360) runs on
        // 64-channel tiles instead: twice the workgroups, half the matrix work per workgroup -- a tile's time is its K loop (Cin / 32 slices x 9 taps x
        // 2 RI MFMAs per wave, ~35 us at Cin = 512 whatever the grid), so the layer takes about half as long.  Same arithmetic per output element.
        static int small_thr = -1;
        if (small_thr < 0) { const char* e = getenv("LXO_CONV_SMALL"); small_thr = e ? atoi(e) : 256; }
        if (p.N % 128 == 0 && (int)g4.x >= small_thr) {
            if (epi == 0) hipLaunchKernelGGL((conv_halo2wg_kernel<4, 0>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
            else if (epi == 1) hipLaunchKernelGGL((conv_halo2wg_kernel<4, 1>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
            else if (epi == 2) hipLaunchKernelGGL((conv_halo2wg_kernel<4, 2>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
            else if (epi == 4 && p.pool_h == 2 && p.pool_w == 2) hipLaunchKernelGGL((conv_halo2wg_kernel<4, 4, 2, 2>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
            else if (epi == 4 && p.pool_h == 2) hipLaunchKernelGGL((conv_halo2wg_kernel<4, 4, 2, 1>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
            else if (epi == 4) hipLaunchKernelGGL((conv_halo2wg_kernel<4, 4, 1, 2>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
            else hipLaunchKernelGGL((conv_halo2wg_kernel<4, 3>), g4, dim3(WTHR), LDS4, s, p, p.N / 128, tiles_x, tiles_y);
        } else {
            if (epi == 0) hipLaunchKernelGGL((conv_halo2wg_kernel<2, 0>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
            else if (epi == 1) hipLaunchKernelGGL((conv_halo2wg_kernel<2, 1>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
            else if (epi == 2) hipLaunchKernelGGL((conv_halo2wg_kernel<2, 2>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
            else if (epi == 4 && p.pool_h == 2 && p.pool_w == 2) hipLaunchKernelGGL((conv_halo2wg_kernel<2, 4, 2, 2>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
            else if (epi == 4 && p.pool_h == 2) hipLaunchKernelGGL((conv_halo2wg_kernel<2, 4, 2, 1>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
            else if (epi == 4) hipLaunchKernelGGL((conv_halo2wg_kernel<2, 4, 1, 2>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
            else hipLaunchKernelGGL((conv_halo2wg_kernel<2, 3>), g2, dim3(WTHR), LDS2, s, p, p.N / 64, tiles_x, tiles_y);
        }
        if (p.colsum_part) return lxo_k_det_reduce(p.colsum_part, B * tiles_x * tiles_y, p.N, p.N, p.colsum, s);      // the tiles' sums in tile order
        return (int)hipGetLastError();
    }
    if (p.pool_out || !p.C) return -2;                    // the fused pool lives in conv_halo2wg_kernel only
    if (p.colsum_part) return -7;                         // no slot form here: the caller runs it without the fused sum + an ordered pass
    // everything the two-workgroup kernel does not take (Cout % 64 != 0, tensors of 2 GB and more, tanh, alpha != 1): the general halo kernel
    {
        constexpr int LDSB = 2 * HPATCH + 3 * HB_STAGE, LDSB64 = 2 * HPATCH + 3 * (HB_STAGE / 2);
        if (attr_needed(2)) {
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<bf16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<bf16_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB64));
        }
        const int B = p.M / (p.Ho * p.Wo);
        const int tiles_x = cdiv(p.Wo, HTW), tiles_y = cdiv(p.Ho, HTH);
        if (p.N <= 64) {     // a 64-channel-wide tile: no MFMA work on padding channels (conv2's dgrad)
            const int tiles_n = cdiv(p.N, 64);
            hipLaunchKernelGGL((conv_halo_kernel<bf16_t, 1>), dim3(B * tiles_x * tiles_y * tiles_n), dim3(CTH), LDSB64, s, p, tiles_n, tiles_x, tiles_y);
            return (int)hipGetLastError();
        }
        const int tiles_n = cdiv(p.N, CBN);
        hipLaunchKernelGGL((conv_halo_kernel<bf16_t, 2>), dim3(B * tiles_x * tiles_y * tiles_n), dim3(CTH), LDSB, s, p, tiles_n, tiles_x, tiles_y);
        return (int)hipGetLastError();
    }
}
