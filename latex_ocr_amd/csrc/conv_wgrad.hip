// 3x3 convolution weight gradient on the bf16 MFMA units, gfx950:
//     dW[tap][ci][co] += sum over pixels  in[pixel + tap][ci] * d_out[pixel][co]
//
// A workgroup owns a (64 ci) x (128 co) tile of ALL NINE taps (9 x 32x32 accumulator tiles per
// wave = 144 registers) and walks a range of 2 x 64 pixel blocks.  Per block it stages, by
// LDS-DMA and double buffered, the (2+2) x (64+2) input patch for its 64 channels and the 128
// d_out pixels for its 128 channels -- each once, serving all nine taps (262 FLOP per staged
// byte).  The contraction runs over PIXELS, which are the strided index of NHWC, so MFMA
// operands are fetched with the transposing LDS read ds_read_b64_tr_b16 (4 consecutive pixels
// of one channel per lane); the kw = 0 and kw = 2 windows of a patch row are read from LDS, the
// kw = 1 window is four v_perm of the two.  Partial tiles are reduced with f32 atomics; the pixel
// ranges of the workgroups have unequal lengths so that those epilogues do not collide.
#include "gemm.h"
#include "api_util.h"
#include "lxo_debug.h"
#include <stdlib.h>

namespace {

constexpr int WTH = 2, WTW = 64, WPW = WTW + 2, WPH = WTH + 2, WPROWS = WPH * WPW;   // 264 patch pixels
constexpr int WTHREADS = 512;
constexpr int WPATCH = 5 * WTHREADS * 16;           // 40960 B: 2560 slots >= 264 * 8
constexpr int WDY = WTH * WTW * 256;                // 32768 B: 128 pixels x 128 co
constexpr int WSTAGE = WPATCH + WDY;
constexpr int WCI = 64, WCO = 128;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short v4s_t;
typedef __attribute__((address_space(3))) v4s_t* ltr_t;
LXO_DEV void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)g, (lptr_t)(uintptr_t)l, 16, 0, 0);
}
// 4 consecutive pixels (rows of the LDS image) of this lane's channel, as two dwords of bf16 pairs
LXO_DEV u32x2 tr_read(const char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(uintptr_t)p));
}

}  // namespace

HIP_DYNAMIC_SHARED(char, lxo_wgrad_lds)

namespace {

__global__ __launch_bounds__(512) void conv_wgrad_kernel(GemmTN p, int tiles_co, int tiles_x, int tiles_y, int nblocks, int nsplit, float stagger) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wci = wave >> 2, wco = wave & 3;            // 2 x 4 waves: 32 ci x 32 co each (x 9 taps)
    // XCD-aware order: workgroup L runs on XCD L % 8 (each XCD has its own L2).  The `ntiles` (ci, co) tiles of ONE pixel
    // range read the same input and d_out blocks, so they are made neighbours on ONE XCD (the n-th workgroup of XCD x is
    // tile n % ntiles of pixel range (n / ntiles) * 8 + x): every block is then fetched once per XCD and served to the other
    // tiles from L2, instead of every XCD pulling every block through the fabric (grid (tile, split) put tile t on XCD t % 8).
    const int ntiles = p.nbatch;                                // launcher passes the tile count here (nbatch is unused for conv)
    const int L = blockIdx.x, xcd = L & 7, seq = L >> 3;
    const int tile = seq % ntiles, split = (seq / ntiles) * 8 + xcd;
    const int ci0 = (tile / tiles_co) * WCI, co0 = (tile % tiles_co) * WCO;
    const int Cout = p.J;
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ DY = reinterpret_cast<const bf16_t*>(p.B);
    // Pixel ranges of UNEQUAL length: every workgroup ends with 73 728 f32 atomics per lane-row into the same few MB, and with equal
    // ranges all 256 of them arrive there together -- the atomic units then serve 18.9 M lane-atomics while every matrix pipe idles
    // (stamps: 44-65 k cycles, 10-20 % of the launch).  Range length grows linearly with the split index (1 -/+ stagger at the ends),
    // so the workgroups finish spread over about the time the atomics take and the epilogues of the early ones run under the K
    // loops of the late ones.  boundary(s) = nblocks * (u + stagger * (u * u - u)), u = s / nsplit.
    if (split >= nsplit) return;                           // (slots >= nsplit do not exist)
    auto boundary = [&](int sidx) {
        const float u = (float)sidx / (float)nsplit;
        const int v = (int)((float)nblocks * (u + stagger * (u * u - u)) + 0.5f);
        return sidx >= nsplit ? nblocks : min(nblocks, max(0, v));
    };
    const int pb_beg = __builtin_amdgcn_readfirstlane(boundary(split)), pb_end = __builtin_amdgcn_readfirstlane(boundary(split + 1));
    if (pb_beg >= pb_end) {
        if (p.det_slab) {                                      // deterministic mode: an empty range still owns a slot: zeros
            float* const sb = p.det_slab + ((long long)split * ntiles + tile) * (9 * WCI * WCO);
            for (int i = tid; i < 9 * WCI * WCO / 4; i += WTHREADS) reinterpret_cast<f32x4*>(sb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // ---- the LDS-DMA of one pixel block: 5 (wave 0) / 4 requests for the patch, 4 for d_out, through buffer resources.
    // The request addresses used to be rebuilt from the block index for every request (divisions, 64-bit multiplies, a select
    // against a zero line: ~25 VALU instructions, a third of them quarter rate); in-kernel stamps of a build without the DMA put
    // that at 2.1 k of a block's 7.6 k cycles (the matrix pipe waits while a wave issues VALU work).  Now: everything that depends
    // on the lane is computed ONCE (byte offset of the lane's chunk from the block's patch origin; its patch row / column, packed),
    // everything that depends on the block is SCALAR (soffset of the request; the valid row / column window of the block), and
    // zero padding = an out-of-range offset (the buffer returns zeros): 4 VALU instructions per patch request, 2 per d_out request.
    typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
    const unsigned m0b = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lxo_wgrad_lds);
    const int nB = p.M / (p.Ho * p.Wo);
    const unsigned padoff = (unsigned)p.pad * (unsigned)(p.W + 1) * (unsigned)p.Cin * 2u;     // the patch origin of the first block lies `pad` rows and columns before the tensor
    const lxo_rsrc_t rx = lxo_make_rsrc(reinterpret_cast<const char*>(X) - padoff, (unsigned)((long long)nB * p.H * p.W * p.Cin * 2) + padoff);
    const lxo_rsrc_t rdy = lxo_make_rsrc(DY, (unsigned)((long long)nB * p.Ho * p.Wo * Cout * 2));
    unsigned prel[5], ppk[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {                          // patch pixel prow = (tid >> 3) + 64 j, 16-byte chunk tid & 7 (swizzled by the pixel pair)
        const int prow = (tid >> 3) + 64 * j;
        const int py = prow / WPW, px = prow - py * WPW;
        prel[j] = (unsigned)(((py * p.W + px) * p.Cin + (((tid & 7) ^ ((prow >> 1) & 7)) << 3)) * 2);
        ppk[j] = ((unsigned)py << 16) | (unsigned)px;
    }
    // d_out: block pixel k = (tid >> 4) + 32 j = row j >> 1, column (tid >> 4) + 32 (j & 1); 16-byte chunk tid & 15 swizzled by k & 15 (the same for all j)
    const int dgch = ((tid & 15) ^ ((tid >> 4) & 15)) << 3;
    const unsigned drel = (unsigned)(((tid >> 4) * Cout + dgch) * 2);
    const int dxq = (co0 + dgch) < Cout ? (tid >> 4) : 0x7FFF0000;                 // a channel chunk beyond Cout never passes the column test
    int tx_i = pb_beg % tiles_x, ty_i = (pb_beg / tiles_x) % tiles_y, b_i = pb_beg / (tiles_x * tiles_y);   // walked, not divided, from here on
    auto issue = [&](int stage) {                          // the block (b_i, ty_i, tx_i); then steps to the next one
        const int oy0 = ty_i * WTH, ox0 = tx_i * WTW;
        const unsigned sx = (unsigned)((((b_i * p.H + oy0) * p.W + ox0) * p.Cin + ci0) * 2);
        const int loy = max(0, p.pad - oy0), lox = max(0, p.pad - ox0);
        const int ny = min(WPH, p.H + p.pad - oy0) - loy, nx = min(WPW, p.W + p.pad - ox0) - lox;   // >= 1: the block exists
        const unsigned lo = __builtin_amdgcn_readfirstlane(((unsigned)loy << 16) | (unsigned)lox);
        const unsigned lim = __builtin_amdgcn_readfirstlane(((unsigned)(ny - 1) << 16) | (unsigned)(nx - 1));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j == 4 && wave != 0) break;                 // pixels 256..263: wave 0 only (slots beyond 263 are never read)
            const u16x2_t t = __builtin_bit_cast(u16x2_t, ppk[j]) - __builtin_bit_cast(u16x2_t, lo);
            const u16x2_t m = __builtin_elementwise_min(t, __builtin_bit_cast(u16x2_t, lim));
            const bool ok = __builtin_bit_cast(unsigned, m) == __builtin_bit_cast(unsigned, t) && (j < 4 || tid < 64);
            const unsigned voff = ok ? prel[j] : LXO_BLDS_OOB;
            LXO_BLDS16(voff, rx, sx, lxo_wgrad_lds, m0b, stage * WSTAGE + (wave * 64 + 512 * j) * 16);
        }
        const unsigned sd = (unsigned)((((b_i * p.Ho + oy0) * p.Wo + ox0) * Cout + co0) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hx = (oy0 + (j >> 1) < p.Ho) ? p.Wo - ox0 - 32 * (j & 1) : 0;            // scalar: columns of this request's row that exist
            const unsigned voff = dxq < hx ? drel : LXO_BLDS_OOB;
            const unsigned sdj = __builtin_amdgcn_readfirstlane(sd + (unsigned)((((j >> 1) * p.Wo + 32 * (j & 1)) * Cout) * 2));
            LXO_BLDS16(voff, rdy, sdj, lxo_wgrad_lds, m0b, stage * WSTAGE + WPATCH + (wave * 64 + 512 * j) * 16);
        }
        if (++tx_i == tiles_x) { tx_i = 0; if (++ty_i == tiles_y) { ty_i = 0; ++b_i; } }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // lane geometry of the transposing reads: group g = (lane>>4)&1 (16 channels), q = lane&15 -> (j = q>>2: pixel, c = q&3: 4 channels)
    const int h = lane >> 5, g = (lane >> 4) & 1, qj = (lane & 15) >> 2, qc = lane & 3;
    const int a_ch = wci * 32 + 16 * g + 4 * qc;           // channel (within the 64) this lane addresses
    const int a_chunk = a_ch >> 3, a_sub = (a_ch & 7) * 2;  // 16-byte chunk, byte offset inside it
    const int b_ch = wco * 32 + 16 * g + 4 * qc;           // channel (within the 128)
    const int b_chunk = b_ch >> 3, b_sub = (b_ch & 7) * 2;
    // Every LDS address of the K loop = (per-lane register) + (compile-time immediate): the loop used to spend ~14 issue
    // slots per MFMA on swizzle arithmetic (SQ_ACTIVE_INST_ANY 39 %, MFMA pipe 33 % busy).  A patch read touches pixel
    // row prow = Cc + bl with Cc even and compile-time, bl = 8h + qj per lane, so its swizzle key ((prow >> 1) & 7) is
    // (Cc/2 + (bl >> 1)) & 7: eight per-lane offsets, indexed by the compile-time (Cc/2) & 7, cover every read.  A d_out
    // read touches pixel Cd + bl with Cd a multiple of 16, so its key (pixel & 15) depends on the lane only.
    const int bl = 8 * h + qj;
    int poff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) poff[k] = bl * 128 + ((a_chunk ^ ((k + (bl >> 1)) & 7)) << 4) + a_sub;
    const int doff_a = WPATCH + bl * 256 + ((b_chunk ^ (bl & 15)) << 4) + b_sub;
    const int doff_b = WPATCH + (bl + 4) * 256 + ((b_chunk ^ ((bl + 4) & 15)) << 4) + b_sub;

#define WSTAMP(i) do { if (p.dbg && tid == 0 && (i) < 64) p.dbg[(long long)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
    WSTAMP(0);
    issue(0);
    for (int pb = pb_beg; pb < pb_end; ++pb) {
        const int stage = (pb - pb_beg) & 1;
        WSTAMP(1 + 3 * (pb - pb_beg));
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): only this block's loads are outstanding here
        WSTAMP(2 + 3 * (pb - pb_beg));
        __builtin_amdgcn_s_barrier();
        WSTAMP(3 + 3 * (pb - pb_beg));
#ifndef LXO_WG_DIAG
#define LXO_WG_DIAG 0                                      // measurement builds only: 1 = no DMA after the first block, 2 = no LDS reads in the loop, 4 = no v_perm
#endif
        const char* sb = lxo_wgrad_lds + stage * WSTAGE;
        const char* pk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = sb + poff[k];
        const char* da = sb + doff_a;
        const char* db = sb + doff_b;
        // The K loop walks 16 sub-steps (16-pixel column group xg, patch row `row` of the four): patch row `row` is the kh = row
        // operand of output row 0 and the kh = row - 1 operand of output row 1, so it is read ONCE and feeds 3 + 3 MFMAs (rows
        // 0 and 3: 3).  The kw = 0 and the kw = 2 windows (pixels 0..7 and 2..9 of the lane's run) are BOTH read from LDS (2
        // transposing reads each: the second window used to be four v_mov out of an odd register pair), the kw = 1 window is
        // four v_perm of the two (all three from LDS is LDS-bound).  Per 18 MFMAs: 20 LDS reads + 16 v_perm (round 2: 22 + 48);
        // measured with LXO_WG_DIAG builds, these cost ~4 % of the loop -- what cost 30 % was the DMA issue (see `issue`).
        // Raw patch operands are read one sub-step ahead, the d_out operands of both output rows one column group ahead.
        u32x2 ra[2][4], rb[2][4];
        auto read_b = [&](int xg, u32x2 (&b4)[4]) {
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int cd = (ty * 64 + xg * 16) * 256;
                b4[2 * ty] = tr_read(da + cd);
                b4[2 * ty + 1] = tr_read(db + cd);
            }
        };
        auto read_a = [&](int s, u32x2 (&r)[4]) {
            const int cc = (s & 3) * WPW + (s >> 2) * 16;                       // even, compile-time
            r[0] = tr_read(pk[(cc >> 1) & 7] + cc * 128);                       // pixels 0..3
            r[1] = tr_read(pk[((cc + 4) >> 1) & 7] + (cc + 4) * 128);           //        4..7
            r[2] = tr_read(pk[((cc + 2) >> 1) & 7] + (cc + 2) * 128);           //        2..5
            r[3] = tr_read(pk[((cc + 6) >> 1) & 7] + (cc + 6) * 128);           //        6..9
        };
        read_b(0, rb[0]);
        read_a(0, ra[0]);
        __builtin_amdgcn_sched_barrier(0);                  // the first LDS reads fly while the next block's DMA requests are issued
        if (pb + 1 < pb_end && !((LXO_WG_DIAG & 1) && pb > pb_beg)) issue(stage ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int xg = s >> 2, row = s & 3, cur = s & 1, bcur = xg & 1;     // compile-time after unrolling
            // Two waves share a SIMD's MFMA pipe; with the default (age-based) arbitration one runs ahead and then idles at
            // the block barrier while the other finishes alone (2 k of a block's 8.3 k cycles in the stamps).  Priority falls
            // as a wave advances through the block, so the wave that is behind wins the pipe and both arrive together.
            if (s == 0) __builtin_amdgcn_s_setprio(3);
            else if (s == 4) __builtin_amdgcn_s_setprio(2);
            else if (s == 8) __builtin_amdgcn_s_setprio(1);
            else if (s == 12) __builtin_amdgcn_s_setprio(0);
            if (!(LXO_WG_DIAG & 2)) {
                if (s + 1 < 16) read_a(s + 1, ra[cur ^ 1]);
                if (row == 1 && xg + 1 < 4) read_b(xg + 1, rb[bcur ^ 1]);
            } else if (s == 0) { read_a(1, ra[1]); read_b(1, rb[1]); }
            __builtin_amdgcn_sched_barrier(0);              // the reads of sub-step s + 1 stay ahead of the MFMAs of s; nothing is hoisted further
            const u32x2* r = ra[cur];
            const u32x4 a0 = {r[0][0], r[0][1], r[1][0], r[1][1]};
            const u32x4 a2 = {r[2][0], r[2][1], r[3][0], r[3][1]};
            const u32x4 b0 = {rb[bcur][0][0], rb[bcur][0][1], rb[bcur][1][0], rb[bcur][1][1]};   // output row 0: kh = row
            const u32x4 b1 = {rb[bcur][2][0], rb[bcur][2][1], rb[bcur][3][0], rb[bcur][3][1]};   // output row 1: kh = row - 1
            // order: one kw = 0 MFMA first, the four v_perm of the kw = 1 window in its shadow, then the rest
#define WG_MFMA(A, B, T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, B), acc[T], 0, 0, 0)
            if (row <= 2) WG_MFMA(a0, b0, row * 3 + 0); else WG_MFMA(a0, b1, (row - 1) * 3 + 0);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 a1 = (LXO_WG_DIAG & 4) ? u32x4{r[0][1], r[1][0], r[1][1], r[3][1]} :
                             u32x4{__builtin_amdgcn_alignbit(r[0][1], r[0][0], 16), __builtin_amdgcn_alignbit(r[1][0], r[0][1], 16),
                                   __builtin_amdgcn_alignbit(r[1][1], r[1][0], 16), __builtin_amdgcn_alignbit(r[3][1], r[1][1], 16)};
            __builtin_amdgcn_sched_barrier(0);
            if (row <= 2) {
                WG_MFMA(a2, b0, row * 3 + 2);
                WG_MFMA(a1, b0, row * 3 + 1);
                if (row >= 1) WG_MFMA(a0, b1, (row - 1) * 3 + 0);
            }
            if (row >= 1) {
                WG_MFMA(a2, b1, (row - 1) * 3 + 2);
                WG_MFMA(a1, b1, (row - 1) * 3 + 1);
            }
#undef WG_MFMA
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    WSTAMP(61);
    // reduce into dW[(tap*Cin + ci)][co]
    // Every workgroup of a tile adds into the same 73 728 addresses, and all workgroups reach this point together: walking
    // the taps in the same order made them queue on the same few cache lines (the epilogue cost 70 k cycles, 12 % of the
    // kernel).  Each workgroup starts at a different tap (the accumulator index stays a compile-time constant: two
    // unrolled sweeps behind uniform predicates), which spreads the concurrent atomics over nine times as many lines
    // (stamps: 70 k -> 47 k cycles on conv6, 77 k -> 69 k on conv2 where all 256 workgroups share one tile; the launch time moved
    // within noise: what remains is the L2 atomic rate for 18.9 M lane-atomics per launch).
    if (p.det_slab) {
        // deterministic mode: the partial tile goes to this workgroup's slot [tap][ci 64][co 128] of the slab with plain stores;
        // wgrad_slab_reduce_kernel adds the slots of a tile in pixel-range order
        float* const sb = p.det_slab + ((long long)split * ntiles + tile) * (9 * WCI * WCO) + (wci * 32 + 4 * (lane >> 5)) * WCO + wco * 32 + (lane & 31);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) sb[(t * WCI + (e & 3) + 8 * (e >> 2)) * WCO] = acc[t][e];
        return;
    }
    const int co = co0 + wco * 32 + (lane & 31);
    if (co < Cout) {
        const int rot = (int)(blockIdx.x % 9u);
        // one per-lane base pointer, wave-uniform element offsets (Cin % 64 == 0: every ci of the tile exists)
        float* const cbase = p.C + (long long)(ci0 + wci * 32 + 4 * (lane >> 5)) * p.ldc + co;
        auto emit = [&](const f32x16& a, int t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) atomicAdd(cbase + (t * p.Cin + (e & 3) + 8 * (e >> 2)) * p.ldc, a[e]);
        };
        // order rot, rot + 1, ..., 8, 0, ..., rot - 1 with compile-time accumulator indices: two unrolled sweeps, uniform branches
#pragma unroll
        for (int t = 0; t < 9; ++t) if (t >= rot) emit(acc[t], t);
#pragma unroll
        for (int t = 0; t < 9; ++t) if (t < rot) emit(acc[t], t);
    }
    WSTAMP(62);
}

// deterministic mode: dW[(tap * Cin + ci0 + ci)][co0 + co] += sum over the pixel ranges s = 0 .. nsplit-1 (in that order) of slab[s][tile][tap][ci][co];
// slots of empty ranges were zeroed by the launcher
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, int nsplit, int ntiles, int tiles_co, int Cin, int Cout,
                                                               float* __restrict__ dw, int ldc) {
    const int tile = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;                 // element of the 9 x 64 x 128 tile
    if (i >= 9 * WCI * WCO) return;
    const int co = i % WCO, ci = (i / WCO) % WCI, tap = i / (WCO * WCI);
    const int ci0 = (tile / tiles_co) * WCI, co0 = (tile % tiles_co) * WCO;
    float s = 0.f;
    for (int q = 0; q < nsplit; ++q) s += slab[((long long)q * ntiles + tile) * (9 * WCI * WCO) + i];
    if (co0 + co < Cout) dw[(long long)(tap * Cin + ci0 + ci) * ldc + co0 + co] += s;
}

}  // namespace

// bf16 only; Cin % 64 == 0, Cout % 8 == 0
static thread_local unsigned long long* g_wgrad_dbg = nullptr;
extern "C" int lxo_wgrad_debug(unsigned long long* buf) { g_wgrad_dbg = buf; return 0; }
int lxo_launch_conv_wgrad(const GemmTN& p, hipStream_t s) {
    if (!p.conv || p.Cin % 64 || p.J % 8) return -2;
    {   // both tensors are addressed through 32-bit buffer offsets (and 2^31 is the kernel's out-of-range offset)
        const long long nb = p.M / (p.Ho * p.Wo);
        if (nb * p.H * p.W * p.Cin * 2 + 2LL * (p.W + 1) * p.Cin >= (1LL << 31) || nb * p.Ho * p.Wo * p.J * 2 >= (1LL << 31)) return -2;
    }
    {   // per device, not per process (see conv_igemm.hip attr_needed)
        static bool done[64] = {};
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (!known || !done[dev]) {
            HIPRC(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE));
            if (known) done[dev] = true;
        }
    }
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles_ci = p.Cin / WCI, tiles_co = cdiv(p.J, WCO);
    const int tiles_x = cdiv(p.Wo, WTW), tiles_y = cdiv(p.Ho, WTH);
    const int nblocks = B * tiles_x * tiles_y;
    const int tiles = tiles_ci * tiles_co;
    int nsplit = cdiv(256, tiles);
    if (nsplit > nblocks) nsplit = nblocks;
    // every split ends with the atomic epilogue of a whole (64 ci x 128 co x 9 taps) tile -- 295 KB of f32 atomics onto addresses its sibling
    // splits hit too -- whatever its pixel range: on the small maps of the reference's real buckets (configs/data.json: 50 x 120 ... at batch 20)
    // a range of one or two pixel blocks made the launch all epilogue.  At least LXO_WG_MINBLK = 8 blocks per split (the benchmark shape has 16 .. 56: unchanged)
    // (swept over four real buckets at batch 20 / 64, profiles/r06_wgrad_minblk.txt: 1 / 4 / 8 / 16 blocks -> 1.678 / 1.644 / 1.615 / 1.672 ms per step at 50 x 120)
    static const int min_blk = getenv("LXO_WG_MINBLK") ? atoi(getenv("LXO_WG_MINBLK")) : 8;
    if (min_blk > 1 && nsplit > cdiv(nblocks, min_blk)) nsplit = cdiv(nblocks, min_blk);
    if (nsplit < 1) nsplit = 1;
    const int per_split = cdiv(nblocks, nsplit);
    nsplit = cdiv(nblocks, per_split);
    // the end-of-range spread that hides the atomic epilogue in units of blocks at each end of the ramp (swept in-step: 0 / 4.3 / 10 / 14 / 20 -> 0.48 / 0.50 / 0.526 / 0.50 / 0.48 of the MFMA peak; the
    // spread also takes the 256 workgroups out of phase, which evens out the DMA bursts of conv2 -- one tile, no operand shared between workgroups, 4 TB/s)
    static const float stag_scale = getenv("LXO_WG_STAGGER") ? (float)atof(getenv("LXO_WG_STAGGER")) : 10.f;
    float stagger = stag_scale / (float)per_split;
    if (stagger > 0.5f) stagger = 0.5f;
    if (nsplit < 2) stagger = 0.f;
    GemmTN q = p;
    q.nbatch = tiles;
    q.dbg = g_wgrad_dbg;
    // splits are dealt to the 8 XCDs in turn: round the split count up to a multiple of 8 (empty ranges return at once)
    const int nsplit8 = (nsplit + 7) / 8 * 8;
    if (q.det_slab) {
        const size_t need = (size_t)nsplit * tiles * 9 * WCI * WCO;
        if (need > q.det_floats) return -6;
    }
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles * nsplit8), dim3(WTHREADS), 2 * WSTAGE, s, q, tiles_co, tiles_x, tiles_y, nblocks, nsplit, stagger);
    if (q.det_slab)
        hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(9 * WCI * WCO / 256, tiles), dim3(256), 0, s, q.det_slab, nsplit, tiles, tiles_co, p.Cin, p.J, p.C, p.ldc);
    return (int)hipGetLastError();
}
