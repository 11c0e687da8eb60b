// 3x3 convolution weight gradient on the bf16 MFMA units, gfx950:
//     dW[tap][ci][co] += sum over pixels  in[pixel + tap][ci] * d_out[pixel][co]
//
// A workgroup owns a (64 ci) x (128 co) tile of ALL NINE taps (9 x 32x32 accumulator tiles per
// wave = 144 registers) and walks a range of 2 x 64 pixel blocks.  Per block it stages, by
// LDS-DMA and double buffered, the (2+2) x (64+2) input patch for its 64 channels and the 128
// d_out pixels for its 128 channels -- each once, serving all nine taps (262 FLOP per staged
// byte).  The contraction runs over PIXELS, which are the strided index of NHWC, so MFMA
// operands are fetched with the transposing LDS read ds_read_b64_tr_b16 (4 consecutive pixels
// of one channel per lane); the kw = 0/1/2 shifted operands of a patch row come from the same
// three reads through v_alignbit.  Partial tiles are reduced with f32 atomics.
#include "gemm.h"
#include "api_util.h"
#include <stdlib.h>

namespace {

__device__ unsigned lxo_wg_zero_line[8] = {0, 0, 0, 0, 0, 0, 0, 0};

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

__global__ __launch_bounds__(512) void conv_wgrad_kernel(GemmTN p, int tiles_co, int tiles_x, int tiles_y, int nblocks, int per_split) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const char* zline = reinterpret_cast<const char*>(lxo_wg_zero_line);
    const int pb_beg = split * per_split, pb_end = min(nblocks, pb_beg + per_split);
    if (pb_beg >= pb_end) return;

    const int sch = tid & 7;                               // patch: 16-byte chunk (8 channels) of the 64-channel row
    const int dch = tid & 15;                              // d_out: 16-byte chunk of the 128-channel row
    auto issue = [&](int pb, int stage) {
        const int tx_i = pb % tiles_x, ty_i = (pb / tiles_x) % tiles_y, b = pb / (tiles_x * tiles_y);
        const int oy0 = ty_i * WTH, ox0 = tx_i * WTW;
#pragma unroll
        for (int j = 0; j < 5; ++j) {                      // 5 LDS-DMA: patch pixel prow = (tid >> 3) + 64 j
            const int prow = (tid >> 3) + 64 * j;
            const int py = prow / WPW, px = prow - py * WPW;
            const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
            const bool ok = prow < WPROWS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int gch = (sch ^ ((prow >> 1) & 7)) << 3;
            const void* src = ok ? (const void*)(X + (((long long)b * p.H + iy) * p.W + ix) * p.Cin + ci0 + gch) : (const void*)zline;
            LXO_GLDS16_HIDDEN(src, lxo_wgrad_lds, stage * WSTAGE + (wave * 64 + 512 * j) * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // 4 LDS-DMA: block pixel k = (tid >> 4) + 32 j
            const int k = (tid >> 4) + 32 * j;
            const int oy = oy0 + (k >> 6), ox = ox0 + (k & 63);
            const int gch = (dch ^ (k & 15)) << 3;            // the GLOBAL chunk this lane fetches (LDS slot dch holds it)
            const bool ok = oy < p.Ho && ox < p.Wo && (co0 + gch) < Cout;
            const void* src = ok ? (const void*)(DY + (((long long)b * p.Ho + oy) * p.Wo + ox) * Cout + co0 + gch) : (const void*)zline;
            LXO_GLDS16_HIDDEN(src, lxo_wgrad_lds, stage * WSTAGE + WPATCH + (wave * 64 + 512 * j) * 16);
        }
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
    issue(pb_beg, 0);
    for (int pb = pb_beg; pb < pb_end; ++pb) {
        const int stage = (pb - pb_beg) & 1;
        WSTAMP(1 + 3 * (pb - pb_beg));
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): only this block's loads are outstanding here
        WSTAMP(2 + 3 * (pb - pb_beg));
        __builtin_amdgcn_s_barrier();
        WSTAMP(3 + 3 * (pb - pb_beg));
        if (pb + 1 < pb_end) issue(pb + 1, stage ^ 1);
        const char* sb = lxo_wgrad_lds + stage * WSTAGE;
        const char* pk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = sb + poff[k];
        const char* da = sb + doff_a;
        const char* db = sb + doff_b;
        // One K-step of raw operands (2 d_out reads + 9 patch reads = 22 dwords) is read AHEAD: the reads of K-step ks + 1 are
        // issued before the MFMAs of K-step ks, so the LDS latency hides under 288 cycles of matrix work instead of stalling
        // every K-step (SQ_WAIT_ANY was 31 % with the reads and the MFMAs of a K-step back to back).
        u32x2 rb[2][2], rd[2][9];
        auto read_ks = [&](int ks, u32x2 (&b2)[2], u32x2 (&d9)[9]) {
            const int ty = ks >> 2, xc = (ks & 3) * 16;
            const int cd = (ty * 64 + xc) * 256;
            b2[0] = tr_read(da + cd);
            b2[1] = tr_read(db + cd);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) {
                    const int cc = (ty + kh) * WPW + xc + 4 * q3;        // even, compile-time
                    d9[kh * 3 + q3] = tr_read(pk[(cc >> 1) & 7] + cc * 128);
                }
        };
        read_ks(0, rb[0], rd[0]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {                   // 16 pixels per K-step: row ty = ks>>2, x = (ks&3)*16 + 8h ..
            const int cur = ks & 1;                          // compile-time after unrolling
            // Two waves share a SIMD's MFMA pipe; with the default (age-based) arbitration one runs ahead and then idles at
            // the block barrier while the other finishes alone (2 k of a block's 8.3 k cycles in the stamps).  Priority falls
            // as a wave advances through the block, so the wave that is behind wins the pipe and both arrive together.
            if (ks == 0) __builtin_amdgcn_s_setprio(3);
            else if (ks == 2) __builtin_amdgcn_s_setprio(2);
            else if (ks == 4) __builtin_amdgcn_s_setprio(1);
            else if (ks == 6) __builtin_amdgcn_s_setprio(0);
            if (ks + 1 < 8) read_ks(ks + 1, rb[cur ^ 1], rd[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);              // reads of ks + 1 stay ahead of the MFMAs of ks; nothing is hoisted further (144 accumulator registers)
            const u32x4 bfr = {rb[cur][0][0], rb[cur][0][1], rb[cur][1][0], rb[cur][1][1]};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // patch row ty + kh, pixels xk .. xk+11: six dwords of pixel pairs
                const unsigned d[6] = {rd[cur][kh * 3][0], rd[cur][kh * 3][1], rd[cur][kh * 3 + 1][0], rd[cur][kh * 3 + 1][1],
                                       rd[cur][kh * 3 + 2][0], rd[cur][kh * 3 + 2][1]};
                const u32x4 a0 = {d[0], d[1], d[2], d[3]};
                const u32x4 a1 = {__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                                  __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d[4], d[3], 16)};
                const u32x4 a2 = {d[1], d[2], d[3], d[4]};
                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, bfr), acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, bfr), acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2), __builtin_bit_cast(bf16x8_t, bfr), acc[kh * 3 + 2], 0, 0, 0);
            }
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

}  // namespace

// bf16 only; Cin % 64 == 0, Cout % 8 == 0
static thread_local unsigned long long* g_wgrad_dbg = nullptr;
extern "C" int lxo_wgrad_debug(unsigned long long* buf) { g_wgrad_dbg = buf; return 0; }
int lxo_launch_conv_wgrad(const GemmTN& p, hipStream_t s) {
    if (!p.conv || p.Cin % 64 || p.J % 8) return -2;
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
    const int per_split = cdiv(nblocks, nsplit);
    nsplit = cdiv(nblocks, per_split);
    GemmTN q = p;
    q.nbatch = tiles;
    q.dbg = g_wgrad_dbg;
    // splits are dealt to the 8 XCDs in turn: round the split count up to a multiple of 8 (empty ranges return at once)
    const int nsplit8 = (nsplit + 7) / 8 * 8;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles * nsplit8), dim3(WTHREADS), 2 * WSTAGE, s, q, tiles_co, tiles_x, tiles_y, nblocks, per_split);
    return (int)hipGetLastError();
}
