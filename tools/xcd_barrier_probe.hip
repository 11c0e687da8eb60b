// What does a barrier among the 32 workgroups of ONE XCD cost when nothing crosses the die?
// (the decoder's samples are independent through the recurrence, block b runs on XCD b % 8 and an
// XCD's L2 is the point of coherence for its own CUs -- so a chain of B/8 samples per XCD needs no agent-scope fence.)
//
// 256 workgroups (one per CU), rank in XCD = blockIdx / 8.  Per iteration: optional streaming phase (bytes per XCD), each
// workgroup publishes a 128-byte record with PLAIN stores, waits for its own stores (vmcnt(0)), arrives on its XCD's counter,
// polls it with sc1 loads (L1 bypassed, L2 served), then reads the 32 records of its XCD with sc1 loads and counts stale ones.
//   arrive 0: workgroup-scope atomic add (no sc1: executed in this XCD's L2)      arrive 1: agent-scope atomic add     2: no barrier     3: no atomic (flag line)
// Every spin is bounded (a stuck barrier sets a flag and the kernel runs on).
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier_probe.hip -o tools/build/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Res { unsigned long long ticks; unsigned stale, stuck, xcc_mismatch, pad; };

template <int ARRIVE, int CHECK>
__global__ __launch_bounds__(256) void xbar_kernel(unsigned* counters /* 8 x 64 words */, unsigned* records /* 256 x 32 words */,
                                                   const float4* stream, long long stream_f4_per_wg, int iters, Res* res, float* sink) {
    const int b = blockIdx.x, xcd = b & 7, rank = b >> 3, tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __shared__ unsigned s_stuck;
    if (tid == 0) { s_stuck = 0; if ((int)xcc != xcd) atomicAdd(&res[0].xcc_mismatch, 1u); }
    __syncthreads();
    unsigned* cnt = counters + xcd * 64;
    unsigned* myrec = records + (size_t)b * 32;
    unsigned stale = 0;
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (stream_f4_per_wg > 0) {
            const float4* p = stream + (size_t)b * stream_f4_per_wg;
            for (long long i = tid; i < stream_f4_per_wg; i += 256 * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const long long j = i + 256 * u; v[u] = p[j < stream_f4_per_wg ? j : i]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].w;
            }
        }
        if (tid < 32) myrec[tid] = (unsigned)it;                       // plain stores (L1 is write-through)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ARRIVE == 3) {
            // no atomic at all: every workgroup owns one word of its XCD's 128-byte flag line (plain store -> L2), wave 0 polls the
            // whole line with ONE sc1 load per lane (32 words) until every word has reached this iteration
            unsigned* flags = counters + 8 * 64 + xcd * 64;
            if (tid == 0) flags[rank] = (unsigned)it;
            if (tid < 64) {
                int spins = 0;
                for (;;) {
                    const unsigned v = __hip_atomic_load(flags + (tid & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64(v < (unsigned)it) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { s_stuck = 1; break; }
                }
            }
        } else if (tid == 0 && ARRIVE != 2) {
            if (ARRIVE == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = 32u * (unsigned)it;
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { s_stuck = 1; break; }
            }
        }
        __syncthreads();
        if (CHECK) {
            // every thread reads word (tid & 31) of record (tid >> 5) + 8 k of its XCD: 4 sc1 loads
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = (tid >> 5) + 8 * k;
                const unsigned v = __hip_atomic_load(records + (size_t)(r * 8 + xcd) * 32 + (tid & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stale += (v < (unsigned)it) ? 1u : 0u;
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0 && b == 0) res[0].ticks = t1 - t0;
    if (stale) atomicAdd(&res[0].stale, stale);
    if (tid == 0 && s_stuck) atomicAdd(&res[0].stuck, 1u);
    if (acc == 12345.678f) sink[b] = acc;
}

template <int ARRIVE, int CHECK>
static void run(const char* name, unsigned* counters, unsigned* records, const float4* stream, long long f4_per_wg, int iters, Res* res, float* sink) {
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(counters, 0, 16 * 64 * 4)); CK(hipMemset(records, 0, 256 * 32 * 4)); CK(hipMemset(res, 0, sizeof(Res)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((xbar_kernel<ARRIVE, CHECK>), dim3(256), dim3(256), 0, 0, counters, records, stream, f4_per_wg, iters, res, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        Res h; CK(hipMemcpy(&h, res, sizeof(Res), hipMemcpyDeviceToHost));
        printf("%-64s rep %d: %.3f us/iter (events), %.3f us/iter (wall_clock64 of WG 0), stale %u, stuck %u, xcc!=b%%8 %u\n", name, rep,
               ms * 1e3 / iters, (double)h.ticks * 0.01 / iters, h.stale, h.stuck, h.xcc_mismatch);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *counters, *records; Res* res; float* sink; float4* stream;
    const long long MB = 1 << 20;
    const long long stream_bytes = 80 * MB;                                     // 10 MB per XCD
    CK(hipMalloc(&counters, 16 * 64 * 4)); CK(hipMalloc(&records, 256 * 32 * 4)); CK(hipMalloc(&res, sizeof(Res))); CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&stream, stream_bytes)); CK(hipMemset(stream, 0, stream_bytes));
    const long long f4 = stream_bytes / 16 / 256;                               // per workgroup: 320 KB
    run<0, 0>("XCD barrier, L2 (workgroup-scope) arrive, no payload", counters, records, stream, 0, iters, res, sink);
    run<1, 0>("XCD barrier, agent-scope arrive, no payload", counters, records, stream, 0, iters, res, sink);
    run<0, 1>("XCD barrier, L2 arrive, 128-B record per WG re-read (sc1)", counters, records, stream, 0, iters, res, sink);
    run<1, 1>("XCD barrier, agent arrive, 128-B record per WG re-read (sc1)", counters, records, stream, 0, iters, res, sink);
    run<3, 0>("XCD barrier, flag line (store + one sc1 load), no payload", counters, records, stream, 0, iters, res, sink);
    run<3, 1>("XCD barrier, flag line, 128-B record per WG re-read (sc1)", counters, records, stream, 0, iters, res, sink);
    const int si = iters / 10 > 0 ? iters / 10 : 1;
    run<2, 0>("10 MB stream per XCD, NO barrier", counters, records, stream, f4, si, res, sink);
    run<0, 1>("10 MB stream per XCD + L2 arrive + records", counters, records, stream, f4, si, res, sink);
    run<1, 1>("10 MB stream per XCD + agent arrive + records", counters, records, stream, f4, si, res, sink);
    run<3, 1>("10 MB stream per XCD + flag line + records", counters, records, stream, f4, si, res, sink);
    run<2, 0>("2.5 MB stream per XCD, NO barrier", counters, records, stream, f4 / 4, si, res, sink);
    run<0, 1>("2.5 MB stream per XCD + L2 arrive + records", counters, records, stream, f4 / 4, si, res, sink);
    run<3, 1>("2.5 MB stream per XCD + flag line + records", counters, records, stream, f4 / 4, si, res, sink);
    return 0;
}
