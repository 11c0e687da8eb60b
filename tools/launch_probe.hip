// Where do the 4.5-5 us of a decoder-loop launch go?  Stand-alone probe (no torch, no liblxo):
// dependent chains of N launches on one stream, timed host-paired (hipEvent pair + wall clock),
// with the host enqueue time reported separately.
//   chain A: empty kernel, 1 WG            chain B: empty kernel, 256 WGs
//   chain C: 128 WGs x 256 threads, each thread reads 9 f32x4 written by the previous launch
//            (an lstm_fwd-shaped body: loads -> a little math -> one store)
//   chain D: chain C captured into a hipGraph and replayed
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o tools/build/launch_probe
// run:   tools/build/launch_probe [N]      (also try HIP_FORCE_DEV_KERNARG=0/1)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Fat { const float* p; int n; long long stride; int ld; int pad[20]; };   // a ~112-byte by-value argument like Slabs + Drop

__global__ void empty_kernel(Fat f, int x) { (void)f; (void)x; }

__global__ __launch_bounds__(256) void body_kernel(const float* __restrict__ in, float* __restrict__ out, Fat f, int nslab, long long stride) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 t[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) t[s] = (s < nslab) ? *reinterpret_cast<const float4*>(in + s * stride + i) : acc;
#pragma unroll
    for (int s = 0; s < 9; ++s) { acc.x += t[s].x; acc.y += t[s].y; acc.z += t[s].z; acc.w += t[s].w; }
    acc.x = tanhf(acc.x) * 0.1f; acc.y = tanhf(acc.y) * 0.1f; acc.z = tanhf(acc.z) * 0.1f; acc.w = tanhf(acc.w) * 0.1f;
#pragma unroll
    for (int s = 0; s < 9; ++s) if (s < nslab) *reinterpret_cast<float4*>(out + s * stride + i) = acc;
    (void)f;
}

// H: how fast can N workgroups each pull `shared_bytes` that ALL of them read (an A operand) plus `priv_bytes` of their own
// (a weight slice)?  All loads of a thread are issued before the first use.
template <int NSH, int NPR>
__global__ __launch_bounds__(256) void pull_kernel(const float4* __restrict__ shared, const float4* __restrict__ priv, float* __restrict__ out) {
    float4 a[NSH > 0 ? NSH : 1], b[NPR > 0 ? NPR : 1];
#pragma unroll
    for (int j = 0; j < NSH; ++j) a[j] = shared[j * 256 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < NPR; ++j) b[j] = priv[((long long)blockIdx.x * NPR + j) * 256 + threadIdx.x];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NSH; ++j) s += a[j].x + a[j].y + a[j].z + a[j].w;
#pragma unroll
    for (int j = 0; j < NPR; ++j) s += b[j].x + b[j].y + b[j].z + b[j].w;
    if (s == 12345.678f) out[blockIdx.x] = s;
}
// rewrites the shared region (as the producer kernel of a chain would), 32 workgroups
__global__ __launch_bounds__(256) void rewrite_kernel(float4* __restrict__ shared, int n16, float v) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) shared[i] = make_float4(v, v, v, v);
}

// I: epilogue-store shapes.  N workgroups x 256 threads; thread (row = tid/4, part = tid%4) writes `reps` scalars.
//   frag = 1: out[rep][row][blockIdx*4 + part]      (row pitch 2048 floats: 16-byte fragments of lines shared by 8 workgroups)
//   frag = 0: out[rep][blockIdx][row*4 + part]      (each workgroup writes one contiguous 1 KB block per rep)
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, int reps, int frag, float v) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    for (int rep = 0; rep < reps; ++rep) {
        float* base = out + (long long)rep * 64 * 2048;
        if (frag) base[(long long)row * 2048 + blockIdx.x * 4 + part] = v + rep;
        else base[(long long)blockIdx.x * 256 + row * 4 + part] = v + rep;
    }
}

// spins for ~ms milliseconds (s_memtime runs at 100 MHz)
__global__ void blocker_kernel(long long ticks, int* sink) {
    const long long t0 = __builtin_readcyclecounter();
    long long t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(32); t = __builtin_readcyclecounter(); }
    if (sink && ticks < 0) *sink = (int)t;
}

// device-only time of a dependent chain: everything is enqueued while a blocker kernel holds the stream, so the host's
// enqueue cost is off the clock
template <typename F>
static void blocked_chain(const char* name, hipStream_t st, int N, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(blocker_kernel, dim3(1), dim3(64), 0, st, 40000000LL, (int*)nullptr);   // generous: ends when the counter says so
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipEventRecord(e1, st));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        auto t2 = std::chrono::steady_clock::now();
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s rep %d: DEVICE %.2f us/launch (enqueue took %.1f ms, blocker+chain wall %.1f ms)\n", name, rep, ms * 1e3 / N,
               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <typename F>
static void chain(const char* name, hipStream_t st, int N, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipEventRecord(e1, st));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        auto t2 = std::chrono::steady_clock::now();
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        const double wall = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
        printf("%-44s rep %d: event %.2f us/launch, wall %.2f us/launch, host enqueue %.2f us/launch\n", name, rep, ms * 1e3 / N, wall, enq);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2000;
    const char* kv = getenv("HIP_FORCE_DEV_KERNARG");
    printf("launch_probe N=%d HIP_FORCE_DEV_KERNARG=%s\n", N, kv ? kv : "(unset)");
    hipStream_t s_plain, s_nb;
    CK(hipStreamCreate(&s_plain));
    CK(hipStreamCreateWithPriority(&s_nb, hipStreamNonBlocking, 0));      // what torch.cuda.current_stream() pool streams are
    const long long stride = 128 * 256 * 4;                               // floats per slab
    float *a, *b;
    CK(hipMalloc(&a, stride * 9 * sizeof(float))); CK(hipMalloc(&b, stride * 9 * sizeof(float)));
    CK(hipMemset(a, 0, stride * 9 * sizeof(float))); CK(hipMemset(b, 0, stride * 9 * sizeof(float)));
    Fat f = {a, 8, stride, 2048, {0}};
    hipStream_t streams[3] = {nullptr, s_plain, s_nb};
    const char* sn[3] = {"null-stream", "hipStreamCreate", "nonblocking+prio"};
    for (int k = 0; k < 3; ++k) {
        hipStream_t st = streams[k];
        char nm[128];
        snprintf(nm, sizeof nm, "A empty 1 WG        [%s]", sn[k]);
        chain(nm, st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, f, 0); });
        snprintf(nm, sizeof nm, "B empty 256 WG      [%s]", sn[k]);
        chain(nm, st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, f, 0); });
        snprintf(nm, sizeof nm, "C body 128 WG 9 slab[%s]", sn[k]);
        chain(nm, st, N, [&](int i) { hipLaunchKernelGGL(body_kernel, dim3(128), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, f, 9, stride); });
        snprintf(nm, sizeof nm, "C1 body 128 WG 1 slab[%s]", sn[k]);
        chain(nm, st, N, [&](int i) { hipLaunchKernelGGL(body_kernel, dim3(128), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, f, 1, stride); });
    }
    // E: device-only cost of the chains (host enqueue hidden behind a blocker kernel)
    {
        hipStream_t st = s_nb;
        blocked_chain("E empty 1 WG (device only)", st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, f, 0); });
        blocked_chain("E empty 256 WG (device only)", st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, f, 0); });
        blocked_chain("E body 128 WG 9 slab (device only)", st, N, [&](int i) { hipLaunchKernelGGL(body_kernel, dim3(128), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, f, 9, stride); });
        blocked_chain("E body 128 WG 1 slab (device only)", st, N, [&](int i) { hipLaunchKernelGGL(body_kernel, dim3(128), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, f, 1, stride); });
    }
    // H: pull bandwidth of N workgroups (device-only; each timed launch is preceded by a rewrite of the shared region)
    {
        hipStream_t st = s_nb;
        float4 *sh, *pr; float* o;
        CK(hipMalloc(&sh, 512 << 10)); CK(hipMalloc(&pr, 256LL * 512 * 1024)); CK(hipMalloc(&o, 4096));
        CK(hipMemset(sh, 0, 512 << 10)); CK(hipMemset(pr, 0, 256LL * 512 * 1024));
        const int NR = 300;
        blocked_chain("H0 rewrite 128 KB only (32 WG)", st, NR, [&](int i) { hipLaunchKernelGGL(rewrite_kernel, dim3(32), dim3(256), 0, st, sh, 8192, (float)i); });
        const int grids[5] = {16, 32, 64, 128, 256};
        for (int gi = 0; gi < 5; ++gi) {
            const int G = grids[gi];
            char nm[128];
#define PULL(NSH, NPR) \
            snprintf(nm, sizeof nm, "H rewrite + pull shared %3d KB priv %3d KB x %3d WG", NSH * 4, NPR * 4, G); \
            blocked_chain(nm, st, NR, [&](int i) { hipLaunchKernelGGL(rewrite_kernel, dim3(32), dim3(256), 0, st, sh, 8192, (float)i); \
                                                   hipLaunchKernelGGL((pull_kernel<NSH, NPR>), dim3(G), dim3(256), 0, st, sh, pr, o); });
            PULL(32, 0) PULL(0, 8) PULL(32, 8) PULL(16, 4) PULL(0, 40)
#undef PULL
        }
    }
    // I: fragmented vs contiguous epilogue stores, 128 workgroups, 8 output arrays of [64][2048] floats
    {
        hipStream_t st = s_nb;
        float* o;
        CK(hipMalloc(&o, 8LL * 64 * 2048 * 4)); CK(hipMemset(o, 0, 8LL * 64 * 2048 * 4));
        blocked_chain("I stores 128 WG x 8 arrays, 16-B fragments", st, 500, [&](int i) { hipLaunchKernelGGL(store_kernel, dim3(128), dim3(256), 0, st, o, 8, 1, (float)i); });
        blocked_chain("I stores 128 WG x 8 arrays, contiguous 1 KB", st, 500, [&](int i) { hipLaunchKernelGGL(store_kernel, dim3(128), dim3(256), 0, st, o, 8, 0, (float)i); });
        blocked_chain("I stores 128 WG x 1 array, 16-B fragments", st, 500, [&](int i) { hipLaunchKernelGGL(store_kernel, dim3(128), dim3(256), 0, st, o, 1, 1, (float)i); });
        blocked_chain("I stores 128 WG x 1 array, contiguous 1 KB", st, 500, [&](int i) { hipLaunchKernelGGL(store_kernel, dim3(128), dim3(256), 0, st, o, 1, 0, (float)i); });
    }
    // F: hipModuleLaunchKernel (no host-pointer -> function lookup) enqueue cost
    {
        hipFunction_t fn;
        CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(&empty_kernel)));
        int x = 0;
        void* args[] = {&f, &x};
        hipStream_t st = s_nb;
        chain("F empty 1 WG hipModuleLaunchKernel", st, N, [&](int) { CK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, args, nullptr)); });
    }
    // G: two host threads, one stream each: does the enqueue rate scale?
    {
        hipStream_t s2;
        CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, 0));
        auto t0 = std::chrono::steady_clock::now();
        std::thread th([&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2, f, 0); });
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s_nb, f, 0);
        th.join();
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        printf("G two threads x two streams: %.2f us per launch per thread (aggregate %.2f us/launch)\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * N));
    }
    // D: the body chain as a graph
    {
        hipStream_t st = s_nb;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(body_kernel, dim3(128), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, f, 9, stride);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            auto t1 = std::chrono::steady_clock::now();
            printf("%-44s rep %d: wall %.2f us/launch\n", "D body chain as hipGraph [nonblocking+prio]", rep, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
    }
    return 0;
}
