// hipsim runtime -- TEST INFRASTRUCTURE ONLY.  See shim/hip/hip_runtime.h.
// One OS thread; each GPU thread of the workgroup being interpreted is a fiber
// with its own stack.  A fiber runs until it reaches a rendezvous
// (__syncthreads / wave exchange) or returns; the scheduler releases a
// rendezvous when every live fiber of the workgroup (or wave) has arrived.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <vector>

extern "C" void hipsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipsim_switch,.-hipsim_switch
)");

namespace hipsim {
Idx g_threadIdx, g_blockIdx;
alignas(64) char g_dyn_lds[163840];
dim3 g_blockDim, g_gridDim;

enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber { void* sp; int state; };
static const size_t STACK = 256 * 1024;
static const int MAXT = 1024;
static char* g_stacks = nullptr;
static Fiber g_f[MAXT];
static int g_n = 0, g_cur = 0;
static void* g_sched_sp = nullptr;
static const std::function<void()>* g_body = nullptr;
static char g_slots[MAXT / 64][64][256];

static void set_tid(int t) {
    g_threadIdx.x = t % g_blockDim.x;
    g_threadIdx.y = (t / g_blockDim.x) % g_blockDim.y;
    g_threadIdx.z = t / (g_blockDim.x * g_blockDim.y);
}
static void yield() {
    hipsim_switch(&g_f[g_cur].sp, g_sched_sp);
    set_tid(g_cur);
}
static void entry() {
    (*g_body)();
    g_f[g_cur].state = DONE;
    hipsim_switch(&g_f[g_cur].sp, g_sched_sp);
    abort();
}
void syncthreads() { g_f[g_cur].state = WAIT_BLOCK; yield(); }
void wave_sync() { g_f[g_cur].state = WAIT_WAVE; yield(); }
unsigned lane() { return (unsigned)g_cur & 63u; }
char* slot(unsigned l) { return g_slots[g_cur >> 6][l & 63u]; }
unsigned long long live_mask() {
    unsigned long long m = 0;
    int w0 = g_cur & ~63;
    for (int i = 0; i < 64 && w0 + i < g_n; ++i)
        if (g_f[w0 + i].state != DONE) m |= 1ull << i;
    return m;
}

static void run_block() {
    for (int t = 0; t < g_n; ++t) {
        char* top = g_stacks + (size_t)(t + 1) * STACK;   // 16-byte aligned
        void** sp = (void**)(top - 64);
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;       // r15 r14 r13 r12 rbx rbp
        sp[6] = (void*)&entry;                             // return address
        sp[7] = nullptr;
        g_f[t].sp = sp;
        g_f[t].state = RUN;
    }
    for (;;) {
        for (int t = 0; t < g_n; ++t) {
            if (g_f[t].state != RUN) continue;
            g_cur = t;
            set_tid(t);
            hipsim_switch(&g_sched_sp, g_f[t].sp);
        }
        bool released = false, all_done = true;
        for (int w0 = 0; w0 < g_n; w0 += 64) {             // wave rendezvous
            int live = 0, waiting = 0;
            for (int i = w0; i < std::min(w0 + 64, g_n); ++i) {
                if (g_f[i].state != DONE) ++live;
                if (g_f[i].state == WAIT_WAVE) ++waiting;
            }
            if (live && waiting == live) {
                for (int i = w0; i < std::min(w0 + 64, g_n); ++i)
                    if (g_f[i].state == WAIT_WAVE) g_f[i].state = RUN;
                released = true;
            }
        }
        int live = 0, waiting = 0;
        for (int t = 0; t < g_n; ++t) {
            if (g_f[t].state != DONE) { ++live; all_done = false; }
            if (g_f[t].state == WAIT_BLOCK) ++waiting;
        }
        if (all_done) return;
        if (!released && live && waiting == live) {
            for (int t = 0; t < g_n; ++t)
                if (g_f[t].state == WAIT_BLOCK) g_f[t].state = RUN;
            released = true;
        }
        if (!released) {
            fprintf(stderr, "hipsim: deadlock (divergent barrier / wave op) in block (%u,%u,%u)\n",
                    g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
            for (int t = 0; t < g_n; ++t)
                if (g_f[t].state != DONE) { fprintf(stderr, "  thread %d state %d\n", t, g_f[t].state); break; }
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int n = (int)(block.x * block.y * block.z);
    if (n <= 0 || n > MAXT) { fprintf(stderr, "hipsim: bad block size %d\n", n); abort(); }
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, STACK * MAXT, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { perror("mmap"); abort(); }
    }
    g_blockDim = block; g_gridDim = grid; g_n = n; g_body = &body;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                g_blockIdx = {x, y, z};
                run_block();
            }
}
}  // namespace hipsim
