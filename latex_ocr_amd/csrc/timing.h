// Measurement aid (off by default): HIP-event brackets around individual launches INSIDE a real training step, on the
// stream the kernels are launched on, so that bench.py's roofline entries come from in-step kernel durations rather
// than from back-to-back micro-timings on synthetic operands.  Per host thread; enabled by lxo_timing_enable(1).
#pragma once
#include <hip/hip_runtime.h>

// -> slot (>= 0) when timing is enabled, -1 otherwise.  family / name must be string literals.
int lxo_timer_begin(const char* family, const char* name, double work, hipStream_t st);
void lxo_timer_end(int slot, hipStream_t st);
bool lxo_timer_on();          // per-launch brackets are being recorded on this host thread (launches are then kept on ONE stream)

struct LxoTimed {
    int slot; hipStream_t st;
    LxoTimed(const char* family, const char* name, double work, hipStream_t s) : slot(lxo_timer_begin(family, name, work, s)), st(s) {}
    ~LxoTimed() { if (slot >= 0) lxo_timer_end(slot, st); }
};
