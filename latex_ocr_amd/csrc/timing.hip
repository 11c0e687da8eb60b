// In-step kernel timing registry (see timing.h, include/lxo.h lxo_timing_*).
#include "timing.h"
#include "lxo.h"
#include <string.h>
#include <vector>

namespace {
struct Slot { const char* family; const char* name; double work; hipEvent_t e0, e1; bool ended; };
struct Registry {
    bool on = false;
    std::vector<Slot> slots;          // events are created on first use and recycled by lxo_timing_enable(1)
    size_t used = 0;
};
thread_local Registry g_reg;
const size_t kMaxSlots = 8192;
}  // namespace

int lxo_timer_begin(const char* family, const char* name, double work, hipStream_t st) {
    Registry& R = g_reg;
    if (!R.on || R.used >= kMaxSlots) return -1;
    if (R.used == R.slots.size()) {
        Slot s; memset(&s, 0, sizeof(s));
        if (hipEventCreate(&s.e0) != hipSuccess || hipEventCreate(&s.e1) != hipSuccess) return -1;
        R.slots.push_back(s);
    }
    Slot& s = R.slots[R.used];
    s.family = family; s.name = name; s.work = work; s.ended = false;
    if (hipEventRecord(s.e0, st) != hipSuccess) return -1;
    return (int)R.used++;
}
void lxo_timer_end(int slot, hipStream_t st) {
    Registry& R = g_reg;
    if (slot < 0 || (size_t)slot >= R.used) return;
    if (hipEventRecord(R.slots[slot].e1, st) == hipSuccess) R.slots[slot].ended = true;
}

bool lxo_timer_on() { return g_reg.on; }

extern "C" int lxo_timing_enable(int on) {
    g_reg.on = on != 0;
    if (on) g_reg.used = 0;
    return 0;
}
extern "C" int lxo_timing_count(void) { return (int)g_reg.used; }
extern "C" int lxo_timing_get(int i, const char** family, const char** name, double* work, float* ms) {
    Registry& R = g_reg;
    if (i < 0 || (size_t)i >= R.used || !R.slots[i].ended) return -1;
    Slot& s = R.slots[i];
    if (hipEventSynchronize(s.e1) != hipSuccess) return -2;
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.e0, s.e1) != hipSuccess) return -2;
    if (family) *family = s.family;
    if (name) *name = s.name;
    if (work) *work = s.work;
    if (ms) *ms = t;
    return 0;
}
