// Counter-based dropout mask shared by every kernel that applies or re-applies tf.nn.dropout of
// attention_cell.py:72,83 (oracle/ref_model.py drop_mask is the same arithmetic).
#pragma once
#include "decoder_kernels.h"

// dropout mask bit: splitmix64 of the element counter
__device__ __forceinline__ float drop_scale(const Drop& d, unsigned which, int r, int c, int width) {
    if (d.thr == 0u) return 1.f;
    unsigned long long z = ((unsigned long long)((long long)d.t * d.rows_total + d.row0 + r)) * (unsigned long long)width + (unsigned long long)c;
    z += 0x9E3779B97F4A7C15ull * ((((unsigned long long)d.seed) << 2) | which);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((unsigned)(z >> 40) < d.thr) ? d.inv_keep : 0.f;
}
