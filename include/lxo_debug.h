/* Measurement hooks of liblxo.so: in-kernel phase timestamps.
 *
 * NOT part of the drop-in boundary (include/lxo.h is): these five entry points exist so that bench.py and tools/ can
 * report WHERE a kernel of the shipped library spends its time (the per-phase microseconds of the two decoder chains in
 * the bench line's `roofline_attention*` keys; profiles/r0x_*_stamps.txt).  They are exported from the production
 * library because the stamps must come from the kernels the benchmark runs, not from a second build of them.
 *
 * Every hook arms the calling HOST THREAD: the next launches of that kernel family from this thread write
 * 64-bit timestamps (the 100 MHz wall clock / the shader cycle counter, see the tool that reads them) into `buf`
 * (DEVICE memory owned by the caller, sized as the tool documents); NULL disarms.  A kernel launched with a NULL
 * buffer takes one wave-uniform branch per stamp site and writes nothing.  No hook changes a numeric result.
 *
 * There is no fault-injection hook: tests exercise the chains' fall-back by writing the error word of ws region
 * "xdec_sync" themselves (tests/test_gpu_xdec.py).
 */
#ifndef LXO_DEBUG_H
#define LXO_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* forward / backward decoder chain (csrc/xdec.hip): buf = [256 workgroups][T][16] stamps at the phase boundaries of every step
 * (tools/xdec_stamps.py, tools/xdec_stamps_bwd.py, bench.py: chain_phases) */
int lxo_xdec_debug(unsigned long long* buf);
int lxo_xdec_debug_bwd(unsigned long long* buf);
/* implicit-GEMM conv (csrc/conv_igemm.hip): per-workgroup prologue / slice / epilogue stamps + HW_ID (tools/conv_stamps.py, conv_timeline.py) */
int lxo_conv_debug(unsigned long long* buf);
/* conv weight gradient (csrc/conv_wgrad.hip): per pixel block and epilogue (tools/wgrad_stamps.py) */
int lxo_wgrad_debug(unsigned long long* buf);
/* fused recurrent-step kernels (csrc/rstep.hip): launches whose epilogue == epi are stamped (tools/rstep_stamps.py); epi < 0 disarms */
int lxo_rstep_debug(unsigned long long* buf, int epi);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
