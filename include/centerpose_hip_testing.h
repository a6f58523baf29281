/*
 * centerpose_hip_testing.h -- test hooks of libcenterpose_hip.so.  NOT part of the product ABI (centerpose_hip.h): nothing
 * in centerpose_amd/lib or bench.py's timed regions calls these; tests/ and the A/B tools under tools/ do.
 */
#ifndef CENTERPOSE_HIP_TESTING_H
#define CENTERPOSE_HIP_TESTING_H

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel SELECTION switches (process-global; 0 = the engine's own choice).  Every bit picks another implementation of the
 * same layer among the ones the library ships -- per-head launches instead of the grouped head launch, the element-wise
 * split-K epilogue instead of the quad form, dcn16p instead of dcn16s, the generic DCN kernel instead of the fused ones,
 * 128-row tiles for small launches, ... (the list is next to g_dbg in csrc/engine.hip) -- so that the parity tests can compare
 * two implementations on one input.  Every combination computes the layer correctly (to the summation-order round-off the
 * tests state); the timing ablations of earlier rounds, under which results were wrong, no longer exist in the library.
 * Bit 512 (operands of the f16x3 products used without the |max| pre-scale) is correct only for inputs inside binary16's
 * range and exists for the range-safety tests. */
int cp_set_debug(int flags);

#ifdef __cplusplus
}
#endif
#endif
