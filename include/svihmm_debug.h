/* svihmm_debug.h -- measurement and test hooks of libsvihmm_hip.so.
 *
 * NOT the drop-in boundary (that is include/svihmm.h): nothing here replaces a reference interface.
 * bench.py uses the per-kernel HIP-event timing for the roofline object, the test suite uses
 * svihmm_set_variant to reach kernel generations the default dispatch would not pick for a shape, and
 * tools/ use both for A/B runs.  The symbols are exported by the same library so that the measured
 * binary IS the product binary; knobs that change what a call computes in a way that makes its results
 * invalid (skipped launches, knocked-out loads / stores) are compiled only with -DSVIHMM_MEASURE
 * (make -C pysvihmm_amd/csrc measure -> build_exp/libsvihmm_measure.so, never shipped).
 *
 * Versioning: SVIHMM_ABI_VERSION / svihmm_abi_version() cover include/svihmm.h ONLY (the symbols a caller of the
 * product binds; 3 since the hooks below left that header in round 5).  This header is not a stable interface: it
 * changes with the library build (round 6 added svihmm_svi_recoveries and variant codes 0 = 2 / 3, 4 = 3 / 5, 13 = 3),
 * and its only users are the repository's own tests, bench.py and tools/, which travel with the library.
 */
#ifndef SVIHMM_DEBUG_H
#define SVIHMM_DEBUG_H

#include "svihmm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement ------------------------------------------------------------------ */
/* When enabled, every kernel launched by the handle is bracketed by HIP events on
 * the handle's stream; svihmm_profile_read returns accumulated milliseconds and
 * launch counts per kernel slot since the last reset. */
#define SVIHMM_NKERN 12
/* on = 0: off; 1: every slot; SVIHMM_PROF_SLOTS | (1 << slot) | ...: those slots only (an event
 * pair between two dependent kernels costs a few microseconds of dispatch -- measured 0.055 ms on
 * the 3.15 ms bench step with all slots -- so a timed region that wants one kernel's duration
 * brackets only that kernel). */
#define SVIHMM_PROF_SLOTS 0x40000000
int svihmm_profile_enable(svihmm_ctx* h, int32_t on);
int svihmm_profile_reset(svihmm_ctx* h);
int svihmm_profile_read(svihmm_ctx* h, double ms_out[SVIHMM_NKERN],
                        int64_t count_out[SVIHMM_NKERN]);
const char* svihmm_kernel_name(int32_t slot);
/* The kernel function the slot's LAST launch dispatched, as rocprofv3 prints it (e.g.
 * "k_stats_mfma4<5, 2, 2, 3, true, false, double, double, 3>"; "" when unknown): bench.py checks it against
 * the kernel name in the committed profile before it quotes that profile's duration. */
const char* svihmm_last_kernel_name(svihmm_ctx* h, int32_t slot);
/* Selects the kernel generation for A/B measurement (0 = default/best; which < 24).
 * which 0 the resident SVI loop's dependency mechanism (0: device-side counters where kernels of two streams run
 *      side by side, else stream events; 1: stream events; 2: counters, and the loop behaves as if the device stopped
 *      running kernels concurrently at iteration 3 -- exercises the mid-loop switch to stream events; 3: counters, one gate
 *      of iteration 3 waits for a count that never comes with a 2 ms bound -- exercises the bounded-wait recovery)
 * | 1 statistics (2 double-buffered MFMA, else the pipelined MFMA kernels)
 * | 2 sweeps (1 wave-per-window, 2 log-domain MFMA, 3 scaled linear-domain MFMA)
 * | 3 emission row tiles per wave | 4 E-step launch structure (0: sweeps + statistics of minibatch-sized K = 64 batches in one
 *      fused launch where that is ahead, else one launch after the other; 1: never fused; 2: the two-stream pipeline of
 *      round 2; 3: the fused launch for every batch it can take, whatever its size or precision mode, WITH the emission
 *      tiles computed inside it where the batch's emission kernel is the 16-row fp64 one (D % 8 == 0, D <= 32) -- tests;
 *      4: as 0; 5: as 0 plus the emission tiles inside the launch -- measured, not ahead: tu_fused.hip, sweep_emission_ok;
 *      6: as 0, but the fp32 mode keeps float messages + its bf16 statistics kernel for minibatch-sized batches instead of
 *      the fused launch on fp64 messages behind float emission rows)
 * | 6 blocked scan for one long window (B = 1, Lm >= 2048, K <= 64; 1 = off)
 * | 9 automatic centring of the resident observations at upload (1 = off: c = 0)
 * | 12 barrier-free statistics GEMM with three LDS buffers (1 = off: the double-buffered kernel)
 * | 11 svihmm_allreduce_packed forms the sum in caller coordinates also at one rank (1 = on: the
 *      multi-rank path's coordinate round trip, exercised on a single GPU)
 * | 7 scaled sweeps' kernel family (K > 128: 1 one state tile per wave, 2 two tiles per wave for every K)
 * | 13 wide models' sweeps with 32 windows per workgroup (1 = off: 16); 2: NIW -> theta for 16 < D <= 32 by the builder that
 *      leaves half the wave idle (rounds 2-5) instead of k_niw_to_theta_wave32s -- same results bit for bit; 3: the
 *      split builder, but the resident loop's global step stays a launch of its own (k_svi_global_step, not merged
 *      into k_svi_step_theta32s)
 * | 16 the register-resident minibatch sweep (k_wave_linr, the sweep workgroups of k_sweep_stats) re-normalises its vector
 *      every fourth step where the transition expectations lie inside a float's range (1 = off: every step; same
 *      results bit for bit)
 * | 14 wide models' transition statistic in 128 x 64 blocks (1 = off: 64 x 64)
 * | 15 wide models' statistics GEMM forms q = ah bh scale itself (1 = off: separate posterior pass;
 *      2: a separate pass for every K -- valid results, slower)
 * | 10 statistics GEMM tiling (1: five feature tiles per wave for every shape) and, in the fp32 mode, its
 *      pipe (2: the fp32-input MFMA kernel instead of the three-term bf16 one; 3: the bf16 kernel also
 *      below its batch-size floor of 32 768 rows)
 * | 7 = 3: the LDS-broadcast one-wave minibatch sweep k_wave_lin instead of the register-resident k_wave_linr (fp64)
 *      / the four-wave k_wave_lin4 (fp32 mode)
 * Codes that make results INVALID exist only in a -DSVIHMM_MEASURE build of the library (make measure):
 * | 7 = 9: the scaled sweeps are skipped, the statistics read stale messages (tools/r4_overlap_probe.py);
 *   a product build rejects them with an error. */
int svihmm_set_variant(svihmm_ctx* h, int32_t which, int32_t value);
/* How often the current device-resident SVI loop left its device-side counters for stream events mid-way
 * (a gate's bounded wait ran out and the lost iterations were replayed, or the debug variants 0 = 2 / 3). */
int svihmm_svi_recoveries(svihmm_ctx* h, int32_t* out);

/* ---- diagnostics ------------------------------------------------------------------- */
/* One v_mfma_f64_16x16x4_f64 on A[16,4] x B[4,16] -> C[16,16] (operand-layout check). */
int svihmm_selftest_mfma(svihmm_ctx* h, const double* A16x4, const double* B4x16,
                         double* C16x16);

#ifdef __cplusplus
}
#endif
#endif /* SVIHMM_DEBUG_H */
