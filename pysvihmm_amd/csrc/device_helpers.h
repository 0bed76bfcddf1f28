// device_helpers.h -- wave / row reductions, fp64 exp/log, DPP helpers shared by the kernels.
// Part of libsvihmm_hip.so; included by every translation unit (host.h lists them).
#pragma once

// ------------------------------------------------------------------------------------
//  device helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double nan_to_num(double v) {
  // np.nan_to_num: NaN -> 0, +-inf -> +-DBL_MAX  (hmmbase.py:220)
  if (v != v) return 0.0;
  if (isinf(v)) return v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
  return v;
}
// global row g=(b,t) of the flattened window batch -> obs row
__device__ __forceinline__ int64_t obs_row(const int64_t* __restrict__ starts, int Lm,
                                           int64_t g) {
  int64_t b = g / Lm;
  return starts[b] + (g - b * Lm);
}

// fp64 transcendentals for the fused sweeps.  On gfx950 fp64 MFMA and fp64 VALU share one
// pipe (tools/peak_probe.py: their times add), so every fp64 VALU instruction in the time
// loop costs matrix throughput; ocml's log() alone is ~90 of them.  These are plain
// range-reduction + Horner versions, accurate to ~2 ulp (tests compare against the oracle).
__device__ __forceinline__ double fmax_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // no NaN canonicalisation pair
  return r;
}
__device__ __forceinline__ double fast_exp(double x) {
  x = fmax_raw(x, -800.0);                       // also maps -inf to exp -> 0
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.0 / 479001600.0;                  // Taylor to r^12: |r| <= 0.3466 -> 1.7e-16
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}
// fast_exp with its constants held in caller-provided VGPRs.  hipcc otherwise rematerialises
// every fp64 literal (two v_mov_b32 each) at each use: in a 32-exp epilogue that is ~800 moves.
struct ExpConsts { double c[15]; };
__device__ __forceinline__ void exp_consts_init(ExpConsts& k) {
  const double v[15] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
                        1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0,
                        1.0 / 6.0, 0.5, 1.4426950408889634074, -6.93147180369123816490e-01,
                        -1.90821492927058770002e-10, -800.0};
#pragma unroll
  for (int i = 0; i < 15; ++i) { k.c[i] = v[i]; asm volatile("" : "+v"(k.c[i])); }
}
__device__ __forceinline__ double fast_exp_k(double x, const ExpConsts& k) {
  x = fmax_raw(x, k.c[14]);
  const double n = __builtin_rint(x * k.c[11]);
  double r = fma(n, k.c[12], x);
  r = fma(n, k.c[13], r);
  double p = k.c[0];
#pragma unroll
  for (int i = 1; i <= 10; ++i) p = fma(p, r, k.c[i]);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}
__device__ __forceinline__ double fast_log(double x) {   // x >= 0, finite
  int e;
  double m = frexp(x, &e);                       // m in [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;                        // z <= 0.0295
  double p = 1.0 / 23.0;
  p = fma(p, z, 1.0 / 21.0);
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  // log(m) = 2s + 2s*z*p ; log(x) = e*ln2_hi + (log(m) + e*ln2_lo)
  const double two_s = s + s;
  const double ed = (double)e;
  const double t = fma(two_s * z, p, fma(ed, 1.90821492927058770002e-10, two_s));
  const double r = fma(ed, 6.93147180369123816490e-01, t);
  return x > 0.0 ? r : -INFINITY;               // log(0) = -inf (an unreachable state)
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// all-lanes reductions over each row of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror)
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);
  v += dpp_mov_f64<0x4E>(v);
  v += dpp_mov_f64<0x141>(v);
  v += dpp_mov_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ double row16_max(double v) {
  v = fmax_raw(v, dpp_mov_f64<0xB1>(v));
  v = fmax_raw(v, dpp_mov_f64<0x4E>(v));
  v = fmax_raw(v, dpp_mov_f64<0x141>(v));
  v = fmax_raw(v, dpp_mov_f64<0x140>(v));
  return v;
}
// wave-wide reductions without LDS round trips: DPP within rows of 16, v_readlane across rows
// (result uniform).  ~25 instructions instead of six ds_bpermute round trips.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Reading a lane that is inactive where the value was produced is undefined, and the compiler
// is free to sink "x = valid ? f(..) : 0" plus everything that consumes x only in valid lanes
// into the divergent region -- a v_readlane of x then sees garbage in the other lanes (seen
// with ragged K in k_fb_wave).  pin_all_lanes() is an opaque volatile no-op: the value must
// exist in every lane at this point of the (uniform) control flow.
__device__ __forceinline__ void pin_all_lanes(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ double wave_sum_dpp(double v) {
  pin_all_lanes(v);
  v = row16_sum(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max_dpp(double v) {
  pin_all_lanes(v);
  v = row16_max(v);
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double wave64_max_fast(double v) {
  v = row16_max(v);
  v = fmax_raw(v, __shfl_xor(v, 16, 64));
  v = fmax_raw(v, __shfl_xor(v, 32, 64));
  return v;
}

// digamma (recurrence to x >= 10 + asymptotic series) and the triangular feature index, shared by the
// theta builders (kernels_emission.h) and the SVI loop's kernels (kernels_svi.h)
__device__ __forceinline__ double digamma_d(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  const double t = f * (-1.0 / 12 + f * (1.0 / 120 + f * (-1.0 / 252 + f * (1.0 / 240 +
                   f * (-1.0 / 132 + f * (691.0 / 32760 + f * (-1.0 / 12)))))));
  return r + log(x) - 0.5 / x + t;
}
__device__ __forceinline__ int feat_index_d(int a, int b, int D) {
  return a * (D + 1) - a * (a - 1) / 2 + (b - a);
}

// ------------------------------------------------------------------------------------
//  Device-side dependencies of the resident SVI loop (round 5).  A stream-order event between two kernels
//  of an iteration's chain costs ~7 us of dispatch (record) or 4-10 us (a wait that was satisfied long
//  before); the loop's cross-stream edges -- global step -> globals kernel, globals -> sweeps, theta ->
//  ELBO kernels, ELBO kernels -> next global step -- and its timing marks are therefore carried by
//  monotonic counters in HBM: a producer's workgroups ARRIVE (agent-scope release) when their stores are
//  done, a consumer's workgroups GATE on the expected total (agent-scope acquire; bounded spin -- after
//  60 s a timeout raises the status word instead of hanging the queue), side streams start their kernels behind a
//  one-wave k_svi_gate so that nothing squats on a CU while it waits, and the iteration boundaries are
//  wall_clock64() stamps written by the kernels themselves.
// ------------------------------------------------------------------------------------
#define SVI_SYNC_TIMEOUT (1 << 22)
// upper limit of a gate's bound in ticks of the 100 MHz device wall clock (60 s).  The host passes each gate its own
// bound (SviSync::ticks): a gate launched early legitimately waits for as long as the iteration in front of it
// runs, so the bound follows the loop's measured iteration period (svihmm_hip.hip, svi_gate_ticks) -- 64 periods,
// at least 50 ms, and this limit while no period is known yet.
#define SVI_SYNC_TICKS 6000000000ull
// Returns true when the kernel may run its body (uniform over the workgroup).  false: the loop is dead -- this gate
// or an earlier one gave up; the caller skips the body but still arrives, so that nothing behind it waits.
__device__ __forceinline__ bool svi_gate(const SviSync& sy) {
  if (sy.early && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    __hip_atomic_fetch_add(sy.early, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!sy.gate && !sy.dead) return true;
  int failed = 0;
  if (threadIdx.x == 0) {
    // (relaxed polls, one acquire fence at the end: an acquire load invalidates the caches on every poll)
    // (the first poll and the flag travel together: one L2 round trip on the path that finds the count already there)
    unsigned seen = sy.gate ? __hip_atomic_load(sy.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (sy.dead && __hip_atomic_load(sy.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) failed = 1;
    if (sy.gate && !failed) {
      const unsigned long long t0 = wall_clock64();
      const unsigned long long bound = sy.ticks ? sy.ticks : SVI_SYNC_TICKS;
      unsigned n = 0;
      // (counters and targets are 32-bit and only ever grow: compared by their signed difference, so a long loop may wrap them)
      for (; (int)(seen - sy.gate_tgt) < 0; seen = __hip_atomic_load(sy.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (++n < 256u) { __builtin_amdgcn_s_sleep(4); continue; }       // ~0.1 us naps first, ~2 us later
        __builtin_amdgcn_s_sleep(64);
        if ((n & 63u) == 0u) {
          if (sy.dead && __hip_atomic_load(sy.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = 1; break; }
          if (wall_clock64() - t0 > bound) {
            if (sy.dead) __hip_atomic_store(sy.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sy.status) atomicMax(sy.status, SVI_SYNC_TIMEOUT);
            failed = 1;
            break;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  return __syncthreads_or(failed) == 0;
}
// a kernel on the read-after-write path of the loop's state that could not run: the iteration poison_val stands for
// (and every later one) must not reach the state
__device__ __forceinline__ void svi_poison(const SviSync& sy) {
  if (sy.poison && threadIdx.x == 0) atomicMax(sy.poison, sy.poison_val);
}
// the global-step kernels: skip from the poisoned iteration on (uniform: see SviSync::poison); the kernel's own gate
// guards a write-after-read hazard only (the previous iteration's ELBO kernels still reading what the step rewrites),
// so a gate that gives up does not stop the step -- it costs that iteration's ELBO entry, not the state
__device__ __forceinline__ bool svi_step_gate(const SviSync& sy) {
  // (the poison word is requested by every thread up front and looked at AFTER the gate: its round trip rides with the
  //  gate's own loads instead of standing in front of them -- 3 us of a 6 us kernel otherwise)
  unsigned poisoned = 0u;
  if (sy.poison) poisoned = __hip_atomic_load(sy.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)svi_gate(sy);
  return poisoned == 0u;
}
__device__ __forceinline__ void svi_arrive(const SviSync& sy) {
  if (!sy.arrive) return;
  __syncthreads();                         // every thread's stores are issued and waited for (workgroup scope)
  if (threadIdx.x == 0) {
    // (release: the workgroup's stores -- ordered before this thread by the barrier -- are written back first)
    const unsigned before = __hip_atomic_fetch_add(sy.arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (sy.stamp && before + 1u == sy.stamp_at) *sy.stamp = wall_clock64();
  }
}

// entry e of the transition factor (shared by the families' global-step kernels)
__device__ __forceinline__ void svi_tran_step(int e, int K, const double* __restrict__ packed,
                                              const double* __restrict__ prior_tran, double* __restrict__ var_tran,
                                              double rho, double bA, double nwin, double* __restrict__ ada_G) {
  if (e >= K * K) return;
  const double a_inter = packed[e] + nwin * (prior_tran[e] - 1.0);
  const double nat_old = var_tran[e] - 1.0;
  if (ada_G) {
    // AdaGrad-scaled step of the transition factor (hmmsgd_metaobs.py:1036-1040): the
    // accumulated squared natural parameters set a per-entry step 1 / G^(1/4); rho is not used
    const double g = ada_G[e] + nat_old * nat_old;
    ada_G[e] = g;
    const double am = sqrt(sqrt(g));
    var_tran[e] = ((1.0 - 1.0 / am) * nat_old + (bA * a_inter) / am) + 1.0;
  } else {
    var_tran[e] = ((1.0 - rho) * nat_old + rho * (bA * a_inter)) + 1.0;
  }
}
