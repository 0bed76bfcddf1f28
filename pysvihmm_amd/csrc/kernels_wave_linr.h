// kernels_wave_linr.h -- the register-resident one-wave scaled sweep of minibatch-sized batches (k_wave_linr, round 5)
// and what it shares with the other scaled sweeps (arithmetic-type traits, the initial message).  Its own header since
// round 6: the fused sweep + statistics kernel of tu_stats.hip (kernels_fused.h) runs the same body in its sweep
// workgroups.  Device templates and inline functions only -- safe to include from several translation units.
#pragma once
#include <type_traits>
#ifndef LN2_D
#define LN2_D 0.69314718055994530942
#endif
#define LOG2E_D 1.4426950408889634074
#define LN2_HI_D 6.93147180369123816490e-01
#define LN2_LO_D 1.90821492927058770002e-10

// arithmetic type of a sweep: the MFMA, its C-register <-> window map, frexp / ldexp.  double:
// v_mfma_f64_16x16x4_f64, register r of lane group lg = window lg + 4 r; float (fp32 mode, round 4):
// v_mfma_f32_16x16x4_f32 at twice the rate, register r = window 4 lg + r.
typedef float float4_lv __attribute__((ext_vector_type(4)));
template <typename T> struct LV;
template <> struct LV<double> {
  typedef double4_t v4;
  static __device__ __forceinline__ v4 mma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lg, int r) { return lg + 4 * r; }
  static __device__ __forceinline__ int fexp(double x) { return __builtin_amdgcn_frexp_exp(x); }
  static __device__ __forceinline__ double ldx(double x, int e) { return ldexp(x, e); }
};
template <> struct LV<float> {
  typedef float4_lv v4;
  static __device__ __forceinline__ v4 mma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lg, int r) { return 4 * lg + r; }
  static __device__ __forceinline__ int fexp(float x) { return __builtin_amdgcn_frexp_expf(x); }
  static __device__ __forceinline__ float ldx(float x, int e) { return ldexpf(x, e); }
};
// Initial message of a window, lane = state (K <= 64): lalpha_0 = mod_init + ll_0 combined in the
// log domain (see k_lin_init), scaled by its own binary exponent H; returns a0[j], sets
// h0 = H - k0 (k0: the emission exponent of the window's first row).
// (COH: l0 was written by another workgroup of the SAME launch with agent-scope stores -- kernels_fused.h)
template <bool COH = false>
__device__ __forceinline__ double lin_init_lane(const double* __restrict__ mod_init,
                                                const double* __restrict__ l0, int jc, bool valid,
                                                double k0, double& h0) {
  const double l0v = COH ? __hip_atomic_load(l0 + jc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : l0[jc];
  const double v = valid ? mod_init[jc] + l0v : -INFINITY;
  const double m = wave_max(v);
  const double H = (m > -1e300 && m < 1e300) ? ceil(m * LOG2E_D) : 0.0;
  h0 = H - k0;
  return valid ? exp(fma(-H, LN2_LO_D, fma(-H, LN2_HI_D, v))) : 0.0;
}

// ------------------------------------------------------------------------------------
//  K2g (round 5), minibatch-sized batches: ONE wavefront per (window, direction) with the whole
//  mat-vec in registers -- no LDS round trip and no workgroup barrier on the 257-step chain.
//  Lane l = 16 r + c holds state l of the entering vector.  It forms the partial sums of the FOUR
//  targets 16 q + c (q = 0..3) over its source block [16 r, 16 r + 16): 64 DPP FMAs whose source
//  operand is a row broadcast of the wave's own registers (row_newbcast:N = lane 16 r + N), four
//  independent chains.  The four rows' partials of a target meet through the two lane-swap
//  instructions gfx950 has (v_permlane32_swap: upper half of one register <-> lower half of another;
//  v_permlane16_swap: odd rows <-> even rows): swap(P0, P2), swap(P1, P3), two adds, swap, one add --
//  and lane 16 r + c holds the finished sum of target 16 r + c, i.e. the next entering vector in
//  place.  A step of k_wave_lin4 was 16 FMAs + an LDS round trip + a barrier between four waves
//  (~740 cycles); this one is ~110 instructions of one wave.  Same inputs / outputs as k_wave_lin4;
//  the 64-term sum is associated as 4 blocks x 16, blocks (r, r + 2) first.
//  fp32 mode (ST = float): the same step on v_fmac_f32_dpp; exponents and the local bound's running
//  product stay double.
// ------------------------------------------------------------------------------------
#ifndef WLR_KO
#define WLR_KO 0      // measurement knock-outs / add-ons of k_wave_linr (tools/probe/wlr_probe.hip); 0 in the product
#endif
typedef unsigned wr_u2 __attribute__((ext_vector_type(2)));
template <int N, bool FIRST = false>
__device__ __forceinline__ void fmac_row_bcast(float& acc, float p, float a) {
  if (FIRST)
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(p), "v"(a), "n"(N));
  else
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(p), "v"(a), "n"(N));
}
// lanes 0..31: x(l) + x(l + 32);  lanes 32..63: y(l - 32) + y(l)
__device__ __forceinline__ double swap_add32(double x, double y) {
  const wr_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const wr_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ float swap_add32(float x, float y) {
  const wr_u2 v = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(v.x) + __uint_as_float(v.y);
}
// even rows: x(l) + x(l + 16);  odd rows: y(l - 16) + y(l)
__device__ __forceinline__ double swap_add16(double x, double y) {
  const wr_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const wr_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ float swap_add16(float x, float y) {
  const wr_u2 v = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(v.x) + __uint_as_float(v.y);
}
// One eighth of a step's mat-vec: the DPP FMAs of sources N0, N0 + 1 into the four target accumulators.
// One asm statement per group, so that the exponent reduction's instructions can be placed between the
// groups by hand (a lone wave issues in order: what matters is the instruction count, and that no
// dependent pair sits back to back).
template <int N0, bool FIRST>
__device__ __forceinline__ void wlr_fma2(double& p0, double& p1, double& p2, double& p3, double pc,
                                         double a00, double a10, double a20, double a30,
                                         double a01, double a11, double a21, double a31) {
#define WLR_F(ACC, CO, NN) "v_fmac_f64_dpp %" #ACC ", %4, %" #CO " row_newbcast:%" #NN " row_mask:0xf bank_mask:0xf\n\t"
  if (FIRST)
    asm volatile("s_nop 4\n\t" WLR_F(0, 5, 13) WLR_F(1, 6, 13) WLR_F(2, 7, 13) WLR_F(3, 8, 13)
                 WLR_F(0, 9, 14) WLR_F(1, 10, 14) WLR_F(2, 11, 14) WLR_F(3, 12, 14)
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                 : "v"(pc), "v"(a00), "v"(a10), "v"(a20), "v"(a30), "v"(a01), "v"(a11), "v"(a21), "v"(a31),
                   "n"(N0), "n"(N0 + 1));
  else
    asm volatile(WLR_F(0, 5, 13) WLR_F(1, 6, 13) WLR_F(2, 7, 13) WLR_F(3, 8, 13)
                 WLR_F(0, 9, 14) WLR_F(1, 10, 14) WLR_F(2, 11, 14) WLR_F(3, 12, 14)
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                 : "v"(pc), "v"(a00), "v"(a10), "v"(a20), "v"(a30), "v"(a01), "v"(a11), "v"(a21), "v"(a31),
                   "n"(N0), "n"(N0 + 1));
#undef WLR_F
}
template <int N0, bool FIRST>
__device__ __forceinline__ void wlr_fma2(float& p0, float& p1, float& p2, float& p3, float pc,
                                         float a00, float a10, float a20, float a30,
                                         float a01, float a11, float a21, float a31) {
#define WLR_F(ACC, CO, NN) "v_fmac_f32_dpp %" #ACC ", %4, %" #CO " row_newbcast:%" #NN " row_mask:0xf bank_mask:0xf\n\t"
  if (FIRST)
    asm volatile("s_nop 4\n\t" WLR_F(0, 5, 13) WLR_F(1, 6, 13) WLR_F(2, 7, 13) WLR_F(3, 8, 13)
                 WLR_F(0, 9, 14) WLR_F(1, 10, 14) WLR_F(2, 11, 14) WLR_F(3, 12, 14)
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                 : "v"(pc), "v"(a00), "v"(a10), "v"(a20), "v"(a30), "v"(a01), "v"(a11), "v"(a21), "v"(a31),
                   "n"(N0), "n"(N0 + 1));
  else
    asm volatile(WLR_F(0, 5, 13) WLR_F(1, 6, 13) WLR_F(2, 7, 13) WLR_F(3, 8, 13)
                 WLR_F(0, 9, 14) WLR_F(1, 10, 14) WLR_F(2, 11, 14) WLR_F(3, 12, 14)
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                 : "v"(pc), "v"(a00), "v"(a10), "v"(a20), "v"(a30), "v"(a01), "v"(a11), "v"(a21), "v"(a31),
                   "n"(N0), "n"(N0 + 1));
#undef WLR_F
}
// The normalising exponent: a uniform integer e with  sum_j p[j] 2^-e  in [1/2, 64) -- the largest
// biased exponent field of the vector's entries (zeros and denormals count as the smallest), as frexp's
// exponent.  32-bit DPP max: four steps inside the rows of 16, row_bcast:15 / :31 across them, one
// v_readlane of lane 63: eight instructions against ~26 for the fp64 wave sum whose exponent the other
// sweep kernels use (any integer keeps the books exact; the vector only has to stay in range).  The
// steps are separate asm statements: the caller places them between the groups of the mat-vec.
// (asm volatile: the field is consumed by the DPP instructions of wlr_emax, which the compiler's hazard recogniser does
//  not see inside inline asm -- a VALU write needs two wait states before a DPP read of the same register.  As plain
//  C++ the extraction was free to sink right in front of the first v_max_i32_dpp (it did once the call became
//  conditional: round 6, RN) and the reduction then read the register's previous contents.  As a volatile statement
//  it stays where it is written: in front of the step's first group of eight FMAs.)
__device__ __forceinline__ int wlr_expfield(double v) {
  int e;
  asm volatile("v_bfe_u32 %0, %1, 20, 11" : "=v"(e) : "v"(__double2hiint(v)));
  return e;
}
__device__ __forceinline__ int wlr_expfield(float v) {
  int e;
  asm volatile("v_bfe_u32 %0, %1, 23, 8" : "=v"(e) : "v"(__float_as_uint(v)));
  return e;
}
template <int STEP>
__device__ __forceinline__ void wlr_emax(int& e) {
  if (STEP == 0) asm volatile("v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(e));
  if (STEP == 1) asm volatile("v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(e));
  if (STEP == 2) asm volatile("v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(e));
  if (STEP == 3) asm volatile("v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(e));
  if (STEP == 4) asm volatile("v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(e));
  if (STEP == 5) asm volatile("v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(e));
}
template <typename CT> struct WlrBias;
template <> struct WlrBias<double> { static constexpr int v = 1022; };
template <> struct WlrBias<float> { static constexpr int v = 126; };
template <typename CT> struct WlrRing { CT v[64][65]; };      // [step & 63][state]: a lane later sums ITS step's row
// Progress of a sweep, published for the statistics workgroups of the fused kernel (kernels_fused.h, round 6): when
// the wave has stored the row of sweep step thr[i] it adds one to counter i (agent-scope release in front: the rows
// are visible to whoever sees the count).  Both directions use the same step thresholds.
#define WLR_MAX_BANDS 12
#ifndef PIPE_PD
#define PIPE_PD 12
#endif
struct WlrPub {
  unsigned* cnt;                // counter of band i at cnt + 16 i (64 bytes apart)
  int nb;                       // bands
  int thr[WLR_MAX_BANDS];       // ascending sweep-step thresholds
  unsigned long long* dbgw;     // measurement only: per sweep workgroup 32 stamps (wave 0), nullptr in normal runs
  // the other direction (EMW variant of the body): the emission rows are produced INSIDE the launch, by the statistics
  // workgroups before their first band opens, in rounds of outside-in priority min(t, Lm - 1 - t); round i is complete
  // when counter em_cnt + 16 i has reached em_tgt[i], and then every row of priority <= em_thr[i] is there
  const unsigned* em_cnt;
  int em_n;                     // rounds (0: the rows were written by an earlier launch)
  int em_thr[WLR_MAX_BANDS];    // ascending; the last round's is INT_MAX
  unsigned em_tgt[WLR_MAX_BANDS];
};
// RN (round 6): the vector is re-normalised every RN-th step only (RN divides 12).  The normalising exponent -- six DPP
// max steps, a v_readlane, two v_ldexp and the books -- is ~18 of a step's ~115 instructions, and fp64 has the range to
// go without it for a few steps when no transition expectation lies below SVIHMM_LTRAN_F32_MIN = -60 nats (the host's
// f32_ok): a step shrinks the vector's largest entry by at most 2^-88 (the transition factor and the emission row's
// maximum >= 1/2), the normalisation lags one step, so with RN = 4 a stored vector's largest entry stays above 2^-440
// and the consumers' product ah bh above 2^-880 -- RN = 6 would not leave that product inside fp64.  Scaling by powers of
// two is exact: the stored mantissas and every statistic are those of RN = 1 bit for bit; the local bound (a sum of
// logarithms of the differently scaled sums + the exponent books) agrees to rounding.
// SE: storage type of the emission rows Eh where it differs from the messages' (fp32 mode inside the fused kernel: float Eh
// from the bf16 emission kernel, fp64 messages).
template <bool FWD, bool FULLK, typename ST, typename CT, bool PUB = false, bool EMW = false, int RN = 1, typename SE = ST>
__device__ __forceinline__ void wave_linr_body(
    const SE* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Am, const double* __restrict__ mod_init,
    const double* __restrict__ ll0, size_t l0stride, int Lm, int K, ST* __restrict__ out,
    double* __restrict__ xout, double* __restrict__ local_lb, double* __restrict__ logz,
    double2* __restrict__ zfac, WlrRing<CT>& ring, const int b, const int j,
    const WlrPub* __restrict__ pub = nullptr) {
  // (CT = arithmetic type of the mat-vec; ST = storage type of Eh and of the messages; b = window, j = lane)
  int pub_i = 0, pub_next = 0x7fffffff;
  auto publish = [&](int s) {           // every band whose threshold the sweep has reached (uniform)
    if constexpr (PUB) {
      if (__builtin_amdgcn_readfirstlane(s) >= __builtin_amdgcn_readfirstlane(pub_next)) {
        // (no fence, no wait: the caller only names steps whose rows are KNOWN to be complete -- see the loop below)
        while (s >= pub_next) {
          if (j == 0) __hip_atomic_fetch_add(pub->cnt + 16 * pub_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ++pub_i;
          pub_next = __builtin_amdgcn_readfirstlane(pub_i < pub->nb ? pub->thr[pub_i] : 0x7fffffff);
        }
      }
    }
  };
  if constexpr (PUB) pub_next = pub->nb > 0 ? pub->thr[0] : 0x7fffffff;
  // EMW: which emission rows are there.  em_lim = every row of priority <= em_lim is complete (rounds [0, em_r) seen).
  // The sweep requests Eh rows 2 PD steps ahead of their use; the round counter is read asynchronously -- requested at
  // the head of a block of PD steps, looked at behind it -- and only a sweep that has caught up with the emission
  // rounds waits (bounded like every gate of the loop: a round that never completes must not hang the queue).
  int em_r = 0, em_lim = EMW ? -1 : 0x7fffffff;
  unsigned em_seen_v = 0;
  auto em_wait = [&](int need) {
    if constexpr (EMW) {
      while (em_lim < need && em_r < pub->em_n) {
        const unsigned tgt = pub->em_tgt[em_r];
        const unsigned long long t0 = wall_clock64();
        unsigned n = 0;
        while ((int)(__builtin_amdgcn_readfirstlane(__hip_atomic_load(pub->em_cnt + 16 * em_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - tgt) < 0) {
          __builtin_amdgcn_s_sleep(8);
          if ((++n & 1023u) == 0u && wall_clock64() - t0 > SVI_SYNC_TICKS) break;
        }
        em_lim = __builtin_amdgcn_readfirstlane(pub->em_thr[em_r]);
        ++em_r;
      }
    }
  };
  const int em_half = (Lm - 1) >> 1;              // the largest priority there is
  auto em_need = [&](int srow) { return srow < em_half ? srow : em_half; };
  auto eload = [&](const SE* p) -> CT {
    if constexpr (EMW) return (CT)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return (CT)*p;
  };
  auto kload = [&](const double* p) -> double {
    if constexpr (EMW) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
  };
  const int r = j >> 4, c = j & 15;
  const bool valid = FULLK || j < K;
  const int jc = valid ? j : 0;
  CT a[4][16];                                  // a[q][N] = A[16 r + N][16 q + c] (fwd) / A[16 q + c][16 r + N] (bwd)
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int src = 16 * r + i, tgt = 16 * q + c;
      a[q][i] = (FULLK || (tgt < K && src < K)) ? (CT)Am[(size_t)src * K + tgt] : (CT)0;
    }
  const size_t wrow = (size_t)b * Lm;
  // (FULLK: the row stride is the constant 64 -- the unrolled block's stores and row requests become immediate offsets
  //  from one pointer per block instead of a 64-bit vector add per step)
  const ptrdiff_t KS = FULLK ? (ptrdiff_t)64 : (ptrdiff_t)K;
  const ptrdiff_t dstep = FWD ? KS : -KS;
  const size_t row0 = FWD ? 0 : (size_t)(Lm - 1);
  // uniform row pointers (scalar registers) + the lane's column: loads and stores take base + offset
  const SE* __restrict__ ep = Eh + (wrow + row0) * KS;      // Eh row of sweep step 0
  ST* __restrict__ op = out + (wrow + row0) * KS;
  double* __restrict__ xb = xout + wrow;
  auto rowof = [&](int s) { return FWD ? s : Lm - 1 - s; };
  // exponent books: h = the current vector's binary exponent (an exact integer: |h| <= ~1100 Lm),
  // hsum = sum of the h of all vectors so far, lbacc = this lane's share of sum_t log(sum_j ah_t[j])
  int h = 0;
  long long hsum = 0;
  double lbacc = 0.0;
  CT pcur;
  constexpr int PD = PUB ? PIPE_PD : 12;
  em_wait(em_need(PD));                         // row 0 (ll0, kexp, Eh of the backward sweep) and the first PD rows
  {
    CT o;
    if (FWD) {
      double h0;
      o = (CT)lin_init_lane<EMW>(mod_init, ll0 + (size_t)b * l0stride, jc, valid, kload(kexp + wrow), h0);
      h = __builtin_amdgcn_readfirstlane((int)h0);
      pcur = o;
      if constexpr (!PUB) ring.v[0][j] = o;
    } else {
      const CT e0 = eload(ep + jc);
      o = valid ? (CT)1 : (CT)0;
      pcur = valid ? e0 : (CT)0;
    }
#ifdef PIPE_PLAIN_STORES
    if (valid) op[jc] = o;
#else
    if (valid) { if constexpr (PUB) __hip_atomic_store(&op[jc], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else op[jc] = o; }
#endif
  }
  int hkeep = h;                                // lane (s & 63) keeps the exponent of sweep step s
  // the local bound's terms of the vectors in ring rows [0, n): lane t sums row t
  auto ring_flush = [&](int n) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
      t0 += (double)ring.v[j][i]; t1 += (double)ring.v[j][i + 1];
      t2 += (double)ring.v[j][i + 2]; t3 += (double)ring.v[j][i + 3];
    }
    const double tot = (t0 + t1) + (t2 + t3);
    if (j < n) lbacc += fast_log(tot);
    __builtin_amdgcn_wave_barrier();
  };
  // (rows of Eh requested ahead of their step: 12 where the wave has the SIMD's whole register file -- the stand-alone
  //  kernel --, 6 inside the fused kernel's 256-register budget, where 12 spilled into the step loop)
  auto eclamped = [&](int s) { return eload(ep + (ptrdiff_t)(s < Lm ? s : Lm - 1) * dstep + jc); };
  CT eq[PD];
#pragma unroll
  for (int u = 0; u < PD; ++u) eq[u] = eclamped(1 + u);
  static_assert(RN >= 1 && PD % RN == 0, "RN divides the unroll depth");
  constexpr bool LAZY = RN > 1 && 64 % RN == 0;
  ST* opl = op + jc;                            // the lane's own output pointer: one 64-bit add per step
  // (tried: two accumulator sets, the next step's zeroed between this step's last FMA group and the lane swaps, where the
  //  compiler pads four s_nop after the asm it cannot look into -- it moved the zeroing back in front of the next FMAs
  //  and, with a scheduling barrier to pin it, padded eight)
  auto step = [&](int s, CT et, const bool rn) {      // rn: a compile-time constant after unrolling (see the loops)
    CT p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    int ef = rn ? wlr_expfield(pcur) : 0;       // (the exponent comes from the ENTERING vector: off the chain)
#define WLR_G(N0, FI) wlr_fma2<N0, FI>(p0, p1, p2, p3, pcur, a[0][N0], a[1][N0], a[2][N0], a[3][N0], \
                                       a[0][N0 + 1], a[1][N0 + 1], a[2][N0 + 1], a[3][N0 + 1]);
    WLR_G(0, true)
#if !(WLR_KO & 1)
    if (rn) wlr_emax<0>(ef);
    WLR_G(2, false)
    if (rn) wlr_emax<1>(ef);
    WLR_G(4, false)
    if (rn) wlr_emax<2>(ef);
    WLR_G(6, false)
    if (rn) wlr_emax<3>(ef);
    WLR_G(8, false)
    if (rn) wlr_emax<4>(ef);
    WLR_G(10, false)
    if (rn) wlr_emax<5>(ef);
    WLR_G(12, false)
    WLR_G(14, false)
#else
    wlr_emax<0>(ef); wlr_emax<1>(ef); wlr_emax<2>(ef); wlr_emax<3>(ef); wlr_emax<4>(ef); wlr_emax<5>(ef);
#endif
#undef WLR_G
#if WLR_KO & 2
    const int e2 = 0;
#else
    int e2 = 0;
    if (rn) {
      const int em = __builtin_amdgcn_readlane(ef, 63);
      e2 = em ? em - WlrBias<CT>::v : 0;
    }
#endif
    const CT q0 = swap_add32(p0, p2), q1 = swap_add32(p1, p3);
    const CT acc = swap_add16(q0, q1);
    // (LAZY: h only changes at re-normalising steps, so the books of the RN - 1 steps before one are written there)
    if constexpr (!LAZY) { if (FWD) hsum += h; }
    else { if (FWD && rn) hsum += (long long)RN * h; }
    const int h_old = h;
    CT o;
    if (FWD) { o = rn ? LV<CT>::ldx(acc * et, -e2) : acc * et; if (!FULLK) o = valid ? o : (CT)0; pcur = o; }
    else { o = rn ? LV<CT>::ldx(acc, -e2) : acc; if (!FULLK) o = valid ? o : (CT)0; pcur = et * o; }
    if (rn) h += e2;
    opl += dstep;
#if !(WLR_KO & 4)
    // (PUB: agent-scope atomic store = written through to where every CU of the device reads it coherently; "complete"
    //  then means "visible", and the statistics workgroups read the row with agent-scope atomic loads -- no cache
    //  maintenance on either side.  Tried and measured (tools/r6_fused_trace.py, tools/probe/wlr_pack_probe.hip): an
    //  acquire fence in every statistics wave halved the sweeps' speed (208 workgroups x 8 waves invalidating the L2s
    //  at every band); a release fence per band in the sweep wave drains its prefetch queue: 2 us each)
#ifdef PIPE_PLAIN_STORES
    if (FULLK || valid) *opl = o;
#else
    if (FULLK || valid) { if constexpr (PUB) __hip_atomic_store(opl, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *opl = o; }
#endif
#endif
    if constexpr (!LAZY) {
      hkeep = (j == (s & 63)) ? h : hkeep;
      if (FWD && !PUB) ring.v[s & 63][j] = o;
      if ((s & 63) == 63) {                     // uniform: the books (and the ring's rows) of steps s - 63 .. s
        if constexpr (PUB) { if (pub->dbgw && threadIdx.x == 0) pub->dbgw[(size_t)blockIdx.x * 32 + 2 + 2 * (s >> 6)] = wall_clock64(); }
        xb[rowof(s - 63 + j)] = (double)hkeep;
        if (FWD && !PUB) ring_flush(64);
        if constexpr (PUB) { if (pub->dbgw && threadIdx.x == 0) pub->dbgw[(size_t)blockIdx.x * 32 + 3 + 2 * (s >> 6)] = wall_clock64(); }
      }
    } else {
      // LAZY: the books only change at re-normalising steps, so the lanes of steps s - RN + 1 .. s - 1 get the old book
      // HERE, the 64-step flush rides in the re-normalising step that opens the next group of 64 (one test per RN
      // steps instead of four scalar instructions in every step), and then the lane of step s gets the new book
      if (rn) {
        hkeep = ((unsigned)(j - ((s - RN + 1) & 63)) < (unsigned)(RN - 1)) ? h_old : hkeep;
        if ((s & 63) == 0) {                    // uniform: steps s - 64 .. s - 1 are complete
          if constexpr (PUB) { if (pub->dbgw && threadIdx.x == 0) pub->dbgw[(size_t)blockIdx.x * 32 + 2 * (s >> 6)] = wall_clock64(); }
          xb[rowof(s - 64 + j)] = (double)hkeep;
          if (FWD && !PUB) ring_flush(64);      // (before this step's vector takes ring row 0)
        }
        hkeep = (j == (s & 63)) ? h : hkeep;
      }
      if (FWD && !PUB) ring.v[s & 63][j] = o;
    }
#if WLR_KO & 48
    // measurement only (tools/probe/wlr_probe.hip): what publishing the sweep's progress would cost -- an
    // agent-scope release every 32 (bit 16) or 16 (bit 32) steps + one relaxed store of the step
    if ((s & ((WLR_KO & 32) ? 15 : 31)) == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (j == 0) __hip_atomic_store(reinterpret_cast<int*>(zfac + b) + (FWD ? 0 : 1), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
  };
  int s = 1;
  em_wait(em_need(2 * PD));                      // the first block requests rows PD + 1 .. 2 PD
  // Blocks whose requests all lie inside the window (INNER) take the row pointer as a running scalar; only the last one
  // or two clamp the row index (the clamp's 64-bit scalar multiply cost every step nine scalar instructions).
  const SE* enext = ep + (ptrdiff_t)(1 + PD) * dstep + jc;      // the lane's entry of the first request of block s = 1
  auto run_block = [&](auto inner_c) {
    constexpr bool INNER = decltype(inner_c)::value;
    if constexpr (EMW) {
      if (em_r < pub->em_n) em_seen_v = __hip_atomic_load(pub->em_cnt + 16 * em_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const CT et = eq[u];
#if !(WLR_KO & 8)
      if constexpr (INNER) { eq[u] = eload(enext); enext += dstep; }
      else eq[u] = eclamped(s + u + PD);
#endif
      step(s + u, et, RN == 1 || (1 + u) % RN == 0);      // (s = 1 mod PD: step s + u re-normalises iff RN | s + u)
    }
    // Progress, once per block of PD steps.  Which rows are complete?  Loads and stores of a wave retire through ONE
    // in-order counter: the last step of this block waited for the Eh row requested PD steps earlier, and everything
    // issued before that request -- the stores of the steps up to s - 2 -- has retired with it.  So the sweep publishes
    // step s - 2 here, ~PD steps (2 us) behind its real position, and never waits for its own stores: the bands open a
    // little later, the 257-step chain does not stall.  (The last rows are published after the drain below.)
    publish(s - 2);
    if constexpr (EMW) {
      if (em_r < pub->em_n) {
        if ((int)(__builtin_amdgcn_readfirstlane(em_seen_v) - pub->em_tgt[em_r]) >= 0) { em_lim = __builtin_amdgcn_readfirstlane(pub->em_thr[em_r]); ++em_r; }
        // (tried: the same test in front of every step's request instead of once per block -- 23 steps less lookahead; the
        //  branch in the unrolled step loop cost the chain 0.53 us a step instead of 0.31)
        em_wait(em_need(s + 3 * PD - 1));       // the next block requests rows up to (s + PD) + 2 PD - 1
      }
    }
  };
  // (only where a workgroup runs ONE direction -- the fused kernel: with both directions in one kernel, as in k_wave_linr,
  //  the second copy of the unrolled block pushes the step loops out of the instruction cache: fp32-mode iteration 0.168 ->
  //  0.177 ms, measured)
  if constexpr (PUB) { for (; s + 2 * PD <= Lm; s += PD) run_block(std::true_type{}); }
  for (; s + PD <= Lm; s += PD) run_block(std::false_type{});
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (s + u < Lm) step(s + u, eq[u], RN == 1 || (1 + u) % RN == 0);
  em_wait(em_half);                             // (short windows: the epilogue reads kexp of every row)
  if constexpr (PUB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  publish(0x7ffffffe);                          // (whatever is left: nobody may wait for a band the sweep never names)
  if constexpr (LAZY) { if (FWD) hsum += (long long)((Lm - 1) % RN) * h; }     // the steps behind the last re-normalisation
  {
    const int sl = Lm - 1, s0 = sl & ~63;
    if constexpr (LAZY) hkeep = (s0 + j > sl - sl % RN && s0 + j <= sl) ? h : hkeep;
    // (LAZY: a full last group is flushed here as well -- no re-normalising step follows it)
    if ((LAZY || (sl & 63) != 63) && s0 + j <= sl) xb[rowof(s0 + j)] = (double)hkeep;
    if (FWD && !PUB && (LAZY || (sl & 63) != 63)) ring_flush((sl & 63) + 1);
  }
  // (PUB variant: no row-sum ring, no flushes inside the chain -- the terms sum_t log(sum_j ah_t[j]) of the local bound are
  //  formed by the statistics workgroups of the fused kernel, which read every stored vector anyway (kernels_fused.h);
  //  local_lb[b] below then carries the exponent books only)
  if constexpr (PUB) { if (pub->dbgw && threadIdx.x == 0) pub->dbgw[(size_t)blockIdx.x * 32 + 12] = wall_clock64(); }
  if (!FWD) return;
  double ks = 0.0, kk = 0.0;
  for (int t = j; t < Lm; t += 64) {
    const double kv = kload(kexp + wrow + t);
    ks += kv;
    kk += kv * (double)(Lm - t);
  }
  ks = wave_sum_dpp(ks);
  kk = wave_sum_dpp(kk);
  lbacc = wave_sum_dpp(lbacc);
  const double tot = wave_sum_dpp((double)pcur);
  if (j == 0) {
    const double zm = __builtin_amdgcn_frexp_mant(tot);
    const double zexp = (double)__builtin_amdgcn_frexp_exp(tot);
    const double hd = (double)h;
    local_lb[b] = lbacc + ((double)hsum + hd + kk) * LN2_D;
    logz[b] = log(zm) + (hd + ks + zexp) * LN2_D;
    zfac[b] = make_double2(1.0 / zm, hd + zexp);
  }
}
template <typename ST = double, typename CT = ST, int RN = 1>
__global__ __launch_bounds__(64) void k_wave_linr(
    const ST* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Aexp, const double* __restrict__ AexpT,
    const double* __restrict__ mod_init, const double* __restrict__ ll0, size_t l0stride, int Lm,
    int K, ST* __restrict__ ah,
    ST* __restrict__ bh, double* __restrict__ hx, double* __restrict__ gx,
    double* __restrict__ local_lb, double* __restrict__ logz, double2* __restrict__ zfac,
    SviSync sy = SviSync{nullptr, 0u, nullptr, nullptr, nullptr}) {
  __shared__ WlrRing<CT> ring;
  if (!svi_gate(sy)) { svi_poison(sy); return; }   // (SVI loop: the globals kernel of the side stream has arrived)
  if (blockIdx.y == 0) {
    if (K == 64) wave_linr_body<true, true, ST, CT, false, false, RN>(Eh, kexp, Aexp, mod_init, ll0, l0stride, Lm, K, ah, hx, local_lb, logz, zfac, ring, blockIdx.x, threadIdx.x);
    else wave_linr_body<true, false, ST, CT, false, false, RN>(Eh, kexp, Aexp, mod_init, ll0, l0stride, Lm, K, ah, hx, local_lb, logz, zfac, ring, blockIdx.x, threadIdx.x);
  } else {
    if (K == 64) wave_linr_body<false, true, ST, CT, false, false, RN>(Eh, kexp, AexpT, mod_init, ll0, l0stride, Lm, K, bh, gx, local_lb, logz, zfac, ring, blockIdx.x, threadIdx.x);
    else wave_linr_body<false, false, ST, CT, false, false, RN>(Eh, kexp, AexpT, mod_init, ll0, l0stride, Lm, K, bh, gx, local_lb, logz, zfac, ring, blockIdx.x, threadIdx.x);
  }
}

