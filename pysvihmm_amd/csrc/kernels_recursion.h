// kernels_recursion.h -- K2 forward/backward sweeps (wave-per-window, generic, batched fp64 MFMA), K3 posterior, K6 FFBS sampling.
// Part of libsvihmm_hip.so; compiled in tu_recursion.hip.
#pragma once

// ------------------------------------------------------------------------------------
//  K2a: forward / backward messages, one wavefront per (window, direction), K <= KMAX<=64.
//       The transition column (fwd) / row (bwd) of exp(ltran) lives in registers,
//       the shifted probabilities p_i are exchanged through LDS.
//       grid (B, ndir), block 64.
// ------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(64) void k_fb_wave(
    const double* __restrict__ ll, const double* __restrict__ Aexp,
    const double* __restrict__ mod_init, int Lm, int K, int dir0,
    double* __restrict__ la_out, double* __restrict__ lb_out) {
  __shared__ double p_s[2][KMAX];
  const int b = blockIdx.x, dir = dir0 + blockIdx.y, j = threadIdx.x;
  const bool valid = j < K;
  double a[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) {
    double v = 0.0;
    if (valid && i < K) v = (dir == 0) ? Aexp[i * K + j] : Aexp[j * K + i];
    a[i] = v;
  }
  const double* llb = ll + (size_t)b * Lm * K;
  const double NEG_INF = -INFINITY;
  int cur = 0;
  if (dir == 0) {
    double* out = la_out + (size_t)b * Lm * K;
    double la = valid ? mod_init[j] + llb[j] : NEG_INF;
    if (valid) out[j] = la;
    double llnext = (valid && Lm > 1) ? llb[K + j] : 0.0;
    for (int t = 1; t < Lm; ++t) {
      const double llt = llnext;
      if (valid && t + 1 < Lm) llnext = llb[(size_t)(t + 1) * K + j];
      const double m = wave64_max_fast(la);
      const double p = valid ? fast_exp(la - m) : 0.0;
      if (j < KMAX) p_s[cur][j] = p;
      __syncthreads();
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < KMAX; i += 2) {
        s0 = fma(p_s[cur][i], a[i], s0);
        s1 = fma(p_s[cur][i + 1], a[i + 1], s1);
      }
      cur ^= 1;
      la = valid ? fast_log(s0 + s1) + m + llt : NEG_INF;
      if (valid) out[(size_t)t * K + j] = la;
    }
  } else {
    double* out = lb_out + (size_t)b * Lm * K;
    double lb = 0.0;
    if (valid) out[(size_t)(Lm - 1) * K + j] = 0.0;
    double llnext = valid ? llb[(size_t)(Lm - 1) * K + j] : 0.0;
    for (int t = Lm - 2; t >= 0; --t) {
      const double u = valid ? lb + llnext : NEG_INF;
      if (valid && t >= 1) llnext = llb[(size_t)t * K + j];
      const double m = wave64_max_fast(u);
      const double p = valid ? fast_exp(u - m) : 0.0;
      if (j < KMAX) p_s[cur][j] = p;
      __syncthreads();
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < KMAX; i += 2) {
        s0 = fma(p_s[cur][i], a[i], s0);
        s1 = fma(p_s[cur][i + 1], a[i + 1], s1);
      }
      cur ^= 1;
      lb = fast_log(s0 + s1) + m;
      if (valid) out[(size_t)t * K + j] = lb;
    }
  }
}

// ------------------------------------------------------------------------------------
//  K2b: forward / backward, generic K (block = roundup(K,64) threads, thread = state).
//       Transition matrix (fwd: A, bwd: A^T) is read from LDS when it fits, else HBM/L2.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double block_max(double v, double* red, int nw) {
  v = wave_max(v);
  if (nw == 1) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double m = red[0];
  for (int i = 1; i < nw; ++i) m = fmax(m, red[i]);
  return m;
}

__global__ void k_fb_generic(const double* __restrict__ ll, const double* __restrict__ Aexp,
                             const double* __restrict__ AexpT,
                             const double* __restrict__ mod_init, int Lm, int K, int dir0,
                             int m_in_lds, double* __restrict__ la_out,
                             double* __restrict__ lb_out) {
  extern __shared__ double sm[];
  double* p_s = sm;            // [2][K]
  double* red = sm + 2 * K;    // [16]
  double* M_s = red + 16;      // [K][K] if m_in_lds
  const int b = blockIdx.x, dir = dir0 + blockIdx.y, j = threadIdx.x;
  const int nw = (blockDim.x + 63) >> 6;
  const bool valid = j < K;
  // M[i][j] such that out_j = sum_i p_i M[i][j]:  fwd M = A ; bwd M[jj][i] = A[i][jj] = A^T
  const double* Mg = (dir == 0) ? Aexp : AexpT;
  const double* M = Mg;
  if (m_in_lds) {
    for (int e = threadIdx.x; e < K * K; e += blockDim.x) M_s[e] = Mg[e];
    M = M_s;
  }
  __syncthreads();
  const double* llb = ll + (size_t)b * Lm * K;
  const double NEG_INF = -INFINITY;
  int cur = 0;
  if (dir == 0) {
    double* out = la_out + (size_t)b * Lm * K;
    double la = valid ? mod_init[j] + llb[j] : NEG_INF;
    if (valid) out[j] = la;
    for (int t = 1; t < Lm; ++t) {
      const double llt = valid ? llb[(size_t)t * K + j] : 0.0;
      const double m = block_max(la, red, nw);
      if (valid) p_s[cur * K + j] = exp(la - m);
      __syncthreads();
      double s = 0.0;
      if (valid)
        for (int i = 0; i < K; ++i) s = fma(p_s[cur * K + i], M[(size_t)i * K + j], s);
      cur ^= 1;
      la = valid ? log(s) + m + llt : NEG_INF;
      if (valid) out[(size_t)t * K + j] = la;
    }
  } else {
    double* out = lb_out + (size_t)b * Lm * K;
    double lb = 0.0;
    if (valid) out[(size_t)(Lm - 1) * K + j] = 0.0;
    for (int t = Lm - 2; t >= 0; --t) {
      const double u = valid ? lb + llb[(size_t)(t + 1) * K + j] : NEG_INF;
      const double m = block_max(u, red, nw);
      if (valid) p_s[cur * K + j] = exp(u - m);
      __syncthreads();
      double s = 0.0;
      if (valid)
        for (int i = 0; i < K; ++i) s = fma(p_s[cur * K + i], M[(size_t)i * K + j], s);
      cur ^= 1;
      lb = log(s) + m;
      if (valid) out[(size_t)t * K + j] = lb;
    }
  }
}

// ------------------------------------------------------------------------------------
//  K2x: the reference's recursion literally -- np.logaddexp.reduce over (message + ltran)
//       (hmmbase.py:295, 319): one exp per (source, target) pair and step.  Every other
//       sweep kernel works with exp(ltran), which cannot represent transition expectations
//       below the range of exp() (Dirichlet pseudo-counts of ~1e-3: psi(1e-3) = -1000); models
//       like that are routed here (svihmm_set_globals decides).  Thread = state, block =
//       roundup(K, 64); ltran in LDS (row stride K + 1: both directions conflict-free) when it
//       fits.  grid (B, ndir).
// ------------------------------------------------------------------------------------
__global__ void k_fb_exact(const double* __restrict__ ll, const double* __restrict__ ltran,
                           const double* __restrict__ mod_init, int Lm, int K, int dir0,
                           int lt_in_lds, double* __restrict__ la_out, double* __restrict__ lb_out) {
  extern __shared__ double sm[];
  double* v_s = sm;              // [2][K]
  double* lt_s = sm + 2 * K;     // [K][K + 1] if lt_in_lds
  const int b = blockIdx.x, dir = dir0 + blockIdx.y, j = threadIdx.x;
  const bool valid = j < K;
  const int ls = lt_in_lds ? K + 1 : K;
  const double* lt = ltran;
  if (lt_in_lds) {
    for (int e = threadIdx.x; e < K * K; e += blockDim.x) lt_s[(e / K) * (K + 1) + e % K] = ltran[e];
    lt = lt_s;
  }
  const double* llb = ll + (size_t)b * Lm * K;
  // element (source i, target j) of the transition expectations as seen from thread j:
  // forward lt[i][j], backward (thread = source state) lt[j][i]
  const size_t si = dir == 0 ? (size_t)ls : 1, sj = dir == 0 ? 1 : (size_t)ls;
  const double* mycol = lt + (valid ? j : 0) * sj;
  int cur = 0;
  auto lse = [&](const double* v) {
    double m = -INFINITY;
    for (int i = 0; i < K; ++i) m = fmax(m, v[i] + mycol[i * si]);
    if (!(m > -INFINITY)) return m;            // all terms -inf (or NaN): as np.logaddexp.reduce
    if (!(m < INFINITY)) return m;
    // terms more than 50 nats below the maximum cannot change the sum (K e^-50 < 2^-53): in the
    // sparse models this kernel exists for that is most of them, and exp is the step's cost
    double s = 0.0;
    for (int i = 0; i < K; ++i) {
      const double d = v[i] + mycol[i * si] - m;
      if (d > -50.0) s += exp(d);
    }
    return m + log(s);
  };
  if (dir == 0) {
    double* out = la_out + (size_t)b * Lm * K;
    double la = valid ? mod_init[j] + llb[j] : -INFINITY;
    if (valid) out[j] = la;
    for (int t = 1; t < Lm; ++t) {
      if (valid) v_s[cur * K + j] = la;
      __syncthreads();
      if (valid) {
        la = lse(v_s + cur * K) + llb[(size_t)t * K + j];
        out[(size_t)t * K + j] = la;
      }
      cur ^= 1;
    }
  } else {
    double* out = lb_out + (size_t)b * Lm * K;
    double lb = 0.0;
    if (valid) out[(size_t)(Lm - 1) * K + j] = 0.0;
    for (int t = Lm - 2; t >= 0; --t) {
      if (valid) v_s[cur * K + j] = lb + llb[(size_t)(t + 1) * K + j];
      __syncthreads();
      if (valid) {
        lb = lse(v_s + cur * K);
        out[(size_t)t * K + j] = lb;
      }
      cur ^= 1;
    }
  }
}

// ------------------------------------------------------------------------------------
//  K2c/K2d: forward and backward(+posterior) sweeps as batched fp64 MFMA mat-mats.
//  A workgroup owns 16 windows (the M dimension of v_mfma_f64_16x16x4_f64); wave s owns
//  the 16-state tile n0=16*s (K <= 64 -> NW = Kp/16 waves).  Per time step
//      out[w][j] = sum_i P[w][i] * M[i][j],   P = exp(prev message - shift[w]) via LDS,
//  M = exp(ltran) (forward) / its transpose (backward) held in registers as the B operand.
//  The per-window shift is c_t = c_{t-1} + ln2*frexp_exp(sum_i P_{t-1}[i]) + max_j ll_t[j]:
//  an upper bound of max_j message_t[j] that is at most ~|min ltran| above it, built only
//  from tile reductions of the PREVIOUS step, so there is one barrier per step and no
//  reduction on the critical path.  sum_t LSE_j lalpha (quirk Q4) is accumulated as a
//  running (mantissa, exponent) product of the per-step sums.
// ------------------------------------------------------------------------------------
#ifndef LN2_D
#define LN2_D 0.69314718055994530942
#endif

template <int NW>
struct FbShared {
  static constexpr int Kp = 16 * NW;
  static constexpr int PS = Kp + 2;
  double __attribute__((aligned(16))) P[2][16][PS];
  // tile reductions: every lane of a 16-lane row holds the same value after the DPP
  // reduction and writes its own slot (branch-free, conflict-free); readers use slot 0
  double tsum[2][16][NW][16];
  double tmll[2][16][NW][16];
  double tq[2][16][NW][16];
};

template <int NW>
__device__ __forceinline__ double4_t fb_matmul(const FbShared<NW>& sh, int cur, int li, int lg,
                                               const double (&Bv)[4 * NW]) {
  constexpr int KS = 4 * NW;
  double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double* prow = &sh.P[cur][li][2 * lg];
#pragma unroll
  for (int c = 0; c < KS / 2; c += 2) {
    const double2 x = *reinterpret_cast<const double2*>(prow + 8 * c);
    const double2 y = *reinterpret_cast<const double2*>(prow + 8 * (c + 1));
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, Bv[2 * c], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, Bv[2 * c + 1], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.x, Bv[2 * c + 2], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.y, Bv[2 * c + 3], a3, 0, 0, 0);
  }
  return (a0 + a1) + (a2 + a3);
}

// FULL: K == 16*NW (no padded states).  Windows beyond B are clamped to B-1 (they redo the
// last window and rewrite identical values), so the time loop has no per-lane predicate
// and compiles to a single basic block: loads issued two steps ahead are waited for with a
// counted vmcnt instead of a full drain.
template <int NW, bool FULL>
__global__ __launch_bounds__(64 * NW) void k_fwd_mfma(
    const double* __restrict__ ll, const double* __restrict__ Aexp,
    const double* __restrict__ mod_init, int B, int Lm, int K, double* __restrict__ la_out,
    double* __restrict__ local_lb, double* __restrict__ logz) {
  // Critical path per step: LDS read -> MFMA -> p = acc * w -> LDS write -> row sum -> barrier.
  // w = exp(ll_t - d) (d = shift increment) does not depend on the MFMA result and the
  // lalpha store (log) of step t is issued during step t+1, so every transcendental runs
  // in the shadow of the matrix pipe.
  constexpr int KS = 4 * NW;
  __shared__ FbShared<NW> sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int j = wave * 16 + li;
  const bool vj = FULL || (j < K);
  const int jc = vj ? j : 0;
  const int b0 = blockIdx.x * 16;
  double Bv[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = 8 * (kk >> 1) + 2 * lg + (kk & 1);
    Bv[kk] = (k < K && vj) ? Aexp[(size_t)k * K + jc] : 0.0;
  }
  size_t base[4];
  int gwc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gw = b0 + lg + 4 * r;
    gwc[r] = gw < B ? gw : B - 1;
    base[r] = (size_t)gwc[r] * Lm * K + jc;
  }
  const double NEG_INF = -INFINITY;
  double c[4], csum[4], mant[4], lln[4], ll2[4];  // ll_{t+1}, ll_{t+2}: two steps in flight
  double pacc[4], pc[4], pll[4];                  // delayed lalpha store of the previous step
  int ex[4];
  const size_t K1 = (size_t)K * (Lm > 1 ? 1 : 0), K2 = (size_t)K * (Lm > 2 ? 2 : (Lm > 1 ? 1 : 0));
  // ---- t = 0
  {
    double tm[4], la0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double x0 = mod_init[jc] + ll[base[r]];
      la0[r] = vj ? x0 : NEG_INF;
      if (vj) la_out[base[r]] = la0[r];
      tm[r] = row16_max(la0[r]);
      const double x1 = ll[base[r] + K1], x2 = ll[base[r] + K2];
      lln[r] = vj ? x1 : NEG_INF;
      ll2[r] = vj ? x2 : NEG_INF;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh.tq[0][lg + 4 * r][wave][li] = tm[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      double m = sh.tq[0][w][0][0];
#pragma unroll
      for (int s2 = 1; s2 < NW; ++s2) m = fmax(m, sh.tq[0][w][s2][0]);
      c[r] = m;
      csum[r] = m;
      mant[r] = 1.0;
      ex[r] = 0;
      const double pv = vj ? exp(la0[r] - m) : 0.0;
      sh.P[0][w][j] = pv;
      sh.tsum[0][w][wave][li] = row16_sum(pv);
      sh.tmll[1][w][wave][li] = row16_max(lln[r]);
      pacc[r] = 1.0; pc[r] = 0.0; pll[r] = la0[r];   // re-stores lalpha[0] at t = 1
    }
    __syncthreads();
  }
  for (int t = 1; t < Lm; ++t) {
    const int cur = (t - 1) & 1, nxt = t & 1;
    const size_t o2 = (size_t)(t + 2 < Lm ? t + 2 : Lm - 1) * K;
    double llv[4], wgt[4], cn[4];
    // (a) everything that does not need the MFMA result
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      llv[r] = lln[r];
      lln[r] = ll2[r];   // loaded one step ago: its row max below does not wait on HBM
      const double x2 = ll[base[r] + o2];
      ll2[r] = vj ? x2 : NEG_INF;
      double tot = sh.tsum[cur][w][0][0], mll = sh.tmll[nxt][w][0][0];
#pragma unroll
      for (int s2 = 1; s2 < NW; ++s2) {
        tot += sh.tsum[cur][w][s2][0];
        mll = fmax_raw(mll, sh.tmll[nxt][w][s2][0]);
      }
      int e1, e2;
      mant[r] = frexp(mant[r] * tot, &e1);
      ex[r] += e1;
      (void)frexp(tot, &e2);
      const double d = (double)e2 * LN2_D + mll;
      cn[r] = c[r] + d;
      wgt[r] = vj ? fast_exp(llv[r] - d) : 0.0;
    }
    // (b) matrix pipe
    const double4_t acc = fb_matmul<NW>(sh, cur, li, lg, Bv);
    // (c) delayed lalpha store of step t-1 (independent of acc: overlaps the MFMAs)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double lav = fast_log(pacc[r]) + pc[r] + pll[r];
      if (FULL || vj) la_out[base[r] + (size_t)(t - 1) * K] = lav;
    }
    // (d) critical tail
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      const double pv = acc[r] * wgt[r];
      sh.P[nxt][w][j] = pv;
      sh.tsum[nxt][w][wave][li] = row16_sum(pv);
      sh.tmll[cur][w][wave][li] = row16_max(lln[r]);
      pacc[r] = acc[r]; pc[r] = c[r]; pll[r] = llv[r];
      c[r] = cn[r];
      csum[r] += cn[r];
    }
    __syncthreads();
  }
  if (Lm > 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (vj) la_out[base[r] + (size_t)(Lm - 1) * K] = fast_log(pacc[r]) + pc[r] + pll[r];
  }
  // ---- epilogue: LSE of the last step, per-window totals
  if (wave == 0 && li == 0) {
    const int last = (Lm - 1) & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      double tot = sh.tsum[last][w][0][0];
#pragma unroll
      for (int s2 = 1; s2 < NW; ++s2) tot += sh.tsum[last][w][s2][0];
      const double lz = c[r] + log(tot);
      int e1;
      const double mm = frexp(mant[r] * tot, &e1);
      local_lb[gwc[r]] = csum[r] + log(mm) + (double)(ex[r] + e1) * LN2_D;
      logz[gwc[r]] = lz;
    }
  }
}

template <int NW, bool FULL, bool WANT_LB>
__global__ __launch_bounds__(64 * NW) void k_bwd_mfma(
    const double* __restrict__ ll, const double* __restrict__ AexpT,
    const double* __restrict__ la_in, const double* __restrict__ logz, int B, int Lm, int K,
    double* __restrict__ lb_out, double* __restrict__ q_out) {
  constexpr int KS = 4 * NW;
  __shared__ FbShared<NW> sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int j = wave * 16 + li;
  const bool vj = FULL || (j < K);
  const int jc = vj ? j : 0;
  const int b0 = blockIdx.x * 16;
  double Bv[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = 8 * (kk >> 1) + 2 * lg + (kk & 1);
    Bv[kk] = (k < K && vj) ? AexpT[(size_t)k * K + jc] : 0.0;
  }
  size_t base[4];
  double sz[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gw = b0 + lg + 4 * r;
    const int g = gw < B ? gw : B - 1;
    base[r] = (size_t)g * Lm * K + jc;
    sz[r] = logz[g];
  }
  const double NEG_INF = -INFINITY;
  double c[4], eprev[4], lln[4], lan[4], ll2[4], la2[4];
  const size_t top = (size_t)(Lm - 1) * K;
  const size_t K1 = (size_t)K * (Lm > 1 ? 1 : 0), K2 = (size_t)K * (Lm > 2 ? 2 : (Lm > 1 ? 1 : 0));
  // ---- t = Lm-1: lbeta = 0
  {
    double tm[4], u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double x0 = ll[base[r] + top], a0 = la_in[base[r] + top];
      const double x1 = ll[base[r] + top - K1], x2 = ll[base[r] + top - K2];
      const double a1 = la_in[base[r] + top - K1];
      la2[r] = la_in[base[r] + top - K2];
      if (WANT_LB && vj) lb_out[base[r] + top] = 0.0;
      u[r] = vj ? x0 : NEG_INF;
      tm[r] = row16_max(u[r]);
      eprev[r] = vj ? exp(a0 - sz[r]) : 0.0;
      lln[r] = vj ? x1 : NEG_INF;
      ll2[r] = vj ? x2 : NEG_INF;
      lan[r] = a1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh.tq[1][lg + 4 * r][wave][li] = tm[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      double m = sh.tq[1][w][0][0];
#pragma unroll
      for (int s2 = 1; s2 < NW; ++s2) m = fmax(m, sh.tq[1][w][s2][0]);
      c[r] = m;
      const double pv = vj ? exp(u[r] - m) : 0.0;
      sh.P[0][w][j] = pv;
      sh.tsum[0][w][wave][li] = row16_sum(pv);
      sh.tmll[1][w][wave][li] = row16_max(lln[r]);
      sh.tq[0][w][wave][li] = row16_sum(eprev[r]);
    }
    __syncthreads();
  }
  int step = 1;
  for (int t = Lm - 2; t >= 0; --t, ++step) {
    const int cur = (step - 1) & 1, nxt = step & 1;
    const size_t o2 = (size_t)(t >= 2 ? t - 2 : 0) * K;
    double wp[4], we[4], cn[4], rq[4];
    // (a) independent of the MFMA result
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      const double llv = lln[r], lav = lan[r];
      lln[r] = ll2[r];
      const double x2 = ll[base[r] + o2];
      ll2[r] = vj ? x2 : NEG_INF;
      lan[r] = la2[r];
      la2[r] = la_in[base[r] + o2];
      double tot = sh.tsum[cur][w][0][0], mll = sh.tmll[nxt][w][0][0], totq = sh.tq[cur][w][0][0];
#pragma unroll
      for (int s2 = 1; s2 < NW; ++s2) {
        tot += sh.tsum[cur][w][s2][0];
        mll = fmax_raw(mll, sh.tmll[nxt][w][s2][0]);
        totq += sh.tq[cur][w][s2][0];
      }
      rq[r] = 1.0 / totq;
      int e2;
      (void)frexp(tot, &e2);
      const double d = (double)e2 * LN2_D + mll;
      cn[r] = c[r] + d;
      wp[r] = vj ? fast_exp(llv - d) : 0.0;                          // P'_t = acc * wp
      we[r] = vj ? fast_exp(fmin(lav + c[r] - sz[r], 700.0)) : 0.0;  // e_t  = acc * we
    }
    // (b) matrix pipe
    const double4_t acc = fb_matmul<NW>(sh, cur, li, lg, Bv);
    // (c) posterior of row t+1, normalised like hmmbase.py:226-229 (overlaps the MFMAs)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double qv = eprev[r] * rq[r];
      if (FULL || vj) q_out[base[r] + (size_t)(t + 1) * K] = qv;
    }
    // (d) critical tail
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      const double pv = acc[r] * wp[r];
      sh.P[nxt][w][j] = pv;
      eprev[r] = acc[r] * we[r];
      sh.tsum[nxt][w][wave][li] = row16_sum(pv);
      sh.tq[nxt][w][wave][li] = row16_sum(eprev[r]);
      sh.tmll[cur][w][wave][li] = row16_max(lln[r]);
    }
    if (WANT_LB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double lbv = fast_log(acc[r]) + c[r];
        if (FULL || vj) lb_out[base[r] + (size_t)t * K] = lbv;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = cn[r];
    __syncthreads();
  }
  // ---- flush the posterior of row 0
  {
    const int last = (step - 1) & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = lg + 4 * r;
      double totq = sh.tq[last][w][0][0];
#pragma unroll
      for (int s2 = 1; s2 < NW; ++s2) totq += sh.tq[last][w][s2][0];
      if (vj) q_out[base[r]] = eprev[r] / totq;
    }
  }
}

// ------------------------------------------------------------------------------------
//  K2e: scaled linear-domain sweeps -- the E-step fast path (K <= 64, B >= 192).
//  Same batching as K2c/K2d (16 windows per workgroup, wave s owns state tile 16*s), but
//  nothing in the time loop is a transcendental: the emission kernel hands over
//      Eh[t][j] = exp(ll[t][j]) * 2^-k_t,   k_t = ceil(max_j ll[t][j] / ln 2)  (integer),
//  and the messages are carried as (vector, integer binary exponent):
//      alpha_t[j] = ah_t[j] * 2^(h_t + K_t),   beta_t[j] = bh_t[j] * 2^(g_t + K_top - K_t),
//  K_t = k_0 + .. + k_t.  Forward:  ah_t = (ah_{t-1} . A) * Eh_t * 2^-e,  h_t = h_{t-1} + e,
//  e = frexp exponent of sum_i ah_{t-1}[i].  Backward: v_{t+1} = Eh_{t+1} * bh_{t+1},
//  bh_t = (A . v_{t+1}) * 2^-e,  g_t = g_{t+1} + e,  e = frexp exponent of sum_j v_{t+1}[j].
//  The posterior needs no k at all:
//      q_t[j] = ah_t[j] * bh_t[j] * 2^(h_t + g_t - h_top - zexp) / zm,   Z = zm * 2^(..+zexp)
//  (hmmbase.py:226-229 normalises by the row sum, which equals Z); it is formed by its
//  consumers (the statistics GEMM's staging threads, or k_lin_posterior for API reads).
//  The two directions are independent, so ONE launch runs both (blockIdx.y = direction):
//  a forward and a backward workgroup share each CU, two waves per SIMD, and one direction's
//  matrix work fills the other's LDS / barrier / memory-issue gaps.
//  The row sum needed for e costs one extra MFMA: lane (li, lg) adds up the 16 A-operand
//  values it loads anyway and multiplies against a ones operand, so every accumulator
//  register r holds the sum of window lg + 4r -- the layout the per-window scalars live in.
//  One barrier per step, no reduction through LDS.  local_lb = sum_t LSE_j lalpha_t[j]
//  (quirk Q4) is a running (mantissa, exponent) product.  Log-domain lalpha / lbeta / lliks
//  are materialised on demand by the log-domain kernels above.
//  Addressing: the workgroup's 16 windows are contiguous rows [b0*Lm, (b0+16)*Lm), so every
//  access is (uniform base advanced by t) + (32-bit per-lane offset fixed for the kernel) --
//  no 64-bit vector address arithmetic in the loop (host guarantees 16*Lm*K*8 < 2^32).
//  The time loop is unrolled by two with one register set per parity: a value loaded at step
//  t is first touched at step t+2, so its wait is a counted vmcnt a full step behind the
//  stores (gfx9 counts loads and stores in one in-order counter).
// ------------------------------------------------------------------------------------
// Measurement-only knock-outs of the scaled sweeps (tools/r4_overlap_probe.py builds its own copy of
// the library with -DSVIHMM_KO_SWEEP=n; never set in the product): bit 0 = no ah / bh stores, bit 1 =
// no Eh loads inside the time loop.  Results are then meaningless; the launch's time shows what the
// HBM side of the sweeps costs.
#ifndef SVIHMM_KO_SWEEP
#define SVIHMM_KO_SWEEP 0
#endif
#include "kernels_wave_linr.h"
template <int NW, typename T = double>
struct LinShared {
  static constexpr int PS = 16 * NW + 2;
  T __attribute__((aligned(16))) P[2][16][PS];
};

template <typename T> struct LPair;
template <> struct LPair<double> { typedef double2 t; };
template <> struct LPair<float> { typedef float2 t; };
// Which source state k lane group lg feeds into k-slot kk of a step's mat-vec (kk = 0 .. 4 NW - 1; the A operand
// P[window][k] and the B operand A[k][target] only have to agree).  Four state tiles (K = 64, the bench shape; round 6):
// k = 32 (lg & 1) + 16 (lg >> 1) + kk -- each lane group reads 16 CONSECUTIVE entries of the window's row, and the groups
// lg = 0 / 1 (2 / 3), which one LDS pass of a ds_read_b128 (ds_read_b64 for float) serves together, start 32 entries
// apart: with the odd slot stride of P's rows the pass touches every bank once.  Rounds 2-5 interleaved the groups
// (k = 8 (kk >> 1) + 2 lg + (kk & 1)): lanes of lg = 0 and lg = 1 met on one bank in every pass, and every operand read
// took twice its LDS cycles (tools/probe/lds_probe.hip patterns 20 / 21, profiles/r06a_lds_probe_pmc.txt).  The other
// tile counts keep the interleaved order (their rows are too short for the 32-entry distance).
template <int NW>
__device__ __forceinline__ int lin_kslot(int lg, int kk) {
#ifdef SVIHMM_AB_KSLOT_OLD      // (A/B builds only: the interleaved order of rounds 2-5)
  return 8 * (kk >> 1) + 2 * lg + (kk & 1);
#else
  return NW == 4 ? 32 * (lg & 1) + 16 * (lg >> 1) + kk : 8 * (kk >> 1) + 2 * lg + (kk & 1);
#endif
}
template <int NW, typename T>
__device__ __forceinline__ void lin_matmul(const LinShared<NW, T>& sh, int cur, int li, int lg,
                                           const T (&Bv)[4 * NW], typename LV<T>::v4& acc,
                                           typename LV<T>::v4& tot) {
  constexpr int KS = 4 * NW;
#ifdef SVIHMM_AB_KSLOT_OLD
  constexpr int PSTEP = 8;
#else
  constexpr int PSTEP = NW == 4 ? 2 : 8;        // distance of two consecutive k-slot pairs in the row
#endif
  typedef typename LV<T>::v4 v4;
  typedef typename LPair<T>::t T2;
  v4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const T* prow = &sh.P[cur][li][lin_kslot<NW>(lg, 0)];
  T s0 = 0, s1 = 0;
#pragma unroll
  for (int c = 0; c < KS / 2; c += 2) {
    const T2 x = *reinterpret_cast<const T2*>(prow + PSTEP * c);
    const T2 y = *reinterpret_cast<const T2*>(prow + PSTEP * (c + 1));
    a0 = LV<T>::mma(x.x, Bv[2 * c], a0);
    a1 = LV<T>::mma(x.y, Bv[2 * c + 1], a1);
    a2 = LV<T>::mma(y.x, Bv[2 * c + 2], a2);
    a3 = LV<T>::mma(y.y, Bv[2 * c + 3], a3);
    s0 = (c == 0) ? x.x + y.x : s0 + (x.x + y.x);
    s1 = (c == 0) ? x.y + y.y : s1 + (x.y + y.y);
  }
  const v4 z = {0, 0, 0, 0};
  tot = LV<T>::mma(s0 + s1, (T)1, z);
  acc = (a0 + a1) + (a2 + a3);
}
template <int NW, typename T>
__device__ __forceinline__ typename LV<T>::v4 lin_rowsum(const LinShared<NW, T>& sh, int cur, int li, int lg) {
  typedef typename LPair<T>::t T2;
  const T* prow = &sh.P[cur][li][2 * lg];
  T s = 0;
#pragma unroll
  for (int c = 0; c < 2 * NW; ++c) {
    const T2 x = *reinterpret_cast<const T2*>(prow + 8 * c);
    s += x.x + x.y;
  }
  const typename LV<T>::v4 z = {0, 0, 0, 0};
  return LV<T>::mma(s, (T)1, z);
}

// K > 64: the B operand (the transition matrix tile of this wave, K x 16 doubles) no longer
// fits the register budget of 16 waves per workgroup, so it is streamed from L2 every step
// (A + jc points at column jc; rows beyond K are clamped -- their P entries are zero).
template <int NW, bool FULL>
__device__ __forceinline__ void lin_matmul_stream(const LinShared<NW>& sh, int cur, int li, int lg,
                                                  const double* __restrict__ Bcol, int K,
                                                  double4_t& acc, double4_t& tot) {
  // A rolled loop with running pointers (unroll 4): fully unrolled, the compiler turns the
  // lane addresses of the streamed tile into loop invariants of the time loop and spills
  // them.  Rows beyond K come from the zeroed slack behind the matrices.
  constexpr int KS = 4 * NW;
  double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double* pr = &sh.P[cur][li][2 * lg];
  const double* __restrict__ pb = Bcol + (size_t)(2 * lg) * K;
  const size_t K1 = (size_t)K, K8 = (size_t)8 * K, K9 = (size_t)9 * K, K16 = (size_t)16 * K;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
  for (int c = 0; c < KS / 4; ++c) {
    const double b0 = pb[0], b1 = pb[K1], b2 = pb[K8], b3 = pb[K9];
    pb += K16;
    const double2 x = *reinterpret_cast<const double2*>(pr);
    const double2 y = *reinterpret_cast<const double2*>(pr + 8);
    pr += 16;
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, b0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, b1, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.x, b2, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y.y, b3, a3, 0, 0, 0);
    s0 += x.x + y.x;
    s1 += x.y + y.y;
  }
  const double4_t z = {0, 0, 0, 0};
  tot = __builtin_amdgcn_mfma_f64_16x16x4f64(s0 + s1, 1.0, z, 0, 0, 0);
  acc = (a0 + a1) + (a2 + a3);
}

// per-lane state shared by both directions.  Windows of a launch start `wstride` rows apart
// (normal batches: wstride = Lm; chain chunks overlap their terminal row: wstride = Lm - 1).
template <int NW, bool FULL, typename CT = double>
struct LinLane {
  int lane, wave, li, lg, j, jc, b0;
  bool vj;
  unsigned oE[4], oR[4], oRw;   // 32-bit element offsets of (window lg+4r, state j) / row; oRw: window lg+4*wave
  int gwc[4];
  __device__ __forceinline__ void init(int bfirst, int B, int wstride, int K, bool shared_rows) {
    lane = threadIdx.x & 63; wave = threadIdx.x >> 6;
    li = lane & 15; lg = lane >> 4;
    j = wave * 16 + li;
    vj = FULL || (j < K);
    jc = vj ? j : 0;
    b0 = bfirst;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gw = b0 + LV<CT>::crow(lg, r);
      gwc[r] = gw < B ? gw : B - 1;
      oR[r] = shared_rows ? 0u : (unsigned)(gwc[r] - b0) * (unsigned)wstride;
      oE[r] = oR[r] * (unsigned)K + (unsigned)jc;
    }
    const int gww = b0 + LV<CT>::crow(lg, wave & 3);
    oRw = shared_rows ? 0u : (unsigned)((gww < B ? gww : B - 1) - b0) * (unsigned)wstride;
  }
};
// value of register array v[r] for r == w (wave-uniform select)
__device__ __forceinline__ double sel4(const double (&v)[4], int w) {
  return w == 0 ? v[0] : (w == 1 ? v[1] : (w == 2 ? v[2] : v[3]));
}

// Chain extension (one window = the whole sequence, "100 % sequential in t").
// The scaled recursion is a product of K x K matrices G_s = A diag(Eh_s), so a long chain is
// cut into chunks of L steps and processed as an exact blocked scan:
//   S1  per chunk, M_c = G_{cL+1} ... G_{cL+L} by the forward sweep itself run on 64 unit
//       vectors (MODE 2: four workgroups of 16 pseudo-windows share the chunk's rows);
//   S2  k_chunk_scan: alpha at the chunk starts (alpha_{b+1} = alpha_b M_c) and beta at the
//       chunk ends (beta_b = M_c beta_{b+1}) -- the same matrices serve both directions;
//   S3  the sweeps below with the boundary vectors as initial / terminal condition (MODE 1):
//       every chunk is then an independent window, all of them run concurrently.
// Same arithmetic as the sequential sweep up to the association order of the products.
struct LinChain {
  const double* init_vec;   // MODE 1 fwd: [B][K] initial ah of each window, init_exp[B] its exponent
  const double* init_exp;
  const double* term_vec;   // MODE 1 bwd: [B][K] bh at the (virtual) top row, term_exp[B]
  const double* term_exp;
  const double* kbefore;    // [B] sum of the emission row exponents before the window (local_lb, logz)
  double* Mout;             // MODE 2: [chunks][16*NW][K] chunk matrix, MoutT its transpose
  double* MoutT;
  double* Mh;               // MODE 2: [chunks][16*NW] per-row exponents
};

// MODE 0: windows start from the initial distribution.  MODE 1: from chain.init_vec.
// MODE 2: chunk matrices (unit initial vectors, all pseudo-windows read the chunk's rows,
// nothing but the final matrix is stored; blockIdx.x = chunk * NW + row group).
// CT (arithmetic type): float where the storage is float (fp32 mode: ordinary window batches, B in
// registers); double everywhere else.
template <int MODE, bool BS, typename ST>
struct LinCT { typedef typename std::conditional<std::is_same<ST, float>::value && MODE == 0 && !BS, float, double>::type t; };
template <int NW, bool FULL, int MODE, bool BS = false, typename ST = double>
__device__ __forceinline__ void fwd_lin_body(
    LinShared<NW, typename LinCT<MODE, BS, ST>::t>& sh, const ST* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Aexp, const double* __restrict__ a0v,
    const double* __restrict__ a0e, int B, int Lm,
    int wstride, int K, ST* __restrict__ ah, double* __restrict__ hx,
    double* __restrict__ local_lb, double* __restrict__ logz, double2* __restrict__ zfac,
    const LinChain& ch) {
  constexpr int KS = 4 * NW;
  typedef typename LinCT<MODE, BS, ST>::t CT;
  typedef typename LV<CT>::v4 cv4;
  LinLane<NW, FULL, CT> L;
  const int chunk = MODE == 2 ? blockIdx.x / NW : 0, rgrp = MODE == 2 ? blockIdx.x % NW : 0;
  L.init(MODE == 2 ? chunk : blockIdx.x * 16, MODE == 2 ? chunk + 1 : B, wstride, K, MODE == 2);
  const int li = L.li, lg = L.lg, j = L.j, jc = L.jc, wave = L.wave;
  const bool vj = L.vj;
  CT Bv[BS ? 1 : KS];
  if (!BS) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int k = lin_kslot<NW>(lg, kk);
      Bv[BS ? 0 : kk] = (k < K && vj) ? (CT)Aexp[(size_t)k * K + jc] : (CT)0;
    }
  }
  const double* __restrict__ Bcol = Aexp + jc;
  const size_t wrow = (size_t)L.b0 * wstride;
  const ST* __restrict__ Eb = Eh + wrow * K;
  ST* __restrict__ ab = ah + wrow * K;
  double* __restrict__ hb = hx + wrow;
  const int i1 = Lm > 1 ? 1 : 0, i2 = Lm > 2 ? 2 : i1;
  double h[4], mant[4], hsum[4];
  CT ea[4], eb[4];
  int ex[4];
  {
    CT a0[4];
    if (MODE == 0) {
      // initial message and its exponent from k_lin_init (mod_init + ll_0, combined in the log domain)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a0[r] = vj ? (CT)a0v[(size_t)L.gwc[r] * K + jc] : (CT)0;
        h[r] = a0e[L.gwc[r]];
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a0[r] = vj ? (CT)ch.init_vec[(size_t)L.gwc[r] * K + jc] : (CT)0;
        h[r] = ch.init_exp[L.gwc[r]];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) { a0[r] = (vj && j == rgrp * 16 + LV<CT>::crow(lg, r)) ? (CT)1 : (CT)0; h[r] = 0.0; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ea[r] = (Eb + (size_t)i1 * K)[L.oE[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) eb[r] = (Eb + (size_t)i2 * K)[L.oE[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE != 2) {
        if (NW % 4 != 0) hb[L.oR[r]] = h[r];
        if (FULL || vj) ab[L.oE[r]] = a0[r];
      }
      sh.P[0][LV<CT>::crow(lg, r)][j] = a0[r];
      mant[r] = 1.0; ex[r] = 0; hsum[r] = 0.0;
    }
    if (MODE != 2 && NW % 4 == 0) hb[L.oRw] = sel4(h, wave & 3);
  }
  __syncthreads();
  // one time step: reads P[CUR], writes P[1-CUR]; er holds Eh_t, refilled with step t+2
  auto step = [&](const int t, auto curc, CT (&er)[4]) {
    constexpr int CUR = decltype(curc)::value, NXT = 1 - CUR;
    const int t2 = t + 2 < Lm ? t + 2 : Lm - 1;
    cv4 acc, tot;
    if constexpr (BS) lin_matmul_stream<NW, FULL>(sh, CUR, li, lg, Bcol, K, acc, tot);
    else lin_matmul<NW, CT>(sh, CUR, li, lg, Bv, acc, tot);
    ST* __restrict__ at = ab + (size_t)t * K;
    double* __restrict__ ht = hb + t;
    const ST* __restrict__ E2 = Eb + (size_t)t2 * K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e2 = LV<CT>::fexp(tot[r]);
      CT av = LV<CT>::ldx(acc[r] * er[r], -e2);
      if (BS && !FULL) av = vj ? av : (CT)0;   // streamed B has no zero columns for padded states
      sh.P[NXT][LV<CT>::crow(lg, r)][j] = av;
      if (MODE != 2) {
#if !(SVIHMM_KO_SWEEP & 1)
        if (FULL || vj) at[L.oE[r]] = av;
#endif
        // LSE of step t-1: log(tot) + (h_{t-1} + K_{t-1}) ln 2, accumulated as a product
        const double mm = mant[r] * (double)tot[r];
        ex[r] += __builtin_amdgcn_frexp_exp(mm);
        mant[r] = __builtin_amdgcn_frexp_mant(mm);
        hsum[r] += h[r];
      }
      h[r] += (double)e2;
      if (MODE != 2 && NW % 4 != 0) ht[L.oR[r]] = h[r];
    }
    if (MODE != 2 && NW % 4 == 0) ht[L.oRw] = sel4(h, wave & 3);
#if !(SVIHMM_KO_SWEEP & 2)
#pragma unroll
    for (int r = 0; r < 4; ++r) er[r] = E2[L.oE[r]];
#endif
    __syncthreads();
  };
  {
    int t = 1;
    for (; t + 1 < Lm; t += 2) {
      step(t, std::integral_constant<int, 0>{}, ea);
      step(t + 1, std::integral_constant<int, 1>{}, eb);
    }
    if (t < Lm) step(t, std::integral_constant<int, 0>{}, ea);
  }
  const int last = (Lm - 1) & 1;
  if (MODE == 2) {
    // the chunk matrix row group: M[i][j] = P[last][i - 16*rgrp][j], and its transpose
    const int Kp = 16 * NW;
    double* __restrict__ Mo = ch.Mout + (size_t)chunk * Kp * K;
    double* __restrict__ Mt = ch.MoutT + (size_t)chunk * Kp * K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = rgrp * 16 + LV<CT>::crow(lg, r);
      const double v = sh.P[last][LV<CT>::crow(lg, r)][j];
      if (vj && i < K) { Mo[(size_t)i * K + j] = v; Mt[(size_t)j * K + i] = v; }
      if (wave == 0 && li == 0 && i < K) ch.Mh[(size_t)chunk * Kp + i] = h[r];
    }
    return;
  }
  // ---- epilogue: K_top = sum_t k_t and sum_t K_t = sum_t (Lm - t) k_t per window (the only
  // place the emission exponents enter), Z = sum_j alpha_{Lm-1}[j], local_lb
  {
    double* scr = reinterpret_cast<double*>(&sh.P[1 - last][0][0]);     // free buffer: [0,16) K_top, [16,32) sum_t K_t
    const double* __restrict__ kbw = kexp + wrow;
    for (int w = threadIdx.x >> 4; w < 16; w += 4 * NW) {
      const int gw = L.b0 + w;
      const unsigned o = (unsigned)((gw < B ? gw : B - 1) - L.b0) * (unsigned)wstride;
      double a = 0.0, c = 0.0;
      for (int t = li; t < Lm; t += 16) {
        const double kv = kbw[o + t];
        a += kv;
        c += kv * (double)(Lm - t);
      }
      a = row16_sum(a);
      c = row16_sum(c);
      if (li == 0) { scr[w] = a; scr[16 + w] = c; }
    }
    __syncthreads();
    const cv4 totc = lin_rowsum<NW, CT>(sh, last, li, lg);
    if (wave == 0 && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int w = LV<CT>::crow(lg, r);
        const double totr = (double)totc[r];
        const double kb4 = (MODE == 1 && ch.kbefore) ? ch.kbefore[L.gwc[r]] : 0.0;
        const double Ktop = scr[w] + kb4, KK = scr[16 + w] + kb4 * (double)Lm;
        const double mm = mant[r] * totr;
        const int exf = ex[r] + __builtin_amdgcn_frexp_exp(mm);
        const double mf = __builtin_amdgcn_frexp_mant(mm);
        const double zm = __builtin_amdgcn_frexp_mant(totr);
        const double zexp = (double)__builtin_amdgcn_frexp_exp(totr);
        local_lb[L.gwc[r]] = log(mf) + ((double)exf + hsum[r] + h[r] + KK) * LN2_D;
        logz[L.gwc[r]] = log(zm) + (h[r] + Ktop + zexp) * LN2_D;
        zfac[L.gwc[r]] = make_double2(1.0 / zm, h[r] + zexp);
      }
    }
  }
}

// MODE 0: beta = 1 at the window's last row.  MODE 1 (chain chunks): the window has Lm rows
// of which the top one is virtual -- it belongs to the next chunk and only supplies Eh and the
// boundary vector chain.term_vec; nothing is stored for it.
template <int NW, bool FULL, int MODE, bool BS = false, typename ST = double>
__device__ __forceinline__ void bwd_lin_body(
    LinShared<NW, typename LinCT<MODE, BS, ST>::t>& sh, const ST* __restrict__ Eh, const double* __restrict__ AexpT, int B,
    int Lm, int wstride, int K, ST* __restrict__ bh, double* __restrict__ gx,
    const LinChain& ch) {
  constexpr int KS = 4 * NW;
  typedef typename LinCT<MODE, BS, ST>::t CT;
  typedef typename LV<CT>::v4 cv4;
  LinLane<NW, FULL, CT> L;
  L.init(blockIdx.x * 16, B, wstride, K, false);
  const int li = L.li, lg = L.lg, j = L.j, jc = L.jc, wave = L.wave;
  const bool vj = L.vj;
  CT Bv[BS ? 1 : KS];
  if (!BS) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int k = lin_kslot<NW>(lg, kk);
      Bv[BS ? 0 : kk] = (k < K && vj) ? (CT)AexpT[(size_t)k * K + jc] : (CT)0;
    }
  }
  const double* __restrict__ Bcol = AexpT + jc;
  const size_t wrow = (size_t)L.b0 * wstride;
  const ST* __restrict__ Eb = Eh + wrow * K;
  ST* __restrict__ bb = bh + wrow * K;
  double* __restrict__ gb = gx + wrow;
  const int top = Lm - 1;
  const int i1 = Lm > 1 ? top - 1 : top, i2 = Lm > 2 ? top - 2 : i1;
  double g[4];
  CT ea[4], eb[4];
  {
    CT e0[4], b0v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) e0[r] = (Eb + (size_t)top * K)[L.oE[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) ea[r] = (Eb + (size_t)i1 * K)[L.oE[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) eb[r] = (Eb + (size_t)i2 * K)[L.oE[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 1) {
        b0v[r] = vj ? (CT)ch.term_vec[(size_t)L.gwc[r] * K + jc] : (CT)0;
        g[r] = ch.term_exp[L.gwc[r]];
      } else {
        b0v[r] = (CT)1;
        g[r] = 0.0;
        if (NW % 4 != 0) (gb + top)[L.oR[r]] = 0.0;
        if (FULL || vj) (bb + (size_t)top * K)[L.oE[r]] = 1.0;
      }
      sh.P[0][LV<CT>::crow(lg, r)][j] = vj ? e0[r] * b0v[r] : (CT)0;
    }
    if (MODE != 1 && NW % 4 == 0) (gb + top)[L.oRw] = 0.0;
  }
  __syncthreads();
  // one step: bh of row t from P[CUR] = Eh_{t+1} * bh_{t+1}; er holds Eh_t
  auto step = [&](const int t, auto curc, CT (&er)[4]) {
    constexpr int CUR = decltype(curc)::value, NXT = 1 - CUR;
    const int t2 = t >= 2 ? t - 2 : 0;
    cv4 acc, tot;
    if constexpr (BS) lin_matmul_stream<NW, FULL>(sh, CUR, li, lg, Bcol, K, acc, tot);
    else lin_matmul<NW, CT>(sh, CUR, li, lg, Bv, acc, tot);
    ST* __restrict__ bt = bb + (size_t)t * K;
    double* __restrict__ gt = gb + t;
    const ST* __restrict__ E2 = Eb + (size_t)t2 * K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e2 = LV<CT>::fexp(tot[r]);
      CT bv = LV<CT>::ldx(acc[r], -e2);
      if (BS && !FULL) bv = vj ? bv : (CT)0;
      sh.P[NXT][LV<CT>::crow(lg, r)][j] = er[r] * bv;
#if !(SVIHMM_KO_SWEEP & 1)
      if (FULL || vj) bt[L.oE[r]] = bv;
#endif
      g[r] += (double)e2;
      if (NW % 4 != 0) gt[L.oR[r]] = g[r];
    }
    if (NW % 4 == 0) gt[L.oRw] = sel4(g, wave & 3);
#if !(SVIHMM_KO_SWEEP & 2)
#pragma unroll
    for (int r = 0; r < 4; ++r) er[r] = E2[L.oE[r]];
#endif
    __syncthreads();
  };
  {
    int t = Lm - 2;
    for (; t >= 1; t -= 2) {
      step(t, std::integral_constant<int, 0>{}, ea);
      step(t - 1, std::integral_constant<int, 1>{}, eb);
    }
    if (t == 0) step(0, std::integral_constant<int, 0>{}, ea);
  }
}

// grid (ceil(B/16), 2): blockIdx.y = 0 forward, 1 backward.  MODE 1: interior chain chunks,
// forward windows have Lm - 1 rows (the top row is the next chunk's), backward ones Lm with a
// virtual top row.  MODE 3: the chain's last chunk (boundary initial vector, ordinary end).
// MODE 2: grid (chunks * NW, 1), forward only (chunk matrices).
// ST: storage type of Eh / ah / bh (float in the fp32 mode: the launch is HBM-bound, half the
// bytes; the arithmetic of the recursion stays fp64 with its exact binary exponents).
template <int NW, bool FULL, int MODE, bool BS = false, typename ST = double>
__global__ __launch_bounds__(64 * NW) void k_sweeps_lin(
    const ST* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Aexp, const double* __restrict__ AexpT,
    const double* __restrict__ a0v, const double* __restrict__ a0e, int B, int Lm, int wstride, int K,
    ST* __restrict__ ah, ST* __restrict__ bh, double* __restrict__ hx,
    double* __restrict__ gx, double* __restrict__ local_lb, double* __restrict__ logz,
    double2* __restrict__ zfac, LinChain ch) {
  extern __shared__ double __attribute__((aligned(16))) lin_smem[];   // sizeof(LinShared<NW>) (float P: half of it used)
  typedef typename LinCT<(MODE == 3 ? 1 : MODE), BS, ST>::t CT;
  LinShared<NW, CT>& sh = *reinterpret_cast<LinShared<NW, CT>*>(lin_smem);
  if (blockIdx.y == 0)
    fwd_lin_body<NW, FULL, (MODE == 3 ? 1 : MODE), BS, ST>(sh, Eh, kexp, Aexp, a0v, a0e, B,
                                                   MODE == 1 ? Lm - 1 : Lm, wstride, K, ah, hx,
                                                   local_lb, logz, zfac, ch);
  else
    bwd_lin_body<NW, FULL, (MODE == 1 ? 1 : 0), BS, ST>(sh, Eh, AexpT, B, Lm, wstride, K, bh, gx, ch);
}

// ------------------------------------------------------------------------------------
//  K2e', wide models (128 < K <= 256): the same scaled sweeps with TWO state tiles per wave.
//  With one tile per wave a K = 256 workgroup needs 16 waves, i.e. four per SIMD and only
//  128 VGPRs each -- no room to prefetch the streamed transition tile, the kernel is then
//  bound by L2 round trips (23 us per step).  Eight waves of two tiles get 256 VGPRs, share
//  every A operand between their two MFMAs, and the compiler can keep a whole group of B
//  values in flight.  Ordinary windows only (no chain modes); both directions in one launch.
// ------------------------------------------------------------------------------------
// WT = window tiles (16 windows each) per wave.  WT = 2 (round 3, more than one 16-window
// workgroup per CU): the 32 windows share every streamed B value, the loads per MFMA halve.  It
// only pays with the exponent books kept per wave (below): with all four C registers' books in
// every wave the kernel spilled 130 registers and ran 6.7 ms on configs[4] against 6.35 ms of
// WT = 1; with one register's books per wave 5.3 ms (WT = 1: 6.1 ms).  (Two wave groups of 124
// registers sharing the B stream through L1 in one 1024-thread workgroup: 12.5 ms.)
// T = double: the fp64 kernel.  T = float (fp32 mode, round 5): Eh / ah / bh stored as float, the
// transition matrix read from its float copy, v_mfma_f32_16x16x4_f32 (C register r of lane group lg =
// window row 4 lg + r instead of lg + 4 r: LV<T>::crow); exponents and the local bound's books stay double.
// BREG (fp32 only): the wave's share of the transition matrix -- K rows x its 32 columns = 128 floats per lane --
// stays in registers for the whole sweep instead of being streamed from L2 every step (in fp64 that share is the
// whole register file, which is why the kernel streams; as floats it fits beside the accumulators).
template <int NW, bool FULL, int WT = 1, typename T = double, bool BREG = false>
__global__ __launch_bounds__(64 * NW) void k_sweeps_lin2(
    const T* __restrict__ Eh, const double* __restrict__ kexp,
    const T* __restrict__ Aexp, const T* __restrict__ AexpT,
    const double* __restrict__ a0v, const double* __restrict__ a0e, int B, int Lm, int K,
    T* __restrict__ ah,
    T* __restrict__ bh, double* __restrict__ hx, double* __restrict__ gx,
    double* __restrict__ local_lb, double* __restrict__ logz, double2* __restrict__ zfac) {
  constexpr int NT = 2 * NW, KS = 4 * NT, PS = 16 * NT + 2, WR = 16 * WT;
  static_assert(NW % 4 == 0, "a wave tracks the exponents of window rows crow(lg, wave & 3)");
  typedef typename LV<T>::v4 v4;
  typedef typename LPair<T>::t T2;
  extern __shared__ double __attribute__((aligned(16))) lin_smem[];
  typedef T PRow[PS];
  PRow* P0 = reinterpret_cast<PRow*>(lin_smem);             // P[2][WR][PS]
  auto Pb = [&](int buf) { return P0 + buf * WR; };
  const bool fwd = blockIdx.y == 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int rw = wave & 3;         // the C register whose window rows this wave keeps the books of
  const int j0 = wave * 32 + li, j1 = j0 + 16;
  const bool v0 = FULL || j0 < K, v1 = FULL || j1 < K;
  const int jc0 = v0 ? j0 : 0, jc1 = v1 ? j1 : 0;
  const int b0 = blockIdx.x * WR;
  const size_t wrow = (size_t)b0 * Lm;
  const T* __restrict__ Eb = Eh + wrow * K;
  T* __restrict__ ob = (fwd ? ah : bh) + wrow * K;      // stored vector: ah (fwd) / bh (bwd)
  double* __restrict__ xb = (fwd ? hx : gx) + wrow;          // its exponent stream
  // B operand addressing: uniform row base (scalar registers) + one 32-bit lane offset per
  // tile; rows beyond K are read from the zeroed slack the host keeps behind the matrices
  const T* __restrict__ Bm = fwd ? Aexp : AexpT;
  const int lo0 = 2 * lg * K + jc0, lo1 = 2 * lg * K + jc1;
  auto gwin = [&](int wt, int r) { const int gw = b0 + 16 * wt + LV<T>::crow(lg, r); return gw < B ? gw : B - 1; };
  unsigned oE[WT][4], oRw[WT];
#pragma unroll
  for (int wt = 0; wt < WT; ++wt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) oE[wt][r] = (unsigned)(gwin(wt, r) - b0) * (unsigned)Lm * (unsigned)K;
    oRw[wt] = (unsigned)(gwin(wt, rw) - b0) * (unsigned)Lm;
  }
  // row touched by sweep step s (s = 0: the initial row)
  auto rowof = [&](int s) { return fwd ? s : Lm - 1 - s; };
  // exponent books of window rows lg + 4 rw (every wave keeps one of the four C registers' rows;
  // the sums are exact in double): h = exponent of the stored vector, hsum = sum_s (h_s + the
  // exponent split off the running mantissa), mant = that mantissa
  double h[WT], mant[WT], hsum[WT];
  {
    const size_t ro = (size_t)rowof(0) * K;
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T e0 = (Eb + ro)[oE[wt][r] + jc0], e1 = (Eb + ro)[oE[wt][r] + jc1];
        // forward: ah_0 = the initial message of k_lin_init (stored), P = ah_0;
        // backward: bh_top = 1 (stored), P = Eh_top
        const int gw = gwin(wt, r);
        const T s0v = fwd ? (T)a0v[(size_t)gw * K + jc0] : (T)1;
        const T s1v = fwd ? (T)a0v[(size_t)gw * K + jc1] : (T)1;
        if (v0) (ob + ro)[oE[wt][r] + jc0] = s0v;
        if (v1) (ob + ro)[oE[wt][r] + jc1] = s1v;
        Pb(0)[16 * wt + LV<T>::crow(lg, r)][j0] = v0 ? (fwd ? s0v : e0) : (T)0;
        Pb(0)[16 * wt + LV<T>::crow(lg, r)][j1] = v1 ? (fwd ? s1v : e1) : (T)0;
      }
      h[wt] = fwd ? a0e[gwin(wt, rw)] : 0.0; mant[wt] = 1.0; hsum[wt] = 0.0;
      (xb + rowof(0))[oRw[wt]] = h[wt];
    }
  }
  // BREG: rows 2 lg + {0, 1, 8, 9} + 16 c of the wave's two column tiles (rows beyond K: zero)
  T breg0[BREG ? KS / 4 : 1][4], breg1[BREG ? KS / 4 : 1][4];
  if (BREG) {
#pragma unroll
    for (int c = 0; c < KS / 4; ++c) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = 2 * lg + 16 * c + (u & 1) + 8 * (u >> 1);
        breg0[c][u] = (v0 && row < K) ? Bm[(size_t)row * K + jc0] : (T)0;
        breg1[c][u] = (v1 && row < K) ? Bm[(size_t)row * K + jc1] : (T)0;
      }
    }
  }
  __syncthreads();
  for (int s = 1; s < Lm; ++s) {
    const int cur = (s - 1) & 1, nxt = s & 1;
    const int t = rowof(s);
    const size_t ro = (size_t)t * K;
    // out[w][j] = sum_i P[w][i] M[i][j] for this wave's two state tiles and WT window tiles;
    // tot[w] = sum_i P[w][i]
    v4 p0[WT], p1[WT], q0[WT], q1[WT];
    const T* pr[WT];
    T sa[WT], sb[WT];
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) {
      p0[wt] = (v4){0, 0, 0, 0}; p1[wt] = p0[wt]; q0[wt] = p0[wt]; q1[wt] = p0[wt];
      pr[wt] = &Pb(cur)[16 * wt + li][2 * lg];
      sa[wt] = 0; sb[wt] = 0;
    }
    // A rolled loop with running pointers: fully unrolled, the compiler materialises all 256
    // lane addresses of the streamed tile as loop invariants of the time loop and spills them.
    const T* __restrict__ pb0 = Bm + lo0;
    const T* __restrict__ pb1 = Bm + lo1;
    const size_t K1 = (size_t)K, K8 = (size_t)8 * K, K9 = (size_t)9 * K, K16 = (size_t)16 * K;
    auto ktrip = [&](T b00, T b01, T b02, T b03, T b10, T b11, T b12, T b13) {
#pragma unroll
      for (int wt = 0; wt < WT; ++wt) {
        const T2 x = *reinterpret_cast<const T2*>(pr[wt]);
        const T2 y = *reinterpret_cast<const T2*>(pr[wt] + 8);
        pr[wt] += 16;
        p0[wt] = LV<T>::mma(x.x, b00, p0[wt]);
        q0[wt] = LV<T>::mma(x.x, b10, q0[wt]);
        p1[wt] = LV<T>::mma(x.y, b01, p1[wt]);
        q1[wt] = LV<T>::mma(x.y, b11, q1[wt]);
        p0[wt] = LV<T>::mma(y.x, b02, p0[wt]);
        q0[wt] = LV<T>::mma(y.x, b12, q0[wt]);
        p1[wt] = LV<T>::mma(y.y, b03, p1[wt]);
        q1[wt] = LV<T>::mma(y.y, b13, q1[wt]);
        sa[wt] += x.x + y.x;
        sb[wt] += x.y + y.y;
      }
    };
    if (BREG) {
#pragma unroll
      for (int c = 0; c < KS / 4; ++c)
        ktrip(breg0[c][0], breg0[c][1], breg0[c][2], breg0[c][3], breg1[c][0], breg1[c][1], breg1[c][2], breg1[c][3]);
    } else
#pragma unroll 4
    for (int c = 0; c < KS / 4; ++c) {        // 4 k-steps (16 transition rows) per trip
      const T b00 = pb0[0], b01 = pb0[K1], b02 = pb0[K8], b03 = pb0[K9];
      const T b10 = pb1[0], b11 = pb1[K1], b12 = pb1[K8], b13 = pb1[K9];
      pb0 += K16; pb1 += K16;
#pragma unroll
      for (int wt = 0; wt < WT; ++wt) {
        const T2 x = *reinterpret_cast<const T2*>(pr[wt]);
        const T2 y = *reinterpret_cast<const T2*>(pr[wt] + 8);
        pr[wt] += 16;
        p0[wt] = LV<T>::mma(x.x, b00, p0[wt]);
        q0[wt] = LV<T>::mma(x.x, b10, q0[wt]);
        p1[wt] = LV<T>::mma(x.y, b01, p1[wt]);
        q1[wt] = LV<T>::mma(x.y, b11, q1[wt]);
        p0[wt] = LV<T>::mma(y.x, b02, p0[wt]);
        q0[wt] = LV<T>::mma(y.x, b12, q0[wt]);
        p1[wt] = LV<T>::mma(y.y, b03, p1[wt]);
        q1[wt] = LV<T>::mma(y.y, b13, q1[wt]);
        sa[wt] += x.x + y.x;
        sb[wt] += x.y + y.y;
      }
    }
    // the step's Eh values are requested BEHIND the B stream: vmcnt retires in order, so loads issued
    // in front of the GEMM loop make its first operand wait for their HBM latency, and they would
    // hold 16 WT registers through the loop
    __builtin_amdgcn_sched_barrier(0);
    T e0[WT][4], e1[WT][4];
#pragma unroll
    for (int wt = 0; wt < WT; ++wt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { e0[wt][r] = (Eb + ro)[oE[wt][r] + jc0]; e1[wt][r] = (Eb + ro)[oE[wt][r] + jc1]; }
    __builtin_amdgcn_sched_barrier(0);
    const v4 z = {0, 0, 0, 0};
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) {
      const v4 tot = LV<T>::mma(sa[wt] + sb[wt], (T)1, z);
      const v4 acc0 = p0[wt] + p1[wt], acc1 = q0[wt] + q1[wt];
      const double totw = (double)(rw == 0 ? tot[0] : (rw == 1 ? tot[1] : (rw == 2 ? tot[2] : tot[3])));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e2 = LV<T>::fexp(tot[r]);
        T o0, o1, n0, n1;      // stored value, next P value
        if (fwd) {
          o0 = v0 ? LV<T>::ldx(acc0[r] * e0[wt][r], -e2) : (T)0; o1 = v1 ? LV<T>::ldx(acc1[r] * e1[wt][r], -e2) : (T)0;
          n0 = o0; n1 = o1;
        } else {
          o0 = v0 ? LV<T>::ldx(acc0[r], -e2) : (T)0; o1 = v1 ? LV<T>::ldx(acc1[r], -e2) : (T)0;
          n0 = e0[wt][r] * o0; n1 = e1[wt][r] * o1;
        }
        Pb(nxt)[16 * wt + LV<T>::crow(lg, r)][j0] = n0;
        Pb(nxt)[16 * wt + LV<T>::crow(lg, r)][j1] = n1;
        if (v0) (ob + ro)[oE[wt][r] + jc0] = o0;
        if (v1) (ob + ro)[oE[wt][r] + jc1] = o1;
      }
      if (fwd) {
        const double mm = mant[wt] * totw;
        hsum[wt] += h[wt] + (double)__builtin_amdgcn_frexp_exp(mm);
        mant[wt] = __builtin_amdgcn_frexp_mant(mm);
      }
      h[wt] += (double)__builtin_amdgcn_frexp_exp(totw);
      (xb + t)[oRw[wt]] = h[wt];
    }
    __syncthreads();
  }
  if (!fwd) return;
  // ---- forward epilogue: K sums, Z, local_lb (as in fwd_lin_body)
  {
    const int last = (Lm - 1) & 1;
    double* scr = reinterpret_cast<double*>(&Pb(1 - last)[0][0]);   // free buffer (>= 2 WR doubles): [0, WR) K_top, [WR, 2 WR) sum_t K_t
    const double* __restrict__ kbw = kexp + wrow;
    for (int w = threadIdx.x >> 4; w < WR; w += 4 * NW) {
      const int gw = b0 + w;
      const unsigned o = (unsigned)((gw < B ? gw : B - 1) - b0) * (unsigned)Lm;
      double a = 0.0, c = 0.0;
      for (int t = li; t < Lm; t += 16) {
        const double kv = kbw[o + t];
        a += kv;
        c += kv * (double)(Lm - t);
      }
      a = row16_sum(a);
      c = row16_sum(c);
      if (li == 0) { scr[w] = a; scr[WR + w] = c; }
    }
    __syncthreads();
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) {
      T sr = 0;
      const T* prow = &Pb(last)[16 * wt + li][2 * lg];
#pragma unroll
      for (int c = 0; c < 2 * NT; ++c) {
        const T2 x = *reinterpret_cast<const T2*>(prow + 8 * c);
        sr += x.x + x.y;
      }
      const v4 z = {0, 0, 0, 0};
      const v4 tot = LV<T>::mma(sr, (T)1, z);
      if (wave < 4 && li == 0) {          // wave rw: window rows crow(lg, rw)
        const double totw = (double)(rw == 0 ? tot[0] : (rw == 1 ? tot[1] : (rw == 2 ? tot[2] : tot[3])));
        const int w = 16 * wt + LV<T>::crow(lg, rw);
        const int gw = gwin(wt, rw);
        const double Ktop = scr[w], KK = scr[WR + w];
        const double mm = mant[wt] * totw;
        const double exf = (double)__builtin_amdgcn_frexp_exp(mm);
        const double mf = __builtin_amdgcn_frexp_mant(mm);
        const double zm = __builtin_amdgcn_frexp_mant(totw);
        const double zexp = (double)__builtin_amdgcn_frexp_exp(totw);
        local_lb[gw] = log(mf) + (exf + hsum[wt] + h[wt] + KK) * LN2_D;
        logz[gw] = log(zm) + (h[wt] + Ktop + zexp) * LN2_D;
        zfac[gw] = make_double2(1.0 / zm, h[wt] + zexp);
      }
    }
  }
}

// ------------------------------------------------------------------------------------
//  K2e'', small batches (K <= 64): the scaled recursion with one wavefront per (window,
//  direction), lane = state.  Same inputs / outputs as k_sweeps_lin (Eh, kexp -> ah, hx / bh,
//  gx, local_lb, logz, zfac), so the consumers (statistics GEMM, k_lin_posterior) do not care
//  which of the two produced them.  The MFMA kernel needs >= 16 windows per workgroup and
//  ~0.9 us per step; this one has no exp / log / max on the step's critical path either
//  (LDS broadcast of the previous vector, 64 FMAs in four chains against the transition
//  column held in registers, exponent from the previous vector's sum) and takes 0.56 us per
//  step (0.6 - 0.7 with 64 - 512 windows x 2 directions in flight) -- the 64-window minibatch of
//  configs[2] and every short single chain.
// ------------------------------------------------------------------------------------
// FULLK (K == 64: every lane is a state): no per-lane predicates in the time loop, which then
// is a single basic block -- stores and the prefetched Eh load are waited for with counted
// vmcnt instead of a full drain per step.  The exponent stream is written 64 steps at a time
// (lane s & 63 keeps h_s; one coalesced store per 64 steps): 64 lanes storing one address
// every step serialise in the memory pipeline.
template <int KMAX, bool FULLK, typename ST = double>
__global__ __launch_bounds__(64) void k_wave_lin(
    const ST* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Aexp, const double* __restrict__ AexpT,
    const double* __restrict__ mod_init, const double* __restrict__ ll0, size_t l0stride, int Lm,
    int K, ST* __restrict__ ah,
    ST* __restrict__ bh, double* __restrict__ hx, double* __restrict__ gx,
    double* __restrict__ local_lb, double* __restrict__ logz, double2* __restrict__ zfac) {
  __shared__ double p_s[2][64];
  const int b = blockIdx.x, j = threadIdx.x;
  const bool fwd = blockIdx.y == 0;
  static_assert(!FULLK || KMAX == 64, "FULLK: all 64 lanes are states");
  const bool valid = FULLK || j < K;
  const int jc = valid ? j : 0;
  // fwd: a[i] = A[i][j] (column j);  bwd: a[i] = A[j][i] = AexpT[i][j]
  const double* __restrict__ Am = fwd ? Aexp : AexpT;
  double a[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) a[i] = (valid && i < K) ? Am[(size_t)i * K + jc] : 0.0;
  const size_t wrow = (size_t)b * Lm;
  const ST* __restrict__ Eb = Eh + wrow * K + jc;
  ST* __restrict__ ob = (fwd ? ah : bh) + wrow * K + jc;
  double* __restrict__ xb = (fwd ? hx : gx) + wrow;
  auto rowof = [&](int s) { return fwd ? s : Lm - 1 - s; };
  double h = 0.0, mant = 1.0, hsum = 0.0;
  int ex = 0;
  double pcur;      // the vector entering the next mat-vec: ah_{t-1} (fwd) / Eh_{t+1} bh_{t+1} (bwd)
  {
    const int t = rowof(0);
    const double e0 = Eb[(size_t)t * K];
    double o;
    if (fwd) {      // mod_init + ll_0 combined in the log domain (every wave of the window alike)
      o = lin_init_lane(mod_init, ll0 + (size_t)b * l0stride, jc, valid, kexp[wrow], h);
      pcur = o;
    } else {
      o = valid ? 1.0 : 0.0;
      pcur = valid ? e0 : 0.0;
    }
    if (valid) ob[(size_t)t * K] = o;
  }
  double hkeep = h;     // lane (s & 63) keeps the exponent of sweep step s
  // Eh rows are fetched PD steps ahead (one register each): a single window has nothing else
  // to hide the HBM latency of a long chain behind
  constexpr int PD = 4;
  auto eload = [&](int s) { return Eb[(size_t)rowof(s < Lm ? s : Lm - 1) * K]; };
  double eq[PD];
#pragma unroll
  for (int u = 0; u < PD; ++u) eq[u] = eload(1 + u);
  auto step = [&](int s, double et) {
    const int cur = s & 1;
    const int t = rowof(s);
    p_s[cur][j] = pcur;
    __syncthreads();
    // exponent and bookkeeping from the entering vector (off the mat-vec's dependency chain)
    const double tot = wave_sum_dpp(pcur);
    const int e2 = __builtin_amdgcn_frexp_exp(tot);
    if (fwd) {
      const double mm = mant * tot;
      ex += __builtin_amdgcn_frexp_exp(mm);
      mant = __builtin_amdgcn_frexp_mant(mm);
      hsum += h;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int i = 0; i < KMAX; i += 4) {
      s0 = fma(p_s[cur][i], a[i], s0);
      s1 = fma(p_s[cur][i + 1], a[i + 1], s1);
      s2 = fma(p_s[cur][i + 2], a[i + 2], s2);
      s3 = fma(p_s[cur][i + 3], a[i + 3], s3);
    }
    const double acc = (s0 + s1) + (s2 + s3);
    double o;
    if (fwd) { o = valid ? ldexp(acc * et, -e2) : 0.0; pcur = o; }
    else { o = valid ? ldexp(acc, -e2) : 0.0; pcur = et * o; }
    h += (double)e2;
    if (FULLK || valid) ob[(size_t)t * K] = o;
    hkeep = (j == (s & 63)) ? h : hkeep;
    if ((s & 63) == 63) xb[rowof(s - 63 + j)] = hkeep;      // uniform branch, coalesced store
  };
  int s = 1;
  for (; s + PD <= Lm; s += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const double et = eq[u];
      eq[u] = eload(s + u + PD);
      step(s + u, et);
    }
  }
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (s + u < Lm) step(s + u, eq[u]);
  {   // remaining exponents of the last partial group (and step 0 when Lm < 64)
    const int sl = Lm - 1, s0 = sl & ~63;
    if ((sl & 63) != 63 && s0 + j <= sl) xb[rowof(s0 + j)] = hkeep;
  }
  if (!fwd) return;
  // ---- forward epilogue: K sums over the window's rows, Z, local_lb
  double ks = 0.0, kk = 0.0;
  for (int t = j; t < Lm; t += 64) {
    const double kv = kexp[wrow + t];
    ks += kv;
    kk += kv * (double)(Lm - t);
  }
  ks = wave_sum_dpp(ks);
  kk = wave_sum_dpp(kk);
  const double tot = wave_sum_dpp(pcur);
  if (j == 0) {
    const double mm = mant * tot;
    const int exf = ex + __builtin_amdgcn_frexp_exp(mm);
    const double mf = __builtin_amdgcn_frexp_mant(mm);
    const double zm = __builtin_amdgcn_frexp_mant(tot);
    const double zexp = (double)__builtin_amdgcn_frexp_exp(tot);
    local_lb[b] = log(mf) + ((double)exf + hsum + h + kk) * LN2_D;
    logz[b] = log(zm) + (h + ks + zexp) * LN2_D;
    zfac[b] = make_double2(1.0 / zm, h + zexp);
  }
}

// ------------------------------------------------------------------------------------
//  K2f, minibatch-sized batches (K <= 64, up to a few hundred windows): the same scaled
//  recursion with FOUR wavefronts per (window, direction).  A 64-window minibatch gives the
//  one-wave kernel 128 wavefronts for 1024 SIMDs, each walking 64 broadcast reads + 64 FMAs per
//  step (0.56 us).  Here every wave keeps the whole entering vector in registers (lane l = state
//  l) and wave w owns the TARGET states [16w, 16w+16): lane (r, c) = 16 r + c forms the partial
//  sum of target 16 w + c over the source block [16r, 16r+16) -- the sources reach it as DPP
//  row broadcasts of the wave's own registers (row_newbcast:N hands lane 16r+N's value to the
//  16 lanes of row r), so a step has no LDS round trip for the mat-vec operand.  The partials go
//  through LDS once (part[r][target], conflict-free both ways), ONE workgroup barrier, and
//  every wave adds up the four partials of all 64 targets redundantly and renormalises, so the
//  next step's entering vector is already in each wave's registers.  (Round 2 split the SOURCES
//  over the waves and broadcast them through a wave-private LDS copy: two LDS round trips per
//  step, 0.43 us; this form: one.)  Same inputs / outputs as k_wave_lin, and the same
//  arithmetic per element up to the association of the 64-term sum (4 blocks x (8 + 8)).
// ------------------------------------------------------------------------------------
// acc += (lane 16 r + N of p, broadcast over row r) * a: one DP-ALU DPP instruction (the builtin
// route costs two v_mov_b32_dpp + the FMA).  FIRST: two wait states in front, the hazard between a
// VALU write of p and its DPP read is the compiler's to keep and it cannot see into the asm.
template <int N, bool FIRST = false>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double p, double a) {
  if (FIRST)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(p), "v"(a), "n"(N));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(p), "v"(a), "n"(N));
}
// The body is instantiated per direction and for K = 64 exactly (FULLK): at one wave per SIMD a
// step is instruction-issue bound, so the direction selects, validity selects and 64-bit row
// address products of a generic body are what it would spend its time on (running pointers here).
template <bool FWD, bool FULLK, typename ST>
__device__ __forceinline__ void wave_lin4_body(
    const ST* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Am, const double* __restrict__ mod_init,
    const double* __restrict__ ll0, size_t l0stride, int Lm, int K, ST* __restrict__ out,
    double* __restrict__ xout, double* __restrict__ local_lb, double* __restrict__ logz,
    double2* __restrict__ zfac, double (&part)[2][4][64]) {
  const int b = blockIdx.x, j = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = j >> 4, c = j & 15;             // source block, target within the wave's 16
  const int tgt = 16 * w + c;
  const bool valid = FULLK || j < K;
  const int jc = valid ? j : 0;
  double a[16];                                 // a[N] = A[16 r + N][tgt] (fwd) / A[tgt][16 r + N] (bwd)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int src = 16 * r + i;
    a[i] = (FULLK || (tgt < K && src < K)) ? Am[(size_t)src * K + tgt] : 0.0;
  }
  const size_t wrow = (size_t)b * Lm;
  const ptrdiff_t dstep = FWD ? (ptrdiff_t)K : -(ptrdiff_t)K;          // one sweep step in elements
  const size_t row0 = FWD ? 0 : (size_t)(Lm - 1);
  const ST* __restrict__ ep = Eh + (wrow + row0) * K + jc;            // Eh row of sweep step 0
  ST* __restrict__ op = out + (wrow + row0) * K + jc;
  double* __restrict__ xb = xout + wrow;
  auto rowof = [&](int s) { return FWD ? s : Lm - 1 - s; };
  double h = 0.0, mant = 1.0, hsum = 0.0;
  int ex = 0;
  double pcur;
  {
    double o;
    if (FWD) {      // mod_init + ll_0 combined in the log domain (every wave of the window alike)
      o = lin_init_lane(mod_init, ll0 + (size_t)b * l0stride, jc, valid, kexp[wrow], h);
      pcur = o;
    } else {
      const double e0 = *ep;
      o = valid ? 1.0 : 0.0;
      pcur = valid ? e0 : 0.0;
    }
    if (valid && w == 0) *op = o;
  }
  double hkeep = h;
  constexpr int PD = 12;    // Eh rows in flight (one register each): measured 102.8 -> 96.4 us against PD = 4
  // loads run PD steps ahead of the step: a running pointer in the main loop (where step s + PD is
  // still inside the window), clamped row arithmetic in the <= 2 PD - 1 steps at the window's end
  const ST* __restrict__ lp = ep + dstep;
  auto eclamped = [&](int s) { return ep[(ptrdiff_t)(s < Lm ? s : Lm - 1) * dstep]; };
  double eq[PD];
#pragma unroll
  for (int u = 0; u < PD; ++u) eq[u] = eclamped(1 + u);
  lp += (ptrdiff_t)PD * dstep;                  // -> row of step 1 + PD
  // Order of a step (the wave issues in order): the mat-vec on the entering vector and its LDS write
  // first -- they are the dependency chain --, then, in the shadow of that write and of the barrier,
  // everything that only has to be done by the next step: the 4-level DPP reduction for the exponent,
  // the ELBO books, and the store / exponent bookkeeping of the vector the PREVIOUS step produced
  // (`deferred`).  With the reduction and the stores in front of the mat-vec the first FMA of every
  // step waited ~150 cycles longer.
  double olast = 0.0;
  auto deferred = [&](int sp) {      // vector of step sp (>= 1): store it, keep its exponent
    op += dstep;
    if (valid && w == (sp & 3)) *op = olast;               // the four waves take turns storing
    hkeep = (j == (sp & 63)) ? h : hkeep;
    if ((sp & 63) == 63 && w == 0) xb[rowof(sp - 63 + j)] = hkeep;
  };
  auto step = [&](int s, double et) {
    const int cur = s & 1;
    double s0 = 0.0, s1 = 0.0;
    fmac_row_bcast<0, true>(s0, pcur, a[0]);
    fmac_row_bcast<1>(s1, pcur, a[1]);
#define WL4_PAIR(N) fmac_row_bcast<N>(s0, pcur, a[N]); fmac_row_bcast<N + 1>(s1, pcur, a[N + 1]);
    WL4_PAIR(2) WL4_PAIR(4) WL4_PAIR(6) WL4_PAIR(8) WL4_PAIR(10) WL4_PAIR(12) WL4_PAIR(14)
#undef WL4_PAIR
    part[cur][r][tgt] = s0 + s1;
    __builtin_amdgcn_sched_barrier(0);
    const double tot = wave_sum_dpp(pcur);
    const int e2 = __builtin_amdgcn_frexp_exp(tot);
    if (FWD) {
      const double mm = mant * tot;
      ex += __builtin_amdgcn_frexp_exp(mm);
      mant = __builtin_amdgcn_frexp_mant(mm);
      hsum += h;
    }
    if (s > 1) deferred(s - 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const double acc = (part[cur][0][j] + part[cur][1][j]) + (part[cur][2][j] + part[cur][3][j]);
    double o;
    if (FWD) { o = ldexp(acc * et, -e2); if (!FULLK) o = valid ? o : 0.0; pcur = o; }
    else { o = ldexp(acc, -e2); if (!FULLK) o = valid ? o : 0.0; pcur = et * o; }
    h += (double)e2;
    olast = o;
  };
  int s = 1;
  for (; s + 2 * PD <= Lm; s += PD) {           // every load of this trip: row s + u + PD <= Lm - 1
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const double et = eq[u];
      eq[u] = *lp;
      lp += dstep;
      step(s + u, et);
    }
  }
  for (; s + PD <= Lm; s += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const double et = eq[u];
      eq[u] = eclamped(s + u + PD);
      step(s + u, et);
    }
  }
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (s + u < Lm) step(s + u, eq[u]);
  if (Lm > 1) deferred(Lm - 1);
  if (w != 0) return;
  {
    const int sl = Lm - 1, s0 = sl & ~63;
    if ((sl & 63) != 63 && s0 + j <= sl) xb[rowof(s0 + j)] = hkeep;
  }
  if (!FWD) return;
  double ks = 0.0, kk = 0.0;
  for (int t = j; t < Lm; t += 64) {
    const double kv = kexp[wrow + t];
    ks += kv;
    kk += kv * (double)(Lm - t);
  }
  ks = wave_sum_dpp(ks);
  kk = wave_sum_dpp(kk);
  const double tot = wave_sum_dpp(pcur);
  if (j == 0) {
    const double mm = mant * tot;
    const int exf = ex + __builtin_amdgcn_frexp_exp(mm);
    const double mf = __builtin_amdgcn_frexp_mant(mm);
    const double zm = __builtin_amdgcn_frexp_mant(tot);
    const double zexp = (double)__builtin_amdgcn_frexp_exp(tot);
    local_lb[b] = log(mf) + ((double)exf + hsum + h + kk) * LN2_D;
    logz[b] = log(zm) + (h + ks + zexp) * LN2_D;
    zfac[b] = make_double2(1.0 / zm, h + zexp);
  }
}
template <int KMAX, typename ST = double>
__global__ __launch_bounds__(256) void k_wave_lin4(
    const ST* __restrict__ Eh, const double* __restrict__ kexp,
    const double* __restrict__ Aexp, const double* __restrict__ AexpT,
    const double* __restrict__ mod_init, const double* __restrict__ ll0, size_t l0stride, int Lm,
    int K, ST* __restrict__ ah,
    ST* __restrict__ bh, double* __restrict__ hx, double* __restrict__ gx,
    double* __restrict__ local_lb, double* __restrict__ logz, double2* __restrict__ zfac,
    SviSync sy = SviSync{nullptr, 0u, nullptr, nullptr, nullptr}) {
  static_assert(KMAX == 64, "lane = state, four source blocks of 16");
  __shared__ double part[2][4][64];             // [step parity][source block][target], double-buffered
  if (!svi_gate(sy)) { svi_poison(sy); return; }   // (SVI loop: the globals kernel of the side stream has arrived)
  if (blockIdx.y == 0) {
    if (K == 64) wave_lin4_body<true, true, ST>(Eh, kexp, Aexp, mod_init, ll0, l0stride, Lm, K, ah, hx, local_lb, logz, zfac, part);
    else wave_lin4_body<true, false, ST>(Eh, kexp, Aexp, mod_init, ll0, l0stride, Lm, K, ah, hx, local_lb, logz, zfac, part);
  } else {
    if (K == 64) wave_lin4_body<false, true, ST>(Eh, kexp, AexpT, mod_init, ll0, l0stride, Lm, K, bh, gx, local_lb, logz, zfac, part);
    else wave_lin4_body<false, false, ST>(Eh, kexp, AexpT, mod_init, ll0, l0stride, Lm, K, bh, gx, local_lb, logz, zfac, part);
  }
}

// S2 of the chain scan: boundary vectors.  grid 2 (0: alpha at chunk starts, 1: beta at chunk
// ends), one wavefront each, lane = state.  Chunk c spans rows [c*L, (c+1)*L] (the last one up
// to T-1); bnd row b of the outputs belongs to chain row min(b*L, T-1).
//   forward:  a_{c+1} = a_c M_c          a_0 = pi * Eh_0
//   backward: b_c = M_c b_{c+1}          b_C = 1
// with M_c[i][j] = Mm[c][i][j] * 2^Mh[c][i]; vectors renormalised to sum in [1/2, 1) with their
// own binary exponent.  Also: kbefore[c] = sum of kexp over rows < c*L, and Z / logZ.
// The dependent chain per chunk is kept short: the row exponents / their maximum are taken
// one chunk ahead, four partial accumulators, DPP + readlane reductions (no LDS), and the
// renormalisation exponent comes from the PREVIOUS vector's sum (any integer keeps the
// bookkeeping exact; the sum only has to stay in range), so no reduction sits between two
// mat-vecs.
// Workgroup = 4 waves.  Wave 0 runs the chain; waves 1..3 stream the chunk matrices from HBM
// into a 4-slot LDS ring three chunks ahead (each helper owns every third chunk: it issues
// the loads on its turn and writes them to LDS two iterations later), so the HBM latency of
// a 32 KB matrix is off the sequential path.  One barrier per chunk.
template <int KMAX>
__device__ __forceinline__ double scan_matvec(const double* __restrict__ ms, int K, int lane, double w) {
  pin_all_lanes(w);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
  for (int i = 0; i < KMAX; i += 4) {
    a0 = fma(readlane_f64(w, i), ms[(i) * 64 + lane], a0);
    a1 = fma(readlane_f64(w, i + 1), ms[(i + 1) * 64 + lane], a1);
    a2 = fma(readlane_f64(w, i + 2), ms[(i + 2) * 64 + lane], a2);
    a3 = fma(readlane_f64(w, i + 3), ms[(i + 3) * 64 + lane], a3);
  }
  double r = (a0 + a1) + (a2 + a3);
  pin_all_lanes(r);
  return r;
}

template <int KMAX>
__global__ __launch_bounds__(256) void k_chunk_scan(
    const double* __restrict__ Mm, const double* __restrict__ MmT, const double* __restrict__ Mh,
    int C, int Kp, int K, const double* __restrict__ Eh, const double* __restrict__ ksum,
    const double* __restrict__ a0v, const double* __restrict__ a0e, double* __restrict__ abnd,
    double* __restrict__ aexp,
    double* __restrict__ bbnd, double* __restrict__ bexp, double* __restrict__ kbefore,
    double2* __restrict__ zfac, double* __restrict__ logz) {
  extern __shared__ double ring[];            // [4][KMAX][64]
  constexpr int SLOT = KMAX * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vl = lane < K;
  const int lc = vl ? lane : 0;
  const size_t MS = (size_t)Kp * K;
  const bool fwd = blockIdx.x == 0;
  const double* __restrict__ Msrc = fwd ? Mm : MmT;
  // position p = 0..C-1 in processing order -> chunk index
  auto chunk_of = [&](int p) { return fwd ? p : C - 1 - p; };
  // helper: load chunk at position p into registers / write registers to its ring slot
  double hr[KMAX];
  auto h_load = [&](int p) {
    const double* __restrict__ M = Msrc + (size_t)chunk_of(p < C ? p : C - 1) * MS;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) hr[i] = M[(size_t)(i < K ? i : K - 1) * K + lc];
  };
  auto h_store = [&](int p) {
    double* __restrict__ dst = ring + (p & 3) * SLOT;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) dst[i * 64 + lane] = hr[i];
  };
  // prologue: helper h brings in position h-1 (slots 0..2)
  if (wave >= 1) { h_load(wave - 1); h_store(wave - 1); }
  double a = 0.0, Hx = 0.0, mh = 0.0, hmax = 0.0;
  int e_lag = 0;
  if (wave == 0) {
    if (fwd) {
      a = vl ? a0v[lc] : 0.0;      // initial message of k_lin_init (chain = window 0)
      Hx = a0e[0];
      if (vl) abnd[lane] = a;
      if (lane == 0) aexp[0] = Hx;
      mh = vl ? Mh[lane] : -INFINITY;
      hmax = wave_max_dpp(mh);
    } else {
      a = vl ? 1.0 : 0.0;
      if (vl) bbnd[(size_t)C * K + lane] = a;
      if (lane == 0) bexp[C] = 0.0;
    }
    e_lag = __builtin_amdgcn_frexp_exp(wave_sum_dpp(a));
  }
  // helper turn bookkeeping: helper h loads positions p = h-1+3, h-1+6, ... ; the load for
  // position p is issued at iteration p-3 and written at iteration p-1
  if (wave >= 1) h_load(wave - 1 + 3);
  __syncthreads();
  for (int p = 0; p < C; ++p) {
    if (wave == 0) {
      const double* __restrict__ ms = ring + (p & 3) * SLOT;
      const int c = chunk_of(p);
      if (fwd) {
        // a <- a M_c
        const double w = vl ? ldexp(a, (int)(mh - hmax) - e_lag) : 0.0;
        const double acc = scan_matvec<KMAX>(ms, K, lane, w);
        Hx += hmax + (double)e_lag;
        a = vl ? acc : 0.0;
        if (vl) abnd[(size_t)(c + 1) * K + lane] = a;
        if (lane == 0) aexp[c + 1] = Hx;
        e_lag = __builtin_amdgcn_frexp_exp(wave_sum_dpp(a));
        mh = vl ? Mh[(size_t)(c + 1 < C ? c + 1 : c) * Kp + lane] : -INFINITY;
        hmax = wave_max_dpp(mh);
      } else {
        // b <- M_c b = 2^Mh[i] * sum_j M[i][j] b[j]  (rows of the transposed copy)
        const double w = vl ? ldexp(a, -e_lag) : 0.0;
        const double acc = scan_matvec<KMAX>(ms, K, lane, w);
        const double mhc = vl ? Mh[(size_t)c * Kp + lane] : -INFINITY;
        const double hm = wave_max_dpp(mhc);
        a = vl ? ldexp(acc, (int)(mhc - hm)) : 0.0;
        Hx += hm + (double)e_lag;
        if (vl) bbnd[(size_t)c * K + lane] = a;
        if (lane == 0) bexp[c] = Hx;
        e_lag = __builtin_amdgcn_frexp_exp(wave_sum_dpp(a));
      }
    } else {
      // position q = p + 2 is written now by its owner (slot (p+2)&3 was freed at iteration
      // p - 2), then the owner issues the loads of q + 3
      const int q = p + 2;
      if ((q % 3) == wave - 1 && q >= 3) {
        if (q < C) h_store(q);
        h_load(q + 3);
      }
    }
    __syncthreads();
  }
  if (wave == 0 && fwd) {
    // emission exponents before each chunk (prefix sums of the per-chunk sums)
    double kb = 0.0;
    if (lane == 0)
      for (int cc = 0; cc < C; ++cc) { kbefore[cc] = kb; kb += ksum[cc]; }
    kb = readlane_f64(kb, 0);
    const double tot = wave_sum_dpp(vl ? a : 0.0);   // a is alpha at the last row: Z
    if (lane == 0) {
      const double zm = __builtin_amdgcn_frexp_mant(tot);
      const double zexp = (double)__builtin_amdgcn_frexp_exp(tot);
      zfac[0] = make_double2(1.0 / zm, Hx + zexp);
      logz[0] = log(zm) + (Hx + zexp + kb) * LN2_D;
    }
  }
}

// sum of the emission row exponents of each chunk's own rows [c*Lc, (c+1)*Lc) (last: up to T)
// S2 for 64 < K <= 256: same recurrences and outputs as k_chunk_scan, thread = state, the chunk
// matrix streamed straight from HBM / L2 (512 KB per chunk at K = 256: ~10 us per chunk, a few
// ms per million rows -- S1's T K^3 flops dominate the wide scan by two orders of magnitude,
// so no ring, no helper waves).  grid 2 (forward / backward), block 256.
__device__ __forceinline__ double block256_sum(double v, double* red) {
  v = wave_sum_dpp(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block256_max(double v, double* red) {
  v = wave_max_dpp(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__global__ __launch_bounds__(256) void k_chunk_scan_wide(
    const double* __restrict__ Mm, const double* __restrict__ MmT, const double* __restrict__ Mh,
    int C, int Kp, int K, const double* __restrict__ Eh, const double* __restrict__ ksum,
    const double* __restrict__ a0v, const double* __restrict__ a0e, double* __restrict__ abnd,
    double* __restrict__ aexp,
    double* __restrict__ bbnd, double* __restrict__ bexp, double* __restrict__ kbefore,
    double2* __restrict__ zfac, double* __restrict__ logz) {
  __shared__ double ws[256];
  __shared__ double red[4];
  const int j = threadIdx.x;
  const bool vl = j < K;
  const int jc = vl ? j : 0;
  const size_t MS = (size_t)Kp * K;
  const bool fwd = blockIdx.x == 0;
  const double* __restrict__ Msrc = fwd ? Mm : MmT;
  auto matvec = [&](const double* __restrict__ M, double w) {
    __syncthreads();
    ws[j] = w;
    __syncthreads();
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double* __restrict__ col = M + jc;
    int i = 0;
    for (; i + 16 <= K; i += 16) {
      double m[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) m[u] = col[(size_t)(i + u) * K];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        a0 = fma(ws[i + u], m[u], a0);
        a1 = fma(ws[i + u + 1], m[u + 1], a1);
        a2 = fma(ws[i + u + 2], m[u + 2], a2);
        a3 = fma(ws[i + u + 3], m[u + 3], a3);
      }
    }
    for (; i < K; ++i) a0 = fma(ws[i], col[(size_t)i * K], a0);
    return (a0 + a1) + (a2 + a3);
  };
  double a, Hx = 0.0;
  int e_lag;
  if (fwd) {
    a = vl ? a0v[jc] : 0.0;      // initial message of k_lin_init (chain = window 0)
    Hx = a0e[0];
    if (vl) abnd[j] = a;
    if (j == 0) aexp[0] = Hx;
  } else {
    a = vl ? 1.0 : 0.0;
    if (vl) bbnd[(size_t)C * K + j] = a;
    if (j == 0) bexp[C] = 0.0;
  }
  e_lag = __builtin_amdgcn_frexp_exp(block256_sum(a, red));
  for (int p = 0; p < C; ++p) {
    const int c = fwd ? p : C - 1 - p;
    const double mh = vl ? Mh[(size_t)c * Kp + j] : -INFINITY;
    const double hm = block256_max(mh, red);
    if (fwd) {       // a <- a M_c, rows of M_c carry their own exponents
      const double w = vl ? ldexp(a, (int)(mh - hm) - e_lag) : 0.0;
      const double acc = matvec(Msrc + (size_t)c * MS, w);
      Hx += hm + (double)e_lag;
      a = vl ? acc : 0.0;
      if (vl) abnd[(size_t)(c + 1) * K + j] = a;
      if (j == 0) aexp[c + 1] = Hx;
    } else {         // b <- M_c b (rows of the transposed copy)
      const double w = vl ? ldexp(a, -e_lag) : 0.0;
      const double acc = matvec(Msrc + (size_t)c * MS, w);
      a = vl ? ldexp(acc, (int)(mh - hm)) : 0.0;
      Hx += hm + (double)e_lag;
      if (vl) bbnd[(size_t)c * K + j] = a;
      if (j == 0) bexp[c] = Hx;
    }
    e_lag = __builtin_amdgcn_frexp_exp(block256_sum(a, red));
  }
  if (fwd) {
    const double tot = block256_sum(vl ? a : 0.0, red);
    if (j == 0) {
      double kb = 0.0;
      for (int cc = 0; cc < C; ++cc) { kbefore[cc] = kb; kb += ksum[cc]; }
      const double zm = __builtin_amdgcn_frexp_mant(tot);
      const double zexp = (double)__builtin_amdgcn_frexp_exp(tot);
      zfac[0] = make_double2(1.0 / zm, Hx + zexp);
      logz[0] = log(zm) + (Hx + zexp + kb) * LN2_D;
    }
  }
}

__global__ __launch_bounds__(64) void k_chunk_ksum(const double* __restrict__ kexp, int C, int Lc,
                                                  int64_t T, double* __restrict__ ksum) {
  const int c = blockIdx.x;
  const int64_t r0 = (int64_t)c * Lc, r1 = (c == C - 1) ? T : r0 + Lc;
  double ks = 0.0;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 64) ks += kexp[r];
  ks = wave_sum_dpp(ks);
  if (threadIdx.x == 0) ksum[c] = ks;
}

// posterior marginals from the scaled messages (API reads of var_x; the statistics GEMM
// forms the same product in its staging threads and never needs this array).
// One 16-lane row per (window, t) row.
// Initial message of every window of the scaled sweeps.  The reference forms
// lalpha_0 = mod_init + lliks_0 in the log domain (hmmbase.py:292, hmmsgd_metaobs.py:800); a
// product of two separately shifted exponentials loses the row when the state that carries the
// first observation is rare in the initial distribution (psi(var_init) of a state with
// stationary mass 1e-3 is -1000) -- so the sum is taken here, in the log domain, from the first
// row's unscaled log-likelihoods (ll0: row b at b * stride), and scaled afterwards:
//   a0[b][j] = exp(lalpha_0[j] - H_b ln 2),  H_b = ceil(max_j lalpha_0[j] / ln 2),
//   a0exp[b] = H_b - k_0   (alpha_0 = a0 * 2^(a0exp + k_0), k_0 = the emission exponent of row 0).
// One wavefront per window.
__global__ __launch_bounds__(256) void k_lin_init(const double* __restrict__ mod_init,
                                                  const double* __restrict__ ll0, size_t stride,
                                                  const double* __restrict__ kexp, int B, int Lm, int K,
                                                  double* __restrict__ a0, double* __restrict__ a0exp) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const double* __restrict__ l0 = ll0 + (size_t)b * stride;
  double m = -INFINITY;
  for (int j = lane; j < K; j += 64) m = fmax(m, mod_init[j] + l0[j]);
  m = wave_max(m);
  const double H = (m > -1e300 && m < 1e300) ? ceil(m * LOG2E_D) : 0.0;
  for (int j = lane; j < K; j += 64)
    a0[(size_t)b * K + j] = exp(fma(-H, LN2_LO_D, fma(-H, LN2_HI_D, mod_init[j] + l0[j])));
  if (lane == 0) a0exp[b] = H - kexp[(size_t)b * Lm];
}

template <int KT, typename ST = double>
__global__ __launch_bounds__(256) void k_lin_posterior(
    const ST* __restrict__ ah, const ST* __restrict__ bh, const double* __restrict__ hx,
    const double* __restrict__ gx, const double2* __restrict__ zfac, int64_t nrows, int Lm,
    int K, double* __restrict__ q) {
  const int li = threadIdx.x & 15;
  const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (g >= nrows) return;
  const double2 zf = zfac[g / Lm];
  const double s = ldexp(zf.x, (int)(hx[g] + gx[g] - zf.y));
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int k = li + 16 * c;
    if (k < K) q[g * K + k] = ((double)ah[g * K + k] * (double)bh[g * K + k]) * s;
  }
}

// host-supplied lliks (generic emission plugins) -> (Eh, kexp) for the scaled sweeps.
// One 16-lane row per (window, t) row, KT = ceil(K/16) values per lane.
template <int KT>
__global__ __launch_bounds__(256) void k_scale_ll(const double* __restrict__ ll, int64_t nrows,
                                                  int K, double* __restrict__ Eh,
                                                  double* __restrict__ kexp) {
  const int li = threadIdx.x & 15;
  const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t gc = g < nrows ? g : nrows - 1;
  double v[KT], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int k = li + 16 * c;
    v[c] = k < K ? ll[gc * K + k] : -INFINITY;
    mx = fmax_raw(mx, v[c]);
  }
  mx = row16_max(mx);
  const double kx = (mx > -1e300 && mx < 1e300) ? ceil(mx * LOG2E_D) : 0.0;
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int k = li + 16 * c;
    if (k < K && g < nrows)
      Eh[g * K + k] = fast_exp(fma(-kx, LN2_LO_D, fma(-kx, LN2_HI_D, v[c])));
  }
  if (li == 0 && g < nrows) kexp[g] = kx;
}

// fp32 mode, wide models (round 5): the plain float log-likelihoods of k_emission_bf16x3d<true> -> the scaled
// float Eh IN PLACE (a row is read completely by its 16 lanes before it is written) + the row exponents
template <int KT>
__global__ __launch_bounds__(256) void k_scale_ll_f32(float* __restrict__ ll, int64_t nrows, int K,
                                                      double* __restrict__ kexp) {
  const int li = threadIdx.x & 15;
  const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t gc = g < nrows ? g : nrows - 1;
  float v[KT], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int k = li + 16 * c;
    v[c] = k < K ? ll[gc * K + k] : -INFINITY;
    mx = fmaxf(mx, v[c]);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const double kx = (fabsf(mx) < 1e30f) ? ceil((double)mx * LOG2E_D) : 0.0;
  const float fr = (float)fma((double)mx, LOG2E_D, -kx);       // in (-1, 0], formed in double (see k_emission_bf16x3)
#pragma unroll
  for (int c = 0; c < KT; ++c) {
    const int k = li + 16 * c;
    if (k < K && g < nrows) ll[g * K + k] = __builtin_amdgcn_exp2f(fmaf(v[c] - mx, 1.44269504f, fr));
  }
  if (li == 0 && g < nrows) kexp[g] = kx;
}
__global__ __launch_bounds__(256) void k_f64_to_f32(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (float)src[i];
}
__global__ void k_sum_lb(const double* __restrict__ local_lb, int B, double* __restrict__ lb_total) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) acc += local_lb[b];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *lb_total = red[0];
}

// ------------------------------------------------------------------------------------
//  K3: posterior marginals q = softmax_k(la+lb) and per-row LSE_k(la) partial sums.
//      grid (B*nseg), block 256 = 4 waves, one wave per row; seg = rps rows (the host picks
//      short segments for small batches so that a 64-window minibatch still fills the chip).
// ------------------------------------------------------------------------------------
template <int KPL>  // states per lane (K <= 64*KPL)
__global__ __launch_bounds__(256) void k_posterior(const double* __restrict__ la,
                                                   const double* __restrict__ lb, int Lm,
                                                   int K, int nseg, int rps,
                                                   double* __restrict__ q,
                                                   double* __restrict__ lse_part) {
  __shared__ double wsum[4];
  const int b = blockIdx.x / nseg, seg = blockIdx.x - b * nseg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = seg * rps;
  const int t1 = min(Lm, t0 + rps);
  double lse_acc = 0.0;
  for (int t = t0 + wave; t < t1; t += 4) {
    const size_t base = ((size_t)b * Lm + t) * K;
    double u[KPL], a[KPL];
    double mu = -INFINITY, ma = -INFINITY;
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
      const int k = lane + 64 * c;
      if (k < K) {
        a[c] = la[base + k];
        u[c] = a[c] + lb[base + k];
      } else {
        a[c] = -INFINITY;
        u[c] = -INFINITY;
      }
      mu = fmax(mu, u[c]);
      ma = fmax(ma, a[c]);
    }
    mu = wave_max(mu);
    ma = wave_max(ma);
    double su = 0.0, sa = 0.0;
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
      u[c] = exp(u[c] - mu);
      su += u[c];
      sa += exp(a[c] - ma);
    }
    su = wave_sum(su);
    sa = wave_sum(sa);
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
      const int k = lane + 64 * c;
      if (k < K) q[base + k] = u[c] / su;
    }
    lse_acc += ma + log(sa);
  }
  if (lane == 0) wsum[wave] = lse_acc;
  __syncthreads();
  if (threadIdx.x == 0)
    lse_part[(size_t)b * nseg + seg] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// ------------------------------------------------------------------------------------
//  K6: FFBS backward sampling (hmm_fast.pyx:97-122), one wavefront, K <= 64 in-lane,
//      larger K through a serial tail in lane 0.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ffbs_sample(const double* __restrict__ la,
                                                    const double* __restrict__ logA,
                                                    const double* __restrict__ unif, int64_t T,
                                                    int K, int64_t* __restrict__ z) {
  const int lane = threadIdx.x;
  extern __shared__ double ps[];  // [K] for K > 64
  int64_t znext = 0;
  for (int64_t t = T - 1; t >= 0; --t) {
    if (K <= 64) {
      double lp = -INFINITY;
      if (lane < K) {
        lp = la[t * K + lane];
        if (t < T - 1) lp += logA[(size_t)lane * K + znext];
      }
      const double m = wave_max(lp);
      double p = (lane < K) ? exp(lp - m) : 0.0;
      const double tot = wave_sum(p);
      p /= tot;
      // inclusive scan in lane order
      double c = p;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double v = __shfl_up(c, o, 64);
        if (lane >= o) c += v;
      }
      const double r = unif[t];
      const unsigned long long bal = __ballot(lane < K && r <= c);
      int zz = bal ? (__ffsll((long long)bal) - 1) : (K - 1);
      znext = zz;
    } else {
      double mloc = -INFINITY;
      for (int k = lane; k < K; k += 64) {
        double lp = la[t * K + k];
        if (t < T - 1) lp += logA[(size_t)k * K + znext];
        ps[k] = lp;
        mloc = fmax(mloc, lp);
      }
      const double m = wave_max(mloc);
      double sl = 0.0;
      for (int k = lane; k < K; k += 64) {
        const double e = exp(ps[k] - m);
        ps[k] = e;
        sl += e;
      }
      const double tot = wave_sum(sl);
      __syncthreads();
      int zz = K - 1;
      if (lane == 0) {
        const double r = unif[t];
        double rs = 0.0;
        for (int k = 0; k < K; ++k) {
          rs += ps[k] / tot;
          if (r <= rs) { zz = k; break; }
        }
      }
      zz = __shfl(zz, 0, 64);
      znext = zz;
      __syncthreads();
    }
    if (lane == 0) z[t] = znext;
  }
}

// ------------------------------------------------------------------------------------
//  K6b: FFBS for long chains (hmm_fast.pyx:43-124) without a sequential pass over T.
//  Forward filter: the exact blocked scan of the scaled sweeps (launch_fb_chain), turned into
//  lalpha by k_chain_lalpha:  lalpha_t[j] = log(ah_t[j]) + (h_t + K_t) ln 2,  K_t the running
//  sum of the emission row exponents (integers, so any summation order is exact).
//  Backward sampling: z_t = F_t(z_{t+1}) with F_t the inverse-CDF draw of
//  softmax_k(lalpha[t,k] + logA[k, z_{t+1}]) at the recorded uniform u_t -- a composition of
//  maps {0..K-1} -> {0..K-1}, which is associative:
//    P1 k_ffbs_paths: one wave per chunk of rows, lane = the chunk's entry state (the state
//       at the first row of the next chunk); every lane walks its own path down the chunk
//       and records it, path[t][entry] (one byte).  As soon as all lanes agree (the paths
//       couple, after a few rows on peaked posteriors) the wave switches to one cooperative
//       draw per row (lane = state, as K6); coupled paths stay coupled.
//    P2 k_ffbs_compose: suffix scan of the chunk maps G_c = path[first row of c][.]
//       (Hillis-Steele, log2 C rounds) -> the entry state of every chunk.
//    P3 k_ffbs_gather: z[t] = path[t][entry of t's chunk].
//  The draws use  u_t * sum_k p_k <= cumsum_k p_k  (no division); within 1 ulp of a CDF
//  boundary the state can differ from K6's -- as it can between K6 and NumPy's summation.
// ------------------------------------------------------------------------------------
// ktop != nullptr: the backward messages instead,  lbeta_t[i] = log(bh_t[i]) + (g_t + K_top - K_t) ln 2
// (ah = bh, hx = gx).
__global__ void k_ksum_all(const double* __restrict__ kexp, int64_t T, double* __restrict__ ktop) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t t = threadIdx.x; t < T; t += 256) acc += kexp[t];     // integers: exact in any order
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *ktop = red[0];
}
__global__ __launch_bounds__(256) void k_chain_lalpha(
    const double* __restrict__ ah, const double* __restrict__ hx, const double* __restrict__ kexp,
    const double* __restrict__ kbefore, int L, int C, int64_t T, int K, double* __restrict__ out,
    const double* __restrict__ ktop = nullptr) {
  // entries whose scaled message underflowed (ah = 0 or denormal: more than ~460 nats below the
  // row's total) come out as -inf / imprecise here; k_lalpha_fix recomputes exactly those
  __shared__ double kc[1025];
  __shared__ double part[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int64_t r0 = (int64_t)c * L;
  const int n = (int)((c == C - 1 ? T : r0 + L) - r0);       // <= L + 1 <= 1025
  const int per = (n + 255) / 256;
  const int i0 = tid * per, i1 = i0 + per < n ? i0 + per : n;
  double s = 0.0;
  for (int i = i0; i < i1; ++i) s += kexp[r0 + i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    double run = kbefore[c];
    for (int i = 0; i < 256; ++i) { const double v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  s = part[tid];
  for (int i = i0; i < i1; ++i) { s += kexp[r0 + i]; kc[i] = s; }
  __syncthreads();
  for (int64_t e = tid; e < (int64_t)n * K; e += 256) {
    const int i = (int)(e / K);
    const double hk = hx[r0 + i] + (ktop ? *ktop - kc[i] : kc[i]);
    out[r0 * K + e] = fma(hk, LN2_HI_D, fma(hk, LN2_LO_D, log(ah[r0 * K + e])));
  }
}

// lalpha entries the scaled messages cannot represent (ah < 1e-200 relative to a row total of
// order one) recomputed in the log domain from the previous row:
//   lalpha_t[j] = LSE_i(lalpha_{t-1}[i] + ltran[i,j]) + ll_t[j]       (hmmbase.py:292-295)
// The LSE is carried by the row's large entries, which `src` (k_chain_lalpha's output) has to
// full precision; what the underflowed entries of row t-1 would add is below e^-460 relative.
// Every row independently: one wave per row, lane = state j, ltran column in registers.
// dst = src where no fix is needed (separate buffers: no read/write overlap between rows).
template <int KMAX>
__global__ __launch_bounds__(256) void k_lalpha_fix(
    const double* __restrict__ src, const double* __restrict__ ah, const double* __restrict__ ll,
    const double* __restrict__ ltran, const double* __restrict__ mod_init, int64_t T, int K,
    double* __restrict__ dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vj = lane < K;
  const int jc = vj ? lane : 0;
  double lt[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) lt[i] = (i < K && vj) ? ltran[(size_t)i * K + jc] : -INFINITY;
  const int64_t rows_per_block = 64;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int64_t t = r0 + wave; t < r0 + rows_per_block && t < T; t += 4) {
    const double a = vj ? ah[t * K + lane] : 1.0;
    double v = vj ? src[t * K + lane] : 0.0;
    const bool need = vj && a < 1e-200;
    if (__ballot(need) != 0ull) {
      if (t == 0) {
        if (need) v = mod_init[lane] + ll[lane];
      } else {
        const double* __restrict__ pr = src + (t - 1) * K;     // uniform row: scalar loads
        double m = -INFINITY;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) m = fmax(m, (i < K ? pr[i] : -INFINITY) + lt[i]);
        double sm = 0.0;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) sm += exp((i < K ? pr[i] : -INFINITY) + lt[i] - m);
        const double fixed = (m > -INFINITY ? m + log(sm) : -INFINITY) + ll[t * K + jc];
        if (need) v = fixed;
      }
    }
    if (vj) dst[t * K + lane] = v;
  }
}

// the backward counterpart:  lbeta_t[i] = LSE_j(ltran[i,j] + lbeta_{t+1}[j] + ll_{t+1}[j])
// (hmmbase.py:316-320), lbeta_{T-1} = 0; lane = state i, ltran row in registers.
template <int KMAX>
__global__ __launch_bounds__(256) void k_lbeta_fix(
    const double* __restrict__ src, const double* __restrict__ bh, const double* __restrict__ ll,
    const double* __restrict__ ltran, int64_t T, int K, double* __restrict__ dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vi = lane < K;
  const int ic = vi ? lane : 0;
  double lt[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) lt[j] = (j < K && vi) ? ltran[(size_t)ic * K + j] : -INFINITY;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int64_t t = r0 + wave; t < r0 + 64 && t < T; t += 4) {
    const double b = vi ? bh[t * K + lane] : 1.0;
    double v = vi ? src[t * K + lane] : 0.0;
    const bool need = vi && b < 1e-200;
    if (__ballot(need) != 0ull) {
      if (t == T - 1) {
        if (need) v = 0.0;
      } else {
        const double* __restrict__ pr = src + (t + 1) * K;
        const double* __restrict__ pl = ll + (t + 1) * K;
        double m = -INFINITY;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) m = fmax(m, (j < K ? pr[j] + pl[j] : -INFINITY) + lt[j]);
        double sm = 0.0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) sm += exp((j < K ? pr[j] + pl[j] : -INFINITY) + lt[j] - m);
        const double fixed = m > -INFINITY ? m + log(sm) : -INFINITY;
        if (need) v = fixed;
      }
    }
    if (vi) dst[t * K + lane] = v;
  }
}

// The same two repairs for wide models (64 < K <= 1024): one wave per row, lane = states lane + 64 q;
// the transition expectations are read from L2 (column / row accesses coalesce over the lanes) --
// only rows that hold an underflowed entry pay for it.
__global__ __launch_bounds__(256) void k_lalpha_fix_wide(
    const double* __restrict__ src, const double* __restrict__ ah, const double* __restrict__ ll,
    const double* __restrict__ ltran, const double* __restrict__ mod_init, int64_t T, int K,
    double* __restrict__ dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int64_t t = r0 + wave; t < r0 + 64 && t < T; t += 4) {
    bool any = false;
    for (int j = lane; j < K; j += 64) any = any || ah[t * K + j] < 1e-200;
    const bool fix_row = __ballot(any) != 0ull;
    const double* __restrict__ pr = src + (t > 0 ? t - 1 : 0) * K;
    for (int j0 = 0; j0 < K; j0 += 64) {
      const int j = j0 + lane;
      const bool vj = j < K;
      const int jc = vj ? j : 0;
      double v = vj ? src[t * K + j] : 0.0;
      const bool need = fix_row && vj && ah[t * K + j] < 1e-200;
      if (__ballot(need) != 0ull) {
        double fixed;
        if (t == 0) fixed = mod_init[jc] + ll[jc];
        else {
          double m = -INFINITY;
          for (int i = 0; i < K; ++i) m = fmax(m, pr[i] + ltran[(size_t)i * K + jc]);
          double sm = 0.0;
          for (int i = 0; i < K; ++i) sm += exp(pr[i] + ltran[(size_t)i * K + jc] - m);
          fixed = (m > -INFINITY ? m + log(sm) : -INFINITY) + ll[t * K + jc];
        }
        if (need) v = fixed;
      }
      if (vj) dst[t * K + j] = v;
    }
  }
}
__global__ __launch_bounds__(256) void k_lbeta_fix_wide(
    const double* __restrict__ src, const double* __restrict__ bh, const double* __restrict__ ll,
    const double* __restrict__ ltran, int64_t T, int K, double* __restrict__ dst) {
  extern __shared__ double fw_s[];              // [4 waves][K]: lbeta_{t+1} + ll_{t+1} of the wave's row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* __restrict__ nx = fw_s + (size_t)wave * K;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int64_t t = r0 + wave; t < r0 + 64 && t < T; t += 4) {
    bool any = false;
    for (int i = lane; i < K; i += 64) any = any || bh[t * K + i] < 1e-200;
    const bool fix_row = __ballot(any) != 0ull;
    if (fix_row && t < T - 1)
      for (int j = lane; j < K; j += 64) nx[j] = src[(t + 1) * K + j] + ll[(t + 1) * K + j];
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < K; i0 += 64) {
      const int i = i0 + lane;
      const bool vi = i < K;
      const int ic = vi ? i : 0;
      double v = vi ? src[t * K + i] : 0.0;
      const bool need = fix_row && vi && bh[t * K + i] < 1e-200;
      if (need) {                                   // (rows of ltran: a lane walks its own row)
        if (t == T - 1) v = 0.0;
        else {
          const double* __restrict__ lr = ltran + (size_t)ic * K;
          double m = -INFINITY;
          for (int j = 0; j < K; ++j) m = fmax(m, nx[j] + lr[j]);
          double sm = 0.0;
          for (int j = 0; j < K; ++j) sm += exp(nx[j] + lr[j] - m);
          v = m > -INFINITY ? m + log(sm) : -INFINITY;
        }
      }
      if (vi) dst[t * K + i] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int KMAX>
__global__ __launch_bounds__(64) void k_ffbs_paths(
    const double* __restrict__ la, const double* __restrict__ logA, const double* __restrict__ unif,
    int64_t T, int K, int Ls, unsigned char* __restrict__ path) {
  extern __shared__ double fsm[];
  double* lAT = fsm;                       // [K][KMAX + 1]: lAT[s][k] = logA[k][s]
  double* row = lAT + K * (KMAX + 1);      // [2][KMAX] the lalpha row of the step
  const int lane = threadIdx.x;
  for (int e = lane; e < K * K; e += 64) {
    const int k = e / K, sidx = e - k * K;
    lAT[sidx * (KMAX + 1) + k] = logA[e];
  }
  const int64_t lo = (int64_t)blockIdx.x * Ls;
  const int64_t hi = lo + Ls < T ? lo + Ls : T;
  int cur = lane < K ? lane : 0;           // lane s: the state at row t+1 on the path entered at s
  bool coupled = false;
  __syncthreads();
  for (int64_t t = hi - 1; t >= lo; --t) {
    const bool first = t == T - 1;         // no transition term: every entry state draws alike
    const double r = unif[t];
    double* rw = row + (t & 1) * KMAX;
    if (lane < KMAX) rw[lane] = lane < K ? la[t * K + lane] : -INFINITY;
    if (!coupled) {
      const double* tr = lAT + cur * (KMAX + 1);
      double lp[KMAX], m = -INFINITY;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        lp[k] = rw[k] + ((first || k >= K) ? 0.0 : tr[k]);
        m = fmax(m, lp[k]);
      }
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { lp[k] = k < K ? exp(lp[k] - m) : 0.0; tot += lp[k]; }
      const double thr = r * tot;
      double c = 0.0;
      int zz = K - 1;
      bool found = false;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        c += lp[k];
        const bool hit = !found && k < K && thr <= c;
        zz = hit ? k : zz;
        found = found || hit;
      }
      cur = zz;
      const int c0 = __builtin_amdgcn_readfirstlane(cur);
      coupled = __ballot(lane < K && cur != c0) == 0ull;
    } else {
      // cooperative draw, lane = state k
      double lp = -INFINITY;
      if (lane < K) lp = rw[lane] + (first ? 0.0 : lAT[cur * (KMAX + 1) + lane]);
      const double m = wave_max(lp);
      const double pk = lane < K ? exp(lp - m) : 0.0;
      const double tot = wave_sum(pk);
      double c = pk;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double v = __shfl_up(c, o, 64);
        if (lane >= o) c += v;
      }
      const unsigned long long bal = __ballot(lane < K && r * tot <= c);
      cur = bal ? (__ffsll((long long)bal) - 1) : (K - 1);
    }
    if (lane < KMAX) path[t * KMAX + lane] = (unsigned char)cur;
  }
}

// P1 for wide models (64 < K <= 256; states still fit the one-byte path entries): one workgroup of
// 256 threads per chunk, thread = entry state.  Until the paths have coupled every thread draws for
// its own current state (three passes over the K source states: maximum, total, inverse CDF -- the
// transition column comes from L2); afterwards one cooperative draw per row, thread = state, with
// workgroup-wide maximum / total / inclusive scan through LDS.  Same arithmetic per draw as
// k_ffbs_paths (u * sum_k p_k <= cumsum_k p_k, terms in state order).
__global__ __launch_bounds__(256) void k_ffbs_paths_wide(
    const double* __restrict__ la, const double* __restrict__ logA, const double* __restrict__ unif,
    int64_t T, int K, int Ls, unsigned char* __restrict__ path) {
  __shared__ double rw[256];          // the lalpha row of the step
  __shared__ double red[8];           // per-wave partials of the cooperative draw
  __shared__ int flag[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t lo = (int64_t)blockIdx.x * Ls;
  const int64_t hi = lo + Ls < T ? lo + Ls : T;
  const bool vs = tid < K;
  int cur = vs ? tid : 0;
  bool coupled = false;
  for (int64_t t = hi - 1; t >= lo; --t) {
    const bool first = t == T - 1;
    const double r = unif[t];
    __syncthreads();                                  // (previous row's readers are done)
    rw[tid] = vs ? la[t * K + tid] : -INFINITY;
    if (tid == 0) { flag[0] = 0; }
    __syncthreads();
    if (!coupled) {
      const double* __restrict__ col = logA + cur;    // logA[k][cur] = col[k * K]
      double m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmax(m, rw[k] + (first ? 0.0 : col[(size_t)k * K]));
      double tot = 0.0;
      for (int k = 0; k < K; ++k) tot += exp(rw[k] + (first ? 0.0 : col[(size_t)k * K]) - m);
      const double thr = r * tot;
      double c = 0.0;
      int zz = K - 1;
      for (int k = 0; k < K; ++k) {
        c += exp(rw[k] + (first ? 0.0 : col[(size_t)k * K]) - m);
        if (thr <= c) { zz = k; break; }
      }
      cur = zz;
      if (tid == 0) flag[1] = cur;
      __syncthreads();
      if (vs && cur != flag[1]) flag[0] = 1;          // benign race: any writer writes 1
      __syncthreads();
      coupled = flag[0] == 0;
    } else {
      // cooperative draw, thread = state k (all threads hold the same `cur`)
      const double lp = vs ? rw[tid] + (first ? 0.0 : logA[(size_t)tid * K + cur]) : -INFINITY;
      double wm = wave_max(lp);
      if (lane == 0) red[wave] = wm;
      __syncthreads();
      const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      const double pk = vs ? exp(lp - m) : 0.0;
      double c = pk;                                  // inclusive scan inside the wave, state order
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double v = __shfl_up(c, o, 64);
        if (lane >= o) c += v;
      }
      if (lane == 63) red[4 + wave] = c;
      __syncthreads();
      double before = 0.0;
      for (int w2 = 0; w2 < wave; ++w2) before += red[4 + w2];
      const double tot = ((red[4] + red[5]) + red[6]) + red[7];
      c += before;
      const unsigned long long bal = __ballot(vs && r * tot <= c);
      if (lane == 0) red[wave] = bal ? (double)(wave * 64 + __ffsll((long long)bal) - 1) : 1e9;
      __syncthreads();
      const double f = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
      cur = f < 1e8 ? (int)f : K - 1;
    }
    path[t * 256 + tid] = (unsigned char)cur;
  }
}

// P2: maps[c][s] = state at the first row of chunk c given entry state s (= path row lo_c);
// suffix composition H_c = G_c o G_{c+1} o ... ; entry[c] = H_{c+1}(0) (last chunk: 0, unused).
__global__ __launch_bounds__(1024) void k_ffbs_compose(const unsigned char* __restrict__ path, int64_t T,
                                                       int KS, int Ls, int C, unsigned char* __restrict__ mA,
                                                       unsigned char* __restrict__ mB,
                                                       unsigned char* __restrict__ entry) {
  const int n = C * KS;
  for (int e = threadIdx.x; e < n; e += 1024) {
    const int c = e / KS, x = e - c * KS;
    mA[e] = path[(int64_t)c * Ls * KS + x];
  }
  __threadfence_block();
  __syncthreads();
  unsigned char* src = mA;
  unsigned char* dst = mB;
  for (int d = 1; d < C; d <<= 1) {
    for (int e = threadIdx.x; e < n; e += 1024) {
      const int c = e / KS, x = e - c * KS;
      dst[e] = (c + d < C) ? src[c * KS + src[(c + d) * KS + x]] : src[e];
    }
    __threadfence_block();
    __syncthreads();
    unsigned char* t = src; src = dst; dst = t;
  }
  for (int c = threadIdx.x; c < C; c += 1024) entry[c] = (c + 1 < C) ? src[(c + 1) * KS] : 0;
}

__global__ __launch_bounds__(256) void k_ffbs_gather(const unsigned char* __restrict__ path,
                                                     const unsigned char* __restrict__ entry, int64_t T,
                                                     int KS, int Ls, int64_t* __restrict__ z) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  z[t] = path[t * KS + entry[t / Ls]];
}

__global__ void k_reduce_lb(const double* __restrict__ lse_part, int B, int nseg,
                            double* __restrict__ local_lb, double* __restrict__ lb_total) {
  // single block; deterministic order
  __shared__ double red[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    double s = 0.0;
    for (int i = 0; i < nseg; ++i) s += lse_part[(size_t)b * nseg + i];
    local_lb[b] = s;
    acc += s;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && lb_total) *lb_total = red[0];
}
