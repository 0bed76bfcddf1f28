// kernels_svi.h -- the global side of one SVI iteration on the device (hmmsgd_metaobs.py:347-445):
// stationary initial vector + psi-expectations before the E-step, the natural-gradient global
// step and the global part of the ELBO after it.  With these the variational state (var_tran,
// the K NIW factors) lives in HBM for the whole of infer(): per iteration only the window starts
// and the learning rate go to the device and nothing has to come back.
// Part of libsvihmm_hip.so; compiled in svihmm_hip.hip.
#pragma once

#define SVI_EPS 1e-9     // the reference's eps (hmmbase.py:30) inside digamma / gammaln

// ------------------------------------------------------------------------------------
//  G1: globals of an iteration, K + 1 workgroups of 512 threads.
//    (a) psi-expectations of the transition factor (hmmsgd_metaobs.py:502-504):
//          ltran[i][j] = psi(var_tran[i][j] + eps) - psi(sum_j var_tran[i][j] + eps),
//        plus exp(ltran) and its transpose for the scaled sweeps (what k_exp_transpose makes
//        from host-supplied globals);
//    (b) stationary initial vector (hmmsgd_metaobs.py:413-418): the reference takes |v| of the
//        top eigenvector of the row-normalised var_tran^T from np.linalg.eig -- for a positive
//        stochastic matrix the Perron vector, unit L2 norm.  Here: Grassmann-Taksar-Heyman
//        elimination (state reduction without subtractions: every intermediate is a sum of
//        products of positive numbers, component-wise relative accuracy ~K eps, no pivoting),
//        K-1 dependent steps of a rank-one update on the shrinking leading block;
//    (c) mod_init[k] = psi(var_init[k] + eps) - psi(sum var_init + eps)  (quirk Q5: the unit-L2
//        vector goes into psi as if it were Dirichlet parameters).
//  `work` = 2 K (K|1) doubles (LDS when they fit, else global scratch).
// ------------------------------------------------------------------------------------
// GTH elimination for K <= 64 by ONE wavefront, matrix in registers: lane i holds row i (64
// doubles), the loops over states are unrolled completely so that every register index is a
// compile-time constant.  Step N (N = K-1 .. 1): s = sum_{j<N} P[N][j] (each lane sums its own row,
// lane N's sum is the one read), column N is scaled by 1/s and stays in place as Q[i][N], and
// P[i][j] += Q[i][N] P[N][j] with the pivot row's entries read by v_readlane -- no LDS, no barrier:
// ~4 N + 40 instructions per step instead of an LDS round trip + workgroup barrier (round 2:
// ~1.4 us per step, 90 us in all).  Then pi_0 = 1, pi_j = sum_{i<j} pi_i Q[i][j] as one wave
// reduction per state.  Sums of positive products only, as before.
template <int N>
struct GthStep {
  static __device__ __forceinline__ void run(double (&r)[64], int K) {
    if (N < K) {                                   // (uniform: states beyond K do not exist)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if ((j & 3) == 0) s0 += r[j]; else if ((j & 3) == 1) s1 += r[j]; else if ((j & 3) == 2) s2 += r[j]; else s3 += r[j];
      }
      double s = (s0 + s1) + (s2 + s3);
      pin_all_lanes(s);
      const double inv = 1.0 / readlane_f64(s, N);
      const double c = r[N] * inv;
      r[N] = c;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        double pj = r[j];
        pin_all_lanes(pj);
        r[j] = fma(c, readlane_f64(pj, N), r[j]);
      }
    }
    GthStep<N - 1>::run(r, K);
  }
};
template <>
struct GthStep<0> {
  static __device__ __forceinline__ void run(double (&)[64], int) {}
};
template <int J>
struct GthBack {
  static __device__ __forceinline__ void run(const double (&r)[64], int K, int lane, double& p) {
    GthBack<J - 1>::run(r, K, lane, p);
    if (J < K) {
      const double t = wave_sum_dpp(lane < J ? p * r[J] : 0.0);
      p = lane == J ? t : p;
    }
  }
};
template <>
struct GthBack<0> {
  static __device__ __forceinline__ void run(const double (&)[64], int, int, double&) {}
};

template <bool LDSW, int NWV>
__device__ __forceinline__ void k_svi_globals_body(
    const double* __restrict__ var_tran, int K, double* __restrict__ work_g,
    double* __restrict__ ltran, double* __restrict__ Aexp, double* __restrict__ AexpT,
    double* __restrict__ var_init, double* __restrict__ mod_init) {
  extern __shared__ double svi_lds[];
  __shared__ double rs[1024];      // row sums, later the stationary vector (K <= 1024)
  __shared__ double psum[2][8];    // partial sums of the next pivot row (double-buffered)
  __shared__ double sc[4];
  constexpr int NT = 64 * NWV, CPT = 64 / NWV;     // threads; register columns per thread (K <= 64 path)
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  if ((int)blockIdx.x < K) {
    // ---- (a) psi-expectations of transition row i = blockIdx.x: 4096 digamma / exp evaluations
    // are ~40k cycles of fp64 work on one CU, so they are spread over K workgroups that run
    // beside the elimination (the last workgroup)
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = tid; j < K; j += NT) s += var_tran[(size_t)i * K + j];
    s = wave_sum(s);
    if (lane == 0) rs[w] = s;
    __syncthreads();
    double rsum = 0.0;
    for (int u = 0; u < NWV; ++u) rsum += rs[u];
    const double dgs = digamma_d(rsum + SVI_EPS);
    for (int j = tid; j < K; j += NT) {
      const size_t e = (size_t)i * K + j;
      const double l = digamma_d(var_tran[e] + SVI_EPS) - dgs;
      const double x = exp(l);
      ltran[e] = l;
      Aexp[e] = x;
      AexpT[(size_t)j * K + i] = x;
    }
    return;
  }
  // ---- (b) stationary vector by GTH elimination, (c) mod_init: the last workgroup
  const int LD = K | 1;                            // odd row stride: a column walks all LDS banks
  // LDSW is a template parameter so that the LDS instantiation keeps ds_* instructions (a
  // run-time choice between the two spaces compiles to flat loads: 2x slower for this kernel)
  auto Pm = [&]() -> double* { if constexpr (LDSW) return svi_lds; else return work_g; };
  double* P = Pm();                                // [K][LD] working copy, then its reduced form
  double* Q = P + (size_t)K * LD;                  // [K][LD] scaled columns (back-substitution)
  for (int i = w; i < K; i += NWV) {               // row sums (one wave per row, round-robin)
    double s = 0.0;
    for (int j = lane; j < K; j += 64) s += var_tran[(size_t)i * K + j];
    s = wave_sum(s);
    if (lane == 0) rs[i] = s;
  }
  __syncthreads();
  for (int e0 = 0; e0 < K * K; e0 += 8 * NT) {     // row-stochastic mean transition matrix
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT + tid;
      v[u] = e < K * K ? var_tran[e] : 1.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT + tid;
      if (e < K * K) {
        const int i = e / K, j = e - i * K;
        P[(size_t)i * LD + j] = v[u] / rs[i];
      }
    }
  }
  __syncthreads();
  if (K <= 64) {
    // one wavefront, matrix in registers (GthStep / GthBack above); the other waves are done
    if (w == 0) {
      double r[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) r[j] = (lane < K && j < K) ? P[(size_t)lane * LD + j] : 0.0;
      GthStep<63>::run(r, K);
      double pst = lane == 0 ? 1.0 : 0.0;
      GthBack<63>::run(r, K, lane, pst);
      if (lane < K) rs[lane] = pst;
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  } else {
    // GTH: eliminate states K-1 .. 1.  Thread (row r = lane (+64, ..), column phase c = wave)
    // owns columns c, c+8, .. of its rows: a wave's accesses to 64 different rows are 65 doubles
    // apart (two lanes per LDS bank pair, the minimum for 8-byte words), the pivot row is a
    // broadcast.  One barrier per step (row n and column n are not written in step n); the sum
    // of the NEXT pivot row is gathered by its owners while they update it.
    const int c0 = w;
    {
      double s = 0.0;
      if (lane == ((K - 1) & 63))
        for (int j = c0; j < K - 1; j += NWV) s += P[(size_t)(K - 1) * LD + j];
      if (lane == ((K - 1) & 63)) psum[(K - 1) & 1][c0] = s;
    }
    __syncthreads();
    for (int n = K - 1; n >= 1; --n) {
      const double* __restrict__ rown = P + (size_t)n * LD;
      const double* __restrict__ ps = psum[n & 1];
      double tot = 0.0;
      for (int u = 0; u < NWV; ++u) tot += ps[u];
      const double inv = 1.0 / tot;
      for (int i = lane; i < n; i += 64) {
        double* __restrict__ rowi = P + (size_t)i * LD;
        const double c = rowi[n] * inv;
        double s = 0.0;
        for (int jb = c0; jb < n; jb += 4 * NWV) {        // four of this thread's columns per trip, loads first
          double rn[4], rv[4];
  #pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = jb + NWV * u, jc = j < n ? j : n;      // clamped to column n: read, never written
            rn[u] = rown[jc];
            rv[u] = rowi[jc];
          }
  #pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = jb + NWV * u;
            const double v = fma(c, rn[u], rv[u]);
            if (j < n) rowi[j] = v;
            s += (j < n - 1) ? v : 0.0;
          }
        }
        if (c0 == 0) Q[(size_t)i * LD + n] = c;
        if (i == n - 1) psum[(n - 1) & 1][c0] = s;    // the next pivot row's partial sums
      }
      __syncthreads();
    }
  }
  // back-substitution: pi_0 = 1, pi_j = sum_{i<j} pi_i Q[i][j]
  if (tid < 64) {
    if (K <= 64) {
      // (rs already holds the un-normalised stationary vector: the register path above)
    } else {
      if (lane == 0) rs[0] = 1.0;
      for (int j = 1; j < K; ++j) {
        double s = 0.0;
        for (int i = lane; i < j; i += 64) s = fma(rs[i], Q[(size_t)i * LD + j], s);
        s = wave_sum_dpp(s);
        if (lane == 0) rs[j] = s;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
      }
    }
    double n2 = 0.0, n1 = 0.0;
    for (int i = lane; i < K; i += 64) n2 = fma(rs[i], rs[i], n2);
    n2 = wave_sum(n2);
    const double rn = 1.0 / sqrt(n2);
    for (int i = lane; i < K; i += 64) { const double v = rs[i] * rn; rs[i] = v; n1 += v; }
    n1 = wave_sum(n1);
    if (lane == 0) sc[1] = n1;
  }
  __syncthreads();
  const double dsum = digamma_d(sc[1] + SVI_EPS);
  for (int k = tid; k < K; k += NT) {
    const double v = rs[k];
    var_init[k] = v;
    mod_init[k] = digamma_d(v + SVI_EPS) - dsum;
  }
}
template <bool LDSW, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_svi_globals(
    const double* __restrict__ var_tran, int K, double* __restrict__ work_g,
    double* __restrict__ ltran, double* __restrict__ Aexp, double* __restrict__ AexpT,
    double* __restrict__ var_init, double* __restrict__ mod_init, SviSync sy) {
  // (skipped -- the loop is dead, its gate kernel gave up -- the iteration these globals are for is poisoned; the
  //  kernel still arrives so that nothing waits for it)
  // (round 6: this kernel runs on a side stream BESIDE the emission GEMM of the main stream, whose waves share its
  //  SIMDs, and the minibatch sweeps gate on it: the elimination's 64 dependent steps ended 2.7 us after the emission
  //  kernel -- profiles/r06z_svi_iteration_trace.txt.  Its few waves take issue priority over their neighbours.)
  __builtin_amdgcn_s_setprio(3);
  if (svi_gate(sy)) k_svi_globals_body<LDSW, NWV>(var_tran, K, work_g, ltran, Aexp, AexpT, var_init, mod_init);
  else svi_poison(sy);
  svi_arrive(sy);
}

// ------------------------------------------------------------------------------------
//  G2: natural-gradient global step (hmmsgd_metaobs.py:1010-1069, util.py:28-60) from the packed
//  statistics [A_raw | xbar | neff | S | lb] of the minibatch (after the all-reduce, if any):
//    transitions: var_tran <- (1-rho)(var_tran - 1) + rho * bA * (A_raw + nwin (prior_tran - 1)) + 1
//                 (quirk Q2: prior_tran - 1 sits in every window's A_i); with ada_G (AdaGrad, :1036-1040)
//                 G += (var_tran - 1)^2, step 1 / G^(1/4) per entry instead of rho
//    emissions  : eta = [kappa mu, kappa, sigma + kappa mu mu', nu + 2 + D];
//                 eta <- (1-rho) eta + rho (eta_0 + bE * [xbar, neff, S, neff]);  back to moments.
//  grid: K workgroups (one NIW factor each) + ceil(K^2 / 256) for the transition factor.
//  niw = [mu K*D | sigma K*D*D | kappa K | nu K] (the E-step's parameter block, updated in place);
//  prior = [mu0 K*D | sigma0 K*D*D | kappa0 K | nu0 K].
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void k_svi_global_step_body(
    const double* __restrict__ packed, const double* __restrict__ prior_tran, double* __restrict__ var_tran,
    double* __restrict__ niw, const double* __restrict__ prior, int K, int D, double rho, double bA,
    double bE, double nwin, double* __restrict__ lb_keep, double* __restrict__ ada_G) {
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= K) {
    const int e = ((int)blockIdx.x - K) * 256 + tid;
    // the minibatch's local bound, kept for the ELBO kernel (which runs on a side stream while
    // the next E-step already rewrites `packed`)
    if (e == 0) *lb_keep = packed[(size_t)K * K + nmu + K + nsg];
    svi_tran_step(e, K, packed, prior_tran, var_tran, rho, bA, nwin, ada_G);
    return;
  }
  const int k = blockIdx.x;
  extern __shared__ double gs_lds[];       // mu_old [D] | mu_new [D] | mu_0 [D]
  double* mo = gs_lds;
  double* mn = mo + D;
  double* m0 = mn + D;
  double* mu = niw + (size_t)k * D;
  double* sg = niw + nmu + (size_t)k * D * D;
  double* kap = niw + nmu + nsg;
  double* nu = kap + K;
  const double* mu0 = prior + (size_t)k * D;
  const double* sg0 = prior + nmu + (size_t)k * D * D;
  const double ka0 = prior[nmu + nsg + k], nu0 = prior[nmu + nsg + K + k];
  const double* xbar = packed + (size_t)K * K + (size_t)k * D;
  const double neff = packed[(size_t)K * K + nmu + k];
  const double* S = packed + (size_t)K * K + nmu + K + (size_t)k * D * D;
  const double ka = kap[k], nuo = nu[k];
  const double e2 = (1.0 - rho) * ka + rho * (ka0 + bE * neff);                        // kappa'
  const double e4 = (1.0 - rho) * (nuo + 2 + D) + rho * ((nu0 + 2 + D) + bE * neff);
  for (int a = tid; a < D; a += 256) {
    const double m = mu[a], p = mu0[a];
    mo[a] = m; m0[a] = p;
    mn[a] = ((1.0 - rho) * (ka * m) + rho * (ka0 * p + bE * xbar[a])) / e2;            // mu' = e1 / e2
  }
  __syncthreads();
  for (int e = tid; e < D * D; e += 256) {
    const int a = e / D, b = e - a * D;
    const double e3o = sg[e] + (mo[a] * mo[b]) * ka;
    const double e3p = sg0[e] + (m0[a] * m0[b]) * ka0;
    const double e3 = (1.0 - rho) * e3o + rho * (e3p + bE * S[e]);
    sg[e] = e3 - (mn[a] * mn[b]) * e2;                                                 // sigma'
  }
  for (int a = tid; a < D; a += 256) mu[a] = mn[a];
  if (tid == 0) { kap[k] = e2; nu[k] = e4 - 2 - D; }
}
__global__ __launch_bounds__(256) void k_svi_global_step(
    const double* __restrict__ packed, const double* __restrict__ prior_tran, double* __restrict__ var_tran,
    double* __restrict__ niw, const double* __restrict__ prior, int K, int D, double rho, double bA,
    double bE, double nwin, double* __restrict__ lb_keep, double* __restrict__ ada_G, SviSync sy) {
  if (svi_step_gate(sy)) k_svi_global_step_body(packed, prior_tran, var_tran, niw, prior, K, D, rho, bA, bE, nwin, lb_keep, ada_G);
  svi_arrive(sy);
}

// ------------------------------------------------------------------------------------
//  G3: the NIW factors' term of global_lower_bound (hmmsgd_metaobs.py:294 sum_k get_vlb();
//  formulas of distributions.niw_vlb_batch: Bishop 10.74 + 10.77) for the CURRENT factors, from
//  what k_niw_to_theta_wave just produced: theta holds W = (nu/2) sigma_mf^-1 in feature form,
//  logdet = log det sigma_mf.  One wave per state -> vlb[k].
//  prior_logpart[k] = invwishart_log_partitionfunction(sigma_0[k], nu_0[k]) (host, constant);
//  zsign: +1 pybasicbayes' sign of that term, -1 Bishop's (see distributions.Gaussian.get_vlb).
// ------------------------------------------------------------------------------------
// Dirichlet energy + entropy of transition row i for the UPDATED var_tran (one wave)
__device__ __forceinline__ void svi_rowterm(int i, int K, int lane, const double* __restrict__ prior_tran,
                                            const double* __restrict__ var_tran, double* __restrict__ rowterm) {
  double sv = 0.0;
  for (int j = lane; j < K; j += 64) sv += var_tran[(size_t)i * K + j];
  sv = wave_sum(sv);
  const double dgs = digamma_d(sv + SVI_EPS);
  double acc = 0.0;
  for (int j = lane; j < K; j += 64) {
    const double q = var_tran[(size_t)i * K + j], p = prior_tran[(size_t)i * K + j];
    const double elog = digamma_d(q + SVI_EPS) - dgs;
    acc += ((p - 1.0) - (q - 1.0)) * elog + lgamma(q + SVI_EPS);
  }
  acc = wave_sum(acc);
  if (lane == 0) rowterm[i] = acc - lgamma(sv + SVI_EPS);
}
__device__ __forceinline__ void k_svi_vlb_body(
    const double* __restrict__ theta, const int* __restrict__ fab, int F, int D, int Kp,
    const double* __restrict__ niw, const double* __restrict__ logdet, const double* __restrict__ prior,
    const double* __restrict__ prior_logpart, double zsign, int K, double* __restrict__ vlb,
    const double* __restrict__ prior_tran, const double* __restrict__ var_tran,
    double* __restrict__ rowterm) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= K) {
    // Dirichlet energy + entropy of transition row i for the UPDATED var_tran (hmmbase.
    // dirichlet_elbo, reference hmmsgd_metaobs.py:277-292) minus the prior-only constants
    // (lgamma of the prior row: added once by the host-supplied prior_const in k_svi_elbo)
    svi_rowterm((int)blockIdx.x - K, K, lane, prior_tran, var_tran, rowterm);
    return;
  }
  const int k = blockIdx.x;
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  const double* m = niw + (size_t)k * D;
  const double ka = niw[nmu + nsg + k], nu = niw[nmu + nsg + K + k];
  const double* m0 = prior + (size_t)k * D;
  const double* s0 = prior + nmu + (size_t)k * D * D;
  const double ka0 = prior[nmu + nsg + k], nu0 = prior[nmu + nsg + K + k];
  double tr = 0.0, qd = 0.0;
  for (int f = lane; f < F; f += 64) {
    const int ab = fab[f], a = ab & 0xffff, b = ab >> 16;
    if (b >= D) continue;                    // linear and constant features
    const double t = theta[(size_t)f * Kp + k];
    tr = fma(t, s0[a * D + b], tr);
    qd = fma(t, (m[a] - m0[a]) * (m[b] - m0[b]), qd);
  }
  double dg = 0.0, lg = 0.0;
  for (int i = lane; i < D; i += 64) {
    dg += digamma_d(0.5 * (nu - i));
    lg += lgamma(0.5 * (nu - i));
  }
  tr = wave_sum(tr); qd = wave_sum(qd); dg = wave_sum(dg); lg = wave_sum(lg);
  if (lane == 0) {
    const double c = -2.0 / nu;
    const double tr_s0 = c * tr, quad = c * qd, half_ld = 0.5 * logdet[k];
    const double LN2 = 0.69314718055994530942, LNPI = 1.1447298858494001741, LN2PI = 1.8378770664093454836;
    const double l_mf = dg + D * LN2 - 2.0 * half_ld;
    const double logpart_mf = -(nu * half_ld - (nu * D / 2.0 * LN2 + D * (D - 1) / 4.0 * LNPI + lg));
    const double iw_entropy = logpart_mf - (nu - D - 1) / 2.0 * l_mf + nu * D / 2.0;
    const double q_entropy = -0.5 * (l_mf + D * ((log(ka) - LN2PI) - 1.0)) + iw_entropy;
    const double p_avgengy = 0.5 * (D * (log(ka0) - LN2PI) + l_mf - D * ka0 / ka - ka0 * nu * quad)
                             + zsign * prior_logpart[k] + (nu0 - D - 1) / 2.0 * l_mf - 0.5 * nu * tr_s0;
    vlb[k] = p_avgengy + q_entropy;
  }
}
// Round 6: in the counter choreography the ELBO total (k_svi_elbo below) rides in this launch -- the workgroup that
// arrives LAST on the side counter (its value then equals `at`) forms it.  Why: beside the fused sweep + statistics
// launch, which fills 240 CUs for ~110 us, the 2 K one-wave workgroups of this kernel are handed to the shader engines
// round-robin and those bound for an engine without a free CU wait for that launch to END; a second one-workgroup
// kernel behind them (12-15 us of launch + a dependent-load chain) then ended AFTER the main stream's finalize, and the
// next global step -- which may not overwrite the factors before these kernels have read them -- waited 2-3 us for it
// every iteration (profiles/r06z_svi_iteration_trace.txt).  Same sums in the same order as k_svi_elbo_body.
struct SviElboTail {
  double* out;                 // nullptr: no total in this launch
  const double* lb;
  double prior_const;
  unsigned at;                 // value of the arrival counter once every workgroup of the launch has arrived
};
__device__ __forceinline__ void svi_arrive_elbo(const SviSync& sy, bool go, const SviElboTail& et, int K,
                                                const double* __restrict__ vlb, const double* __restrict__ rowterm) {
  if (!sy.arrive) return;
  __shared__ int last_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = __hip_atomic_fetch_add(sy.arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (sy.stamp && before + 1u == sy.stamp_at) *sy.stamp = wall_clock64();
    last_s = (et.out != nullptr && go && before + 1u == et.at) ? 1 : 0;
  }
  __syncthreads();
  if (!last_s || threadIdx.x >= 64) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // the other workgroups' terms (released by their arrivals)
  const int lane = threadIdx.x;
  double v = 0.0, d = 0.0;
  for (int base = 0; base < K; base += 64) {               // k ascending, one add per term: k_svi_elbo_body's order
    const double x = base + lane < K ? __hip_atomic_load(vlb + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    const double y = base + lane < K ? __hip_atomic_load(rowterm + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    const int n = K - base < 64 ? K - base : 64;
    for (int j = 0; j < n; ++j) { v += __shfl(x, j, 64); d += __shfl(y, j, 64); }
  }
  if (lane == 0) *et.out = et.lb[0] + (d + et.prior_const) + v;
}
__global__ __launch_bounds__(64) void k_svi_vlb(
    const double* __restrict__ theta, const int* __restrict__ fab, int F, int D, int Kp,
    const double* __restrict__ niw, const double* __restrict__ logdet, const double* __restrict__ prior,
    const double* __restrict__ prior_logpart, double zsign, int K, double* __restrict__ vlb,
    const double* __restrict__ prior_tran, const double* __restrict__ var_tran,
    double* __restrict__ rowterm, SviSync sy, SviElboTail et) {
  const bool go = svi_gate(sy);
  if (go) k_svi_vlb_body(theta, fab, F, D, Kp, niw, logdet, prior, prior_logpart, zsign, K, vlb, prior_tran, var_tran, rowterm);
  svi_arrive_elbo(sy, go, et, K, vlb, rowterm);
}

// ------------------------------------------------------------------------------------
//  G4: elbo_vec[it] = lb + global_lower_bound()  (hmmsgd_metaobs.py:436-445, 273-296):
//  lb = packed[last] (sum of the windows' local bounds) + the transition rows' Dirichlet terms
//  (rowterm[i] from k_svi_vlb + prior_const = sum_i [lgamma(sum_j p_ij + eps) - sum_j lgamma(p_ij
//  + eps)], a constant of the prior computed once by the host) + sum_k vlb[k]; fixed order.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void k_svi_elbo_body(int K, const double* __restrict__ vlb,
                                                 const double* __restrict__ rowterm, double prior_const,
                                                 const double* __restrict__ lb, double* __restrict__ elbo_out) {
  if (threadIdx.x == 0) {
    double v = 0.0, d = 0.0;
    for (int k = 0; k < K; ++k) v += vlb[k];
    for (int i = 0; i < K; ++i) d += rowterm[i];
    *elbo_out = lb[0] + (d + prior_const) + v;
  }
}
__global__ __launch_bounds__(64) void k_svi_elbo(int K, const double* __restrict__ vlb,
                                                 const double* __restrict__ rowterm, double prior_const,
                                                 const double* __restrict__ lb, double* __restrict__ elbo_out, SviSync sy) {
  if (svi_gate(sy)) k_svi_elbo_body(K, vlb, rowterm, prior_const, lb, elbo_out);
  svi_arrive(sy);
}


// ------------------------------------------------------------------------------------
//  G2' / G3': the same two steps for the families whose factors are element-wise (round 4):
//    fam 1, diagonal Gaussian (distributions.DiagonalGaussian: per dimension a normal-inverse-gamma
//      factor; blocks [mu | nus | alphas | betas], each [K][W = D]); packed = [A_raw | xbar K*D | neff K |
//      xsq K*D | lb].  Blend in the natural parameters eta = [nu mu, nu, 2 beta + nu mu^2, 2 alpha]
//      (hmmsgd_metaobs.py:1050-1069 with this family's eta): eta <- (1-rho) eta + rho (eta_0 +
//      bE [xbar, neff, xsq, neff]); ELBO term: -KL(q || prior) summed over the dimensions
//      (distributions.DiagonalGaussian.get_vlb).
//    fam 2, Categorical (hmmsgd_metaobs.py:907-926, 1071-1084): Dirichlet factors alpha[K][W = V];
//      packed = [A_raw | counts K*V | lb]; every window contributes alpha_0 + counts - 1, so
//      alpha <- (1-rho)(alpha - 1) + rho bE (nwin (alpha_0 - 1) + counts) + 1; ELBO term as
//      distributions.Categorical.get_vlb.
//  grid: nem = ceil(K W / 256) element blocks + ceil(K^2 / 256) transition blocks.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void k_svi_global_step_simple_body(
    int fam, const double* __restrict__ packed, const double* __restrict__ prior_tran,
    double* __restrict__ var_tran, double* __restrict__ blk, const double* __restrict__ prior, int K, int W,
    double rho, double bA, double bE, double nwin, double* __restrict__ lb_keep, double* __restrict__ ada_G,
    int nem) {
  const int tid = threadIdx.x;
  const size_t n = (size_t)K * W;
  if ((int)blockIdx.x >= nem) {
    const int e = ((int)blockIdx.x - nem) * 256 + tid;
    if (e == 0) *lb_keep = packed[(size_t)K * K + (fam == 1 ? 2 * n + K : n)];
    svi_tran_step(e, K, packed, prior_tran, var_tran, rho, bA, nwin, ada_G);
    return;
  }
  const size_t e = (size_t)blockIdx.x * 256 + tid;
  if (e >= n) return;
  if (fam == 2) {
    const double counts = packed[(size_t)K * K + e];
    const double inter = nwin * (prior[e] - 1.0) + counts;
    blk[e] = ((1.0 - rho) * (blk[e] - 1.0) + (rho * bE) * inter) + 1.0;
    return;
  }
  const int k = (int)(e / W);
  const double xb = packed[(size_t)K * K + e], ne = packed[(size_t)K * K + n + k];
  const double xs = packed[(size_t)K * K + n + K + e];
  const double m = blk[e], nu = blk[n + e], al = blk[2 * n + e], be = blk[3 * n + e];
  const double m0 = prior[e], nu0 = prior[n + e], al0 = prior[2 * n + e], be0 = prior[3 * n + e];
  const double e0 = (1.0 - rho) * (nu * m) + rho * (nu0 * m0 + bE * xb);
  const double e1 = (1.0 - rho) * nu + rho * (nu0 + bE * ne);
  const double e2 = (1.0 - rho) * (2.0 * be + nu * m * m) + rho * ((2.0 * be0 + nu0 * m0 * m0) + bE * xs);
  const double e3 = (1.0 - rho) * (2.0 * al) + rho * (2.0 * al0 + bE * ne);
  const double mn = e0 / e1;
  blk[e] = mn; blk[n + e] = e1; blk[2 * n + e] = 0.5 * e3; blk[3 * n + e] = 0.5 * (e2 - e1 * mn * mn);
}
__global__ __launch_bounds__(256) void k_svi_global_step_simple(
    int fam, const double* __restrict__ packed, const double* __restrict__ prior_tran,
    double* __restrict__ var_tran, double* __restrict__ blk, const double* __restrict__ prior, int K, int W,
    double rho, double bA, double bE, double nwin, double* __restrict__ lb_keep, double* __restrict__ ada_G,
    int nem, SviSync sy) {
  if (svi_step_gate(sy)) k_svi_global_step_simple_body(fam, packed, prior_tran, var_tran, blk, prior, K, W, rho, bA, bE, nwin, lb_keep, ada_G, nem);
  svi_arrive(sy);
}

// grid 2 K waves: [0, K) the factors' ELBO terms vlb[k], [K, 2K) the transition rows' terms
__device__ __forceinline__ void k_svi_vlb_simple_body(
    int fam, const double* __restrict__ blk, const double* __restrict__ prior, int K, int W,
    double* __restrict__ vlb, const double* __restrict__ prior_tran, const double* __restrict__ var_tran,
    double* __restrict__ rowterm) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= K) {
    svi_rowterm((int)blockIdx.x - K, K, lane, prior_tran, var_tran, rowterm);
    return;
  }
  const int k = blockIdx.x;
  const size_t n = (size_t)K * W, o = (size_t)k * W;
  if (fam == 2) {
    double sa = 0.0, s0 = 0.0;
    for (int v = lane; v < W; v += 64) { sa += blk[o + v]; s0 += prior[o + v]; }
    sa = wave_sum(sa); s0 = wave_sum(s0);
    const double dgs = digamma_d(sa);
    double acc = 0.0;
    for (int v = lane; v < W; v += 64) {
      const double a = blk[o + v], a0 = prior[o + v];
      const double el = digamma_d(a) - dgs;
      acc += (a0 - a) * el - lgamma(a0) + lgamma(a);      // ((a0-1) - (a-1)) el - gammaln(a0) + gammaln(a)
    }
    acc = wave_sum(acc);
    if (lane == 0) vlb[k] = acc + lgamma(s0) - lgamma(sa);
    return;
  }
  const double LN2PI = 1.8378770664093454836;
  double acc = 0.0;
  for (int d = lane; d < W; d += 64) {
    const double m = blk[o + d], nu = blk[n + o + d], al = blk[2 * n + o + d], be = blk[3 * n + o + d];
    const double m0 = prior[o + d], nu0 = prior[n + o + d], al0 = prior[2 * n + o + d], be0 = prior[3 * n + o + d];
    const double elog = log(be) - digamma_d(al);          // E log sigma^2
    const double prec = al / be;                          // E 1 / sigma^2
    const double dm = m - m0;
    const double pp = 0.5 * (log(nu0) - LN2PI) - (al0 + 1.5) * elog - 0.5 * nu0 * (1.0 / nu + dm * dm * prec)
                      + al0 * log(be0) - lgamma(al0) - be0 * prec;
    const double qq = 0.5 * (log(nu) - LN2PI) - (al + 1.5) * elog - 0.5 + al * log(be) - lgamma(al) - al;
    acc += pp - qq;
  }
  acc = wave_sum(acc);
  if (lane == 0) vlb[k] = acc;
}
__global__ __launch_bounds__(64) void k_svi_vlb_simple(
    int fam, const double* __restrict__ blk, const double* __restrict__ prior, int K, int W,
    double* __restrict__ vlb, const double* __restrict__ prior_tran, const double* __restrict__ var_tran,
    double* __restrict__ rowterm, SviSync sy, SviElboTail et) {
  const bool go = svi_gate(sy);
  if (go) k_svi_vlb_simple_body(fam, blk, prior, K, W, vlb, prior_tran, var_tran, rowterm);
  svi_arrive_elbo(sy, go, et, K, vlb, rowterm);
}

// E log theta[v][k] = psi(alpha[k][v]) - psi(sum_v alpha[k][v]): the Categorical lookup table
// (layout [V][K], see k_emission_cat) from the resident Dirichlet factors; one wave per state
__device__ __forceinline__ void k_cat_table_body(const double* __restrict__ alpha, int K, int V,
                                                  double* __restrict__ table) {
  const int k = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int v = lane; v < V; v += 64) s += alpha[(size_t)k * V + v];
  s = wave_sum(s);
  const double dgs = digamma_d(s);
  for (int v = lane; v < V; v += 64) table[(size_t)v * K + k] = digamma_d(alpha[(size_t)k * V + v]) - dgs;
}
__global__ __launch_bounds__(64) void k_cat_table(const double* __restrict__ alpha, int K, int V,
                                                  double* __restrict__ table, SviSync sy) {
  if (svi_gate(sy)) k_cat_table_body(alpha, K, V, table);
  svi_arrive(sy);
}

// one wave that waits for a counter: what a side stream runs in front of a kernel whose inputs another stream
// produces (instead of hipStreamWaitEvent on an event recorded between two kernels of the main chain)
__global__ __launch_bounds__(64) void k_svi_gate(SviSync sy) { svi_gate(sy); }
// Can a kernel of one stream run while a kernel of another stream spins?  (svi_begin_common: a tool that lets one
// kernel at a time onto the device -- rocprofv3 --pmc, AMD_SERIALIZE_KERNEL -- may dispatch a gate before the
// kernel it waits for and then never lets that kernel in; the loop falls back to stream events there.)  The waiter
// reports that it runs, then waits up to `ticks` of the device wall clock for the setter's flag.
__global__ __launch_bounds__(64) void k_svi_probe_wait(const unsigned* flag, unsigned* started, unsigned* result,
                                                       unsigned long long ticks) {
  if (threadIdx.x != 0) return;
  __hip_atomic_store(started, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  unsigned seen = 0;
  while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < ticks)
    __builtin_amdgcn_s_sleep(8);
  __hip_atomic_store(result, seen ? 1u : 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void k_svi_probe_set(unsigned* flag) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// iteration-begin stamp when the loop does not run back to back (first iteration, after a host-side hook)
__global__ __launch_bounds__(64) void k_svi_stamp(unsigned long long* ts) { if (threadIdx.x == 0) *ts = wall_clock64(); }
