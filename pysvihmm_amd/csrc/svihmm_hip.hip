// svihmm_hip.hip -- MI355X (gfx950 / CDNA4) SVI-HMM E-step engine: kernels + C ABI.
//
// Hot path of dillonalaird/pysvihmm (see include/svihmm.h for the reference
// file:line each entry point replaces).  Everything is fp64 (the reference is
// float64 throughout, hmmbase.py:102-103).
//
// Data layout in HBM (all row-major float64):
//   obs   [T][D]                  resident for the life of the handle
//   theta [Fp][Kp]                emission parameters in "augmented feature" form:
//                                 feature f=(a,b), 0<=a<=b<=D over x~=(x_0..x_{D-1},1):
//                                 phi_f(x) = x~_a * x~_b ;  ll[t,k] = sum_f phi_f(x_t) theta[f,k]
//   ll / la / lb / q [B*Lm][K]    per-window intermediates (window-major == API layout)
//   part  [nchunk][Ftot][Kp]      per-workgroup partial statistics, Ftot = Fp + Kp
//                                 rows [0,Fp): sum_t phi_f(x_t) q[t,k]  (S, xbar, neff)
//                                 rows [Fp,Fp+Kp): sum_t q[t-1,i] q[t,k] (transition stat)
//   packed [K*K + K*D + K + K*D*D + 1]
//
// Recursions: log-domain storage, linear-domain mat-vec.  With m = max_i la[t-1,i],
//   la[t,j] = m + log( sum_i exp(la[t-1,i]-m) * exp(ltran[i,j]) ) + ll[t,j]
// is algebraically the reference's LSE_i(la[t-1,i] + ltran[i,j]) + ll[t,j]
// (hmmbase.py:295) with K exps + K logs per step instead of K^2.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/svihmm.h"

// ------------------------------------------------------------------------------------
//  error handling
// ------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess)                                                          \
      return fail(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ \
                  ":" + std::to_string(__LINE__) + ")");                           \
  } while (0)
#define NCCLCK(x)                                                                  \
  do {                                                                             \
    ncclResult_t r_ = (x);                                                         \
    if (r_ != ncclSuccess)                                                         \
      return fail(std::string(#x) + ": " + ncclGetErrorString(r_));                \
  } while (0)
#define CK(x)              \
  do {                     \
    if (int r__ = (x)) return r__; \
  } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }
// global row g=(b,t) of the flattened window batch -> obs row
__device__ __forceinline__ int64_t obs_row(const int64_t* __restrict__ starts, int Lm,
                                           int64_t g) {
  int64_t b = g / Lm;
  return starts[b] + (g - b * Lm);
}

// ------------------------------------------------------------------------------------
//  K1a: emission, VALU outer-product form (generic fallback).  lane = row.
//       grid (ceil(n/128), Kp/16), block 128, LDS (D+1)*129*8 bytes.
// ------------------------------------------------------------------------------------
#define EM_R 128
__global__ __launch_bounds__(EM_R) void k_emission_outer(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kp,
    const double* __restrict__ theta, uint32_t flags, double* __restrict__ ll) {
  extern __shared__ double xs[];  // [(D+1)][EM_R+1], transposed
  const int S = EM_R + 1;
  const int tid = threadIdx.x;
  const int64_t g0 = (int64_t)blockIdx.x * EM_R;
  const int k0 = blockIdx.y * 16;
  for (int e = tid; e < EM_R * D; e += EM_R) {
    int r = e / D, i = e - r * D;
    int64_t g = g0 + r;
    double v = 0.0;
    if (g < nrows) v = obs[obs_row(starts, Lm, g) * D + i];
    xs[i * S + r] = v;
  }
  xs[D * S + tid] = 1.0;
  __syncthreads();
  const int64_t g = g0 + tid;
  bool bad = false;
  if (g < nrows && (flags & SVIHMM_MASK_AS_NAN) && mask)
    bad = mask[obs_row(starts, Lm, g)] != 0;
  double acc[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc[kk] = 0.0;
  const double* th = theta + k0;
  int f = 0;
  for (int a = 0; a <= D; ++a) {
    const double xa = xs[a * S + tid];
    bad |= (xa != xa);
    for (int b = a; b <= D; ++b) {
      const double phi = xa * xs[b * S + tid];
      const double* row = th + (size_t)f * Kp;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc[kk] = fma(phi, row[kk], acc[kk]);
      ++f;
    }
  }
  if (g < nrows) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      if (k0 + kk < K) ll[g * K + k0 + kk] = bad ? 0.0 : nan_to_num(acc[kk]);
  }
}

// ------------------------------------------------------------------------------------
//  K1b: emission as an fp64 MFMA GEMM  ll[rows x K] = Phi[rows x Fp] * theta[Fp x Kp]
//       with Phi generated on the fly from x rows staged in LDS.
//       v_mfma_f64_16x16x4_f64: A lane l -> A[i=l&15][k=l>>4]; B lane l -> B[k=l>>4][j=l&15];
//       C/D lane l reg r -> C[row=(l>>4)+4r][col=l&15].
//       Workgroup = 4 waves x (MT=2 row tiles) = 128 rows; NT n-tiles of 16 states.
//       grid (ceil(n/128), Kp/(16*NT)), block 256.
// ------------------------------------------------------------------------------------
#define EMM_ROWS 128
template <int NT>
__global__ __launch_bounds__(256) void k_emission_mfma(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kp,
    int Fp, const double* __restrict__ theta, const int* __restrict__ fab,
    uint32_t flags, double* __restrict__ ll) {
  extern __shared__ double smem[];
  const int DS = (D + 2) | 1;  // odd row stride (doubles); slot D = 1.0, slot D+1 = 0.0
  double* xs = smem;                              // [EMM_ROWS][DS]
  int* fabs_ = (int*)(xs + EMM_ROWS * DS);        // [Fp] packed (a | b<<16)
  unsigned char* bad_s = (unsigned char*)(fabs_ + Fp);  // [EMM_ROWS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * EMM_ROWS;
  const int n0 = blockIdx.y * (16 * NT);

  if (tid < EMM_ROWS) {
    int64_t g = g0 + tid;
    unsigned char bd = 0;
    if (g < nrows && (flags & SVIHMM_MASK_AS_NAN) && mask)
      bd = mask[obs_row(starts, Lm, g)] != 0;
    bad_s[tid] = bd;
    xs[tid * DS + D] = 1.0;
    xs[tid * DS + D + 1] = 0.0;
  }
  for (int e = tid; e < Fp; e += 256) fabs_[e] = fab[e];
  __syncthreads();
  for (int e = tid; e < EMM_ROWS * D; e += 256) {
    int r = e / D, i = e - r * D;
    int64_t g = g0 + r;
    double v = 0.0;
    if (g < nrows) v = obs[obs_row(starts, Lm, g) * D + i];
    if (v != v) { bad_s[r] = 1; v = 0.0; }
    xs[r * DS + i] = v;
  }
  __syncthreads();

  const int li = lane & 15, lg = lane >> 4;
  const int r0 = wave * 32 + li;  // row of m-tile 0 for this lane; m-tile 1 = +16
  double4_t acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const double* xr0 = xs + r0 * DS;
  const double* xr1 = xs + (r0 + 16) * DS;
  const double* thl = theta + n0 + li;
  const int nks = Fp >> 2;
#pragma unroll 2
  for (int s = 0; s < nks; ++s) {
    const int f = (s << 2) + lg;
    const int ab = fabs_[f];
    const int a = ab & 0xffff, b = ab >> 16;
    const double A0 = xr0[a] * xr0[b];
    const double A1 = xr1[a] * xr1[b];
    const double* trow = thl + (size_t)f * Kp;
    double Bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) Bv[n] = trow[n * 16];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, Bv[n], acc[0][n], 0, 0, 0);
      acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, Bv[n], acc[1][n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = wave * 32 + m * 16 + lg + 4 * r;
      const int64_t g = g0 + rl;
      if (g < nrows) {
        const bool bd = bad_s[rl] != 0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int k = n0 + n * 16 + li;
          if (k < K) ll[g * K + k] = bd ? 0.0 : nan_to_num(acc[m][n][r]);
        }
      }
    }
  }
}

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
      const double m = wave_max(la);
      const double p = valid ? exp(la - m) : 0.0;
      if (j < KMAX) p_s[cur][j] = p;
      __syncthreads();
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < KMAX; i += 2) {
        s0 = fma(p_s[cur][i], a[i], s0);
        s1 = fma(p_s[cur][i + 1], a[i + 1], s1);
      }
      cur ^= 1;
      la = valid ? log(s0 + s1) + m + llt : NEG_INF;
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
      const double m = wave_max(u);
      const double p = valid ? exp(u - m) : 0.0;
      if (j < KMAX) p_s[cur][j] = p;
      __syncthreads();
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < KMAX; i += 2) {
        s0 = fma(p_s[cur][i], a[i], s0);
        s1 = fma(p_s[cur][i + 1], a[i + 1], s1);
      }
      cur ^= 1;
      lb = log(s0 + s1) + m;
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
//  K3: posterior marginals q = softmax_k(la+lb) and per-row LSE_k(la) partial sums.
//      grid (B*nseg), block 256 = 4 waves, one wave per row; seg = PS_ROWS rows.
// ------------------------------------------------------------------------------------
#define PS_ROWS 256
template <int KPL>  // states per lane (K <= 64*KPL)
__global__ __launch_bounds__(256) void k_posterior(const double* __restrict__ la,
                                                   const double* __restrict__ lb, int Lm,
                                                   int K, int nseg,
                                                   double* __restrict__ q,
                                                   double* __restrict__ lse_part) {
  __shared__ double wsum[4];
  const int b = blockIdx.x / nseg, seg = blockIdx.x - b * nseg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = seg * PS_ROWS;
  const int t1 = min(Lm, t0 + PS_ROWS);
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
//  K4a: statistics, VALU outer-product form (generic fallback).  One wave per
//       (row chunk, 16-feature chunk, 64-state chunk); lane = state.
//       feature f < Fp : phi = x~_a x~_b (0 on masked rows);  f >= Fp : phi = q[prev][f-Fp]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_stats_outer(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kp,
    int Fp, int F, const int* __restrict__ fab, const double* __restrict__ q,
    int64_t rows_per_chunk, uint32_t flags, int Lq, int off, double* __restrict__ part) {
  // rows g enumerate (window b, inner step t<Lm); q row = b*Lq+off+t, obs row = starts[b]+off+t
  const int lane = threadIdx.x;
  const int f0 = blockIdx.y * 16;
  const int k = blockIdx.z * 64 + lane;
  const int Ftot = Fp + Kp;
  const int64_t g0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t g1 = imin64(nrows, g0 + rows_per_chunk);
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  const bool is_trans = f0 >= Fp;
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t bwin = g / Lm;
    const int64_t t = g - bwin * Lm;
    const int64_t qrow = bwin * Lq + off + t;
    const double qk = (k < K) ? q[qrow * K + k] : 0.0;
    if (!is_trans) {
      const int64_t orow = starts[bwin] + off + t;
      if (mask && mask[orow]) continue;
      const double* x = obs + orow * D;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int f = f0 + i;
        double phi = 0.0;
        if (f < F) {
          const int ab = fab[f];
          const int a = ab & 0xffff, b = ab >> 16;
          const double xa = (a < D) ? x[a] : 1.0;
          const double xb = (b < D) ? x[b] : 1.0;
          phi = xa * xb;
        }
        acc[i] = fma(phi, qk, acc[i]);
      }
    } else {
      int64_t gp;
      if (t > 0) gp = qrow - 1;
      else if (flags & SVIHMM_TRANS_WRAP) gp = qrow + Lm - 1;
      else continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ii = f0 - Fp + i;
        const double phi = (ii < K) ? q[gp * K + ii] : 0.0;
        acc[i] = fma(phi, qk, acc[i]);
      }
    }
  }
  if (k < Kp) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      part[((size_t)blockIdx.x * Ftot + f0 + i) * Kp + k] = acc[i];
  }
}

// ------------------------------------------------------------------------------------
//  K4b: statistics as an fp64 MFMA GEMM  out[Ftot x Kp] = Phi^T[Ftot x rows] * q[rows x Kp]
//       Per workgroup: 4 waves, each MT m-tiles (16 features) x NT n-tiles (16 states);
//       rows of the chunk staged through LDS in blocks of ST_RB.
//       grid (nchunk, ceil(Ftot/16 / (4*MT)), Kp/(16*NT)), block 256.
// ------------------------------------------------------------------------------------
#define ST_RB 32
template <int MT, int NT>
__global__ __launch_bounds__(256) void k_stats_mfma(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kp,
    int Fp, int F, const int* __restrict__ fab, const double* __restrict__ q,
    int64_t rows_per_chunk, uint32_t flags, int Lq, int off, double* __restrict__ part) {
  extern __shared__ double smem[];
  const int DS = (D + 2) | 1;
  const int QS = 16 * NT + 1;  // padded q row stride
  double* xs = smem;                  // [ST_RB][DS]   augmented, masked rows zeroed
  double* qs = xs + ST_RB * DS;       // [ST_RB][QS]   q[t][n0..]
  double* qp = qs + ST_RB * QS;       // [ST_RB][Kp+1] q[prev(t)][all states] (transition tiles)
  const int QPS = Kp + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int Ftot = Fp + Kp;
  const int mt0 = (blockIdx.y * 4 + wave) * MT;  // first m-tile of this wave
  const int n0 = blockIdx.z * 16 * NT;
  const int wg_m0 = blockIdx.y * 4 * MT * 16, wg_m1 = wg_m0 + 4 * MT * 16;
  const bool need_x = wg_m0 < Fp;
  const bool need_qp = wg_m1 > Fp;

  // per-lane feature descriptors for each m-tile (constant for the whole kernel)
  int fa[MT], fb[MT], ftype[MT];  // ftype 0: emission feature, 1: transition
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int f = (mt0 + m) * 16 + li;
    if (f < F) {
      const int ab = fab[f];
      fa[m] = ab & 0xffff; fb[m] = ab >> 16; ftype[m] = 0;
    } else if (f >= Fp && f < Fp + K) {
      fa[m] = f - Fp; fb[m] = 0; ftype[m] = 1;
    } else if (f >= Fp) {
      fa[m] = Kp; fb[m] = 0; ftype[m] = 1;   // qp[r][Kp] is a zero column
    } else {
      fa[m] = D + 1; fb[m] = D + 1; ftype[m] = 0;  // zero slot
    }
  }
  double4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int64_t c0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t c1 = imin64(nrows, c0 + rows_per_chunk);
  for (int64_t s0 = c0; s0 < c1; s0 += ST_RB) {
    __syncthreads();
    // ---- stage ST_RB rows
    if (need_x) {
      for (int e = tid; e < ST_RB * (D + 2); e += 256) {
        const int r = e / (D + 2), i = e - r * (D + 2);
        const int64_t g = s0 + r;
        double v = 0.0;
        if (g < c1) {
          const int64_t bw = g / Lm;
          const int64_t orow = starts[bw] + off + (g - bw * Lm);
          const bool msk = mask && mask[orow];
          if (!msk) v = (i < D) ? obs[orow * D + i] : (i == D ? 1.0 : 0.0);
        }
        xs[r * DS + i] = v;
      }
    }
    for (int e = tid; e < ST_RB * 16 * NT; e += 256) {
      const int r = e / (16 * NT), c = e - r * (16 * NT);
      const int64_t g = s0 + r;
      const int k = n0 + c;
      double v = 0.0;
      if (g < c1 && k < K) {
        const int64_t bw = g / Lm;
        v = q[(bw * Lq + off + (g - bw * Lm)) * K + k];
      }
      qs[r * QS + c] = v;
    }
    if (need_qp) {
      for (int e = tid; e < ST_RB * (Kp + 1); e += 256) {
        const int r = e / (Kp + 1), c = e - r * (Kp + 1);
        const int64_t g = s0 + r;
        double v = 0.0;
        if (g < c1 && c < K) {
          const int64_t bwin = g / Lm;
          const int64_t t = g - bwin * Lm;
          const int64_t qrow = bwin * Lq + off + t;
          if (t > 0) v = q[(qrow - 1) * K + c];
          else if (flags & SVIHMM_TRANS_WRAP) v = q[(qrow + Lm - 1) * K + c];
        }
        qp[r * QPS + c] = v;
      }
    }
    __syncthreads();
    // ---- ST_RB/4 k-steps of 4 rows
#pragma unroll 2
    for (int ks = 0; ks < ST_RB / 4; ++ks) {
      const int r = ks * 4 + lg;
      double Bv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) Bv[n] = qs[r * QS + n * 16 + li];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        double A;
        if ((mt0 + m) * 16 < Fp) A = xs[r * DS + fa[m]] * xs[r * DS + fb[m]];  // wave-uniform
        else A = qp[r * QPS + fa[m]];
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, Bv[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // ---- write partials: C[row=(l>>4)+4r -> feature][col=l&15 -> state]
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = (mt0 + m) * 16 + lg + 4 * r;
      if (f < Ftot) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int k = n0 + n * 16 + li;
          part[((size_t)blockIdx.x * Ftot + f) * Kp + k] = acc[m][n][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
//  K5: deterministic reduction of the per-chunk partials + scatter into the packed layout
//      packed = [A_raw K*K | xbar K*D | neff K | S K*D*D | lb]
// ------------------------------------------------------------------------------------
__global__ void k_finalize(const double* __restrict__ part, int nchunk, int D, int K,
                           int Kp, int Fp, int F, const int* __restrict__ fab,
                           double* __restrict__ packed) {
  const int Ftot = Fp + Kp;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Ftot * Kp) return;
  const int f = idx / Kp, k = idx - (int64_t)f * Kp;
  if (k >= K) return;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += part[((size_t)c * Ftot + f) * Kp + k];
  double* A = packed;
  double* xbar = A + (size_t)K * K;
  double* neff = xbar + (size_t)K * D;
  double* S = neff + K;
  if (f < F) {
    const int ab = fab[f];
    const int a = ab & 0xffff, b = ab >> 16;
    if (b < D) {  // a <= b < D
      S[((size_t)k * D + a) * D + b] = s;
      S[((size_t)k * D + b) * D + a] = s;
    } else if (a < D) {
      xbar[(size_t)k * D + a] = s;
    } else {
      neff[k] = s;
    }
  } else if (f >= Fp && f - Fp < K) {
    A[(size_t)(f - Fp) * K + k] = s;
  }
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

// small utility kernels
__global__ void k_exp_transpose(const double* __restrict__ ltran, int K, double* __restrict__ A,
                                double* __restrict__ AT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * K) return;
  const int i = idx / K, j = idx - i * K;
  const double v = exp(ltran[idx]);
  A[idx] = v;
  AT[(size_t)j * K + i] = v;
}

__global__ void k_selftest_mfma(const double* __restrict__ A, const double* __restrict__ Bm,
                                double* __restrict__ C) {
  // A[16][4], B[4][16] row-major -> C[16][16]
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = Bm[(l >> 4) * 16 + (l & 15)];
  double4_t c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

// fp64 throughput micro-benchmarks (peak calibration for the roofline)
__global__ __launch_bounds__(256) void k_peak_mfma_f64(double* out, int iters) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void k_peak_fma_f64(double* out, int iters) {
  double c[8];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = fma(c[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------
//  host side
// ------------------------------------------------------------------------------------
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};
static int ensure(Buf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return 0;
  if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + bytes / 8 + 256;
  HIPCK(hipMalloc(&b.p, want));
  b.cap = want;
  return 0;
}
static void release(Buf& b) {
  if (b.p) hipFree(b.p);
  b.p = nullptr; b.cap = 0;
}

enum { KS_EMISSION = 0, KS_FB, KS_POSTERIOR, KS_STATS, KS_FINALIZE, KS_FFBS, KS_MISC,
       KS_ALLREDUCE, KS_H2D, KS_D2H, KS_RES0, KS_RES1 };
static const char* kKernNames[SVIHMM_NKERN] = {
    "emission", "forward_backward", "posterior", "stats", "finalize", "ffbs_sample",
    "misc", "allreduce", "h2d", "d2h", "reserved0", "reserved1"};

struct Pending { int slot; hipEvent_t e0, e1; };

struct svihmm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // data
  int64_t T = 0; int D = 0; bool have_mask = false;
  Buf obs, mask;
  // globals
  int K = 0;
  Buf mod_init, ltran, Aexp, AexpT;
  bool have_globals = false;
  // emission
  int eK = 0, eD = 0, Kp = 0, F = 0, Fp = 0;
  Buf theta, fab;
  bool have_emission = false;
  // work
  Buf starts, ll, la, lb, q, lse_part, local_lb, part, packed, scratch;
  int lastB = 0, lastLm = 0;       // shape of the intermediates currently held
  int hostB = 0, hostLm = 0;       // shape of host-uploaded lliks
  bool have_host_ll = false;
  bool have_packed = false;
  // variants: [0] emission (0 auto,1 outer,2 mfma) [1] stats (0 auto,1 outer,2 mfma)
  int variant[4] = {0, 0, 0, 0};
  // profiling
  bool prof = false;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  double ms[SVIHMM_NKERN] = {0};
  int64_t cnt[SVIHMM_NKERN] = {0};
  // comm
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
};

struct ProfScope {
  svihmm_ctx* h; int slot; hipEvent_t e0 = nullptr, e1 = nullptr; bool on;
  ProfScope(svihmm_ctx* h_, int slot_) : h(h_), slot(slot_), on(h_->prof) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!h->pool.empty()) { e = h->pool.back(); h->pool.pop_back(); }
      else hipEventCreate(&e);
      return e;
    };
    e0 = get(); e1 = get();
    hipEventRecord(e0, h->stream);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(e1, h->stream);
    h->pending.push_back({slot, e0, e1});
  }
};

static int set_device(svihmm_ctx* h) {
  HIPCK(hipSetDevice(h->device));
  return 0;
}

extern "C" {

const char* svihmm_last_error(void) { return g_err.c_str(); }
int svihmm_abi_version(void) { return SVIHMM_ABI_VERSION; }
const char* svihmm_kernel_name(int32_t slot) {
  return (slot >= 0 && slot < SVIHMM_NKERN) ? kKernNames[slot] : "";
}

int svihmm_device_count(int* n_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *n_out = 0; return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *n_out = n;
  return 0;
}

int svihmm_create(int device_id, svihmm_ctx** out) {
  if (!out) return fail("svihmm_create: out is NULL");
  int n = 0;
  HIPCK(hipGetDeviceCount(&n));
  if (n <= 0) return fail("svihmm_create: no HIP device visible (the HIP path has no CPU fallback)");
  if (device_id < 0 || device_id >= n) return fail("svihmm_create: bad device id");
  svihmm_ctx* h = new svihmm_ctx();
  h->device = device_id;
  HIPCK(hipSetDevice(device_id));
  HIPCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  *out = h;
  return 0;
}

int svihmm_destroy(svihmm_ctx* h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  for (auto& p : h->pending) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
  for (auto e : h->pool) hipEventDestroy(e);
  Buf* bufs[] = {&h->obs, &h->mask, &h->mod_init, &h->ltran, &h->Aexp, &h->AexpT, &h->theta,
                 &h->fab, &h->starts, &h->ll, &h->la, &h->lb, &h->q, &h->lse_part,
                 &h->local_lb, &h->part, &h->packed, &h->scratch};
  for (Buf* b : bufs) release(*b);
  hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int svihmm_sync(svihmm_ctx* h) {
  CK(set_device(h));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_set_obs(svihmm_ctx* h, const double* obs, int64_t T, int32_t D,
                   const uint8_t* mask) {
  if (!h || !obs || T <= 0 || D <= 0) return fail("svihmm_set_obs: bad arguments");
  if (D > 4095) return fail("svihmm_set_obs: D too large");
  CK(set_device(h));
  ProfScope ps(h, KS_H2D);
  CK(ensure(h->obs, (size_t)T * D * sizeof(double)));
  HIPCK(hipMemcpyAsync(h->obs.p, obs, (size_t)T * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->have_mask = mask != nullptr;
  if (mask) {
    CK(ensure(h->mask, (size_t)T));
    HIPCK(hipMemcpyAsync(h->mask.p, mask, (size_t)T, hipMemcpyHostToDevice, h->stream));
  }
  HIPCK(hipStreamSynchronize(h->stream));
  h->T = T; h->D = D;
  return 0;
}

int svihmm_set_globals(svihmm_ctx* h, int32_t K, const double* mod_init, const double* ltran) {
  if (!h || K <= 0 || !mod_init || !ltran) return fail("svihmm_set_globals: bad arguments");
  if (K > 1024) return fail("svihmm_set_globals: K > 1024 unsupported");
  CK(set_device(h));
  const size_t kk = (size_t)K * K * sizeof(double);
  CK(ensure(h->mod_init, K * sizeof(double)));
  CK(ensure(h->ltran, kk));
  CK(ensure(h->Aexp, kk));
  CK(ensure(h->AexpT, kk));
  HIPCK(hipMemcpyAsync(h->mod_init.p, mod_init, K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->ltran.p, ltran, kk, hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, KS_MISC);
    hipLaunchKernelGGL(k_exp_transpose, dim3((K * K + 255) / 256), dim3(256), 0, h->stream,
                       (const double*)h->ltran.p, K, (double*)h->Aexp.p, (double*)h->AexpT.p);
  }
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(h->stream));
  h->K = K; h->have_globals = true;
  return 0;
}

// ---- host NIW -> theta ------------------------------------------------------------
static double digamma_h(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  // asymptotic series: ln x - 1/2x - sum B_2n / (2n x^2n)
  const double t = f * (-1.0 / 12 + f * (1.0 / 120 + f * (-1.0 / 252 + f * (1.0 / 240 +
                   f * (-1.0 / 132 + f * (691.0 / 32760 + f * (-1.0 / 12)))))));
  return r + std::log(x) - 0.5 / x + t;
}

static int cholesky_lower(std::vector<double>& a, int n) {
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (!(d > 0.0)) return 1;
    d = std::sqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[i * n + j];
      for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s / d;
    }
    for (int i = 0; i < j; ++i) a[i * n + j] = 0.0;
  }
  return 0;
}

static inline int feat_index(int a, int b, int D) {  // 0 <= a <= b <= D
  return a * (D + 1) - a * (a - 1) / 2 + (b - a);
}

int svihmm_set_emission_niw(svihmm_ctx* h, int32_t K, int32_t D, const double* mu,
                            const double* sigma, const double* kappa, const double* nu) {
  if (!h || K <= 0 || D <= 0 || !mu || !sigma || !kappa || !nu)
    return fail("svihmm_set_emission_niw: bad arguments");
  CK(set_device(h));
  const int F = (D + 1) * (D + 2) / 2;
  const int Fp = (F + 15) / 16 * 16;
  const int Kp = (K + 15) / 16 * 16;
  std::vector<double> theta((size_t)Fp * Kp, 0.0);
  std::vector<int> fab(Fp, (D << 16) | D);
  for (int a = 0; a <= D; ++a)
    for (int b = a; b <= D; ++b) fab[feat_index(a, b, D)] = a | (b << 16);
  for (int f = F; f < Fp; ++f) fab[f] = (D + 1) | ((D + 1) << 16);  // padding -> zero slot
  std::vector<double> L((size_t)D * D), Li((size_t)D * D), W((size_t)D * D);
  const double LOG2PI = 1.8378770664093454835606594728112;
  for (int k = 0; k < K; ++k) {
    const double* S = sigma + (size_t)k * D * D;
    const double* m = mu + (size_t)k * D;
    for (int i = 0; i < D * D; ++i) L[i] = S[i];
    if (cholesky_lower(L, D))
      return fail("svihmm_set_emission_niw: sigma_mf[" + std::to_string(k) + "] is not positive definite");
    // Li = L^-1 (lower)
    std::fill(Li.begin(), Li.end(), 0.0);
    for (int c = 0; c < D; ++c) {
      Li[c * D + c] = 1.0 / L[c * D + c];
      for (int r = c + 1; r < D; ++r) {
        double s = 0.0;
        for (int j = c; j < r; ++j) s -= L[r * D + j] * Li[j * D + c];
        Li[r * D + c] = s / L[r * D + r];
      }
    }
    // W = (nu/2) * Li^T Li
    const double hn = 0.5 * nu[k];
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j) {
        double s = 0.0;
        for (int r = j; r < D; ++r) s += Li[r * D + i] * Li[r * D + j];
        W[i * D + j] = W[j * D + i] = hn * s;
      }
    double logdet = 0.0;
    for (int i = 0; i < D; ++i) logdet += std::log(L[i * D + i]);
    double llt = D * std::log(2.0) - 2.0 * logdet;
    for (int i = 0; i < D; ++i) llt += digamma_h(0.5 * (nu[k] - i));
    const double cst = 0.5 * llt - D / (2.0 * kappa[k]) - 0.5 * D * LOG2PI;
    double mWm = 0.0;
    for (int i = 0; i < D; ++i) {
      double wm = 0.0;
      for (int j = 0; j < D; ++j) wm += W[i * D + j] * m[j];
      theta[(size_t)feat_index(i, D, D) * Kp + k] = 2.0 * wm;  // linear term v_i
      mWm += m[i] * wm;
    }
    theta[(size_t)feat_index(D, D, D) * Kp + k] = cst - mWm;  // constant
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j)
        theta[(size_t)feat_index(i, j, D) * Kp + k] = (i == j) ? -W[i * D + i] : -2.0 * W[i * D + j];
  }
  CK(ensure(h->theta, theta.size() * sizeof(double)));
  CK(ensure(h->fab, fab.size() * sizeof(int)));
  HIPCK(hipMemcpyAsync(h->theta.p, theta.data(), theta.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->fab.p, fab.data(), fab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->eK = K; h->eD = D; h->Kp = Kp; h->F = F; h->Fp = Fp; h->have_emission = true;
  return 0;
}

// feature table is also needed by the statistics kernels when only host lliks are used
static int ensure_feature_table(svihmm_ctx* h) {
  if (h->have_emission && h->eD == h->D && h->eK == h->K) return 0;
  const int D = h->D, K = h->K;
  const int F = (D + 1) * (D + 2) / 2, Fp = (F + 15) / 16 * 16, Kp = (K + 15) / 16 * 16;
  std::vector<int> fab(Fp, (D + 1) | ((D + 1) << 16));
  for (int a = 0; a <= D; ++a)
    for (int b = a; b <= D; ++b) fab[feat_index(a, b, D)] = a | (b << 16);
  CK(ensure(h->fab, fab.size() * sizeof(int)));
  HIPCK(hipMemcpyAsync(h->fab.p, fab.data(), fab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->F = F; h->Fp = Fp; h->Kp = Kp;
  return 0;
}

int svihmm_set_lliks(svihmm_ctx* h, const double* lliks, int32_t B, int32_t Lm) {
  if (!h || !lliks || B <= 0 || Lm <= 0) return fail("svihmm_set_lliks: bad arguments");
  if (!h->have_globals) return fail("svihmm_set_lliks: call svihmm_set_globals first (K unknown)");
  CK(set_device(h));
  const size_t n = (size_t)B * Lm * h->K * sizeof(double);
  CK(ensure(h->ll, n));
  HIPCK(hipMemcpyAsync(h->ll.p, lliks, n, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->hostB = B; h->hostLm = Lm; h->have_host_ll = true;
  return 0;
}

// ---- launch helpers -----------------------------------------------------------------
static int check_windows(svihmm_ctx* h, const int64_t* starts, int B, int Lm, bool need_obs) {
  if (B <= 0 || Lm <= 0) return fail("bad window batch (B, Lm must be positive)");
  if (need_obs) {
    if (h->T <= 0) return fail("no observations: call svihmm_set_obs first");
    if (!starts) return fail("starts is NULL");
    for (int b = 0; b < B; ++b)
      if (starts[b] < 0 || starts[b] + Lm > h->T)
        return fail("window " + std::to_string(b) + " out of range");
  }
  return 0;
}

static int upload_starts(svihmm_ctx* h, const int64_t* starts, int B) {
  CK(ensure(h->starts, (size_t)B * sizeof(int64_t)));
  HIPCK(hipMemcpyAsync(h->starts.p, starts, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  return 0;
}

static int launch_emission(svihmm_ctx* h, int B, int Lm, uint32_t flags) {
  if (!h->have_emission) return fail("no emission parameters: call svihmm_set_emission_niw");
  if (h->eD != h->D) return fail("emission D does not match obs D");
  if (!h->have_globals || h->eK != h->K) return fail("emission K does not match globals K");
  const int64_t n = (int64_t)B * Lm;
  const int D = h->D, K = h->K, Kp = h->Kp;
  CK(ensure(h->ll, (size_t)n * K * sizeof(double)));
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  ProfScope ps(h, KS_EMISSION);
  int var = h->variant[0];
  if (var == 0) var = 2;
  if (var == 2) {
    const int DS = (D + 2) | 1;
    const size_t lds = (size_t)EMM_ROWS * DS * 8 + (size_t)h->Fp * 4 + EMM_ROWS;
    if (lds > 160 * 1024) var = 1;
    else {
      const int ntile = Kp / 16;
      const int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
      dim3 grid((unsigned)((n + EMM_ROWS - 1) / EMM_ROWS), ntile / NT);
#define EMM_LAUNCH(NTV)                                                                     \
  hipLaunchKernelGGL(k_emission_mfma<NTV>, grid, dim3(256), lds, h->stream,                 \
                     (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K, \
                     Kp, h->Fp, (const double*)h->theta.p, (const int*)h->fab.p, flags,     \
                     (double*)h->ll.p)
      if (lds > 64 * 1024) {
        hipFuncSetAttribute((const void*)k_emission_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)k_emission_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)k_emission_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
      if (NT == 4) EMM_LAUNCH(4); else if (NT == 2) EMM_LAUNCH(2); else EMM_LAUNCH(1);
#undef EMM_LAUNCH
    }
  }
  if (var == 1) {
    const size_t lds = (size_t)(D + 1) * (EM_R + 1) * 8;
    if (lds > 160 * 1024) return fail("emission: D too large for the LDS-staged kernels");
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_emission_outer, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)((n + EM_R - 1) / EM_R), Kp / 16);
    hipLaunchKernelGGL(k_emission_outer, grid, dim3(EM_R), lds, h->stream,
                       (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K,
                       Kp, (const double*)h->theta.p, flags, (double*)h->ll.p);
  }
  HIPCK(hipGetLastError());
  return 0;
}

static int launch_fb(svihmm_ctx* h, int B, int Lm, int dir0, int ndir) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const int K = h->K;
  const size_t n = (size_t)B * Lm * K * sizeof(double);
  if (dir0 == 0) CK(ensure(h->la, n));
  if (dir0 + ndir > 1) CK(ensure(h->lb, n));
  ProfScope ps(h, KS_FB);
  dim3 grid(B, ndir);
  const double* ll = (const double*)h->ll.p;
  const double* A = (const double*)h->Aexp.p;
  const double* mi = (const double*)h->mod_init.p;
  double* la = (double*)h->la.p;
  double* lb = (double*)h->lb.p;
  if (K <= 16)
    hipLaunchKernelGGL(k_fb_wave<16>, grid, dim3(64), 0, h->stream, ll, A, mi, Lm, K, dir0, la, lb);
  else if (K <= 32)
    hipLaunchKernelGGL(k_fb_wave<32>, grid, dim3(64), 0, h->stream, ll, A, mi, Lm, K, dir0, la, lb);
  else if (K <= 64)
    hipLaunchKernelGGL(k_fb_wave<64>, grid, dim3(64), 0, h->stream, ll, A, mi, Lm, K, dir0, la, lb);
  else {
    const int threads = (K + 63) / 64 * 64;
    const int in_lds = ((size_t)K * K * 8 + 2 * K * 8 + 128) <= 150 * 1024;
    const size_t lds = (2 * (size_t)K + 16) * 8 + (in_lds ? (size_t)K * K * 8 : 0);
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_fb_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_fb_generic, grid, dim3(threads), lds, h->stream, ll, A,
                       (const double*)h->AexpT.p, mi, Lm, K, dir0, in_lds, la, lb);
  }
  HIPCK(hipGetLastError());
  return 0;
}

static int launch_posterior(svihmm_ctx* h, int B, int Lm, bool total) {
  const int K = h->K;
  const int nseg = (Lm + PS_ROWS - 1) / PS_ROWS;
  CK(ensure(h->q, (size_t)B * Lm * K * sizeof(double)));
  CK(ensure(h->lse_part, (size_t)B * nseg * sizeof(double)));
  CK(ensure(h->local_lb, (size_t)B * sizeof(double)));
  CK(ensure(h->packed, (size_t)svihmm_packed_size(K, h->D > 0 ? h->D : 1) * sizeof(double)));
  ProfScope ps(h, KS_POSTERIOR);
  dim3 grid((unsigned)((size_t)B * nseg));
#define POST_LAUNCH(KPL)                                                                  \
  hipLaunchKernelGGL(k_posterior<KPL>, grid, dim3(256), 0, h->stream, (const double*)h->la.p, \
                     (const double*)h->lb.p, Lm, K, nseg, (double*)h->q.p, (double*)h->lse_part.p)
  if (K <= 64) POST_LAUNCH(1);
  else if (K <= 256) POST_LAUNCH(4);
  else POST_LAUNCH(16);
#undef POST_LAUNCH
  double* lbtot = nullptr;
  if (total) lbtot = (double*)h->packed.p + (svihmm_packed_size(K, h->D) - 1);
  hipLaunchKernelGGL(k_reduce_lb, dim3(1), dim3(256), 0, h->stream, (const double*)h->lse_part.p,
                     B, nseg, (double*)h->local_lb.p, lbtot);
  HIPCK(hipGetLastError());
  return 0;
}

static int launch_stats(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  // statistics over the inner segment [off, off+Lm) of each window of length Lq
  CK(ensure_feature_table(h));
  const int D = h->D, K = h->K, Kp = h->Kp, Fp = h->Fp, F = h->F;
  const int Ftot = Fp + Kp;
  const int64_t n = (int64_t)B * Lm;
  // row chunking: ~256 row chunks (x feature/state tiles => >= 1024 workgroups at D=32)
  // so that small minibatches still spread over the 256 CUs; chunk = multiple of ST_RB
  int64_t rpc = (n + 255) / 256;
  rpc = (rpc + ST_RB - 1) / ST_RB * ST_RB;
  int64_t nchunk = (n + rpc - 1) / rpc;
  CK(ensure(h->part, (size_t)nchunk * Ftot * Kp * sizeof(double)));
  CK(ensure(h->packed, (size_t)svihmm_packed_size(K, D) * sizeof(double)));
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  int var = h->variant[1];
  if (var == 0) var = 2;
  {
    ProfScope ps(h, KS_STATS);
    if (var == 2) {
      const int ntile = Kp / 16;
      const int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
      const int MT = 3;
      const int DS = (D + 2) | 1;
      const size_t lds = ((size_t)ST_RB * DS + (size_t)ST_RB * (16 * NT + 1) + (size_t)ST_RB * (Kp + 1)) * 8;
      if (lds > 160 * 1024) var = 1;
      else {
        const int mtiles = Ftot / 16;
        dim3 grid((unsigned)nchunk, (mtiles + 4 * MT - 1) / (4 * MT), ntile / NT);
#define ST_LAUNCH(NTV)                                                                        \
  do {                                                                                        \
    if (lds > 64 * 1024)                                                                      \
      hipFuncSetAttribute((const void*)k_stats_mfma<3, NTV>,                                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL((k_stats_mfma<3, NTV>), grid, dim3(256), lds, h->stream,               \
                       (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K, \
                       Kp, Fp, F, (const int*)h->fab.p, (const double*)h->q.p, rpc, flags,    \
                       Lq, off, (double*)h->part.p);                                          \
  } while (0)
        if (NT == 4) ST_LAUNCH(4); else if (NT == 2) ST_LAUNCH(2); else ST_LAUNCH(1);
#undef ST_LAUNCH
      }
    }
    if (var == 1) {
      dim3 grid((unsigned)nchunk, Ftot / 16, (Kp + 63) / 64);
      hipLaunchKernelGGL(k_stats_outer, grid, dim3(64), 0, h->stream, (const double*)h->obs.p, mk,
                         (const int64_t*)h->starts.p, n, Lm, D, K, Kp, Fp, F,
                         (const int*)h->fab.p, (const double*)h->q.p, rpc, flags, Lq, off,
                         (double*)h->part.p);
    }
    HIPCK(hipGetLastError());
  }
  {
    ProfScope ps(h, KS_FINALIZE);
    const int64_t tot = (int64_t)Ftot * Kp;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (const double*)h->part.p, (int)nchunk, D, K, Kp, Fp, F,
                       (const int*)h->fab.p, (double*)h->packed.p);
    HIPCK(hipGetLastError());
  }
  return 0;
}

static int d2h(svihmm_ctx* h, void* dst, const void* src, size_t bytes) {
  ProfScope ps(h, KS_D2H);
  HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  return 0;
}

int64_t svihmm_packed_size(int32_t K, int32_t D) {
  return (int64_t)K * K + (int64_t)K * D + K + (int64_t)K * D * D + 1;
}

static int prepare_ll(svihmm_ctx* h, const int64_t* starts, int B, int Lm, uint32_t flags,
                      bool need_obs_for_stats) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const bool host_ll = flags & SVIHMM_USE_HOST_LLIKS;
  CK(check_windows(h, starts, B, Lm, !host_ll || need_obs_for_stats));
  if (starts) CK(upload_starts(h, starts, B));
  if (host_ll) {
    if (!h->have_host_ll || h->hostB != B || h->hostLm != Lm)
      return fail("SVIHMM_USE_HOST_LLIKS: no uploaded lliks of shape [B,Lm,K]");
  } else {
    CK(launch_emission(h, B, Lm, flags));
    h->have_host_ll = false;
  }
  return 0;
}

int svihmm_loglik(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm, uint32_t flags,
                  double* out_lliks) {
  if (!h || !out_lliks) return fail("svihmm_loglik: bad arguments");
  CK(set_device(h));
  CK(prepare_ll(h, starts, B, Lm, flags & ~SVIHMM_USE_HOST_LLIKS, false));
  CK(d2h(h, out_lliks, h->ll.p, (size_t)B * Lm * h->K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  h->lastB = B; h->lastLm = Lm;
  return 0;
}

int svihmm_forward_backward(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                            uint32_t flags, double* out_lalpha, double* out_lbeta,
                            double* out_var_x, double* out_local_lb) {
  if (!h) return fail("svihmm_forward_backward: NULL handle");
  CK(set_device(h));
  CK(prepare_ll(h, starts, B, Lm, flags, false));
  CK(launch_fb(h, B, Lm, 0, 2));
  CK(launch_posterior(h, B, Lm, false));
  const size_t n = (size_t)B * Lm * h->K * sizeof(double);
  if (out_lalpha) CK(d2h(h, out_lalpha, h->la.p, n));
  if (out_lbeta) CK(d2h(h, out_lbeta, h->lb.p, n));
  if (out_var_x) CK(d2h(h, out_var_x, h->q.p, n));
  if (out_local_lb) CK(d2h(h, out_local_lb, h->local_lb.p, (size_t)B * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  h->lastB = B; h->lastLm = Lm;
  return 0;
}

int svihmm_estep_minibatch(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                           uint32_t flags, double* out_packed) {
  return svihmm_estep_minibatch_ex(h, starts, B, Lm, 0, Lm, flags, out_packed);
}

int svihmm_estep_minibatch_ex(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                              int32_t inner_off, int32_t inner_len, uint32_t flags,
                              double* out_packed) {
  if (!h) return fail("svihmm_estep_minibatch: NULL handle");
  CK(set_device(h));
  if (B == 0) {  // empty shard of a multi-GPU minibatch: all-zero statistics
    if (!h->have_globals || h->D <= 0) return fail("svihmm_estep_minibatch: set obs/globals first");
    const size_t nb = (size_t)svihmm_packed_size(h->K, h->D) * sizeof(double);
    CK(ensure(h->packed, nb));
    HIPCK(hipMemsetAsync(h->packed.p, 0, nb, h->stream));
    h->have_packed = true;
    if (out_packed) {
      CK(d2h(h, out_packed, h->packed.p, nb));
      HIPCK(hipStreamSynchronize(h->stream));
    }
    return 0;
  }
  if (inner_off < 0 || inner_len <= 0 || inner_off + inner_len > Lm)
    return fail("svihmm_estep_minibatch_ex: inner segment out of range");
  CK(prepare_ll(h, starts, B, Lm, flags, true));
  CK(launch_fb(h, B, Lm, 0, 2));
  CK(launch_posterior(h, B, Lm, true));
  CK(launch_stats(h, B, Lm, inner_off, inner_len, flags));
  h->have_packed = true;
  h->lastB = B; h->lastLm = Lm;
  if (out_packed) {
    CK(d2h(h, out_packed, h->packed.p, (size_t)svihmm_packed_size(h->K, h->D) * sizeof(double)));
    HIPCK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

int svihmm_read_packed(svihmm_ctx* h, double* out_packed) {
  if (!h || !out_packed) return fail("svihmm_read_packed: bad arguments");
  if (!h->have_packed) return fail("svihmm_read_packed: no statistics computed yet");
  CK(set_device(h));
  CK(d2h(h, out_packed, h->packed.p, (size_t)svihmm_packed_size(h->K, h->D) * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_read_intermediate(svihmm_ctx* h, int32_t what, double* out) {
  if (!h || !out) return fail("svihmm_read_intermediate: bad arguments");
  if (h->lastB <= 0) return fail("svihmm_read_intermediate: nothing computed yet");
  CK(set_device(h));
  Buf* src[] = {&h->ll, &h->la, &h->lb, &h->q};
  if (what < 0 || what > 3) return fail("svihmm_read_intermediate: bad selector");
  const size_t n = (size_t)h->lastB * h->lastLm * h->K * sizeof(double);
  if (!src[what]->p || src[what]->cap < n) return fail("svihmm_read_intermediate: buffer not available");
  CK(d2h(h, out, src[what]->p, n));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_read_rows(svihmm_ctx* h, int32_t what, int64_t row0, int64_t nrows, double* out) {
  if (!h || !out || row0 < 0 || nrows <= 0) return fail("svihmm_read_rows: bad arguments");
  if (h->lastB <= 0) return fail("svihmm_read_rows: nothing computed yet");
  if (what < 0 || what > 3) return fail("svihmm_read_rows: bad selector");
  if (row0 + nrows > (int64_t)h->lastB * h->lastLm) return fail("svihmm_read_rows: out of range");
  CK(set_device(h));
  Buf* src[] = {&h->ll, &h->la, &h->lb, &h->q};
  if (!src[what]->p) return fail("svihmm_read_rows: buffer not available");
  CK(d2h(h, out, (const double*)src[what]->p + (size_t)row0 * h->K, (size_t)nrows * h->K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_ffbs(svihmm_ctx* h, const double* logA, const double* uniforms, uint32_t flags,
                int64_t* out_z, double* out_lalpha) {
  if (!h || !logA || !uniforms || !out_z) return fail("svihmm_ffbs: bad arguments");
  CK(set_device(h));
  const int64_t T = h->T;
  if (T <= 0) return fail("svihmm_ffbs: no observations");
  if (T > 2147483647LL) return fail("svihmm_ffbs: T too large");
  int64_t st0 = 0;
  CK(prepare_ll(h, &st0, 1, (int)T, flags, false));
  CK(launch_fb(h, 1, (int)T, 0, 1));
  const int K = h->K;
  CK(ensure(h->scratch, ((size_t)K * K + (size_t)T) * sizeof(double) + (size_t)T * sizeof(int64_t)));
  double* dlogA = (double*)h->scratch.p;
  double* dun = dlogA + (size_t)K * K;
  int64_t* dz = (int64_t*)(dun + T);
  HIPCK(hipMemcpyAsync(dlogA, logA, (size_t)K * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(dun, uniforms, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, KS_FFBS);
    hipLaunchKernelGGL(k_ffbs_sample, dim3(1), dim3(64), K > 64 ? (size_t)K * 8 : 0, h->stream,
                       (const double*)h->la.p, (const double*)dlogA, (const double*)dun, T, K, dz);
    HIPCK(hipGetLastError());
  }
  CK(d2h(h, out_z, dz, (size_t)T * sizeof(int64_t)));
  if (out_lalpha) CK(d2h(h, out_lalpha, h->la.p, (size_t)T * K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  h->lastB = 1; h->lastLm = (int)T;
  return 0;
}

// ---- multi-GPU -------------------------------------------------------------------------
int svihmm_comm_unique_id(char uid_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLCK(ncclGetUniqueId(&id));
  std::memcpy(uid_out, &id, 128);
  return 0;
}

int svihmm_comm_init(svihmm_ctx* h, const char uid[128], int32_t rank, int32_t nranks) {
  if (!h || !uid || nranks <= 0 || rank < 0 || rank >= nranks)
    return fail("svihmm_comm_init: bad arguments");
  CK(set_device(h));
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  ncclUniqueId id;
  std::memcpy(&id, uid, 128);
  NCCLCK(ncclCommInitRank(&h->comm, nranks, id, rank));
  h->rank = rank; h->nranks = nranks;
  return 0;
}

int svihmm_comm_destroy(svihmm_ctx* h) {
  if (!h) return 0;
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  h->nranks = 1; h->rank = 0;
  return 0;
}

int svihmm_allreduce_packed(svihmm_ctx* h) {
  if (!h) return fail("svihmm_allreduce_packed: NULL handle");
  if (!h->have_packed) return fail("svihmm_allreduce_packed: no statistics computed yet");
  if (!h->comm) return fail("svihmm_allreduce_packed: communicator not initialised");
  CK(set_device(h));
  ProfScope ps(h, KS_ALLREDUCE);
  const size_t n = (size_t)svihmm_packed_size(h->K, h->D);
  NCCLCK(ncclAllReduce(h->packed.p, h->packed.p, n, ncclDouble, ncclSum, h->comm, h->stream));
  return 0;
}

int svihmm_allreduce_host(svihmm_ctx* h, double* buf, int64_t n, int32_t op) {
  if (!h || !buf || n <= 0) return fail("svihmm_allreduce_host: bad arguments");
  if (!h->comm) return fail("svihmm_allreduce_host: communicator not initialised");
  CK(set_device(h));
  Buf tmp;
  CK(ensure(tmp, (size_t)n * sizeof(double)));
  int rc = 0;
  do {
    if (hipMemcpyAsync(tmp.p, buf, n * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = fail("allreduce_host: H2D failed"); break; }
    ncclResult_t r = ncclAllReduce(tmp.p, tmp.p, (size_t)n, ncclDouble, op == 1 ? ncclMax : ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) { rc = fail(std::string("ncclAllReduce: ") + ncclGetErrorString(r)); break; }
    if (hipMemcpyAsync(buf, tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rc = fail("allreduce_host: D2H failed"); break; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail("allreduce_host: sync failed"); break; }
  } while (0);
  release(tmp);
  return rc;
}

// ---- measurement --------------------------------------------------------------------------
static void drain(svihmm_ctx* h) {
  for (auto& p : h->pending) {
    hipEventSynchronize(p.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      h->ms[p.slot] += ms;
      h->cnt[p.slot] += 1;
    }
    h->pool.push_back(p.e0);
    h->pool.push_back(p.e1);
  }
  h->pending.clear();
}

int svihmm_profile_enable(svihmm_ctx* h, int32_t on) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  if (!on) drain(h);
  h->prof = on != 0;
  return 0;
}
int svihmm_profile_reset(svihmm_ctx* h) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  drain(h);
  for (int i = 0; i < SVIHMM_NKERN; ++i) { h->ms[i] = 0; h->cnt[i] = 0; }
  return 0;
}
int svihmm_profile_read(svihmm_ctx* h, double ms_out[SVIHMM_NKERN], int64_t count_out[SVIHMM_NKERN]) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  HIPCK(hipStreamSynchronize(h->stream));
  drain(h);
  for (int i = 0; i < SVIHMM_NKERN; ++i) { ms_out[i] = h->ms[i]; count_out[i] = h->cnt[i]; }
  return 0;
}
int svihmm_set_variant(svihmm_ctx* h, int32_t which, int32_t value) {
  if (!h || which < 0 || which >= 4) return fail("svihmm_set_variant: bad arguments");
  h->variant[which] = value;
  return 0;
}

// ---- diagnostics ---------------------------------------------------------------------------
int svihmm_selftest_mfma(svihmm_ctx* h, const double* A16x4, const double* B4x16, double* C16x16) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  CK(ensure(h->scratch, (64 + 64 + 256) * sizeof(double)));
  double* dA = (double*)h->scratch.p; double* dB = dA + 64; double* dC = dB + 64;
  HIPCK(hipMemcpyAsync(dA, A16x4, 64 * 8, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(dB, B4x16, 64 * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, h->stream, dA, dB, dC);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(C16x16, dC, 256 * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

// which 0: v_mfma_f64_16x16x4_f64, 1: v_fma_f64.  Returns achieved TFLOP/s.
int svihmm_peak_fp64(svihmm_ctx* h, int32_t which, double* tflops_out) {
  if (!h || !tflops_out) return fail("bad arguments");
  CK(set_device(h));
  const int blocks = 256 * 8, threads = 256, iters = 20000;
  CK(ensure(h->scratch, (size_t)blocks * threads * sizeof(double)));
  hipEvent_t e0, e1;
  HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCK(hipEventRecord(e0, h->stream));
    if (which == 0)
      hipLaunchKernelGGL(k_peak_mfma_f64, dim3(blocks), dim3(threads), 0, h->stream, (double*)h->scratch.p, iters);
    else
      hipLaunchKernelGGL(k_peak_fma_f64, dim3(blocks), dim3(threads), 0, h->stream, (double*)h->scratch.p, iters);
    HIPCK(hipEventRecord(e1, h->stream));
    HIPCK(hipEventSynchronize(e1));
  }
  float ms = 0.f;
  HIPCK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  double flops;
  if (which == 0) flops = (double)blocks * (threads / 64) * (double)iters * 4.0 * 2048.0;
  else flops = (double)blocks * threads * (double)iters * 8.0 * 2.0;
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  return 0;
}

}  // extern "C"
