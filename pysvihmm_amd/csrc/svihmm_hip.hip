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
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/svihmm.h"
#include "svihmm_common.h"

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
template <int NT, int MT>
__global__ __launch_bounds__(256) void k_emission_mfma(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kp,
    int Fp, const double* __restrict__ theta, const int* __restrict__ fab,
    uint32_t flags, double* __restrict__ ll) {
  // workgroup = 4 waves x MT row tiles of 16 rows
  constexpr int ROWS = 64 * MT;
  extern __shared__ double smem[];
  const int DS = (D + 2) | 1;  // odd row stride (doubles); slot D = 1.0, slot D+1 = 0.0
  double* xs = smem;                              // [ROWS][DS]
  int* fabs_ = (int*)(xs + ROWS * DS);            // [Fp] packed (a | b<<16)
  unsigned char* bad_s = (unsigned char*)(fabs_ + Fp);  // [ROWS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * ROWS;
  const int n0 = blockIdx.y * (16 * NT);

  for (int r = tid; r < ROWS; r += 256) {
    int64_t g = g0 + r;
    unsigned char bd = 0;
    if (g < nrows && (flags & SVIHMM_MASK_AS_NAN) && mask)
      bd = mask[obs_row(starts, Lm, g)] != 0;
    bad_s[r] = bd;
    xs[r * DS + D] = 1.0;
    xs[r * DS + D + 1] = 0.0;
  }
  for (int e = tid; e < Fp; e += 256) fabs_[e] = fab[e];
  __syncthreads();
  for (int e = tid; e < ROWS * D; e += 256) {
    int r = e / D, i = e - r * D;
    int64_t g = g0 + r;
    double v = 0.0;
    if (g < nrows) v = obs[obs_row(starts, Lm, g) * D + i];
    if (v != v) { bad_s[r] = 1; v = 0.0; }
    xs[r * DS + i] = v;
  }
  __syncthreads();

  const int li = lane & 15, lg = lane >> 4;
  const int r0 = wave * 16 * MT + li;  // row of m-tile 0 for this lane; m-tile m = +16m
  double4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const double* xr = xs + r0 * DS;
  const double* thl = theta + n0 + li;
  // Fp is a multiple of 16 -> the k-step count is a multiple of 4: the loop is unrolled by
  // hand so that the theta (B operand) loads of four k-steps are in flight together
  const int nks = Fp >> 2;
  for (int s = 0; s < nks; s += 4) {
    double Bv[4][NT], Av[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = ((s + u) << 2) + lg;
      const double* trow = thl + (size_t)f * Kp;
#pragma unroll
      for (int n = 0; n < NT; ++n) Bv[u][n] = trow[n * 16];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = ((s + u) << 2) + lg;
      const int ab = fabs_[f];
      const int a = ab & 0xffff, b = ab >> 16;
#pragma unroll
      for (int m = 0; m < MT; ++m) Av[u][m] = xr[m * 16 * DS + a] * xr[m * 16 * DS + b];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(Av[u][m], Bv[u][n], acc[m][n], 0, 0, 0);
  }
  // epilogue on plain VGPR copies: keeps the loop-carried accumulators in AGPRs (otherwise
  // hipcc shuttles all of them VGPR<->AGPR around every k-step)
  double outv[MT][NT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) outv[m][n][r] = acc[m][n][r];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = wave * 16 * MT + m * 16 + lg + 4 * r;
      const int64_t g = g0 + rl;
      if (g < nrows) {
        const bool bd = bad_s[rl] != 0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int k = n0 + n * 16 + li;
          if (k < K) ll[g * K + k] = bd ? 0.0 : nan_to_num(outv[m][n][r]);
        }
      }
    }
  }
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
__device__ __forceinline__ double wave64_max_fast(double v) {
  v = row16_max(v);
  v = fmax_raw(v, __shfl_xor(v, 16, 64));
  v = fmax_raw(v, __shfl_xor(v, 32, 64));
  return v;
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
#define LN2_D 0.69314718055994530942

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
    int64_t rows_per_chunk, uint32_t flags, int Lq, int off, double* __restrict__ part,
    int mt_base) {
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
  const int mt0 = mt_base + (blockIdx.y * 4 + wave) * MT;  // first m-tile of this wave
  const int n0 = blockIdx.z * 16 * NT;
  const int wg_m0 = (mt_base + blockIdx.y * 4 * MT) * 16, wg_m1 = wg_m0 + 4 * MT * 16;
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
//  K4c: statistics GEMM, software-pipelined, VGPR-form accumulators (K <= 64).
//  Same math as K4b.  Design points:
//   * fp64 MFMA with AGPR accumulators runs at ~63 % of the VGPR-form rate on gfx950
//     (tools/peak_probe.py: 49 vs 77.6 TF/s), so the accumulators must fit the 256
//     architected VGPRs: a workgroup is 4 m-groups x NSPLIT n-groups of waves, each wave
//     MT x NTW tiles (5 x 2 x 8 = 80 accumulator registers at K = 64);
//   * the next 32-row stage is fetched from HBM into registers while the current stage
//     runs on the matrix pipe (global -> reg early, reg -> LDS after the compute);
//   * row bookkeeping (obs row, q row, wrap predecessor, mask) is computed once per stage
//     by 32 lanes instead of per element (no integer divisions in the copy loops);
//   * the 36 emission + 4 transition tiles of K=64, D=32 split into two balanced
//     workgroup passes, so q is read twice.
//  grid (nchunk, ceil(Ftot/16 / (4*MT))), block 256*NSPLIT.
// ------------------------------------------------------------------------------------
struct StRow {
  long long orow;   // obs row, -1: out of range or masked (x~ = 0)
  long long qrow;   // q row, -1: out of range
  long long prow;   // predecessor q row, -1: none
};

template <int MT, int NTW, int NSPLIT, int XK>
__global__ __launch_bounds__(256 * NSPLIT) void k_stats_mfma3(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Fp, int F,
    const int* __restrict__ fab, const double* __restrict__ q, int64_t rows_per_chunk,
    uint32_t flags, int Lq, int off, double* __restrict__ part, int KpTot, int mt_limit) {
  // KpTot: padded state count of the whole problem (partials stride); this workgroup covers
  // states [blockIdx.z*Kp, +Kp); only m-tiles < mt_limit are produced (K > 64: the
  // transition tiles are left to k_stats_mfma)
  constexpr int NT = NTW * NSPLIT;
  constexpr int Kp = 16 * NT;
  constexpr int QS = Kp + 1;
  constexpr int TPR = 8 * NSPLIT;          // staging threads per row
  constexpr int QK = (Kp + TPR - 1) / TPR;  // q columns per staging thread
  extern __shared__ double smem[];
  // One LDS row per time step holds every A-operand source, so that each operand is the
  // branch-free product row[fa] * row[fb]:
  //   [0, D)      x (0 on masked rows)        D        1.0 (0 on masked rows)
  //   D+1  ZERO   0.0                          D+2      ONE = 1.0 (always)
  //   QP0 + i     q[prev(t), i], i < Kp  (transition features: row[QP0+i] * row[ONE])
  const int ZERO = D + 1, ONE = D + 2, QP0 = D + 3;
  const int RS = (QP0 + Kp) | 1;   // odd stride
  double* rb0 = smem;                    // [2][32][RS]
  double* qs0 = rb0 + 2 * ST_RB * RS;    // [2][32][QS]
  StRow* rinfo = reinterpret_cast<StRow*>(qs0 + 2 * ST_RB * QS);  // [4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int mg = wave & 3, ng = wave >> 2;
  const int Ftot = Fp + KpTot;
  const int kbase = blockIdx.z * Kp;
  const int mt0 = (blockIdx.y * 4 + mg) * MT;
  const int nt0 = ng * NTW;
  const int wg_m0 = blockIdx.y * 4 * MT * 16, wg_m1 = wg_m0 + 4 * MT * 16;
  const bool need_x = wg_m0 < Fp;
  const bool need_qp = wg_m1 > Fp && mt_limit * 16 > Fp;
  const int sr = tid / TPR, sc = tid % TPR;   // staging role: row sr, columns sc + TPR*k

  int fa[MT], fb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int f = (mt0 + m) * 16 + li;
    fa[m] = ZERO; fb[m] = ZERO;
    if (f < F) { const int ab = fab[f]; fa[m] = ab & 0xffff; fb[m] = ab >> 16; }
    else if (f >= Fp && f - Fp < K && mt_limit * 16 > Fp) { fa[m] = QP0 + (f - Fp); fb[m] = ONE; }
  }
  double4_t acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int64_t c0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t c1 = imin64(nrows, c0 + rows_per_chunk);
  const int nstage = (int)((c1 - c0 + ST_RB - 1) / ST_RB);

  auto row_info = [&](int64_t s0, int buf) {
    if (tid < ST_RB) {
      const int64_t g = s0 + tid;
      StRow ri; ri.orow = -1; ri.qrow = -1; ri.prow = -1;
      if (g < c1) {
        const int64_t bw = g / Lm;
        const int64_t t = g - bw * Lm;
        ri.qrow = bw * Lq + off + t;
        const int64_t orow = starts[bw] + off + t;
        ri.orow = (mask && mask[orow]) ? -1 : orow;
        if (t > 0) ri.prow = ri.qrow - 1;
        else if (flags & SVIHMM_TRANS_WRAP) ri.prow = ri.qrow + Lm - 1;
      }
      rinfo[buf * ST_RB + tid] = ri;
    }
  };
  double rx[XK], rq[QK], rp[QK];
  auto fetch = [&](int buf) {
    const StRow ri = rinfo[buf * ST_RB + sr];
    if (need_x) {
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int c = sc + TPR * k;
        double v = 0.0;
        if (ri.orow >= 0) {
          if (c < D) v = obs[ri.orow * D + c];
          else if (c == D) v = 1.0;
        }
        rx[k] = v;
      }
    }
#pragma unroll
    for (int k = 0; k < QK; ++k) {
      const int c = sc + TPR * k;
      rq[k] = (ri.qrow >= 0 && kbase + c < K) ? q[ri.qrow * K + kbase + c] : 0.0;
    }
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < QK; ++k) {
        const int c = sc + TPR * k;
        rp[k] = (ri.prow >= 0 && c < K) ? q[ri.prow * K + c] : 0.0;
      }
    }
  };
  auto commit = [&](int bufi) {
    double* rb = rb0 + bufi * ST_RB * RS;
    double* qs = qs0 + bufi * ST_RB * QS;
    if (need_x) {
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int c = sc + TPR * k;
        if (c <= D) rb[sr * RS + c] = rx[k];
      }
    }
#pragma unroll
    for (int k = 0; k < QK; ++k) {
      const int c = sc + TPR * k;
      if (c < Kp) qs[sr * QS + c] = rq[k];
    }
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < QK; ++k) {
        const int c = sc + TPR * k;
        if (c < Kp) rb[sr * RS + QP0 + c] = rp[k];
      }
    }
  };
  if (sc == 0) {
    rb0[sr * RS + ZERO] = 0.0; rb0[sr * RS + ONE] = 1.0;
    rb0[(ST_RB + sr) * RS + ZERO] = 0.0; rb0[(ST_RB + sr) * RS + ONE] = 1.0;
  }
  // Pipeline: LDS tiles are double buffered and there is ONE barrier per 32-row stage.
  // During stage st every wave also writes stage st+1 (held in registers) into the other
  // buffer and fetches stage st+2 from HBM; the two waves that share a SIMD do this at
  // opposite ends of the stage (role B first, role A last), so one of them always feeds
  // the matrix pipe.  Row bookkeeping runs three stages ahead.
  const bool roleB = (NSPLIT == 2) && (ng == 1);
  row_info(c0, 0);
  row_info(c0 + ST_RB, 1);
  row_info(c0 + 2 * ST_RB, 2);
  __syncthreads();
  fetch(0);
  commit(0);
  if (nstage > 1) fetch(1);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int cur = st & 1;
    if (roleB) {
      if (st + 1 < nstage) commit(cur ^ 1);
      if (st + 2 < nstage) fetch((st + 2) & 3);
    }
    const double* rb = rb0 + cur * ST_RB * RS;
    const double* qs = qs0 + cur * ST_RB * QS;
    // k-steps, software pipelined by hand: the LDS reads of k-step ks+1 are issued before
    // the MFMAs of k-step ks, so their latency is covered by this wave's own matrix work
    double Bv[NTW], Ax[MT], Ay[MT];
    {
      const double* row = rb + lg * RS;
#pragma unroll
      for (int n = 0; n < NTW; ++n) Bv[n] = qs[lg * QS + (nt0 + n) * 16 + li];
#pragma unroll
      for (int m = 0; m < MT; ++m) { Ax[m] = row[fa[m]]; Ay[m] = row[fb[m]]; }
    }
#pragma unroll
    for (int ks = 0; ks < ST_RB / 4; ++ks) {
      double Bn[NTW], Axn[MT], Ayn[MT];
      if (ks + 1 < ST_RB / 4) {
        const int r = (ks + 1) * 4 + lg;
        const double* row = rb + r * RS;
#pragma unroll
        for (int n = 0; n < NTW; ++n) Bn[n] = qs[r * QS + (nt0 + n) * 16 + li];
#pragma unroll
        for (int m = 0; m < MT; ++m) { Axn[m] = row[fa[m]]; Ayn[m] = row[fb[m]]; }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const double A = Ax[m] * Ay[m];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, Bv[n], acc[m][n], 0, 0, 0);
      }
      if (ks + 1 < ST_RB / 4) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) Bv[n] = Bn[n];
#pragma unroll
        for (int m = 0; m < MT; ++m) { Ax[m] = Axn[m]; Ay[m] = Ayn[m]; }
      }
      // role A stages in the middle of its compute phase (role B did it before), so that
      // at the end of the stage both waves of a SIMD are still feeding the matrix pipe
      if (ks == ST_RB / 8 - 1 && !roleB) {
        if (st + 1 < nstage) commit(cur ^ 1);
        if (st + 2 < nstage) fetch((st + 2) & 3);
      }
    }
    row_info(c0 + (int64_t)(st + 3) * ST_RB, (st + 3) & 3);
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = (mt0 + m) * 16 + lg + 4 * r;
      if (f < Ftot && (mt0 + m) < mt_limit) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          part[((size_t)blockIdx.x * Ftot + f) * KpTot + kbase + (nt0 + n) * 16 + li] = acc[m][n][r];
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
  // fixed summation order (4 interleaved partial sums) -> bit-reproducible
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const size_t stride = (size_t)Ftot * Kp;
  const double* pp = part + (size_t)f * Kp + k;
  int c = 0;
  for (; c + 4 <= nchunk; c += 4) {
    s0 += pp[(size_t)c * stride];
    s1 += pp[(size_t)(c + 1) * stride];
    s2 += pp[(size_t)(c + 2) * stride];
    s3 += pp[(size_t)(c + 3) * stride];
  }
  for (; c < nchunk; ++c) s0 += pp[(size_t)c * stride];
  const double s = (s0 + s1) + (s2 + s3);
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

// ------------------------------------------------------------------------------------
//  K0: NIW mean-field factors -> theta (one workgroup per state).  Cholesky of sigma_mf,
//      W = (nu/2) sigma^-1 = (nu/2) L^-T L^-1, E log|Lambda| (digamma), linear and constant
//      terms of the quadratic form.  status[0] = 1 + k if sigma_k is not positive definite.
// ------------------------------------------------------------------------------------
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
__global__ __launch_bounds__(256) void k_niw_to_theta(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status) {
  extern __shared__ double sm[];
  const int S = D + 1;
  double* Lm_ = sm;            // [D][S] Cholesky factor (lower)
  double* Li = Lm_ + D * S;    // [D][S] its inverse (lower)
  double* W = Li + D * S;      // [D][S]
  double* wm = W + D * S;      // [D]
  __shared__ int bad;
  const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const double* Sg = sigma + (size_t)k * D * D;
  const double* m = mu + (size_t)k * D;
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    Lm_[i * S + j] = Sg[e];
    Li[i * S + j] = 0.0;
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  // right-looking Cholesky
  for (int j = 0; j < D; ++j) {
    const double djj = Lm_[j * S + j];
    if (!(djj > 0.0)) { if (tid == 0) bad = 1; }
    __syncthreads();
    if (bad) break;
    const double d = sqrt(djj);
    for (int i = j + 1 + tid; i < D; i += nt) Lm_[i * S + j] /= d;
    __syncthreads();
    if (tid == 0) Lm_[j * S + j] = d;
    const int n = D - 1 - j;
    for (int e = tid; e < n * n; e += nt) {
      const int a = j + 1 + e / n, b = j + 1 + e % n;
      if (b <= a) Lm_[a * S + b] -= Lm_[a * S + j] * Lm_[b * S + j];
    }
    __syncthreads();
  }
  if (bad) {
    if (tid == 0) atomicMax(status, 1 + k);
    return;
  }
  // Li = L^-1, one column per thread
  for (int c = tid; c < D; c += nt) {
    Li[c * S + c] = 1.0 / Lm_[c * S + c];
    for (int r = c + 1; r < D; ++r) {
      double s = 0.0;
      for (int jj = c; jj < r; ++jj) s -= Lm_[r * S + jj] * Li[jj * S + c];
      Li[r * S + c] = s / Lm_[r * S + r];
    }
  }
  __syncthreads();
  const double hn = 0.5 * nu[k];
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    if (j < i) continue;
    double s = 0.0;
    for (int r = j; r < D; ++r) s += Li[r * S + i] * Li[r * S + j];
    W[i * S + j] = hn * s;
    W[j * S + i] = hn * s;
  }
  __syncthreads();
  for (int i = tid; i < D; i += nt) {
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += W[i * S + j] * m[j];
    wm[i] = s;
    theta[(size_t)feat_index_d(i, D, D) * Kp + k] = 2.0 * s;
  }
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    if (j < i) continue;
    theta[(size_t)feat_index_d(i, j, D) * Kp + k] = (i == j) ? -W[i * S + i] : -2.0 * W[i * S + j];
  }
  __syncthreads();
  if (tid == 0) {
    double logdet = 0.0, llt = D * log(2.0), mWm = 0.0;
    for (int i = 0; i < D; ++i) {
      logdet += log(Lm_[i * S + i]);
      llt += digamma_d(0.5 * (nu[k] - i));
      mWm += m[i] * wm[i];
    }
    llt -= 2.0 * logdet;
    const double cst = 0.5 * llt - D / (2.0 * kappa[k]) - 0.5 * D * 1.8378770664093454835606594728112;
    theta[(size_t)feat_index_d(D, D, D) * Kp + k] = cst - mWm;
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
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) out[(size_t)gridDim.x * blockDim.x] = (double)(t1 - t0);
}
template <int NACC>
__global__ __launch_bounds__(256) void k_peak_mfma_chain(double* out, int iters) {
  double4_t c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = (double4_t){0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i][i & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[(size_t)gridDim.x * blockDim.x] = (double)(t1 - t0);
}
// MFMA + fp64 VALU overlap probe: per iteration 8 MFMAs and NF*8 independent v_fma_f64
template <int NF, bool MF>
__global__ __launch_bounds__(256) void k_peak_mix(double* out, int iters) {
  double4_t c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = (double4_t){0, 0, 0, 0};
  double f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = i;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MF) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NF; ++k) f[(i + k) & 7] = fma(f[(i + k) & 7], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i][i & 3] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
  Buf theta, fab, niw;
  int tabD = -1;
  void* pin = nullptr; size_t pin_cap = 0;   // pinned host staging for parameter uploads
  int* pin_status = nullptr;                 // pinned: NIW factorisation status (lazy check)
  bool status_pending = false;
  bool have_emission = false;
  // work
  Buf starts, ll, la, lb, q, lse_part, local_lb, logz, part, packed, scratch;
  int lastB = 0, lastLm = 0;       // shape of the intermediates currently held
  int hostB = 0, hostLm = 0;       // shape of host-uploaded lliks
  bool have_host_ll = false;
  bool have_packed = false;
  bool have_lb = false;           // lbeta materialised by the last call
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
  Buf* bufs[] = {&h->obs, &h->mask, &h->mod_init, &h->ltran, &h->Aexp, &h->AexpT, &h->theta, &h->niw,
                 &h->fab, &h->starts, &h->ll, &h->la, &h->lb, &h->q, &h->lse_part,
                 &h->local_lb, &h->logz, &h->part, &h->packed, &h->scratch};
  for (Buf* b : bufs) release(*b);
  if (h->pin) hipHostFree(h->pin);
  if (h->pin_status) hipHostFree(h->pin_status);
  hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

static int check_emission_status(svihmm_ctx* h);
static int pinned(svihmm_ctx* h, size_t bytes, void** out);
int svihmm_sync(svihmm_ctx* h) {
  CK(set_device(h));
  HIPCK(hipStreamSynchronize(h->stream));
  return check_emission_status(h);
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
  HIPCK(hipStreamSynchronize(h->stream));   // staging buffer free again
  CK(check_emission_status(h));
  void* pin = nullptr;
  CK(pinned(h, kk + K * sizeof(double), &pin));
  std::memcpy(pin, ltran, kk);
  std::memcpy((char*)pin + kk, mod_init, K * sizeof(double));
  HIPCK(hipMemcpyAsync(h->ltran.p, pin, kk, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->mod_init.p, (char*)pin + kk, K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, KS_MISC);
    hipLaunchKernelGGL(k_exp_transpose, dim3((K * K + 255) / 256), dim3(256), 0, h->stream,
                       (const double*)h->ltran.p, K, (double*)h->Aexp.p, (double*)h->AexpT.p);
  }
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(h->stream));   // both setters share the staging buffer
  h->K = K; h->have_globals = true;
  return 0;
}

static int pinned(svihmm_ctx* h, size_t bytes, void** out) {
  if (bytes > h->pin_cap) {
    if (h->pin) hipHostFree(h->pin);
    h->pin = nullptr; h->pin_cap = 0;
    HIPCK(hipHostMalloc(&h->pin, bytes + 4096, hipHostMallocDefault));
    h->pin_cap = bytes + 4096;
  }
  *out = h->pin;
  return 0;
}
// call after a stream synchronisation: reports a failed NIW factorisation of the last
// svihmm_set_emission_niw (which itself returns without waiting for the device)
static int check_emission_status(svihmm_ctx* h) {
  if (!h->status_pending) return 0;
  h->status_pending = false;
  const int st = h->pin_status ? *h->pin_status : 0;
  if (st != 0) {
    h->have_emission = false;
    return fail("svihmm_set_emission_niw: sigma_mf[" + std::to_string(st - 1) + "] is not positive definite");
  }
  return 0;
}

// ---- NIW -> theta on the device (k_niw_to_theta) -------------------------------------
static inline int feat_index(int a, int b, int D) {  // 0 <= a <= b <= D
  return a * (D + 1) - a * (a - 1) / 2 + (b - a);
}

static int upload_feature_table(svihmm_ctx* h, int D, int K) {
  const int F = (D + 1) * (D + 2) / 2, Fp = (F + 15) / 16 * 16, Kp = (K + 15) / 16 * 16;
  if (h->tabD == D && h->Fp == Fp) { h->F = F; h->Kp = Kp; return 0; }
  std::vector<int> fab(Fp, (D + 1) | ((D + 1) << 16));  // padding -> zero slot
  for (int a = 0; a <= D; ++a)
    for (int b = a; b <= D; ++b) fab[feat_index(a, b, D)] = a | (b << 16);
  CK(ensure(h->fab, fab.size() * sizeof(int)));
  HIPCK(hipMemcpyAsync(h->fab.p, fab.data(), fab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->tabD = D; h->F = F; h->Fp = Fp; h->Kp = Kp;
  return 0;
}

int svihmm_set_emission_niw(svihmm_ctx* h, int32_t K, int32_t D, const double* mu,
                            const double* sigma, const double* kappa, const double* nu) {
  if (!h || K <= 0 || D <= 0 || !mu || !sigma || !kappa || !nu)
    return fail("svihmm_set_emission_niw: bad arguments");
  if ((size_t)(3 * D * (D + 1) + D) * 8 > 150 * 1024)
    return fail("svihmm_set_emission_niw: D too large");
  CK(set_device(h));
  CK(upload_feature_table(h, D, K));
  const int Fp = h->Fp, Kp = h->Kp;
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  const size_t nin = nmu + nsg + 2 * (size_t)K;
  CK(ensure(h->niw, nin * sizeof(double) + 64));
  CK(ensure(h->theta, (size_t)Fp * Kp * sizeof(double)));
  double* dmu = (double*)h->niw.p;
  double* dsg = dmu + nmu;
  double* dka = dsg + nsg;
  double* dnu = dka + K;
  int* dstatus = (int*)(dnu + K);
  // one pinned staging buffer, one H2D copy; the previous upload must have left it
  HIPCK(hipStreamSynchronize(h->stream));
  CK(check_emission_status(h));
  void* pin = nullptr;
  CK(pinned(h, nin * sizeof(double), &pin));
  double* hp = (double*)pin;
  std::memcpy(hp, mu, nmu * sizeof(double));
  std::memcpy(hp + nmu, sigma, nsg * sizeof(double));
  std::memcpy(hp + nmu + nsg, kappa, K * sizeof(double));
  std::memcpy(hp + nmu + nsg + K, nu, K * sizeof(double));
  if (!h->pin_status) HIPCK(hipHostMalloc((void**)&h->pin_status, 64, hipHostMallocDefault));
  *h->pin_status = 0;
  HIPCK(hipMemcpyAsync(dmu, hp, nin * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemsetAsync(dstatus, 0, sizeof(int), h->stream));
  HIPCK(hipMemsetAsync(h->theta.p, 0, (size_t)Fp * Kp * sizeof(double), h->stream));
  {
    ProfScope ps(h, KS_MISC);
    const size_t lds = (size_t)(3 * D * (D + 1) + D) * sizeof(double);
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_niw_to_theta, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_niw_to_theta, dim3(K), dim3(256), lds, h->stream, (const double*)dmu,
                       (const double*)dsg, (const double*)dka, (const double*)dnu, K, D, Kp,
                       (double*)h->theta.p, dstatus);
    HIPCK(hipGetLastError());
  }
  // status comes back asynchronously; it is examined at the next synchronising call
  HIPCK(hipMemcpyAsync(h->pin_status, dstatus, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  h->status_pending = true;
  h->eK = K; h->eD = D; h->have_emission = true;
  return 0;
}

// feature table is also needed by the statistics kernels when only host lliks are used
static int ensure_feature_table(svihmm_ctx* h) {
  return upload_feature_table(h, h->D, h->K);
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
    int MT = h->variant[3] > 0 ? h->variant[3] : 2;
    if (MT != 2 && MT != 4) MT = 2;
    size_t lds = (size_t)(64 * MT) * DS * 8 + (size_t)h->Fp * 4 + 64 * MT;
    if (lds > 150 * 1024 && MT == 4) { MT = 2; lds = (size_t)128 * DS * 8 + (size_t)h->Fp * 4 + 128; }
    if (lds > 150 * 1024) var = 1;
    else {
      const int ntile = Kp / 16;
      const int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
      const int rows = 64 * MT;
      dim3 grid((unsigned)((n + rows - 1) / rows), ntile / NT);
#define EMM_LAUNCH(NTV, MTV)                                                                 \
  do {                                                                                        \
    if (lds > 64 * 1024)                                                                      \
      hipFuncSetAttribute((const void*)k_emission_mfma<NTV, MTV>,                             \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL((k_emission_mfma<NTV, MTV>), grid, dim3(256), lds, h->stream,          \
                       (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K, \
                       Kp, h->Fp, (const double*)h->theta.p, (const int*)h->fab.p, flags,     \
                       (double*)h->ll.p);                                                     \
  } while (0)
      if (MT == 4) { if (NT == 4) EMM_LAUNCH(4, 4); else if (NT == 2) EMM_LAUNCH(2, 4); else EMM_LAUNCH(1, 4); }
      else { if (NT == 4) EMM_LAUNCH(4, 2); else if (NT == 2) EMM_LAUNCH(2, 2); else EMM_LAUNCH(1, 2); }
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

// forward sweep, then backward sweep with the posterior fused (K <= 64).  want_lb: also
// materialise lbeta (API readback); total: write sum_b local_lb into packed[last].
static int launch_fb_fused(svihmm_ctx* h, int B, int Lm, bool want_lb, bool total) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const int K = h->K;
  const size_t n = (size_t)B * Lm * K * sizeof(double);
  CK(ensure(h->la, n));
  CK(ensure(h->q, n));
  if (want_lb) CK(ensure(h->lb, n));
  CK(ensure(h->local_lb, (size_t)B * sizeof(double)));
  CK(ensure(h->logz, (size_t)B * sizeof(double)));
  CK(ensure(h->packed, (size_t)svihmm_packed_size(K, h->D > 0 ? h->D : 1) * sizeof(double)));
  const int NW = (K + 15) / 16;
  dim3 grid((B + 15) / 16);
  const double* ll = (const double*)h->ll.p;
  double* la = (double*)h->la.p;
  double* lb = want_lb ? (double*)h->lb.p : nullptr;
  double* q = (double*)h->q.p;
  double* llb = (double*)h->local_lb.p;
  double* lz = (double*)h->logz.p;
  {
    ProfScope ps(h, KS_FB);
#define FWD(NWV, F) hipLaunchKernelGGL((k_fwd_mfma<NWV, F>), grid, dim3(64 * NWV), 0, h->stream, ll, \
                                       (const double*)h->Aexp.p, (const double*)h->mod_init.p, B, Lm, K, la, llb, lz)
    const bool full = (K == 16 * NW);
    if (NW == 1) { if (full) FWD(1, true); else FWD(1, false); }
    else if (NW == 2) { if (full) FWD(2, true); else FWD(2, false); }
    else if (NW == 3) { if (full) FWD(3, true); else FWD(3, false); }
    else { if (full) FWD(4, true); else FWD(4, false); }
#undef FWD
    HIPCK(hipGetLastError());
  }
  {
    ProfScope ps(h, KS_POSTERIOR);
    const bool full = (K == 16 * NW);
#define BWD(NWV, F, W) hipLaunchKernelGGL((k_bwd_mfma<NWV, F, W>), grid, dim3(64 * NWV), 0, h->stream, ll, \
                                          (const double*)h->AexpT.p, (const double*)la, (const double*)lz, B, Lm, K, lb, q)
#define BWD2(NWV) do { if (full) { if (want_lb) BWD(NWV, true, true); else BWD(NWV, true, false); } \
                       else { if (want_lb) BWD(NWV, false, true); else BWD(NWV, false, false); } } while (0)
    if (NW == 1) BWD2(1); else if (NW == 2) BWD2(2); else if (NW == 3) BWD2(3); else BWD2(4);
#undef BWD2
#undef BWD
    if (total) {
      double* lbtot = (double*)h->packed.p + (svihmm_packed_size(K, h->D) - 1);
      hipLaunchKernelGGL(k_sum_lb, dim3(1), dim3(256), 0, h->stream, (const double*)llb, B, lbtot);
    }
    HIPCK(hipGetLastError());
  }
  return 0;
}

// messages + posterior for a window batch: picks the fused MFMA sweeps or the
// wave-per-window kernels (concurrent directions; better latency for small batches)
static int run_fb(svihmm_ctx* h, int B, int Lm, bool want_lb, bool total) {
  int var = h->variant[2];
  if (h->K > 64) var = 1;
  if (var == 0) var = (B >= 192) ? 2 : 1;
  h->have_lb = (var != 2) || want_lb;
  if (var == 2) return launch_fb_fused(h, B, Lm, want_lb, total);
  CK(launch_fb(h, B, Lm, 0, 2));
  return launch_posterior(h, B, Lm, total);
}

static int launch_stats(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  // statistics over the inner segment [off, off+Lm) of each window of length Lq
  CK(ensure_feature_table(h));
  const int D = h->D, K = h->K, Kp = h->Kp, Fp = h->Fp, F = h->F;
  const int Ftot = Fp + Kp;
  const int64_t n = (int64_t)B * Lm;
  // row chunking: ~256 row chunks (x feature/state tiles => >= 1024 workgroups at D=32)
  // so that small minibatches still spread over the 256 CUs; chunk = multiple of ST_RB
  // (one resident workgroup per CU for the pipelined kernel: 128 chunks x 2 passes = 256)
  const int target_chunks = (n >= 128 * 1024) ? 128 : 256;
  int64_t rpc = (n + target_chunks - 1) / target_chunks;
  rpc = (rpc + ST_RB - 1) / ST_RB * ST_RB;
  int64_t nchunk = (n + rpc - 1) / rpc;
  CK(ensure(h->part, (size_t)nchunk * Ftot * Kp * sizeof(double)));
  CK(ensure(h->packed, (size_t)svihmm_packed_size(K, D) * sizeof(double)));
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  int var = h->variant[1];
  if (var == 0) var = 3;
  {
    ProfScope ps(h, KS_STATS);
    if (var == 3) {
      // pipelined VGPR-form GEMM.  K <= 64: all tiles (statistics + transition) in one launch.
      // K > 64: state groups of 64 in grid.z for the emission-statistics tiles; the K x K
      // transition tiles (which need q[t-1] of ALL states as operand rows) go to k_stats_mfma.
      const bool big = Kp > 64;
      const int NTt = big ? 4 : Kp / 16;                // n-tiles per workgroup
      const int KpW = 16 * NTt;
      const int NSPLIT = (NTt == 4) ? 2 : 1;
      const int TPR = 8 * NSPLIT;
      const int RS = (D + 3 + KpW) | 1;
      const size_t lds = 2 * ((size_t)ST_RB * RS + (size_t)ST_RB * (KpW + 1)) * 8 + 4 * ST_RB * sizeof(StRow);
      const int mtiles = Ftot / 16;
      const int mt_limit = big ? Fp / 16 : mtiles;
      const int xk = (D + 1 + TPR - 1) / TPR;
      if (lds > 150 * 1024 || xk > 9 || (big && Kp % 64 != 0)) var = 2;
      else {
        dim3 grid((unsigned)nchunk, (mt_limit + 4 * 5 - 1) / (4 * 5), big ? Kp / 64 : 1);
#define ST3(NTW, NS, XKV)                                                                        \
  do {                                                                                           \
    if (lds > 64 * 1024)                                                                         \
      hipFuncSetAttribute((const void*)k_stats_mfma3<5, NTW, NS, XKV>,                           \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    hipLaunchKernelGGL((k_stats_mfma3<5, NTW, NS, XKV>), grid, dim3(256 * NS), lds, h->stream,   \
                       (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K, Fp, \
                       F, (const int*)h->fab.p, (const double*)h->q.p, rpc, flags, Lq, off,       \
                       (double*)h->part.p, Kp, mt_limit);                                         \
  } while (0)
#define ST3X(NTW, NS) do { if (xk <= 1) ST3(NTW, NS, 1); else if (xk <= 3) ST3(NTW, NS, 3); else if (xk <= 5) ST3(NTW, NS, 5); else ST3(NTW, NS, 9); } while (0)
        if (NTt == 4) ST3X(2, 2); else if (NTt == 3) ST3X(3, 1); else if (NTt == 2) ST3X(2, 1); else ST3X(1, 1);
#undef ST3X
#undef ST3
        if (big) {   // transition tiles [Fp/16, Ftot/16)
          const int NT = 4, MT = 3;
          const int DS = (D + 2) | 1;
          const size_t lds2 = ((size_t)ST_RB * DS + (size_t)ST_RB * (16 * NT + 1) + (size_t)ST_RB * (Kp + 1)) * 8;
          if (lds2 > 150 * 1024) return fail("statistics: K too large for the LDS-staged transition kernel");
          if (lds2 > 64 * 1024)
            hipFuncSetAttribute((const void*)k_stats_mfma<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
          const int ttiles = Kp / 16;
          dim3 g2((unsigned)nchunk, (ttiles + 4 * MT - 1) / (4 * MT), Kp / 64);
          hipLaunchKernelGGL((k_stats_mfma<3, 4>), g2, dim3(256), lds2, h->stream,
                             (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K,
                             Kp, Fp, F, (const int*)h->fab.p, (const double*)h->q.p, rpc, flags,
                             Lq, off, (double*)h->part.p, Fp / 16);
        }
      }
    }
    if (var == 2) {
      const int ntile = Kp / 16;
      const int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
      const int MT = 3;
      const int DS = (D + 2) | 1;
      const size_t lds = ((size_t)ST_RB * DS + (size_t)ST_RB * (16 * NT + 1) + (size_t)ST_RB * (Kp + 1)) * 8;
      if (lds > 150 * 1024) var = 1;
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
                       Lq, off, (double*)h->part.p, 0);                                       \
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
  CK(check_emission_status(h));
  h->lastB = B; h->lastLm = Lm;
  return 0;
}

int svihmm_forward_backward(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                            uint32_t flags, double* out_lalpha, double* out_lbeta,
                            double* out_var_x, double* out_local_lb) {
  if (!h) return fail("svihmm_forward_backward: NULL handle");
  CK(set_device(h));
  CK(prepare_ll(h, starts, B, Lm, flags, false));
  CK(run_fb(h, B, Lm, out_lbeta != nullptr, false));
  const size_t n = (size_t)B * Lm * h->K * sizeof(double);
  if (out_lalpha) CK(d2h(h, out_lalpha, h->la.p, n));
  if (out_lbeta) CK(d2h(h, out_lbeta, h->lb.p, n));
  if (out_var_x) CK(d2h(h, out_var_x, h->q.p, n));
  if (out_local_lb) CK(d2h(h, out_local_lb, h->local_lb.p, (size_t)B * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  CK(check_emission_status(h));
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
  CK(run_fb(h, B, Lm, (flags & SVIHMM_KEEP_LBETA) != 0, true));
  CK(launch_stats(h, B, Lm, inner_off, inner_len, flags));
  h->have_packed = true;
  h->lastB = B; h->lastLm = Lm;
  if (out_packed) {
    CK(d2h(h, out_packed, h->packed.p, (size_t)svihmm_packed_size(h->K, h->D) * sizeof(double)));
    HIPCK(hipStreamSynchronize(h->stream));
    CK(check_emission_status(h));
  }
  return 0;
}

int svihmm_read_packed(svihmm_ctx* h, double* out_packed) {
  if (!h || !out_packed) return fail("svihmm_read_packed: bad arguments");
  if (!h->have_packed) return fail("svihmm_read_packed: no statistics computed yet");
  CK(set_device(h));
  CK(d2h(h, out_packed, h->packed.p, (size_t)svihmm_packed_size(h->K, h->D) * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  CK(check_emission_status(h));
  return 0;
}

int svihmm_read_intermediate(svihmm_ctx* h, int32_t what, double* out) {
  if (!h || !out) return fail("svihmm_read_intermediate: bad arguments");
  if (h->lastB <= 0) return fail("svihmm_read_intermediate: nothing computed yet");
  CK(set_device(h));
  Buf* src[] = {&h->ll, &h->la, &h->lb, &h->q};
  if (what < 0 || what > 3) return fail("svihmm_read_intermediate: bad selector");
  if (what == 2 && !h->have_lb)
    return fail("svihmm_read_intermediate: lbeta was not materialised (pass SVIHMM_KEEP_LBETA)");
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
  if (what == 2 && !h->have_lb)
    return fail("svihmm_read_rows: lbeta was not materialised (pass SVIHMM_KEEP_LBETA)");
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
  // which: 0 mfma (8 blocks/CU), 1 fma, 2 mfma 1 block/CU (1 wave/SIMD), 3 mfma 2 blocks/CU;
  // +16: return s_memtime ticks per loop iteration of block 0 instead of TFLOP/s
  const bool ticks = (which & 16) != 0;
  if (which >= 200) {   // 200 + 10*nf + mf : mix probe, 2 blocks/CU; returns ns per loop iteration
    const int nf = (which - 200) / 10, mf = (which - 200) % 10;
    const int blocks = 512, threads = 256, iters = 4000;
    CK(ensure(h->scratch, ((size_t)blocks * threads + 8) * sizeof(double)));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      HIPCK(hipEventRecord(e0, h->stream));
#define PM(N, M) hipLaunchKernelGGL((k_peak_mix<N, M>), dim3(blocks), dim3(threads), 0, h->stream, (double*)h->scratch.p, iters)
      if (nf == 0) PM(0, true);
      else if (nf == 8) { if (mf) PM(8, true); else PM(8, false); }
      else { if (mf) PM(16, true); else PM(16, false); }
#undef PM
      HIPCK(hipEventRecord(e1, h->stream));
      HIPCK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *tflops_out = (double)ms * 1e6 / iters;
    return 0;
  }
  if (which >= 100) {   // 100 + nacc: 1 block/CU (1 wave/SIMD), nacc independent accumulators
    const int nacc = which - 100, blocks = 256, threads = 256, iters = 4000;
    CK(ensure(h->scratch, ((size_t)blocks * threads + 8) * sizeof(double)));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      HIPCK(hipEventRecord(e0, h->stream));
#define PK(N) hipLaunchKernelGGL(k_peak_mfma_chain<N>, dim3(blocks), dim3(threads), 0, h->stream, (double*)h->scratch.p, iters)
      if (nacc == 1) PK(1); else if (nacc == 2) PK(2); else if (nacc == 4) PK(4); else if (nacc == 8) PK(8);
      else if (nacc == 12) PK(12); else PK(16);
#undef PK
      HIPCK(hipEventRecord(e1, h->stream));
      HIPCK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    // nanoseconds per MFMA per SIMD
    *tflops_out = (double)ms * 1e6 / ((double)iters * (nacc == 1 || nacc == 2 || nacc == 4 || nacc == 8 || nacc == 12 ? nacc : 16));
    return 0;
  }
  which &= 15;
  const int bpc = which == 2 ? 1 : which == 3 ? 2 : 8;
  const int blocks = 256 * bpc, threads = 256, iters = 20000;
  CK(ensure(h->scratch, ((size_t)blocks * threads + 8) * sizeof(double)));
  hipEvent_t e0, e1;
  HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCK(hipEventRecord(e0, h->stream));
    if (which != 1)
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
  if (which != 1) flops = (double)blocks * (threads / 64) * (double)iters * 4.0 * 2048.0;
  else flops = (double)blocks * threads * (double)iters * 8.0 * 2.0;
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  if (ticks && which != 1) {
    double tk = 0;
    HIPCK(hipMemcpy(&tk, (double*)h->scratch.p + (size_t)blocks * threads, sizeof(double), hipMemcpyDeviceToHost));
    *tflops_out = tk / iters;
  }
  return 0;
}

}  // extern "C"
