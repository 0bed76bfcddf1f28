// kernels_misc.h -- exp/transpose of ltran, MFMA layout self-test, small utility kernels.
// Part of libsvihmm_hip.so; compiled in svihmm_hip.hip.
#pragma once

// small utility kernels
// `src` = [ltran K*K | mod_init K] in a pinned host slot (device-visible): the kernel pulls the
// globals over PCIe itself (device copies of both + exp(ltran) and its transpose).  A kernel
// reading mapped host memory starts ~8 us after its predecessor; a hipMemcpyAsync of the same
// few KB took 20-40 us of stream time on this stack (kernel trace, tools/trace_gaps.py).
__global__ void k_exp_transpose(const double* __restrict__ src, int K, double* __restrict__ ltran,
                                double* __restrict__ mod_init, double* __restrict__ A,
                                double* __restrict__ AT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < K) mod_init[idx] = src[(size_t)K * K + idx];
  if (idx >= K * K) return;
  const int i = idx / K, j = idx - i * K;
  const double l = src[idx];
  const double v = exp(l);
  ltran[idx] = l;
  A[idx] = v;
  AT[(size_t)j * K + i] = v;
}
// small host -> device upload as a kernel: dst[i] = src[i], src in a pinned (device-visible)
// host slot, 8-byte words
__global__ __launch_bounds__(256) void k_pull(const unsigned long long* __restrict__ src,
                                              unsigned long long* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dst[i] = src[i];
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

// ------------------------------------------------------------------------------------
//  K7: predictive log-probability of the held-out rows (hmmsgd_metaobs.py:1086-1145,
//      hmmbase.py:322-340):  sum over masked rows of LSE_k( log(var_x[t,k] + 1e-9) + ll[t,k] )
//      with ll evaluated on the TRUE observations.  One 16-lane row per (window, t) row,
//      fixed row -> block assignment and per-block partials: deterministic.
//      grid (nblk), block 256; part[2*b] = sum, part[2*b+1] = count.
// ------------------------------------------------------------------------------------
#define PRED_ROWS_PER_BLOCK 2048
template <int KT>
__global__ __launch_bounds__(256) void k_pred_logprob(
    const double* __restrict__ q, const double* __restrict__ ll, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int K, double* __restrict__ part) {
  __shared__ double red[2][16];
  const int li = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * PRED_ROWS_PER_BLOCK;
  const int64_t r1 = imin64(nrows, r0 + PRED_ROWS_PER_BLOCK);
  double acc = 0.0, cnt = 0.0;
  for (int64_t g = r0 + rg; g < r1; g += 16) {
    const int64_t b = g / Lm;
    const int64_t orow = starts[b] + (g - b * Lm);
    if (!mask[orow]) continue;             // uniform within the 16-lane row
    double v[KT], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const int k = li + 16 * c;
      v[c] = k < K ? log(q[g * K + k] + 1e-9) + ll[g * K + k] : -INFINITY;
      mx = fmax(mx, v[c]);
    }
    mx = row16_max(mx);
    double sm = 0.0;
#pragma unroll
    for (int c = 0; c < KT; ++c) sm += (li + 16 * c < K) ? exp(v[c] - mx) : 0.0;
    sm = row16_sum(sm);
    acc += mx + log(sm);
    cnt += 1.0;
  }
  if (li == 0) { red[0][rg] = acc; red[1][rg] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int i = 0; i < 16; ++i) { a += red[0][i]; c += red[1][i]; }
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = c;
  }
}
__global__ void k_pred_final(const double* __restrict__ part, int nblk, double* __restrict__ out2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int i = 0; i < nblk; ++i) { a += part[2 * i]; c += part[2 * i + 1]; }
    out2[0] = c > 0.0 ? a / c : NAN;
    out2[1] = c;
  }
}

// ------------------------------------------------------------------------------------
//  State decoding for the evaluation metrics (hmmbase.py:346-355 hamming_dist:
//  np.argmax(full_var_x, axis=1) -> util.munkres_match's K x K count matrix, util.py:236-277).
//  One wave per row, lane = state (strided by 64 above 64 states); the first maximum wins like
//  np.argmax.  With true labels: conf[pred * K + true] += 1, counted in an LDS table per
//  workgroup (K <= 64) and flushed with one atomic per touched cell -- T = 1e6 rows move
//  4 MB of labels instead of the 512 MB of var_x the host route reads back.
// ------------------------------------------------------------------------------------
#define ARGMAX_ROWS_PER_BLOCK 1024
__global__ __launch_bounds__(256) void k_state_argmax(
    const double* __restrict__ q, int64_t nrows, int K, const int32_t* __restrict__ true_sts,
    int32_t* __restrict__ z, unsigned long long* __restrict__ conf) {
  extern __shared__ unsigned int ctab[];         // [K*K] when K <= 64 and labels are given
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool lds_tab = true_sts != nullptr && K <= 64;
  if (lds_tab) {
    for (int i = threadIdx.x; i < K * K; i += 256) ctab[i] = 0u;
    __syncthreads();
  }
  const int64_t r0 = (int64_t)blockIdx.x * ARGMAX_ROWS_PER_BLOCK;
  const int64_t r1 = imin64(nrows, r0 + ARGMAX_ROWS_PER_BLOCK);
  for (int64_t g = r0 + wave; g < r1; g += 4) {
    const double* __restrict__ row = q + g * K;
    double best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
      const double v = row[k];
      if (v > best || bi == 0x7fffffff) { best = v; bi = k; }   // strictly greater: first maximum
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      const bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi));
      best = take ? ov : best;
      bi = take ? oi : bi;
    }
    if (lane == 0) {
      if (z) z[g] = bi;
      if (true_sts) {
        const int tr = true_sts[g];
        if (tr >= 0 && tr < K) {
          if (lds_tab) atomicAdd(&ctab[bi * K + tr], 1u);
          else atomicAdd(&conf[(size_t)bi * K + tr], 1ull);
        }
      }
    }
  }
  if (lds_tab) {
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += 256) {
      const unsigned int c = ctab[i];
      if (c) atomicAdd(&conf[i], (unsigned long long)c);
    }
  }
}

// ------------------------------------------------------------------------------------
//  Synthetic sequences on the device (gen_synthetic.py:27-44 generate_data: start in state 0,
//  z_t ~ tran[z_{t-1}] by np.random.choice's inverse CDF -- cdf = cumsum(p) / cumsum(p)[-1],
//  searchsorted(cdf, u, side='right') -- and x_t = mean[z_t] + chol[z_t] n_t, n_t ~ N(0, I)).
//  Randomness is counter based (Philox4x32-10, key = seed, counter = (row, stream)): row t
//  draws its transition uniform from stream 0 and its normals, two per call by Box-Muller,
//  from streams 1 + j/2 -- any thread can recompute any draw, and the NumPy oracle does the
//  same.  The state chain is a composition of per-row maps z_t = F_t(z_{t-1}) like the FFBS
//  sampler: k_gen_paths (one wave per chunk, lane = state before the chunk, binary search in
//  the CDF row), k_gen_compose (prefix composition of the chunk maps), k_gen_gather.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long row,
                                              unsigned stream, unsigned (&o)[4]) {
  unsigned c0 = (unsigned)row, c1 = (unsigned)(row >> 32), c2 = stream, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ double u53(unsigned hi, unsigned lo) {   // [0, 1), 53 bits
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

template <int KMAX>
__global__ __launch_bounds__(64) void k_gen_paths(const double* __restrict__ cdf, unsigned long long seed,
                                                  int64_t T, int K, int Ls,
                                                  unsigned char* __restrict__ path) {
  extern __shared__ double gcdf[];          // [K][KMAX + 1]
  const int lane = threadIdx.x;
  for (int e = lane; e < K * K; e += 64) gcdf[(e / K) * (KMAX + 1) + (e % K)] = cdf[e];
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * Ls;
  const int64_t hi = lo + Ls < T ? lo + Ls : T;
  int cur = lane < K ? lane : 0;
  for (int64_t t = lo; t < hi; ++t) {
    unsigned w[4];
    philox4x32_10(seed, (unsigned long long)t, 0u, w);
    const double u = u53(w[0], w[1]);
    // number of CDF entries <= u (searchsorted side='right'), clipped to K - 1
    const double* __restrict__ row = gcdf + cur * (KMAX + 1);
    int a = 0, b = K;                       // answer in [a, b]
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int mid = (a + b) >> 1;
      const bool le = (a < b) && row[mid < K ? mid : K - 1] <= u;
      a = le ? mid + 1 : a;
      b = ((a < b) && !le) ? mid : b;
    }
    cur = t == 0 ? 0 : (a < K ? a : K - 1);
    if (lane < KMAX) path[t * KMAX + lane] = (unsigned char)cur;
  }
}
// maps[c][s] = state at the LAST row of chunk c given the state s before its first row;
// prefix composition; entry[c] = state before chunk c (chunk 0: 0, unused -- row 0 is state 0)
__global__ __launch_bounds__(1024) void k_gen_compose(const unsigned char* __restrict__ path, int64_t T,
                                                      int KS, int Ls, int C, unsigned char* __restrict__ mA,
                                                      unsigned char* __restrict__ mB,
                                                      unsigned char* __restrict__ entry) {
  const int n = C * KS;
  for (int e = threadIdx.x; e < n; e += 1024) {
    const int c = e / KS, x = e - c * KS;
    const int64_t last = ((int64_t)(c + 1) * Ls < T ? (int64_t)(c + 1) * Ls : T) - 1;
    mA[e] = path[last * KS + x];
  }
  __threadfence_block();
  __syncthreads();
  unsigned char* src = mA;
  unsigned char* dst = mB;
  for (int d = 1; d < C; d <<= 1) {
    for (int e = threadIdx.x; e < n; e += 1024) {
      const int c = e / KS, x = e - c * KS;
      dst[e] = (c - d >= 0) ? src[c * KS + src[(c - d) * KS + x]] : src[e];
    }
    __threadfence_block();
    __syncthreads();
    unsigned char* t = src; src = dst; dst = t;
  }
  for (int c = threadIdx.x; c < C; c += 1024) entry[c] = c > 0 ? src[(c - 1) * KS] : 0;
}
__global__ __launch_bounds__(256) void k_gen_gather(const unsigned char* __restrict__ path,
                                                    const unsigned char* __restrict__ entry, int64_t T,
                                                    int KS, int Ls, int32_t* __restrict__ z) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  z[t] = path[t * KS + entry[t / Ls]];
}
// x_t = mean[z_t] + chol[z_t] n_t (lower triangular), one thread per row
__global__ __launch_bounds__(256) void k_gen_obs(const int32_t* __restrict__ z, const double* __restrict__ means,
                                                 const double* __restrict__ chols, unsigned long long seed,
                                                 int64_t T, int D, double* __restrict__ obs) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int s = z[t];
  const double* __restrict__ m = means + (size_t)s * D;
  const double* __restrict__ Lc = chols + (size_t)s * D * D;
  double* __restrict__ x = obs + t * D;
  for (int i = 0; i < D; ++i) x[i] = m[i];
  for (int p = 0; 2 * p < D; ++p) {
    unsigned w[4];
    philox4x32_10(seed, (unsigned long long)t, 1u + (unsigned)p, w);
    const double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
    const double r = sqrt(-2.0 * log(1.0 - u1));
    double sn, cs;
    sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
    const double n0 = r * cs, n1 = r * sn;
    const int j0 = 2 * p, j1 = 2 * p + 1;
    for (int i = j0; i < D; ++i) x[i] = fma(Lc[i * D + j0], n0, x[i]);
    if (j1 < D)
      for (int i = j1; i < D; ++i) x[i] = fma(Lc[i * D + j1], n1, x[i]);
  }
}

// packed statistics -> host-visible (pinned, mapped) mirror.  An ordinary kernel launch right
// behind k_finalize / the all-reduce: the runtime's D2H copy command starts ~0.1 ms after its
// producer in the kernel trace, this one after the usual ~6 us.
__global__ void k_mirror(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// obs[t][d] -= shift[d], in place, for n consecutive entries of whole rows (the handle keeps the
// resident copy centred; svihmm_shift_obs).  ROUND: the result is an integer-valued symbol column
// again (a Categorical table follows data that had been centred for a NIW family).
template <bool ROUND>
__global__ __launch_bounds__(256) void k_shift_obs(double* __restrict__ obs, int64_t n, int D,
                                                   const double* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const double v = obs[i] - shift[i % D];
    obs[i] = ROUND ? rint(v) : v;
  }
}
// rows of [K][D] means -= shift (the NIW factors / prior resident on the device follow a change of
// the handle's shift)
__global__ __launch_bounds__(256) void k_shift_means(double* __restrict__ mu, int n, int D,
                                                     const double* __restrict__ shift) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) mu[i] -= shift[i % D];
}

// Packed NIW statistics [A_raw | xbar | neff | S | lb] of observations x - c  ->  the same
// statistics of x (sgn = +1), or back (sgn = -1; with cs = sgn * c in both cases):
//   xbar_k += n_k cs,   S_k += cs xbar_k' + xbar_k cs' + n_k cs cs'      (xbar_k of the SOURCE).
// Out of place; dst may be the host-visible mirror (then this is k_mirror with the handle's shift
// undone: the C ABI hands out statistics in the caller's coordinates).
// diag: layout [A_raw | xbar | neff | xsq K*D | lb]:  xsq_ka += 2 cs_a xbar_ka + n_k cs_a^2.
__global__ __launch_bounds__(256) void k_packed_shift(const double* __restrict__ src, double* __restrict__ dst,
                                                      int n, int K, int D, const double* __restrict__ c,
                                                      double sgn, int diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int oX = K * K, oN = oX + K * D, oS = oN + K;
  double v = src[i];
  if (i >= oX && i < oN) {
    const int k = (i - oX) / D, a = (i - oX) - k * D;
    v += src[oN + k] * (sgn * c[a]);
  } else if (diag && i >= oS && i < oS + K * D) {
    const int k = (i - oS) / D, a = (i - oS) - k * D;
    const double ca = sgn * c[a];
    v += 2.0 * ca * src[oX + k * D + a] + src[oN + k] * ca * ca;
  } else if (!diag && i >= oS && i < oS + K * D * D) {
    const int e = i - oS;
    const int k = e / (D * D), r = e - k * D * D, a = r / D, b = r - a * D;
    const double ca = sgn * c[a], cb = sgn * c[b];
    // order-symmetric in (a, b) with every product rounded (no FMA contraction: fused, the two
    // cross terms round differently for (a, b) and (b, a)): S stays bit-symmetric
    {
#pragma clang fp contract(off)
      const double p1 = ca * src[oX + k * D + b], p2 = cb * src[oX + k * D + a];
      const double cross = fmin(p1, p2) + fmax(p1, p2);
      const double quad = src[oN + k] * (ca * cb);
      v = v + (cross + quad);
    }
  }
  dst[i] = v;
}

// column means of the finite entries of rows 0, stride, 2 stride, ... (nsamp of them): the centre of a
// resident copy whose upload skipped the centring (a Categorical table was active); block = column
__global__ __launch_bounds__(256) void k_col_mean(const double* __restrict__ obs, int D, int64_t stride,
                                                  int64_t nsamp, double* __restrict__ c) {
  __shared__ double rs[256];
  __shared__ double rn[256];
  const int d = blockIdx.x;
  double s = 0.0, n = 0.0;
  for (int64_t i = threadIdx.x; i < nsamp; i += 256) {
    const double v = obs[(size_t)(i * stride) * D + d];
    if (v > -1.7e308 && v < 1.7e308) { s += v; n += 1.0; }
  }
  rs[threadIdx.x] = s; rn[threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rn[threadIdx.x] += rn[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double m = rn[0] > 0.0 ? rs[0] / rn[0] : 0.0;
    c[d] = (m > -1.7e308 && m < 1.7e308) ? m : 0.0;
  }
}
