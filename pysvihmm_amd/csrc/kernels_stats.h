// kernels_stats.h -- K4 expected sufficient statistics (VALU fallback + fp64 MFMA GEMMs), K5 deterministic finalize.
// Part of libsvihmm_hip.so; included by svihmm_hip.hip (single translation unit).
#pragma once

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
