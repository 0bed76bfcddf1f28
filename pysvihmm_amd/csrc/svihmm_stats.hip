// svihmm_stats.hip -- fp64 MFMA statistics kernels (own translation unit, see Makefile).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "svihmm_common.h"

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
//  K4c: statistics GEMM, software-pipelined (K <= 64).  Same math as K4b; differences:
//   * the next 32-row stage is fetched from HBM into registers while the current stage
//     runs on the matrix pipe (global -> reg early, reg -> LDS after the compute);
//   * row bookkeeping (obs row, q row, wrap predecessor, mask) is computed once per stage
//     by 32 lanes instead of per element (no integer divisions in the copy loops);
//   * MT = 5 m-tiles per wave: the 36 emission + 4 transition tiles of K=64, D=32 split
//     into two balanced workgroup passes, so q is read twice instead of four times.
//  grid (nchunk, ceil(Ftot/16 / (4*MT))), block 256.
// ------------------------------------------------------------------------------------
struct StRow {
  long long orow;   // obs row, -1: out of range or masked (x~ = 0)
  long long qrow;   // q row, -1: out of range
  long long prow;   // predecessor q row, -1: none
};

template <int MT, int NT, int XK>
__global__ __launch_bounds__(256) void k_stats_mfma2(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Fp, int F,
    const int* __restrict__ fab, const double* __restrict__ q, int64_t rows_per_chunk,
    uint32_t flags, int Lq, int off, double* __restrict__ part) {
  constexpr int Kp = 16 * NT;
  constexpr int QS = Kp + 1;
  extern __shared__ double smem[];
  const int DS = (D + 2) | 1;
  double* xs = smem;               // [32][DS]
  double* qs = xs + ST_RB * DS;    // [32][QS]
  double* qp = qs + ST_RB * QS;    // [32][QS]  (column Kp is a zero column)
  StRow* rinfo = reinterpret_cast<StRow*>(qp + ST_RB * QS);  // [2][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int Ftot = Fp + Kp;
  const int mt0 = (blockIdx.y * 4 + wave) * MT;
  const int wg_m0 = blockIdx.y * 4 * MT * 16, wg_m1 = wg_m0 + 4 * MT * 16;
  const bool need_x = wg_m0 < Fp;
  const bool need_qp = wg_m1 > Fp;
  const int sr = tid >> 3, sc = tid & 7;   // staging role: row sr, columns sc + 8k

  int fa[MT], fb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int f = (mt0 + m) * 16 + li;
    if (f < F) { const int ab = fab[f]; fa[m] = ab & 0xffff; fb[m] = ab >> 16; }
    else if (f >= Fp) { fa[m] = (f - Fp < K) ? f - Fp : Kp; fb[m] = 0; }
    else { fa[m] = D + 1; fb[m] = D + 1; }
  }
  double4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

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
  double rx[XK], rq[2 * NT], rp[2 * NT];
  auto fetch = [&](int buf) {
    const StRow ri = rinfo[buf * ST_RB + sr];
    if (need_x) {
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int c = sc + 8 * k;
        double v = 0.0;
        if (ri.orow >= 0) {
          if (c < D) v = obs[ri.orow * D + c];
          else if (c == D) v = 1.0;
        }
        rx[k] = v;
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * NT; ++k) {
      const int c = sc + 8 * k;
      rq[k] = (ri.qrow >= 0 && c < K) ? q[ri.qrow * K + c] : 0.0;
    }
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < 2 * NT; ++k) {
        const int c = sc + 8 * k;
        rp[k] = (ri.prow >= 0 && c < K) ? q[ri.prow * K + c] : 0.0;
      }
    }
  };
  auto commit = [&]() {
    if (need_x) {
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int c = sc + 8 * k;
        if (c < D + 2) xs[sr * DS + c] = rx[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * NT; ++k) qs[sr * QS + sc + 8 * k] = rq[k];
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < 2 * NT; ++k) qp[sr * QS + sc + 8 * k] = rp[k];
      if (sc == 0) qp[sr * QS + Kp] = 0.0;
    }
  };

  row_info(c0, 0);
  __syncthreads();
  fetch(0);
  for (int st = 0; st < nstage; ++st) {
    const int64_t s0 = c0 + (int64_t)st * ST_RB;
    __syncthreads();            // previous compute finished reading LDS
    commit();
    row_info(s0 + ST_RB, (st + 1) & 1);
    __syncthreads();
    if (st + 1 < nstage) fetch((st + 1) & 1);   // in flight during the MFMAs below
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
        else A = qp[r * QS + fa[m]];
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, Bv[n], acc[m][n], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = (mt0 + m) * 16 + lg + 4 * r;
      if (f < Ftot) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
          part[((size_t)blockIdx.x * Ftot + f) * Kp + n * 16 + li] = acc[m][n][r];
      }
    }
  }
}


int svihmm_launch_stats_mfma(const StatsLaunch* a, int pipelined) {
  const int D = a->D, K = a->K, Kp = a->Kp, Fp = a->Fp, F = a->F, Lm = a->Lm;
  const int Ftot = Fp + Kp;
  const int DS = (D + 2) | 1;
  const int mtiles = Ftot / 16;
  if (pipelined) {
    const int NT = Kp / 16;
    const size_t lds = ((size_t)ST_RB * DS + 2 * (size_t)ST_RB * (Kp + 1)) * 8 + 2 * ST_RB * sizeof(StRow);
    if (lds > 150 * 1024 || Kp > 64) return 2;
    dim3 grid((unsigned)a->nchunk, (mtiles + 4 * 5 - 1) / (4 * 5));
    const int xk = (D + 2 + 7) / 8;
#define ST2(NTV, XKV)                                                                          \
  do {                                                                                         \
    if (lds > 64 * 1024)                                                                       \
      (void)hipFuncSetAttribute((const void*)k_stats_mfma2<5, NTV, XKV>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
    hipLaunchKernelGGL((k_stats_mfma2<5, NTV, XKV>), grid, dim3(256), lds, a->stream, a->obs,  \
                       a->mask, a->starts, a->n, Lm, D, K, Fp, F, a->fab, a->q, a->rpc,        \
                       a->flags, a->Lq, a->off, a->part);                                      \
  } while (0)
#define ST2X(NTV) do { if (xk <= 2) ST2(NTV, 2); else if (xk <= 5) ST2(NTV, 5); else ST2(NTV, 9); } while (0)
    if (NT == 1) ST2X(1); else if (NT == 2) ST2X(2); else if (NT == 3) ST2X(3); else ST2X(4);
#undef ST2X
#undef ST2
  } else {
    const int ntile = Kp / 16;
    const int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
    const int MT = 3;
    const size_t lds = ((size_t)ST_RB * DS + (size_t)ST_RB * (16 * NT + 1) + (size_t)ST_RB * (Kp + 1)) * 8;
    if (lds > 150 * 1024) return 2;
    dim3 grid((unsigned)a->nchunk, (mtiles + 4 * MT - 1) / (4 * MT), ntile / NT);
#define ST_LAUNCH(NTV)                                                                         \
  do {                                                                                         \
    if (lds > 64 * 1024)                                                                       \
      (void)hipFuncSetAttribute((const void*)k_stats_mfma<3, NTV>,                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
    hipLaunchKernelGGL((k_stats_mfma<3, NTV>), grid, dim3(256), lds, a->stream, a->obs,        \
                       a->mask, a->starts, a->n, Lm, D, K, Kp, Fp, F, a->fab, a->q, a->rpc,    \
                       a->flags, a->Lq, a->off, a->part);                                      \
  } while (0)
    if (NT == 4) ST_LAUNCH(4); else if (NT == 2) ST_LAUNCH(2); else ST_LAUNCH(1);
#undef ST_LAUNCH
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
