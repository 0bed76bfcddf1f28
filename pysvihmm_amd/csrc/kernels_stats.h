// kernels_stats.h -- K4 expected sufficient statistics (VALU fallback + fp64 MFMA GEMMs), K5 deterministic finalize.
// Part of libsvihmm_hip.so; compiled in tu_stats.hip.
#pragma once

// ------------------------------------------------------------------------------------
//  K4b: statistics as an fp64 MFMA GEMM  out[Ftot x Kp] = Phi^T[Ftot x rows] * q[rows x Kp]
//       Per workgroup: 4 waves, each MT m-tiles (16 features) x NT n-tiles (16 states);
//       rows of the chunk staged through LDS in blocks of ST_RB.
//       grid (nchunk, ceil(Ftot/16 / (4*MT)), Kp/(16*NT)), block 256.
// ------------------------------------------------------------------------------------
#ifndef ST_RB
#define ST_RB 32
#endif
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
//  K4d: statistics GEMM, software-pipelined, VGPR-form accumulators (K <= 64 per state
//  group).  Same math as K4b.  Design points:
//   * fp64 MFMA with AGPR accumulators runs at ~63 % of the VGPR-form rate on gfx950
//     (tools/peak_probe.py: 49 vs 77.6 TF/s), so the accumulators must fit the 256
//     architected VGPRs: a workgroup is 4 m-groups x NSPLIT n-groups of waves, each wave
//     MT x NTW tiles (5 x 2 x 8 = 80 accumulator registers at K = 64);
//   * the next 32-row stage is fetched from HBM into registers while the current stage
//     runs on the matrix pipe (global -> reg early, reg -> LDS after the compute); LDS tiles
//     are double buffered with ONE barrier per stage; the two waves that share a SIMD stage
//     at opposite ends of the stage (role B first, role A in the middle), so one of them
//     always feeds the matrix pipe;
//   * row bookkeeping (obs row, q row, wrap predecessor, mask) is computed once per stage
//     by 32 lanes, three stages ahead, in phases so that its dependent global loads never
//     sit in front of the stage barrier;
//   * the 36 emission + 4 transition tiles of K=64, D=32 split into two balanced
//     workgroup passes, so q is read twice.
//  On gfx950 every VALU instruction a wave issues competes with the fp64 MFMAs of the SIMD
//  (tools/peak_probe.py: the times add); an earlier version (K4c) spent ~600 VALU
//  instructions per 80-MFMA stage on address arithmetic, 64-bit divisions and predicated
//  copies.  Here (~170 per stage):
//   * the A-operand tile is stored column-major, both LDS buffers interleaved, the 32 stage
//     rows permuted:  element (buffer u, row r, column c) at  c*67 + u*33 + ST_SLOT(r&3) + (r>>2),
//     so the operand of k-step ks for lane (li, lg) is  base(lane, m) + [u*33 + ks]  -- one
//     VGPR per (m-tile, factor) computed once per kernel, everything else an immediate, and
//     the 32 lanes of an LDS pass spread over all bank pairs (column stride 67 = 3 mod 32, the
//     slots of lg = 0 / 1 and of lg = 2 / 3 sixteen apart: ST_SLOT, round 6);
//   * row bookkeeping is 32-bit arithmetic relative to the chunk start (one unsigned
//     division per row instead of a 64-bit one); q rows become 32-bit element offsets
//     from a per-thread base pointer;
//   * staging threads copy with unconditional loads from clamped addresses and value
//     selects (no exec-mask branches, so waits stay counted);
//   * the stage loop is unrolled by two so that the LDS buffer is a compile-time choice.
//  grid (nchunk, ceil(Ftot/16 / (4*MT)), state groups), block 256*NSPLIT.
//  Host guarantees rows_per_chunk * max(K, Lq/Lm * K) < 2^31.
// ------------------------------------------------------------------------------------
struct StRow4 {
  long long ooff;   // obs element offset (row * D), -1: out of range or masked
  int qoff;         // q element offset relative to the chunk's first q row, -1: out of range
  int poff;         // predecessor q element offset (may be negative: see pok), validity in pok
  int pok, pad_;
  double sq, sp;    // LIN: posterior scale of the two rows
};
#include "kernels_stats_layout.h"
// TRONLY (K > 64): only the transition statistic sum_t q[t-1, pbase + i] q[t, kbase + j] of
// one (64 MT) x 64 block of (previous state, state) pairs: blockIdx.y = previous-state group,
// blockIdx.z = state group, the m-tiles are the 64 MT q[prev] columns (MT per wave), no obs
// columns in the tile (ZERO, ONE, then q[prev]) and no second A factor.
// CT: arithmetic type of the GEMM (LDS tiles, MFMA operands and accumulators): double =
// v_mfma_f64_16x16x4_f64, float = v_mfma_f32_16x16x4_f32 (the fp32 mode: twice the matrix rate,
// half the LDS traffic; a chunk's sums are accumulated in fp32 and leave as fp64 partials).
// ST: storage type of the scaled messages q = ah and bh (LIN) as the sweeps wrote them.
template <typename T> struct MF;
template <> struct MF<double> {
  typedef double4_t v4;
  static __device__ __forceinline__ v4 mma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lg, int r) { return lg + 4 * r; }    // C row of register r
};
typedef float float4_mf __attribute__((ext_vector_type(4)));
template <> struct MF<float> {
  typedef float4_mf v4;
  static __device__ __forceinline__ v4 mma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lg, int r) { return 4 * lg + r; }
};
// NB: LDS buffers of the staged tiles.  2: double buffering with one workgroup barrier per stage.
// 3 (round 3): NO barrier in the stage loop.  A stage's data are complete when every wave has
// committed its share; each wave counts its commit in an LDS counter (release) and a wave enters
// stage s once the counter shows all commits of stage s - 1 (acquire) -- a wait that normally finds
// the count already there, since the commits happen at the start / in the middle of the previous
// stage.  The third buffer makes the write side safe without any wait: stage s's commits go to
// buffer (s + 1) % 3, last read in stage s - 2, and a wave can only be in stage s after everyone
// has committed in stage s - 1, i.e. has left stage s - 2.  The row-info slot of stage s + 3 is
// committed by wave 0 BEFORE its data commit of the same stage, so the same counter covers it.
// (Knock-out measurement on the bench shape: without the barrier the kernel runs 1.41 -> 1.31 ms.)
template <int MT, int NTW, int NSPLIT, int XK, bool LIN, bool TRONLY = false, typename CT = double,
          typename ST = double, int NB = 2>
__global__ __launch_bounds__(256 * NSPLIT) void k_stats_mfma4(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Fp, int F,
    const int* __restrict__ fab, const ST* __restrict__ q, int64_t rows_per_chunk,
    uint32_t flags, int Lq, int off, double* __restrict__ part, int KpTot, int mt_limit,
    const ST* __restrict__ bh, const double* __restrict__ hx, const double* __restrict__ gx,
    const double2* __restrict__ zfac, double* __restrict__ qout) {
  // qout (LIN, wide models): the first feature group's workgroups also write the posteriors
  // q = ah bh scale they form while staging, for the transition-block launch that follows
  static_assert(ST_RB == 32, "row permutation assumes 32-row stages");
  static_assert(NB == 2 || NB == 3, "double or triple buffering");
  // column stride of the A-operand tile: NB buffers of 33 row slots + padding such that 16
  // consecutive columns fall on 16 different bank pairs (67 = 3 mod 32 doubles; 101 = 5 mod 32)
  constexpr int CC = NB == 2 ? ST_CC : 3 * ST_CS + 2;
  constexpr int NWV = 4 * NSPLIT;          // waves of the workgroup
  constexpr int NT = NTW * NSPLIT;
  constexpr int Kp = 16 * NT;
  // q tile row stride: the k-rows lg = 0, 1 (and 2, 3) of a B-operand read are served in one LDS pass of 32 lanes
  // and must fall on different banks: 16 words apart modulo the 32 (ST_QS)
  constexpr int QS = TRONLY ? ST_QS_TR(Kp, MT) : (NB == 3 ? ST_QS3(Kp, XK) : ST_QS(Kp));
  constexpr int TPR = 8 * NSPLIT;          // staging threads per row (block / 32)
  constexpr int QK = Kp / TPR;             // q columns per staging thread (exact)
  static_assert(QK * TPR == Kp, "staging split");
  extern __shared__ double smem[];
  // columns of the A-operand tile: [0,D) x | D: 1 (0 on masked rows) | D+1 ZERO | D+2 ONE | QP0+i: q[prev][i]
  constexpr int PG = TRONLY ? MT : 1;      // previous-state groups of Kp columns staged
  const int ZERO = TRONLY ? 0 : D + 1, ONE = ZERO + 1, QP0 = ZERO + 2;
  const int C = QP0 + Kp * PG;
  CT* rb0 = reinterpret_cast<CT*>(smem);   // [C][CC]
  CT* qs0 = rb0 + C * CC;                  // [NB][32][QS]
  StRow4* rinfo = reinterpret_cast<StRow4*>(     // [4][32], 8-byte aligned behind the tiles
      smem + (((size_t)(C * CC + NB * ST_RB * QS) * sizeof(CT) + 7) / 8));
  int* commits = reinterpret_cast<int*>(rinfo + 4 * ST_RB);   // NB == 3: commits counted so far
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int mg = wave & 3, ng = wave >> 2;
  const int Ftot = Fp + KpTot;
  const int kbase = blockIdx.z * Kp;
  const int pbase = TRONLY ? blockIdx.y * Kp * PG : 0;     // first previous state of this block
  const int mt0 = TRONLY ? mg * MT : (blockIdx.y * 4 + mg) * MT;
  const int nt0 = ng * NTW;
  const int wg_m0 = blockIdx.y * 4 * MT * 16, wg_m1 = wg_m0 + 4 * MT * 16;
  const bool need_x = !TRONLY && wg_m0 < Fp;
  const bool need_qp = TRONLY || (wg_m1 > Fp && mt_limit * 16 > Fp);
  const int sr = tid / TPR, sc = tid % TPR;   // staging role: row sr, columns sc + TPR*k
  const int psr = ST_SLOT(sr & 3) + (sr >> 2);   // row slot of row sr = 4 ks + lg (see ST_SLOT)

  // A-operand element index of buffer 0, k-step 0 (buffer u, k-step ks: + u*33 + ks)
  int oa[MT], ob[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int f = (mt0 + m) * 16 + li;
    int fa = ZERO, fb = ZERO;
    if (TRONLY) { if (f < Kp * PG && pbase + f < K) { fa = QP0 + f; fb = ONE; } }
    else if (f < F) { const int ab = fab[f]; fa = ab & 0xffff; fb = ab >> 16; }
    else if (f >= Fp && f - Fp < K && mt_limit * 16 > Fp) { fa = QP0 + (f - Fp); fb = ONE; }
    oa[m] = fa * CC + ST_SLOT(lg); ob[m] = fb * CC + ST_SLOT(lg);
  }
  const int obq = lg * QS + nt0 * 16 + li;   // B operand: qs[(4ks+lg)*QS + (nt0+n)*16 + li]
  typename MF<CT>::v4 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[m][n] = (typename MF<CT>::v4){0, 0, 0, 0};

  const int64_t c0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t c1 = imin64(nrows, c0 + rows_per_chunk);
  const int nrow = (int)(c1 - c0);
  const int nstage = (nrow + ST_RB - 1) / ST_RB;
  // chunk origin: window bw0, step t0, q row Q0 (all uniform)
  const int64_t bw0 = c0 / Lm;
  const unsigned t0 = (unsigned)(c0 - bw0 * Lm);
  const int64_t Q0 = bw0 * Lq + off;
  const ST* __restrict__ qthr = q + Q0 * K + kbase + sc;   // per-thread bases
  const ST* __restrict__ bthr = LIN ? bh + Q0 * K + kbase + sc : nullptr;
  double* __restrict__ qothr = (LIN && qout && blockIdx.y == 0) ? qout + Q0 * K + kbase + sc : nullptr;
  const ST* __restrict__ pthr = q + Q0 * K + pbase + sc;
  const ST* __restrict__ bpthr = LIN ? bh + Q0 * K + pbase + sc : nullptr;

  // ---- row bookkeeping, three stages ahead, in phases; threads 0..31
  unsigned ri_bwr = 0, ri_t = 0;
  int ri_qr = 0, ri_pr = 0, ri_pok = 0;
  int64_t ri_start = 0, ri_o = -1;
  uint8_t ri_m = 0;
  double2 ri_zf = make_double2(0.0, 0.0);
  double ri_hq = 0.0, ri_gq = 0.0, ri_hp = 0.0, ri_gp = 0.0;
  bool ri_ok = false;
  auto ri_phase1 = [&](int s0) {        // s0: first row of the stage relative to c0
    if (tid < ST_RB) {
      const int jrow = s0 + tid;
      ri_ok = jrow < nrow;
      const unsigned x = t0 + (unsigned)(ri_ok ? jrow : 0);
      ri_bwr = x / (unsigned)Lm;
      ri_t = x - ri_bwr * (unsigned)Lm;
      ri_qr = (int)(ri_bwr * (unsigned)Lq + ri_t);          // q row relative to Q0
      ri_pok = (ri_t > 0 || (flags & SVIHMM_TRANS_WRAP)) ? 1 : 0;
      ri_pr = ri_t > 0 ? ri_qr - 1 : ri_qr + Lm - 1;
      const int64_t bw = bw0 + ri_bwr;
      ri_start = starts[bw];
      if (LIN) {
        ri_zf = zfac[bw];
        ri_hq = hx[Q0 + ri_qr]; ri_gq = gx[Q0 + ri_qr];
        const int pr = ri_pok ? ri_pr : ri_qr;
        ri_hp = hx[Q0 + pr]; ri_gp = gx[Q0 + pr];
      }
    }
  };
  auto ri_phase2 = [&]() {
    if (tid < ST_RB) {
      ri_o = ri_start + off + ri_t;
      ri_m = mask ? mask[ri_o] : (uint8_t)0;
    }
  };
  auto ri_commit = [&](int buf) {
    if (tid < ST_RB) {
      StRow4 ri;
      ri.ooff = (ri_ok && !ri_m) ? ri_o * D : -1;
      ri.qoff = ri_ok ? ri_qr * K : -1;
      ri.poff = ri_pr * K;
      ri.pok = (ri_ok && ri_pok) ? 1 : 0;
      ri.pad_ = 0;
      ri.sq = 0.0; ri.sp = 0.0;
      if (LIN) {   // invalid rows: scale 0 (their loads come from a clamped, finite row)
        ri.sq = ri_ok ? ldexp(ri_zf.x, (int)(ri_hq + ri_gq - ri_zf.y)) : 0.0;
        ri.sp = ri.pok ? ldexp(ri_zf.x, (int)(ri_hp + ri_gp - ri_zf.y)) : 0.0;
      }
      rinfo[buf * ST_RB + tid] = ri;
    }
  };
  auto row_info = [&](int s0, int buf) { ri_phase1(s0); ri_phase2(); ri_commit(buf); };

  // ---- staging: unconditional loads from clamped addresses, selects at commit time
  int xcc[XK], xwi[XK];  // clamped obs column; LDS index (buffer 0) of the column this thread writes
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int c = sc + TPR * k;
    xcc[k] = c < D ? c : D - 1;
    xwi[k] = (c <= D ? c : ZERO) * CC + psr;   // beyond the ones slot: rewrite ZERO with 0.0
  }
  const int qwi = sr * QS + sc;                 // q tile element of this thread (column 0)
  const int pwi = (QP0 + sc) * CC + psr;     // q[prev] column of this thread (buffer 0)
  constexpr int PK = QK * PG;               // q[prev] columns per staging thread
  double rx[XK], rq[QK], rp[PK];
  double rq2[LIN ? QK : 1], rp2[LIN ? PK : 1], rsq = 0.0, rsp = 0.0;
  bool okx = false, okq = false, okp = false;
  int qo_row = 0;                            // q offset of the row fetched last (qout)
  auto fetch = [&](int buf) {
    const StRow4 ri = rinfo[buf * ST_RB + sr];
    okx = ri.ooff >= 0; okq = ri.qoff >= 0; okp = ri.pok != 0;
    if (LIN) { rsq = ri.sq; rsp = ri.sp; }
    if (need_x) {
      const double* __restrict__ xo = obs + (okx ? ri.ooff : 0);
#pragma unroll
      for (int k = 0; k < XK; ++k) rx[k] = xo[xcc[k]];
    }
    {
      const int o = okq ? ri.qoff : 0;
      qo_row = o;
#pragma unroll
      for (int k = 0; k < QK; ++k) rq[k] = qthr[o + TPR * k];
      if (LIN) {
#pragma unroll
        for (int k = 0; k < QK; ++k) rq2[k] = bthr[o + TPR * k];
      }
    }
    if (need_qp) {
      const int o = okp ? ri.poff : 0;
#pragma unroll
      for (int k = 0; k < PK; ++k) rp[k] = pthr[o + TPR * k];
      if (LIN) {
#pragma unroll
        for (int k = 0; k < PK; ++k) rp2[k] = bpthr[o + TPR * k];
      }
    }
  };
  auto commit = [&](auto bufc) {
    constexpr int U = decltype(bufc)::value;
    if (need_x) {
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int c = sc + TPR * k;
        const double v = c < D ? rx[k] : (c == D ? 1.0 : 0.0);
        rb0[xwi[k] + U * ST_CS] = (CT)(okx ? v : 0.0);
      }
    }
    // Padded state columns (k >= K) are not masked: they read finite neighbours (the
    // buffers carry slack past the last row) and only feed output columns / transition
    // features that nothing reads (k_finalize drops k >= K, features >= Fp + K are (0, 0)).
#pragma unroll
    for (int k = 0; k < QK; ++k) {
      const double v = LIN ? (rq[k] * rq2[k]) * rsq : (okq ? rq[k] : 0.0);
      qs0[U * ST_RB * QS + qwi + TPR * k] = (CT)v;
      if (LIN && qothr && okq && kbase + sc + TPR * k < K) qothr[qo_row + TPR * k] = v;
    }
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < PK; ++k) {
        const double v = LIN ? (rp[k] * rp2[k]) * rsp : (okp ? rp[k] : 0.0);
        rb0[pwi + U * ST_CS + TPR * k * CC] = (CT)v;
      }
    }
  };
  // constant columns of both buffers
  if (sc == 0) {
#pragma unroll
    for (int u = 0; u < NB; ++u) { rb0[ZERO * CC + u * ST_CS + psr] = (CT)0; rb0[ONE * CC + u * ST_CS + psr] = (CT)1; }
  }
  if (NB == 3 && tid == 0) *commits = 0;
  const bool roleB = wave >= 4;   // the second wave of each SIMD
  row_info(0, 0);
  row_info(ST_RB, 1);
  row_info(2 * ST_RB, 2);
  __syncthreads();
  fetch(0);
  commit(std::integral_constant<int, 0>{});
  if (nstage > 1) fetch(1);
  __syncthreads();
  // one 32-row stage on LDS buffer CUR: 8 k-steps, software pipelined by hand (the LDS
  // reads of k-step ks+1 are issued before the MFMAs of k-step ks)
  // NB == 3: count this wave's commit of the next stage's data (after the LDS writes of all its lanes)
  auto signal = [&]() {
    if (NB == 3) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(commits, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto stage = [&](const int st, auto curc) {
    constexpr int CUR = decltype(curc)::value;
    constexpr int NXT = NB == 2 ? 1 - CUR : (CUR + 1) % 3;
    constexpr int UO = CUR * ST_CS;
    if (NB == 3 && st > 0) {      // this stage's data: every wave's commit of the previous stage
      const int need = NWV * st;
      while (__hip_atomic_load(commits, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
        __builtin_amdgcn_s_sleep(1);
    }
    if (NB == 2) ri_phase1((st + 3) * ST_RB);
    if (roleB) {
      if (st + 1 < nstage) { commit(std::integral_constant<int, NXT>{}); signal(); }
      if (st + 2 < nstage) fetch((st + 2) & 3);
    }
    const CT* qs = qs0 + CUR * ST_RB * QS + obq;
    CT Bv[NTW], Ax[MT], Ay[MT];
#pragma unroll
    for (int n = 0; n < NTW; ++n) Bv[n] = qs[n * 16];
#pragma unroll
    for (int m = 0; m < MT; ++m) { Ax[m] = rb0[oa[m] + UO]; Ay[m] = rb0[ob[m] + UO]; }
#pragma unroll
    for (int ks = 0; ks < ST_RB / 4; ++ks) {
      CT Bn[NTW], Axn[MT], Ayn[MT];
      if (ks + 1 < ST_RB / 4) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) Bn[n] = qs[(ks + 1) * 4 * QS + n * 16];
#pragma unroll
        for (int m = 0; m < MT; ++m) { Axn[m] = rb0[oa[m] + UO + ks + 1]; Ayn[m] = rb0[ob[m] + UO + ks + 1]; }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const CT A = TRONLY ? Ax[m] : Ax[m] * Ay[m];
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = MF<CT>::mma(A, Bv[n], acc[m][n]);
      }
      if (ks + 1 < ST_RB / 4) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) Bv[n] = Bn[n];
#pragma unroll
        for (int m = 0; m < MT; ++m) { Ax[m] = Axn[m]; Ay[m] = Ayn[m]; }
      }
      // the operand addresses are immediates: without a fence the scheduler hoists the
      // LDS reads of all eight k-steps to the top of the stage and spills
      __builtin_amdgcn_sched_barrier(0);
      // role A stages in the middle of its compute phase (role B did it before), so that
      // at the end of the stage both waves of a SIMD are still feeding the matrix pipe
      if (NB == 3) {
        // row info of stage st + 3: phase 1 ran in the middle of the previous stage, phase 2 here at
        // the start, the slot is written in the middle of this stage BEFORE wave 0 (role A) counts
        // its commit -- every load has half a stage to arrive, nobody waits on it
        if (ks == 0) ri_phase2();
        if (ks == ST_RB / 8 - 1) { ri_commit((st + 3) & 3); ri_phase1((st + 4) * ST_RB); }
      } else if (ks == ST_RB / 8 - 1) ri_phase2();
      if (ks == ST_RB / 8 - 1 && !roleB) {
        if (st + 1 < nstage) { commit(std::integral_constant<int, NXT>{}); signal(); }
        if (st + 2 < nstage) fetch((st + 2) & 3);
      }
    }
    if (NB == 2) {
      ri_commit((st + 3) & 3);
      __syncthreads();
    }
  };
  if (NB == 3) ri_phase1(3 * ST_RB);
  if (NB == 2) {
    int st = 0;
    for (; st + 1 < nstage; st += 2) {
      stage(st, std::integral_constant<int, 0>{});
      stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < nstage) stage(st, std::integral_constant<int, 0>{});
  } else {
    int st = 0;
    for (; st + 2 < nstage; st += 3) {
      stage(st, std::integral_constant<int, 0>{});
      stage(st + 1, std::integral_constant<int, 1>{});
      stage(st + 2, std::integral_constant<int, NB - 1>{});
    }
    if (st < nstage) stage(st, std::integral_constant<int, 0>{});
    if (st + 1 < nstage) stage(st + 1, std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fl = (mt0 + m) * 16 + MF<CT>::crow(lg, r);
      const int f = TRONLY ? Fp + pbase + fl : fl;
      if (TRONLY ? (fl < Kp * PG && pbase + fl < KpTot) : (f < Ftot && (mt0 + m) < mt_limit)) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          part[((size_t)blockIdx.x * Ftot + f) * KpTot + kbase + (nt0 + n) * 16 + li] = (double)acc[m][n][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
//  K5: deterministic reduction of the per-chunk partials + scatter into the packed layout
//      packed = [A_raw K*K | xbar K*D | neff K | S K*D*D | lb]
// ------------------------------------------------------------------------------------
// With `local_lb` the launch carries one extra workgroup that adds up the B per-window ELBO
// terms into the last packed slot (fixed order; saves a kernel on the E-step's critical path).
// diag: the diagonal family's layout [A_raw | xbar K*D | neff K | xsq K*D | lb] (features x_a^2 go
// to xsq[k][a]); otherwise the NIW layout with the full K x D x D second moments.
__global__ void k_finalize(const double* __restrict__ part, int nchunk, int D, int K,
                           int Kp, int Fp, int F, const int* __restrict__ fab,
                           double* __restrict__ packed, const double* __restrict__ local_lb, int B,
                           int diag) {
  const int Ftot = Fp + Kp;
  if (local_lb && blockIdx.x == gridDim.x - 1) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) acc += local_lb[b];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[(size_t)K * K + (size_t)K * D + K + (size_t)K * D * (diag ? 1 : D)] = red[0];
    return;
  }
  // 64 outputs per workgroup, wave q sums the chunks c = q (mod 4) of the whole groups of four (the tail goes
  // to wave 0): the four interleaved partial sums of rounds 1-4 -- same additions in the same order, bit for
  // bit -- with four times the loads in flight (the minibatch's ~100 chunks on 40 960 threads were
  // latency-bound: 12 us of the 64-window iteration)
  __shared__ double fsum[4][64];
  const int q = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = idx < (int64_t)Ftot * Kp;
  const int f = live ? (int)(idx / Kp) : 0, k = live ? (int)(idx - (int64_t)f * Kp) : 0;
  const size_t stride = (size_t)Ftot * Kp;
  {
    const double* pp = part + (size_t)f * Kp + k;
    const int n4 = nchunk & ~3;
    double sq = 0.0;
    if (live && k < K) {
      int c = q;
      for (; c + 12 < n4; c += 16) {
        const double v0 = pp[(size_t)c * stride], v1 = pp[(size_t)(c + 4) * stride];
        const double v2 = pp[(size_t)(c + 8) * stride], v3 = pp[(size_t)(c + 12) * stride];
        sq += v0; sq += v1; sq += v2; sq += v3;
      }
      for (; c < n4; c += 4) sq += pp[(size_t)c * stride];
      if (q == 0)
        for (c = n4; c < nchunk; ++c) sq += pp[(size_t)c * stride];
    }
    fsum[q][threadIdx.x & 63] = sq;
  }
  __syncthreads();
  if (q != 0 || !live || k >= K) return;
  const int ln = threadIdx.x & 63;
  const double s = (fsum[0][ln] + fsum[1][ln]) + (fsum[2][ln] + fsum[3][ln]);
  double* A = packed;
  double* xbar = A + (size_t)K * K;
  double* neff = xbar + (size_t)K * D;
  double* S = neff + K;
  if (f < F) {
    const int ab = fab[f];
    const int a = ab & 0xffff, b = ab >> 16;
    if (b < D) {  // a <= b < D
      if (diag) S[(size_t)k * D + a] = s;
      else {
        S[((size_t)k * D + a) * D + b] = s;
        S[((size_t)k * D + b) * D + a] = s;
      }
    } else if (a < D) {
      xbar[(size_t)k * D + a] = s;
    } else {
      neff[k] = s;
    }
  } else if (f >= Fp && f - Fp < K) {
    A[(size_t)(f - Fp) * K + k] = s;
  }
}

// ------------------------------------------------------------------------------------
//  K8b/K8c: Categorical symbol counts and their finalize (see kernels_emission.h, K8)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_stats_cat(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int K, int Kp, int V,
    const double* __restrict__ q, int64_t rows_per_chunk, int Lq, int off,
    double* __restrict__ partc) {
  extern __shared__ double tab[];             // [V][Kp]
  const int lane = threadIdx.x;
  for (int e = lane; e < V * Kp; e += 64) tab[e] = 0.0;
  __syncthreads();
  const int64_t c0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t c1 = imin64(nrows, c0 + rows_per_chunk);
  for (int64_t g = c0; g < c1; ++g) {
    const int64_t bw = g / Lm;
    const int64_t t = g - bw * Lm;
    const int64_t orow = starts[bw] + off + t;
    const double x = obs[orow];
    if ((mask && mask[orow]) || x != x) continue;      // uniform
    const int v = (int)x;
    if (v < 0 || v >= V) continue;
    const int64_t qrow = bw * Lq + off + t;
    for (int k = lane; k < K; k += 64) tab[v * Kp + k] += q[qrow * K + k];
  }
  __syncthreads();
  double* out = partc + (size_t)blockIdx.x * V * Kp;
  for (int e = lane; e < V * Kp; e += 64) out[e] = tab[e];
}

// packed (Categorical layout) = [A_raw K*K | counts K*V | lb]
__global__ void k_finalize_cat(const double* __restrict__ part, int nchunk, int KpT,
                               const double* __restrict__ partc, int nchunkc, int K, int Kp, int V,
                               double* __restrict__ packed) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nA = (int64_t)K * K, nC = (int64_t)K * V;
  if (idx < nA) {
    const int i = idx / K, k = idx - (int64_t)i * K;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[((size_t)c * KpT + i) * KpT + k];   // Ftot = KpT (Fp = 0)
    packed[idx] = s;
  } else if (idx < nA + nC) {
    const int64_t e = idx - nA;
    const int k = e / V, v = e - (int64_t)k * V;
    double s = 0.0;
    for (int c = 0; c < nchunkc; ++c) s += partc[((size_t)c * V + v) * Kp + k];
    packed[idx] = s;
  }
}


// ------------------------------------------------------------------------------------
//  K4f (round 4): the statistics GEMM of the fp32 mode on the bf16 matrix pipe.
//  Same contraction as K4d -- out[f][k] = sum_t phi_f(x_t) q[t][k], the transition rows
//  f >= Fp with phi = q[prev(t)][f - Fp] -- with BOTH operands carried as three bf16 terms
//  (hi + mid + lo, round-to-nearest each: the fp32 value) and the six products hi hi, hi mid,
//  mid hi, hi lo, lo hi, mid mid accumulated in the MFMA's fp32 accumulators: fp32 arithmetic at
//  16 x the rate of v_mfma_f32_16x16x4_f32, i.e. 2.7 x faster per product than K4d's fp32
//  instance (which ran at 0.60 of that slower pipe).  gfx950 has v_cvt_pk_bf16_f32, so a split
//  costs 11 VALU instructions per PAIR of values.
//  v_mfma_f32_32x32x16_bf16: M = 32 features, N = 32 states, k = 16 rows.
//    A lane 32 h + j: feature j of the tile, rows 8 h .. 8 h + 7 of the k-step;
//    B lane 32 h + n: state n, the same rows;   C lane 32 h + n, reg r: feature 8 (r / 4) + 4 h + r % 4.
//  Operands in LDS, transposed so that a lane's eight consecutive rows are 16 / 32 contiguous bytes:
//    xT  [D + 2][XRS] float    x columns, the ones column (0 on masked rows), a zero column;
//    qT  3 x [64][QRS] bf16    q = ah bh scale as its three terms, state-major;
//    pT  3 x [64][QRS] bf16    q of the predecessor row (wrap / none at window starts).
//  A feature tile's A terms are formed by the wave that owns it: x_a x_b in fp32, split; the
//  two transition tiles read pT as they are.  q is multiplied, scaled and split ONCE per
//  (row, state) by the staging threads (thread = state, eight consecutive rows: one 16-byte
//  LDS write per plane).  Workgroup = 8 waves, ONE per row chunk and CU: wave w owns feature
//  tiles w, w + 8, w + 16 (Ftot / 32 = 20 at K = 64, D = 32: three on waves 0..3, two on 4..7,
//  five per SIMD) x both state tiles -- every A operand feeds 12 MFMAs.  Stages of 64 rows,
//  double-buffered, one barrier per stage.  Requires K == 64 (Kp = 64), D <= 32, Fp % 32 == 0.
//  Partials leave as fp64 in K4d's layout part[chunk][f][k] (k_finalize unchanged).
// ------------------------------------------------------------------------------------
#define SB_ROWS 64
#define SB_XRS 68         // xT row stride in floats
#define SB_QRS 72         // plane row stride in bf16 (144 B: eight lanes' 16-byte reads cover all banks)
typedef __attribute__((ext_vector_type(8))) __bf16 sbf8_t;
typedef __attribute__((ext_vector_type(16))) float sf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 sbf2_t;
typedef __attribute__((ext_vector_type(2))) float sf2_t;
struct SbRow {
  long long ooff;     // obs element offset of the row, -1: invalid or masked
  int qoff, poff;     // element offsets of the row / its predecessor in ah, bh (clamped to 0)
  float sq, sp;       // their posterior scales (0: row invalid / no predecessor)
  int pad0, pad1;
};
// two fp32 values -> their three bf16 terms, packed (low half: a)
__device__ __forceinline__ void sb_split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  sf2_t v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sbf2_t));
  sf2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, sbf2_t));
  sf2_t s = {r.x - __uint_as_float(mid << 16), r.y - __uint_as_float(mid & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(s, sbf2_t));
}
__device__ __forceinline__ void sb_split8(const float (&v)[8], uint4& hi, uint4& mid, uint4& lo) {
  sb_split2(v[0], v[1], hi.x, mid.x, lo.x);
  sb_split2(v[2], v[3], hi.y, mid.y, lo.y);
  sb_split2(v[4], v[5], hi.z, mid.z, lo.z);
  sb_split2(v[6], v[7], hi.w, mid.w, lo.w);
}
__global__ __launch_bounds__(512) void k_stats_bf16x3(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Fp, int F,
    const int* __restrict__ fab, const float* __restrict__ ah, const float* __restrict__ bh,
    int64_t rows_per_chunk, uint32_t flags, int Lq, int off, double* __restrict__ part,
    const double* __restrict__ hx, const double* __restrict__ gx, const double2* __restrict__ zfac) {
  constexpr int Kp = 64, NPL = 6;
  extern __shared__ uint4 sb_smem[];
  const int XC = D + 2;                                      // x columns + ones + zero
  const size_t xbytes = ((size_t)XC * SB_XRS * 4 + 15) & ~(size_t)15;
  const size_t pbytes = (size_t)64 * SB_QRS * 2;             // one plane
  const size_t bufbytes = xbytes + NPL * pbytes;
  char* base = reinterpret_cast<char*>(sb_smem);
  SbRow* rinfo = reinterpret_cast<SbRow*>(base + 2 * bufbytes);     // [3][SB_ROWS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int Ftot = Fp + Kp, NMT = Ftot >> 5, FT = Fp >> 5;   // feature tiles; the first transition tile
  const int64_t c0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t c1 = imin64(nrows, c0 + rows_per_chunk);
  const int nrow = c1 > c0 ? (int)(c1 - c0) : 0;
  const int nstage = (nrow + SB_ROWS - 1) / SB_ROWS;
  const int64_t bw0 = c0 / Lm;
  const unsigned t0 = (unsigned)(c0 - bw0 * Lm);
  const int64_t Q0 = bw0 * Lq + off;
  const float* __restrict__ ah0 = ah + Q0 * K;
  const float* __restrict__ bh0 = bh + Q0 * K;

  // per-lane operand addresses of the wave's feature tiles (float index of row 0 in xT)
  int oa[3], ob[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int mt = wave + 8 * m;
    const int f = 32 * mt + j;
    int a = D + 1, b = D + 1;                                // padding feature: zero column
    if (mt < FT) { if (f < F) { const int ab = fab[f]; a = ab & 0xffff; b = ab >> 16; } }
    else { a = f - Fp; b = 0; }                              // transition tile: previous state a
    oa[m] = a; ob[m] = b;
  }
  sf16_t acc[3][2];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  auto row_info = [&](int st) {
    if (tid < SB_ROWS) {
      const int jrow = st * SB_ROWS + tid;
      const bool ok = jrow < nrow;
      const unsigned x = t0 + (unsigned)(ok ? jrow : 0);
      const unsigned bwr = x / (unsigned)Lm, t = x - bwr * (unsigned)Lm;
      const int qr = (int)(bwr * (unsigned)Lq + t);
      const bool pok = ok && (t > 0 || (flags & SVIHMM_TRANS_WRAP));
      const int pr = t > 0 ? qr - 1 : qr + Lm - 1;
      const int64_t bw = bw0 + bwr;
      const int64_t orow = starts[bw] + off + t;
      const bool msk = mask && mask[orow];
      const double2 zf = zfac[bw];
      SbRow ri;
      ri.ooff = (ok && !msk) ? orow * D : -1;
      ri.qoff = ok ? qr * K : 0;
      ri.poff = pok ? pr * K : 0;
      ri.sq = ok ? (float)ldexp(zf.x, (int)(hx[Q0 + qr] + gx[Q0 + qr] - zf.y)) : 0.0f;
      const int prc = pok ? pr : qr;
      ri.sp = pok ? (float)ldexp(zf.x, (int)(hx[Q0 + prc] + gx[Q0 + prc] - zf.y)) : 0.0f;
      ri.pad0 = 0; ri.pad1 = 0;
      rinfo[(st % 3) * SB_ROWS + tid] = ri;
    }
  };
  // staged data of one stage, in registers: thread = (state / x column `lane`, rows 8 wave .. 8 wave + 7)
  float ra[8], rb[8], pa[8], pb[8], rsq[8], rsp[8];
  double rx[8];
  bool xok[8];
  auto fetch = [&](int st) {
    const SbRow* ri = rinfo + (st % 3) * SB_ROWS + 8 * wave;
    const int xc = lane < D ? lane : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const SbRow r = ri[e];                                 // (uniform address: broadcast)
      ra[e] = ah0[r.qoff + lane]; rb[e] = bh0[r.qoff + lane];
      pa[e] = ah0[r.poff + lane]; pb[e] = bh0[r.poff + lane];
      rsq[e] = r.sq; rsp[e] = r.sp;
      xok[e] = r.ooff >= 0;
      rx[e] = obs[(xok[e] ? r.ooff : 0) + xc];
    }
  };
  auto commit = [&](int buf) {
    char* bb = base + (size_t)buf * bufbytes;
    float qv[8], pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { qv[e] = (ra[e] * rb[e]) * rsq[e]; pv[e] = (pa[e] * pb[e]) * rsp[e]; }
    uint4 t3[3];
    sb_split8(qv, t3[0], t3[1], t3[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      *reinterpret_cast<uint4*>(bb + xbytes + s * pbytes + ((size_t)lane * SB_QRS + 8 * wave) * 2) = t3[s];
    sb_split8(pv, t3[0], t3[1], t3[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      *reinterpret_cast<uint4*>(bb + xbytes + (3 + s) * pbytes + ((size_t)lane * SB_QRS + 8 * wave) * 2) = t3[s];
    if (lane < XC) {
      float xf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = lane < D ? (float)rx[e] : (lane == D ? 1.0f : 0.0f);
        xf[e] = xok[e] ? v : 0.0f;
      }
      float* xt = reinterpret_cast<float*>(bb) + lane * SB_XRS + 8 * wave;
      *reinterpret_cast<float4*>(xt) = make_float4(xf[0], xf[1], xf[2], xf[3]);
      *reinterpret_cast<float4*>(xt + 4) = make_float4(xf[4], xf[5], xf[6], xf[7]);
    }
  };
  constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};
  auto compute = [&](int buf) {
    const char* bb = base + (size_t)buf * bufbytes;
    const float* xT = reinterpret_cast<const float*>(bb);
    const char* qpl = bb + xbytes;
#pragma unroll 1
    for (int ks = 0; ks < SB_ROWS / 16; ++ks) {
      const int r0 = 16 * ks + 8 * hh;
      sbf8_t b3[2][3];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          b3[n][s] = __builtin_bit_cast(sbf8_t, *reinterpret_cast<const uint4*>(qpl + s * pbytes + ((size_t)(32 * n + j) * SB_QRS + r0) * 2));
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int mt = wave + 8 * m;
        if (mt >= NMT) continue;                             // (uniform per wave)
        sbf8_t a3[3];
        if (mt < FT) {
          const float4 xa0 = *reinterpret_cast<const float4*>(xT + oa[m] * SB_XRS + r0);
          const float4 xa1 = *reinterpret_cast<const float4*>(xT + oa[m] * SB_XRS + r0 + 4);
          const float4 xb0 = *reinterpret_cast<const float4*>(xT + ob[m] * SB_XRS + r0);
          const float4 xb1 = *reinterpret_cast<const float4*>(xT + ob[m] * SB_XRS + r0 + 4);
          const float p[8] = {xa0.x * xb0.x, xa0.y * xb0.y, xa0.z * xb0.z, xa0.w * xb0.w,
                              xa1.x * xb1.x, xa1.y * xb1.y, xa1.z * xb1.z, xa1.w * xb1.w};
          uint4 t3[3];
          sb_split8(p, t3[0], t3[1], t3[2]);
#pragma unroll
          for (int s = 0; s < 3; ++s) a3[s] = __builtin_bit_cast(sbf8_t, t3[s]);
        } else {
#pragma unroll
          for (int s = 0; s < 3; ++s)
            a3[s] = __builtin_bit_cast(sbf8_t, *reinterpret_cast<const uint4*>(qpl + (3 + s) * pbytes + ((size_t)oa[m] * SB_QRS + r0) * 2));
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int pi = 0; pi < 6; ++pi)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[TA[pi]], b3[n][TB[pi]], acc[m][n], 0, 0, 0);
      }
    }
  };
  if (nstage > 0) {
    row_info(0);
    if (nstage > 1) row_info(1);
    __syncthreads();
    fetch(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
      if (st + 1 < nstage) fetch(st + 1);
      if (st + 2 < nstage) row_info(st + 2);
      compute(st & 1);
      if (st + 1 < nstage) commit((st + 1) & 1);
      __syncthreads();
    }
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int mt = wave + 8 * m;
    if (mt >= NMT) continue;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * mt + 8 * (r >> 2) + 4 * hh + (r & 3);
        part[((size_t)blockIdx.x * Ftot + f) * Kp + 32 * n + j] = (double)acc[m][n][r];
      }
  }
}

// ------------------------------------------------------------------------------------
//  K4g (round 5): the same GEMM for the shapes K4f does not take -- wide models (K > 64) and
//  D > 32 (more than 22 feature tiles).  A workgroup is (row chunk, feature group gy, state group gz
//  of 64): it stages q for ITS 64 states, owns up to 22 feature tiles of 32 and -- when gy names a
//  group of previous states -- the two transition tiles (previous states 64 gy .. 64 gy + 63) x its 64
//  states, for which it stages q of the predecessor rows for THAT group.  Tile l = wave + 8 m of the
//  workgroup's list (features first).  Everything else as K4f: three bf16 terms per operand, six
//  products in fp32 accumulators, 64-row stages, double-buffered.  ah / bh rows have Kfull entries;
//  partials part[chunk][Fp + Kpf][Kpf] (k_finalize's layout for wide models).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_stats_bf16x3w(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kpf, int Fp, int F,
    const int* __restrict__ fab, const float* __restrict__ ah, const float* __restrict__ bh,
    int64_t rows_per_chunk, uint32_t flags, int Lq, int off, double* __restrict__ part,
    const double* __restrict__ hx, const double* __restrict__ gx, const double2* __restrict__ zfac, int TPG) {
  constexpr int NPL = 6;
  extern __shared__ uint4 sb_smem[];
  const int XC = D + 2;                                      // x columns + ones + zero
  const size_t xbytes = ((size_t)XC * SB_XRS * 4 + 15) & ~(size_t)15;
  const size_t pbytes = (size_t)64 * SB_QRS * 2;             // one plane
  const size_t bufbytes = xbytes + NPL * pbytes;
  char* base = reinterpret_cast<char*>(sb_smem);
  SbRow* rinfo = reinterpret_cast<SbRow*>(base + 2 * bufbytes);     // [3][SB_ROWS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int gy = blockIdx.y, gz = blockIdx.z;
  const int FT = (Fp + 31) >> 5;                             // (Fp is a multiple of 16: the last tile may be half padding)
  const int ft0 = gy * TPG;
  const int nft = ft0 >= FT ? 0 : (FT - ft0 < TPG ? FT - ft0 : TPG);    // feature tiles of this workgroup
  const bool has_tr = 64 * gy < Kpf;                                    // ... and its two transition tiles
  const int ntile = nft + (has_tr ? 2 : 0);
  const int FtotF = Fp + Kpf;
  const int ks0 = 64 * gz, kp0 = 64 * gy;
  const int64_t c0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t c1 = imin64(nrows, c0 + rows_per_chunk);
  const int nrow = c1 > c0 ? (int)(c1 - c0) : 0;
  const int nstage = (nrow + SB_ROWS - 1) / SB_ROWS;
  const int64_t bw0 = c0 / Lm;
  const unsigned t0 = (unsigned)(c0 - bw0 * Lm);
  const int64_t Q0 = bw0 * Lq + off;
  const float* __restrict__ ah0 = ah + Q0 * K;
  const float* __restrict__ bh0 = bh + Q0 * K;
  // the lane's state of the staged groups (clamped; states beyond K contribute zero)
  const bool sq_ok = ks0 + lane < K, sp_ok = has_tr && kp0 + lane < K;
  const int sq_c = sq_ok ? ks0 + lane : 0, sp_c = sp_ok ? kp0 + lane : 0;

  int oa[3], ob[3], tkind[3];                                // tkind: 0 idle, 1 feature, 2 transition
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int l = wave + 8 * m;
    int a = D + 1, b = D + 1;
    tkind[m] = l < nft ? 1 : (l < ntile ? 2 : 0);
    if (tkind[m] == 1) {
      const int f = 32 * (ft0 + l) + j;
      if (f < F) { const int ab = fab[f]; a = ab & 0xffff; b = ab >> 16; }
    } else if (tkind[m] == 2) { a = 32 * (l - nft) + j; b = 0; }   // previous state a of the staged group
    oa[m] = a; ob[m] = b;
  }
  sf16_t acc[3][2];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  auto row_info = [&](int st) {
    if (tid < SB_ROWS) {
      const int jrow = st * SB_ROWS + tid;
      const bool ok = jrow < nrow;
      const unsigned x = t0 + (unsigned)(ok ? jrow : 0);
      const unsigned bwr = x / (unsigned)Lm, t = x - bwr * (unsigned)Lm;
      const int qr = (int)(bwr * (unsigned)Lq + t);
      const bool pok = ok && (t > 0 || (flags & SVIHMM_TRANS_WRAP));
      const int pr = t > 0 ? qr - 1 : qr + Lm - 1;
      const int64_t bw = bw0 + bwr;
      const int64_t orow = starts[bw] + off + t;
      const bool msk = mask && mask[orow];
      const double2 zf = zfac[bw];
      SbRow ri;
      ri.ooff = (ok && !msk) ? orow * D : -1;
      ri.qoff = ok ? qr * K : 0;
      ri.poff = pok ? pr * K : 0;
      ri.sq = ok ? (float)ldexp(zf.x, (int)(hx[Q0 + qr] + gx[Q0 + qr] - zf.y)) : 0.0f;
      const int prc = pok ? pr : qr;
      ri.sp = pok ? (float)ldexp(zf.x, (int)(hx[Q0 + prc] + gx[Q0 + prc] - zf.y)) : 0.0f;
      ri.pad0 = 0; ri.pad1 = 0;
      rinfo[(st % 3) * SB_ROWS + tid] = ri;
    }
  };
  float ra[8], rb[8], pa[8], pb[8], rsq[8], rsp[8];
  double rx[8];
  bool xok[8];
  auto fetch = [&](int st) {
    const SbRow* ri = rinfo + (st % 3) * SB_ROWS + 8 * wave;
    const int xc = lane < D ? lane : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const SbRow r = ri[e];                                 // (uniform address: broadcast)
      ra[e] = ah0[r.qoff + sq_c]; rb[e] = bh0[r.qoff + sq_c];
      if (has_tr) { pa[e] = ah0[r.poff + sp_c]; pb[e] = bh0[r.poff + sp_c]; }
      rsq[e] = sq_ok ? r.sq : 0.0f; rsp[e] = sp_ok ? r.sp : 0.0f;
      xok[e] = r.ooff >= 0;
      rx[e] = obs[(xok[e] ? r.ooff : 0) + xc];
    }
  };
  auto commit = [&](int buf) {
    char* bb = base + (size_t)buf * bufbytes;
    float qv[8], pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = (ra[e] * rb[e]) * rsq[e];
    uint4 t3[3];
    sb_split8(qv, t3[0], t3[1], t3[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      *reinterpret_cast<uint4*>(bb + xbytes + s * pbytes + ((size_t)lane * SB_QRS + 8 * wave) * 2) = t3[s];
    if (has_tr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = (pa[e] * pb[e]) * rsp[e];
      sb_split8(pv, t3[0], t3[1], t3[2]);
#pragma unroll
      for (int s = 0; s < 3; ++s)
        *reinterpret_cast<uint4*>(bb + xbytes + (3 + s) * pbytes + ((size_t)lane * SB_QRS + 8 * wave) * 2) = t3[s];
    }
    {
      float xf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = lane < D ? (float)rx[e] : (lane == D ? 1.0f : 0.0f);
        xf[e] = xok[e] ? v : 0.0f;
      }
      if (lane < XC) {
        float* xt = reinterpret_cast<float*>(bb) + lane * SB_XRS + 8 * wave;
        *reinterpret_cast<float4*>(xt) = make_float4(xf[0], xf[1], xf[2], xf[3]);
        *reinterpret_cast<float4*>(xt + 4) = make_float4(xf[4], xf[5], xf[6], xf[7]);
      }
      // columns 64, 65 (D = 63, 64): the ones column (0 on masked rows) and the zero column
      const int c2 = lane + 64;
      if (c2 < XC) {
        float* xt = reinterpret_cast<float*>(bb) + c2 * SB_XRS + 8 * wave;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (c2 == D && xok[e]) ? 1.0f : 0.0f;
        *reinterpret_cast<float4*>(xt) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(xt + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  };
  constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};
  auto compute = [&](int buf) {
    const char* bb = base + (size_t)buf * bufbytes;
    const float* xT = reinterpret_cast<const float*>(bb);
    const char* qpl = bb + xbytes;
#pragma unroll 1
    for (int ks = 0; ks < SB_ROWS / 16; ++ks) {
      const int r0 = 16 * ks + 8 * hh;
      sbf8_t b3[2][3];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          b3[n][s] = __builtin_bit_cast(sbf8_t, *reinterpret_cast<const uint4*>(qpl + s * pbytes + ((size_t)(32 * n + j) * SB_QRS + r0) * 2));
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (tkind[m] == 0) continue;                         // (uniform per wave)
        sbf8_t a3[3];
        if (tkind[m] == 1) {
          const float4 xa0 = *reinterpret_cast<const float4*>(xT + oa[m] * SB_XRS + r0);
          const float4 xa1 = *reinterpret_cast<const float4*>(xT + oa[m] * SB_XRS + r0 + 4);
          const float4 xb0 = *reinterpret_cast<const float4*>(xT + ob[m] * SB_XRS + r0);
          const float4 xb1 = *reinterpret_cast<const float4*>(xT + ob[m] * SB_XRS + r0 + 4);
          const float p[8] = {xa0.x * xb0.x, xa0.y * xb0.y, xa0.z * xb0.z, xa0.w * xb0.w,
                              xa1.x * xb1.x, xa1.y * xb1.y, xa1.z * xb1.z, xa1.w * xb1.w};
          uint4 t3[3];
          sb_split8(p, t3[0], t3[1], t3[2]);
#pragma unroll
          for (int s = 0; s < 3; ++s) a3[s] = __builtin_bit_cast(sbf8_t, t3[s]);
        } else {
#pragma unroll
          for (int s = 0; s < 3; ++s)
            a3[s] = __builtin_bit_cast(sbf8_t, *reinterpret_cast<const uint4*>(qpl + (3 + s) * pbytes + ((size_t)oa[m] * SB_QRS + r0) * 2));
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int pi = 0; pi < 6; ++pi)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[TA[pi]], b3[n][TB[pi]], acc[m][n], 0, 0, 0);
      }
    }
  };
  if (nstage > 0 && ntile > 0) {
    row_info(0);
    if (nstage > 1) row_info(1);
    __syncthreads();
    fetch(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
      if (st + 1 < nstage) fetch(st + 1);
      if (st + 2 < nstage) row_info(st + 2);
      compute(st & 1);
      if (st + 1 < nstage) commit((st + 1) & 1);
      __syncthreads();
    }
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (tkind[m] == 0) continue;
    const int l = wave + 8 * m;
    const int fbase = tkind[m] == 1 ? 32 * (ft0 + l) : Fp + kp0 + 32 * (l - nft);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fbase + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (tkind[m] == 1 && f >= Fp) continue;              // padding rows of the last feature tile: the transition rows start there
        part[((size_t)blockIdx.x * FtotF + f) * Kpf + ks0 + 32 * n + j] = (double)acc[m][n][r];
      }
  }
}
