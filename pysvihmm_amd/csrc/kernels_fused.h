// kernels_fused.h -- the minibatch E-step's sweeps and statistics as ONE launch (round 6; VERDICT r5 next #1a; DESIGN 4
// "The S = 64 iteration").
//
// A minibatch of S = 64 windows keeps 128 sweep waves busy for 257 dependent steps (~80 us) while 7/8 of the chip
// idles, and its statistics GEMM (~40 us, 206 one-per-CU workgroups) can only start when the sweeps are over -- it needs
// alpha_t AND beta_t of a row.  But row t has both once the forward sweep has passed t and the backward sweep Lm - 1 - t:
// the middle rows at HALF the sweep time, the rows next to a window's ends only at the very end.  This kernel puts both
// kinds of workgroup into one grid of 256-thread workgroups (every wave has its SIMD's whole register file):
//
//   workgroups [0, nsw)            sweep workgroups: the four waves run k_wave_linr's body (kernels_wave_linr.h) for
//                                  windows wpb (b >> 1) + w in direction b & 1 (K = 64 only; one direction per workgroup:
//                                  the two unrolled step loops together do not fit a CU's instruction cache).  A sweep
//                                  wave PUBLISHES its progress: once per block of 12 steps it adds one to the counter of
//                                  every band whose threshold it has passed -- two steps behind its real position, so that
//                                  the rows it names have retired through the wave's in-order memory counter: no fence;
//   workgroups [nsw, nsw + nst)    statistics workgroups (chunk c, feature group g): four waves x five feature tiles x all
//                                  four state tiles of the fp64 MFMA GEMM, rows taken in READINESS ORDER: stage s of every
//                                  chunk draws 32 rows from band s -- the rows of its windows whose order index (distance
//                                  from the window's middle) lies in [s Lb, (s + 1) Lb) -- and starts when band counter s
//                                  shows that all 2 B sweeps have passed the band's threshold.
//
// All workgroups are resident at once (one per CU: the host asks for > 80 KB of LDS and launches the kernel only when
// nsw + nst fits the device with CUs to spare for the loop's side streams) and the sweep workgroups have the lower
// indices, so they are placed first: a statistics workgroup never waits for a sweep that has no CU.  No cross-stream
// edge, no host round trip: the pipelining lives inside one launch and works the same under a profiler that serialises
// kernels.
//
// Coherence without fences: the sweeps store their rows with agent-scope atomic stores (written through), the statistics
// read them with agent-scope atomic loads.  (Measured alternatives: an acquire fence in every statistics wave halved the
// sweeps' speed, a release fence per band in the sweep costs 2 us of drained prefetch queue each.)
//
// Posteriors: the statistics kernels of rounds 1-5 formed q_t = ah_t bh_t 2^(hx + gx - zexp) / zmant with the window's
// normaliser Z -- which the forward sweep only knows at its END.  Here every row normalises itself,
// q_t[j] = ah_t[j] bh_t[j] / sum_j ah_t[j] bh_t[j]  (the sum IS Z for every t; the binary exponents cancel), a DPP sum
// over the eight lanes that stage a row.  Same quantity, rounded differently in the last place.  The local bound's
// sum_t log sum_j ah_t[j] is formed from the same staged rows (lbpart), which takes the row-sum ring and its flushes out
// of the sweep's chain.
//
// A statistics stage = 160 MFMAs per wave = 4.3 us of the fp64 matrix pipe at one wave per SIMD, 8.5 us with its loads and
// LDS commits; the readiness bands of a 257-row window are 7.7 us apart at five stages.  The stages are double-buffered:
// the rows of band s + 1 are requested as soon as its counter shows it open (looked at between k-steps), so the memory
// latency runs beside the matrix work.  What is left behind the sweeps is the last band's stage and the epilogue.
//
// EMW instantiation (off by default, tu_fused.hip: sweep_emission_ok): the statistics workgroups first compute the batch's
// emission tiles (kernels_emission_ks.h: emission_orbit_ks_rounds), in outside-in priority order and in rounds with one
// arrival counter each; the sweeps then read Eh / kexp / ll0 coherently and take a round's gate before they request rows
// of its priority levels.
#pragma once
#include "kernels_stats_layout.h"
#include "kernels_wave_linr.h"
#include "kernels_emission_ks.h"

#define PIPE_MAX_STAGES WLR_MAX_BANDS
#define PIPE_AS1(T) __attribute__((address_space(1))) T*
struct PipePlan {
  int nsw;                          // sweep workgroups (wpb windows of one direction each)
  int wpb;                          // active sweep waves per workgroup (4)
  int exp;                          // measurement only (SVIHMM_PIPE_EXP): 1 statistics workgroups leave at once, 2 they sleep
                                    // 120 us on the clock and leave, 3 they take their gates but skip loads and k-steps
  int nchunk, ngrp;                 // statistics chunks x feature groups of 20 tiles = statistics workgroups
  int NS, Lb;                       // stages per chunk = readiness bands; rows of a window per band
  unsigned tgt[PIPE_MAX_STAGES];    // band counter value that says "all 2 B sweeps are past this band"
  const int* ord;                   // [len]: inner row of readiness order o (host table: stable sort by need)
  // emission inside the launch (EMW instantiation): the 16-row tiles of the batch in outside-in priority order; statistics
  // workgroup sb computes tiles em_tiles[r nst + sb], r = 0 .. em_nround - 1, and adds one to the round's counter after each
  const int* em_tiles;
  int em_ntile, em_nround;
  int64_t* starts_copy;             // SVI loop: the window starts are read from the pinned slot; the device copy is left here
  WlrPub pub;
  unsigned long long* dbg;          // measurement only (SVIHMM_PIPE_DBG): wall-clock stamps, nullptr in normal runs --
                                    // [workgroup][16]: sweep workgroups (wave 0) begin / end; statistics workgroups begin,
                                    // then per stage the gate's opening and the end of its k-steps
};
struct PipeRow {
  long long ooff;   // obs element offset (row * D), -1: out of range or masked
  int qoff;         // element offset of the row in ah / bh, -1: out of range
  int poff;         // predecessor row's
  int pok;
  int pad_;
};

// ---- statistics workgroup ------------------------------------------------------------------------------------
// Four waves (one per SIMD, 256-thread workgroups: every wave of the kernel may use the SIMD's whole register file --
// the sweep body needs 274 registers, and under the 256 of a 512-thread workgroup it spilled into its step loop or lost
// half its prefetch depth).  Wave mg owns MT = 5 feature tiles x all four state tiles (Kp = 64): 160 accumulator
// registers, 160 MFMAs per stage on 14 LDS operand reads per k-step.  LDS: two buffers of { A tile rb[C][CC1]
// (column-major, CC1 = 33 row slots), q tile qs[32][80] }, the row records of all stages, flags.
#define PIPE_CC1 33
// sum over the eight lanes that stage one row (quad_perm xor 1, xor 2, row_half_mirror)
__device__ __forceinline__ double pipe_row8_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);
  v += dpp_mov_f64<0x4E>(v);
  v += dpp_mov_f64<0x141>(v);
  return v;
}
template <int XK, typename ST>
__device__ __forceinline__ void pipe_stats_body(
    double* __restrict__ smem, const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int B, int Lm, int Lq, int off, int D, int K, int Fp, int F,
    const int* __restrict__ fab, const ST* __restrict__ ah, const ST* __restrict__ bh, uint32_t flags,
    double* __restrict__ part, double* __restrict__ lbpart, const PipePlan& pl, const int* __restrict__ ord,
    const unsigned* bcnt, PIPE_AS1(unsigned long long) dbg0, const int chunk, const int fg) {
  const bool lb_here = fg == 0 && lbpart != nullptr;
  constexpr int MT = 5, NTW = 4, Kp = 64, QS = ST_QS(64), TPR = 8, QK = 8, CC = PIPE_CC1;
  const int ZERO = D + 1, ONE = ZERO + 1, QP0 = ZERO + 2;
  const int C = QP0 + Kp;
  const int TB = C * CC + ST_RB * QS;               // doubles of one tile buffer: A tile [C][CC] | q tile [32][QS]
  double* rb0 = smem;                               // buffer u at rb0 + u TB
  double* qs0 = rb0 + C * CC;
  PipeRow* rinfo = reinterpret_cast<PipeRow*>(smem + 2 * TB);      // [NS][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int mg = wave;
  const int Ftot = Fp + Kp, mtiles = Ftot / 16;
  const int mt0 = (fg * 4 + mg) * MT, nt0 = 0;
  const int wg_m0 = fg * 4 * MT * 16, wg_m1 = wg_m0 + 4 * MT * 16;
  const bool need_x = wg_m0 < Fp;
  const bool need_qp = wg_m1 > Fp;
  const int sr = tid / TPR, sc = tid % TPR;         // staging role: row sr, columns sc + 8 k
  const int psr = ST_SLOT(sr & 3) + (sr >> 2);      // row slot of row sr = 4 ks + lg
  const int NS = pl.NS, Lb = pl.Lb;

  // ---- row records of every stage, up front (nothing here depends on the sweeps): thread (s, r).  The chunk's 32 rows per
  //      stage lie in at most 33 windows, the same for every stage: their starts go through LDS (in the SVI loop `starts`
  //      is the pinned slot the host wrote: one read per window, not one per row record, crosses the bus)
  volatile int* lflag = reinterpret_cast<volatile int*>(rinfo + NS * 32);   // [s]: band s is open (written by thread 0)
  double* lred = reinterpret_cast<double*>(const_cast<int*>(lflag) + PIPE_MAX_STAGES + 2);   // [4] per-wave sums of the local bound's terms
  int64_t* wst = reinterpret_cast<int64_t*>(lred + 4);                       // [34]
  const int w0 = (chunk * 32) / Lb;
  if (tid < 34) wst[tid] = starts[w0 + tid < B ? w0 + tid : 0];
  __syncthreads();
  for (int i = tid; i < 32 * NS; i += 256) {
    const int s = i >> 5, rr = i & 31;
    const int y = chunk * 32 + rr;
    const int w = y / Lb, oo = s * Lb + (y - w * Lb);
    const bool ok = w < B && oo < Lm;
    const int t = ok ? ord[oo] : 0;
    const int64_t start = wst[ok ? w - w0 : 0];
    const int64_t orow = start + off + t;
    const uint8_t m = (ok && mask) ? mask[orow] : (uint8_t)0;
    const int qrow = (ok ? w : 0) * Lq + off + t;
    const bool pok = ok && (t > 0 || (flags & SVIHMM_TRANS_WRAP));
    PipeRow ri;
    ri.ooff = (ok && !m) ? orow * D : -1;
    ri.qoff = ok ? qrow * K : -1;
    ri.poff = (t > 0 ? qrow - 1 : qrow + Lm - 1) * K;
    ri.pok = pok ? 1 : 0;
    ri.pad_ = 0;
    rinfo[i] = ri;
  }
  // A-operand element index (k-step 0)
  int oa[MT], ob[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int f = (mt0 + m) * 16 + li;
    int fa = ZERO, fb = ZERO;
    if (f < F) { const int ab = fab[f]; fa = ab & 0xffff; fb = ab >> 16; }
    else if (f >= Fp && f - Fp < K) { fa = QP0 + (f - Fp); fb = ONE; }
    oa[m] = fa * CC + ST_SLOT(lg); ob[m] = fb * CC + ST_SLOT(lg);
  }
  const int obq = lg * QS + nt0 * 16 + li;
  double4_t acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[m][n] = (double4_t){0, 0, 0, 0};
  int xcc[XK], xwi[XK];
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int c = sc + TPR * k;
    xcc[k] = c < D ? c : D - 1;
    xwi[k] = (c <= D ? c : ZERO) * CC + psr;        // beyond the ones slot: rewrite ZERO with 0.0
  }
  const int qwi = sr * QS + sc;
  const int pwi = (QP0 + sc) * CC + psr;
  if (sc == 0) { rb0[ZERO * CC + psr] = 0.0; rb0[ONE * CC + psr] = 1.0; rb0[TB + ZERO * CC + psr] = 0.0; rb0[TB + ONE * CC + psr] = 1.0; }
  const ST* __restrict__ athr = ah + sc;
  const ST* __restrict__ bthr = bh + sc;

  // (stamps through a pointer declared global: a FLAT store may alias LDS, and every LDS read behind one would wait for it)
  PIPE_AS1(unsigned long long) dbg = dbg0 ? dbg0 + (size_t)blockIdx.x * 32 : nullptr;
  if (dbg && tid == 0) dbg[0] = wall_clock64();
  // ---- the stage loop, double-buffered.  Stage s = 8 k-steps (160 MFMAs per wave: 4.3 us of the fp64 matrix pipe) on the
  //      32 rows of band s in LDS buffer s & 1.  The rows of band s + 1 are requested as soon as that band is open --
  //      thread 0 looks at its counter between k-steps and raises a flag in LDS, every thread tests the flag at the same
  //      places and issues its loads itself -- so their ~2 us of memory latency run beside the matrix work; whoever has
  //      not seen the flag by the end of the k-steps waits for it there.  Then normalise + commit into the other buffer,
  //      one barrier, next stage.  (Measured first with one buffer and gate -> loads -> commit -> k-steps in sequence:
  //      8.4 us per stage against bands 7.7 us apart -- the statistics fell further behind with every stage and ended
  //      20 us after the sweeps.)
  if (tid < PIPE_MAX_STAGES) lflag[tid] = 0;
  double rx[XK], va[QK], vb[QK], pa[QK], pb[QK];
  bool okx = false, okq = false, okp = false;
  double lbp = 0.0;
  auto band_open = [&](int s) {
    return (int)(__hip_atomic_load(bcnt + 16 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - pl.tgt[s]) >= 0;
  };
  auto fetch = [&](int s) {
    // (no acquire fence: the sweeps store their rows with agent-scope atomic stores, these are the matching loads)
    const PipeRow ri = rinfo[s * 32 + sr];
    okx = ri.ooff >= 0; okq = ri.qoff >= 0; okp = ri.pok != 0;
    if (need_x) {
      const double* __restrict__ xo = obs + (okx ? ri.ooff : 0);
#pragma unroll
      for (int k = 0; k < XK; ++k) rx[k] = xo[xcc[k]];
    }
    {
      const int o = okq ? ri.qoff : 0;
#pragma unroll
      for (int k = 0; k < QK; ++k) {
        va[k] = (double)__hip_atomic_load(athr + o + TPR * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vb[k] = (double)__hip_atomic_load(bthr + o + TPR * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (need_qp) {
      const int o = okp ? ri.poff : 0;
#pragma unroll
      for (int k = 0; k < QK; ++k) {
        pa[k] = (double)__hip_atomic_load(athr + o + TPR * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pb[k] = (double)__hip_atomic_load(bthr + o + TPR * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  auto commit = [&](int u) {
    double* rbu = rb0 + u * TB;
    double* qsu = qs0 + u * TB;
    // self-normalised posteriors (columns >= K of a ragged model read a neighbour's entries: masked)
    double vq[QK], vp[QK], sq = 0.0, sp = 0.0, sa = 0.0;
#pragma unroll
    for (int k = 0; k < QK; ++k) {
      const bool live = sc + TPR * k < K;
      vq[k] = (okq && live) ? va[k] * vb[k] : 0.0;
      sq += vq[k];
      sa += (okq && live) ? va[k] : 0.0;
    }
    sq = pipe_row8_sum(sq);
    const double iq = sq > 0.0 ? 1.0 / sq : 0.0;
    if (lb_here) {                                   // the local bound's term of this row: log sum_j ah_t[j]
      sa = pipe_row8_sum(sa);
      if (okq && sc == 0) lbp += fast_log(sa);
    }
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < QK; ++k) {
        const bool live = sc + TPR * k < K;
        vp[k] = (okp && live) ? pa[k] * pb[k] : 0.0;
        sp += vp[k];
      }
      sp = pipe_row8_sum(sp);
    }
    const double ip = sp > 0.0 ? 1.0 / sp : 0.0;
    if (need_x) {
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int c = sc + TPR * k;
        const double v = c < D ? rx[k] : (c == D ? 1.0 : 0.0);
        rbu[xwi[k]] = okx ? v : 0.0;
      }
    }
#pragma unroll
    for (int k = 0; k < QK; ++k) qsu[qwi + TPR * k] = vq[k] * iq;
    if (need_qp) {
#pragma unroll
      for (int k = 0; k < QK; ++k) rbu[pwi + TPR * k * CC] = vp[k] * ip;
    }
  };
  // every thread: wait until band s is flagged open (thread 0 does the looking: bounded like the loop's other gates --
  // a count that never comes must not hang the queue; the statistics of such a launch are garbage, its sweeps poisoned
  // the step).  Polling discipline (measured): 208 workgroups polling one counter every 0.2 us saturate its memory
  // channel; the first band is polled every ~1.7 us, later ones every ~0.4 us.
  auto wait_band = [&](int s) {
    // (wave-uniform on purpose: with lane 0 polling in one branch and lanes 1..63 of the same wave spinning on the flag
    //  in the other, the wave never gets back to lane 0 -- all lanes of wave 0 read the counter, one request)
    if (wave == 0) {
      if (!lflag[s]) {
        const unsigned long long t0 = wall_clock64();
        unsigned n = 0;
        while (!band_open(s)) {
          if (s == 0) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(16);
          if ((++n & 1023u) == 0u && wall_clock64() - t0 > SVI_SYNC_TICKS) break;
        }
        if (lane == 0) lflag[s] = 1;
      }
    } else {
      while (!lflag[s]) __builtin_amdgcn_s_sleep(2);
    }
  };
  __syncthreads();                                   // row records and flags are in place
  wait_band(0);
  if (dbg && tid == 0) dbg[1] = wall_clock64();
  fetch(0);
  commit(0);
  __syncthreads();
  for (int s = 0; s < NS; ++s) {
    if (pl.exp == 3) {                                // (measurement: gates only)
      if (dbg && tid == 0) dbg[2 + 2 * s] = wall_clock64();
      if (s + 1 < NS) { wait_band(s + 1); if (dbg && tid == 0) dbg[3 + 2 * s] = wall_clock64(); }
      continue;
    }
    const bool more = s + 1 < NS;
    bool issued = false;
    unsigned seen = 0;                                // thread 0: the band counter as last read (asynchronously)
    const unsigned* cnt_next = bcnt + 16 * (more ? s + 1 : s);
    const unsigned tgt_next = pl.tgt[more ? s + 1 : s];
    // ---- 8 k-steps on buffer s & 1, the LDS reads of k-step ks + 1 issued before the MFMAs of k-step ks
    {
      const double* rbu = rb0 + (s & 1) * TB;
      const double* qs = qs0 + (s & 1) * TB + obq;
      double Bv[NTW], Ax[MT], Ay[MT];
#pragma unroll
      for (int n = 0; n < NTW; ++n) Bv[n] = qs[n * 16];
#pragma unroll
      for (int m = 0; m < MT; ++m) { Ax[m] = rbu[oa[m]]; Ay[m] = rbu[ob[m]]; }
#pragma unroll
      for (int ks = 0; ks < ST_RB / 4; ++ks) {
        // the next band: thread 0 reads its counter (the value is looked at one k-step later: no wait on the load),
        // everybody tests the flag
        // (twice per stage only: the flag is an LDS read, and waiting for it also waits for the operand reads in flight.
        //  Measured, round 6: three looks per stage -- a band that opens 0.2 us after the second look costs the stage
        //  its prefetch -- made the stages slower (k-steps of stage 1 end at 72.2 instead of 69.6 us); committing the
        //  prefetched rows inside the k-loop, in the shadow of its last MFMAs, changed nothing: for stages 1-3 the band
        //  opens during or after the stage's own k-steps, the fetch comes behind them)
        if (more && !issued && (ks == 1 || ks == 4)) {
          if (tid == 0 && !lflag[s + 1] && (int)(seen - tgt_next) >= 0) lflag[s + 1] = 1;
          if (lflag[s + 1]) { fetch(s + 1); issued = true; if (dbg && tid == 0) dbg[14 + 2 * s] = wall_clock64(); }
        }
        if (more && !issued && (ks == 0 || ks == 3) && tid == 0)
          seen = __hip_atomic_load(cnt_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double Bn[NTW], Axn[MT], Ayn[MT];
        if (ks + 1 < ST_RB / 4) {
#pragma unroll
          for (int n = 0; n < NTW; ++n) Bn[n] = qs[(ks + 1) * 4 * QS + n * 16];
#pragma unroll
          for (int m = 0; m < MT; ++m) { Axn[m] = rbu[oa[m] + ks + 1]; Ayn[m] = rbu[ob[m] + ks + 1]; }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const double A = Ax[m] * Ay[m];
#pragma unroll
          for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, Bv[n], acc[m][n], 0, 0, 0);
        }
        if (ks + 1 < ST_RB / 4) {
#pragma unroll
          for (int n = 0; n < NTW; ++n) Bv[n] = Bn[n];
#pragma unroll
          for (int m = 0; m < MT; ++m) { Ax[m] = Axn[m]; Ay[m] = Ayn[m]; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (dbg && tid == 0) dbg[2 + 2 * s] = wall_clock64();
    if (more) {
      if (!issued) { wait_band(s + 1); fetch(s + 1); if (dbg && tid == 0) dbg[14 + 2 * s] = wall_clock64(); }
      if (dbg && tid == 0) dbg[3 + 2 * s] = wall_clock64();
      commit((s + 1) & 1);
      __syncthreads();                               // buffer (s + 1) & 1 complete; everybody is through buffer s & 1
      if (dbg && tid == 0) dbg[15 + 2 * s] = wall_clock64();
    }
  }
  // ---- the chunk's share of sum_t log(sum_j ah_t[j]) (first feature group only: every row once)
  if (lb_here) {
    const double w = wave_sum_dpp(lbp);
    if (lane == 0) lred[wave] = w;
    __syncthreads();
    if (tid == 0) lbpart[chunk] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
  }
  // ---- partial sums of the chunk: part[chunk][f][k]
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = (mt0 + m) * 16 + lg + 4 * r;
      if (f < Ftot && (mt0 + m) < mtiles) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          part[((size_t)chunk * Ftot + f) * Kp + (nt0 + n) * 16 + li] = acc[m][n][r];
      }
    }
  }
}

// LDS of the kernel
inline size_t pipe_lds_bytes(int D, int NS) {
  // (statistics: tiles + row records + flag / reduction words; the sweep workgroups use no LDS since the row sums of
  //  the local bound moved to the statistics side)
  return 2 * ((size_t)(D + 3 + 64) * PIPE_CC1 + (size_t)ST_RB * ST_QS(64)) * 8 + (size_t)NS * 32 * sizeof(PipeRow) +
         (PIPE_MAX_STAGES + 2) * sizeof(int) + 4 * sizeof(double) + 34 * sizeof(int64_t) + 64;
}

// 256-thread workgroups: four waves, one per SIMD, each with the SIMD's whole register file (512).  Both roles are
// inlined into the kernel body -- the sweep's row arithmetic wants the kernel's pointer arguments where they are: in
// scalar registers, known to be global (as a non-inlined function it took them as flat pointers in vector registers:
// flat loads / stores with 64-bit vector address arithmetic in every step; with eight-wave workgroups, i.e. 256
// registers, the inlined roles spilled 100-400 registers).
// The statistics role alone is a separate (non-inlined) function: inlined beside the sweep bodies the allocator sees the
// union of all live ranges and spills 40-140 registers; as a function it gets its own allocation.  It finds the
// workgroup's dynamic LDS itself (a pointer argument would make every LDS access a flat one), and its global pointers are
// declared in the global address space (generic parameters made every load of the stage loop a flat one).
template <int XK, typename ST>
__device__ __attribute__((noinline)) void pipe_stats_role(
    PIPE_AS1(const double) obs, PIPE_AS1(const uint8_t) mask, PIPE_AS1(const int64_t) starts, int B, int Lm,
    int Lq, int off, int D, int K, int Fp, int F, PIPE_AS1(const int) fab, PIPE_AS1(const ST) ah,
    PIPE_AS1(const ST) bh, uint32_t flags, PIPE_AS1(double) part, PIPE_AS1(double) lbpart, PIPE_AS1(const int) ord,
    PIPE_AS1(const unsigned) bcnt, PIPE_AS1(unsigned long long) dbg, const PipePlan* pl, int chunk, int fg) {
  extern __shared__ double smem_role[];
  pipe_stats_body<XK, ST>(smem_role, (const double*)obs, (const uint8_t*)mask, (const int64_t*)starts, B, Lm, Lq, off, D, K, Fp, F,
                          (const int*)fab, (const ST*)ah, (const ST*)bh, flags, (double*)part, (double*)lbpart, *pl, (const int*)ord,
                          (const unsigned*)bcnt, dbg, chunk, fg);
}
// The emission role of a statistics workgroup: k_emission_orbit_ks's tile (kernels_emission.h: the same arithmetic, bit for
// bit) with its results stored coherently, one arrival per tile.  A function of its own for the same reason as the
// statistics role; its pointer parameters are declared in the global address space, which gives the inlined body its global -- not flat -- accesses back.
__device__ __attribute__((noinline)) void pipe_emission_role(
    PIPE_AS1(const double) obs_, PIPE_AS1(const uint8_t) mask_, PIPE_AS1(const int64_t) starts_, int64_t nrows, int Lm,
    int D, int K, PIPE_AS1(const double) orb_, uint32_t flags, PIPE_AS1(double) ll_, PIPE_AS1(double) kexp_,
    PIPE_AS1(double) ll0_, PIPE_AS1(const int) tiles_, PIPE_AS1(unsigned) cnt_, PIPE_AS1(unsigned long long) dbg0,
    const PipePlan* pl, int sb, int nst) {
  extern __shared__ double smem_em[];
  const double* __restrict__ obs = (const double*)obs_;
  const uint8_t* __restrict__ mask = (const uint8_t*)mask_;
  const int64_t* __restrict__ starts = (const int64_t*)starts_;
  const double* __restrict__ orb = (const double*)orb_;
  double* __restrict__ ll = (double*)ll_;
  double* __restrict__ kexp = (double*)kexp_;
  double* __restrict__ ll0 = (double*)ll0_;
  const int* __restrict__ tiles = (const int*)tiles_;
  unsigned* cnt = (unsigned*)cnt_;
  PIPE_AS1(unsigned long long) dbg = dbg0 ? dbg0 + (size_t)blockIdx.x * 32 : nullptr;
  if (dbg && threadIdx.x == 0) dbg[28] = wall_clock64();
  if (sb == 0 && pl->starts_copy) {
    PIPE_AS1(int64_t) sc = (PIPE_AS1(int64_t))pl->starts_copy;
    const int B = (int)(nrows / Lm);
    for (int i = threadIdx.x; i < B; i += 256) sc[i] = starts[i];
  }
  emission_orbit_ks_rounds<4>(smem_em, obs, mask, starts, nrows, Lm, D, K, orb, flags, ll, kexp, ll0, tiles, pl->em_ntile,
                              pl->em_nround, nst, sb, cnt, dbg);
  if (dbg && threadIdx.x == 0) dbg[29] = wall_clock64();
}
template <int XK, typename ST, bool EMW = false, int RN = 1, typename SE = ST>
__global__ __launch_bounds__(256) void k_sweep_stats(
    // sweeps (k_wave_linr's arguments; SE: storage type of Eh where it is not the messages' -- fp32 mode with fp64 messages)
    const SE* __restrict__ Eh, const double* __restrict__ kexp, const double* __restrict__ Aexp,
    const double* __restrict__ AexpT, const double* __restrict__ mod_init, const double* __restrict__ ll0,
    size_t l0stride, int Lq, int K, ST* __restrict__ ah, ST* __restrict__ bh, double* __restrict__ hx,
    double* __restrict__ gx, double* __restrict__ local_lb, double* __restrict__ logz, double2* __restrict__ zfac,
    SviSync sy,
    // statistics
    const double* __restrict__ obs, const uint8_t* __restrict__ mask, const int64_t* __restrict__ starts, int B,
    int Lm, int off, int D, int Fp, int F, const int* __restrict__ fab, uint32_t flags, double* __restrict__ part,
    PipePlan pl,
    // emission (EMW only: Eh, kexp, ll0 above are then this launch's own products)
    const double* __restrict__ orb, uint32_t eflags) {
  extern __shared__ double smem[];
  // Nobody else on this kernel's SIMDs: naming the last accumulation register makes the kernel's register block the whole
  // file (512 per lane), so no wave of another kernel fits beside a sweep or a statistics wave.  In the resident SVI loop
  // the ELBO kernels of the previous iteration and the next iteration's globals kernel run on side streams at this
  // time; spread over this kernel's CUs they took issue slots from the 257-step chain and the matrix work (trace:
  // k_svi_vlb 104 us instead of 20, this kernel 109 instead of 100) -- they belong on the CUs the launch leaves free.
  asm volatile("" ::: "a255");
  const int bx = blockIdx.x;
  if (bx < pl.nsw) {
    // ---- sweep workgroup
    // (SVI loop: the globals kernel of the side stream has arrived -- the gate ends in a workgroup barrier, so all
    //  four waves take it before those without a window leave)
    const bool go = svi_gate(sy);
    // (a workgroup's four waves run ONE direction -- even workgroups forward, odd backward, four windows each: the
    //  unrolled step loops of the two directions are 40 + 75 KB of code, more than a CU's instruction cache holds)
    const int wave = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int b = pl.wpb * (bx >> 1) + wave;
    const bool fwd = (bx & 1) == 0;
    if (wave >= pl.wpb || b >= B) return;
    if (!go) {
      // the loop is dead (device_helpers.h): the iteration is poisoned, and every band is published so that no
      // statistics workgroup waits for a sweep that will not run
      if (threadIdx.x == 0) svi_poison(sy);
      if (j == 0)
        for (int i = 0; i < pl.pub.nb; ++i) __hip_atomic_fetch_add(pl.pub.cnt + 16 * i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    WlrRing<double>& ring = *reinterpret_cast<WlrRing<double>*>(smem);     // (not touched by the publishing variant of the body)
    if (pl.dbg && threadIdx.x == 0) pl.dbg[(size_t)bx * 32] = wall_clock64();
    if (fwd) wave_linr_body<true, true, ST, double, true, EMW, RN, SE>(Eh, kexp, Aexp, mod_init, ll0, l0stride, Lq, K, ah, hx, local_lb, logz, zfac, ring, b, j, &pl.pub);
    else wave_linr_body<false, true, ST, double, true, EMW, RN, SE>(Eh, kexp, AexpT, mod_init, ll0, l0stride, Lq, K, bh, gx, local_lb, logz, zfac, ring, b, j, &pl.pub);
    if (pl.dbg && threadIdx.x == 0) pl.dbg[(size_t)bx * 32 + 1] = wall_clock64();
    return;
  }
  const int sb = bx - pl.nsw;
  if (pl.exp == 1) return;
  if (pl.exp == 2) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 12000ull) __builtin_amdgcn_s_sleep(64); return; }
  if constexpr (EMW) {
    if constexpr (sizeof(SE) == 8)
      pipe_emission_role((PIPE_AS1(const double))obs, (PIPE_AS1(const uint8_t))mask, (PIPE_AS1(const int64_t))starts, (int64_t)B * Lq,
                         Lq, D, K, (PIPE_AS1(const double))orb, eflags, (PIPE_AS1(double)) const_cast<double*>(reinterpret_cast<const double*>(Eh)),
                         (PIPE_AS1(double)) const_cast<double*>(kexp), (PIPE_AS1(double)) const_cast<double*>(ll0),
                         (PIPE_AS1(const int))pl.em_tiles, (PIPE_AS1(unsigned)) const_cast<unsigned*>(pl.pub.em_cnt),
                         (PIPE_AS1(unsigned long long))pl.dbg, &pl, sb,
                         pl.nchunk * pl.ngrp);
    __syncthreads();                                 // (the roles share the workgroup's LDS)
  }
  pipe_stats_role<XK, ST>((PIPE_AS1(const double))obs, (PIPE_AS1(const uint8_t))mask, (PIPE_AS1(const int64_t))starts, B, Lm, Lq, off, D,
                          K, Fp, F, (PIPE_AS1(const int))fab, (PIPE_AS1(const ST))ah, (PIPE_AS1(const ST))bh, flags,
                          (PIPE_AS1(double))part, (PIPE_AS1(double))(local_lb + B), (PIPE_AS1(const int))pl.ord,
                          (PIPE_AS1(const unsigned))pl.pub.cnt, (PIPE_AS1(unsigned long long))pl.dbg, &pl, sb / pl.ngrp, sb % pl.ngrp);
}
