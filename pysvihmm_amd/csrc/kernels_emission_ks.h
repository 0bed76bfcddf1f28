// kernels_emission_ks.h -- the 16-row tile of the minibatch emission kernel K1c' (k_emission_orbit_ks, described in
// kernels_emission.h) as a device function: the stand-alone kernel (tu_emission.hip) and the statistics workgroups of the
// fused E-step kernel (kernels_fused.h, round 6) run the same body.  Device templates only -- safe to include from several
// translation units.
#pragma once
#include "device_helpers.h"
#ifndef EMO_KO
#define EMO_KO 0      // measurement knock-outs (tools/probe/emo_probe.hip): 1 no k-steps, 2 no theta loads, 4 no exp
#endif
// COH (round 6): the tile's results are stored with agent-scope atomic stores -- written through to where every CU reads
// them coherently -- for consumers inside the SAME launch (the sweep workgroups of the fused E-step kernel,
// kernels_fused.h); `tile` = which 16 rows (the stand-alone kernel: blockIdx.x).
template <int NT, bool COH>
__device__ __forceinline__ void emission_orbit_ks_body(
    double* __restrict__ smem, const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const double* __restrict__ orb, uint32_t flags, double* __restrict__ ll,
    double* __restrict__ kexp, double* __restrict__ ll0, const int tile) {
  constexpr int KP = 16 * NT;
  const int N = D + 1, c = D >> 2, nd = (D >> 1) + 1, nleft = (nd + 3) >> 2;
  const int LEN = D + (D >> 1) + 1;                 // slots -1 .. 3D/2 - 1 (odd count)
  // [0, R0) doubles: the rows (16 x LEN), later the partial sums (3 NT 256); then the small arrays
  double* xs = smem;
  double* red = smem;                               // [dest tile][source slot 0..2][r][lane]
  const int R0 = 3 * NT * 256 > 16 * LEN + 1 ? 3 * NT * 256 : ((16 * LEN + 1) & ~1);
  double* mxs = smem + R0;                          // [NT][16] per-tile row maxima
  long long* rowoff = (long long*)(mxs + 16 * NT);
  unsigned char* bad_s = (unsigned char*)(rowoff + 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t g0 = (int64_t)tile * 16;
  if (tid < 16) {
    const int r = tid;
    const int64_t bw0 = g0 / Lm;
    const unsigned t0 = (unsigned)(g0 - bw0 * Lm);
    const bool valid = g0 + r < nrows;
    const unsigned x = t0 + (unsigned)(valid ? r : 0);
    const unsigned bwr = x / (unsigned)Lm;
    const int64_t orow = starts[bw0 + bwr] + (x - bwr * (unsigned)Lm);
    unsigned char bd = 0;
    if (valid && (flags & SVIHMM_MASK_AS_NAN) && mask) bd = mask[orow] != 0;
    rowoff[r] = valid ? orow * D : -1;
    bad_s[r] = bd | ((valid && x == bwr * (unsigned)Lm) ? 2 : 0);   // bit 1: step 0 of its window
    xs[r * LEN] = 1.0;
    xs[r * LEN + D + 1] = 1.0;
  }
  __syncthreads();
  {
    const int sh = 32 - __builtin_clz((unsigned)(D - 1));
    for (int e = tid; e < (16 << sh); e += 256) {
      const int r = e >> sh, i = e & ((1 << sh) - 1);
      const long long o = rowoff[r];
      if (i < D) {
        double v = o >= 0 ? obs[o + i] : 0.0;
        if (v != v) { bad_s[r] |= 1; v = 0.0; }     // (every writer ORs the same bit)
        xs[r * LEN + i + 1] = v;
        if (i + N <= LEN - 2) xs[r * LEN + i + N + 1] = v;
      }
    }
  }
  __syncthreads();

  const int li = lane & 15, lg = lane >> 4;
  double4_t acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const double* rowp = xs + li * LEN + 1;                      // slot 0 of this lane's row
  const double* pa0 = rowp + c * lg;
  const double* p2 = rowp + lg - 1;
  const double* tl = orb + (unsigned)(lg * KP + li * NT);      // lane part of the theta address
  // this wave's k-steps: [s0, s1) of the schedule's nmain = c nd orbit steps followed by nleft leftover steps
  const int nmain = c * nd, S = nmain + nleft;
#if EMO_KO & 1
  const int s0 = 0, s1 = 0;
#else
  const int s0 = (S * wave) >> 2, s1 = (S * (wave + 1)) >> 2;
#endif
  auto loadB = [&](int s, double (&Bv)[NT]) {
    const double* trow = tl + (size_t)(s < S ? s : S - 1) * (4 * KP);
#if EMO_KO & 2
#pragma unroll
    for (int n = 0; n < NT; ++n) Bv[n] = (double)(s + n) * 1e-3;
    (void)trow;
#else
    if constexpr (NT % 2 == 0) {
#pragma unroll
      for (int n = 0; n < NT; n += 2) {
        const double2 t = *reinterpret_cast<const double2*>(trow + n);
        Bv[n] = t.x; Bv[n + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) Bv[n] = trow[n];
    }
#endif
  };
  int sd = s0 / c, sa = s0 - sd * c;                 // (delta, a0) of the next main step
  // blocks of four k-steps: the block's eight LDS operands first, then its 4 NT MFMAs; the B operands of block
  // i + 1 are requested before the MFMAs of block i (two register sets, loop unrolled by two blocks).
  // Round 6: the two LDS operand addresses of a k-step are SELECTED (main step / leftover step), the (uniform) position
  // in the feature schedule is stepped with scalar selects -- the branch per k-step this loop had cost a lone wave 7.1
  // against 5.0 us for its 36 k-steps (kernels_fused.h's emission role, same loop) and here it shared the SIMD's issue
  // port with three other waves' MFMAs.  Same operands, same order: bit-identical.
  const double* pxl = rowp + (N - 1);
  auto block = [&](int s, const double (&Bv)[4][NT]) {
    double xa[4], xb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool mainstep = s + u < nmain;
      int jl = s + u - nmain;
      jl = jl < 0 ? 0 : (jl < nleft ? jl : nleft - 1);
      const double* pa = mainstep ? pa0 + sa : pxl;
      const double* pb = mainstep ? pa0 + sa + sd : p2 + 4 * jl;
      xa[u] = *pa; xb[u] = *pb;
      const bool wrap = sa + 1 == c;
      sa = wrap ? 0 : sa + 1;
      sd += wrap ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double A = (s + u < s1) ? xa[u] * xb[u] : 0.0;        // (past the wave's range: adds B x 0)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, Bv[u][n], acc[n], 0, 0, 0);
    }
  };
  auto loadblk = [&](int s, double (&Bv)[4][NT]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) loadB(s + u, Bv[u]);
  };
  {
    double B0[4][NT], B1[4][NT];
    loadblk(s0, B0);
    for (int s = s0; s < s1; s += 8) {
      loadblk(s + 4, B1);
      block(s, B0);
      loadblk(s + 8, B0);
      if (s + 4 < s1) block(s + 4, B1);
    }
  }
  // partial sums: wave w keeps state tile w and leaves the other tiles for their owners
  __syncthreads();                                   // (the rows are no longer needed: red overlays xs)
#pragma unroll
  for (int n = 0; n < NT; ++n)
    if (n != wave) {
      const int slot = wave < n ? wave : wave - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((n * 3 + slot) * 4 + r) * 64 + lane] = acc[n][r];
    }
  __syncthreads();
  double own[4] = {0.0, 0.0, 0.0, 0.0};
  if (wave < NT) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (n == wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) own[r] = acc[n][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {                  // source waves in index order
        const int slot = w < wave ? w : w - 1;
        const double pv = (w == wave) ? own[r] : red[((wave * 3 + slot) * 4 + r) * 64 + lane];
        sum = (w == 0) ? pv : sum + pv;
      }
      own[r] = sum;
    }
  }
  // epilogue: emission_scaled_epilogue with the row maximum taken over the workgroup's waves
  ExpConsts ek;
  exp_consts_init(ek);
  double big = 1.7976931348623157e308, l2e = 1.4426950408889634074;
  asm volatile("" : "+v"(big));
  asm volatile("" : "+v"(l2e));
  const int k = wave * 16 + li;
  const int st32 = (flags >> 16) & 1;
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rl = lg + 4 * r;
    const bool bdr = (bad_s[rl] & 1) != 0;
    double x = own[r];
    x = fmax_raw(-big, x);
    x = -fmax_raw(-big, -x);
    x = (own[r] != own[r] || bdr) ? 0.0 : x;
    v[r] = (wave < NT && k < K) ? x : -INFINITY;
    const double mx = row16_max(v[r]);
    if (li == 0 && wave < NT) mxs[wave * 16 + rl] = mx;
  }
  __syncthreads();
  if (wave >= NT) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rl = lg + 4 * r;
    const int64_t g = g0 + rl;
    double mx = mxs[rl];
#pragma unroll
    for (int w = 1; w < NT; ++w) mx = fmax_raw(mx, mxs[w * 16 + rl]);
    const double kx = (mx > -1e300 && mx < 1e300) ? ceil(mx * l2e) : 0.0;
#if EMO_KO & 4
    const double e = v[r] - kx;
#else
    const double e = fast_exp_k(fma(kx, ek.c[13], fma(kx, ek.c[12], v[r])), ek);
#endif
    if (g < nrows && k < K) {
      if (COH) __hip_atomic_store(ll + g * K + k, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (st32) reinterpret_cast<float*>(ll)[g * K + k] = (float)e; else ll[g * K + k] = e;
    }
    if (wave == 0 && li == 0 && g < nrows) { if (COH) __hip_atomic_store(kexp + g, kx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else kexp[g] = kx; }
    if (ll0 && (bad_s[rl] & 2) && g < nrows && k < K) {
      if (COH) __hip_atomic_store(ll0 + (g / Lm) * K + k, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else ll0[(g / Lm) * K + k] = v[r];
    }
  }
}

// ------------------------------------------------------------------------------------
//  The same tile for a workgroup that has a CU to itself and several tiles to do (the statistics workgroups of the fused
//  E-step kernel before their first band opens, kernels_fused.h).  Alone on its CU the tile above is a chain of memory
//  round trips -- window start (over the bus in the SVI loop), mask, rows, nine blocks of theta operands one block
//  ahead, the stores' retirement: 15 us a tile, measured, against 3.9 us of matrix work -- that the stand-alone kernel
//  hides behind three other workgroups per CU.  Here:
//    * the wave's theta operands are the same for EVERY tile (wave w owns k-steps [s0, s1) of the schedule): all nine
//      blocks are loaded once, 288 registers of the 512 a wave of that kernel has, and stay;
//    * the row records (start, mask) of all the workgroup's tiles are formed up front: one round trip;
//    * the rows of tile r + 1 are requested before the k-steps of tile r and reach LDS after its epilogue;
//    * tile r's arrival is sent at the top of tile r + 1, when its stores have retired beside the next rows' loads.
//  Arithmetic: the k-steps, their order, the reduction and the epilogue are the tile's above -- bit-identical results
//  (tests/test_gpu_fused.py compares the two paths for equality).  NT = 4 (K = 64), S <= 144 k-steps (D <= 32).
//  Results are stored coherently (agent scope), like the COH form above.
// ------------------------------------------------------------------------------------
#define EMKS_MAXBLK 9
template <int NT>
__device__ __forceinline__ void emission_orbit_ks_rounds(
    double* __restrict__ smem, const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const double* __restrict__ orb, uint32_t flags, double* __restrict__ ll,
    double* __restrict__ kexp, double* __restrict__ ll0, const int* __restrict__ tiles, int ntile, int nround,
    int nst, int sb, unsigned* cnt, __attribute__((address_space(1))) unsigned long long* dbg) {
  static_assert(NT == 4, "one state tile per wave of the workgroup");
  constexpr int KP = 16 * NT;
  const int N = D + 1, c = D >> 2, nd = (D >> 1) + 1, nleft = (nd + 3) >> 2;
  const int LEN = D + (D >> 1) + 1;
  double* xs = smem;
  double* red = smem;
  const int R0 = 3 * NT * 256 > 16 * LEN + 1 ? 3 * NT * 256 : ((16 * LEN + 1) & ~1);
  double* mxs = smem + R0;                          // [NT][16]
  long long* roff_all = (long long*)(mxs + 16 * NT);             // [nround][16] obs element offset of the row, -1: none
  unsigned char* bad_all = (unsigned char*)(roff_all + 16 * nround);   // [nround][16] bit 0 missing / NaN, bit 1 first row of its window
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  // ---- this wave's theta operands, once (requested first: the row records' chain of round trips -- tile, window start, mask --
  //      runs beside their 295 KB)
  const double* tl = orb + (unsigned)(lg * KP + li * NT);
  const int nmain = c * nd, S = nmain + nleft;
  const int s0 = (S * wave) >> 2, s1 = (S * (wave + 1)) >> 2;
  double Ball[EMKS_MAXBLK][4][NT];
#pragma unroll
  for (int i = 0; i < EMKS_MAXBLK; ++i)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + 4 * i + u;
      const double* trow = tl + (size_t)(s < S ? s : S - 1) * (4 * KP);
#pragma unroll
      for (int n = 0; n < NT; n += 2) {
        const double2 t = *reinterpret_cast<const double2*>(trow + n);
        Ball[i][u][n] = t.x; Ball[i][u][n + 1] = t.y;
      }
    }
  // ---- row records of every tile of this workgroup
  for (int i = tid; i < 16 * nround; i += 256) {
    const int rd = i >> 4, r = i & 15;
    const int ti = rd * nst + sb;
    long long ro = -1;
    unsigned char bd = 0;
    if (ti < ntile) {
      const int64_t g0 = (int64_t)tiles[ti] * 16;
      const int64_t bw0 = g0 / Lm;
      const unsigned t0 = (unsigned)(g0 - bw0 * Lm);
      const bool valid = g0 + r < nrows;
      const unsigned x = t0 + (unsigned)(valid ? r : 0);
      const unsigned bwr = x / (unsigned)Lm;
      const int64_t orow = starts[bw0 + bwr] + (x - bwr * (unsigned)Lm);
      if (valid && (flags & SVIHMM_MASK_AS_NAN) && mask) bd = mask[orow] != 0;
      ro = valid ? orow * D : -1;
      bd |= (valid && x == bwr * (unsigned)Lm) ? 2 : 0;
    }
    roff_all[i] = ro;
    bad_all[i] = bd;
  }
  __syncthreads();
  // ---- rows of a tile: element e = tid + 256 q of the 16 x 2^sh grid (sh = ceil log2 D <= 5)
  const int sh = 32 - __builtin_clz((unsigned)(D - 1));
  const int nel = 16 << sh;
  double pv[2];
  auto request = [&](int rd) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q;
      const int r = (e >> sh) & 15, i = e & ((1 << sh) - 1);
      const long long o = roff_all[rd * 16 + r];
      pv[q] = (e < nel && i < D && o >= 0) ? obs[o + i] : 0.0;
    }
  };
  const int k = wave * 16 + li;
  const double* rowp = xs + li * LEN + 1;
  const double* pa0 = rowp + c * lg;
  const double* p2 = rowp + lg - 1;
  request(0);
  int rd = 0;
  for (; rd < nround; ++rd) {
    const int ti = rd * nst + sb;
    if (ti >= ntile) break;                           // (uniform)
    const int64_t g0 = (int64_t)tiles[ti] * 16;
    unsigned char* bad_s = bad_all + rd * 16;
    // the tile's rows into LDS (requested a tile ago); behind the wait the previous tile's stores have retired too
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg && tid == 0 && rd == 1) dbg[23] = wall_clock64();
    if (tid < 16) { xs[tid * LEN] = 1.0; xs[tid * LEN + D + 1] = 1.0; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q;
      const int r = e >> sh, i = e & ((1 << sh) - 1);
      if (e < nel && i < D) {
        double v = pv[q];
        if (v != v) { bad_s[r] |= 1; v = 0.0; }
        xs[r * LEN + i + 1] = v;
        if (i + N <= LEN - 2) xs[r * LEN + i + N + 1] = v;
      }
    }
    __syncthreads();
    if ((rd + 1) * nst + sb < ntile) request(rd + 1);
    // (the arrival after the request: nothing behind it touches memory before the next tile's top, nobody waits for it)
    if (rd > 0 && tid == 0) __hip_atomic_fetch_add(cnt + 16 * (rd - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dbg && tid == 0 && rd >= 1 && rd <= 2) dbg[29 + rd] = wall_clock64();
    // ---- k-steps
    double4_t acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // Straight-line schedule for a wave that has its SIMD to itself: the (uniform) position in the feature schedule is
    // stepped with scalar selects, the two LDS operand addresses of a k-step are selected, not branched to, and the
    // operands of block i + 1 are requested before the MFMAs of block i -- the stand-alone kernel's branchy block leaves
    // a lone wave waiting on every LDS read (measured: 7.1 us for the 36 k-steps against 3.9 us of matrix work).
    int sd = s0 / c, sa = s0 - sd * c;
    const double* pxl = rowp + (N - 1);
    auto operands = [&](int s, double (&xa)[4], double (&xb)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool mainstep = s + u < nmain;
        int jl = s + u - nmain;
        jl = jl < 0 ? 0 : (jl < nleft ? jl : nleft - 1);
        const double* pa = mainstep ? pa0 + sa : pxl;
        const double* pb = mainstep ? pa0 + sa + sd : p2 + 4 * jl;
        xa[u] = *pa; xb[u] = *pb;
        const bool wrap = sa + 1 == c;
        sa = wrap ? 0 : sa + 1;
        sd += wrap ? 1 : 0;
      }
    };
    double xa[4], xb[4];
    operands(s0, xa, xb);
#pragma unroll
    for (int i = 0; i < EMKS_MAXBLK; ++i) {
      const int s = s0 + 4 * i;
      double xan[4], xbn[4];
      if (i + 1 < EMKS_MAXBLK) operands(s + 4, xan, xbn);
      if (s < s1) {                                   // (uniform; a skipped block would only add B x 0)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double A = (s + u < s1) ? xa[u] * xb[u] : 0.0;
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A, Ball[i][u][n], acc[n], 0, 0, 0);
        }
      }
      if (i + 1 < EMKS_MAXBLK) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { xa[u] = xan[u]; xb[u] = xbn[u]; }
      }
    }
    if (dbg && tid == 0 && rd == 1) dbg[26] = wall_clock64();
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (n != wave) {
        const int slot = wave < n ? wave : wave - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((n * 3 + slot) * 4 + r) * 64 + lane] = acc[n][r];
      }
    __syncthreads();
    double own[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (n == wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) own[r] = acc[n][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int slot = w < wave ? w : w - 1;
        const double p = (w == wave) ? own[r] : red[((wave * 3 + slot) * 4 + r) * 64 + lane];
        sum = (w == 0) ? p : sum + p;
      }
      own[r] = sum;
    }
    // (the epilogue's constants are formed per tile: kept across the k-steps they cost 34 of the registers the resident
    //  theta operands need)
    ExpConsts ek;
    exp_consts_init(ek);
    double big = 1.7976931348623157e308, l2e = 1.4426950408889634074;
    asm volatile("" : "+v"(big));
    asm volatile("" : "+v"(l2e));
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = lg + 4 * r;
      const bool bdr = (bad_s[rl] & 1) != 0;
      double x = own[r];
      x = fmax_raw(-big, x);
      x = -fmax_raw(-big, -x);
      x = (own[r] != own[r] || bdr) ? 0.0 : x;
      v[r] = k < K ? x : -INFINITY;
      const double mx = row16_max(v[r]);
      if (li == 0) mxs[wave * 16 + rl] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = lg + 4 * r;
      const int64_t g = g0 + rl;
      double mx = mxs[rl];
#pragma unroll
      for (int w = 1; w < NT; ++w) mx = fmax_raw(mx, mxs[w * 16 + rl]);
      const double kx = (mx > -1e300 && mx < 1e300) ? ceil(mx * l2e) : 0.0;
      const double e = fast_exp_k(fma(kx, ek.c[13], fma(kx, ek.c[12], v[r])), ek);
      if (g < nrows && k < K) __hip_atomic_store(ll + g * K + k, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wave == 0 && li == 0 && g < nrows) __hip_atomic_store(kexp + g, kx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ll0 && (bad_s[rl] & 2) && g < nrows && k < K)
        __hip_atomic_store(ll0 + (g / Lm) * K + k, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dbg && tid == 0 && rd == 1) dbg[12] = wall_clock64();
    // (the next tile's rows overlay `red`: every read of it lies before the barrier above; mxs is written again two
    //  barriers from here)
  }
  if (rd > 0) {                                       // the last tile's arrival
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(cnt + 16 * (rd - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
