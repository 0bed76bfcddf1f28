// kernels_emission.h -- K0 NIW -> theta, K1 emission expected log-likelihood (VALU fallback + fp64 MFMA GEMM).
// Part of libsvihmm_hip.so; compiled in tu_emission.hip.
#pragma once

// The emission GEMM evaluates the NIW quadratic form expanded around the origin: its terms are of
// size m'Wm = (nu/2) mu' sigma^-1 mu and cancel down to the (x - mu)'W(x - mu) the reference
// computes centred.  Beyond this size the cancellation costs more than 5e-7 in the
// log-likelihoods (measured: error ~ 5e-16 m'Wm) and the upload is refused (status word).  (A
// common offset of the data reaches it at |mu| / sigma ~ 5e3; states separated by that many
// standard deviations do too, where the loss is harmless -- such callers shift as well.)
#define NIW_CANCEL_LIMIT 1.0e9

// ---- fp32 mode: operand format of k_emission_bf16x3 (K1d) -----------------------------
// A value is carried as three bf16 terms hi + mid + lo (round-to-nearest each, residuals formed
// in double): 24+ significant bits, i.e. the value to fp32 accuracy, on the bf16 matrix pipe.
// States travel in PAIRS (k, k') = (2 p, 2 p + 1).  U is lower triangular: the components 0..15 of a
// state do not see the dimensions 16..31, so the k-block of those dimensions serves the upper
// components of BOTH states of a pair in one MFMA (rows 0..15: components 16..31 of k, rows 16..31:
// of k') -- three instead of four instructions per pair and product term.
constexpr int EMB_BLOCKS = 9;         // per pair: 3 terms x {k dims 0..15, k' dims 0..15, both dims 16..31} of 64 lanes x 16 B
constexpr int EMB_REC = 10240;        // bytes per pair record: the nine blocks, 2 x 32 bias floats, the two constants, padding
constexpr int EMB_NREC = 34;          // records in the parameter buffer: 32 pairs + what the copies run ahead
__device__ __forceinline__ uint32_t bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf16_up(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ void bf16_split3(double v, uint32_t (&t)[3]) {
  t[0] = bf16_rne((float)v);
  double r = v - (double)bf16_up(t[0]);
  t[1] = bf16_rne((float)r);
  r -= (double)bf16_up(t[1]);
  t[2] = bf16_rne((float)r);
}

// ------------------------------------------------------------------------------------
//  K1b: emission as an fp64 MFMA GEMM  ll[rows x K] = Phi[rows x Fp] * theta[Fp x Kp]
//       with Phi generated on the fly from x rows staged in LDS.
//       v_mfma_f64_16x16x4_f64: A lane l -> A[i=l&15][k=l>>4]; B lane l -> B[k=l>>4][j=l&15];
//       C/D lane l reg r -> C[row=(l>>4)+4r][col=l&15].
//       Workgroup = 4 waves x (MT=2 row tiles) = 128 rows; NT n-tiles of 16 states.
//       grid (ceil(n/128), Kp/(16*NT)), block 256.
// ------------------------------------------------------------------------------------
//       SCALED (grid.y == 1, the workgroup owns whole rows): instead of ll the kernel
//       writes Eh = exp(ll) * 2^-k with k = ceil(max_j ll / ln 2) and the per-row k
//       (kexp) -- the input format of the scaled linear-domain sweeps K2e/K2f, so that
//       no exp is left in their time loops.
// Scaled epilogue of the emission GEMMs: Eh = exp(ll) * 2^-k, k = ceil(max_j ll / ln 2) per row.
template <int NT, int MT>
__device__ __forceinline__ void emission_scaled_epilogue(
    const double (&outv)[MT][NT][4], const unsigned char* bad_s, int wave, int li, int lg,
    int64_t g0, int64_t nrows, int K, int n0, double* __restrict__ ll, double* __restrict__ kexp,
    int st32, double* __restrict__ ll0, int Lm) {
    // One exp per (row, state) is the algorithmic minimum of transcendental work on the
    // whole E-step; keep it lean: constants pinned in VGPRs, branch-free NaN/inf handling.
    ExpConsts ek;
    exp_consts_init(ek);
    double big = 1.7976931348623157e308, l2e = 1.4426950408889634074;
    asm volatile("" : "+v"(big));
    asm volatile("" : "+v"(l2e));
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = wave * 16 * MT + m * 16 + lg + 4 * r;
        const int64_t g = g0 + rl;
        const bool bd = (bad_s[rl] & 1) != 0;
        double v[NT], mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int k = n0 + n * 16 + li;
          double x = outv[m][n][r];
          x = fmax_raw(-big, x);                 // -inf -> -DBL_MAX, NaN -> -DBL_MAX ...
          x = -fmax_raw(-big, -x);               // +inf -> DBL_MAX
          x = (outv[m][n][r] != outv[m][n][r] || bd) ? 0.0 : x;   // ... NaN and masked rows -> 0
          v[n] = k < K ? x : -INFINITY;
          mx = fmax_raw(mx, v[n]);
        }
        mx = row16_max(mx);   // all lanes: the DPP reduction stays outside the store guards
        const double kx = (mx > -1e300 && mx < 1e300) ? ceil(mx * l2e) : 0.0;
        double* orow = ll + g * K + n0 + li;
        float* orow32 = reinterpret_cast<float*>(ll) + g * K + n0 + li;   // fp32 mode: Eh stored as float
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const double e = fast_exp_k(fma(kx, ek.c[13], fma(kx, ek.c[12], v[n])), ek);
          if (g < nrows && n0 + n * 16 + li < K) {
            if (st32) orow32[n * 16] = (float)e; else orow[n * 16] = e;      // uniform branch
          }
        }
        if (li == 0 && g < nrows) kexp[g] = kx;
        // first row of a window: its log-likelihoods unscaled, for the initial message
        // (mod_init + ll_0 is combined in the log domain: k_lin_init)
        if (ll0 && (bad_s[rl] & 2) && g < nrows) {
          double* __restrict__ o0 = ll0 + (g / Lm) * K + n0 + li;
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if (n0 + n * 16 + li < K) o0[n * 16] = v[n];
        }
      }
    }
}

template <int NT, int MT, bool SCALED>
__global__ __launch_bounds__(256) void k_emission_mfma(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K, int Kp,
    int Fp, const double* __restrict__ theta, const int* __restrict__ fab,
    uint32_t flags, double* __restrict__ ll, double* __restrict__ kexp, double* __restrict__ ll0) {
  // workgroup = 4 waves x MT row tiles of 16 rows
  constexpr int ROWS = 64 * MT;
  extern __shared__ double smem[];
  const int DS = (D + 2) | 1;  // odd row stride (doubles); slot D = 1.0, slot D+1 = 0.0
  double* xs = smem;                              // [ROWS][DS]
  int* fabs_ = (int*)(xs + ROWS * DS);            // [Fp] packed (a | b<<16)
  long long* rowoff = (long long*)(fabs_ + Fp);   // [ROWS] obs element offset of the row, -1: none (Fp is even)
  unsigned char* bad_s = (unsigned char*)(rowoff + ROWS);  // [ROWS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * ROWS;
  const int n0 = blockIdx.y * (16 * NT);

  // Row bookkeeping once per row, in 32-bit arithmetic relative to the tile's first row (the
  // copy loop below used to pay a 64-bit division per ELEMENT; VALU work of a prologue wave
  // blocks the MFMAs of the three waves it shares the SIMD with).
  {
    const int64_t bw0 = g0 / Lm;
    const unsigned t0 = (unsigned)(g0 - bw0 * Lm);
    for (int r = tid; r < ROWS; r += 256) {
      const bool valid = g0 + r < nrows;
      const unsigned x = t0 + (unsigned)(valid ? r : 0);
      const unsigned bwr = x / (unsigned)Lm;
      const int64_t orow = starts[bw0 + bwr] + (x - bwr * (unsigned)Lm);
      unsigned char bd = 0;
      if (valid && (flags & SVIHMM_MASK_AS_NAN) && mask) bd = mask[orow] != 0;
      rowoff[r] = valid ? orow * D : -1;
      bad_s[r] = bd | ((valid && x == bwr * (unsigned)Lm) ? 2 : 0);   // bit 1: step 0 of its window
      xs[r * DS + D] = 1.0;
      xs[r * DS + D + 1] = 0.0;
    }
  }
  for (int e = tid; e < Fp; e += 256) fabs_[e] = fab[e];
  __syncthreads();
  {
    // (row, column) from shifts: DP = D rounded up to a power of two (<= 256, since the tile
    // must fit LDS), 256 / DP rows per pass, consecutive lanes read consecutive doubles
    const int sh = 32 - __builtin_clz((unsigned)(D > 1 ? D - 1 : 1));
    const int rpp = 256 >> sh, i = tid & ((1 << sh) - 1), rr = tid >> sh;
    // all of a thread's loads are issued before the first LDS write (branch-free, clamped
    // addresses): written as load -> wait -> store per row the compiler serialises 16 HBM
    // round trips per workgroup, 9 % of the kernel on the bench shape
    constexpr int CH = 16;
    for (int rb = 0; rb < ROWS; rb += rpp * CH) {
      double v[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int r = rb + j * rpp + rr;
        const long long o = r < ROWS ? rowoff[r] : -1;
        const bool ok = i < D && o >= 0;
        const double t = obs[ok ? o + i : 0];
        v[j] = ok ? t : 0.0;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int r = rb + j * rpp + rr;
        if (i < D && r < ROWS) {   // (256 >> sh rows per pass can exceed a 64-row tile)
          if (v[j] != v[j]) { bad_s[r] |= 1; v[j] = 0.0; }
          xs[r * DS + i] = v[j];
        }
      }
    }
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
  if (SCALED) {
    emission_scaled_epilogue<NT, MT>(outv, bad_s, wave, li, lg, g0, nrows, K, n0, ll, kexp, (flags >> 16) & 1, ll0, Lm);
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = wave * 16 * MT + m * 16 + lg + 4 * r;
        const int64_t g = g0 + rl;
        const bool bd = (bad_s[rl] & 1) != 0;
        if (g < nrows) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int k = n0 + n * 16 + li;
            if (k < K) ll[g * K + k] = bd ? 0.0 : nan_to_num(outv[m][n][r]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
//  K1c: the scaled emission GEMM with an address-free feature schedule (D % 8 == 0, K <= 64).
//  Measured on K1b (bench shape): every non-MFMA instruction a SIMD issues costs MFMA issue
//  slots -- generating the A operands (table lookup, unpacking, four LDS addresses per k-step)
//  took 8.5 % of the kernel, the 16 B-operand loads per four k-steps 5 %.  Here the features
//  are ordered along orbits of the index shift: with N = D + 1 (odd), c = D / 4 and
//  delta = 0 .. D/2, k-step s = delta * c + a0 gives lane group lg the product
//       x[a] * x[(a + delta) mod N],      a = a0 + c * lg        (a0 = 0 .. c-1)
//  -- every unordered pair {a, b}, a, b < N - 1 or not, exactly once, the leftover a = N - 1
//  of each delta in ceil((D/2+1)/4) extra k-steps (group lg: delta = 4j + lg).  The row is
//  stored in LDS as x[-1 .. 3D/2 - 1] with x[i] = x[i mod N] (slot N-1 = 1.0), so "mod N" and
//  the per-group shift c * lg are part of a per-lane constant base and each operand is ONE
//  ds_read_b128 at base + immediate (the two row tiles of the wave are interleaved: the
//  b128 holds the same column of rows r and r + 16).  theta is re-laid for it by
//  k_theta_orbit: rows in schedule order, the NT state tiles of a lane adjacent (b128 loads).
//  Per k-step: 2 LDS reads, 2 v_mul_f64, NT/2 global loads, 2 NT MFMAs; no table, no
//  unpacking, two address adds per U k-steps.  141 k-steps at D = 32 instead of 144.
// ------------------------------------------------------------------------------------
// theta entry of the feature pair (i <= j <= D) for state k, in both layouts: the canonical
// row feat_index(i, j) and -- when `orb` is given -- its place in the orbit schedule
__device__ __forceinline__ void theta_store(double* __restrict__ theta, double* __restrict__ orb,
                                            int i, int j, int D, int Kp, int k, double v) {
  theta[(size_t)(i * (D + 1) - i * (i - 1) / 2 + (j - i)) * Kp + k] = v;
  if (orb) {
    const int N = D + 1, c = D >> 2, nd = (D >> 1) + 1, NT = Kp >> 4;
    const int d = j - i;
    const int a = d <= (D >> 1) ? i : j, dl = d <= (D >> 1) ? d : N - d;
    int lg, st;
    if (a < N - 1) { lg = a / c; st = dl * c + (a - lg * c); }
    else { lg = dl & 3; st = c * nd + (dl >> 2); }
    orb[(size_t)(4 * st + lg) * Kp + (k & 15) * NT + (k >> 4)] = v;
  }
}
__global__ void k_theta_orbit(const double* __restrict__ theta, int D, int Kp, int NT,
                              double* __restrict__ orb) {
  const int fo = blockIdx.x, s = fo >> 2, lg = fo & 3, k = threadIdx.x;
  const int N = D + 1, c = D >> 2, nd = (D >> 1) + 1;
  int a, b;
  bool valid = true;
  if (s < c * nd) {
    const int d = s / c, a0 = s - d * c;
    a = a0 + c * lg;
    b = (a + d) % N;
  } else {
    const int d = 4 * (s - c * nd) + lg;
    valid = d < nd;
    a = N - 1;
    b = (N - 1 + d) % N;
  }
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  const int f = lo * (D + 1) - lo * (lo - 1) / 2 + (hi - lo);
  if (k < Kp) orb[(size_t)fo * Kp + (k & 15) * NT + (k >> 4)] = valid ? theta[(size_t)f * Kp + k] : 0.0;
}

// MT = row tiles per wave: 2 (128 rows per workgroup) for batches that fill the chip, 1 (64 rows,
// round 3) for minibatches with fewer than one 128-row workgroup per CU -- the SVI iteration of 64
// windows is 129 workgroups of 128 rows on 256 CUs.
template <int NT, int U, int MT = 2>
__global__ __launch_bounds__(256) void k_emission_orbit(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const double* __restrict__ orb, uint32_t flags, double* __restrict__ ll,
    double* __restrict__ kexp, double* __restrict__ ll0,
    int64_t* __restrict__ starts_copy = nullptr, int nstarts = 0) {
  constexpr int ROWS = 64 * MT, KP = 16 * NT;
  // SVI loop (round 5): `starts` is the host's pinned, device-visible slot -- the window starts are read over
  // PCIe here once (a separate k_pull launch in front of this kernel cost 6 us + a 7 us dispatch gap on the
  // iteration's critical path); workgroup 0 leaves the device copy the statistics kernel reads
  if (starts_copy && blockIdx.x == 0)
    for (int i = threadIdx.x; i < nstarts; i += 256) starts_copy[i] = starts[i];
  typedef typename std::conditional<MT == 2, double2, double>::type XV;   // one column of the wave's row tiles
  extern __shared__ double smem[];
  const int N = D + 1, c = D >> 2, nd = (D >> 1) + 1, nleft = (nd + 3) >> 2;
  const int LEN = D + (D >> 1) + 1;                 // slots -1 .. 3D/2 - 1 (odd count)
  XV* xs2 = (XV*)smem;                              // [64][LEN]: MT = 2 (row r, row r + 16) pairs
  // LDS banking (round 6; tools/probe/lds_probe.hip patterns 30 / 31): a ds_read_b128 is served in 16-lane groups that
  // mix the lane rows 0-3, 12-15 of one lane group with rows 4-11 of the next, whose operands start c = D / 4 slots
  // further.  At D = 32 (c = 8, row pitch 49 = 1 mod 16 slots) rows li and li + 8 of neighbouring groups met on one
  // bank in EVERY read (twice the LDS cycles, SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE): every second block of
  // eight rows starts eight slots later now (EO_SHIFT), which moves its reads onto the banks the other block leaves
  // free.  D = 64 (c = 16) never conflicted; the other widths keep the plain pitch.
#ifdef SVIHMM_AB_EMIS_OLD       // (A/B builds only)
  const int SHIFT8 = 0;
#else
  const int SHIFT8 = (MT == 2 && c == 8) ? 8 : 0;
#endif
#define EO_SHIFT(ROW) (SHIFT8 * ((ROW) >> 3))      // ROW = 16 wave + lane row
  long long* rowoff = (long long*)(xs2 + 64 * LEN + 8 * SHIFT8);
  unsigned char* bad_s = (unsigned char*)(rowoff + ROWS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g0 = (int64_t)blockIdx.x * ROWS;
  double* xs1 = (double*)xs2;
  // row r of the tile -> wave r >> 5, row tile (r >> 4) & 1, lane row r & 15
  auto slot = [&](int r, int idx) {
    return MT == 2 ? ((((r >> 5) * 16 + (r & 15)) * LEN + EO_SHIFT((r >> 5) * 16 + (r & 15)) + idx + 1) << 1) + ((r >> 4) & 1)
                   : r * LEN + idx + 1;
  };
  {
    const int64_t bw0 = g0 / Lm;
    const unsigned t0 = (unsigned)(g0 - bw0 * Lm);
    for (int r = tid; r < ROWS; r += 256) {
      const bool valid = g0 + r < nrows;
      const unsigned x = t0 + (unsigned)(valid ? r : 0);
      const unsigned bwr = x / (unsigned)Lm;
      const int64_t orow = starts[bw0 + bwr] + (x - bwr * (unsigned)Lm);
      unsigned char bd = 0;
      if (valid && (flags & SVIHMM_MASK_AS_NAN) && mask) bd = mask[orow] != 0;
      rowoff[r] = valid ? orow * D : -1;
      bad_s[r] = bd | ((valid && x == bwr * (unsigned)Lm) ? 2 : 0);   // bit 1: step 0 of its window
      xs1[slot(r, -1)] = 1.0;
      xs1[slot(r, D)] = 1.0;
    }
  }
  __syncthreads();
  {
    const int sh = 32 - __builtin_clz((unsigned)(D - 1));
    const int rpp = 256 >> sh, i = tid & ((1 << sh) - 1), rr = tid >> sh;
    constexpr int CH = 16;      // loads first, LDS writes after (see K1b)
    for (int rb = 0; rb < ROWS; rb += rpp * CH) {
      double v[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int r = rb + j * rpp + rr;
        const long long o = r < ROWS ? rowoff[r] : -1;
        const bool ok = i < D && o >= 0;
        const double t = obs[ok ? o + i : 0];
        v[j] = ok ? t : 0.0;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int r = rb + j * rpp + rr;
        if (i < D && r < ROWS) {
          if (v[j] != v[j]) { bad_s[r] |= 1; v[j] = 0.0; }
          xs1[slot(r, i)] = v[j];
          if (i + N <= LEN - 2) xs1[slot(r, i + N)] = v[j];
        }
      }
    }
  }
  __syncthreads();

  const int li = lane & 15, lg = lane >> 4;
  double4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const XV* rowp = xs2 + (wave * 16 + li) * LEN + EO_SHIFT(wave * 16 + li) + 1;   // slot 0 of this lane's row (pair)
#undef EO_SHIFT
  const XV* pa0 = rowp + c * lg;
  const unsigned loff = (unsigned)(lg * KP + li * NT);         // lane part of the theta address
  auto kstep = [&](const XV xa, const XV xb, const double (&Bv)[NT]) {
    if constexpr (MT == 2) {
      const double A0 = xa.x * xb.x, A1 = xa.y * xb.y;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, Bv[n], acc[0][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, Bv[n], acc[1][n], 0, 0, 0);
    } else {
      const double A0 = xa * xb;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, Bv[n], acc[0][n], 0, 0, 0);
    }
  };
  auto loadB = [&](const double* trow, double (&Bv)[NT]) {
    if constexpr (NT % 2 == 0) {
#pragma unroll
      for (int n = 0; n < NT; n += 2) {
        const double2 t = *reinterpret_cast<const double2*>(trow + loff + n);
        Bv[n] = t.x; Bv[n + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) Bv[n] = trow[loff + n];
    }
  };
  // blocks of U k-steps, (delta, a0) in schedule order; the B operands of block i + 1 are
  // requested before the MFMAs of block i (two register sets, loop unrolled by two blocks)
  const int nblk = nd * (c / U);
  int bd = 0, ba = 0;                                // (delta, a0) of the next block to compute
  auto block = [&](const double (&Bv)[U][NT]) {
    const XV* pa = pa0 + ba;
    const XV* pb = pa + bd;
    XV xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { xa[u] = pa[u]; xb[u] = pb[u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) kstep(xa[u], xb[u], Bv[u]);
    ba += U;
    if (ba == c) { ba = 0; ++bd; }
  };
  auto loadblk = [&](int bi, double (&Bv)[U][NT]) {
    const double* t = orb + (size_t)(bi < nblk ? bi : nblk - 1) * (U * 4 * KP);
#pragma unroll
    for (int u = 0; u < U; ++u) loadB(t + (size_t)u * 4 * KP, Bv[u]);
  };
  {
    double B0[U][NT], B1[U][NT];
    loadblk(0, B0);
    for (int bi = 0; bi < nblk; bi += 2) {
      loadblk(bi + 1, B1);
      block(B0);
      loadblk(bi + 2, B0);
      if (bi + 1 < nblk) block(B1);
    }
  }
  const double* tbs = orb + (size_t)nblk * (U * 4 * KP);
  {
    const XV xl = rowp[N - 1];
    const XV* p2 = rowp + lg - 1;
    for (int j = 0; j < nleft; ++j) {
      double Bv[NT];
      loadB(tbs, Bv);
      kstep(xl, p2[4 * j], Bv);
      tbs += (size_t)4 * KP;
    }
  }
  double outv[MT][NT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) outv[m][n][r] = acc[m][n][r];
  __builtin_amdgcn_sched_barrier(0);
  emission_scaled_epilogue<NT, MT>(outv, bad_s, wave, li, lg, g0, nrows, K, 0, ll, kexp, (flags >> 16) & 1, ll0, Lm);
}

// ------------------------------------------------------------------------------------
//  K1c', the minibatch form of K1c (round 5).  With about one 16-row tile per SIMD the row-tile kernel above is
//  bound by its busiest SIMD, not by the matrix pipe: a wave's 16 rows x 64 states are 564 fp64 MFMAs of 64
//  cycles (15 us), and the SVI iteration's 16448 rows are 1028 such waves on 1024 SIMDs -- four SIMDs run two
//  (tools/probe/emo_probe.hip: 63 windows 29 us, 64 windows 45 us; a deeper prefetch ring changed nothing).
//  Here a workgroup is ONE 16-row tile and its four waves split the feature schedule's k-steps (141 at D = 32:
//  35 / 35 / 35 / 36), so the unit of work is a quarter tile: at most five per SIMD where the row-tile form has
//  two whole ones.  The four partial sums meet in LDS (fixed order: deterministic); wave w then owns state tile
//  w in the epilogue, and the rows' maximum over the states crosses the waves through LDS as well.
//  (First attempt, measured and dropped: one wave per STATE tile -- each k-step's two LDS reads then feed one
//  MFMA instead of four and the kernel is LDS-bound at 36 TF/s whatever the batch.)
//  Same theta layout and LDS row format as K1c.  blockDim = 256, K <= 64 (NT = Kp / 16 state tiles per wave).
// ------------------------------------------------------------------------------------
#ifndef EMO_KO
#define EMO_KO 0      // measurement knock-outs (tools/probe/emo_probe.hip): 1 no k-steps, 2 no theta loads, 4 no exp
#endif
#include "kernels_emission_ks.h"
template <int NT>
__global__ __launch_bounds__(256) void k_emission_orbit_ks(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const double* __restrict__ orb, uint32_t flags, double* __restrict__ ll,
    double* __restrict__ kexp, double* __restrict__ ll0,
    int64_t* __restrict__ starts_copy, int nstarts) {
  if (starts_copy && blockIdx.x == 0)       // (SVI loop: see k_emission_orbit)
    for (int i = threadIdx.x; i < nstarts; i += 256) starts_copy[i] = starts[i];
  extern __shared__ double smem[];
  emission_orbit_ks_body<NT, false>(smem, obs, mask, starts, nrows, Lm, D, K, orb, flags, ll, kexp, ll0, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------
//  K1d: the scaled emission kernel of the fp32 mode (K <= 64, D <= 32, NIW family).
//  The expanded feature form cannot run below fp64 (its terms cancel); here the quadratic form is
//  evaluated CENTRED, ll[t][k] = c_k - |U_k x_t + b_k|^2 with U_k = sqrt(nu_k / 2) L_k^-1 and
//  b_k = -U_k m_k (k_niw_to_theta_wave writes both), as a GEMM on the bf16 matrix pipe: x and U
//  travel as three bf16 terms each (hi + mid + lo = the value to fp32 accuracy) and the six
//  products hi hi, hi mid, mid hi, hi lo, lo hi, mid mid accumulate in fp32 -- what is left out
//  is below 2^-24 of a product.  bf16 MFMA runs at 16x the fp32-input MFMA rate on gfx950, so the
//  six-term product is 2.7x faster than an fp32 MFMA GEMM of the same shape.
//  v_mfma_f32_32x32x16_bf16: M = 32 components (A = U blocks from LDS), N = 32 rows of the batch
//  (B = x, held in registers for the whole state loop), k = 16 of the observation's dimensions;
//  C = the bias.  U is lower triangular, so the states go in pairs: per product term one MFMA per
//  state for the dimensions 0..15 (all 32 components) and ONE for the dimensions 16..31 of both
//  (rows 0..15: components 16..31 of the first state, rows 16..31: of the second) -- 18 MFMAs per
//  pair and row tile instead of 24.  A lane's results are components of ONE row: the square sum is
//  in-lane plus one exchange between the wave's halves, which at the same time deals the two
//  states of the pair to the two halves.
//  Workgroup = 4 waves x 64 rows, two workgroups per CU (so that one's prologue / epilogue runs
//  under the other's MFMAs).  The pair records (nine operand blocks + biases + constants, EMB_REC
//  bytes) stream global -> LDS directly (global_load_lds_dwordx4) through ONE buffer: operands
//  LDS -> registers, barrier, the next record's copy under this pair's 36 MFMAs, barrier.  The
//  wave's 64 x 64 ll tile is transposed through LDS (XOR-swizzled columns) so that the scaled
//  epilogue (row maximum, exp2, float store) writes whole rows.
//  Measured (bench shape, 1e6 rows, K = 64, D = 32): 0.52 ms against 1.17 ms of the fp64 feature
//  GEMM; on full-range random operands (tools/probe/emb_probe.hip) the bf16 pipe is power-bound:
//  ~1000-1080 TF/s at 1.74 GHz with the MFMA pipe 64 % busy, whatever the schedule (staging,
//  barriers, operand prefetch knocked out: +-3 %), which is why the levers that paid were the ones
//  that remove MFMAs (the pairing) or idle phases (two independent workgroups per CU).
// ------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 embf8_t;
typedef __attribute__((ext_vector_type(16))) float emf16_t;
__device__ __forceinline__ embf8_t emb_cast(const uint4 v) { return __builtin_bit_cast(embf8_t, v); }
__device__ __forceinline__ float row16_max_f32(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
// fp32 value -> three bf16 terms, residuals exact in fp32
__device__ __forceinline__ void bf16_split3f(float f, uint32_t (&t)[3]) {
  t[0] = bf16_rne(f);
  float r = f - bf16_up(t[0]);
  t[1] = bf16_rne(r);
  r -= bf16_up(t[1]);
  t[2] = bf16_rne(r);
}
__device__ __forceinline__ void emb_glds16(const char* src, char* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

// End of a pair's step: the wave's own global -> LDS copies have landed, then the workgroup meets
// (raw barrier with the explicit wait; the step's other barrier, behind the operand reads, waits
// for lgkmcnt the same way).  The sched_barriers keep the MFMAs in front of it: without them the
// compiler hoists the barrier -- and with it the wait for the copy just issued -- above the MFMAs.
__device__ __forceinline__ void emb_step_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(EMB_KO) && (EMB_KO & 4))
  __builtin_amdgcn_s_barrier();
#endif
  __builtin_amdgcn_sched_barrier(0);
}

// MT = row tiles of 32 per wave: 2 (large batches: 256-row workgroups, two per CU) or 1 (round 5, batches with
// fewer than one 256-row workgroup per CU -- the 64-window minibatch: 128-row workgroups, half the chain per pair)
// NH = groups of four waves that split the state pairs of the same 128 rows (minibatch form, MT = 1: the chain of
// a wave is npair / NH steps instead of npair, the groups' waves share the SIMDs and fill each other's VALU
// phases; each group stages its own pair record, every wave finishes the rows / NH of its tile in the epilogue)
template <int MT, int NH, int RW>
__device__ __forceinline__ void emission_bf16x3_body(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const char* __restrict__ uw, uint32_t flags,
    float* __restrict__ Eh, double* __restrict__ kexp, double* __restrict__ ll0,
    int64_t* __restrict__ starts_copy, int nstarts) {
  constexpr int ROWS = 32 * RW * MT, WR = 32 * MT;        // rows per workgroup / per wave
  // (SVI loop: `starts` is the host's pinned slot, see k_emission_orbit; workgroup 0 leaves the device copy)
  if (starts_copy && blockIdx.x == 0)
    for (int i = threadIdx.x; i < nstarts; i += 64 * RW * NH) starts_copy[i] = starts[i];
  extern __shared__ uint4 smem4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wv % RW, sh = wv / RW;                            // row wave, pair group
  char* stage = reinterpret_cast<char*>(smem4) + sh * EMB_REC;       // [NH][EMB_REC]: one pair record per group
  float* tile_s = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + NH * EMB_REC);   // [RW waves][WR rows][64 states]
  const int t = lane & 31, hh = lane >> 5;
  const int64_t g0 = (int64_t)blockIdx.x * ROWS;
  const int npair = (K + 1) >> 1;
  const int nh = (npair + NH - 1) / NH, sp0 = sh * nh;               // this group's pairs: [sp0, sp0 + nh)

  // pair record p -> the group's LDS buffer, branch-free: the group's RW waves issue ceil(10 / RW) 1 KB copies each
  // (chunks wave, RW + wave, ..., clamped to the last chunk, which some waves copy alike)
  auto stage_load = [&](int pr) {
#if !(defined(EMB_KO) && (EMB_KO & 2))
    const char* src = uw + (size_t)pr * EMB_REC + lane * 16;
#pragma unroll
    for (int c = 0; c < (10 + RW - 1) / RW; ++c) {
      const int ch = (wave + c * RW < 9 ? wave + c * RW : 9) * 1024;
      emb_glds16(src + ch, stage + ch);
    }
#endif
  };
  stage_load(sp0);

  // ---- the lane's two rows: dimensions 16 c + 8 hh + e as three bf16 terms (B operands)
  embf8_t xb[MT][3][2];
  int bflag[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = wave * WR + m * 32 + t;
    const int64_t g = g0 + r;
    const bool valid = g < nrows;
    const int64_t gg = valid ? g : 0;
    const int64_t wi = gg / Lm;
    const int tt = (int)(gg - wi * Lm);
    const int64_t orow = starts[wi] + tt;
    int bd = 0;
    if (valid && (flags & SVIHMM_MASK_AS_NAN) && mask) bd = mask[orow] != 0;
    const double* xp = obs + orow * D;
    bool isnan_ = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v[8];
      const int i0 = 16 * c + 8 * hh;
      if ((D & 7) == 0) {                                            // whole groups of eight: 16-byte loads
        const bool ok = valid && i0 < D;
        const double2* x2 = reinterpret_cast<const double2*>(xp + (ok ? i0 : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double2 q2 = x2[e];
          v[2 * e] = ok ? q2.x : 0.0; v[2 * e + 1] = ok ? q2.y : 0.0;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = valid && i0 + e < D;
          const double x = xp[ok ? i0 + e : 0];
          v[e] = ok ? x : 0.0;
        }
      }
      uint32_t w3[3][4];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (v[e] != v[e]) { isnan_ = true; v[e] = 0.0; }
        uint32_t t3[3];
        bf16_split3f((float)v[e], t3);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          if (e & 1) w3[s3][e >> 1] |= t3[s3] << 16; else w3[s3][e >> 1] = t3[s3];
        }
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
        xb[m][s3][c] = emb_cast(make_uint4(w3[s3][0], w3[s3][1], w3[s3][2], w3[s3][3]));
    }
    int bf = bd | (isnan_ ? 1 : 0);
    bf |= __shfl_xor(bf, 32, 64);                                    // the row's other half of the dimensions
    bflag[m] = bf | ((valid && tt == 0) ? 2 : 0);                    // bit 1: step 0 of its window
  }
  __syncthreads();                                                   // (carries the wait for record 0)

  // One step per pair: operands LDS -> registers, barrier (the buffer is free), the next record's copy
  // on its way under this pair's 36 MFMAs, barrier (it has landed).
  float* tw = tile_s + wave * WR * 64;
  const emf16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int si = 0; si < nh; ++si) {
    const int sp = sp0 + si;
    embf8_t a[3][3];
    emf16_t bv[2];
    float cst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(stage + 9216 + (32 * u + 8 * q + 4 * hh) * 4);
        bv[u][4 * q] = b4.x; bv[u][4 * q + 1] = b4.y; bv[u][4 * q + 2] = b4.z; bv[u][4 * q + 3] = b4.w;
      }
      cst[u] = *reinterpret_cast<const float*>(stage + 9216 + (64 + u) * 4);
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int b3 = 0; b3 < 3; ++b3) a[s3][b3] = emb_cast(*reinterpret_cast<const uint4*>(stage + ((s3 * 3 + b3) * 64 + lane) * 16));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage_load(sp + 1);                                              // (behind the last pair: a spare record)
    __builtin_amdgcn_sched_barrier(0);
    emf16_t acc[MT][3];                                              // [row tile][k, k', upper components of both]
    // the six products: (U term, x term)
    constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m][0] = bv[0]; acc[m][1] = bv[1]; acc[m][2] = zero16;
#pragma unroll
      for (int pi = 0; pi < 6; ++pi) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[pi]][0], xb[m][TB[pi]][0], acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[pi]][1], xb[m][TB[pi]][0], acc[m][1], 0, 0, 0);
        acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[pi]][2], xb[m][TB[pi]][1], acc[m][2], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // |y|^2 of the lane half's components: registers 0..7 are components 0..15 (dims 0..15 only),
    // registers 8..15 components 16..31 = the state's own block + its half of the shared block
    float pp[2][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // (scalar v_fmac_f32: the packed form, v_pk_add_f32 / v_pk_fma_f32 on register pairs, measured
        //  7 % SLOWER for the whole kernel -- 0.624 against 0.583 ms on the probe's operands)
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          p0 = fmaf(acc[m][u][r], acc[m][u][r], p0); p1 = fmaf(acc[m][u][r + 1], acc[m][u][r + 1], p1);
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          const float y0 = acc[m][u][8 + r] + acc[m][2][8 * u + r], y1 = acc[m][u][9 + r] + acc[m][2][8 * u + r + 1];
          p0 = fmaf(y0, y0, p0); p1 = fmaf(y1, y1, p1);
        }
        pp[u][m] = cst[u] - (p0 + p1);                               // the lane half's share of cst - |y|^2
      }
    }
    // the halves exchange: lanes 0..31 finish the pair's first state, lanes 32..63 its second
    const int kl = 2 * sp + hh;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float send = hh ? pp[0][m] : pp[1][m];
      const float keep = hh ? pp[1][m] : pp[0][m];
      const float recv = __shfl_xor(send, 32, 64);
      const int r = m * 32 + t;
      if (NH == 1 || sp < npair) tw[r * 64 + (kl ^ ((r & 15) << 2))] = keep + recv;
    }
    emb_step_barrier();
  }
  if constexpr (NH == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();                                                 // the other groups' columns of the tile
  }

  // ---- scaled epilogue: Eh = exp(ll) 2^-k, k = ceil(max_j ll / ln 2); 16 lanes per row of the
  //      wave's own tile, four adjacent states each
  const int sg = lane & 15, rq = lane >> 4;
  const double L2E = 1.4426950408889634074;
  const float l2e = 1.44269504f, fbig = 3.0e38f;
  for (int it = sh * (8 * MT / NH); it < (sh + 1) * (8 * MT / NH); ++it) {
    const int rl = it * 4 + rq;
    const int64_t g = g0 + wave * WR + rl;
    const float4 v4 = *reinterpret_cast<const float4*>(tw + rl * 64 + ((4 * sg) ^ ((rl & 15) << 2)));
    const int bf = __shfl(bflag[it >> 3], rl & 31, 64);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = v[i];
      x = (x != x || (bf & 1)) ? 0.0f : fminf(fmaxf(x, -fbig), fbig);
      v[i] = (4 * sg + i < K) ? x : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
    mx = row16_max_f32(mx);
    // k and the row constant mx log2 e - k in (-1, 0] are formed in double: a row far from every state
    // has |mx| ~ 1e14 and more, where a float product would miss the integer by millions
    const double kx = (fabsf(mx) < 1e30f) ? ceil((double)mx * L2E) : 0.0;
    const float fr = (float)fma((double)mx, L2E, -kx);
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(fmaf(v[i] - mx, l2e, fr));
    if (g < nrows) {
      float* orow = Eh + g * K + 4 * sg;
      if (K == 64) *reinterpret_cast<float4*>(orow) = make_float4(e[0], e[1], e[2], e[3]);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * sg + i < K) orow[i] = e[i];
      }
      if (sg == 0) kexp[g] = kx;
      if (ll0 && (bf & 2)) {
        double* o0 = ll0 + (g / Lm) * K + 4 * sg;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * sg + i < K) o0[i] = (double)v[i];
      }
    }
  }
}
template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_emission_bf16x3(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const char* __restrict__ uw, uint32_t flags,
    float* __restrict__ Eh, double* __restrict__ kexp, double* __restrict__ ll0,
    int64_t* __restrict__ starts_copy = nullptr, int nstarts = 0) {
  emission_bf16x3_body<MT, 1, 4>(obs, mask, starts, nrows, Lm, D, K, uw, flags, Eh, kexp, ll0, starts_copy, nstarts);
}
template <int NH, int RW>
__global__ __launch_bounds__(64 * RW * NH) void k_emission_bf16x3h(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const char* __restrict__ uw, uint32_t flags,
    float* __restrict__ Eh, double* __restrict__ kexp, double* __restrict__ ll0,
    int64_t* __restrict__ starts_copy = nullptr, int nstarts = 0) {
  emission_bf16x3_body<1, NH, RW>(obs, mask, starts, nrows, Lm, D, K, uw, flags, Eh, kexp, ll0, starts_copy, nstarts);
}

// ------------------------------------------------------------------------------------
//  K1e (round 5): the centred bf16 x 3 emission for 32 < D <= 64 and for wide models (K > 64; any
//  D <= 64 through this kernel's WIDE form).  Same arithmetic as K1d: ll[t][k] = c_k - |U_k x_t + b_k|^2,
//  x and U as three bf16 terms, six products in fp32 accumulators.  U (64 components x 64 dimensions,
//  lower triangular) is cut into 32 x 16 blocks; per state the blocks that are not identically zero are
//      B0: comps  0..31 x dims  0..15      B1: comps 32..63 x dims  0..15
//      B2: comps 32..63 x dims 16..31      B3: comps 32..63 x dims 32..47
//  and the two half blocks comps 16..31 x dims 16..31 and comps 48..63 x dims 48..63, which the two
//  states of a pair share (rows 0..15: first state, rows 16..31: second): S0, S1 -- ten MFMAs per pair
//  and product term instead of sixteen.  Record of a pair: 3 terms x 10 blocks x 1 KB, then 2 x 64 bias
//  floats and the two constants (EMD_REC = 32 KB, 32 chunks of 1 KB: four per wave).
//  Workgroup = 8 waves x 32 rows (one 32-row tile per wave: six accumulators = 96 registers), ONE per
//  CU; the pair records stream global -> LDS through two buffers (the ten blocks of a pair are read
//  just in time, three 16-byte LDS reads in front of their six MFMAs), one barrier per pair.
//  grid.y = group of 64 states.  WIDE: the plain log-likelihoods go out as float [rows][K] (the wide
//  models' scaling pass and sweeps follow); otherwise (K <= 64) the scaled epilogue of K1d.
// ------------------------------------------------------------------------------------
constexpr int EMD_BLK = 10;
constexpr int EMD_REC = 32768;
constexpr int EMD_BIAS = 3 * EMD_BLK * 1024;      // byte offset of the bias floats in a record
typedef unsigned emd_u4 __attribute__((ext_vector_type(4)));
template <bool WIDE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_emission_bf16x3d(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int D, int K,
    const char* __restrict__ uw, uint32_t flags,
    float* __restrict__ Eh, double* __restrict__ kexp, double* __restrict__ ll0) {
  constexpr int ROWS = 256;
  extern __shared__ uint4 smem4[];
  char* stage = reinterpret_cast<char*>(smem4);                      // [2][EMD_REC]
  float* tile_s = reinterpret_cast<float*>(stage + 2 * EMD_REC);     // [8 waves][32 rows][64 states]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 31, hh = lane >> 5;
  const int64_t g0 = (int64_t)blockIdx.x * ROWS;
  const int k0 = 64 * blockIdx.y;                                    // first state of this workgroup's group
  const int kg = K - k0 < 64 ? K - k0 : 64;                          // states in the group
  const int npair = (kg + 1) >> 1;
  const char* __restrict__ urec = uw + (size_t)(k0 >> 1) * EMD_REC;
  const bool d48 = D <= 48;                                          // (uniform) no dimensions 48..63: S1 is all zero

  auto stage_load = [&](int pr, int buf) {
    const char* src = urec + (size_t)pr * EMD_REC + lane * 16;
    char* dst = stage + buf * EMD_REC;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) emb_glds16(src + (wave + 8 * c4) * 1024, dst + (wave + 8 * c4) * 1024);
  };
  stage_load(0, 0);

  // ---- the lane's row: dimensions 16 c + 8 hh + e as three bf16 terms (B operands)
  embf8_t xb[3][4];
  int bflag;
  {
    const int64_t g = g0 + wave * 32 + t;
    const bool valid = g < nrows;
    const int64_t gg = valid ? g : 0;
    const int64_t wi = gg / Lm;
    const int tt = (int)(gg - wi * Lm);
    const int64_t orow = starts[wi] + tt;
    int bd = 0;
    if (valid && (flags & SVIHMM_MASK_AS_NAN) && mask) bd = mask[orow] != 0;
    const double* xp = obs + orow * D;
    bool isnan_ = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v[8];
      const int i0 = 16 * c + 8 * hh;
      if ((D & 7) == 0) {
        const bool ok = valid && i0 < D;
        const double2* x2 = reinterpret_cast<const double2*>(xp + (ok ? i0 : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double2 q2 = x2[e];
          v[2 * e] = ok ? q2.x : 0.0; v[2 * e + 1] = ok ? q2.y : 0.0;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = valid && i0 + e < D;
          const double x = xp[ok ? i0 + e : 0];
          v[e] = ok ? x : 0.0;
        }
      }
      uint32_t w3[3][4];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (v[e] != v[e]) { isnan_ = true; v[e] = 0.0; }
        uint32_t t3[3];
        bf16_split3f((float)v[e], t3);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          if (e & 1) w3[s3][e >> 1] |= t3[s3] << 16; else w3[s3][e >> 1] = t3[s3];
        }
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) xb[s3][c] = emb_cast(make_uint4(w3[s3][0], w3[s3][1], w3[s3][2], w3[s3][3]));
    }
    int bf = bd | (isnan_ ? 1 : 0);
    bf |= __shfl_xor(bf, 32, 64);
    bflag = bf | ((valid && tt == 0) ? 2 : 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                   // record 0 has landed

  float* tw = tile_s + wave * 32 * 64;
  const emf16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int TA[6] = {0, 0, 1, 0, 2, 1}, TB[6] = {0, 1, 0, 2, 0, 1};
  for (int sp = 0; sp < npair; ++sp) {
    const char* st = stage + (sp & 1) * EMD_REC;
    emf16_t a0[2], a1[2], s0 = zero16, s1 = zero16;                  // comps 0..31 / 32..63 of the two states; shared halves
    float cst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(st + EMD_BIAS + (64 * u + 8 * q + 4 * hh) * 4);
        const float4 c4 = *reinterpret_cast<const float4*>(st + EMD_BIAS + (64 * u + 32 + 8 * q + 4 * hh) * 4);
        a0[u][4 * q] = b4.x; a0[u][4 * q + 1] = b4.y; a0[u][4 * q + 2] = b4.z; a0[u][4 * q + 3] = b4.w;
        a1[u][4 * q] = c4.x; a1[u][4 * q + 1] = c4.y; a1[u][4 * q + 2] = c4.z; a1[u][4 * q + 3] = c4.w;
      }
      cst[u] = *reinterpret_cast<const float*>(st + EMD_BIAS + (128 + u) * 4);
    }
    // (the C++ LDS reads above sit in front of the copy: the compiler waits for an outstanding copy
    //  before any LDS load it can see)
    __builtin_amdgcn_sched_barrier(0);
    stage_load(sp + 1, (sp + 1) & 1);                                // (behind the last pair: a spare record)
    __builtin_amdgcn_sched_barrier(0);
    // The ten blocks, software-pipelined: block i + 1's three terms are requested (16-byte LDS reads)
    // before block i's six MFMAs.  The reads are inline asm with their own lgkmcnt waits: for a C++
    // LDS load behind the global -> LDS copy just issued the compiler would first wait for that copy.
    emd_u4 A[2][3];
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)const_cast<char*>(st) + lane * 16;
#define EMD_RD(SET, BLK) do {                                                                                   \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[SET][0]) : "v"(sbase), "n"((0 * EMD_BLK + BLK) * 1024)); \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[SET][1]) : "v"(sbase), "n"((1 * EMD_BLK + BLK) * 1024)); \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[SET][2]) : "v"(sbase), "n"((2 * EMD_BLK + BLK) * 1024)); \
    } while (0)
#define EMD_WAIT(SET, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A[SET][0]), "+v"(A[SET][1]), "+v"(A[SET][2]))
#define EMD_MM(SET, C, ACC) do {                                                                                \
      _Pragma("unroll") for (int pi = 0; pi < 6; ++pi)                                                          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(embf8_t, A[SET][TA[pi]]), xb[TB[pi]][C], ACC, 0, 0, 0); \
    } while (0)
    EMD_RD(0, 0);
    EMD_RD(1, 1); EMD_WAIT(0, 3); EMD_MM(0, 0, a0[0]);
    EMD_RD(0, 2); EMD_WAIT(1, 3); EMD_MM(1, 0, a0[1]);
    EMD_RD(1, 3); EMD_WAIT(0, 3); EMD_MM(0, 0, a1[0]);
    EMD_RD(0, 8); EMD_WAIT(1, 3); EMD_MM(1, 0, a1[1]);
    EMD_RD(1, 4); EMD_WAIT(0, 3); EMD_MM(0, 1, s0);
    EMD_RD(0, 5); EMD_WAIT(1, 3); EMD_MM(1, 1, a1[0]);
    EMD_RD(1, 6); EMD_WAIT(0, 3); EMD_MM(0, 1, a1[1]);
    EMD_RD(0, 7); EMD_WAIT(1, 3); EMD_MM(1, 2, a1[0]);
    if (!d48) {
      EMD_RD(1, 9); EMD_WAIT(0, 3); EMD_MM(0, 2, a1[1]);
      EMD_WAIT(1, 0); EMD_MM(1, 3, s1);
    } else {
      EMD_WAIT(0, 0); EMD_MM(0, 2, a1[1]);
    }
#undef EMD_RD
#undef EMD_WAIT
#undef EMD_MM
    // |y|^2 of the lane half's components (registers 0..7: the block's rows 0..15, 8..15: rows 16..31)
    float pp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        p0 = fmaf(a0[u][r], a0[u][r], p0); p1 = fmaf(a0[u][r + 1], a0[u][r + 1], p1);
        p0 = fmaf(a1[u][r], a1[u][r], p0); p1 = fmaf(a1[u][r + 1], a1[u][r + 1], p1);
      }
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        const float y0 = a0[u][8 + r] + s0[8 * u + r], y1 = a0[u][9 + r] + s0[8 * u + r + 1];
        const float z0 = a1[u][8 + r] + s1[8 * u + r], z1 = a1[u][9 + r] + s1[8 * u + r + 1];
        p0 = fmaf(y0, y0, p0); p1 = fmaf(y1, y1, p1);
        p0 = fmaf(z0, z0, p0); p1 = fmaf(z1, z1, p1);
      }
      pp[u] = cst[u] - (p0 + p1);
    }
    // the halves exchange: lanes 0..31 finish the pair's first state, lanes 32..63 its second
    {
      const int kl = 2 * sp + hh;
      const float send = hh ? pp[0] : pp[1];
      const float keep = hh ? pp[1] : pp[0];
      const float recv = __shfl_xor(send, 32, 64);
      tw[t * 64 + (kl ^ ((t & 15) << 2))] = keep + recv;
    }
    emb_step_barrier();                                              // next record landed, this one free
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- epilogue: 16 lanes per row of the wave's own tile, four adjacent states each
  const int sg = lane & 15, rq = lane >> 4;
  const double L2E = 1.4426950408889634074;
  const float l2e = 1.44269504f, fbig = 3.0e38f;
  for (int it = 0; it < 8; ++it) {
    const int rl = it * 4 + rq;
    const int64_t g = g0 + wave * 32 + rl;
    const float4 v4 = *reinterpret_cast<const float4*>(tw + rl * 64 + ((4 * sg) ^ ((rl & 15) << 2)));
    const int bf = __shfl(bflag, rl, 64);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = v[i];
      x = (x != x || (bf & 1)) ? 0.0f : fminf(fmaxf(x, -fbig), fbig);
      v[i] = (WIDE || 4 * sg + i < kg) ? x : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
    if (WIDE) {
      // plain log-likelihoods of this group's states (nan_to_num'ed; rows masked as missing: 0)
      if (g < nrows) {
        float* orow = Eh + g * K + k0 + 4 * sg;
        if (kg == 64 && (K & 3) == 0) *reinterpret_cast<float4*>(orow) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) if (4 * sg + i < kg) orow[i] = v[i];
        }
        if (ll0 && (bf & 2)) {
          double* o0 = ll0 + (g / Lm) * K + k0 + 4 * sg;
#pragma unroll
          for (int i = 0; i < 4; ++i) if (4 * sg + i < kg) o0[i] = (double)v[i];
        }
      }
      continue;
    }
    mx = row16_max_f32(mx);
    const double kx = (fabsf(mx) < 1e30f) ? ceil((double)mx * L2E) : 0.0;
    const float fr = (float)fma((double)mx, L2E, -kx);
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(fmaf(v[i] - mx, l2e, fr));
    if (g < nrows) {
      float* orow = Eh + g * K + 4 * sg;
      if (K == 64) *reinterpret_cast<float4*>(orow) = make_float4(e[0], e[1], e[2], e[3]);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * sg + i < K) orow[i] = e[i];
      }
      if (sg == 0) kexp[g] = kx;
      if (ll0 && (bf & 2)) {
        double* o0 = ll0 + (g / Lm) * K + 4 * sg;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * sg + i < K) o0[i] = (double)v[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
//  K0: NIW mean-field factors -> theta (one workgroup per state).  Cholesky of sigma_mf,
//      W = (nu/2) sigma^-1 = (nu/2) L^-T L^-1, E log|Lambda| (digamma), linear and constant
//      terms of the quadratic form.  status[0] = 1 + k if sigma_k is not positive definite.
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
//  K0d: diagonal-covariance Gaussian factors (pybasicbayes DiagonalGaussian: per dimension a
//       normal-inverse-gamma mean-field factor  sigma_d^2 ~ InvGamma(alpha_d, beta_d),
//       mu_d | sigma_d^2 ~ N(m_d, sigma_d^2 / nu_d))  ->  theta in the DIAGONAL feature order
//       (upload_feature_table: f < D: x_f^2, D <= f < 2D: x_{f-D}, f = 2D: 1):
//         E_q log N(x | mu, diag sigma^2) = sum_d [ -1/2 (alpha_d/beta_d) x_d^2 + m_d (alpha_d/beta_d) x_d
//            - 1/2 (1/nu_d + m_d^2 alpha_d/beta_d) - 1/2 (log beta_d - psi(alpha_d)) ] - D/2 log 2 pi.
//       2 D + 1 features instead of (D+1)(D+2)/2: the same table-driven GEMM kernels (emission,
//       statistics) run it, at D = 32 on 80 instead of 576 feature rows.
//       One wavefront per state; status as k_niw_to_theta (1 + k: a non-positive parameter;
//       NIW_STATUS_RANGE + 1 + k: the factor lies too far from the data's centre).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void k_diag_to_theta_body(
    const double* __restrict__ mu, const double* __restrict__ nus, const double* __restrict__ alphas,
    const double* __restrict__ betas, int K, int D, int Kp, double* __restrict__ theta,
    int* __restrict__ status) {
  const int k = blockIdx.x, lane = threadIdx.x;
  double cs = 0.0, mwm = 0.0;
  bool bad = false;
  for (int d = lane; d < D; d += 64) {
    const size_t e = (size_t)k * D + d;
    const double m = mu[e], nu = nus[e], a = alphas[e], b = betas[e];
    if (!(nu > 0.0) || !(a > 0.0) || !(b > 0.0) || !(m == m)) bad = true;
    const double prec = a / b;
    theta[(size_t)d * Kp + k] = -0.5 * prec;
    theta[(size_t)(D + d) * Kp + k] = m * prec;
    cs += -0.5 * (1.0 / nu + m * m * prec) - 0.5 * (log(b) - digamma_d(a));
    mwm += 0.5 * m * m * prec;
  }
  cs = wave_sum(cs);
  mwm = wave_sum(mwm);
  const bool anybad = __any(bad);
  if (lane == 0) {
    theta[(size_t)(2 * D) * Kp + k] = cs - 0.5 * D * 1.8378770664093453;   // log(2 pi)
    if (anybad) atomicMax(status, 1 + k);
    else if (mwm > NIW_CANCEL_LIMIT) atomicMax(status, NIW_STATUS_RANGE + 1 + k);
  }
}
__global__ __launch_bounds__(64) void k_diag_to_theta(
    const double* __restrict__ mu, const double* __restrict__ nus, const double* __restrict__ alphas,
    const double* __restrict__ betas, int K, int D, int Kp, double* __restrict__ theta,
    int* __restrict__ status,
    SviSync sy = SviSync{nullptr, 0u, nullptr, nullptr, nullptr}) {
  if (svi_gate(sy)) k_diag_to_theta_body(mu, nus, alphas, betas, K, D, Kp, theta, status);
  svi_arrive(sy);
}

// generic D (workgroup per state, matrices in LDS): used for D > 64 only
__device__ __forceinline__ void k_niw_to_theta_generic_body(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out) {
  extern __shared__ double sm[];
  const int S = D + 1;
  double* Lm_ = sm;            // [D][S] Cholesky factor (lower); once its inverse stands: W
  double* Li = Lm_ + D * S;    // [D][S] its inverse (lower)
  double* W = Lm_;             // [D][S] (nu/2) sigma^-1 takes the factor's place (its diagonal is kept in ld)
  double* wm = Li + D * S;     // [D]
  double* ld = wm + D;         // [D] diagonal of the Cholesky factor
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
  for (int i = tid; i < D; i += nt) ld[i] = Lm_[i * S + i];
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
    theta_store(theta, orb, i, D, D, Kp, k, 2.0 * s);
  }
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    if (j < i) continue;
    theta_store(theta, orb, i, j, D, Kp, k, (i == j) ? -W[i * S + i] : -2.0 * W[i * S + j]);
  }
  __syncthreads();
  if (tid == 0) {
    double logdet = 0.0, llt = D * log(2.0), mWm = 0.0;
    for (int i = 0; i < D; ++i) {
      logdet += log(ld[i]);
      llt += digamma_d(0.5 * (nu[k] - i));
      mWm += m[i] * wm[i];
    }
    llt -= 2.0 * logdet;
    const double cst = 0.5 * llt - D / (2.0 * kappa[k]) - 0.5 * D * 1.8378770664093454835606594728112;
    theta_store(theta, orb, D, D, D, Kp, k, cst - mWm);
    if (logdet_out) logdet_out[k] = 2.0 * logdet;     // log det sigma_mf (ELBO terms)
    if (mWm > NIW_CANCEL_LIMIT) atomicMax(status, NIW_STATUS_RANGE + 1 + k);
  }
}
__global__ __launch_bounds__(256) void k_niw_to_theta_generic(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out,
    SviSync sy = SviSync{nullptr, 0u, nullptr, nullptr, nullptr}) {
  if (svi_gate(sy)) k_niw_to_theta_generic_body(mu, sigma, kappa, nu, K, D, Kp, theta, status, orb, logdet_out);
  svi_arrive(sy);
}

// D <= DMAX <= 64: one wavefront per state, lane a owns row a of sigma / column a of L^-1 in
// registers (loops fully unrolled, so all register indices are static).  Cross-lane data
// moves as LDS broadcasts (every lane reads the same address): the D dependent Cholesky
// columns cost one v_readlane + one LDS round trip each instead of workgroup barriers, and
// the triangular inverse / Gram product stream independent broadcast reads.  ~10x faster
// than the generic kernel at D = 32; it sits on the E-step's critical path every iteration.
template <int DMAX>
__device__ __forceinline__ void k_niw_to_theta_wave_body(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out, uint4* __restrict__ uw) {
  __shared__ double col[64];
  __shared__ double Ls[DMAX][DMAX + 1];    // L (row-major), later X = L^-1 stored as Ls[c][r]
  __shared__ double ms[DMAX];
  const int k = blockIdx.x, a = threadIdx.x;
  const bool va = a < D;
  const double* Sg = sigma + (size_t)k * D * D;
  double A[DMAX];
#pragma unroll
  for (int j = 0; j < DMAX; ++j) A[j] = (va && j < D) ? Sg[(size_t)a * D + j] : (a == j ? 1.0 : 0.0);
  if (a < DMAX) ms[a] = va ? mu[(size_t)k * D + a] : 0.0;
  // ---- right-looking Cholesky; padded rows/columns form an identity block
  bool bad = false;
  double rdv[DMAX];                       // 1 / L[j][j] (uniform): the triangular inverse multiplies by it
#pragma unroll
  for (int j = 0; j < DMAX; ++j) {
    const double djj = readlane_f64(A[j], j);
    bad |= !(djj > 0.0);
    // sqrt and reciprocal sqrt together: v_rsq_f64 + two coupled Newton steps (g -> sqrt x, hh -> 1 / (2 sqrt x))
    // and a last correction of g: ~12 instructions in place of the ~40 of sqrt() and a division
    double d, rd;
    {
      const double y0 = __builtin_amdgcn_rsq(djj);
      double g = djj * y0, hh = 0.5 * y0;
      double r = fma(-g, hh, 0.5);
      g = fma(g, r, g); hh = fma(hh, r, hh);
      r = fma(-g, hh, 0.5);
      g = fma(g, r, g); hh = fma(hh, r, hh);
      g = fma(fma(-g, g, djj), hh, g);
      d = g; rd = hh + hh;
    }
    rdv[j] = rd;
    const double l = (a == j) ? d : A[j] * rd;
    A[j] = l;
    col[a] = l;
    __syncthreads();
#pragma unroll
    for (int b = j + 1; b < DMAX; ++b) A[b] = fma(-l, col[b], A[b]);   // upper part: unused garbage
    __syncthreads();
  }
  if (bad) {   // uniform: djj is a broadcast value
    if (a == 0 && status) atomicMax(status, 1 + k);
    return;
  }
  double logdet;
  {   // the lane's diagonal element through a select chain (no dynamic register index)
    double dg = 1.0;
#pragma unroll
    for (int j = 0; j < DMAX; ++j) dg = (a == j) ? A[j] : dg;
    logdet = va ? log(dg) : 0.0;
  }
  if (a < DMAX) {
#pragma unroll
    for (int j = 0; j < DMAX; ++j) Ls[a][j] = A[j];
  }
  __syncthreads();
  // ---- X = L^-1, lane c owns column c: X[r][c] = -(sum_{j=c}^{r-1} L[r][j] X[j][c]) / L[r][r]
  double X[DMAX];
  const int c = a;
#pragma unroll
  for (int r = 0; r < DMAX; ++r) {
    double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < r; ++j) s = fma(-Ls[r][j], X[j], s);   // X[j] = 0 for j < c
    X[r] = (r >= c) ? s * rdv[r] : 0.0;
  }
  __syncthreads();
  if (a < DMAX) {
#pragma unroll
    for (int r = 0; r < DMAX; ++r) Ls[a][r] = X[r];            // Ls[c][r] = X[r][c]
  }
  __syncthreads();
  // ---- W = (nu/2) X^T X, lane i owns row i; theta entries
  const double hn = 0.5 * nu[k];
  if constexpr (DMAX <= 32) {
    // centred factor for the fp32-mode emission kernel (k_emission_bf16x3): U = sqrt(nu/2) L^-1 as
    // three bf16 terms per entry in the A-operand order of v_mfma_f32_32x32x16_bf16 (lane = 32 h + j,
    // entries e = 0..7: U[j][16 c + 8 h + e]), bias b = -U m in fp32
    if (uw) {
      const double shn = sqrt(hn);
      const int j = a & 31, hh = a >> 5, u = k & 1;
      uint4* blk = uw + (size_t)(k >> 1) * (EMB_REC / 16);
      float* ub = reinterpret_cast<float*>(blk + EMB_BLOCKS * 64);
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        uint32_t w3[3][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cl = 16 * cc + 8 * hh + e;
          const double v = (j < D && cl < D) ? shn * Ls[cl < DMAX ? cl : 0][j < DMAX ? j : 0] : 0.0;   // X[j][cl]
          uint32_t t3[3];
          bf16_split3(v, t3);
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3) {
            if (e & 1) w3[s3][e >> 1] |= t3[s3] << 16; else w3[s3][e >> 1] = t3[s3];
          }
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          const uint4 q = make_uint4(w3[s3][0], w3[s3][1], w3[s3][2], w3[s3][3]);
          if (cc == 0) blk[(s3 * 3 + u) * 64 + a] = q;                                   // dims 0..15, all components
          else if (j >= 16) blk[(s3 * 3 + 2) * 64 + 32 * hh + 16 * u + (j - 16)] = q;    // dims 16..31, components 16..31
        }
      }
      if (a < 32) {
        double b = 0.0;
        if (a < D) {
#pragma unroll
          for (int cl = 0; cl < DMAX; ++cl) b = fma(Ls[cl][a < DMAX ? a : 0], ms[cl], b);           // sum_c X[a][c] m_c
        }
        ub[32 * u + a] = (float)(-shn * b);
      }
    }
  }
  if constexpr (DMAX == 64) {
    // centred factor in the record layout of k_emission_bf16x3d (32 < D <= 64, wide models): lane
    // 32 hh + jj writes, for both component blocks cb (component 32 cb + jj) and the four dimension
    // blocks c (entries e = 0..7: U[comp][16 c + 8 hh + e]), the 16 bytes of the blocks that are not
    // identically zero: B0 (cb 0, c 0), B1..B3 (cb 1, c 0..2), and its 16 rows of the half blocks the
    // pair shares, S0 (cb 0, c 1, components 16..31) and S1 (cb 1, c 3, components 48..63)
    if (uw) {
      const double shn = sqrt(hn);
      const int jj = a & 31, hh = a >> 5, u = k & 1;
      uint4* blk = uw + (size_t)(k >> 1) * (EMD_REC / 16);
      float* ub = reinterpret_cast<float*>(reinterpret_cast<char*>(blk) + EMD_BIAS);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (cb == 0 && c >= 2) continue;
          const bool half = (cb == 0 && c == 1) || (cb == 1 && c == 3);
          const int idx = cb == 0 ? (c == 0 ? u : 8) : (c < 3 ? 2 + 2 * c + u : 9);
          const int pos = half ? 32 * hh + 16 * u + (jj - 16) : a;
          const int comp = 32 * cb + jj;
          uint32_t w3[3][4];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int cl = 16 * c + 8 * hh + e;
            const double v = (comp < D && cl < D) ? shn * Ls[cl][comp] : 0.0;      // X[comp][cl]
            uint32_t t3[3];
            bf16_split3(v, t3);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
              if (e & 1) w3[s3][e >> 1] |= t3[s3] << 16; else w3[s3][e >> 1] = t3[s3];
            }
          }
          if (!half || jj >= 16) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
              blk[(s3 * EMD_BLK + idx) * 64 + pos] = make_uint4(w3[s3][0], w3[s3][1], w3[s3][2], w3[s3][3]);
          }
        }
      }
      double b = 0.0;
      if (a < D) {
#pragma unroll
        for (int cl = 0; cl < DMAX; ++cl) b = fma(Ls[cl][a], ms[cl], b);                  // sum_c X[a][c] m_c
      }
      ub[64 * u + a] = (float)(-shn * b);
    }
  }
  double wmi = 0.0;
  const int i = a;
  // theta_store's index arithmetic, once per lane and per column instead of once per entry (the kernel is one
  // wave issuing ~8000 instructions in a row; the per-entry divisions were a third of them).  Canonical row of
  // the pair (i, j): fbase + j; orbit row: with S4 = 4 c Kp and t4(a) = 4 (a mod c) + a / c,
  //   j - i <= D/2:  (t4(i) Kp + kk - i S4) + j S4        j - i > D/2:  (i S4 + kk) + (N - j) S4 + t4(j) Kp
  const int Nn = D + 1, cq = (D >> 2) > 0 ? (D >> 2) : 1, S4 = 4 * cq * Kp;
  const int kk = (k & 15) * (Kp >> 4) + (k >> 4);
  const int t4 = 4 * (a % cq) + a / cq;
  const int fbase = (a * (D + 1) - a * (a - 1) / 2 - a) * Kp + k;
  const int oA = t4 * Kp + kk - a * S4, oB = a * S4 + kk;
#pragma unroll
  for (int j = 0; j < DMAX; ++j) {
    double s = 0.0;
#pragma unroll
    for (int r = j; r < DMAX; ++r) s = fma(X[r], Ls[j][r], s);  // X[r][i] * X[r][j]; zero for r < max(i,j)
    const double w = hn * s;
    wmi = fma(w, ms[j], wmi);
    if (theta && va && j >= i && j < D) {
      const double tv = (i == j) ? -w : -2.0 * w;
      theta[fbase + j * Kp] = tv;
      if (orb) {
        const int t4j = __builtin_amdgcn_readlane(t4, j);
        orb[(j - i <= (D >> 1)) ? oA + j * S4 : oB + (Nn - j) * S4 + t4j * Kp] = tv;
      }
    }
  }
  if (theta && va) {                      // the linear term (i, D): column D is the orbit's leftover block
    const double tv = 2.0 * wmi;
    theta[fbase + D * Kp] = tv;
    if (orb) {
      const int dl = i + 1;
      orb[(D - i <= (D >> 1)) ? oA + D * S4
                              : (4 * (cq * ((D >> 1) + 1) + (dl >> 2)) + (dl & 3)) * Kp + kk] = tv;
    }
  }
  // ---- constant term
  double dgm = va ? digamma_d(0.5 * (nu[k] - a)) : 0.0;
  double mWm = va ? ms[a < DMAX ? a : 0] * wmi : 0.0;
  logdet = wave_sum(logdet); dgm = wave_sum(dgm); mWm = wave_sum(mWm);
  if (a == 0) {
    const double llt = D * log(2.0) + dgm - 2.0 * logdet;
    const double cst = 0.5 * llt - D / (2.0 * kappa[k]) - 0.5 * D * 1.8378770664093454835606594728112;
    if (theta) theta_store(theta, orb, D, D, D, Kp, k, cst - mWm);
    if (logdet_out) logdet_out[k] = 2.0 * logdet;
    if (status && mWm > NIW_CANCEL_LIMIT) atomicMax(status, NIW_STATUS_RANGE + 1 + k);
    if constexpr (DMAX <= 32) {   // ll = cst - |U x + b|^2; each half of the wave adds its share of the squares to cst / 2
      if (uw) reinterpret_cast<float*>(uw + (size_t)(k >> 1) * (EMB_REC / 16) + EMB_BLOCKS * 64)[64 + (k & 1)] = (float)(0.5 * cst);
    }
    if constexpr (DMAX == 64) {
      if (uw) reinterpret_cast<float*>(reinterpret_cast<char*>(uw + (size_t)(k >> 1) * (EMD_REC / 16)) + EMD_BIAS)[128 + (k & 1)] = (float)(0.5 * cst);
    }
  }
}
// D <= 32 (round 6): the same factorisation with BOTH halves of the wave at work.  The one-wave builder above leaves
// lanes 32..63 idle at DMAX = 32 -- and it is 25 us of the S = 64 iteration's critical path (one wave per state, ~6900
// dependent-ish instructions).  Here lane (r = a & 31, h = a >> 5) owns the columns b = 2 i + h of row r in the Cholesky
// sweep (half the rank-one updates and LDS broadcasts per column step) and the columns j = 2 jj + h of row r in the Gram
// product W = (nu / 2) X'X (half the dot products); the triangular inverse runs in both halves alike (32 columns, one
// per lane: nothing to split).  Every ELEMENT goes through the same operations in the same order as above -- the
// products of a row with the mean are summed in column order from LDS -- so theta, log det and the status word come
// out bit for bit the same (tests/test_emission_formula.py pins them to SciPy; tests/test_gpu_theta_split.py compares
// the two builders directly).
__device__ __forceinline__ void k_niw_to_theta_wave32s_body(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out, uint4* __restrict__ uw) {
  constexpr int DMAX = 32;
  __shared__ double col[64];
  __shared__ double Ls[DMAX][DMAX + 1];    // L (row-major), later X = L^-1 stored as Ls[c][r]
  __shared__ double Wr[DMAX][DMAX + 1];    // W rows
  __shared__ double ms[DMAX];
  const int k = blockIdx.x, a = threadIdx.x, hh = a >> 5, r_ = a & 31;
  const bool vr = r_ < D, va = a < D;
  const double* Sg = sigma + (size_t)k * D * D;
  double Ah[16];                           // A[r_][2 i + hh]
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int b = 2 * i + hh;
    Ah[i] = (vr && b < D) ? Sg[(size_t)r_ * D + b] : (r_ == b ? 1.0 : 0.0);
  }
  if (a < DMAX) ms[a] = va ? mu[(size_t)k * D + a] : 0.0;
  // ---- right-looking Cholesky; padded rows/columns form an identity block
  bool bad = false;
  double rdv[DMAX];
#pragma unroll
  for (int j = 0; j < DMAX; ++j) {
    const double djj = readlane_f64(Ah[j >> 1], j + 32 * (j & 1));   // A[j][j]: row j, half j & 1
    bad |= !(djj > 0.0);
    double d, rd;
    {
      const double y0 = __builtin_amdgcn_rsq(djj);
      double g = djj * y0, hf = 0.5 * y0;
      double r = fma(-g, hf, 0.5);
      g = fma(g, r, g); hf = fma(hf, r, hf);
      r = fma(-g, hf, 0.5);
      g = fma(g, r, g); hf = fma(hf, r, hf);
      g = fma(fma(-g, g, djj), hf, g);
      d = g; rd = hf + hf;
    }
    rdv[j] = rd;
    if (hh == (j & 1)) {
      const double lo = (r_ == j) ? d : Ah[j >> 1] * rd;
      Ah[j >> 1] = lo;
      col[r_] = lo;
    }
    __syncthreads();
    const double l = col[r_];
    const double* cb = col + hh;
    if ((j & 1) == 0) {                    // column j + 1 = 2 (j / 2) + 1: the odd half's entry of slot j / 2
      const double t = fma(-l, col[j + 1 < DMAX ? j + 1 : j], Ah[j >> 1]);
      Ah[j >> 1] = hh ? t : Ah[j >> 1];
    }
#pragma unroll
    for (int i = (j >> 1) + 1; i < 16; ++i) Ah[i] = fma(-l, cb[2 * i], Ah[i]);   // upper part: unused garbage
    __syncthreads();
  }
  if (bad) {   // uniform: djj is a broadcast value
    if (a == 0 && status) atomicMax(status, 1 + k);
    return;
  }
  double logdet;
  {   // the diagonal element of row r_ lives in half r_ & 1: handed to lane r_ through LDS
    double dgh = 1.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) dgh = ((r_ >> 1) == i) ? Ah[i] : dgh;
    if (hh == (r_ & 1)) col[r_] = dgh;
    __syncthreads();
    logdet = va ? log(col[a < DMAX ? a : 0]) : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) Ls[r_][2 * i + hh] = Ah[i];
  __syncthreads();
  // ---- X = L^-1, column c = r_ (both halves compute it): X[r][c] = -(sum_{j=c}^{r-1} L[r][j] X[j][c]) / L[r][r]
  double X[DMAX];
  const int c = r_;
#pragma unroll
  for (int r = 0; r < DMAX; ++r) {
    double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < r; ++j) s = fma(-Ls[r][j], X[j], s);   // X[j] = 0 for j < c
    X[r] = (r >= c) ? s * rdv[r] : 0.0;
  }
  __syncthreads();
  if (a < DMAX) {
#pragma unroll
    for (int r = 0; r < DMAX; ++r) Ls[a][r] = X[r];            // Ls[c][r] = X[r][c]
  }
  __syncthreads();
  const double hn = 0.5 * nu[k];
  if (uw) {
    // centred factor for the fp32-mode emission kernel (see k_niw_to_theta_wave_body: same record, same values)
    const double shn = sqrt(hn);
    const int j = a & 31, u = k & 1;
    uint4* blk = uw + (size_t)(k >> 1) * (EMB_REC / 16);
    float* ub = reinterpret_cast<float*>(blk + EMB_BLOCKS * 64);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      uint32_t w3[3][4];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cl = 16 * cc + 8 * hh + e;
        const double v = (j < D && cl < D) ? shn * Ls[cl < DMAX ? cl : 0][j < DMAX ? j : 0] : 0.0;   // X[j][cl]
        uint32_t t3[3];
        bf16_split3(v, t3);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          if (e & 1) w3[s3][e >> 1] |= t3[s3] << 16; else w3[s3][e >> 1] = t3[s3];
        }
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const uint4 q = make_uint4(w3[s3][0], w3[s3][1], w3[s3][2], w3[s3][3]);
        if (cc == 0) blk[(s3 * 3 + u) * 64 + a] = q;
        else if (j >= 16) blk[(s3 * 3 + 2) * 64 + 32 * hh + 16 * u + (j - 16)] = q;
      }
    }
    if (a < 32) {
      double b = 0.0;
      if (a < D) {
#pragma unroll
        for (int cl = 0; cl < DMAX; ++cl) b = fma(Ls[cl][a < DMAX ? a : 0], ms[cl], b);
      }
      ub[32 * u + a] = (float)(-shn * b);
    }
  }
  // ---- W = (nu/2) X^T X: lane (i = r_, hh) forms W[i][j] for the columns j = 2 jj + hh
  const int i = r_;
  const int Nn = D + 1, cq = (D >> 2) > 0 ? (D >> 2) : 1, S4 = 4 * cq * Kp;
  const int kk = (k & 15) * (Kp >> 4) + (k >> 4);
  const int t4 = 4 * (i % cq) + i / cq;
  const int fbase = (i * (D + 1) - i * (i - 1) / 2 - i) * Kp + k;
  const int oA = t4 * Kp + kk - i * S4, oB = i * S4 + kk;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int j = 2 * jj + hh;
    const double* lj = &Ls[0][0] + j * (DMAX + 1);
    double s = 0.0;
    {
      const double t = fma(X[2 * jj], lj[2 * jj], 0.0);       // r = 2 jj: part of the sum for the even column only
      s = hh ? 0.0 : t;
    }
#pragma unroll
    for (int r = 2 * jj + 1; r < DMAX; ++r) s = fma(X[r], lj[r], s);   // X[r][i] * X[r][j]; zero for r < max(i,j)
    const double w = hn * s;
    Wr[i][j] = w;
    if (theta && vr && j >= i && j < D) {
      const double tv = (i == j) ? -w : -2.0 * w;
      theta[fbase + j * Kp] = tv;
      if (orb) {
        const int t4j = 4 * (j % cq) + j / cq;
        orb[(j - i <= (D >> 1)) ? oA + j * S4 : oB + (Nn - j) * S4 + t4j * Kp] = tv;
      }
    }
  }
  __syncthreads();
  double wmi = 0.0;
  if (a < DMAX) {
#pragma unroll
    for (int j = 0; j < DMAX; ++j) wmi = fma(Wr[a][j], ms[j], wmi);
  }
  if (theta && va) {                      // the linear term (i, D): column D is the orbit's leftover block
    const double tv = 2.0 * wmi;
    theta[fbase + D * Kp] = tv;
    if (orb) {
      const int dl = i + 1;
      orb[(D - i <= (D >> 1)) ? oA + D * S4
                              : (4 * (cq * ((D >> 1) + 1) + (dl >> 2)) + (dl & 3)) * Kp + kk] = tv;
    }
  }
  // ---- constant term
  double dgm = va ? digamma_d(0.5 * (nu[k] - a)) : 0.0;
  double mWm = va ? ms[a < DMAX ? a : 0] * wmi : 0.0;
  logdet = wave_sum(logdet); dgm = wave_sum(dgm); mWm = wave_sum(mWm);
  if (a == 0) {
    const double llt = D * log(2.0) + dgm - 2.0 * logdet;
    const double cst = 0.5 * llt - D / (2.0 * kappa[k]) - 0.5 * D * 1.8378770664093454835606594728112;
    if (theta) theta_store(theta, orb, D, D, D, Kp, k, cst - mWm);
    if (logdet_out) logdet_out[k] = 2.0 * logdet;
    if (status && mWm > NIW_CANCEL_LIMIT) atomicMax(status, NIW_STATUS_RANGE + 1 + k);
    if (uw) reinterpret_cast<float*>(uw + (size_t)(k >> 1) * (EMB_REC / 16) + EMB_BLOCKS * 64)[64 + (k & 1)] = (float)(0.5 * cst);
  }
}
// The natural-gradient global step of the resident SVI loop AND the theta builder in one launch (round 6; VERDICT r5
// next #1b): workgroup k < K updates NIW factor k (hmmsgd_metaobs.py:1048-1069, util.py:28-60 -- the expressions of
// k_svi_global_step, kernels_svi.h, term by term) with the sigma' entries already where the split Cholesky wants them
// (lane (r, h): columns 2 i + h of row r), writes the factor back for the ELBO kernels and the next step, arrives on the
// step counter (the side stream's globals kernel may go), and carries straight on into the factorisation; workgroups
// K .. K + ceil(K^2 / 64) - 1 update the transition factor.  One kernel boundary and the 6-9 us step kernel less on the
// iteration's chain; the separate kernels remain for every other family / shape.
__global__ __launch_bounds__(64) void k_svi_step_theta32s(
    const double* __restrict__ packed, const double* __restrict__ prior_tran, double* __restrict__ var_tran,
    double* __restrict__ niw, const double* __restrict__ prior, int K, int D, double rho, double bA, double bE,
    double nwin, double* __restrict__ lb_keep, double* __restrict__ ada_G, SviSync ssy,
    int Kp, double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out, uint4* __restrict__ uw, SviSync tsy) {
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  const int a = threadIdx.x;
  const bool go = svi_step_gate(ssy);
  if ((int)blockIdx.x >= K) {
    if (go) {
      const int e = ((int)blockIdx.x - K) * 64 + a;
      if (e == 0) *lb_keep = packed[(size_t)K * K + nmu + K + nsg];
      svi_tran_step(e, K, packed, prior_tran, var_tran, rho, bA, nwin, ada_G);
    }
    svi_arrive(ssy);
    return;
  }
  const int k = blockIdx.x;
  double* mu = niw + (size_t)k * D;
  double* sg = niw + nmu + (size_t)k * D * D;
  double* kap = niw + nmu + nsg;
  double* nu = kap + K;
  if (go) {
    __shared__ double mo[32], mn[32], m0[32];
    const int hh = a >> 5, r_ = a & 31;
    const double* mu0 = prior + (size_t)k * D;
    const double* sg0 = prior + nmu + (size_t)k * D * D;
    const double ka0 = prior[nmu + nsg + k], nu0 = prior[nmu + nsg + K + k];
    const double* xbar = packed + (size_t)K * K + (size_t)k * D;
    const double neff = packed[(size_t)K * K + nmu + k];
    const double* S = packed + (size_t)K * K + nmu + K + (size_t)k * D * D;
    const double ka = kap[k], nuo = nu[k];
    const int ac = a < D ? a : 0;
    const double m_ld = mu[ac], p_ld = mu0[ac], x_ld = xbar[ac];
    const double e2 = (1.0 - rho) * ka + rho * (ka0 + bE * neff);                        // kappa'
    const double e4 = (1.0 - rho) * (nuo + 2 + D) + rho * ((nu0 + 2 + D) + bE * neff);
    if (a < D) {
      const double m = m_ld, p = p_ld;
      mo[a] = m; m0[a] = p;
      mn[a] = ((1.0 - rho) * (ka * m) + rho * (ka0 * p + bE * x_ld)) / e2;               // mu' = e1 / e2
    }
    __syncthreads();
    {
      // (all 48 loads of the lane's 16 entries in flight at once -- clamped addresses, the stores predicated: one
      //  memory round trip instead of one per entry)
      const int rr = r_ < D ? r_ : 0;
      double vs[16], v0[16], vS[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int b = 2 * i + hh, e = rr * D + (b < D ? b : 0);
        vs[i] = sg[e]; v0[i] = sg0[e]; vS[i] = S[e];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int b = 2 * i + hh, bc = b < D ? b : 0;
        const double e3o = vs[i] + (mo[rr] * mo[bc]) * ka;
        const double e3p = v0[i] + (m0[rr] * m0[bc]) * ka0;
        const double e3 = (1.0 - rho) * e3o + rho * (e3p + bE * vS[i]);
        if (r_ < D && b < D) sg[rr * D + b] = e3 - (mn[rr] * mn[bc]) * e2;                  // sigma'
      }
    }
    __syncthreads();
    if (a < D) mu[a] = mn[a];
    if (a == 0) { kap[k] = e2; nu[k] = e4 - 2 - D; }
  }
  // (only the transition workgroups arrive on the step counter: what waits on it -- the side stream's globals kernel, 60 us
  //  that must be over before the next sweeps -- needs var_tran alone and should not wait for the K factor updates)
  // theta from the factor just written (this workgroup's own stores, in program order)
  __syncthreads();
  k_niw_to_theta_wave32s_body(niw, niw + nmu, kap, nu, K, D, Kp, theta, status, orb, logdet_out, uw);
  svi_arrive(tsy);
}
// SPLIT: 0 the one-wave-half builder above, 1 both halves (D <= 32 only)
__global__ __launch_bounds__(64) void k_niw_to_theta_wave32s(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out, uint4* __restrict__ uw = nullptr,
    SviSync sy = SviSync{nullptr, 0u, nullptr, nullptr, nullptr}) {
  if (svi_gate(sy)) k_niw_to_theta_wave32s_body(mu, sigma, kappa, nu, K, D, Kp, theta, status, orb, logdet_out, uw);
  svi_arrive(sy);
}
template <int DMAX>
__global__ __launch_bounds__(64) void k_niw_to_theta_wave(
    const double* __restrict__ mu, const double* __restrict__ sigma,
    const double* __restrict__ kappa, const double* __restrict__ nu, int K, int D, int Kp,
    double* __restrict__ theta, int* __restrict__ status, double* __restrict__ orb,
    double* __restrict__ logdet_out, uint4* __restrict__ uw = nullptr,
    SviSync sy = SviSync{nullptr, 0u, nullptr, nullptr, nullptr}) {
  if (svi_gate(sy)) k_niw_to_theta_wave_body<DMAX>(mu, sigma, kappa, nu, K, D, Kp, theta, status, orb, logdet_out, uw);
  svi_arrive(sy);
}

// The three data-dependent scalars of the NIW factor's ELBO term (Gaussian.get_vlb, Bishop
// 10.74 / 10.77) per state, from what the E-step already keeps on the device:
//   out[k] = log det sigma_mf,  out[K+k] = tr(sigma_mf^-1 sigma_0),
//   out[2K+k] = (mu_mf - mu_0)' sigma_mf^-1 (mu_mf - mu_0).
// theta holds W = (nu/2) sigma_mf^-1 in feature form (-W_aa, -2 W_ab), so both contractions are
// sums over the quadratic features.  One wave per state; `out` is host-visible pinned memory.
__global__ __launch_bounds__(64) void k_niw_vlb_terms(
    const double* __restrict__ theta, const int* __restrict__ fab, int F, int D, int Kp,
    const double* __restrict__ mu, const double* __restrict__ nu, const double* __restrict__ logdet,
    const double* __restrict__ mu0, const double* __restrict__ sigma0, int K,
    double* __restrict__ out) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const double* __restrict__ m = mu + (size_t)k * D;
  const double* __restrict__ m0 = mu0 + (size_t)k * D;
  const double* __restrict__ s0 = sigma0 + (size_t)k * D * D;
  double tr = 0.0, qd = 0.0;
  for (int f = lane; f < F; f += 64) {
    const int ab = fab[f], a = ab & 0xffff, b = ab >> 16;
    if (b >= D) continue;                    // linear and constant features
    const double t = theta[(size_t)f * Kp + k];
    tr = fma(t, s0[a * D + b], tr);
    qd = fma(t, (m[a] - m0[a]) * (m[b] - m0[b]), qd);
  }
  tr = wave_sum(tr);
  qd = wave_sum(qd);
  if (lane == 0) {
    const double c = -2.0 / nu[k];
    out[k] = logdet[k];
    out[K + k] = c * tr;
    out[2 * K + k] = c * qd;
  }
}

// ------------------------------------------------------------------------------------
//  K8: Categorical emissions (reference Categorical branches hmmsgd_metaobs.py:907-926,
//      1071-1084).  obs is [T][1] with the symbol index stored as a double.
//      K8a  ll[row][k] = table[x_row][k]  (table = E log theta, transposed to [V][K]);
//           masked (MASK_AS_NAN) or NaN rows -> 0 for every state, like the Gaussian path.
//      K8b  counts[v][k] = sum over unmasked rows with x = v of q[row][k]: one wavefront per
//           row chunk walks its rows in order (lane = state, LDS table [V][Kp]), so the sums
//           are deterministic; per-chunk partials, reduced by k_finalize_cat.
//      Gather / HBM-bound work: no MFMA here (the transition statistic still runs on the
//      pipelined GEMM in its transition-only mode).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emission_cat(
    const double* __restrict__ obs, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ starts, int64_t nrows, int Lm, int K, int V,
    const double* __restrict__ table, uint32_t flags, double* __restrict__ ll) {
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= nrows) return;
  const int64_t b = g / Lm;
  const int64_t orow = starts[b] + (g - b * Lm);
  const double x = obs[orow];
  bool bad = (x != x) || ((flags & SVIHMM_MASK_AS_NAN) && mask && mask[orow]);
  const int v = bad ? 0 : (int)x;
  bad |= v < 0 || v >= V;
  for (int k = lane; k < K; k += 64) ll[g * K + k] = bad ? 0.0 : table[(size_t)v * K + k];
}

