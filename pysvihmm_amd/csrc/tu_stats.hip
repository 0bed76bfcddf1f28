// tu_stats.hip -- statistics GEMMs, finalize: kernels and launchers
// One of the translation units of libsvihmm_hip.so (see host.h).
#include "host.h"
#include "device_helpers.h"
#include "kernels_stats.h"

extern "C" {

// m-tiles (16 features) per wave of the pipelined statistics GEMM: 5 is the instance tuned for
// the bench shape (40 tiles = 2 workgroups of 4 x 5); narrow models have far fewer tiles (K = 16,
// D = 8: 4; K = 64, D = 8: 7) and would run 5-tile waves mostly on padding, so they take 1 / 2 / 4.
// The small instances exist for the staging widths narrow observations need (xk <= 3) only.
static int stats_mt(const svihmm_ctx* h) {
  const int Kp = h->Kp, Fp = h->Fp, D = h->D;
  if (Kp > 64 || h->variant[10] == 1) return 5;
  const int NSPLIT = (Kp / 16 == 4) ? 2 : 1;
  const int xk = (D + 1 + 8 * NSPLIT - 1) / (8 * NSPLIT);
  const int mt = (Fp + Kp) / 16;
  if (xk > 3) return 5;
  if (mt <= 16) return mt <= 4 ? 1 : mt <= 8 ? 2 : 4;
  return (mt + 15) / 16 * 16 < (mt + 19) / 20 * 20 ? 4 : 5;    // whichever pads less (K = 64, D = 24: 25 tiles)
}
// row chunking of the statistics GEMM.  One pipelined workgroup is resident per CU, so the launch
// should be a whole number of rounds of 256 workgroups: chunks x feature groups (grid.y) = 256 r.
// 128 chunks x 2 feature groups at the bench shape; more chunks only add partial-sum traffic
// (64 windows: statistics + finalize 74 -> 56 us, tools/chunk_sweep.py).  chunk = multiple of ST_RB.
// fp32 mode, K = 64 / D <= 32 / whole 32-feature tiles: the statistics GEMM on the bf16 matrix pipe
// (k_stats_bf16x3; variant[10] = 2: the fp32-input MFMA kernel instead)
static bool stats_bf16_ok(const svihmm_ctx* h, int64_t n) {
  // (batches below 32 768 rows -- the 64-window minibatch -- keep the fp32-input kernel's many small
  //  workgroups: one 8-wave workgroup per chunk would leave most CUs idle)
  return h->cur_f32 && h->lin_mode && !h->q_valid && h->K == 64 && h->Kp == 64 && h->D <= 32 && h->Fp > 0 &&
         h->Fp % 32 == 0 && !h->emis_cat && !h->emis_diag && h->variant[10] != 2 && h->variant[1] == 0 &&
         (n >= cu_scaled(h, 32768) || h->variant[10] == 3);     // (variant[10] = 3: tests force it on small batches)
}
// ... and k_stats_bf16x3w (round 5) for what that kernel does not take: wide models (64 < K <= 256) and more
// than 22 feature tiles (D > 32); D <= 64 (two stage buffers of x^T + six planes in 160 KB of LDS)
// feature groups (grid.y): each owns up to 22 feature tiles + (the first Kp / 64 of them) two transition tiles.
// (Measured on configs[4], 68 feature tiles: five groups of 14 + 2 -- two full rounds of the 8 waves each
//  instead of four of 17 + 2 = three ragged rounds -- ran 8.5 against 7.8 ms: every group restages q and
//  re-forms its A terms, which costs more than the idle slots.)
// Small batches (below the large-batch kernels' floor: the 64-window minibatch): six feature tiles per group, so that
// a workgroup's list is one round of its eight waves and the launch has chunks x groups >= ~200 workgroups of four
// stages each instead of ~50 (k_stats_bf16x3 with its floor lifted: 47 us; the fp32-input MFMA kernel: 30 us).
static int bw_feature_groups(const svihmm_ctx* h, int64_t n) {
  const int FT = (h->Fp + 31) / 32, NGz = (h->Kp + 63) / 64;
  if (n < cu_scaled(h, 32768)) return std::max(NGz, (FT + 5) / 6);
  return std::max(NGz, (FT + 21) / 22);
}
static size_t bw_lds(const svihmm_ctx* h) {
  const size_t xb = (((size_t)(h->D + 2) * SB_XRS * 4) + 15) & ~(size_t)15;
  return 2 * (xb + 6 * (size_t)64 * SB_QRS * 2) + 3 * SB_ROWS * sizeof(SbRow);
}
// the part of the test that does not depend on the batch in flight: prepare_ll decides with it whether a wide model's
// batch may enter the fp32 format at all (f32_wide_ok, tu_emission.hip) -- ONE predicate, so the emission / sweep side
// can never commit to float for a batch this side has no kernel for
bool stats_bf16w_shape_ok(const svihmm_ctx* h, int64_t n) {
  // (a wide model that runs in the fp32 format has no other statistics kernel: no batch-size floor there)
  // (D <= 64: the kernel stages x columns 0..63 from the observations and treats columns 64, 65 as the ones / zero
  //  columns -- found by the fuzz at K = 64, D = 79 with the floors lifted)
  return h->K <= 256 && h->D <= 64 && h->Kp % 64 == 0 && h->Fp > 0 && !h->emis_cat &&
         !h->emis_diag && h->variant[10] != 2 && h->variant[1] == 0 && bw_lds(h) <= 160 * 1024 &&
         (h->K > 64 || ((h->Fp + 31) / 32 + 2 > 24 && (n >= cu_scaled(h, 32768) || h->variant[10] == 3)) ||
          (n >= cu_scaled(h, 8192) && n < cu_scaled(h, 32768) && h->variant[10] != 3));      // (minibatch-sized batches at K = 64: the six-tile groups)
}
static bool stats_bf16w_ok(const svihmm_ctx* h, int64_t n) {
  return h->cur_f32 && h->lin_mode && !h->q_valid && stats_bf16w_shape_ok(h, n);
}
StatsPlan stats_plan(const svihmm_ctx* h, int64_t n, int forced) {
  int target_chunks = forced > 0 ? forced : (int)cu_scaled(h, 128);
  if (forced <= 0 && !stats_bf16_ok(h, n) && stats_bf16w_ok(h, n)) {
    // chunks x feature groups x state groups = whole rounds of 256 one-per-CU workgroups, >= 32 chunks
    const int per_chunk = bw_feature_groups(h, n) * ((h->Kp + 63) / 64);
    const int ncu = h->ncu;
    const int R = std::max(1, (32 * per_chunk + ncu - 1) / ncu);
    int64_t tc = std::max(1, ncu * R / per_chunk);
    if (tc > n / (4 * SB_ROWS)) tc = std::max<int64_t>(1, n / (4 * SB_ROWS));
    int64_t rpc = ((n + tc - 1) / tc + SB_ROWS - 1) / SB_ROWS * SB_ROWS;
    return {rpc, (n + rpc - 1) / rpc};
  }
  if (forced <= 0 && stats_bf16_ok(h, n)) {
    // one 8-wave workgroup per chunk covers all feature tiles: a chunk per CU (small batches: chunks of
    // at least four 64-row stages)
    int64_t tc = n / (4 * SB_ROWS);
    if (tc < 1) tc = 1;
    if (tc > h->ncu) tc = h->ncu;
    int64_t rpc = ((n + tc - 1) / tc + SB_ROWS - 1) / SB_ROWS * SB_ROWS;
    return {rpc, (n + rpc - 1) / rpc};
  }
  if (forced <= 0 && h->Kp <= 64 && h->Fp > 0 && !h->emis_cat) {
    const int gy = ((h->Fp + h->Kp) / 16 + 4 * stats_mt(h) - 1) / (4 * stats_mt(h));
    const int ncu = h->ncu;
    const int r = std::max(1, (ncu / 2 * gy + ncu / 2) / ncu);     // rounds: round((CUs / 2) gy / CUs)
    // one state tile (K <= 16): the 4-wave workgroups are small enough for two per CU, and the
    // launch is latency- rather than MFMA-bound (K = 16, D = 32: 0.61 -> 0.46 ms); wider models: one
    const int per_cu = (h->Kp == 16 && n >= (int64_t)1 << 18) ? 2 : 1;   // (small batches: more chunks only add partial sums)
    target_chunks = std::max(1, ncu * r * per_cu / gy);
  }
  int64_t rpc = (n + target_chunks - 1) / target_chunks;
  const int rb = (stats_bf16_ok(h, n) || stats_bf16w_ok(h, n)) ? SB_ROWS : ST_RB;     // (a forced chunk count: the bf16 kernels' stage)
  rpc = (rpc + rb - 1) / rb * rb;
  return {rpc, (n + rpc - 1) / rpc};
}
// partial statistics of windows [b0, b0+nb) (inner segment [off, off+Lm) of each window of
// length Lq) into partial slots [chunk_base, chunk_base + plan.nchunk) on `stream`
int launch_stats_range(svihmm_ctx* h, int b0, int nb, int Lq, int off, int Lm, uint32_t flags,
                              StatsPlan plan, int64_t chunk_base, hipStream_t stream) {
  const int D = h->D, K = h->K, Kp = h->Kp, Fp = h->Fp, F = h->F;
  const int Ftot = Fp + Kp;
  const int64_t n = (int64_t)nb * Lm;
  const int64_t rpc = plan.rpc, nchunk = plan.nchunk;
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  CK(ensure_starts_pulled(h));    // (an E-step without an emission launch -- host lliks -- still owes the device copy)
  const int64_t* starts_dev = (const int64_t*)h->starts.p + b0;
  int var = h->variant[1];
  if (var != 2) var = 3;      // (2: the double-buffered generation; the VALU generation of round 1 is gone)
  if (var == 3) {   // feasibility of the pipelined kernel (same test as below)
    const int KpW = Kp > 64 ? 64 : Kp;
    const int TPR = 8 * ((KpW / 16 == 4) ? 2 : 1);
    const size_t lds = ((size_t)(D + 3 + KpW) * ST_CC + 2 * (size_t)ST_RB * ST_QS(KpW)) * 8 + 4 * ST_RB * sizeof(StRow4);
    if (lds > 150 * 1024 || (D + 1 + TPR - 1) / TPR > 9 || (Kp > 64 && Kp % 64 != 0)) var = 2;
  }
  // scaled sweeps: the pipelined kernel forms q = ah * bh * scale itself; the others read var_x
  // (wide models too, round 3: the separate posterior pass costs more than the second operand's loads;
  //  variant[15] = 1: K > 64 through q as before)
  // (variant[15] = 2, measurement only: posteriors by their own pass + the GEMM on plain q for every K)
  const bool lin = h->lin_mode && !h->q_valid && var == 3 && (Kp <= 64 || h->variant[15] != 1) && h->variant[15] != 2;
  if (h->lin_mode && !lin) CK(ensure_q(h, h->curB, Lq, stream));
  const size_t qo = (size_t)b0 * Lq * K;
  const double* qv = (const double*)(lin ? h->la.p : h->q.p) + qo;   // (reassigned: see the transition blocks)
  const double* bhv = lin ? (const double*)h->lb.p + qo : nullptr;
  const double* hxv = lin ? (const double*)h->hx.p + (size_t)b0 * Lq : nullptr;
  const double* gxv = lin ? (const double*)h->gx.p + (size_t)b0 * Lq : nullptr;
  const double2* zfv = lin ? (const double2*)h->zfac.p + b0 : nullptr;
  double* partv = (double*)h->part.p + (size_t)chunk_base * Ftot * Kp;
  {
    ProfScope ps(h, KS_STATS, stream);
    if (var == 3) {
      // pipelined VGPR-form GEMM.  K <= 64: all tiles (statistics + transition) in one launch.
      // K > 64: state groups of 64 in grid.z for the emission-statistics tiles; the K x K
      // transition tiles (which need q[t-1] of ALL states as operand rows) go to k_stats_mfma.
      const bool big = Kp > 64;
      const bool bw = lin && h->cur_f32 && !stats_bf16_ok(h, n) && stats_bf16w_ok(h, n) && rpc % SB_ROWS == 0 && nchunk * rpc >= n;
      if (h->cur_f32 && big && !bw) return fail("internal: fp32-mode batch of a wide model without its statistics kernel");
      // wide models, scaled sweeps: the feature launch leaves q = ah bh scale behind for the
      // transition-block launch (whose little matrix work per staged row cannot carry two more
      // operand streams: 3.3 against 2.4 ms on configs[4])
      double* qoutv = nullptr;
      if (big && lin && !bw) {
        CK(ensure(h->q, (size_t)h->curB * Lq * K * sizeof(double)));
        qoutv = (double*)h->q.p + qo;
      }
      const int NTt = big ? 4 : Kp / 16;                // n-tiles per workgroup
      const int KpW = 16 * NTt;
      const int NSPLIT = (NTt == 4) ? 2 : 1;
      const int TPR = 8 * NSPLIT;
      const size_t lds = ((size_t)(D + 3 + KpW) * ST_CC + 2 * (size_t)ST_RB * ST_QS(KpW)) * 8 + 4 * ST_RB * sizeof(StRow4);
      const int mtiles = Ftot / 16;
      const int mt_limit = big ? Fp / 16 : mtiles;
      const int xk = (D + 1 + TPR - 1) / TPR;
      if (bw) {
        const int KpF = (Kp + 63) / 64 * 64, NGf = bw_feature_groups(h, n), FT = (Fp + 31) / 32;
        const int TPG = (FT + NGf - 1) / NGf;
        const size_t ldsb = bw_lds(h);
        if (KpF != Kp) return fail("internal: wide fp32 statistics need states padded in groups of 64");
        h->last_kernel[KS_STATS] = "k_stats_bf16x3w";
        hipFuncSetAttribute((const void*)k_stats_bf16x3w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipLaunchKernelGGL(k_stats_bf16x3w, dim3((unsigned)nchunk, NGf, KpF / 64), dim3(512), ldsb, stream, (const double*)h->obs.p, mk,
                           starts_dev, n, Lm, D, K, KpF, Fp, F, (const int*)h->fab.p, (const float*)h->la.p + qo,
                           (const float*)h->lb.p + qo, rpc, flags, Lq, off, partv, hxv, gxv, zfv, TPG);
      }
      else if (lds > 150 * 1024 || xk > 9 || (big && Kp % 64 != 0)) var = 2;
      else if (lin && h->cur_f32 && !big && stats_bf16_ok(h, n) && rpc % SB_ROWS == 0 && nchunk * rpc >= n) {
        const size_t xb = (((size_t)(D + 2) * SB_XRS * 4) + 15) & ~(size_t)15;
        const size_t ldsb = 2 * (xb + 6 * (size_t)64 * SB_QRS * 2) + 3 * SB_ROWS * sizeof(SbRow);
        h->last_kernel[KS_STATS] = "k_stats_bf16x3";
        hipFuncSetAttribute((const void*)k_stats_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipLaunchKernelGGL(k_stats_bf16x3, dim3((unsigned)nchunk), dim3(512), ldsb, stream, (const double*)h->obs.p, mk,
                           starts_dev, n, Lm, D, K, Fp, F, (const int*)h->fab.p, (const float*)h->la.p + qo,
                           (const float*)h->lb.p + qo, rpc, flags, Lq, off, partv, hxv, gxv, zfv);
      }
      else if (lin && h->cur_f32 && !big) {
        // fp32 mode: float LDS tiles, v_mfma_f32_16x16x4_f32, ah / bh read as float
        const size_t ldsf = ((size_t)(D + 3 + KpW) * ST_CC + 2 * (size_t)ST_RB * ST_QS(KpW)) * 4 + 8 +
                            4 * ST_RB * sizeof(StRow4);
        const int MTs = stats_mt(h);
        dim3 grid((unsigned)nchunk, (mt_limit + 4 * MTs - 1) / (4 * MTs), 1);
#define ST3F(MTV, NTW, NS, XKV)                                                                   \
  do {                                                                                           \
    h->last_kernel[KS_STATS] = "k_stats_mfma4<" #MTV ", " #NTW ", " #NS ", " #XKV ", true, false, float, float, 2>"; \
    if (ldsf > 64 * 1024)                                                                        \
      hipFuncSetAttribute((const void*)k_stats_mfma4<MTV, NTW, NS, XKV, true, false, float, float>, \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);                \
    hipLaunchKernelGGL((k_stats_mfma4<MTV, NTW, NS, XKV, true, false, float, float>), grid,      \
                       dim3(256 * NS), ldsf, stream, (const double*)h->obs.p, mk, starts_dev, n, \
                       Lm, D, K, Fp, F, (const int*)h->fab.p, (const float*)h->la.p, rpc, flags, \
                       Lq, off, partv, Kp, mt_limit, (const float*)h->lb.p, hxv, gxv, zfv, (double*)nullptr); \
  } while (0)
#define ST3FX(NTW, NS) do { if (xk <= 1) ST3F(5, NTW, NS, 1); else if (xk <= 3) ST3F(5, NTW, NS, 3); else if (xk <= 5) ST3F(5, NTW, NS, 5); else ST3F(5, NTW, NS, 9); } while (0)
#define ST3FS(MTV, NTW, NS) do { if (xk <= 1) ST3F(MTV, NTW, NS, 1); else ST3F(MTV, NTW, NS, 3); } while (0)
#define ST3FM(NTW, NS) do { if (MTs == 1) ST3FS(1, NTW, NS); else if (MTs == 2) ST3FS(2, NTW, NS); else if (MTs == 4) ST3FS(4, NTW, NS); else ST3FX(NTW, NS); } while (0)
        // (the barrier-free three-buffer variant measured slower here: 0.85 against 0.83 ms -- the fp32
        //  stage is half as long, the counter wait bites; this mode keeps the stage barrier)
        if (NTt == 4) ST3FM(2, 2); else if (NTt == 3) ST3FM(3, 1); else if (NTt == 2) ST3FM(2, 1); else ST3FM(1, 1);
#undef ST3FM
#undef ST3FS
#undef ST3FX
#undef ST3F
      } else {
        const int MTs = stats_mt(h);
        dim3 grid((unsigned)nchunk, (mt_limit + 4 * MTs - 1) / (4 * MTs), big ? Kp / 64 : 1);
        // four state tiles, five feature tiles per wave, scaled sweeps (the K = 64 epoch shapes): the
        // barrier-free stage loop with three LDS buffers, where they fit (D <= 55)
        const int xk3 = xk <= 1 ? 1 : xk <= 3 ? 3 : xk <= 5 ? 5 : 9;      // (the XK instance ST3T picks below)
        const size_t lds3 = ((size_t)(D + 3 + KpW) * (3 * ST_CS + 2) + 3 * (size_t)ST_RB * ST_QS3(KpW, xk3)) * 8 +
                            4 * ST_RB * sizeof(StRow4) + 16;
        const bool tb = NTt == 4 && MTs == 5 && lds3 <= 160 * 1024 && h->variant[12] != 1;
        if (tb) {
#define ST3TL(XKV, LN)                                                                                         \
  do {                                                                                                         \
    h->last_kernel[KS_STATS] = "k_stats_mfma4<5, 2, 2, " #XKV ", " #LN ", false, double, double, 3>";          \
    hipFuncSetAttribute((const void*)k_stats_mfma4<5, 2, 2, XKV, LN, false, double, double, 3>,                \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);                                \
    hipLaunchKernelGGL((k_stats_mfma4<5, 2, 2, XKV, LN, false, double, double, 3>), grid, dim3(512), lds3,     \
                       stream, (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, Fp, F,                    \
                       (const int*)h->fab.p, qv, rpc, flags, Lq, off, partv, Kp, mt_limit, bhv, hxv, gxv, zfv, qoutv); \
  } while (0)
#define ST3T(XKV) do { if (lin) ST3TL(XKV, true); else ST3TL(XKV, false); } while (0)
          if (xk <= 1) ST3T(1); else if (xk <= 3) ST3T(3); else if (xk <= 5) ST3T(5); else ST3T(9);
#undef ST3T
#undef ST3TL
        } else {
#define ST3L(MTV, NTW, NS, XKV, LN)                                                               \
  do {                                                                                           \
    h->last_kernel[KS_STATS] = "k_stats_mfma4<" #MTV ", " #NTW ", " #NS ", " #XKV ", " #LN ", false, double, double, 2>"; \
    if (lds > 64 * 1024)                                                                         \
      hipFuncSetAttribute((const void*)k_stats_mfma4<MTV, NTW, NS, XKV, LN>,                     \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    hipLaunchKernelGGL((k_stats_mfma4<MTV, NTW, NS, XKV, LN>), grid, dim3(256 * NS), lds, stream, \
                       (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, Fp,                  \
                       F, (const int*)h->fab.p, qv, rpc, flags, Lq, off,                          \
                       partv, Kp, mt_limit, bhv, hxv, gxv, zfv, qoutv);                           \
  } while (0)
#define ST3(MTV, NTW, NS, XKV) do { if (lin) ST3L(MTV, NTW, NS, XKV, true); else ST3L(MTV, NTW, NS, XKV, false); } while (0)
#define ST3X(NTW, NS) do { if (xk <= 1) ST3(5, NTW, NS, 1); else if (xk <= 3) ST3(5, NTW, NS, 3); else if (xk <= 5) ST3(5, NTW, NS, 5); else ST3(5, NTW, NS, 9); } while (0)
#define ST3S(MTV, NTW, NS) do { if (xk <= 1) ST3(MTV, NTW, NS, 1); else ST3(MTV, NTW, NS, 3); } while (0)
#define ST3M(NTW, NS) do { if (MTs == 1) ST3S(1, NTW, NS); else if (MTs == 2) ST3S(2, NTW, NS); else if (MTs == 4) ST3S(4, NTW, NS); else ST3X(NTW, NS); } while (0)
        if (NTt == 4) ST3M(2, 2); else if (NTt == 3) ST3M(3, 1); else if (NTt == 2) ST3M(2, 1); else ST3M(1, 1);
#undef ST3M
#undef ST3S
#undef ST3X
#undef ST3
#undef ST3L
        }
        if (big) {
          // transition tiles: one (64 MTt) x 64 (previous state, state) block per workgroup; two
          // m-tiles per wave where the state count allows (round 3: 4 MFMAs on 4 LDS reads per
          // k-step instead of 2 on 4; variant[14] = 1: one)
          const int MTt = (Kp % 128 == 0 && h->variant[14] != 1) ? 2 : 1;
          dim3 g2((unsigned)nchunk, Kp / (64 * MTt), Kp / 64);
          const size_t ldt = ((size_t)(2 + 64 * MTt) * ST_CC + 2 * (size_t)ST_RB * ST_QS_TR(64, MTt)) * 8 + 4 * ST_RB * sizeof(StRow4);
          const size_t ldt3 = ((size_t)(2 + 64 * MTt) * (3 * ST_CS + 2) + 3 * (size_t)ST_RB * ST_QS_TR(64, MTt)) * 8 +
                              4 * ST_RB * sizeof(StRow4) + 16;
#define STT(MTV, LN)                                                                              \
  do {                                                                                           \
    hipFuncSetAttribute((const void*)k_stats_mfma4<MTV, 2, 2, 1, LN, true>,                      \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldt);                   \
    hipLaunchKernelGGL((k_stats_mfma4<MTV, 2, 2, 1, LN, true>), g2, dim3(512), ldt, stream,      \
                       (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, Fp,                  \
                       F, (const int*)h->fab.p, qv, rpc, flags, Lq, off,                          \
                       partv, Kp, mt_limit, bhv, hxv, gxv, zfv, qoutv);                           \
  } while (0)
#define STT3(MTV, LN)                                                                             \
  do {                                                                                           \
    hipFuncSetAttribute((const void*)k_stats_mfma4<MTV, 2, 2, 1, LN, true, double, double, 3>,   \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldt3);                  \
    hipLaunchKernelGGL((k_stats_mfma4<MTV, 2, 2, 1, LN, true, double, double, 3>), g2, dim3(512), ldt3, stream, \
                       (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, Fp,                  \
                       F, (const int*)h->fab.p, qv, rpc, flags, Lq, off,                          \
                       partv, Kp, mt_limit, bhv, hxv, gxv, zfv, qoutv);                           \
  } while (0)
          // (no obs columns in these tiles: XK = 1 always, the three-buffer loop always fits)
          if (lin) qv = qoutv;      // written by the feature launch above
          if (h->variant[12] != 1) { if (MTt == 2) STT3(2, false); else STT3(1, false); }
          else { if (MTt == 2) STT(2, false); else STT(1, false); }
          if (lin && off == 0 && Lm == Lq && b0 == 0 && nb == h->curB) h->q_valid = true;
#undef STT3
#undef STT
        }
      }
    }
    if (var == 2) {
      const int ntile = Kp / 16;
      const int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
      const int MT = 3;
      const int DS = (D + 2) | 1;
      const size_t lds = ((size_t)ST_RB * DS + (size_t)ST_RB * (16 * NT + 1) + (size_t)ST_RB * (Kp + 1)) * 8;
      if (lds > 150 * 1024) return fail("statistics: D too large for the LDS-staged GEMM kernels");
      {
        const int mtiles = Ftot / 16;
        dim3 grid((unsigned)nchunk, (mtiles + 4 * MT - 1) / (4 * MT), ntile / NT);
#define ST_LAUNCH(NTV)                                                                        \
  do {                                                                                        \
    if (lds > 64 * 1024)                                                                      \
      hipFuncSetAttribute((const void*)k_stats_mfma<3, NTV>,                                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL((k_stats_mfma<3, NTV>), grid, dim3(256), lds, stream,                  \
                       (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K,                  \
                       Kp, Fp, F, (const int*)h->fab.p, qv, rpc, flags,                       \
                       Lq, off, partv, 0);                                                    \
  } while (0)
        if (NT == 4) ST_LAUNCH(4); else if (NT == 2) ST_LAUNCH(2); else ST_LAUNCH(1);
#undef ST_LAUNCH
      }
    }
    HIPCK(hipGetLastError());
  }
  return 0;
}
int launch_stats_finalize(svihmm_ctx* h, int64_t nchunk, hipStream_t stream) {
  const int D = h->D, K = h->K, Kp = h->Kp, Fp = h->Fp, F = h->F;
  ProfScope ps(h, KS_FINALIZE, stream);
  const int64_t tot = (int64_t)(Fp + Kp) * Kp;
  const int lbB = h->lb_pending;     // deferred ELBO total of the scaled sweeps rides along
  h->lb_pending = 0;
  hipLaunchKernelGGL(k_finalize, dim3((unsigned)((tot + 63) / 64) + (lbB ? 1 : 0)), dim3(256), 0, stream,
                     (const double*)h->part.p, (int)nchunk, D, K, Kp, Fp, F,
                     (const int*)h->fab.p, (double*)h->packed.p,
                     (const double*)(lbB ? h->local_lb.p : nullptr), lbB, h->emis_diag ? 1 : 0);
  HIPCK(hipGetLastError());
  return 0;
}
int ensure_stats(svihmm_ctx* h, int64_t nchunk_total) {
  CK(ensure_feature_table(h));
  CK(ensure(h->part, (size_t)nchunk_total * (h->Fp + h->Kp) * h->Kp * sizeof(double)));
  CK(ensure(h->packed, (size_t)packed_len(h) * sizeof(double)));
  return 0;
}
// Categorical statistics: transition block on the pipelined GEMM (transition-only mode),
// symbol counts by k_stats_cat, both reduced by k_finalize_cat into [A_raw | counts | lb]
static int launch_stats_cat(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  const int K = h->K, Kp = h->Kp, V = h->V, D = h->D;
  if (K > 256) return fail("Categorical statistics: K > 256 unsupported");
  CK(cat_uncentre(h));
  const int KpT = (K + 63) / 64 * 64;
  const int64_t n = (int64_t)B * Lm;
  hipStream_t stream = h->stream;
  CK(ensure_q(h, h->curB, Lq, stream));
  CK(ensure_starts_pulled(h));
  const StatsPlan plan = stats_plan(h, n);
  const int64_t rpcc = (n + 1023) / 1024 > 64 ? (n + 1023) / 1024 : 64;
  const int nchunkc = (int)((n + rpcc - 1) / rpcc);
  CK(ensure(h->part, (size_t)plan.nchunk * KpT * KpT * sizeof(double)));
  CK(ensure(h->partc, (size_t)nchunkc * V * Kp * sizeof(double)));
  CK(ensure(h->packed, packed_len(h) * sizeof(double)));
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  {
    ProfScope ps(h, KS_STATS, stream);
    const size_t lds = ((size_t)(D + 3 + 64) * ST_CC + 2 * (size_t)ST_RB * ST_QS_TR(64, 1)) * 8 + 4 * ST_RB * sizeof(StRow4);
    hipFuncSetAttribute((const void*)k_stats_mfma4<1, 2, 2, 1, false, true>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 g2((unsigned)plan.nchunk, KpT / 64, KpT / 64);
    hipLaunchKernelGGL((k_stats_mfma4<1, 2, 2, 1, false, true>), g2, dim3(512), lds, stream,
                       (const double*)h->obs.p, mk, (const int64_t*)h->starts.p, n, Lm, D, K, 0, 0,
                       (const int*)nullptr, (const double*)h->q.p, plan.rpc, flags, Lq, off,
                       (double*)h->part.p, KpT, 0, (const double*)nullptr, (const double*)nullptr,
                       (const double*)nullptr, (const double2*)nullptr, (double*)nullptr);
    const size_t ldsc = (size_t)V * Kp * sizeof(double);
    if (ldsc > 150 * 1024) return fail("Categorical statistics: V * K too large for the LDS table");
    if (ldsc > 64 * 1024)
      hipFuncSetAttribute((const void*)k_stats_cat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
    hipLaunchKernelGGL(k_stats_cat, dim3((unsigned)nchunkc), dim3(64), ldsc, stream, (const double*)h->obs.p, mk,
                       (const int64_t*)h->starts.p, n, Lm, K, Kp, V, (const double*)h->q.p, rpcc, Lq, off,
                       (double*)h->partc.p);
    HIPCK(hipGetLastError());
  }
  {
    ProfScope ps(h, KS_FINALIZE, stream);
    const int64_t tot = (int64_t)K * K + (int64_t)K * V;
    hipLaunchKernelGGL(k_finalize_cat, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                       (const double*)h->part.p, (int)plan.nchunk, KpT, (const double*)h->partc.p, nchunkc,
                       K, Kp, V, (double*)h->packed.p);
    HIPCK(hipGetLastError());
  }
  return flush_lb(h, stream);
}

int launch_stats(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  if (h->emis_cat) return launch_stats_cat(h, B, Lq, off, Lm, flags);
  const StatsPlan plan = stats_plan(h, (int64_t)B * Lm, h->variant[8]);
  CK(ensure_stats(h, plan.nchunk));
  CK(launch_stats_range(h, 0, B, Lq, off, Lm, flags, plan, 0, h->stream));
  return launch_stats_finalize(h, plan.nchunk, h->stream);
}

}  // extern "C"
