// tu_fused.hip -- the minibatch E-step's sweeps + statistics as one launch (kernels_fused.h): plan and launcher.
// One of the translation units of libsvihmm_hip.so (see host.h).
#include "host.h"
#include "device_helpers.h"
#include "kernels_fused.h"

#include <algorithm>

extern "C" {

// the statistics GEMM tiling this batch would take as a launch of its own must be the five-tile one the fused kernel's
// statistics workgroups implement (tu_stats.hip, stats_mt: K = 64 with D >= 25, i.e. at least 17 tiles that pad less
// in groups of 20 than in groups of 16)
static bool five_tile_shape(const svihmm_ctx* h) {
  const int Kp = h->Kp, Fp = h->Fp, D = h->D;
  if (Kp != 64 || h->variant[10] != 0) return false;
  const int xk = (D + 1 + 15) / 16, mt = (Fp + Kp) / 16;
  if ((D + 1 + 7) / 8 > 9) return false;         // (the fused kernel's statistics stage: eight threads per row, <= 9 x columns each)
  if (xk > 3) return true;
  if (mt <= 16) return false;
  return !((mt + 15) / 16 * 16 < (mt + 19) / 20 * 20);
}
// stages per chunk: the fewest (>= 3: something to hide; <= 12) with which sweep + statistics workgroups are all
// resident at once and leave CUs for the loop's side-stream kernels; 0: does not fit
// (measurement knobs, read once: SVIHMM_PIPE_WPB = active sweep waves per workgroup, SVIHMM_PIPE_EXP = PipePlan::exp)
static int pipe_wpb() {
  static const int v = [] { const char* e = std::getenv("SVIHMM_PIPE_WPB"); const int x = e ? std::atoi(e) : 4; return x >= 1 && x <= 4 ? x : 4; }();
  return v;
}
static int pipe_exp() {
  static const int v = [] { const char* e = std::getenv("SVIHMM_PIPE_EXP"); return e ? std::atoi(e) : 0; }();
  return v;
}
static int pipe_stages(const svihmm_ctx* h, int B, int Lm, int ngrp, int* nchunk_out, int* Lb_out) {
  const int wpb = pipe_wpb();
  const int nsw = 2 * ((B + wpb - 1) / wpb);
  for (int ns = 3; ns <= PIPE_MAX_STAGES; ++ns) {
    if (ns > Lm) break;
    const int Lb = (Lm + ns - 1) / ns;
    const int64_t nchunk = ((int64_t)B * Lb + 31) / 32;
    if (nsw + nchunk * ngrp <= h->ncu - 8) { *nchunk_out = (int)nchunk; *Lb_out = Lb; return ns; }
  }
  return 0;
}
// Does this batch take the fused launch?  (prepare_ll has run: lin_mode / cur_f32 describe the batch in flight.)
// the part of the test that does not depend on what prepare_ll decides for the batch
static bool fused_shape_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags, int* nst_out) {
  const int K = h->K;
  if (h->variant[4] == 1 || h->variant[1] != 0 || h->variant[2] == 1 || h->variant[2] == 2 || h->variant[7] != 0 ||
      h->variant[8] != 0 || h->variant[15] != 0)
    return false;
  if ((flags & SVIHMM_USE_HOST_LLIKS) || h->emis_cat || h->emis_diag) return false;
  if (K != 64 || h->Fp <= 0 || !five_tile_shape(h)) return false;      // (the sweep workgroups run the all-lanes-valid body)
  if (B < 1 || B > lin_waver_max(h) || Lq > (1 << 20) || use_chain(h, B, Lq)) return false;
  if (off != 0 || Lm != Lq) return false;      // (the local bound covers the whole window: its log terms come from the statistics rows)
  if ((int64_t)B * Lq * K >= ((int64_t)1 << 31)) return false;
  if (pipe_lds_bytes(h->D, PIPE_MAX_STAGES) > 150 * 1024) return false;
  const int ngrp = ((h->Fp + 64) / 16 + 19) / 20;
  int nchunk = 0, Lb = 0;
  if (pipe_stages(h, B, Lm, ngrp, &nchunk, &Lb) <= 0) return false;
  *nst_out = nchunk * ngrp;
  return true;
}
bool sweep_stats_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  int nst = 0;
  if (!fused_shape_ok(h, B, Lq, off, Lm, flags, &nst)) return false;
  if (!h->lin_mode || h->q_valid || h->eh_in_llE) return false;
  // (measured, tools/r6_fused_check.py: from ~16 windows on the fused launch is ahead of sweeps + statistics one after the
  //  other -- 161 against 171 us at 64 windows --, below that the statistics launch is too short to be worth hiding;
  //  float MESSAGES (cur_f32) keep the mode's own sweep + bf16 statistics kernels -- the fp32 mode's minibatches come
  //  here as fp64 batches behind float emission rows (sweep_mixed_ok); variant 4 = 3 forces the fused launch for every
  //  batch it can take -- tests)
  if (h->variant[4] != 3 && (B < 16 || h->cur_f32)) return false;
  return true;
}
// fp32 mode: will this batch take the fused launch with float emission rows and fp64 messages (asked BEFORE prepare_ll)?
// The same thresholds as the fp64 batch's; variant 4 = 6: never (the mode's own sweep + bf16 statistics kernels).
bool sweep_mixed_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  int nst = 0;
  if (h->prec != 1 || !h->f32_ok || h->variant[4] == 6 || h->variant[4] == 3) return false;
  if (!fused_shape_ok(h, B, Lq, off, Lm, flags, &nst)) return false;
  return B >= 16;
}
// Will the fused launch of this batch also compute the emission tiles (asked BEFORE prepare_ll; launch_emission holds
// its launch back only on the fp64 minibatch path whose tile the fused kernel carries)?
// OFF unless asked for (variant 4 = 3: everything the kernel can take, tests; 5: the automatic thresholds + emission).
// Measured at 64 windows (tools/r6_fused_trace.py, r6_fused_check.py, r4_svi_probe.py; profiles/r06g_*): the statistics
// workgroups finish the 1028 tiles 46 - 55 us into the launch (first tile 15 us: 295 KB of theta operands per workgroup,
// 61 MB through the L2s; then 8.5 us a tile against 3.9 us of matrix work -- one wave per SIMD hides nothing) and the
// sweeps, gated round by round, end at 123 us: exactly where 39 us of emission kernel + 84 us of sweeps end.  E-step call
// 170 against 168 us, resident loop 0.190 against 0.185 ms.  The stand-alone emission kernel turned out to be bound by
// the same L2 traffic (every tile re-reads the 270 KB orbit: 277 MB in 39 us = 7.1 TB/s), not by latency.
bool sweep_emission_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  int nst = 0;
  if ((h->variant[4] != 3 && h->variant[4] != 5) || h->variant[5] != 0 || !fused_shape_ok(h, B, Lq, off, Lm, flags, &nst)) return false;
  if (h->prec == 1 && h->f32_ok) return false;           // (fp32 mode: its own emission kernel on the bf16 pipe)
  if (h->D > 32 || h->D % 8 != 0) return false;          // (the resident theta operands of the tile: <= 144 k-steps)
  if (h->variant[4] != 3 && B < 16) return false;
  const int64_t ntile = ((int64_t)B * Lq + 15) / 16;
  return (ntile + nst - 1) / nst <= WLR_MAX_BANDS;
}

// readiness order of the inner rows of a window (cached per (Lq, off, Lm, wrap)): row t of the inner segment has both
// messages -- its own and its predecessor's, which the transition statistic multiplies -- once both sweeps have done
// need(t) = max over {t, pred(t)} of max(t_full, Lq - 1 - t_full) steps
// + the emission tiles of the batch (16 consecutive rows of the [B Lq] row space) in outside-in order: a tile's priority
// is the smallest min(t, Lq - 1 - t) of its rows -- the sweep step at which the first of them is needed; rounds of nst
// tiles; em_thr[r] = every row of priority <= em_thr[r] lies in a tile of rounds 0 .. r
static int pipe_order(svihmm_ctx* h, int B, int nst, int Lq, int off, int Lm, bool wrap, int NS, int Lb, WlrPub* pub,
                      const int** ord_dev, const int** tiles_dev, int* ntile_out, int* nround_out) {
  // cache of device tables: [order | tiles] per (B, nst, Lq, off, Lm, wrap, NS), with the host's copies of the thresholds;
  // a table stays untouched while launches that read it may be in flight.  (The lookup comes first: the two sorts below
  // are ~20 us of host time, and the resident loop calls this once per iteration.)
  const size_t til_off = ((size_t)Lm * sizeof(int) + 15) & ~(size_t)15;
  static_assert(WLR_MAX_BANDS == 12, "PipeTab's threshold arrays");
  for (auto& e : h->pipe_tabs)
    if (e.buf.p && e.Lq == Lq && e.off == off && e.Lm == Lm && e.wrap == wrap && e.NS == NS && e.B == B && e.nst == nst) {
      e.stamp = ++h->pipe_stamp;
      pub->nb = NS;
      for (int i = 0; i < WLR_MAX_BANDS; ++i) { pub->thr[i] = e.thr[i]; pub->em_thr[i] = e.em_thr[i]; }
      pub->em_n = e.em_n; *ntile_out = e.ntile; *nround_out = e.nround;
      *ord_dev = (const int*)e.buf.p; *tiles_dev = (const int*)((const char*)e.buf.p + til_off);
      return 0;
    }
  std::vector<int> need((size_t)Lm), idx((size_t)Lm);
  auto n1 = [&](int t) { const int tf = off + t; return std::max(tf, Lq - 1 - tf); };
  for (int t = 0; t < Lm; ++t) {
    int n = n1(t);
    if (t > 0) n = std::max(n, n1(t - 1));
    else if (wrap) n = std::max(n, n1(Lm - 1));
    need[t] = n; idx[t] = t;
  }
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return need[a] < need[b]; });
  pub->nb = NS;
  for (int s = 0; s < NS; ++s) pub->thr[s] = need[idx[std::min((s + 1) * Lb, Lm) - 1]];
  const int64_t nrows = (int64_t)B * Lq;
  const int ntile = (int)((nrows + 15) / 16), nround = (ntile + nst - 1) / nst;
  std::vector<int> tprio((size_t)ntile), tidx((size_t)ntile);
  for (int i = 0; i < ntile; ++i) {
    int p = 0x7fffffff;
    for (int64_t g = (int64_t)i * 16; g < std::min<int64_t>((int64_t)i * 16 + 16, nrows); ++g) {
      const int t = (int)(g % Lq);
      p = std::min(p, std::min(t, Lq - 1 - t));
    }
    tprio[i] = p; tidx[i] = i;
  }
  std::stable_sort(tidx.begin(), tidx.end(), [&](int a, int b) { return tprio[a] < tprio[b]; });
  *ntile_out = ntile; *nround_out = nround;
  pub->em_n = 0;
  if (nround <= WLR_MAX_BANDS) {
    pub->em_n = nround;
    for (int r = 0; r < nround; ++r)
      pub->em_thr[r] = (r + 1) * nst < ntile ? tprio[tidx[(size_t)(r + 1) * nst]] - 1 : 0x7fffffff;
  }
  svihmm_ctx::PipeTab* slot = nullptr;
  for (auto& e : h->pipe_tabs) if (!e.buf.p) { slot = &e; break; }
  if (!slot) {
    slot = &h->pipe_tabs[0];
    for (auto& e : h->pipe_tabs) if (e.stamp < slot->stamp) slot = &e;
    HIPCK(hipStreamSynchronize(h->stream));           // (the evicted table's readers are done)
  }
  CK(ensure(slot->buf, til_off + (size_t)ntile * sizeof(int)));
  std::vector<char> img(til_off + (size_t)ntile * sizeof(int), 0);
  std::memcpy(img.data(), idx.data(), (size_t)Lm * sizeof(int));
  std::memcpy(img.data() + til_off, tidx.data(), (size_t)ntile * sizeof(int));
  HIPCK(hipMemcpy(slot->buf.p, img.data(), img.size(), hipMemcpyHostToDevice));
  slot->Lq = Lq; slot->off = off; slot->Lm = Lm; slot->wrap = wrap; slot->NS = NS; slot->B = B; slot->nst = nst;
  slot->stamp = ++h->pipe_stamp;
  for (int i = 0; i < WLR_MAX_BANDS; ++i) { slot->thr[i] = i < NS ? pub->thr[i] : 0; slot->em_thr[i] = i < pub->em_n ? pub->em_thr[i] : 0; }
  slot->em_n = pub->em_n; slot->ntile = ntile; slot->nround = nround;
  *ord_dev = (const int*)slot->buf.p; *tiles_dev = (const int*)((const char*)slot->buf.p + til_off);
  return 0;
}

// sweeps of the B windows of length Lq + statistics over their inner segments [off, off + Lm) + finalize
int launch_sweep_stats(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const int K = h->K, D = h->D, Fp = h->Fp, F = h->F;
  const int ngrp = ((Fp + 64) / 16 + 19) / 20;
  PipePlan pl = {};
  pl.wpb = pipe_wpb();
  pl.exp = pipe_exp();
  pl.nsw = 2 * ((B + pl.wpb - 1) / pl.wpb);
  pl.ngrp = ngrp;
  pl.NS = pipe_stages(h, B, Lm, ngrp, &pl.nchunk, &pl.Lb);
  if (pl.NS <= 0) return fail("internal: fused sweep + statistics launch does not fit (sweep_stats_ok)");
  CK(ensure_fb_lin(h, B, Lq));
  CK(ensure(h->local_lb, (size_t)(B + pl.nchunk) * sizeof(double)));     // [B] exponent books per window | [nchunk] the chunks' log terms
  CK(ensure_stats(h, pl.nchunk));
  CK(ensure_starts_pulled(h));
  if (!h->pipe_cnt.p) {      // [band counters | emission round counters], 64 bytes apart
    CK(ensure(h->pipe_cnt, 2 * PIPE_MAX_STAGES * 64));
    HIPCK(hipMemset(h->pipe_cnt.p, 0, 2 * PIPE_MAX_STAGES * 64));
  }
  pl.pub.cnt = (unsigned*)h->pipe_cnt.p;
  pl.pub.em_cnt = (const unsigned*)h->pipe_cnt.p + 16 * PIPE_MAX_STAGES;
  const int nst = pl.nchunk * pl.ngrp;
  CK(pipe_order(h, B, nst, Lq, off, Lm, (flags & SVIHMM_TRANS_WRAP) != 0, pl.NS, pl.Lb, &pl.pub, &pl.ord, &pl.em_tiles,
                &pl.em_ntile, &pl.em_nround));
  for (int s = 0; s < pl.NS; ++s) {
    h->pipe_tgt[s] += 2u * (unsigned)B;            // (monotonic counters: compared by signed difference on the device)
    pl.tgt[s] = h->pipe_tgt[s];
  }
  // the emission tiles inside this launch?  (launch_emission held its launch back: everything else is in place)
  const svihmm_ctx::EmDeferred ed = h->em_def;
  const bool emw = ed.active;
  h->em_def.active = false;
  if (emw) {
    if (ed.B != B || ed.Lm != Lq || pl.pub.em_n <= 0 || h->cur_f32) return fail("internal: deferred emission does not match the fused launch");
    for (int r = 0; r < pl.em_nround; ++r) {
      h->em_tgt[r] += (unsigned)std::min(nst, pl.em_ntile - r * nst);
      pl.pub.em_tgt[r] = h->em_tgt[r];
    }
    pl.starts_copy = ed.starts_copy;
  } else {
    pl.pub.em_n = 0; pl.em_nround = 0;
  }
  const int64_t* starts_arg = emw ? ed.starts : (const int64_t*)h->starts.p;
  if (std::getenv("SVIHMM_PIPE_DBG")) {     // measurement only (tools/r6_fused_trace.py)
    CK(ensure(h->scratch, (size_t)(pl.nsw + pl.nchunk * pl.ngrp) * 32 * 8));
    HIPCK(hipMemsetAsync(h->scratch.p, 0, (size_t)(pl.nsw + pl.nchunk * pl.ngrp) * 32 * 8, h->stream));
    pl.dbg = (unsigned long long*)h->scratch.p;
    pl.pub.dbgw = pl.dbg;
  }
  h->m_nb = 0;
  h->have_lb = true;
  size_t lds = std::max(pipe_lds_bytes(D, pl.NS), (size_t)84 * 1024);     // (> 80 KB: one workgroup per CU)
  {                           // (the emission tile's LDS: tu_emission.hip, k_emission_orbit_ks)
    const int LEN = D + D / 2 + 1, R0 = 3 * 4 * 256 > 16 * LEN + 1 ? 3 * 4 * 256 : ((16 * LEN + 1) & ~1);
    if (emw) lds = std::max(lds, (size_t)(R0 + 16 * 4) * 8 + (size_t)pl.em_nround * 16 * 9 + 64);    // (+ the tiles' row records)
  }
  const SviSync gsy = sweep_gate(h, h->stream);
  const void* Ehv = h->ll.p;
  const double* kx = (const double*)h->kexp.p;
  const double* mi = (const double*)h->mod_init.p;
  const double* l0 = (const double*)h->ll0.p;
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  const int xk = (D + 1 + 7) / 8;
  const dim3 grid((unsigned)(pl.nsw + pl.nchunk * pl.ngrp));
  {
    ProfScope ps(h, KS_FB, h->stream);
#define FZ(XKV, STT, EMV, RNV)                                                                                           \
  do {                                                                                                                   \
    h->last_kernel[KS_FB] = "k_sweep_stats<" #XKV ", " #STT ", " #EMV ", " #RNV ">";                                     \
    hipFuncSetAttribute((const void*)k_sweep_stats<XKV, STT, EMV, RNV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_sweep_stats<XKV, STT, EMV, RNV>), grid, dim3(256), lds, h->stream, (const STT*)Ehv, kx,        \
                       (const double*)h->Aexp.p, (const double*)h->AexpT.p, mi, l0, (size_t)K, Lq, K, (STT*)h->la.p,     \
                       (STT*)h->lb.p, (double*)h->hx.p, (double*)h->gx.p, (double*)h->local_lb.p, (double*)h->logz.p,    \
                       (double2*)h->zfac.p, gsy, (const double*)h->obs.p, mk, starts_arg, B, Lm, off, D,                 \
                       Fp, F, (const int*)h->fab.p, flags, (double*)h->part.p, pl, (const double*)h->theta_orb.p,        \
                       ed.flags);                                                                                        \
  } while (0)
#define FZX(STT, EMV, RNV) do { if (xk <= 5) FZ(5, STT, EMV, RNV); else FZ(9, STT, EMV, RNV); } while (0)
#define FZM(XKV, RNV)                                                                                                    \
  do {                                                                                                                   \
    h->last_kernel[KS_FB] = "k_sweep_stats<" #XKV ", double, false, " #RNV ", float>";                                   \
    hipFuncSetAttribute((const void*)k_sweep_stats<XKV, double, false, RNV, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_sweep_stats<XKV, double, false, RNV, float>), grid, dim3(256), lds, h->stream, (const float*)Ehv, kx, \
                       (const double*)h->Aexp.p, (const double*)h->AexpT.p, mi, l0, (size_t)K, Lq, K, (double*)h->la.p, \
                       (double*)h->lb.p, (double*)h->hx.p, (double*)h->gx.p, (double*)h->local_lb.p, (double*)h->logz.p, \
                       (double2*)h->zfac.p, gsy, (const double*)h->obs.p, mk, starts_arg, B, Lm, off, D,                 \
                       Fp, F, (const int*)h->fab.p, flags, (double*)h->part.p, pl, (const double*)h->theta_orb.p,        \
                       ed.flags);                                                                                        \
  } while (0)
#define FZXM(RNV) do { if (xk <= 5) FZM(5, RNV); else FZM(9, RNV); } while (0)
    // (fp64 messages, transition expectations inside a float's range: the sweeps re-normalise every fourth step --
    //  kernels_wave_linr.h, RN; variant 16 = 1 keeps every step)
    const bool rn4 = !h->cur_f32 && h->f32_ok && h->variant[16] != 1;
    if (h->eh_float) {       // fp32 mode: float emission rows, fp64 messages and statistics
      if (emw || h->cur_f32) return fail("internal: mixed-format fused launch in the wrong state");
      if (rn4) FZXM(4); else FZXM(1);
    }
    else if (h->cur_f32) FZX(float, false, 1);
    else if (emw) { if (rn4) FZX(double, true, 4); else FZX(double, true, 1); }
    else { if (rn4) FZX(double, false, 4); else FZX(double, false, 1); }
#undef FZXM
#undef FZM
#undef FZX
#undef FZ
    HIPCK(hipGetLastError());
  }
  if (pl.dbg) {      // measurement only: the launch's stamps as text
    HIPCK(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> st((size_t)grid.x * 32);
    HIPCK(hipMemcpy(st.data(), pl.dbg, st.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(std::getenv("SVIHMM_PIPE_DBG"), "w")) {
      unsigned long long t0 = ~0ull;
      for (unsigned i = 0; i < grid.x; ++i) if (st[(size_t)i * 32] && st[(size_t)i * 32] < t0) t0 = st[(size_t)i * 32];
      std::fprintf(f, "# nsw %d nchunk %d ngrp %d NS %d Lb %d thr", pl.nsw, pl.nchunk, pl.ngrp, pl.NS, pl.Lb);
      for (int s = 0; s < pl.NS; ++s) std::fprintf(f, " %d", pl.pub.thr[s]);
      std::fprintf(f, "\n");
      for (unsigned i = 0; i < grid.x; ++i) {
        std::fprintf(f, "%u", i);
        for (int k = 0; k < 32; ++k) { const unsigned long long v = st[(size_t)i * 32 + k]; std::fprintf(f, " %.2f", v ? (double)(v - t0) * 0.01 : -1.0); }
        std::fprintf(f, "\n");
      }
      std::fclose(f);
    }
  }
  h->lb_pending = B + pl.nchunk;     // the windows' exponent books + the chunks' log terms: summed by k_finalize's extra workgroup
  return launch_stats_finalize(h, pl.nchunk, h->stream);
}

}  // extern "C"
