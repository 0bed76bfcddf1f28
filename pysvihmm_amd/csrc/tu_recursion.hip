// tu_recursion.hip -- forward / backward recursions, blocked scan, FFBS: kernels and launchers
// One of the translation units of libsvihmm_hip.so (see host.h).
#include "host.h"
#include "device_helpers.h"
#include "kernels_recursion.h"

extern "C" {

int launch_fb(svihmm_ctx* h, int B, int Lm, int dir0, int ndir,
                     const double* ll, double* la, double* lb) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const int K = h->K;
  const size_t n = (size_t)B * Lm * K * sizeof(double);
  if (!ll) {
    if (dir0 == 0) CK(ensure(h->la, n));
    if (dir0 + ndir > 1) CK(ensure(h->lb, n));
    ll = (const double*)h->ll.p;
    la = (double*)h->la.p;
    lb = (double*)h->lb.p;
  }
  ProfScope ps(h, KS_FB);
  dim3 grid(B, ndir);
  const double* A = (const double*)h->Aexp.p;
  const double* mi = (const double*)h->mod_init.p;
  if (h->exact_log) {   // transition expectations outside exp()'s range: the literal recursion
    const int threads = (K + 63) / 64 * 64;
    const int in_lds = ((size_t)K * (K + 1) + 2 * K) * 8 <= 150 * 1024;
    const size_t lds = (2 * (size_t)K + (in_lds ? (size_t)K * (K + 1) : 0)) * 8;
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_fb_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_fb_exact, grid, dim3(threads), lds, h->stream, ll, (const double*)h->ltran.p, mi,
                       Lm, K, dir0, in_lds, la, lb);
    HIPCK(hipGetLastError());
    return 0;
  }
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

int launch_posterior(svihmm_ctx* h, int B, int Lm, bool total) {
  const int K = h->K;
  // rows per workgroup: 256 for big batches, down to 16 so that small ones give >= ~2048 workgroups
  int rps = 256;
  while (rps > 16 && (int64_t)B * ((Lm + rps - 1) / rps) < 2048) rps >>= 1;
  const int nseg = (Lm + rps - 1) / rps;
  CK(ensure(h->q, (size_t)B * Lm * K * sizeof(double)));
  CK(ensure(h->lse_part, (size_t)B * nseg * sizeof(double)));
  CK(ensure(h->local_lb, (size_t)B * sizeof(double)));
  CK(ensure(h->packed, (size_t)packed_len(h) * sizeof(double)));
  ProfScope ps(h, KS_POSTERIOR);
  dim3 grid((unsigned)((size_t)B * nseg));
#define POST_LAUNCH(KPL)                                                                  \
  hipLaunchKernelGGL(k_posterior<KPL>, grid, dim3(256), 0, h->stream, (const double*)h->la.p, \
                     (const double*)h->lb.p, Lm, K, nseg, rps, (double*)h->q.p, (double*)h->lse_part.p)
  if (K <= 64) POST_LAUNCH(1);
  else if (K <= 256) POST_LAUNCH(4);
  else POST_LAUNCH(16);
#undef POST_LAUNCH
  double* lbtot = nullptr;
  if (total) lbtot = (double*)h->packed.p + (packed_len(h) - 1);
  hipLaunchKernelGGL(k_reduce_lb, dim3(1), dim3(256), 0, h->stream, (const double*)h->lse_part.p,
                     B, nseg, (double*)h->local_lb.p, lbtot);
  HIPCK(hipGetLastError());
  return 0;
}

// forward sweep, then backward sweep with the posterior fused (K <= 64).  want_lb: also
// materialise lbeta (API readback); total: write sum_b local_lb into packed[last].
int launch_fb_fused(svihmm_ctx* h, int B, int Lm, bool want_lb, bool total) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const int K = h->K;
  const size_t n = (size_t)B * Lm * K * sizeof(double);
  CK(ensure(h->la, n));
  CK(ensure(h->q, n));
  if (want_lb) CK(ensure(h->lb, n));
  CK(ensure(h->local_lb, (size_t)B * sizeof(double)));
  CK(ensure(h->logz, (size_t)B * sizeof(double)));
  CK(ensure(h->packed, (size_t)packed_len(h) * sizeof(double)));
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
      double* lbtot = (double*)h->packed.p + (packed_len(h) - 1);
      hipLaunchKernelGGL(k_sum_lb, dim3(1), dim3(256), 0, h->stream, (const double*)llb, B, lbtot);
    }
    HIPCK(hipGetLastError());
  }
  return 0;
}

// scaled linear-domain sweeps (K <= 64): Eh / kexp -> ah, (na, k), Z -> var_x
int ensure_fb_lin(svihmm_ctx* h, int B, int Lm) {
  const int K = h->K;
  const size_t n = (size_t)B * Lm * K * sizeof(double);
  CK(ensure(h->la, n));
  CK(ensure(h->lb, n));
  CK(ensure(h->hx, (size_t)B * Lm * sizeof(double)));
  CK(ensure(h->gx, (size_t)B * Lm * sizeof(double)));
  CK(ensure(h->zfac, (size_t)B * sizeof(double2)));
  CK(ensure(h->local_lb, (size_t)B * sizeof(double)));
  CK(ensure(h->logz, (size_t)B * sizeof(double)));
  CK(ensure(h->packed, (size_t)packed_len(h) * sizeof(double)));
  CK(ensure(h->a0v, (size_t)B * K * sizeof(double)));
  CK(ensure(h->a0e, (size_t)B * sizeof(double)));
  return 0;
}
// initial messages of windows [b0, b0+nb): mod_init + ll_0 in the log domain (k_lin_init); the
// first rows' log-likelihoods come from the scaled emission's side output (ll0) or, where the
// plain lliks are kept (host lliks, wide models, Categorical), straight from those
static int launch_lin_init(svihmm_ctx* h, int b0, int nb, int Lm, hipStream_t stream) {
  const int K = h->K;
  const double* src = h->eh_in_llE ? (const double*)h->ll.p + (size_t)b0 * Lm * K
                                   : (const double*)h->ll0.p + (size_t)b0 * K;
  const size_t stride = h->eh_in_llE ? (size_t)Lm * K : (size_t)K;
  hipLaunchKernelGGL(k_lin_init, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, stream,
                     (const double*)h->mod_init.p, src, stride, (const double*)h->kexp.p + (size_t)b0 * Lm,
                     nb, Lm, K, (double*)h->a0v.p + (size_t)b0 * K, (double*)h->a0e.p + b0);
  HIPCK(hipGetLastError());
  return 0;
}
// minibatch-sized batches, 16 < K <= 64: the kernels that split one window over one wave's registers (fp64,
// k_wave_linr) or four waves (fp32 storage, k_wave_lin4)
static bool minibatch_wave_kernel(const svihmm_ctx* h, int K, int nb, int Lm) {
  if (K <= 16 || K > 64 || nb >= lin_wave_max(h) || h->variant[7] == 2 || h->variant[7] == 3) return false;
  return h->cur_f32 ? nb <= lin_wave4_max(h) : (nb <= lin_waver_max(h) && Lm <= (1 << 20));
}
// SVI loop on counters (svihmm_hip.hip, svi_globals): those two kernels wait for the side stream's globals
// kernel themselves -- no stream-order event in front of them
SviSync sweep_gate(svihmm_ctx* h, hipStream_t stream) {
  SviSync sy = {};
  if (h->svi_flags && h->in_svi_estep && h->globals_ev && stream == h->stream && h->svi_sync.p) {
    sy.gate = (const unsigned*)h->svi_sync.p + 16;
    sy.gate_tgt = (unsigned)h->tgt_glob;
    sy.status = h->svi_status_dev;
    // (sweeps that cannot get their globals poison their iteration: the global step behind them does not run)
    sy.dead = (unsigned*)h->svi_sync.p + 16 * 7;
    sy.ticks = h->svi_ticks;
    sy.poison = (unsigned*)h->svi_sync.p + 16 * 5;
    sy.poison_val = SVI_POISON_BASE - (unsigned)(h->svi_cur_it < 0 ? 0 : h->svi_cur_it);
    if (h->variant[0] == 3 && h->svi_cur_it == 3) {      // (debug: a count that never comes, 2 ms bound -- svi_recover's test)
      sy.gate_tgt += 1000u;
      sy.ticks = (unsigned long long)(2.0 * (h->wall_clock_khz > 0.0 ? h->wall_clock_khz : 100000.0));
    }
    h->globals_ev = nullptr;
  }
  if (h->svi_flags && h->in_svi_estep && stream == h->stream && h->svi_sync.p && h->elbo_pending) {
    // the deferred ELBO kernels of the previous iteration start with these sweeps (svihmm_svi_iteration)
    sy.early = (unsigned*)h->svi_sync.p + 16 * 4;
    ++h->tgt_early;
    h->sweep_signalled = true;
  }
  return sy;
}
// both sweeps over windows [b0, b0+nb) of the current batch on `stream` (buffers ensured):
// one launch, blockIdx.y = direction
int launch_fb_lin_range(svihmm_ctx* h, int b0, int nb, int Lm, hipStream_t stream) {
  const int K = h->K;
  const bool wave_kernel = minibatch_wave_kernel(h, K, nb, Lm);
  if (h->svi_flags && h->in_svi_estep && stream == h->stream && !wave_kernel) CK(wait_globals(h));
  // measurement only (tools/r4_overlap_probe.py): variant[7] = 9 skips the sweep launch -- the
  // statistics then read the previous step's messages; bounds what hiding the sweeps could give
#ifdef SVIHMM_MEASURE
  if (h->variant[7] == 9) return 0;
#endif
  const int NW = (K + 15) / 16;
  const bool full = (K == 16 * NW);
  dim3 grid((nb + 15) / 16, 2);
  const size_t ro = (size_t)b0 * Lm;
  const double* Eh = (const double*)(h->eh_in_llE ? h->llE.p : h->ll.p) + ro * K;
  const double* kx = (const double*)h->kexp.p + ro;
  double* ah = (double*)h->la.p + ro * K;
  double* bh = (double*)h->lb.p + ro * K;
  double* hx = (double*)h->hx.p + ro;
  double* gx = (double*)h->gx.p + ro;
  double2* zf = (double2*)h->zfac.p + b0;
  double* llb = (double*)h->local_lb.p + b0;
  double* lz = (double*)h->logz.p + b0;
  const double* a0v = (const double*)h->a0v.p + (size_t)b0 * K;
  const double* a0e = (const double*)h->a0e.p + b0;
  // first-row log-likelihoods of the windows (see launch_lin_init): the wave-per-window kernels
  // form the initial message themselves, the tile kernels take it from k_lin_init
  const double* mi = (const double*)h->mod_init.p;
  const double* l0 = h->eh_in_llE ? (const double*)h->ll.p + ro * K : (const double*)h->ll0.p + (size_t)b0 * K;
  const size_t l0s = h->eh_in_llE ? (size_t)Lm * K : (size_t)K;
  ProfScope ps(h, KS_FB, stream);
  if (h->cur_f32) {
    // fp32 mode (K <= 64, b0 == 0): the same kernels instantiated for float storage
    const float* Ef = (const float*)h->ll.p;
    float* af = (float*)h->la.p;
    float* bf = (float*)h->lb.p;
    if (K > 64) {
      // wide models (round 5): k_sweeps_lin2 on v_mfma_f32_16x16x4_f32, the transition matrix streamed from
      // its float copy (rows up to 256 + slack zeroed once per buffer, K rows converted per launch)
      CK(launch_lin_init(h, b0, nb, Lm, stream));
      const size_t nrow_t = 256 + 16, kk = (size_t)K * K;
      CK(ensure(h->AexpF, nrow_t * K * sizeof(float)));
      CK(ensure(h->AexpTF, nrow_t * K * sizeof(float)));
      HIPCK(hipMemsetAsync((char*)h->AexpF.p + kk * 4, 0, (nrow_t * K - kk) * 4, stream));
      HIPCK(hipMemsetAsync((char*)h->AexpTF.p + kk * 4, 0, (nrow_t * K - kk) * 4, stream));
      hipLaunchKernelGGL(k_f64_to_f32, dim3(64), dim3(256), 0, stream, (const double*)h->Aexp.p, (float*)h->AexpF.p, kk);
      hipLaunchKernelGGL(k_f64_to_f32, dim3(64), dim3(256), 0, stream, (const double*)h->AexpT.p, (float*)h->AexpTF.p, kk);
      // 16 windows per workgroup: with two window tiles per wave the 128 resident transition values + the
      // second tile's accumulators spill (4.0 ms on configs[4]; one tile: 3.17; the streamed kernel 3.70)
      const bool w32 = false;
#define SWP2F(F, WT)                                                                                          \
  do {                                                                                                        \
    const size_t lds = (size_t)(WT) * sizeof(LinShared<16>);                                                  \
    dim3 g2((unsigned)((nb + 16 * (WT) - 1) / (16 * (WT))), 2);                                               \
    hipFuncSetAttribute((const void*)k_sweeps_lin2<8, F, WT, float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_sweeps_lin2<8, F, WT, float, true>), g2, dim3(512), lds, stream, Ef, kx,            \
                       (const float*)h->AexpF.p, (const float*)h->AexpTF.p, a0v, a0e, nb, Lm, K, af, bf, hx, gx, llb, lz, zf); \
  } while (0)
      if (w32) { if (K == 256) SWP2F(true, 2); else SWP2F(false, 2); }
      else if (K == 256) SWP2F(true, 1); else SWP2F(false, 1);
#undef SWP2F
      HIPCK(hipGetLastError());
      return 0;
    }
    if (nb < lin_wave_max(h) && h->variant[7] != 2) {
      dim3 gw((unsigned)nb, 2);
#define WLF(KM, FK) hipLaunchKernelGGL((k_wave_lin<KM, FK, float>), gw, dim3(64), 0, stream, Ef, kx, (const double*)h->Aexp.p, \
                                       (const double*)h->AexpT.p, mi, l0, l0s, Lm, K, af, bf, hx,   \
                                       gx, llb, lz, zf)
#define WL4F(KM) hipLaunchKernelGGL((k_wave_lin4<KM, float>), gw, dim3(256), 0, stream, Ef, kx, (const double*)h->Aexp.p, \
                                    (const double*)h->AexpT.p, mi, l0, l0s, Lm, K, af, bf, hx, gx, \
                                    llb, lz, zf, gsy)
      // (round 5: the register-resident one-wave kernel k_wave_linr of the fp64 path was also built with fp32
      //  arithmetic: 278 against 303 ns per step, but over 257 steps the statistics drift to 1.2e-4 of the fp64
      //  oracle -- inside the mode's 1e-3, above this suite's 1e-4 canary; the fp32 mode keeps the four-wave
      //  kernel, which computes in fp64 on float storage)
      if (wave_kernel && nb <= lin_waver_max(h) && Lm <= (1 << 20) && h->variant[7] != 4) {
        // (round 5, second half: the one-wave register-resident kernel with fp64 arithmetic on the float storage)
        const SviSync gsy = sweep_gate(h, stream);
        hipLaunchKernelGGL((k_wave_linr<float, double>), gw, dim3(64), 0, stream, Ef, kx, (const double*)h->Aexp.p,
                           (const double*)h->AexpT.p, mi, l0, l0s, Lm, K, af, bf, hx, gx, llb, lz, zf, gsy);
      }
      else if (wave_kernel) { const SviSync gsy = sweep_gate(h, stream); WL4F(64); }
      else if (K <= 16) WLF(16, false); else if (K <= 32) WLF(32, false);
      else if (K == 64) WLF(64, true); else WLF(64, false);
#undef WLF
#undef WL4F
    } else {
      CK(launch_lin_init(h, b0, nb, Lm, stream));
      const LinChain none = {};
#define SWF(NWV, F) hipLaunchKernelGGL((k_sweeps_lin<NWV, F, 0, false, float>), grid, dim3(64 * NWV),              \
                                       sizeof(LinShared<NWV>), stream, Ef, kx, (const double*)h->Aexp.p,         \
                                       (const double*)h->AexpT.p, a0v, a0e, nb, Lm, Lm, K,   \
                                       af, bf, hx, gx, llb, lz, zf, none)
      if (NW == 1) { if (full) SWF(1, true); else SWF(1, false); }
      else if (NW == 2) { if (full) SWF(2, true); else SWF(2, false); }
      else if (NW == 3) { if (full) SWF(3, true); else SWF(3, false); }
      else { if (full) SWF(4, true); else SWF(4, false); }
#undef SWF
    }
    HIPCK(hipGetLastError());
    return 0;
  }
  if (K <= 64 && nb < lin_wave_max(h) && h->variant[7] != 2) {
    // small batches: one wavefront per (window, direction)
    dim3 gw((unsigned)nb, 2);
#define WL(KM, FK) hipLaunchKernelGGL((k_wave_lin<KM, FK>), gw, dim3(64), 0, stream, Eh, kx, (const double*)h->Aexp.p, \
                                      (const double*)h->AexpT.p, mi, l0, l0s, Lm, K, ah, bh, hx,  \
                                      gx, llb, lz, zf)
    // up to a few hundred windows the chip is far from full with one wave per (window,
    // direction): split each window's source states over four waves (variant[7] = 3: off)
    // minibatch-sized batches, K > 16 (round 5): one wave per (window, direction) with the mat-vec in registers
    // (k_wave_linr; it replaces round 3's four-wave k_wave_lin4<64, double>: 316 against 319 ns per step with a
    // quarter of the waves and no LDS exchange; variant[7] = 3: the LDS-broadcast one-wave kernel below)
    if (wave_kernel) {
      const SviSync gsy = sweep_gate(h, stream);
      // (transition expectations inside a float's range: re-normalised every fourth step -- kernels_wave_linr.h, RN;
      //  variant 16 = 1: every step)
      if (h->f32_ok && h->variant[16] != 1)
        hipLaunchKernelGGL((k_wave_linr<double, double, 4>), gw, dim3(64), 0, stream, Eh, kx, (const double*)h->Aexp.p,
                           (const double*)h->AexpT.p, mi, l0, l0s, Lm, K, ah, bh, hx, gx, llb, lz, zf, gsy);
      else
        hipLaunchKernelGGL((k_wave_linr<double>), gw, dim3(64), 0, stream, Eh, kx, (const double*)h->Aexp.p,
                           (const double*)h->AexpT.p, mi, l0, l0s, Lm, K, ah, bh, hx, gx, llb, lz, zf, gsy);
    }
    else if (K <= 16) WL(16, false); else if (K <= 32) WL(32, false);
    else if (K == 64) WL(64, true); else WL(64, false);
#undef WL
    HIPCK(hipGetLastError());
    return 0;
  }
  CK(launch_lin_init(h, b0, nb, Lm, stream));
  const LinChain none = {};
#define SWPX(NWV, F, BSV)                                                                                  \
  do {                                                                                                     \
    const size_t lds = sizeof(LinShared<NWV>);                                                             \
    if (lds > 64 * 1024)                                                                                   \
      hipFuncSetAttribute((const void*)k_sweeps_lin<NWV, F, 0, BSV>,                                       \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
    hipLaunchKernelGGL((k_sweeps_lin<NWV, F, 0, BSV>), grid, dim3(64 * NWV), lds, stream, Eh, kx,          \
                       (const double*)h->Aexp.p, (const double*)h->AexpT.p,                                \
                       a0v, a0e, nb, Lm, Lm, K, ah, bh, hx, gx, llb, lz, zf, none);    \
  } while (0)
#define SWP(NWV, F) SWPX(NWV, F, false)
  if (NW == 1) { if (full) SWP(1, true); else SWP(1, false); }
  else if (NW == 2) { if (full) SWP(2, true); else SWP(2, false); }
  else if (NW == 3) { if (full) SWP(3, true); else SWP(3, false); }
  else if (NW == 4) { if (full) SWP(4, true); else SWP(4, false); }
  else if (NW <= 8) { if (K == 128) SWPX(8, true, true); else SWPX(8, false, true); }      // K > 64: B streamed
  else {
    // 128 < K <= 256: eight waves of two state tiles (256 VGPRs each) instead of 16 x 1
#define SWP2(F, WT)                                                                                        \
  do {                                                                                                     \
    const size_t lds = (size_t)(WT) * sizeof(LinShared<16>);                                               \
    dim3 g2((unsigned)((nb + 16 * (WT) - 1) / (16 * (WT))), 2);                                            \
    hipFuncSetAttribute((const void*)k_sweeps_lin2<8, F, WT>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                        (int)lds);                                                                         \
    hipLaunchKernelGGL((k_sweeps_lin2<8, F, WT>), g2, dim3(512), lds, stream, Eh, kx,                      \
                       (const double*)h->Aexp.p, (const double*)h->AexpT.p, a0v, a0e,  \
                       nb, Lm, K, ah, bh, hx, gx, llb, lz, zf);                                            \
  } while (0)
    // more 16-window workgroups than CUs: 32 windows per workgroup share the streamed transition
    // tile (variant[13] = 1: off)
    const bool w32 = 2 * ((nb + 15) / 16) > h->ncu && h->variant[13] != 1;     // more workgroups than CUs
    // 128 < K <= 192: twelve waves of one state tile (three per SIMD; the eight two-tile waves would
    // run four empty tiles: 4.85 against 6.1 ms at K = 192, D = 32, T = 1e6).  variant[7] = 1: one
    // tile per wave for every K, 2: two tiles per wave for every K
    if (NW <= 12 && h->variant[7] != 2) { if (K == 192) SWPX(12, true, true); else SWPX(12, false, true); }
    else if (h->variant[7] == 1) { if (K == 256) SWPX(16, true, true); else SWPX(16, false, true); }
    else if (w32) { if (K == 256) SWP2(true, 2); else SWP2(false, 2); }
    else if (K == 256) SWP2(true, 1); else SWP2(false, 1);
#undef SWP2
  }
#undef SWP
#undef SWPX
  HIPCK(hipGetLastError());
  return 0;
}
// var_x of the last scaled sweep (B windows of length Lm), formed on first use
int ensure_q(svihmm_ctx* h, int B, int Lm, hipStream_t stream) {
  if (!h->lin_mode || h->q_valid) return 0;
  const int K = h->K;
  const int64_t n = (int64_t)B * Lm;
  CK(ensure(h->q, (size_t)n * K * sizeof(double)));
  ProfScope ps(h, KS_POSTERIOR, stream);
  dim3 grid((unsigned)((n + 15) / 16));
#define PQ(KT) hipLaunchKernelGGL(k_lin_posterior<KT>, grid, dim3(256), 0, stream, (const double*)h->la.p, \
                                  (const double*)h->lb.p, (const double*)h->hx.p, (const double*)h->gx.p,   \
                                  (const double2*)h->zfac.p, n, Lm, K, (double*)h->q.p)
#define PQF(KT) hipLaunchKernelGGL((k_lin_posterior<KT, float>), grid, dim3(256), 0, stream, (const float*)h->la.p, \
                                   (const float*)h->lb.p, (const double*)h->hx.p, (const double*)h->gx.p,          \
                                   (const double2*)h->zfac.p, n, Lm, K, (double*)h->q.p)
  if (h->cur_f32) { if (K <= 16) PQF(1); else if (K <= 32) PQF(2); else if (K <= 48) PQF(3); else if (K <= 64) PQF(4);
                    else if (K <= 128) PQF(8); else if (K <= 192) PQF(12); else PQF(16); }
  else if (K <= 16) PQ(1); else if (K <= 32) PQ(2); else if (K <= 48) PQ(3); else if (K <= 64) PQ(4);
  else if (K <= 128) PQ(8); else if (K <= 192) PQ(12); else PQ(16);
#undef PQ
#undef PQF
  HIPCK(hipGetLastError());
  h->q_valid = true;
  return 0;
}
// ELBO total still owed to packed[last] (set by the scaled sweeps, paid by k_finalize or here)
int launch_sum_lb(svihmm_ctx* h, int B, hipStream_t stream) {
  double* lbtot = (double*)h->packed.p + (packed_len(h) - 1);
  hipLaunchKernelGGL(k_sum_lb, dim3(1), dim3(256), 0, stream, (const double*)h->local_lb.p, B, lbtot);
  HIPCK(hipGetLastError());
  return 0;
}
// One long window (B = 1, the full-chain E-step) as an exact blocked scan over chunks of
// CHAIN_L steps: chunk matrices (S1), boundary vectors (S2), all chunks as concurrent windows
// with boundary conditions (S3).  See the comment above LinChain in kernels_recursion.h.
// chunk length: 256 steps give the most concurrent windows in S3; very long chains use 1024 so
// that the sequential boundary scan (S2) stays short (S1's work does not depend on it)
static int chain_len(int Lm) { return Lm >= 512 * 1024 ? 1024 : 256; }
bool use_chain(const svihmm_ctx* h, int B, int Lm) {
  return B == 1 && h->K <= 256 && Lm >= 2048 && h->variant[6] != 1 && !h->exact_log;
}
int launch_fb_chain(svihmm_ctx* h, int Lm, bool total) {
  const int K = h->K, T = Lm, L = chain_len(Lm);
  // state tiles the sweep kernels are instantiated for: exact up to 64 states, 8 / 16 tiles with
  // the transition tile streamed beyond (the chunk matrices are laid out for that width)
  const int NWt = (K + 15) / 16;
  const int NW = NWt <= 4 ? NWt : NWt <= 8 ? 8 : 16, Kp = 16 * NW;
  const bool full = (K == Kp);
  const int Cfull = (T - 2) / L;            // interior chunks; the tail chunk has 1..L steps
  const int C = Cfull + 1;
  const int ltail = T - 1 - Cfull * L;      // steps of the tail chunk (rows Cfull*L .. T-1)
  CK(ensure_fb_lin(h, 1, Lm));
  // chunk matrices + transposes + row exponents | boundary vectors and exponents | per-chunk scalars
  const size_t nM = (size_t)C * Kp * K;
  CK(ensure(h->chain, (2 * nM + (size_t)C * Kp + 2 * (size_t)(C + 1) * K + 7 * (size_t)(C + 1) + 8) * sizeof(double)));
  double* Mm = (double*)h->chain.p;
  double* MmT = Mm + nM;
  double* Mh = MmT + nM;
  double* abnd = Mh + (size_t)C * Kp;
  double* bbnd = abnd + (size_t)(C + 1) * K;
  double* aexp = bbnd + (size_t)(C + 1) * K;
  double* bexp = aexp + (C + 1);
  double* kbef = bexp + (C + 1);
  double* lbw = kbef + (C + 1);             // per-chunk local_lb
  double* lzw = lbw + (C + 1);              // scratch logz of the S3 windows
  double* ksum = lzw + (C + 1);             // per-chunk sums of the emission row exponents
  h->chain_kbef = kbef; h->chain_C = C; h->chain_L = L; h->chain_T = T;
  CK(ensure(h->chain2, (size_t)(C + 1) * sizeof(double2)));
  double2* zfw = (double2*)h->chain2.p;     // scratch zfac of the S3 windows (the global one comes from S2)
  const double* Eh = (const double*)(h->eh_in_llE ? h->llE.p : h->ll.p);
  const double* kx = (const double*)h->kexp.p;
  double* ah = (double*)h->la.p; double* bh = (double*)h->lb.p;
  double* hx = (double*)h->hx.p; double* gx = (double*)h->gx.p;
  const double* A = (const double*)h->Aexp.p; const double* At = (const double*)h->AexpT.p;
  const double* a0v = (const double*)h->a0v.p;
  const double* a0e = (const double*)h->a0e.p;
  hipStream_t st = h->stream;
  ProfScope ps(h, KS_FB, st);
  CK(launch_lin_init(h, 0, 1, Lm, st));
#define SWPM(NWV, F, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH)                       \
  hipLaunchKernelGGL((k_sweeps_lin<NWV, F, MD>), GRID, dim3(64 * NWV), sizeof(LinShared<NWV>), st, EHP, KXP, A, At, \
                     a0v, a0e, BB, LL, WS, K, AH, BH, HX, GX, LB, LZ, ZF, CH)
#define SWPB(NWV, F, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH)                       \
  do {                                                                                                      \
    hipFuncSetAttribute((const void*)k_sweeps_lin<NWV, F, MD, true>,                                        \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LinShared<NWV>));           \
    hipLaunchKernelGGL((k_sweeps_lin<NWV, F, MD, true>), GRID, dim3(64 * NWV), sizeof(LinShared<NWV>), st,  \
                       EHP, KXP, A, At, a0v, a0e, BB, LL, WS, K, AH, BH, HX, GX, LB, LZ, ZF, CH);                 \
  } while (0)
#define SWPD(MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH)                                \
  do {                                                                                                      \
    if (NW == 8) { if (full) SWPB(8, true, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); \
                   else SWPB(8, false, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); }   \
    else if (NW == 16) { if (full) SWPB(16, true, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); \
                         else SWPB(16, false, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); } \
    else if (NW == 1) { if (full) SWPM(1, true, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); \
                   else SWPM(1, false, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); }   \
    else if (NW == 2) { if (full) SWPM(2, true, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); \
                        else SWPM(2, false, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); } \
    else if (NW == 3) { if (full) SWPM(3, true, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); \
                        else SWPM(3, false, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); } \
    else { if (full) SWPM(4, true, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH);         \
           else SWPM(4, false, MD, GRID, BB, LL, WS, EHP, KXP, AH, BH, HX, GX, LB, LZ, ZF, CH); }           \
  } while (0)
  LinChain ch = {};
  ch.Mout = Mm; ch.MoutT = MmT; ch.Mh = Mh;
  // S1: chunk matrices.  Interior chunks: L steps (rows c*L .. c*L+L); tail: ltail steps.
  if (Cfull > 0)
    SWPD(2, dim3((unsigned)(Cfull * NW), 1), Cfull, L + 1, L, Eh, kx, ah, bh, hx, gx, lbw, lzw, zfw, ch);
  {
    LinChain ct = ch;
    ct.Mout = Mm + (size_t)Cfull * Kp * K; ct.MoutT = MmT + (size_t)Cfull * Kp * K; ct.Mh = Mh + (size_t)Cfull * Kp;
    const size_t ro = (size_t)Cfull * L;
    SWPD(2, dim3((unsigned)NW, 1), 1, ltail + 1, L, Eh + ro * K, kx + ro, ah, bh, hx, gx, lbw, lzw, zfw, ct);
  }
  // S2: boundary vectors, Z
  hipLaunchKernelGGL(k_chunk_ksum, dim3(C), dim3(64), 0, st, kx, C, L, (int64_t)T, ksum);
#define SCAN(KM)                                                                                                   \
  do {                                                                                                             \
    const size_t lds = (size_t)4 * KM * 64 * sizeof(double);                                                       \
    if (lds > 64 * 1024)                                                                                           \
      hipFuncSetAttribute((const void*)k_chunk_scan<KM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
    hipLaunchKernelGGL(k_chunk_scan<KM>, dim3(2), dim3(256), lds, st, (const double*)Mm, (const double*)MmT,       \
                       (const double*)Mh, C, Kp, K, Eh, (const double*)ksum, a0v, a0e, abnd, aexp, bbnd, bexp, kbef,     \
                       (double2*)h->zfac.p, (double*)h->logz.p);                                                   \
  } while (0)
  if (K <= 16) SCAN(16); else if (K <= 32) SCAN(32); else if (K <= 64) SCAN(64);
  else
    hipLaunchKernelGGL(k_chunk_scan_wide, dim3(2), dim3(256), 0, st, (const double*)Mm, (const double*)MmT,
                       (const double*)Mh, C, Kp, K, Eh, (const double*)ksum, a0v, a0e, abnd, aexp, bbnd, bexp, kbef,
                       (double2*)h->zfac.p, (double*)h->logz.p);
#undef SCAN
  // S3: every chunk as a window with boundary conditions
  ch.init_vec = abnd; ch.init_exp = aexp; ch.kbefore = kbef;
  ch.term_vec = bbnd + K; ch.term_exp = bexp + 1;       // window c ends at boundary c + 1
  if (Cfull > 0)
    SWPD(1, dim3((unsigned)((Cfull + 15) / 16), 2), Cfull, L + 1, L, Eh, kx, ah, bh, hx, gx, lbw, lzw, zfw, ch);
  {
    LinChain ct = ch;
    ct.init_vec = abnd + (size_t)Cfull * K; ct.init_exp = aexp + Cfull; ct.kbefore = kbef + Cfull;
    const size_t ro = (size_t)Cfull * L;
    SWPD(3, dim3(1, 2), 1, ltail + 1, L, Eh + ro * K, kx + ro, ah + ro * K, bh + ro * K, hx + ro, gx + ro,
         lbw + Cfull, lzw + Cfull, zfw + Cfull, ct);
  }
#undef SWPD
#undef SWPB
#undef SWPM
  HIPCK(hipGetLastError());
  // local_lb[0] = sum of the chunks' parts (fixed order)
  hipLaunchKernelGGL(k_sum_lb, dim3(1), dim3(256), 0, st, (const double*)lbw, C, (double*)h->local_lb.p);
  if (total) {
    double* lbtot = (double*)h->packed.p + (packed_len(h) - 1);
    hipLaunchKernelGGL(k_sum_lb, dim3(1), dim3(256), 0, st, (const double*)lbw, C, lbtot);
  }
  HIPCK(hipGetLastError());
  return 0;
}

int launch_fb_lin(svihmm_ctx* h, int B, int Lm, bool total) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  if (use_chain(h, B, Lm)) {
    if (h->svi_flags && h->in_svi_estep) CK(wait_globals(h));    // (the chain kernels take no gate)
    return launch_fb_chain(h, Lm, total);
  }
  CK(ensure_fb_lin(h, B, Lm));
  CK(launch_fb_lin_range(h, 0, B, Lm, h->stream));
  if (total) h->lb_pending = B;   // summed by k_finalize's extra workgroup (or flush_lb)
  return 0;
}
int flush_lb(svihmm_ctx* h, hipStream_t stream) {
  if (!h->lb_pending) return 0;
  const int B = h->lb_pending;
  h->lb_pending = 0;
  return launch_sum_lb(h, B, stream);
}

int launch_scale_ll(svihmm_ctx* h, int B, int Lm) {
  const int64_t n = (int64_t)B * Lm;
  const int K = h->K;
  CK(ensure(h->llE, (size_t)n * K * sizeof(double)));
  CK(ensure(h->kexp, (size_t)n * sizeof(double)));
  ProfScope ps(h, KS_EMISSION);
  dim3 grid((unsigned)((n + 15) / 16));
#define SC(KT) hipLaunchKernelGGL(k_scale_ll<KT>, grid, dim3(256), 0, h->stream, (const double*)h->ll.p, \
                                  n, K, (double*)h->llE.p, (double*)h->kexp.p)
  if (K <= 16) SC(1); else if (K <= 32) SC(2); else if (K <= 48) SC(3); else if (K <= 64) SC(4);
  else if (K <= 128) SC(8); else if (K <= 192) SC(12); else SC(16);
#undef SC
  HIPCK(hipGetLastError());
  return 0;
}

int launch_scale_ll_f32(svihmm_ctx* h, int B, int Lm) {
  const int64_t n = (int64_t)B * Lm;
  const int K = h->K;
  CK(ensure(h->kexp, (size_t)n * sizeof(double)));
  ProfScope ps(h, KS_EMISSION);
  dim3 grid((unsigned)((n + 15) / 16));
#define SCF(KT) hipLaunchKernelGGL(k_scale_ll_f32<KT>, grid, dim3(256), 0, h->stream, (float*)h->ll.p, n, K, (double*)h->kexp.p)
  if (K <= 128) SCF(8); else if (K <= 192) SCF(12); else SCF(16);
#undef SCF
  HIPCK(hipGetLastError());
  return 0;
}

// Log-domain lliks / lalpha / lbeta of windows [b0, b0+nb) of the last (scaled) sweep,
// recomputed by the log-domain kernels into the m_* side buffers.
int materialise(svihmm_ctx* h, int b0, int nb) {
  if (h->m_nb > 0 && b0 >= h->m_b0 && b0 + nb <= h->m_b0 + h->m_nb) return 0;
  CK(wait_globals(h));
  if (h->lin_stale)
    return fail("log-domain intermediates of the last E-step are rebuilt on demand and the "
                "observations / globals / emission parameters have changed since: read them "
                "before the next parameter upload");
  const int Lm = h->lastLm, K = h->K;
  const size_t n = (size_t)nb * Lm * K * sizeof(double);
  CK(ensure(h->m_la, n));
  CK(ensure(h->m_lb, n));
  const double* ll;
  if (h->eh_in_llE) {   // the plain lliks are still in h->ll
    ll = (const double*)h->ll.p + (size_t)b0 * Lm * K;
  } else {
    CK(ensure(h->m_ll, n));
    CK(launch_emission(h, nb, Lm, h->last_flags, false, (const int64_t*)h->starts.p + b0,
                       (double*)h->m_ll.p));
    ll = (const double*)h->m_ll.p;
  }
  if (h->lastB == 1 && K <= 256 && h->chain_T == Lm && h->chain_kbef && use_chain(h, 1, Lm)) {
    // the blocked scan's messages -> logs (k_chain_lalpha both ways), entries lost to underflow
    // recomputed in the log domain row by row (k_lalpha_fix / k_lbeta_fix): no sequential pass
    const size_t ne = (size_t)Lm * K;
    CK(ensure(h->scratch, (2 * ne + 8) * sizeof(double)));
    double* ta = (double*)h->scratch.p;
    double* tb = ta + ne;
    double* ktop = tb + ne;
    const int64_t T = Lm;
    ProfScope ps(h, KS_FB);
    hipLaunchKernelGGL(k_ksum_all, dim3(1), dim3(256), 0, h->stream, (const double*)h->kexp.p, T, ktop);
    hipLaunchKernelGGL(k_chain_lalpha, dim3(h->chain_C), dim3(256), 0, h->stream, (const double*)h->la.p,
                       (const double*)h->hx.p, (const double*)h->kexp.p, h->chain_kbef, h->chain_L,
                       h->chain_C, T, K, ta, (const double*)nullptr);
    hipLaunchKernelGGL(k_chain_lalpha, dim3(h->chain_C), dim3(256), 0, h->stream, (const double*)h->lb.p,
                       (const double*)h->gx.p, (const double*)h->kexp.p, h->chain_kbef, h->chain_L,
                       h->chain_C, T, K, tb, (const double*)ktop);
    const unsigned nblk = (unsigned)((T + 63) / 64);
#define LFIX(KM)                                                                                                  \
  do {                                                                                                            \
    hipLaunchKernelGGL(k_lalpha_fix<KM>, dim3(nblk), dim3(256), 0, h->stream, (const double*)ta,                  \
                       (const double*)h->la.p, ll, (const double*)h->ltran.p, (const double*)h->mod_init.p, T, K, \
                       (double*)h->m_la.p);                                                                       \
    hipLaunchKernelGGL(k_lbeta_fix<KM>, dim3(nblk), dim3(256), 0, h->stream, (const double*)tb,                   \
                       (const double*)h->lb.p, ll, (const double*)h->ltran.p, T, K, (double*)h->m_lb.p);          \
  } while (0)
    if (K <= 16) LFIX(16); else if (K <= 32) LFIX(32); else if (K <= 64) LFIX(64);
    else {
      hipLaunchKernelGGL(k_lalpha_fix_wide, dim3(nblk), dim3(256), 0, h->stream, (const double*)ta,
                         (const double*)h->la.p, ll, (const double*)h->ltran.p, (const double*)h->mod_init.p, T, K,
                         (double*)h->m_la.p);
      hipLaunchKernelGGL(k_lbeta_fix_wide, dim3(nblk), dim3(256), (size_t)4 * K * sizeof(double), h->stream,
                         (const double*)tb, (const double*)h->lb.p, ll, (const double*)h->ltran.p, T, K,
                         (double*)h->m_lb.p);
    }
#undef LFIX
    HIPCK(hipGetLastError());
  } else {
    CK(launch_fb(h, nb, Lm, 0, 2, ll, (double*)h->m_la.p, (double*)h->m_lb.p));
  }
  h->m_b0 = b0; h->m_nb = nb;
  return 0;
}

// backward sampling from the device-resident lalpha[T,K] (hmm_fast.pyx:97-122): blocked
// composition of the per-row draw maps (K <= 64, T >= 1024), else the sequential single-wave
// sampler.  *dz_out: device int64[T] (in h->scratch), valid until the next call.
static int ffbs_draw(svihmm_ctx* h, const double* la, int64_t T, int K, const double* logA,
                     const double* uniforms, int64_t** dz_out) {
  const bool blocked = K <= 256 && T >= 1024 && h->variant[6] != 1;
  const int KS = K <= 16 ? 16 : K <= 32 ? 32 : K <= 64 ? 64 : 256;      // path entries per row
  const int Ls = T >= 65536 ? 512 : 256;
  const int Cs = (int)((T + Ls - 1) / Ls);
  const size_t base = ((size_t)K * K + (size_t)T) * sizeof(double) + (size_t)T * sizeof(int64_t);
  const size_t extra = blocked ? (size_t)T * KS + 2 * (size_t)Cs * KS + (size_t)Cs + 64 : 0;
  CK(ensure(h->scratch, base + extra));
  double* dlogA = (double*)h->scratch.p;
  double* dun = dlogA + (size_t)K * K;
  int64_t* dz = (int64_t*)(dun + T);
  *dz_out = dz;
  unsigned char* path = (unsigned char*)(dz + T);
  unsigned char* mA = path + (size_t)T * KS;
  unsigned char* mB = mA + (size_t)Cs * KS;
  unsigned char* entry = mB + (size_t)Cs * KS;
  HIPCK(hipMemcpyAsync(dlogA, logA, (size_t)K * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(dun, uniforms, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, KS_FFBS);
    if (blocked) {
#define FPATH(KM)                                                                                          \
  hipLaunchKernelGGL(k_ffbs_paths<KM>, dim3(Cs), dim3(64), ((size_t)K * (KM + 1) + 2 * KM) * sizeof(double), \
                     h->stream, la, (const double*)dlogA, (const double*)dun, T, K, Ls, path)
      if (KS == 16) FPATH(16); else if (KS == 32) FPATH(32); else if (KS == 64) FPATH(64);
      else
        hipLaunchKernelGGL(k_ffbs_paths_wide, dim3(Cs), dim3(256), 0, h->stream, la, (const double*)dlogA,
                           (const double*)dun, T, K, Ls, path);
#undef FPATH
      hipLaunchKernelGGL(k_ffbs_compose, dim3(1), dim3(1024), 0, h->stream, (const unsigned char*)path, T, KS,
                         Ls, Cs, mA, mB, entry);
      hipLaunchKernelGGL(k_ffbs_gather, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, h->stream,
                         (const unsigned char*)path, (const unsigned char*)entry, T, KS, Ls, dz);
    } else {
      hipLaunchKernelGGL(k_ffbs_sample, dim3(1), dim3(64), K > 64 ? (size_t)K * 8 : 0, h->stream,
                         la, (const double*)dlogA, (const double*)dun, T, K, dz);
    }
    HIPCK(hipGetLastError());
  }
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
  const int K = h->K;
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  CK(wait_globals(h));
  // forward filter: long chains through the exact blocked scan (scaled sweeps), then lalpha
  // from (ah, h, K); short ones with the per-window log-domain kernel
  const double* la = nullptr;
  if (K <= 256 && use_chain(h, 1, (int)T)) {
    CK(prepare_ll(h, &st0, 1, (int)T, flags, false, true));
    CK(launch_fb_chain(h, (int)T, false));
    CK(ensure(h->m_la, (size_t)T * K * sizeof(double)));
    h->m_nb = 0;
    ProfScope ps(h, KS_FB);
    hipLaunchKernelGGL(k_chain_lalpha, dim3(h->chain_C), dim3(256), 0, h->stream, (const double*)h->la.p,
                       (const double*)h->hx.p, (const double*)h->kexp.p, h->chain_kbef, h->chain_L,
                       h->chain_C, T, K, (double*)h->m_la.p);
    HIPCK(hipGetLastError());
    la = (const double*)h->m_la.p;
    if (out_lalpha) {
      // the caller wants lalpha itself: entries the scaled messages lost to underflow are
      // recomputed in the log domain (plain lliks into m_ll, corrected copy into m_lb)
      CK(ensure(h->m_lb, (size_t)T * K * sizeof(double)));
      const double* llp = (const double*)h->ll.p;     // host-supplied / two-pass lliks are still there
      if (!h->eh_in_llE) {
        CK(ensure(h->m_ll, (size_t)T * K * sizeof(double)));
        CK(launch_emission(h, 1, (int)T, flags, false, nullptr, (double*)h->m_ll.p));
        llp = (const double*)h->m_ll.p;
      }
      const unsigned nblk = (unsigned)((T + 63) / 64);
#define LFIX(KM) hipLaunchKernelGGL(k_lalpha_fix<KM>, dim3(nblk), dim3(256), 0, h->stream, la, (const double*)h->la.p, \
                                    llp, (const double*)h->ltran.p,                                              \
                                    (const double*)h->mod_init.p, T, K, (double*)h->m_lb.p)
      if (K <= 16) LFIX(16); else if (K <= 32) LFIX(32); else if (K <= 64) LFIX(64);
      else
        hipLaunchKernelGGL(k_lalpha_fix_wide, dim3(nblk), dim3(256), 0, h->stream, la, (const double*)h->la.p, llp,
                           (const double*)h->ltran.p, (const double*)h->mod_init.p, T, K, (double*)h->m_lb.p);
#undef LFIX
      HIPCK(hipGetLastError());
      la = (const double*)h->m_lb.p;
    }
  } else {
    CK(prepare_ll(h, &st0, 1, (int)T, flags, false));
    CK(launch_fb(h, 1, (int)T, 0, 1));
    la = (const double*)h->la.p;
  }
  int64_t* dz = nullptr;
  CK(ffbs_draw(h, la, T, K, logA, uniforms, &dz));
  CK(d2h(h, out_z, dz, (size_t)T * sizeof(int64_t)));
  if (out_lalpha) CK(d2h(h, out_lalpha, la, (size_t)T * K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  h->lastB = 1; h->lastLm = (int)T;
  return 0;
}

// hmm_fast.pyx:80-95: with lalpha_init supplied the reference skips the filter and only samples
int svihmm_ffbs_sample(svihmm_ctx* h, int64_t T, int32_t K, const double* lalpha, const double* logA,
                       const double* uniforms, int64_t* out_z) {
  if (!h || !lalpha || !logA || !uniforms || !out_z || T <= 0 || K <= 0)
    return fail("svihmm_ffbs_sample: bad arguments");
  if (T > 2147483647LL) return fail("svihmm_ffbs_sample: T too large");
  CK(set_device(h));
  const size_t n = (size_t)T * K * sizeof(double);
  CK(ensure(h->m_la, n));
  h->m_nb = 0;
  HIPCK(hipMemcpyAsync(h->m_la.p, lalpha, n, hipMemcpyHostToDevice, h->stream));
  int64_t* dz = nullptr;
  CK(ffbs_draw(h, (const double*)h->m_la.p, T, K, logA, uniforms, &dz));
  CK(d2h(h, out_z, dz, (size_t)T * sizeof(int64_t)));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

}  // extern "C"
