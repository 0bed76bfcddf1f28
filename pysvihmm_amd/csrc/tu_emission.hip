// tu_emission.hip -- emission kernels and their launchers
// One of the translation units of libsvihmm_hip.so (see host.h).
#include "host.h"
#include "device_helpers.h"
#include "kernels_emission.h"

extern "C" {

// NIW parameter block in h->niw ([mu | sigma | kappa | nu], on the device) -> theta (both layouts);
// logdet_out (device, [K]) optionally receives log det sigma_mf.  Asynchronous: a factor that is
// not positive definite is reported by the next synchronising call.
// fp32-mode emission (k_emission_bf16x3): shapes it takes and its parameter buffer -- EMB_NREC state
// records of EMB_REC bytes (zeroed once: the kernel's copies run up to two records + 1 KB ahead)
static bool emb_shape_ok(int K, int D) { return K <= 64 && D <= 32; }
static int emb_buffers(svihmm_ctx* h, uint4** uwp) {
  const size_t nb = (size_t)EMB_NREC * EMB_REC;
  CK(ensure(h->uwb, nb));
  if (h->uw_zero_p != h->uwb.p) {
    HIPCK(hipMemsetAsync(h->uwb.p, 0, nb, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));     // once per handle; the emission launch may sit on another stream
    h->uw_zero_p = h->uwb.p;
    h->uw_valid = false;
  }
  *uwp = (uint4*)h->uwb.p;
  return 0;
}
// ... and of k_emission_bf16x3d (round 5): 32 < D <= 64 at any K <= 1024, and every wide model (K > 64) with
// D <= 64; (K / 2 + 2) records of EMD_REC bytes
static bool emd_shape_ok(int K, int D) { return D <= 64 && K <= 1024 && (K > 64 || D > 32); }
static int emd_buffers(svihmm_ctx* h, int K, uint4** uwp) {
  const size_t nb = ((size_t)((K + 63) / 64 * 64) / 2 + 2) * EMD_REC;
  CK(ensure(h->uwd, nb));
  if (h->uwd_zero_p != h->uwd.p || h->uwd_zero_n < nb) {
    HIPCK(hipMemsetAsync(h->uwd.p, 0, h->uwd.cap, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    h->uwd_zero_p = h->uwd.p; h->uwd_zero_n = h->uwd.cap;
    h->uwd_valid = false;
  }
  *uwp = (uint4*)h->uwd.p;
  return 0;
}
// fp32 mode on a wide model (64 < K <= 256): NIW factors with D <= 64 in batches of at least 32 768 rows --
// k_emission_bf16x3d<true>, k_scale_ll_f32, k_sweeps_lin2<..., float>, k_stats_bf16x3w (round 5)
// (the statistics side of the same decision is stats_bf16w_shape_ok, tu_stats.hip: a batch enters the fp32 format
//  only when BOTH ends have their kernel -- otherwise it runs fp64, as the header promises)
bool f32_wide_ok(const svihmm_ctx* h, int64_t n) {
  return !h->emis_diag && !h->emis_cat && h->K > 64 && h->K <= 256 && h->D <= 64 && h->niw.p != nullptr &&
         h->variant[5] != 3 && h->variant[10] != 2 && (n >= cu_scaled(h, 32768) || h->variant[10] == 3) &&
         stats_bf16w_shape_ok(h, n);
}
// the step may ride in the theta builder's launch: NIW family, the split builder's widths, plain fp64 / fp32 records alike
bool step_theta_ok(const svihmm_ctx* h, int K, int D) {
  return D > 16 && D <= 32 && h->variant[13] != 2 && h->variant[13] != 3 && !(h->prec == 1 && emd_shape_ok(K, D) && !h->svi_active);
}
int launch_niw_to_theta(svihmm_ctx* h, int K, int D, double* logdet_out, const SviStepArgs* step) {
  CK(upload_feature_table(h, D, K));
  const int Fp = h->Fp, Kp = h->Kp;
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  CK(ensure(h->theta, (size_t)Fp * Kp * sizeof(double)));
  double* dmu = (double*)h->niw.p;
  double* dsg = dmu + nmu;
  double* dka = dsg + nsg;
  double* dnu = dka + K;
  // status word: pinned + mapped host memory, written (atomicMax) by the kernel only when a
  // factor is not positive definite -- no device-to-host copy on the critical path
  if (!h->pin_status) {
    HIPCK(hipHostMalloc((void**)&h->pin_status, 64, hipHostMallocMapped));
    *h->pin_status = 0;
  }
  int* dstatus = nullptr;
  HIPCK(hipHostGetDevicePointer((void**)&dstatus, h->pin_status, 0));
  // theta's padded rows / columns are zeroed once per (buffer, shape); k_niw_to_theta
  // rewrites every live entry on each call
  if (h->theta_zero_p != h->theta.p || h->theta_zero_n != (size_t)Fp * Kp) {
    HIPCK(hipMemsetAsync(h->theta.p, 0, (size_t)Fp * Kp * sizeof(double), h->stream));
    h->theta_zero_p = h->theta.p; h->theta_zero_n = (size_t)Fp * Kp;
  }
  // shapes the orbit-schedule emission kernel takes: theta is written in its layout as well
  double* orbp = nullptr;
  if (K <= 64 && D >= 8 && D <= 40 && D % 8 == 0) {
    const size_t nks = (size_t)(D / 4) * (D / 2 + 1) + (D / 2 + 1 + 3) / 4;
    const size_t nb = nks * 4 * Kp * sizeof(double);
    CK(ensure(h->theta_orb, nb));
    if (h->orb_zero_p != h->theta_orb.p || h->orb_zero_n != nb) {   // padding rows / columns: once
      HIPCK(hipMemsetAsync(h->theta_orb.p, 0, nb, h->stream));
      h->orb_zero_p = h->theta_orb.p; h->orb_zero_n = nb;
    }
    orbp = (double*)h->theta_orb.p;
  }
  // fp32 mode: the centred factors of k_emission_bf16x3 come out of the same launch (not inside the
  // device-resident SVI loop: its minibatches stay below that kernel's batch size, and a large batch
  // that follows builds them on demand)
  uint4* uwp = nullptr;
  // (also inside the device-resident SVI loop since round 5: its 64-window minibatches take the 128-row form of the kernel)
  if (h->prec == 1 && emb_shape_ok(K, D)) CK(emb_buffers(h, &uwp));
  const bool emd = h->prec == 1 && emd_shape_ok(K, D) && !h->svi_active;   // (D <= 32 at K > 64: the 64-wide builder writes the records)
  if (emd) CK(emd_buffers(h, K, &uwp));
  {
    ProfScope ps(h, KS_MISC);
#define NIWW(DM) hipLaunchKernelGGL(k_niw_to_theta_wave<DM>, dim3(K), dim3(64), 0, h->stream, (const double*)dmu, \
                                    (const double*)dsg, (const double*)dka, (const double*)dnu, K, D, Kp,   \
                                    (double*)h->theta.p, dstatus, orbp, logdet_out, uwp, h->theta_sy)
    if (step && !emd && step_theta_ok(h, K, D)) {
      const unsigned ntr = (unsigned)((K * K + 63) / 64);
      hipLaunchKernelGGL(k_svi_step_theta32s, dim3((unsigned)K + ntr), dim3(64), 0, h->stream, step->packed, step->prior_tran,
                         step->var_tran, (double*)dmu, step->prior, K, D, step->rho, step->bA, step->bE, step->nwin,
                         step->lb_keep, step->ada_G, step->sy, Kp, (double*)h->theta.p, dstatus, orbp, logdet_out, uwp,
                         h->theta_sy);
    }
    else if (step) return fail("internal: step_theta_ok and launch_niw_to_theta disagree");
    else if (emd) NIWW(64);
    else if (D <= 8) NIWW(8);
    else if (D <= 16) NIWW(16);
    else if (D > 16 && D <= 32 && h->variant[13] != 2)     // (both halves of the wave at work: round 6; variant 13 = 2: the older builder)
      hipLaunchKernelGGL(k_niw_to_theta_wave32s, dim3(K), dim3(64), 0, h->stream, (const double*)dmu, (const double*)dsg,
                         (const double*)dka, (const double*)dnu, K, D, Kp, (double*)h->theta.p, dstatus, orbp, logdet_out,
                         uwp, h->theta_sy);
    else if (D <= 32) NIWW(32);
    else if (D <= 64) NIWW(64);
    else {
      const size_t lds = (size_t)(2 * D * (D + 1) + 2 * D) * sizeof(double);
      if (lds > 64 * 1024)
        hipFuncSetAttribute((const void*)k_niw_to_theta_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_niw_to_theta_generic, dim3(K), dim3(256), lds, h->stream, (const double*)dmu,
                         (const double*)dsg, (const double*)dka, (const double*)dnu, K, D, Kp,
                         (double*)h->theta.p, dstatus, orbp, logdet_out, h->theta_sy);
    }
#undef NIWW
    HIPCK(hipGetLastError());
  }
  // status comes back asynchronously; it is examined at the next synchronising call
  h->status_pending = true;
  h->eK = K; h->eD = D; h->have_emission = true; h->emis_cat = false; h->emis_diag = false;
  h->orb_valid = orbp != nullptr;
  h->uw_valid = uwp != nullptr && !emd;
  h->uwd_valid = emd;
  return 0;
}

// Diagonal family: parameter block [mu | nus | alphas | betas] (each [K][D], means centred) in
// h->niw -> theta in the diagonal feature order.  Same asynchronous status word as the NIW builder.
int launch_diag_to_theta(svihmm_ctx* h, int K, int D) {
  CK(upload_feature_table(h, D, K, true));
  const int Fp = h->Fp, Kp = h->Kp;
  const size_t n = (size_t)K * D;
  CK(ensure(h->theta, (size_t)Fp * Kp * sizeof(double)));
  if (!h->pin_status) {
    HIPCK(hipHostMalloc((void**)&h->pin_status, 64, hipHostMallocMapped));
    *h->pin_status = 0;
  }
  int* dstatus = nullptr;
  HIPCK(hipHostGetDevicePointer((void**)&dstatus, h->pin_status, 0));
  // padded rows / columns zeroed on every family or shape change (the NIW builder keys on the same pair)
  HIPCK(hipMemsetAsync(h->theta.p, 0, (size_t)Fp * Kp * sizeof(double), h->stream));
  h->theta_zero_p = nullptr; h->theta_zero_n = 0;
  const double* p = (const double*)h->niw.p;
  {
    ProfScope ps(h, KS_MISC);
    hipLaunchKernelGGL(k_diag_to_theta, dim3(K), dim3(64), 0, h->stream, p, p + n, p + 2 * n, p + 3 * n, K, D, Kp,
                       (double*)h->theta.p, dstatus, h->theta_sy);
    HIPCK(hipGetLastError());
  }
  h->status_pending = true;
  h->eK = K; h->eD = D; h->have_emission = true; h->emis_cat = false; h->emis_diag = true; h->uw_valid = false; h->uwd_valid = false;
  h->orb_valid = false;
  return 0;
}

// the emission launch launch_emission held back for the fused E-step kernel, sent as a launch of its own after all
// (the batch did not take the fused kernel)
int launch_emission_deferred(svihmm_ctx* h) {
  if (!h->em_def.active) return 0;
  const svihmm_ctx::EmDeferred d = h->em_def;
  h->em_def.active = false;
  const int D = h->D, K = h->K;
  const int NT = 4, LEN = D + D / 2 + 1;
  const int64_t n = (int64_t)d.B * d.Lm;
  const int R0 = 3 * NT * 256 > 16 * LEN + 1 ? 3 * NT * 256 : ((16 * LEN + 1) & ~1);
  const size_t lds = (size_t)(R0 + 16 * NT + 16) * 8 + 16;
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  ProfScope ps(h, KS_EMISSION, h->stream);
  hipLaunchKernelGGL((k_emission_orbit_ks<4>), dim3((unsigned)((n + 15) / 16)), dim3(256), lds, h->stream,
                     (const double*)h->obs.p, mk, d.starts, n, d.Lm, D, K, (const double*)h->theta_orb.p, d.flags, d.out,
                     d.kexp, d.ll0, d.starts_copy, d.nstarts);
  HIPCK(hipGetLastError());
  return 0;
}


// scaled: write (Eh, kexp) for the linear-domain sweeps instead of ll (K <= 64 only).
// starts_dev / out: window starts and destination (default: the handle's buffers).
int launch_emission(svihmm_ctx* h, int B, int Lm, uint32_t flags, bool scaled,
                           const int64_t* starts_dev, double* out,
                           double* kexp_out, hipStream_t stream,
                           size_t min_lds, double* ll0_out) {
  if (!h->have_emission) return fail("no emission parameters: call svihmm_set_emission_niw");
  if (h->eD != h->D) return fail("emission D does not match obs D");
  if (!h->have_globals || h->eK != h->K) return fail("emission K does not match globals K");
  const int64_t n = (int64_t)B * Lm;
  const int D = h->D, K = h->K, Kp = h->Kp;
  if (!out) {
    CK(ensure(h->ll, (size_t)n * K * sizeof(double)));
    out = (double*)h->ll.p;
  }
  // window starts whose pull is still owed (SVI loop, upload_starts): the orbit kernel below takes them from
  // the pinned slot itself and leaves the device copy behind; every other kernel gets the device copy first
  const bool own_starts = starts_dev == nullptr;
  auto device_starts = [&]() -> int {
    if (own_starts) { CK(ensure_starts_pulled(h)); starts_dev = (const int64_t*)h->starts.p; }
    return 0;
  };
  if (!own_starts) CK(ensure_starts_pulled(h));
  if (scaled && !kexp_out) {
    CK(ensure(h->kexp, (size_t)n * sizeof(double)));
    kexp_out = (double*)h->kexp.p;
  }
  if ((scaled || (h->cur_f32 && K > 64)) && !ll0_out) {      // first-row log-likelihoods of every window (k_lin_init)
    CK(ensure(h->ll0, (size_t)B * K * sizeof(double)));
    ll0_out = (double*)h->ll0.p;
  }
  if (!stream) stream = h->stream;
  const uint8_t* mk = h->have_mask ? (const uint8_t*)h->mask.p : nullptr;
  if (scaled && h->cur_f32) flags |= SVIHMM_INT_ST32;
  ProfScope ps(h, KS_EMISSION, stream);
  if (h->emis_cat) {   // table lookup (scaled output: the caller adds the k_scale_ll pass)
    if (scaled) return fail("internal: Categorical emission has no fused scaled output");
    CK(device_starts());
    if (h->shifted) {          // (a shift moved the symbol column after the table was set)
      if (stream != h->stream) return fail("internal: Categorical lookup on a side stream over a centred column");
      CK(cat_uncentre(h));
    }
    hipLaunchKernelGGL(k_emission_cat, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream,
                       (const double*)h->obs.p, mk, starts_dev, n, Lm, K, h->V,
                       (const double*)h->cat_table.p, flags, out);
    HIPCK(hipGetLastError());
    return 0;
  }
  // fp32 mode, large batches of a NIW model with K <= 64, D <= 32: the centred bf16 x 3 kernel
  // (variant[5] = 3: the fp64 feature GEMM also in this mode)
  if (!h->emis_diag && scaled && (flags & SVIHMM_INT_ST32) && emb_shape_ok(K, D) && h->niw.p &&
      h->variant[5] != 3 && min_lds == 0 && (n >= 8192 || h->variant[5] == 4)) {
    uint4* uwp = nullptr;
    const int64_t* pend = nullptr;
    int pend_n = 0;
    if (own_starts && h->starts_pending && h->starts_pending_n == B && stream == h->stream) {
      pend = h->starts_pending; pend_n = B; h->starts_pending = nullptr;
      starts_dev = pend;
    } else CK(device_starts());
    CK(emb_buffers(h, &uwp));
    if (!h->uw_valid) {   // the mode was switched on after the parameter upload: factors from the resident NIW block
      const double* dmu = (const double*)h->niw.p;
      const double* dsg = dmu + (size_t)K * D;
      const double* dka = dsg + (size_t)K * D * D;
      const double* dnu = dka + K;
#define NIWU(DM) hipLaunchKernelGGL(k_niw_to_theta_wave<DM>, dim3(K), dim3(64), 0, stream, dmu, dsg, dka, dnu, K, D, Kp, \
                                    (double*)nullptr, (int*)nullptr, (double*)nullptr, (double*)nullptr, uwp)
      if (D <= 8) NIWU(8); else if (D <= 16) NIWU(16); else NIWU(32);
#undef NIWU
      HIPCK(hipGetLastError());
      h->uw_valid = true;
    }
    // fewer than one 256-row workgroup per CU (the 64-window minibatch): 128-row workgroups, one row tile per wave
    const int MTe = (n + 255) / 256 >= 256 ? 2 : 1;
    const size_t lds = (size_t)EMB_REC + (size_t)4 * (32 * MTe) * 64 * 4;  // MT = 2: two workgroups per CU
    if (!h->emb_attr_set) {   // (per handle = per device)
      HIPCK(hipFuncSetAttribute((const void*)k_emission_bf16x3<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)EMB_REC + (size_t)4 * 64 * 64 * 4)));
      HIPCK(hipFuncSetAttribute((const void*)k_emission_bf16x3<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)EMB_REC + (size_t)4 * 32 * 64 * 4)));
      h->emb_attr_set = true;
    }
    if (MTe == 2)
      hipLaunchKernelGGL(k_emission_bf16x3<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, stream,
                         (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, (const char*)uwp,
                         flags, (float*)out, kexp_out, ll0_out, pend ? (int64_t*)h->starts.p : (int64_t*)nullptr, pend_n);
    else if (h->variant[5] == 8)        // (one group of four waves: the form of the round's first half)
      hipLaunchKernelGGL(k_emission_bf16x3<1>, dim3((unsigned)((n + 127) / 128)), dim3(256), lds, stream,
                         (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, (const char*)uwp,
                         flags, (float*)out, kexp_out, ll0_out, pend ? (int64_t*)h->starts.p : (int64_t*)nullptr, pend_n);
    else {
      // minibatches: the state pairs of a row tile split over groups of waves on different SIMDs -- 32-row
      // workgroups of four one-wave groups up to ~2 tiles per CU (the 64-window minibatch: 514 workgroups, 22 us
      // against 26 for the 128-row form; 30 windows: 12), 64-row workgroups of two two-wave groups above
      // (tools/probe/emb_probe.hip)
#define EMH(NHV, RWV) hipLaunchKernelGGL((k_emission_bf16x3h<NHV, RWV>), dim3((unsigned)((n + 32 * RWV - 1) / (32 * RWV))), \
                         dim3(64 * (NHV) * (RWV)), (size_t)(NHV) * EMB_REC + (size_t)(RWV) * 32 * 64 * 4, stream,      \
                         (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K, (const char*)uwp,                       \
                         flags, (float*)out, kexp_out, ll0_out, pend ? (int64_t*)h->starts.p : (int64_t*)nullptr, pend_n)
      if ((n + 31) / 32 <= 520) EMH(4, 1); else EMH(2, 2);
#undef EMH
    }
    HIPCK(hipGetLastError());
    return 0;
  }
  // fp32 mode, 32 < D <= 64 (K <= 64: scaled output) and wide models (K > 64: plain float output, the
  // scaling pass follows): the centred bf16 x 3 kernel with 64-dimension pair records (round 5)
  const bool wide32 = !scaled && h->cur_f32 && K > 64;
  if (!h->emis_diag && ((scaled && (flags & SVIHMM_INT_ST32) && K <= 64) || wide32) && emd_shape_ok(K, D) && h->niw.p &&
      h->variant[5] != 3 && min_lds == 0 && (n >= cu_scaled(h, 32768) || h->variant[10] == 3)) {
    uint4* uwp = nullptr;
    CK(device_starts());
    CK(emd_buffers(h, K, &uwp));
    if (!h->uwd_valid) {   // the mode was switched on after the parameter upload: records from the resident NIW block
      const double* dmu = (const double*)h->niw.p;
      const double* dsg = dmu + (size_t)K * D;
      const double* dka = dsg + (size_t)K * D * D;
      const double* dnu = dka + K;
      hipLaunchKernelGGL(k_niw_to_theta_wave<64>, dim3(K), dim3(64), 0, stream, dmu, dsg, dka, dnu, K, D, Kp,
                         (double*)nullptr, (int*)nullptr, (double*)nullptr, (double*)nullptr, uwp);
      HIPCK(hipGetLastError());
      h->uwd_valid = true;
    }
    const size_t lds = (size_t)2 * EMD_REC + (size_t)8 * 32 * 64 * 4;
    const int wi = wide32 ? 1 : 0;
    if (!h->emd_attr_set[wi]) {
      if (wide32) HIPCK(hipFuncSetAttribute((const void*)k_emission_bf16x3d<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      else HIPCK(hipFuncSetAttribute((const void*)k_emission_bf16x3d<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      h->emd_attr_set[wi] = true;
    }
    dim3 grid((unsigned)((n + 255) / 256), wide32 ? (unsigned)((K + 63) / 64) : 1u);
    if (wide32)
      hipLaunchKernelGGL(k_emission_bf16x3d<true>, grid, dim3(512), lds, stream, (const double*)h->obs.p, mk, starts_dev, n, Lm,
                         D, K, (const char*)uwp, flags, (float*)out, kexp_out, ll0_out);
    else
      hipLaunchKernelGGL(k_emission_bf16x3d<false>, grid, dim3(512), lds, stream, (const double*)h->obs.p, mk, starts_dev, n, Lm,
                         D, K, (const char*)uwp, flags, (float*)out, kexp_out, ll0_out);
    HIPCK(hipGetLastError());
    return 0;
  }
  int var = 2;      // (the VALU outer-product generation of round 1 is gone: variant[0] is ignored)
  // scaled output, D % 8 == 0: the address-free orbit schedule (variant[5] = 1 keeps K1b)
  if (!h->emis_diag && scaled && K <= 64 && D >= 8 && D <= 40 && D % 8 == 0 && h->variant[5] != 1 && min_lds == 0) {
    const int NT = Kp / 16, LEN = D + D / 2 + 1;
    const int nks = (D / 4) * (D / 2 + 1) + (D / 2 + 1 + 3) / 4;
    const int64_t* pend = nullptr;
    int pend_n = 0;
    if (own_starts && h->starts_pending && h->starts_pending_n == B && stream == h->stream) {
      pend = h->starts_pending; pend_n = B; h->starts_pending = nullptr;
      starts_dev = pend;
    } else CK(device_starts());
    if (!h->orb_valid) {
      CK(ensure(h->theta_orb, (size_t)nks * 4 * Kp * sizeof(double)));
      hipLaunchKernelGGL(k_theta_orbit, dim3(nks * 4), dim3(64), 0, stream, (const double*)h->theta.p,
                         D, Kp, NT, (double*)h->theta_orb.p);
      HIPCK(hipGetLastError());
      h->orb_valid = true;
    }
    // minibatches (fewer than 32768 rows): 16-row workgroups, the four waves split the k-steps
    // (k_emission_orbit_ks; variant[5] = 5: the 64-row form of round 3, 2: always 128 rows)
    if ((n + 127) / 128 < h->ncu && h->variant[5] != 2 && h->variant[5] != 5) {
      const int R0 = 3 * NT * 256 > 16 * LEN + 1 ? 3 * NT * 256 : ((16 * LEN + 1) & ~1);
      const size_t lds = (size_t)(R0 + 16 * NT + 16) * 8 + 16;
      dim3 grid((unsigned)((n + 15) / 16));
      if (h->em_defer_req && NT == 4 && stream == h->stream && out == (double*)h->ll.p && !(flags & SVIHMM_INT_ST32)) {
        // the fused E-step kernel computes these tiles itself (tu_fused.hip); everything but the launch is done
        h->em_def.active = true;
        h->em_def.starts = starts_dev; h->em_def.starts_copy = pend ? (int64_t*)h->starts.p : nullptr; h->em_def.nstarts = pend_n;
        h->em_def.B = B; h->em_def.Lm = Lm; h->em_def.flags = flags;
        h->em_def.out = out; h->em_def.kexp = kexp_out; h->em_def.ll0 = ll0_out;
        return 0;
      }
#define EMK(NTV) hipLaunchKernelGGL((k_emission_orbit_ks<NTV>), grid, dim3(256), lds, stream,                      \
                                    (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K,                          \
                                    (const double*)h->theta_orb.p, flags, out, kexp_out, ll0_out,                  \
                                    pend ? (int64_t*)h->starts.p : (int64_t*)nullptr, pend_n)
      if (NT == 4) EMK(4); else if (NT == 3) EMK(3); else if (NT == 2) EMK(2); else EMK(1);
#undef EMK
      HIPCK(hipGetLastError());
      return 0;
    }
    // fewer than one 128-row workgroup per CU: 64-row workgroups
    const int MTo = ((n + 127) / 128 < h->ncu && h->variant[5] != 2) ? 1 : 2;
    const int rows = 64 * MTo;
    // (+ 1 KB at D = 32 with two row tiles per wave: every second block of eight rows starts eight slots later, EO_SHIFT)
    const size_t lds = (size_t)rows * LEN * 8 + rows * 9 + ((MTo == 2 && D == 32) ? 1024 : 0);
    dim3 grid((unsigned)((n + rows - 1) / rows));
#define EMO(NTV, UV, MTV) hipLaunchKernelGGL((k_emission_orbit<NTV, UV, MTV>), grid, dim3(256), lds, stream,  \
                                        (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K,               \
                                        (const double*)h->theta_orb.p, flags, out, kexp_out, ll0_out,               \
                                        pend ? (int64_t*)h->starts.p : (int64_t*)nullptr, pend_n)
#define EMOM(NTV, UV) do { if (MTo == 1) EMO(NTV, UV, 1); else EMO(NTV, UV, 2); } while (0)
    if (D % 16 == 0) { if (NT == 4) EMOM(4, 4); else if (NT == 3) EMOM(3, 4); else if (NT == 2) EMOM(2, 4); else EMOM(1, 4); }
    else             { if (NT == 4) EMOM(4, 2); else if (NT == 3) EMOM(3, 2); else if (NT == 2) EMOM(2, 2); else EMOM(1, 2); }
#undef EMOM
#undef EMO
    HIPCK(hipGetLastError());
    return 0;
  }
  CK(device_starts());
  if (var == 2) {
    const int DS = (D + 2) | 1;
    int MT = h->variant[3] > 0 ? h->variant[3] : 2;
    if (MT != 2 && MT != 4) MT = 2;
    size_t lds = (size_t)(64 * MT) * DS * 8 + (size_t)h->Fp * 4 + (size_t)(64 * MT) * 9;
    if (lds > 150 * 1024 && MT == 4) { MT = 2; lds = (size_t)128 * DS * 8 + (size_t)h->Fp * 4 + 128 * 9; }
    if (lds < min_lds) lds = min_lds;   // occupancy cap: leave LDS for co-resident sweep workgroups
    if (lds > 150 * 1024 && scaled) return fail("emission: D too large for the scaled sweeps");
    if (lds > 150 * 1024 && h->emis_diag) return fail("emission: D too large for the diagonal family's kernel");
    if (lds > 150 * 1024) return fail("emission: D too large for the LDS-staged GEMM kernel");
    {
      const int ntile = Kp / 16;
      int NT = (ntile % 4 == 0) ? 4 : (ntile % 2 == 0) ? 2 : 1;
      // wide models: eight state tiles per wave -- every generated A operand feeds 8 instead of 4
      // MFMAs (variant[3] = 1: four)
      if (!scaled && MT == 2 && ntile % 8 == 0 && h->variant[3] != 1) NT = 8;
      if (scaled) { NT = ntile; MT = 2; }   // the workgroup must own whole rows (K <= 64)
      const int rows = 64 * MT;
      dim3 grid((unsigned)((n + rows - 1) / rows), ntile / NT);
#define EMM_LAUNCH(NTV, MTV, SC)                                                             \
  do {                                                                                        \
    if (lds > 64 * 1024)                                                                      \
      hipFuncSetAttribute((const void*)k_emission_mfma<NTV, MTV, SC>,                         \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL((k_emission_mfma<NTV, MTV, SC>), grid, dim3(256), lds, stream,         \
                       (const double*)h->obs.p, mk, starts_dev, n, Lm, D, K,                  \
                       Kp, h->Fp, (const double*)h->theta.p, (const int*)h->fab.p, flags,     \
                       out, kexp_out, SC ? ll0_out : (double*)nullptr);                       \
  } while (0)
      if (scaled) {
        if (NT == 4) EMM_LAUNCH(4, 2, true); else if (NT == 3) EMM_LAUNCH(3, 2, true);
        else if (NT == 2) EMM_LAUNCH(2, 2, true); else EMM_LAUNCH(1, 2, true);
      } else if (MT == 4) {
        if (NT == 4) EMM_LAUNCH(4, 4, false); else if (NT == 2) EMM_LAUNCH(2, 4, false); else EMM_LAUNCH(1, 4, false);
      } else {
        if (NT == 8) EMM_LAUNCH(8, 2, false); else if (NT == 4) EMM_LAUNCH(4, 2, false);
        else if (NT == 2) EMM_LAUNCH(2, 2, false); else EMM_LAUNCH(1, 2, false);
      }
#undef EMM_LAUNCH
    }
  }
  HIPCK(hipGetLastError());
  return 0;
}

// theta + log det of a private NIW parameter set, then the three ELBO terms per factor
// (svihmm_niw_vlb_terms; the E-step's own theta stays untouched)
int launch_niw_vlb(svihmm_ctx* h, int K, int D, const double* dmu, const double* dsg, const double* dka,
                   const double* dnu, double* th2, int* dstat, double* ld, const double* p0, double* dout) {
  const int Kp = h->Kp;
  const size_t nmu = (size_t)K * D;
  ProfScope ps(h, KS_MISC);
#define NIWV(DM) hipLaunchKernelGGL(k_niw_to_theta_wave<DM>, dim3(K), dim3(64), 0, h->stream, dmu, dsg, dka, dnu, \
                                    K, D, Kp, th2, dstat, (double*)nullptr, ld)
  if (D <= 8) NIWV(8); else if (D <= 16) NIWV(16); else if (D <= 32) NIWV(32); else if (D <= 64) NIWV(64);
  else {
    // D > 64: the workgroup-per-state factorisation (matrices in LDS), as the E-step uses there
    const size_t lds = (size_t)(2 * D * (D + 1) + 2 * D) * sizeof(double);
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_niw_to_theta_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_niw_to_theta_generic, dim3(K), dim3(256), lds, h->stream, dmu, dsg, dka, dnu, K, D, Kp,
                       th2, dstat, (double*)nullptr, ld);
  }
#undef NIWV
  hipLaunchKernelGGL(k_niw_vlb_terms, dim3(K), dim3(64), 0, h->stream, (const double*)th2,
                     (const int*)h->fab.p, h->F, D, Kp, dmu, dnu, (const double*)ld, p0, p0 + nmu, K, dout);
  HIPCK(hipGetLastError());
  return 0;
}

}  // extern "C"
