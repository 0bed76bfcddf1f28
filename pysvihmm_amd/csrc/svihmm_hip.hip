// svihmm_hip.hip -- MI355X (gfx950 / CDNA4) SVI-HMM E-step engine: kernels + C ABI.
//
// Hot path of dillonalaird/pysvihmm (see include/svihmm.h for the reference
// file:line each entry point replaces).  Everything is fp64 (the reference is
// float64 throughout, hmmbase.py:102-103).
//
// Data layout in HBM (all row-major float64):
//   obs   [T][D]                  resident for the life of the handle
//   theta [Fp][Kp]                emission parameters in "augmented feature" form:
//                                 feature f=(a,b), 0<=a<=b<=D over x~=(x_0..x_{D-1},1):
//                                 phi_f(x) = x~_a * x~_b ;  ll[t,k] = sum_f phi_f(x_t) theta[f,k]
//   ll / la / lb / q [B*Lm][K]    per-window intermediates (window-major == API layout)
//   part  [nchunk][Ftot][Kp]      per-workgroup partial statistics, Ftot = Fp + Kp
//                                 rows [0,Fp): sum_t phi_f(x_t) q[t,k]  (S, xbar, neff)
//                                 rows [Fp,Fp+Kp): sum_t q[t-1,i] q[t,k] (transition stat)
//   packed [K*K + K*D + K + K*D*D + 1]
//
// Recursions: log-domain storage, linear-domain mat-vec.  With m = max_i la[t-1,i],
//   la[t,j] = m + log( sum_i exp(la[t-1,i]-m) * exp(ltran[i,j]) ) + ll[t,j]
// is algebraically the reference's LSE_i(la[t-1,i] + ltran[i,j]) + ll[t,j]
// (hmmbase.py:295) with K exps + K logs per step instead of K^2.
#include "host.h"
#include "device_helpers.h"
#include "kernels_misc.h"
#include "kernels_svi.h"

thread_local std::string g_err;
static const char* kKernNames[SVIHMM_NKERN] = {
    "emission", "forward_backward", "posterior", "stats", "finalize", "ffbs_sample",
    "misc", "allreduce", "h2d", "d2h", "reserved0", "reserved1"};

extern "C" {

const char* svihmm_last_error(void) { return g_err.c_str(); }
int svihmm_abi_version(void) { return SVIHMM_ABI_VERSION; }
const char* svihmm_kernel_name(int32_t slot) {
  return (slot >= 0 && slot < SVIHMM_NKERN) ? kKernNames[slot] : "";
}

int svihmm_device_count(int* n_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *n_out = 0; return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *n_out = n;
  return 0;
}

int svihmm_create(int device_id, svihmm_ctx** out) {
  if (!out) return fail("svihmm_create: out is NULL");
  int n = 0;
  HIPCK(hipGetDeviceCount(&n));
  if (n <= 0) return fail("svihmm_create: no HIP device visible (the HIP path has no CPU fallback)");
  if (device_id < 0 || device_id >= n) return fail("svihmm_create: bad device id");
  svihmm_ctx* h = new svihmm_ctx();
  h->device = device_id;
  HIPCK(hipSetDevice(device_id));
  HIPCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  {   // the device's shape, once: batch-size thresholds and workgroup rounds are expressed through it (host.h, cu_scaled)
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && v > 0) h->ncu = v;
    v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxThreadsPerMultiProcessor, device_id) == hipSuccess && v >= 64)
      h->waves_per_cu = v / 64;
  }
  *out = h;
  return 0;
}

int svihmm_destroy(svihmm_ctx* h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->stream2) hipStreamSynchronize(h->stream2);   // side streams of the SVI loop may still
  if (h->stream3) hipStreamSynchronize(h->stream3);   // hold kernels that touch the buffers below
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  for (auto& p : h->pending) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
  for (auto e : h->pool) hipEventDestroy(e);
  Buf* bufs[] = {&h->obs, &h->mask, &h->mod_init, &h->ltran, &h->Aexp, &h->AexpT, &h->theta, &h->niw,
                 &h->fab, &h->starts, &h->ll, &h->la, &h->lb, &h->q, &h->lse_part,
                 &h->local_lb, &h->logz, &h->part, &h->packed, &h->scratch, &h->kexp, &h->hx, &h->gx,
                 &h->zfac, &h->llE, &h->m_ll, &h->m_la, &h->m_lb, &h->chain, &h->chain2, &h->cat_table, &h->partc, &h->theta_orb, &h->prior, &h->vlb_aux, &h->gen_z,
                 &h->svi_state, &h->svi_prior, &h->svi_work, &h->commtmp, &h->ll0, &h->a0v, &h->a0e,
                 &h->shift_d, &h->uwb, &h->uwd, &h->AexpF, &h->AexpTF, &h->svi_sync, &h->pipe_cnt,
                 &h->pipe_tabs[0].buf, &h->pipe_tabs[1].buf, &h->pipe_tabs[2].buf, &h->pipe_tabs[3].buf,
                 &h->pipe_tabs[4].buf, &h->pipe_tabs[5].buf, &h->pipe_tabs[6].buf, &h->pipe_tabs[7].buf};
  for (Buf* b : bufs) release(*b);
  if (h->vlb_host) { hipHostFree(h->vlb_host); h->vlb_host = nullptr; }
  if (h->svi_elbo) { hipHostFree(h->svi_elbo); h->svi_elbo = nullptr; }
  if (h->svi_ea) hipEventDestroy(h->svi_ea);
  if (h->svi_eb) hipEventDestroy(h->svi_eb);
  if (h->svi_ec) hipEventDestroy(h->svi_ec);
  if (h->svi_ed) hipEventDestroy(h->svi_ed);
  if (h->stream3) hipStreamDestroy(h->stream3);
  for (auto e : h->svi_ev) hipEventDestroy(e);
  for (int i = 0; i < 2; ++i) {
    if (h->ev_em[i]) hipEventDestroy(h->ev_em[i]);
    if (h->ev_sw[i]) hipEventDestroy(h->ev_sw[i]);
  }
  if (h->stream2) hipStreamDestroy(h->stream2);
  for (auto& ps : h->pins) { if (ps.p) hipHostFree(ps.p); if (ps.ev) hipEventDestroy(ps.ev); }
  for (auto& ss : h->svi_starts) if (ss.p) hipHostFree(ss.p);
  if (h->pin_status) hipHostFree(h->pin_status);
  if (h->svi_ts) hipHostFree(h->svi_ts);
  if (h->starts_slot) hipHostFree(h->starts_slot);
  if (h->mirror) hipHostFree(h->mirror);
  hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

static int check_emission_status(svihmm_ctx* h);
static int pinned(svihmm_ctx* h, size_t bytes, void** out, int* slot_out);
static int pull_small(svihmm_ctx* h, void* dst, const void* pin, size_t bytes);
static int pin_release(svihmm_ctx* h, int slot);
static double* svi_ptr(svihmm_ctx* h, int which);
static int wait_side_streams(svihmm_ctx* h);
static void sample_center(const svihmm_ctx* h, const double* obs, int64_t T, int D, std::vector<double>& c);
static int reset_shift(svihmm_ctx* h, const std::vector<double>& c, int64_t row0, int64_t nrows);
static int shift_rows(svihmm_ctx* h, const double* delta, int64_t row0, int64_t nrows, bool round_symbols);
static int store_shift(svihmm_ctx* h, const std::vector<double>& c);
static int params_follow_centre(svihmm_ctx* h, const double* delta);
static int drop_auto_status(svihmm_ctx* h);
static int d2h_sync_small(svihmm_ctx* h, void* dst, const void* src, size_t bytes);
int svihmm_set_precision(svihmm_ctx* h, int32_t mode) {
  if (!h || (mode != SVIHMM_F64 && mode != SVIHMM_F32)) return fail("svihmm_set_precision: mode must be SVIHMM_F64 or SVIHMM_F32");
  h->prec = mode;
  return 0;
}
int svihmm_get_precision(svihmm_ctx* h, int32_t* mode_out, int32_t* last_batch_f32_out) {
  if (!h) return fail("svihmm_get_precision: NULL handle");
  if (mode_out) *mode_out = h->prec;
  if (last_batch_f32_out) *last_batch_f32_out = (h->cur_f32 || h->eh_float) ? 1 : 0;
  return 0;
}
int svihmm_sync(svihmm_ctx* h) {
  CK(set_device(h));
  HIPCK(hipStreamSynchronize(h->stream));
  if (h->stream2) HIPCK(hipStreamSynchronize(h->stream2));
  if (h->stream3) HIPCK(hipStreamSynchronize(h->stream3));
  return check_emission_status(h);
}

int svihmm_set_obs(svihmm_ctx* h, const double* obs, int64_t T, int32_t D,
                   const uint8_t* mask) {
  if (!h || !obs || T <= 0 || D <= 0) return fail("svihmm_set_obs: bad arguments");
  if (D > 4095) return fail("svihmm_set_obs: D too large");
  CK(set_device(h));
  h->lin_stale = true;
  ProfScope ps(h, KS_H2D);
  CK(ensure(h->obs, (size_t)T * D * sizeof(double)));
  HIPCK(hipMemcpyAsync(h->obs.p, obs, (size_t)T * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->have_mask = mask != nullptr;
  if (mask) {
    CK(ensure(h->mask, (size_t)T));
    HIPCK(hipMemcpyAsync(h->mask.p, mask, (size_t)T, hipMemcpyHostToDevice, h->stream));
  }
  h->T = T; h->D = D; h->gen_T = 0;
  h->svi_active = false;               // a resident SVI state belonged to the previous sequence
  h->center_pending = false;
  std::vector<double> c;
  sample_center(h, obs, T, D, c);
  CK(reset_shift(h, c, 0, T));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- the handle's shift (see svihmm_ctx::shift) ---------------------------------------------
// A point inside the data: per column the mean of the finite entries of a strided sample of rows.
// Any vector works (the model is shift-equivariant); what matters is |x - c| being of the size of
// the data's spread, so that the emission GEMM's expanded quadratic form does not cancel.
static void sample_center(const svihmm_ctx* h, const double* obs, int64_t T, int D, std::vector<double>& c) {
  c.assign((size_t)D, 0.0);
  const_cast<svihmm_ctx*>(h)->center_deferred = false;
  if (h->variant[9] == 1 || !obs || T <= 0) return;      // variant 9 = 1: no automatic centring
  // a Categorical table is the active emission: the column holds symbol indices, which the lookup
  // kernels truncate to int -- it stays exactly as uploaded (round-3 advisor finding: set_emission_cat
  // followed by set_obs centred the symbols)
  if (h->have_emission && h->emis_cat) { const_cast<svihmm_ctx*>(h)->center_deferred = true; return; }
  int64_t nsamp = ((int64_t)4 << 20) / D;
  if (nsamp > 65536) nsamp = 65536;
  if (nsamp < 16) nsamp = 16;
  if (nsamp > T) nsamp = T;
  const int64_t stride = T / nsamp;
  std::vector<int64_t> cnt((size_t)D, 0);
  for (int64_t i = 0; i < nsamp; ++i) {
    const double* row = obs + (size_t)(i * stride) * D;
    for (int d = 0; d < D; ++d) {
      const double v = row[d];
      if (v > -1.7e308 && v < 1.7e308) { c[d] += v; ++cnt[d]; }
    }
  }
  for (int d = 0; d < D; ++d) {
    c[d] = cnt[d] > 0 ? c[d] / (double)cnt[d] : 0.0;
    if (!(c[d] > -1.7e308 && c[d] < 1.7e308)) c[d] = 0.0;
  }
}
// rows [row0, row0 + nrows) of the resident copy -= delta[D] (host vector)
static int shift_rows(svihmm_ctx* h, const double* delta, int64_t row0, int64_t nrows, bool round_symbols) {
  const int D = h->D;
  bool any = false;
  for (int d = 0; d < D; ++d) any = any || delta[d] != 0.0;
  if (!any || nrows <= 0) return 0;
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, (size_t)D * sizeof(double), &pin, &slot));
  std::memcpy(pin, delta, (size_t)D * sizeof(double));
  void* dpin = nullptr;
  HIPCK(hipHostGetDevicePointer(&dpin, pin, 0));
  const int64_t n = nrows * (int64_t)D;
  double* base = (double*)h->obs.p + (size_t)row0 * D;
  if (round_symbols)
    hipLaunchKernelGGL(k_shift_obs<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, base, n, D,
                       (const double*)dpin);
  else
    hipLaunchKernelGGL(k_shift_obs<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, base, n, D,
                       (const double*)dpin);
  HIPCK(hipGetLastError());
  CK(pin_release(h, slot));
  return 0;
}
// record c as the handle's shift (host + device copy)
static int store_shift(svihmm_ctx* h, const std::vector<double>& c) {
  const int D = h->D;
  h->shift = c;
  h->shifted = false;
  for (int d = 0; d < D; ++d) h->shifted = h->shifted || c[d] != 0.0;
  ++h->shift_epoch;
  CK(ensure(h->shift_d, (size_t)D * sizeof(double)));
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, (size_t)D * sizeof(double), &pin, &slot));
  std::memcpy(pin, c.data(), (size_t)D * sizeof(double));
  CK(pull_small(h, h->shift_d.p, pin, (size_t)D * sizeof(double)));
  CK(pin_release(h, slot));
  return 0;
}
// The centre moved by delta[D]: device-side parameters kept in centred coordinates follow -- the
// means of the NIW / diagonal factors in h->niw (theta is rebuilt), the prior means of a running
// SVI loop.  What the caller uploaded keeps its meaning.
static int params_follow_centre(svihmm_ctx* h, const double* delta) {
  const int D = h->D;
  bool any = false;
  for (int d = 0; d < D; ++d) any = any || delta[d] != 0.0;
  const bool live = h->have_emission && !h->emis_cat && h->eD == D && h->niw.p;   // (NIW or diagonal: means lead the block)
  const bool svi = h->svi_active && h->svi_D == D && h->svi_family != 2;   // (Categorical: no means)
  if (!any || (!live && !svi)) return 0;
  CK(wait_side_streams(h));
  if (live) CK(drop_auto_status(h));      // this rebuild supersedes what an earlier automatic one reported
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, (size_t)D * sizeof(double), &pin, &slot));
  std::memcpy(pin, delta, (size_t)D * sizeof(double));
  void* dpin = nullptr;
  HIPCK(hipHostGetDevicePointer(&dpin, pin, 0));
  if (live) {
    const int n = h->eK * D;
    hipLaunchKernelGGL(k_shift_means, dim3((n + 255) / 256), dim3(256), 0, h->stream, (double*)h->niw.p, n, D,
                       (const double*)dpin);
  }
  if (svi) {
    const int n = h->svi_K * D;
    hipLaunchKernelGGL(k_shift_means, dim3((n + 255) / 256), dim3(256), 0, h->stream, (double*)h->svi_prior.p, n, D,
                       (const double*)dpin);
  }
  HIPCK(hipGetLastError());
  CK(pin_release(h, slot));
  if (live && h->emis_diag) CK(launch_diag_to_theta(h, h->eK, D));
  else if (live) CK(launch_niw_to_theta(h, h->eK, D, svi ? svi_ptr(h, 4) : nullptr));
  if (live) h->status_auto = true;
  return 0;
}
// An explicit parameter upload supersedes whatever an automatic rebuild of theta reported (new
// observations uploaded under the previous model's factors may lie anywhere relative to them).
static int drop_auto_status(svihmm_ctx* h) {
  if (h->status_pending && h->status_auto) {
    HIPCK(hipStreamSynchronize(h->stream));
    if (h->pin_status) *h->pin_status = 0;
    h->status_pending = false;
  }
  h->status_auto = false;
  return 0;
}
// freshly written rows [row0, row0 + nrows) hold caller coordinates: c becomes the shift, rows move,
// parameters already on the device follow the change of centre
static int reset_shift(svihmm_ctx* h, const std::vector<double>& c, int64_t row0, int64_t nrows) {
  std::vector<double> delta(c);
  if ((int)h->shift.size() == h->D)
    for (int d = 0; d < h->D; ++d) delta[d] -= h->shift[d];
  CK(store_shift(h, c));
  CK(shift_rows(h, c.data(), row0, nrows, false));
  return params_follow_centre(h, delta.data());
}

int svihmm_shift_obs(svihmm_ctx* h, const double* shift) {
  if (!h || !shift) return fail("svihmm_shift_obs: bad arguments");
  if (h->T <= 0 || !h->obs.p) return fail("svihmm_shift_obs: no resident observations");
  CK(set_device(h));
  h->lin_stale = true;
  const int D = h->D;
  for (int d = 0; d < D; ++d)
    if (!(shift[d] > -1.7e308 && shift[d] < 1.7e308)) return fail("svihmm_shift_obs: shift must be finite");
  CK(wait_side_streams(h));
  CK(shift_rows(h, shift, 0, h->T, false));
  std::vector<double> c = h->shift;
  c.resize((size_t)D, 0.0);
  for (int d = 0; d < D; ++d) c[d] += shift[d];
  CK(store_shift(h, c));
  return params_follow_centre(h, shift);
}
int svihmm_get_shift(svihmm_ctx* h, double* shift_out) {
  if (!h || !shift_out) return fail("svihmm_get_shift: bad arguments");
  if (h->T <= 0) return fail("svihmm_get_shift: no resident observations");
  for (int d = 0; d < h->D; ++d) shift_out[d] = d < (int)h->shift.size() ? h->shift[d] : 0.0;
  return 0;
}

// Chunked upload for sequences that arrive in pieces (gen_synthetic.py:188-191 read_data_mmap
// yields [size, D] blocks of the on-disk float64 array): svihmm_alloc_obs sizes the resident
// copy, svihmm_set_obs_rows fills rows [row0, row0 + nrows).  Each call returns once its
// block is on the device (the caller may reuse the block).
int svihmm_alloc_obs(svihmm_ctx* h, int64_t T, int32_t D, int32_t with_mask) {
  if (!h || T <= 0 || D <= 0) return fail("svihmm_alloc_obs: bad arguments");
  if (D > 4095) return fail("svihmm_alloc_obs: D too large");
  CK(set_device(h));
  h->lin_stale = true;
  CK(ensure(h->obs, (size_t)T * D * sizeof(double)));
  h->have_mask = with_mask != 0;
  if (with_mask) {
    CK(ensure(h->mask, (size_t)T));
    HIPCK(hipMemsetAsync(h->mask.p, 0, (size_t)T, h->stream));
  }
  h->T = T; h->D = D; h->gen_T = 0;
  h->svi_active = false;
  CK(reset_shift(h, std::vector<double>((size_t)D, 0.0), 0, 0));
  h->center_pending = true;            // the first block that arrives fixes the shift
  return 0;
}
int svihmm_set_obs_rows(svihmm_ctx* h, int64_t row0, int64_t nrows, const double* obs,
                        const uint8_t* mask) {
  if (!h || !obs || row0 < 0 || nrows <= 0) return fail("svihmm_set_obs_rows: bad arguments");
  if (h->T <= 0 || row0 + nrows > h->T) return fail("svihmm_set_obs_rows: rows outside the allocated sequence");
  if (mask && !h->have_mask) return fail("svihmm_set_obs_rows: sequence was allocated without a mask");
  CK(set_device(h));
  h->lin_stale = true;
  ProfScope ps(h, KS_H2D);
  const size_t D = (size_t)h->D;
  HIPCK(hipMemcpyAsync((double*)h->obs.p + (size_t)row0 * D, obs, (size_t)nrows * D * sizeof(double),
                       hipMemcpyHostToDevice, h->stream));
  if (mask)
    HIPCK(hipMemcpyAsync((uint8_t*)h->mask.p + row0, mask, (size_t)nrows, hipMemcpyHostToDevice, h->stream));
  if (h->center_pending) {
    h->center_pending = false;
    std::vector<double> c;
    sample_center(h, obs, nrows, h->D, c);
    CK(reset_shift(h, c, 0, 0));         // (rows written before this call would be undefined anyway)
  }
  CK(shift_rows(h, h->shift.data(), row0, nrows, false));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

// SVIHMM_LTRAN_LINEAR_MIN: exp(-600) = 1e-261 is a normal double with 1e-47 of headroom for
// products with a scaled message; below it: k_fb_exact
int svihmm_set_globals(svihmm_ctx* h, int32_t K, const double* mod_init, const double* ltran) {
  if (!h || K <= 0 || !mod_init || !ltran) return fail("svihmm_set_globals: bad arguments");
  if (K > 1024) return fail("svihmm_set_globals: K > 1024 unsupported");
  CK(set_device(h));
  CK(wait_globals(h));               // a globals kernel of the SVI loop may still be writing them
  h->svi_globals_ready = false;
  h->lin_stale = true;
  const size_t kk = (size_t)K * K * sizeof(double);
  CK(ensure(h->mod_init, K * sizeof(double)));
  CK(ensure(h->ltran, kk));
  // zero rows behind each matrix: the wide-model sweeps stream whole row tiles of the transition
  // matrix up to the width they are instantiated for (128 rows for 64 < K <= 128, 256 beyond),
  // not just up to K rounded to a tile
  const int reach = K <= 64 ? (K + 15) / 16 * 16 : K <= 128 ? 128 : K <= 256 ? 256 : K;
  const size_t slack = (size_t)(reach - K + 16) * K * sizeof(double);
  CK(ensure(h->Aexp, kk + slack));
  CK(ensure(h->AexpT, kk + slack));
  if (h->slack_a != h->Aexp.p || h->slack_t != h->AexpT.p || h->slack_k != K) {   // once per (buffers, K)
    HIPCK(hipMemsetAsync((char*)h->Aexp.p + kk, 0, slack, h->stream));
    HIPCK(hipMemsetAsync((char*)h->AexpT.p + kk, 0, slack, h->stream));
    h->slack_a = h->Aexp.p; h->slack_t = h->AexpT.p; h->slack_k = K;
  }
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, kk + K * sizeof(double), &pin, &slot));
  std::memcpy(pin, ltran, kk);
  std::memcpy((char*)pin + kk, mod_init, K * sizeof(double));
  {
    ProfScope ps(h, KS_MISC);
    void* dpin = nullptr;
    HIPCK(hipHostGetDevicePointer(&dpin, pin, 0));
    hipLaunchKernelGGL(k_exp_transpose, dim3((K * K + 255) / 256), dim3(256), 0, h->stream,
                       (const double*)dpin, K, (double*)h->ltran.p, (double*)h->mod_init.p,
                       (double*)h->Aexp.p, (double*)h->AexpT.p);
  }
  HIPCK(hipGetLastError());
  CK(pin_release(h, slot));   // guards the slot until the kernel has read it
  h->K = K; h->have_globals = true;
  // dynamic range of the transition expectations (K^2 values, host side): exp(ltran) must stay a
  // normal double with headroom for the scaled recursions, a normal float for the fp32 mode
  double lmin = 0.0;
  bool finite = true;
  for (size_t i = 0; i < (size_t)K * K; ++i) {
    const double v = ltran[i];
    if (!(v > -1.7e308 && v < 1.7e308)) finite = false;
    else if (v < lmin) lmin = v;
  }
  h->exact_log = !finite || lmin < SVIHMM_LTRAN_LINEAR_MIN;
  h->f32_ok = finite && lmin > SVIHMM_LTRAN_F32_MIN;
  return 0;
}

// acquire the next staging slot (waits only if the copy that last used it is still in flight,
// i.e. six uploads ago); pin_release() records the guard event after the copy is enqueued
static int pinned(svihmm_ctx* h, size_t bytes, void** out, int* slot_out) {
  svihmm_ctx::PinSlot& ps = h->pins[h->pin_next];
  *slot_out = h->pin_next;
  h->pin_next = (h->pin_next + 1) % 6;
  if (ps.busy) { HIPCK(hipEventSynchronize(ps.ev)); ps.busy = false; }
  if (bytes > ps.cap) {
    if (ps.p) hipHostFree(ps.p);
    ps.p = nullptr; ps.cap = 0;
    HIPCK(hipHostMalloc(&ps.p, bytes + 4096, hipHostMallocMapped));   // kernels pull from the slots
    ps.cap = bytes + 4096;
  }
  if (!ps.ev) HIPCK(hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming));
  *out = ps.p;
  return 0;
}
static int pin_release(svihmm_ctx* h, int slot) {
  svihmm_ctx::PinSlot& ps = h->pins[slot];
  HIPCK(hipEventRecord(ps.ev, h->stream));
  ps.busy = true;
  return 0;
}
// call after a stream synchronisation: reports a failed NIW factorisation of the last
// svihmm_set_emission_niw (which itself returns without waiting for the device)
static int check_emission_status(svihmm_ctx* h) {
  if (!h->status_pending) return 0;
  h->status_pending = false;
  const int st = h->pin_status ? *h->pin_status : 0;
  if (h->pin_status) *h->pin_status = 0;   // sticky until read: reset only here (stream idle)
  if (st > NIW_STATUS_RANGE) {
    h->have_emission = false;
    return fail("svihmm_set_emission_niw: factor " + std::to_string(st - NIW_STATUS_RANGE - 1) +
                " lies too far from the centre of the resident observations for its spread "
                "((mu-c)' (nu/2 sigma^-1) (mu-c) > 1e9): the expanded quadratic form would lose more than "
                "5e-7 in the log-likelihoods -- svihmm_shift_obs moves the centre (svihmm_get_shift reads it)");
  }
  if (st != 0) {
    h->have_emission = false;
    return fail("svihmm_set_emission_niw: sigma_mf[" + std::to_string(st - 1) + "] is not positive definite");
  }
  return 0;
}

// ---- NIW -> theta on the device (k_niw_to_theta) -------------------------------------
static inline int feat_index(int a, int b, int D) {  // 0 <= a <= b <= D
  return a * (D + 1) - a * (a - 1) / 2 + (b - a);
}

// Features are products x~_a x~_b of the augmented row x~ = (x, 1); the table lists the (a, b) the
// emission family needs.  Full covariance: all a <= b, F = (D+1)(D+2)/2.  Diagonal family:
// x_a^2 (f = a), x_a (f = D + a), 1 (f = 2 D): F = 2 D + 1.  Emission and statistics GEMMs read
// the table, so the family only changes the table and the theta builder.
int upload_feature_table(svihmm_ctx* h, int D, int K, bool diag) {
  const int F = diag ? 2 * D + 1 : (D + 1) * (D + 2) / 2, Fp = (F + 15) / 16 * 16;
  // padded state count: tiles of 16; wide models in groups of 64 (the statistics GEMM's state groups)
  const int Kp = K > 64 ? (K + 63) / 64 * 64 : (K + 15) / 16 * 16;
  if (h->tabD == D && h->Fp == Fp && h->tab_diag == diag) { h->F = F; h->Kp = Kp; return 0; }
  std::vector<int> fab(Fp, (D + 1) | ((D + 1) << 16));  // padding -> zero slot
  if (diag) {
    for (int a = 0; a < D; ++a) { fab[a] = a | (a << 16); fab[D + a] = a | (D << 16); }
    fab[2 * D] = D | (D << 16);
  } else
  for (int a = 0; a <= D; ++a)
    for (int b = a; b <= D; ++b) fab[feat_index(a, b, D)] = a | (b << 16);
  CK(ensure(h->fab, fab.size() * sizeof(int)));
  HIPCK(hipMemcpyAsync(h->fab.p, fab.data(), fab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->tabD = D; h->F = F; h->Fp = Fp; h->Kp = Kp; h->tab_diag = diag;
  return 0;
}


// means[K][D] in the caller's coordinates -> centred coordinates (and back), in place on the host
static void to_centred(const svihmm_ctx* h, double* mu, int K, int D) {
  if (!h->shifted || (int)h->shift.size() != D) return;
  for (int k = 0; k < K; ++k)
    for (int d = 0; d < D; ++d) mu[(size_t)k * D + d] -= h->shift[d];
}
static void from_centred(const svihmm_ctx* h, double* mu, int K, int D) {
  if (!h->shifted || (int)h->shift.size() != D) return;
  for (int k = 0; k < K; ++k)
    for (int d = 0; d < D; ++d) mu[(size_t)k * D + d] += h->shift[d];
}
// A Categorical table was active when the resident copy was uploaded (or un-centred it): the copy
// holds caller coordinates.  A Gaussian family that arrives now gets the centring the upload skipped:
// column means of a strided row sample, taken on the device (k_col_mean), then the usual shift.
static int centre_deferred(svihmm_ctx* h) {
  if (!h->center_deferred) return 0;
  h->center_deferred = false;
  if (h->T <= 0 || !h->obs.p || h->shifted || h->variant[9] == 1) return 0;
  const int D = h->D;
  int64_t nsamp = ((int64_t)4 << 20) / D;
  if (nsamp > 65536) nsamp = 65536;
  if (nsamp < 16) nsamp = 16;
  if (nsamp > h->T) nsamp = h->T;
  const int64_t stride = h->T / nsamp;
  CK(ensure(h->scratch, (size_t)D * sizeof(double)));
  hipLaunchKernelGGL(k_col_mean, dim3((unsigned)D), dim3(256), 0, h->stream, (const double*)h->obs.p, D, stride,
                     nsamp, (double*)h->scratch.p);
  HIPCK(hipGetLastError());
  std::vector<double> c((size_t)D, 0.0);
  CK(d2h_sync_small(h, c.data(), h->scratch.p, (size_t)D * sizeof(double)));
  return reset_shift(h, c, 0, h->T);
}

int svihmm_set_emission_niw(svihmm_ctx* h, int32_t K, int32_t D, const double* mu,
                            const double* sigma, const double* kappa, const double* nu) {
  if (!h || K <= 0 || D <= 0 || !mu || !sigma || !kappa || !nu)
    return fail("svihmm_set_emission_niw: bad arguments");
  if (D > SVIHMM_NIW_MAX_D)   // (k_niw_to_theta_generic keeps two D x (D+1) matrices in LDS)
    return fail("svihmm_set_emission_niw: D > SVIHMM_NIW_MAX_D: evaluate the expected log-likelihoods on the "
                "host and pass them with svihmm_set_lliks / SVIHMM_USE_HOST_LLIKS");
  CK(set_device(h));
  h->lin_stale = true;
  if (h->D == D) CK(centre_deferred(h));
  // (a device loop of another family or shape kept its factors in h->niw: it ends here)
  if (h->svi_active && !(h->svi_family == 0 && h->svi_K == K && h->svi_D == D)) h->svi_active = false;
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  const size_t nin = nmu + nsg + 2 * (size_t)K;
  CK(ensure(h->niw, nin * sizeof(double) + 64));
  double* dmu = (double*)h->niw.p;
  // one pinned staging slot, pulled by a kernel, no stream synchronisation
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, (nin + 1) * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  std::memcpy(hp, mu, nmu * sizeof(double));
  to_centred(h, hp, K, D);             // the resident observations are x - shift: so are the means
  std::memcpy(hp + nmu, sigma, nsg * sizeof(double));
  std::memcpy(hp + nmu + nsg, kappa, K * sizeof(double));
  std::memcpy(hp + nmu + nsg + K, nu, K * sizeof(double));
  // (a live loop whose own family and shape are re-pushed: the previous iteration's ELBO kernels may still be
  //  deferred -- they read the factors this upload rewrites, so they go first)
  if (h->svi_active) CK(svi_flush_elbo(h));
  if (h->vlb_pending) { HIPCK(hipStreamWaitEvent(h->stream, h->svi_ed, 0)); h->vlb_pending = false; }
  CK(drop_auto_status(h));
  CK(pull_small(h, dmu, hp, nin * sizeof(double)));
  CK(pin_release(h, slot));
  CK(launch_niw_to_theta(h, K, D, nullptr));
  return 0;
}

int svihmm_set_emission_diag(svihmm_ctx* h, int32_t K, int32_t D, const double* mu, const double* nus,
                             const double* alphas, const double* betas) {
  if (!h || K <= 0 || D <= 0 || !mu || !nus || !alphas || !betas)
    return fail("svihmm_set_emission_diag: bad arguments");
  if (D > SVIHMM_DIAG_MAX_D)
    return fail("svihmm_set_emission_diag: D > SVIHMM_DIAG_MAX_D: evaluate the expected log-likelihoods on the "
                "host and pass them with svihmm_set_lliks / SVIHMM_USE_HOST_LLIKS");
  CK(set_device(h));
  h->lin_stale = true;
  if (h->D == D) CK(centre_deferred(h));
  const size_t n = (size_t)K * D;
  CK(ensure(h->niw, 4 * n * sizeof(double) + 64));
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, 4 * n * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  std::memcpy(hp, mu, n * sizeof(double));
  to_centred(h, hp, K, D);
  std::memcpy(hp + n, nus, n * sizeof(double));
  std::memcpy(hp + 2 * n, alphas, n * sizeof(double));
  std::memcpy(hp + 3 * n, betas, n * sizeof(double));
  if (h->svi_active) CK(svi_flush_elbo(h));      // (as in svihmm_set_emission_niw)
  if (h->vlb_pending) { HIPCK(hipStreamWaitEvent(h->stream, h->svi_ed, 0)); h->vlb_pending = false; }
  // a running device loop survives a re-push of ITS OWN family and shape (the loop's factor block IS
  // h->niw: the validation hooks of infer() -- full_predprob, adaptive L, growBuffer -- re-upload the
  // factors they just read back); any other loop's resident state is gone with this upload
  if (!(h->svi_active && h->svi_family == 1 && h->svi_K == K && h->svi_D == D)) h->svi_active = false;
  CK(drop_auto_status(h));
  CK(pull_small(h, h->niw.p, hp, 4 * n * sizeof(double)));
  CK(pin_release(h, slot));
  return launch_diag_to_theta(h, K, D);
}

// ---- ELBO terms of the NIW factors from the device-resident parameters --------------------
int svihmm_set_emission_prior(svihmm_ctx* h, int32_t K, int32_t D, const double* mu0, const double* sigma0) {
  if (!h || K <= 0 || D <= 0 || !mu0 || !sigma0) return fail("svihmm_set_emission_prior: bad arguments");
  CK(set_device(h));
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  CK(ensure(h->prior, (nmu + nsg) * sizeof(double)));
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, (nmu + nsg) * sizeof(double), &pin, &slot));
  std::memcpy(pin, mu0, nmu * sizeof(double));
  to_centred(h, (double*)pin, K, D);
  std::memcpy((double*)pin + nmu, sigma0, nsg * sizeof(double));
  CK(pull_small(h, h->prior.p, pin, (nmu + nsg) * sizeof(double)));
  CK(pin_release(h, slot));
  h->prior_K = K; h->prior_D = D;
  h->prior_mu0.assign(mu0, mu0 + nmu);     // re-centred when the handle's shift has moved since
  h->prior_epoch = h->shift_epoch;
  return 0;
}
int svihmm_niw_vlb_terms(svihmm_ctx* h, int32_t K, int32_t D, const double* mu, const double* sigma,
                         const double* kappa, const double* nu, double* out3K) {
  if (!h || K <= 0 || D <= 0 || !mu || !sigma || !kappa || !nu || !out3K)
    return fail("svihmm_niw_vlb_terms: bad arguments");
  if (h->prior_K != K || h->prior_D != D) return fail("svihmm_niw_vlb_terms: call svihmm_set_emission_prior first");
  if (D > SVIHMM_NIW_MAX_D) return fail("svihmm_niw_vlb_terms: D > SVIHMM_NIW_MAX_D");
  CK(set_device(h));
  CK(upload_feature_table(h, D, K));
  const int Fp = h->Fp, Kp = h->Kp;
  if (h->vlb_host_K < K) {
    if (h->vlb_host) hipHostFree(h->vlb_host);
    h->vlb_host = nullptr; h->vlb_host_K = 0;
    HIPCK(hipHostMalloc((void**)&h->vlb_host, (size_t)3 * K * sizeof(double), hipHostMallocMapped));
    h->vlb_host_K = K;
  }
  double* dout = nullptr;
  HIPCK(hipHostGetDevicePointer((void**)&dout, h->vlb_host, 0));
  // a private parameter / theta set: the E-step's own (h->niw, h->theta) stay untouched, so the
  // intermediates of the last E-step remain readable after the ELBO has been evaluated
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D, nin = nmu + nsg + 2 * (size_t)K;
  const size_t nth = (size_t)Fp * Kp;
  CK(ensure(h->vlb_aux, (nin + nth + K + 8) * sizeof(double)));
  double* dmu = (double*)h->vlb_aux.p;
  double* dsg = dmu + nmu;
  double* dka = dsg + nsg;
  double* dnu = dka + K;
  double* th2 = dnu + K;
  double* ld = th2 + nth;
  int* dstat = (int*)(ld + K);
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, nin * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  std::memcpy(hp, mu, nmu * sizeof(double));
  to_centred(h, hp, K, D);     // (only differences mu - mu_0 enter; both travel centred)
  std::memcpy(hp + nmu, sigma, nsg * sizeof(double));
  std::memcpy(hp + nmu + nsg, kappa, K * sizeof(double));
  std::memcpy(hp + nmu + nsg + K, nu, K * sizeof(double));
  CK(pull_small(h, dmu, hp, nin * sizeof(double)));
  CK(pin_release(h, slot));
  if (h->prior_epoch != h->shift_epoch) {
    void* pin2 = nullptr;
    int slot2 = 0;
    CK(pinned(h, nmu * sizeof(double), &pin2, &slot2));
    std::memcpy(pin2, h->prior_mu0.data(), nmu * sizeof(double));
    to_centred(h, (double*)pin2, K, D);
    CK(pull_small(h, h->prior.p, pin2, nmu * sizeof(double)));
    CK(pin_release(h, slot2));
    h->prior_epoch = h->shift_epoch;
  }
  const double* p0 = (const double*)h->prior.p;
  CK(launch_niw_vlb(h, K, D, dmu, dsg, dka, dnu, th2, dstat, ld, p0, dout));
  HIPCK(hipStreamSynchronize(h->stream));
  std::memcpy(out3K, h->vlb_host, (size_t)3 * K * sizeof(double));
  return 0;
}

// The resident column holds symbol indices: a centred copy (an upload cannot know the family that
// will read it; svihmm_shift_obs may have moved it) goes back to exact integers and stays uncentred.
// Called when the table is set and again in front of every lookup / count launch.
int cat_uncentre(svihmm_ctx* h) {
  if (h->T > 0) h->center_deferred = true;     // a Gaussian family that follows centres the copy again
  if (!h->shifted || h->T <= 0) return 0;
  std::vector<double> back(h->shift.size());
  for (size_t d = 0; d < back.size(); ++d) back[d] = -h->shift[d];
  CK(shift_rows(h, back.data(), 0, h->T, true));
  return store_shift(h, std::vector<double>((size_t)h->D, 0.0));
}
// feature table is also needed by the statistics kernels when only host lliks are used
int svihmm_set_emission_cat(svihmm_ctx* h, int32_t K, int32_t V, const double* logp) {
  if (!h || K <= 0 || V <= 0 || !logp) return fail("svihmm_set_emission_cat: bad arguments");
  CK(set_device(h));
  h->lin_stale = true;
  h->center_pending = false;
  // (a device loop of another family or shape ends with this upload, as in svihmm_set_emission_niw / _diag)
  if (h->svi_active && !(h->svi_family == 2 && h->svi_K == K && h->V == V)) h->svi_active = false;
  if (h->svi_active) CK(svi_flush_elbo(h));      // (as in svihmm_set_emission_niw)
  if (h->vlb_pending) { HIPCK(hipStreamWaitEvent(h->stream, h->svi_ed, 0)); h->vlb_pending = false; }
  CK(drop_auto_status(h));
  CK(cat_uncentre(h));
  const size_t n = (size_t)K * V;
  CK(ensure(h->cat_table, n * sizeof(double)));
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, n * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  for (int k = 0; k < K; ++k)              // transpose to [V][K]: a row's states are contiguous
    for (int v = 0; v < V; ++v) hp[(size_t)v * K + k] = logp[(size_t)k * V + v];
  HIPCK(hipMemcpyAsync(h->cat_table.p, hp, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  CK(pin_release(h, slot));
  h->eK = K; h->eD = 1; h->V = V; h->Kp = K > 64 ? (K + 63) / 64 * 64 : (K + 15) / 16 * 16; h->have_emission = true; h->emis_cat = true; h->emis_diag = false; h->uw_valid = false;
  return 0;
}
int64_t svihmm_packed_len(svihmm_ctx* h) { return h ? (int64_t)packed_len(h) : 0; }

int ensure_feature_table(svihmm_ctx* h) {
  return upload_feature_table(h, h->D, h->K, h->emis_diag);
}

int svihmm_set_lliks(svihmm_ctx* h, const double* lliks, int32_t B, int32_t Lm) {
  if (!h || !lliks || B <= 0 || Lm <= 0) return fail("svihmm_set_lliks: bad arguments");
  if (!h->have_globals) return fail("svihmm_set_lliks: call svihmm_set_globals first (K unknown)");
  CK(set_device(h));
  h->lin_stale = true;
  const size_t n = (size_t)B * Lm * h->K * sizeof(double);
  // (host-evaluated emitters do not use the factors an automatic theta rebuild may have complained about)
  CK(drop_auto_status(h));
  CK(ensure(h->ll, n));
  HIPCK(hipMemcpyAsync(h->ll.p, lliks, n, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->hostB = B; h->hostLm = Lm; h->have_host_ll = true;
  return 0;
}

// ---- launch helpers -----------------------------------------------------------------
static int check_windows(svihmm_ctx* h, const int64_t* starts, int B, int Lm, bool need_obs) {
  if (B <= 0 || Lm <= 0) return fail("bad window batch (B, Lm must be positive)");
  if (need_obs) {
    if (h->T <= 0) return fail("no observations: call svihmm_set_obs first");
    if (!starts) return fail("starts is NULL");
    for (int b = 0; b < B; ++b)
      if (starts[b] < 0 || starts[b] + Lm > h->T)
        return fail("window " + std::to_string(b) + " out of range");
  }
  return 0;
}

// small upload through a pinned slot, pulled by a kernel (see k_exp_transpose)
static int pull_small(svihmm_ctx* h, void* dst, const void* pin, size_t bytes) {
  const size_t n = bytes / 8;
  unsigned blocks = (unsigned)((n + 1023) / 1024);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  void* dpin = nullptr;
  HIPCK(hipHostGetDevicePointer(&dpin, const_cast<void*>(pin), 0));
  hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, h->stream, (const unsigned long long*)dpin,
                     (unsigned long long*)dst, n);
  HIPCK(hipGetLastError());
  return 0;
}

int ensure_starts_pulled(svihmm_ctx* h) {
  if (!h->starts_pending) return 0;
  void* dst = h->starts.p;
  const int64_t* src = h->starts_pending;
  h->starts_pending = nullptr;
  hipLaunchKernelGGL(k_pull, dim3(1), dim3(256), 0, h->stream, (const unsigned long long*)src,
                     (unsigned long long*)dst, (size_t)h->starts_pending_n);
  HIPCK(hipGetLastError());
  return 0;
}
static int upload_starts(svihmm_ctx* h, const int64_t* starts, int B) {
  CK(ensure(h->starts, (size_t)B * sizeof(int64_t)));
  h->starts_pending = nullptr;
  const size_t nb = (size_t)B * sizeof(int64_t);
  if (h->svi_upload_it >= 0 && nb <= (size_t)1 << 20) {
    svihmm_ctx::StartSlot& ss = h->svi_starts[h->svi_upload_it % 8];
    if (ss.used_it >= 0 && h->svi_flags && h->svi_ts && 2 * ss.used_it + 1 < h->svi_ts_cap) {
      // (eight iterations back: long complete -- its end stamp is there; bounded wait, then the stream itself)
      volatile unsigned long long* ts = &h->svi_ts[2 * ss.used_it + 1];
      int spins = 0;
      while (*ts == 0 && ++spins < 200000) { for (volatile int w = 0; w < 50; ++w) {} }
      if (*ts == 0) HIPCK(hipStreamSynchronize(h->stream));
    } else if (ss.used_it >= 0 && 2 * ss.used_it + 1 < (int)h->svi_ev.size())
      HIPCK(hipEventSynchronize(h->svi_ev[2 * ss.used_it + 1]));   // (eight iterations back: long complete)
    if (nb > ss.cap) {
      if (ss.p) hipHostFree(ss.p);
      ss.p = nullptr; ss.cap = 0;
      HIPCK(hipHostMalloc(&ss.p, nb + 4096, hipHostMallocMapped));
      ss.cap = nb + 4096;
    }
    std::memcpy(ss.p, starts, nb);
    // the pull is owed, not launched: the orbit-schedule emission kernel reads the slot itself and leaves the
    // device copy behind (launch_emission); every other consumer pays it first (ensure_starts_pulled)
    void* dpin = nullptr;
    HIPCK(hipHostGetDevicePointer(&dpin, ss.p, 0));
    h->starts_pending = (const int64_t*)dpin; h->starts_pending_n = B;
    ss.used_it = h->svi_upload_it;
  } else if (h->starts_sync_call && nb <= (size_t)1 << 20) {
    // (svihmm_estep_minibatch with a read-back: as in the SVI loop the pull is owed, not launched -- a k_pull launch
    //  and the slot's release event in front of the emission kernel are ~10 us of a 64-window step)
    if (h->starts_slot_inflight) { HIPCK(hipStreamSynchronize(h->stream)); h->starts_slot_inflight = false; }
    if (nb > h->starts_slot_cap) {
      if (h->starts_slot) hipHostFree(h->starts_slot);
      h->starts_slot = nullptr; h->starts_slot_cap = 0;
      HIPCK(hipHostMalloc(&h->starts_slot, nb + 4096, hipHostMallocMapped));
      h->starts_slot_cap = nb + 4096;
    }
    std::memcpy(h->starts_slot, starts, nb);
    void* dpin = nullptr;
    HIPCK(hipHostGetDevicePointer(&dpin, h->starts_slot, 0));
    h->starts_pending = (const int64_t*)dpin; h->starts_pending_n = B;
    h->starts_slot_inflight = true;
  } else if (nb <= (size_t)4 << 20) {   // through a pinned slot: no host-side wait for the stream
    void* pin = nullptr;
    int slot = 0;
    CK(pinned(h, nb, &pin, &slot));
    std::memcpy(pin, starts, nb);
    CK(pull_small(h, h->starts.p, pin, nb));
    CK(pin_release(h, slot));
  } else {
    HIPCK(hipMemcpyAsync(h->starts.p, starts, nb, hipMemcpyHostToDevice, h->stream));
  }
  return 0;
}



// Which sweep implementation a batch uses: 1 wave-per-window (log domain; small batches,
// K > 64), 2 log-domain MFMA (callers that want lalpha / lbeta back), 3 scaled
// linear-domain MFMA (the E-step fast path; logs are materialised on demand).
static int pick_fb(const svihmm_ctx* h, int B, int Lm, bool want_logs) {
  int var = h->variant[2];
  if (h->K > 256 || h->exact_log) return 1;
  if (h->K > 64) {   // no log-domain MFMA sweep beyond 64 states: scaled (streamed B) or per-window
    if (var == 2) var = 1;
    // one long chain: blocked scan (its messages convert to logs row by row, see materialise);
    // large batches: scaled sweeps unless the logs themselves are wanted
    if (var == 0) var = (use_chain(h, B, Lm) || (B >= cu_scaled(h, 192) && !want_logs)) ? 3 : 1;
    if (var == 3 && !use_chain(h, B, Lm) && (size_t)16 * Lm * h->K * sizeof(double) >= ((size_t)1 << 32)) var = 1;
    return var;
  }
  if (var == 0) var = want_logs ? (use_chain(h, B, Lm) ? 3 : B >= cu_scaled(h, 192) ? 2 : 1) : 3;   // no logs wanted: scaled sweeps at any batch size; logs of one long chain: blocked scan + conversion
  // the scaled sweeps address a workgroup's 16 windows with 32-bit byte offsets
  // (the wave-per-window kernel of small batches uses 64-bit row offsets)
  if (var == 3 && !use_chain(h, B, Lm) && !(B < lin_wave_max(h) && h->variant[7] != 2) &&
      (size_t)16 * Lm * h->K * sizeof(double) >= ((size_t)1 << 32)) var = 2;
  return var;
}

// messages + posterior for a window batch (mode from pick_fb, fixed before the emission ran)
// the SVI loop's globals kernel runs on a side stream: everything that reads Aexp / mod_init /
// var_init on the main stream waits for it here (no-op otherwise)
int wait_globals(svihmm_ctx* h) {
  if (h->globals_ev) {
    HIPCK(hipStreamWaitEvent(h->stream, h->globals_ev, 0));
    h->globals_ev = nullptr;
  }
  return 0;
}
// The SVI loop's side streams may still read what a host upload is about to rewrite: the ELBO
// kernels (stream3) read niw / theta / logdet / var_tran, the globals kernel (stream2) var_tran.
static int wait_side_streams(svihmm_ctx* h) {
  // (flags mode, inside the loop's E-step: the sweeps gate on the globals counter -- or take the event themselves,
  //  launch_fb_lin_range -- and the global step gates on the ELBO kernels' counter)
  if (h->svi_flags && h->in_svi_estep) return 0;
  CK(svi_flush_elbo(h));
  CK(wait_globals(h));
  if (h->vlb_pending) { HIPCK(hipStreamWaitEvent(h->stream, h->svi_ed, 0)); h->vlb_pending = false; }
  return 0;
}
static int run_fb(svihmm_ctx* h, int B, int Lm, int var, bool want_lb, bool total) {
  h->m_nb = 0;
  // the sweeps need the globals kernel of the SVI loop's side stream; the loop's other cross-stream
  // wait (the previous iteration's ELBO kernels, before the global step rewrites what they read) is
  // taken at the same place: every stream-order event between two kernels of the iteration's chain
  // costs a few microseconds of dispatch, two in a row less than two apart
  CK(wait_side_streams(h));
  if (var == 3) {
    h->have_lb = true;   // materialised lazily
    return launch_fb_lin(h, B, Lm, total);
  }
  h->have_lb = (var != 2) || want_lb;
  if (h->svi_flags && h->in_svi_estep) CK(wait_globals(h));    // (the log-domain kernels take no gate)
  if (var == 2) return launch_fb_fused(h, B, Lm, want_lb, total);
  CK(launch_fb(h, B, Lm, 0, 2));
  return launch_posterior(h, B, Lm, total);
}


int d2h(svihmm_ctx* h, void* dst, const void* src, size_t bytes) {
  ProfScope ps(h, KS_D2H);
  HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  return 0;
}
// small result readback: DMA into a pinned slot queued right behind the producing kernel,
// one stream synchronisation, host copy to the caller's (pageable) buffer.  A direct async
// copy to pageable memory makes the runtime wait for the stream on the host first (~0.1 ms
// of GPU idle per E-step in the kernel trace).
static int d2h_sync_small(svihmm_ctx* h, void* dst, const void* src, size_t bytes) {
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, bytes, &pin, &slot));
  {
    ProfScope ps(h, KS_D2H);
    HIPCK(hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCK(hipStreamSynchronize(h->stream));
  std::memcpy(dst, pin, bytes);
  return 0;
}

int64_t svihmm_packed_size(int32_t K, int32_t D) {
  return (int64_t)K * K + (int64_t)K * D + K + (int64_t)K * D * D + 1;
}


// lin: the batch goes through the scaled linear-domain sweeps (pick_fb == 3)
int prepare_ll(svihmm_ctx* h, const int64_t* starts, int B, int Lm, uint32_t flags,
                      bool need_obs_for_stats, bool lin) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  const bool host_ll = flags & SVIHMM_USE_HOST_LLIKS;
  CK(check_windows(h, starts, B, Lm, !host_ll || need_obs_for_stats));
  if (starts) CK(upload_starts(h, starts, B));
  h->cur_f32 = false;
  h->eh_float = false;
  if (host_ll) {
    if (!h->have_host_ll || h->hostB != B || h->hostLm != Lm)
      return fail("SVIHMM_USE_HOST_LLIKS: no uploaded lliks of shape [B,Lm,K]");
    if (lin) CK(launch_scale_ll(h, B, Lm));
  } else {
    // K <= 64: the emission kernel owns whole rows and writes (Eh, kexp) itself; wider
    // models take the plain kernel plus one scaling pass
    const bool two_pass = lin && (h->Kp > 64 || h->emis_cat);
    // fp32 mode: scaled messages stored as float + fp32 statistics GEMM, for what the mode
    // covers (NIW emission, K <= 64, window batches on the scaled sweeps; round 5: wide NIW models
    // up to K = 256, D = 64 in large batches, f32_wide_ok); everything else runs as fp64
    const bool wide32 = two_pass && lin && h->prec == 1 && h->f32_ok && !use_chain(h, B, Lm) &&
                        f32_wide_ok(h, (int64_t)B * Lm);
    h->cur_f32 = lin && h->prec == 1 && h->f32_ok && (!two_pass || wide32) && !use_chain(h, B, Lm);
    CK(launch_emission(h, B, Lm, flags, lin && !two_pass));
    if (h->cur_f32 && !two_pass && h->f32_fused_req) {     // (float Eh is written; from here on the batch is an fp64 one)
      h->eh_float = true;
      h->cur_f32 = false;
    }
    // wide models: plain log-likelihoods + one scaling pass (fp32 mode: float, in place in h->ll, first rows
    // of the windows in h->ll0 -- the layout of the K <= 64 path, so eh_in_llE stays false)
    if (wide32) CK(launch_scale_ll_f32(h, B, Lm));
    else if (two_pass) CK(launch_scale_ll(h, B, Lm));
    h->have_host_ll = false;
  }
  h->eh_in_llE = lin && (host_ll || (h->Kp > 64 && !h->cur_f32) || h->emis_cat);
  h->lin_mode = lin;
  h->lin_stale = false;
  h->q_valid = false;
  h->curB = B;
  h->last_host_ll = host_ll;
  h->last_flags = flags;
  h->m_nb = 0;
  return 0;
}

// device pointer of row `row0` of intermediate `what` (0 lliks, 1 lalpha, 2 lbeta, 3 var_x)
static int intermediate_ptr(svihmm_ctx* h, int what, int64_t row0, int64_t nrows, const double** out) {
  const int K = h->K, Lm = h->lastLm;
  Buf* src[] = {&h->ll, &h->la, &h->lb, &h->q};
  if (!h->lin_mode || what == 3 || (what == 0 && h->eh_in_llE)) {
    if (what == 3) CK(ensure_q(h, h->lastB, Lm, h->stream));
    if (what == 2 && !h->have_lb) {
      // the fused log-domain backward sweep kept lbeta in registers (no SVIHMM_KEEP_LBETA):
      // one more backward sweep over the lliks still held, as long as they are current
      if (h->lin_stale)
        return fail("lbeta was not kept (SVIHMM_KEEP_LBETA) and the observations / globals / emission "
                    "parameters have changed since: read it before the next parameter upload");
      CK(wait_globals(h));
      CK(launch_fb(h, h->lastB, Lm, 1, 1));
      h->have_lb = true;
    }
    if (!src[what]->p) return fail("intermediate buffer not available");
    *out = (const double*)src[what]->p + (size_t)row0 * K;
    return 0;
  }
  const int b0 = (int)(row0 / Lm), b1 = (int)((row0 + nrows - 1) / Lm) + 1;
  CK(materialise(h, b0, b1 - b0));
  Buf* ms[] = {&h->m_ll, &h->m_la, &h->m_lb};
  *out = (const double*)ms[what]->p + ((size_t)row0 - (size_t)h->m_b0 * Lm) * K;
  return 0;
}

// copy `packed` into the host-visible mirror on the handle's stream
static int launch_mirror(svihmm_ctx* h) {
  const size_t n = (size_t)packed_len(h);
  if (n * sizeof(double) > h->mirror_cap) {
    if (h->mirror) hipHostFree(h->mirror);
    h->mirror = nullptr; h->mirror_cap = 0;
    HIPCK(hipHostMalloc((void**)&h->mirror, n * sizeof(double) + 256, hipHostMallocMapped));
    h->mirror_cap = n * sizeof(double) + 256;
  }
  double* dev = nullptr;
  HIPCK(hipHostGetDevicePointer((void**)&dev, h->mirror, 0));
  ProfScope ps(h, KS_D2H);
  if (h->shifted && !h->emis_cat)      // statistics leave in the caller's coordinates
    hipLaunchKernelGGL(k_packed_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                       (const double*)h->packed.p, dev, (int)n, h->K, h->D, (const double*)h->shift_d.p, 1.0, h->emis_diag ? 1 : 0);
  else
    hipLaunchKernelGGL(k_mirror, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                       (const double*)h->packed.p, dev, (int)n);
  HIPCK(hipGetLastError());
  h->mirror_valid = true;
  return 0;
}
static int read_packed_host(svihmm_ctx* h, double* out) {
  const size_t nb = (size_t)packed_len(h) * sizeof(double);
  if (!h->mirror_valid) CK(launch_mirror(h));
  HIPCK(hipStreamSynchronize(h->stream));
  std::memcpy(out, h->mirror, nb);
  return 0;
}
// Sum of the packed statistics over the ranks.  The buffer holds centred coordinates and every
// rank has its own shift: the sum is formed in the callers' common coordinates and brought back.
// packed_to_common: the statistics in the coordinates every rank shares (the callers'), as a device
// buffer (commtmp; `packed` itself where no shift applies); packed_from_common: the reduced buffer
// back into this handle's centred coordinates.  The RCCL all-reduce and the host-mediated exchange
// (svihmm_export_packed / svihmm_import_packed) run between the same two halves.
static bool common_needs_shift(const svihmm_ctx* h, bool always) {
  return h->shifted && !h->emis_cat && (always || h->nranks > 1 || h->variant[11] == 1);   // (variant 11: rehearsal at one rank)
}
static int packed_to_common(svihmm_ctx* h, bool always, double** buf_out) {
  const size_t n = (size_t)packed_len(h);
  if (!common_needs_shift(h, always)) { *buf_out = (double*)h->packed.p; return 0; }
  CK(ensure(h->commtmp, n * sizeof(double)));
  double* tmp = (double*)h->commtmp.p;
  hipLaunchKernelGGL(k_packed_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                     (const double*)h->packed.p, tmp, (int)n, h->K, h->D, (const double*)h->shift_d.p, 1.0,
                     h->emis_diag ? 1 : 0);
  HIPCK(hipGetLastError());
  *buf_out = tmp;
  return 0;
}
static int packed_from_common(svihmm_ctx* h, const double* buf) {
  if (buf == (const double*)h->packed.p) return 0;
  const size_t n = (size_t)packed_len(h);
  hipLaunchKernelGGL(k_packed_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, buf,
                     (double*)h->packed.p, (int)n, h->K, h->D, (const double*)h->shift_d.p, -1.0,
                     h->emis_diag ? 1 : 0);
  HIPCK(hipGetLastError());
  return 0;
}
static int allreduce_packed_dev(svihmm_ctx* h) {
  double* buf = nullptr;
  CK(packed_to_common(h, false, &buf));
  NCCLCK(ncclAllReduce(buf, buf, (size_t)packed_len(h), ncclDouble, ncclSum, h->comm, h->stream));
  return packed_from_common(h, buf);
}

// ---- two-stream E-step pipeline -------------------------------------------------------------
// The scaled sweeps are latency-bound (one 4-wave workgroup per 16 windows, a serial chain
// of Lm steps that keeps the matrix pipe ~45 % busy) while emission and statistics are
// throughput-bound GEMMs.  For large batches the windows are split in two halves:
//   stream A:  emission(h1)  emission(h2)        statistics(h1)  statistics(h2)  finalize
//   stream B:               sweeps(h1)      sweeps(h2)
// so that the sweeps of one half share the CUs with a GEMM of the other half.  The emission
// kernel's LDS request is padded so that three of its workgroups plus one sweep workgroup fit
// a CU; the sweep kernels raise their wave priority.  Partials stay deterministic (fixed
// chunking per half, one finalize over all slots).
static bool use_pipeline(const svihmm_ctx* h, int B, int Lm, int fbvar, uint32_t flags) {
  const int v = h->variant[4];
  if (v == 1 || fbvar != 3 || (flags & SVIHMM_USE_HOST_LLIKS)) return false;
  if (h->Kp > 64 || h->D > 64 || h->emis_cat) return false;
  const int sv = h->variant[1];
  if (sv != 0 && sv != 3) return false;
  // opt-in only: on MI355X the sweeps' fp64 VALU work queues behind the GEMMs' MFMAs on a
  // shared SIMD and both sides lose more than the overlap wins (experiments)
  return v == 2 && B >= 32;
}
static int estep_pipelined(svihmm_ctx* h, const int64_t* starts, int B, int Lm, int inner_off,
                           int inner_len, uint32_t flags) {
  if (!h->have_globals) return fail("no globals: call svihmm_set_globals");
  CK(check_windows(h, starts, B, Lm, true));
  CK(upload_starts(h, starts, B));
  if (!h->have_emission) return fail("no emission parameters: call svihmm_set_emission_niw");
  if (h->eD != h->D) return fail("emission D does not match obs D");
  if (h->eK != h->K) return fail("emission K does not match globals K");
  if (!h->stream2) {
    HIPCK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIPCK(hipEventCreateWithFlags(&h->ev_em[i], hipEventDisableTiming));
      HIPCK(hipEventCreateWithFlags(&h->ev_sw[i], hipEventDisableTiming));
    }
  }
  const int K = h->K;
  const int64_t n = (int64_t)B * Lm;
  // all buffers first: ensure() may reallocate
  CK(ensure(h->ll, (size_t)n * K * sizeof(double)));
  CK(ensure(h->kexp, (size_t)n * sizeof(double)));
  CK(ensure(h->ll0, (size_t)B * K * sizeof(double)));
  CK(ensure_fb_lin(h, B, Lm));
  int b0[2], nb[2];
  b0[0] = 0; nb[0] = ((B / 2 + 15) / 16) * 16; b0[1] = nb[0]; nb[1] = B - nb[0];
  StatsPlan plan[2] = {stats_plan(h, (int64_t)nb[0] * inner_len), stats_plan(h, (int64_t)nb[1] * inner_len)};
  CK(ensure_stats(h, plan[0].nchunk + plan[1].nchunk));
  h->have_host_ll = false;
  h->lin_mode = true; h->lin_stale = false; h->last_host_ll = false; h->eh_in_llE = false; h->last_flags = flags;
  h->cur_f32 = false;
  h->q_valid = false; h->curB = B;
  h->m_nb = 0; h->have_lb = true;
  hipStream_t A = h->stream, Bs = h->stream2;
  // stream B must not start before earlier work of stream A (parameter uploads) is done
  HIPCK(hipEventRecord(h->ev_sw[1], A));
  HIPCK(hipStreamWaitEvent(Bs, h->ev_sw[1], 0));
  const size_t em_lds = 41 * 1024;   // 3 emission workgroups + 1 sweep workgroup per CU
  for (int c = 0; c < 2; ++c) {
    const size_t ro = (size_t)b0[c] * Lm;
    CK(launch_emission(h, nb[c], Lm, flags, true, (const int64_t*)h->starts.p + b0[c],
                       (double*)h->ll.p + ro * K, (double*)h->kexp.p + ro, A, em_lds,
                       (double*)h->ll0.p + (size_t)b0[c] * K));
    HIPCK(hipEventRecord(h->ev_em[c], A));
    HIPCK(hipStreamWaitEvent(Bs, h->ev_em[c], 0));
    CK(launch_fb_lin_range(h, b0[c], nb[c], Lm, Bs));
    HIPCK(hipEventRecord(h->ev_sw[c], Bs));
  }
  for (int c = 0; c < 2; ++c) {
    HIPCK(hipStreamWaitEvent(A, h->ev_sw[c], 0));
    CK(launch_stats_range(h, b0[c], nb[c], Lm, inner_off, inner_len, flags, plan[c],
                          c == 0 ? 0 : plan[0].nchunk, A));
  }
  CK(launch_stats_finalize(h, plan[0].nchunk + plan[1].nchunk, A));
  CK(launch_sum_lb(h, B, A));
  return 0;
}

int svihmm_loglik(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm, uint32_t flags,
                  double* out_lliks) {
  if (!h || !out_lliks) return fail("svihmm_loglik: bad arguments");
  CK(set_device(h));
  CK(prepare_ll(h, starts, B, Lm, flags & ~SVIHMM_USE_HOST_LLIKS, false));
  CK(d2h(h, out_lliks, h->ll.p, (size_t)B * Lm * h->K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  CK(check_emission_status(h));
  h->lastB = B; h->lastLm = Lm;
  return 0;
}

int svihmm_forward_backward(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                            uint32_t flags, double* out_lalpha, double* out_lbeta,
                            double* out_var_x, double* out_local_lb) {
  if (!h) return fail("svihmm_forward_backward: NULL handle");
  CK(set_device(h));
  const int var = pick_fb(h, B, Lm, out_lalpha != nullptr || out_lbeta != nullptr);
  CK(prepare_ll(h, starts, B, Lm, flags, false, var == 3));
  CK(run_fb(h, B, Lm, var, out_lbeta != nullptr, false));
  h->lastB = B; h->lastLm = Lm;
  if (var == 3 && (out_lalpha || out_lbeta)) CK(materialise(h, 0, B));   // forced variant
  if (var == 3 && out_var_x) CK(ensure_q(h, B, Lm, h->stream));
  const size_t n = (size_t)B * Lm * h->K * sizeof(double);
  if (out_lalpha) CK(d2h(h, out_lalpha, var == 3 ? h->m_la.p : h->la.p, n));
  if (out_lbeta) CK(d2h(h, out_lbeta, var == 3 ? h->m_lb.p : h->lb.p, n));
  if (out_var_x) CK(d2h(h, out_var_x, h->q.p, n));
  if (out_local_lb) CK(d2h(h, out_local_lb, h->local_lb.p, (size_t)B * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  CK(check_emission_status(h));
  h->lastB = B; h->lastLm = Lm;
  return 0;
}

int svihmm_pred_logprob(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                        uint32_t flags, double out2[2]) {
  if (!h || !starts || !out2) return fail("svihmm_pred_logprob: bad arguments");
  if (flags & SVIHMM_USE_HOST_LLIKS) return fail("svihmm_pred_logprob: NIW emission only");
  CK(set_device(h));
  if (!h->have_mask) { out2[0] = NAN; out2[1] = 0.0; return 0; }
  const int var = pick_fb(h, B, Lm, false);
  CK(prepare_ll(h, starts, B, Lm, flags, false, var == 3));
  CK(run_fb(h, B, Lm, var, false, false));
  h->lastB = B; h->lastLm = Lm;
  CK(ensure_q(h, B, Lm, h->stream));
  // emission term on the true observations (no masking), into the side buffer
  const int K = h->K;
  const int64_t n = (int64_t)B * Lm;
  CK(ensure(h->m_ll, (size_t)n * K * sizeof(double)));
  h->m_nb = 0;
  CK(launch_emission(h, B, Lm, flags & ~(uint32_t)SVIHMM_MASK_AS_NAN, false, nullptr, (double*)h->m_ll.p));
  const int nblk = (int)((n + PRED_ROWS_PER_BLOCK - 1) / PRED_ROWS_PER_BLOCK);
  CK(ensure(h->scratch, ((size_t)2 * nblk + 2) * sizeof(double)));
  double* part = (double*)h->scratch.p;
  {
    ProfScope ps(h, KS_MISC);
#define PL(KT) hipLaunchKernelGGL(k_pred_logprob<KT>, dim3(nblk), dim3(256), 0, h->stream, (const double*)h->q.p, \
                                  (const double*)h->m_ll.p, (const uint8_t*)h->mask.p,                            \
                                  (const int64_t*)h->starts.p, n, Lm, K, part)
    if (K <= 16) PL(1); else if (K <= 32) PL(2); else if (K <= 48) PL(3); else if (K <= 64) PL(4);
    else if (K <= 128) PL(8); else if (K <= 256) PL(16);
    else return fail("svihmm_pred_logprob: K > 256 not supported");
#undef PL
    hipLaunchKernelGGL(k_pred_final, dim3(1), dim3(64), 0, h->stream, (const double*)part, nblk, part + 2 * nblk);
    HIPCK(hipGetLastError());
  }
  CK(d2h(h, out2, part + 2 * nblk, 2 * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  CK(check_emission_status(h));
  return 0;
}

// emission -> sweeps -> statistics of a window batch into h->packed (B >= 1), asynchronous
static int estep_core(svihmm_ctx* h, const int64_t* starts, int B, int Lm, int inner_off, int inner_len,
                      uint32_t flags) {
  if (inner_off < 0 || inner_len <= 0 || inner_off + inner_len > Lm)
    return fail("svihmm_estep_minibatch_ex: inner segment out of range");
  const int var = pick_fb(h, B, Lm, false);
  if (use_pipeline(h, B, Lm, var, flags)) {
    CK(estep_pipelined(h, starts, B, Lm, inner_off, inner_len, flags));
  } else {
    // (the fused launch below can compute the emission tiles itself: launch_emission then does everything but launch)
    h->em_def.active = false;
    h->em_defer_req = var == 3 && sweep_emission_ok(h, B, Lm, inner_off, inner_len, flags);
    h->f32_fused_req = var == 3 && sweep_mixed_ok(h, B, Lm, inner_off, inner_len, flags);
    const int prc = prepare_ll(h, starts, B, Lm, flags, true, var == 3);
    h->em_defer_req = false;
    h->f32_fused_req = false;
    if (prc) { h->em_def.active = false; return prc; }
    const bool fused = var == 3 && sweep_stats_ok(h, B, Lm, inner_off, inner_len, flags);
    if (h->eh_float && !fused) return fail("internal: float emission rows without the fused launch that reads them");
    if (!fused) CK(launch_emission_deferred(h));
    if (fused) {
      // minibatch-sized batches of the five-tile shapes: sweeps and statistics in one launch, the statistics'
      // stages behind the sweeps' published progress (tu_fused.hip)
      CK(wait_side_streams(h));
      CK(launch_sweep_stats(h, B, Lm, inner_off, inner_len, flags));
    } else {
      CK(run_fb(h, B, Lm, var, (flags & SVIHMM_KEEP_LBETA) != 0, true));
      CK(launch_stats(h, B, Lm, inner_off, inner_len, flags));
    }
    CK(flush_lb(h, h->stream));   // no-op when k_finalize carried the ELBO total
  }
  h->have_packed = true;
  h->mirror_valid = false;
  h->lastB = B; h->lastLm = Lm;
  return 0;
}

int svihmm_estep_minibatch(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                           uint32_t flags, double* out_packed) {
  return svihmm_estep_minibatch_ex(h, starts, B, Lm, 0, Lm, flags, out_packed);
}

int svihmm_estep_minibatch_ex(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                              int32_t inner_off, int32_t inner_len, uint32_t flags,
                              double* out_packed) {
  if (!h) return fail("svihmm_estep_minibatch: NULL handle");
  CK(set_device(h));
  if (B == 0) {  // empty shard of a multi-GPU minibatch: all-zero statistics
    if (!h->have_globals || h->D <= 0) return fail("svihmm_estep_minibatch: set obs/globals first");
    const size_t nb = (size_t)packed_len(h) * sizeof(double);
    CK(ensure(h->packed, nb));
    HIPCK(hipMemsetAsync(h->packed.p, 0, nb, h->stream));
    h->have_packed = true;
    h->mirror_valid = false;
    if (out_packed) {
      CK(d2h(h, out_packed, h->packed.p, nb));
      HIPCK(hipStreamSynchronize(h->stream));
    }
    return 0;
  }
  h->starts_sync_call = out_packed != nullptr && h->variant[9] != 2;    // (variant 9 = 2: the k_pull route)
  const int rc = estep_core(h, starts, B, Lm, inner_off, inner_len, flags);
  h->starts_sync_call = false;
  if (rc) return rc;
  CK(launch_mirror(h));
  if (out_packed) {
    CK(read_packed_host(h, out_packed));
    h->starts_slot_inflight = false;      // (the stream is idle: the slot's readers are done)
    CK(check_emission_status(h));
  }
  return 0;
}

// ---- device-resident SVI loop (hmmsgd_metaobs.py:347-445) ------------------------------------
// layout of h->svi_state: var_tran K*K | prior_tran K*K | var_init K | vlb K | logdet K | prior_logpart K | rowterm K |
// var_init' K (the stationary vector is double-buffered: the next iteration's is computed ahead)
static double* svi_ptr(svihmm_ctx* h, int which) {
  const size_t K = h->svi_K, kk = K * K;
  double* b = (double*)h->svi_state.p;
  switch (which) {
    case 0: return b;                    // var_tran
    case 1: return b + kk;               // prior_tran
    case 2: return b + 2 * kk;           // var_init
    case 3: return b + 2 * kk + K;       // vlb
    case 4: return b + 2 * kk + 2 * K;   // logdet
    case 5: return b + 2 * kk + 3 * K;   // prior_logpart
    case 6: return b + 2 * kk + 4 * K;   // rowterm (Dirichlet terms of the transition rows)
    case 7: return b + 2 * kk + 5 * K;   // second var_init slot
    case 8: return b + 2 * kk + 6 * K;   // lb of the last two iterations (read by the side-stream ELBO kernel)
    default: return b + 2 * kk + 6 * K + 16;   // ada_G K*K (AdaGrad: accumulated squared natural parameters)
  }
}
#ifndef SVI_GW
#define SVI_GW 8      // wavefronts of the k_svi_globals workgroups
#endif
// counters of the loop's device-side dependencies (host.h): [0] global-step workgroups, [1] globals-kernel
// workgroups, [2] theta-builder workgroups, [3] ELBO side-chain workgroups; the gates' status word is slot 1 of the
// mapped status block (slot 0: the theta builders')
static unsigned* svi_cnt(svihmm_ctx* h, int which) { return (unsigned*)h->svi_sync.p + 16 * which; }
static int* svi_gate_status(svihmm_ctx* h) {
  int* d = nullptr;
  if (h->pin_status && hipHostGetDevicePointer((void**)&d, h->pin_status, 0) == hipSuccess) return d + 1;
  return nullptr;
}
static unsigned long long* svi_stamp_dev(svihmm_ctx* h, int idx) { return h->svi_ts_dev ? h->svi_ts_dev + idx : nullptr; }
// What every gate of the loop carries (flags mode): the status word in mapped host memory, the loop's dead flag
// (counter slot 7) and the bound of its wait.  Counter slots: [0] global-step workgroups, [1] globals-kernel
// workgroups, [2] theta-builder workgroups, [3] ELBO side chain, [4] sweep launches that signalled their start,
// [5] the poison word (SviSync::poison), [6] the concurrency probe's flag, [7] dead.
static SviSync svi_sy(svihmm_ctx* h) {
  SviSync sy = {};
  if (h->svi_flags) { sy.status = h->svi_status_dev; sy.dead = svi_cnt(h, 7); sy.ticks = h->svi_ticks; }
  return sy;
}
// Bound of a gate's wait.  A gate waits for work submitted earlier in host order, and the in-order side streams keep
// it from starting more than one iteration ahead of what it waits for: 64 measured iteration periods (at least 50 ms)
// once the device stamps of a finished iteration are there, 2 s before that, plus 20 us per row of the batch in
// flight (adaptive windows / buffers grow between two measurements); never more than SVI_SYNC_TICKS (60 s).
static unsigned long long svi_gate_ticks(const svihmm_ctx* h, int64_t rows) {
  const double khz = h->wall_clock_khz > 0.0 ? h->wall_clock_khz : 100000.0;
  double t = h->svi_period_ticks ? std::fmax(64.0 * (double)h->svi_period_ticks, 50.0 * khz) : 2000.0 * khz;
  t += 0.02 * (double)rows * khz;
  return t < (double)SVI_SYNC_TICKS ? (unsigned long long)t : SVI_SYNC_TICKS;
}
static int svi_launch_gate(svihmm_ctx* h, hipStream_t st, int which, unsigned long long tgt) {
  if (tgt == 0) return 0;
  SviSync sy = svi_sy(h);
  sy.gate = svi_cnt(h, which); sy.gate_tgt = (unsigned)tgt;
  hipLaunchKernelGGL(k_svi_gate, dim3(1), dim3(64), 0, st, sy);
  HIPCK(hipGetLastError());
  return 0;
}
// One-time probe per handle (see k_svi_probe_wait), for each of the loop's two side streams against the main one:
// the waiter goes first and the setter is launched only once the waiter reports that it is running, so the outcome
// does not depend on which queue a serialising tool serves first.  ~30 us per stream where kernels overlap, 2 ms
// where they do not.
static int svi_probe_concurrency(svihmm_ctx* h) {
  if (h->svi_concurrent >= 0) return 0;
  if (!h->stream2) HIPCK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  if (!h->stream3) HIPCK(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
  unsigned* pin = reinterpret_cast<unsigned*>(h->pin_status);      // words 2, 3: started / result (mapped)
  unsigned* dpin = nullptr;
  HIPCK(hipHostGetDevicePointer((void**)&dpin, pin, 0));
  unsigned* flag = svi_cnt(h, 6);
  volatile unsigned* vp = pin;
  int ok = 1;
  hipStream_t side[2] = {h->stream2, h->stream3};
  for (int i = 0; i < 2 && ok; ++i) {
    vp[2] = 0; vp[3] = 2;
    HIPCK(hipMemset(flag, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(k_svi_probe_wait, dim3(1), dim3(64), 0, side[i], (const unsigned*)flag, dpin + 2, dpin + 3,
                       200000ull);
    HIPCK(hipGetLastError());
    for (int spin = 0; spin < 2000000 && vp[2] == 0; ++spin) { for (volatile int w = 0; w < 20; ++w) {} }
    hipLaunchKernelGGL(k_svi_probe_set, dim3(1), dim3(64), 0, h->stream, flag);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(side[i]));
    HIPCK(hipStreamSynchronize(h->stream));
    ok = vp[3] == 1 ? 1 : 0;
  }
  h->svi_concurrent = ok;
  return 0;
}
static int svi_globals(svihmm_ctx* h, int slot, int for_it) {
  const int K = h->svi_K;
  double* vi_out = svi_ptr(h, slot ? 7 : 2);
  const size_t kk = (size_t)K * K * sizeof(double);
  CK(ensure(h->mod_init, K * sizeof(double)));
  CK(ensure(h->ltran, kk));
  const int reach = K <= 64 ? (K + 15) / 16 * 16 : K <= 128 ? 128 : K <= 256 ? 256 : K;
  const size_t slack = (size_t)(reach - K + 16) * K * sizeof(double);
  CK(ensure(h->Aexp, kk + slack));
  CK(ensure(h->AexpT, kk + slack));
  // The globals kernel depends on var_tran only, i.e. on the previous global step: it is launched
  // on a side stream right after that step and runs beside the rest of the iteration (theta, ELBO
  // kernels) and the next iteration's uploads and emission GEMM; the sweeps wait for it
  // (wait_globals).  ~100 us off the critical path of a 64-window iteration.
  if (!h->stream2) HIPCK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  if (!h->svi_ea) {
    HIPCK(hipEventCreateWithFlags(&h->svi_ea, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&h->svi_eb, hipEventDisableTiming));
  }
  hipStream_t s2 = h->stream2;
  // (a globals kernel that finds the loop dead does not run: the iteration its output is for is poisoned)
  SviSync gsy = svi_sy(h);
  if (h->svi_flags) {
    gsy.arrive = svi_cnt(h, 1); h->tgt_glob += (unsigned)(K + 1);
    gsy.poison = svi_cnt(h, 5); gsy.poison_val = SVI_POISON_BASE - (unsigned)(for_it < 0 ? 0 : for_it);
  }
  if (h->svi_flags && h->tgt_step > 0) {
    // no event on the main stream: the side stream waits for the global steps launched so far in a one-wave gate
    CK(svi_launch_gate(h, s2, 0, h->tgt_step));
  } else {         // (svi_begin: behind the uploads of the main stream)
    HIPCK(hipEventRecord(h->svi_ea, h->stream));
    HIPCK(hipStreamWaitEvent(s2, h->svi_ea, 0));
  }
  if (h->slack_a != h->Aexp.p || h->slack_t != h->AexpT.p || h->slack_k != K) {
    HIPCK(hipMemsetAsync((char*)h->Aexp.p + kk, 0, slack, s2));
    HIPCK(hipMemsetAsync((char*)h->AexpT.p + kk, 0, slack, s2));
    h->slack_a = h->Aexp.p; h->slack_t = h->AexpT.p; h->slack_k = K;
  }
  const size_t work = 2 * (size_t)K * (K | 1) * sizeof(double);
  const int use_lds = work + 9 * 1024 <= 150 * 1024;
  if (!use_lds) CK(ensure(h->svi_work, work));
  {
    ProfScope ps(h, KS_MISC, s2);
    if (use_lds) {
      if (work > 48 * 1024)
        hipFuncSetAttribute((const void*)k_svi_globals<true, SVI_GW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)work);
      hipLaunchKernelGGL((k_svi_globals<true, SVI_GW>), dim3(K + 1), dim3(64 * SVI_GW), work, s2, (const double*)svi_ptr(h, 0), K,
                         (double*)nullptr, (double*)h->ltran.p, (double*)h->Aexp.p, (double*)h->AexpT.p,
                         vi_out, (double*)h->mod_init.p, gsy);
    } else {
      hipLaunchKernelGGL((k_svi_globals<false, SVI_GW>), dim3(K + 1), dim3(64 * SVI_GW), 0, s2, (const double*)svi_ptr(h, 0), K,
                         (double*)h->svi_work.p, (double*)h->ltran.p, (double*)h->Aexp.p, (double*)h->AexpT.p,
                         vi_out, (double*)h->mod_init.p, gsy);
    }
    HIPCK(hipGetLastError());
  }
  HIPCK(hipEventRecord(h->svi_eb, s2));
  h->globals_ev = h->svi_eb;
  h->svi_globals_ready = true; h->svi_globals_slot = slot;
  h->K = K; h->have_globals = true; h->lin_stale = true;
  // (a host svihmm_set_globals between two iterations -- select_L, pred_logprob_full -- decides
  //  these per upload; the loop's own globals are back in the range svi_begin verified)
  h->exact_log = false; h->f32_ok = h->svi_f32_ok;
  return 0;
}
// theta + log det for the factors now in h->niw (main stream: the next emission GEMM needs theta);
// their ELBO term vlb[] and, with elbo_it >= 0, elbo_vec[elbo_it] on the side stream -- only the
// ELBO trace needs them, so they stay off the critical path of the iteration chain.
static int svi_launch_elbo(svihmm_ctx* h, int elbo_it, int lb_slot, bool behind_sweeps, hipEvent_t after_theta);
static int svi_refresh_emission(svihmm_ctx* h, int elbo_it, int lb_slot, hipEvent_t after_theta = nullptr,
                                bool defer_elbo = false, const SviStepArgs* step = nullptr) {
  const int K = h->svi_K, D = h->svi_D, fam = h->svi_family;
  if (!h->stream3) HIPCK(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
  if (!h->svi_ec) {
    HIPCK(hipEventCreateWithFlags(&h->svi_ec, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&h->svi_ed, hipEventDisableTiming));
  }
  // flags mode: the builder's K workgroups arrive on the theta counter (the ELBO kernels' gate) and workgroup 0
  // stamps the end of the iteration
  SviSync tsy = {};
  if (h->svi_flags) {
    tsy.arrive = svi_cnt(h, 2);
    h->tgt_theta += (unsigned)K;
    if (elbo_it >= 0) { tsy.stamp = svi_stamp_dev(h, 2 * elbo_it + 1); tsy.stamp_at = (unsigned)h->tgt_theta; }
  }
  h->theta_sy = tsy;
  if (fam == 0) CK(launch_niw_to_theta(h, K, D, svi_ptr(h, 4), step));
  else if (fam == 1) CK(launch_diag_to_theta(h, K, D));
  else {
    // E log theta[v][k] = psi(alpha_kv) - psi(sum_v alpha_kv) (what hmmbase._push_emission uploads)
    ProfScope ps(h, KS_MISC);
    hipLaunchKernelGGL(k_cat_table, dim3(K), dim3(64), 0, h->stream, (const double*)h->niw.p, K, h->V,
                       (double*)h->cat_table.p, tsy);
    HIPCK(hipGetLastError());
    h->eK = K; h->eD = 1; h->Kp = K > 64 ? (K + 63) / 64 * 64 : (K + 15) / 16 * 16;
    h->have_emission = true; h->emis_cat = true; h->emis_diag = false; h->uw_valid = false;
  }
  h->theta_sy = SviSync{};
  h->lin_stale = true;
#ifdef SVIHMM_MEASURE
  if (h->variant[0] == 9) return 0;    // measurement only: no ELBO kernels at all
#endif
  if (h->svi_flags && defer_elbo && elbo_it >= 0) {
    // launched by the next svihmm_svi_iteration behind its sweeps (svi_launch_elbo), or by whoever needs them first
    h->elbo_pending = true; h->elbo_pend_it = elbo_it; h->elbo_pend_slot = lb_slot;
    return 0;
  }
  return svi_launch_elbo(h, elbo_it, lb_slot, false, after_theta);
}
// The ELBO terms of the factors now in h->niw / theta (side stream).  `behind_sweeps`: additionally gated on the
// start of the sweep launch that carries counter 4.
static int svi_launch_elbo(svihmm_ctx* h, int elbo_it, int lb_slot, bool behind_sweeps, hipEvent_t after_theta) {
  const int K = h->svi_K, D = h->svi_D, fam = h->svi_family;
  hipStream_t s2 = h->stream3;
  // (the ELBO kernels follow their gate kernels in stream order: when a gate gave up they find the loop dead and
  //  leave the iteration's entry NaN -- svi_recover computes it again where the state still allows)
  SviSync vsy = svi_sy(h);
  if (h->svi_flags) {
    CK(svi_launch_gate(h, s2, 2, h->tgt_theta));
    if (behind_sweeps) CK(svi_launch_gate(h, s2, 4, h->tgt_early));
    vsy.arrive = svi_cnt(h, 3);
    h->tgt_side += (unsigned)(2 * K);       // (the ELBO total rides in the last-arriving workgroup: kernels_svi.h)
  } else {
    // (the iteration's end-of-iteration timing event doubles as the fork point of the ELBO kernels)
    hipEvent_t fork = after_theta ? after_theta : h->svi_ec;
    HIPCK(hipEventRecord(fork, h->stream));
    HIPCK(hipStreamWaitEvent(s2, fork, 0));
  }
  {
    ProfScope ps(h, KS_MISC, s2);
    double* delbo = nullptr;
    if (elbo_it >= 0) HIPCK(hipHostGetDevicePointer((void**)&delbo, h->svi_elbo, 0));
    const bool tail = h->svi_flags && elbo_it >= 0;
    const SviElboTail et = {tail ? delbo + elbo_it : (double*)nullptr, (const double*)svi_ptr(h, 8) + lb_slot,
                            h->svi_prior_const, (unsigned)h->tgt_side};
    if (fam == 0)
      hipLaunchKernelGGL(k_svi_vlb, dim3(2 * K), dim3(64), 0, s2, (const double*)h->theta.p,
                         (const int*)h->fab.p, h->F, D, h->Kp, (const double*)h->niw.p,
                         (const double*)svi_ptr(h, 4), (const double*)h->svi_prior.p,
                         (const double*)svi_ptr(h, 5), h->svi_zsign, K, svi_ptr(h, 3),
                         (const double*)svi_ptr(h, 1), (const double*)svi_ptr(h, 0), svi_ptr(h, 6), vsy, et);
    else
      hipLaunchKernelGGL(k_svi_vlb_simple, dim3(2 * K), dim3(64), 0, s2, fam, (const double*)h->niw.p,
                         (const double*)h->svi_prior.p, K, fam == 1 ? D : h->V, svi_ptr(h, 3),
                         (const double*)svi_ptr(h, 1), (const double*)svi_ptr(h, 0), svi_ptr(h, 6), vsy, et);
    if (elbo_it >= 0 && !tail) {      // (stream-event choreography: no arrival counter to tell the last workgroup by)
      hipLaunchKernelGGL(k_svi_elbo, dim3(1), dim3(64), 0, s2, K, (const double*)svi_ptr(h, 3),
                         (const double*)svi_ptr(h, 6), h->svi_prior_const, (const double*)svi_ptr(h, 8) + lb_slot,
                         delbo + elbo_it, vsy);
    }
    HIPCK(hipGetLastError());
  }
  HIPCK(hipEventRecord(h->svi_ed, s2));
  h->vlb_pending = true;
  return 0;
}
// deferred ELBO kernels, now and ungated: somebody is about to wait for the side stream or to rewrite what they read
int svi_flush_elbo(svihmm_ctx* h) {
  if (!h->elbo_pending) return 0;
  h->elbo_pending = false;
  return svi_launch_elbo(h, h->elbo_pend_it, h->elbo_pend_slot, false, nullptr);
}

// What the three families' begin calls share: range check of the transition factor, the resident
// state (var_tran | prior_tran | ...), the ELBO / event rings.  The caller uploads its prior / factor
// blocks between svi_begin_common and svi_begin_finish.
static int svi_begin_common(svihmm_ctx* h, int K, int D, const double* prior_tran, const double* var_tran,
                            int maxit, int family) {
  if (K > 1024) return fail("svihmm_svi_begin: K > 1024 unsupported");
  if (h->D != D) return fail("svihmm_svi_begin: D does not match the resident observations");
  // from here on the previous loop's state is being overwritten: a begin that fails half way must not
  // leave it marked as live (svi_begin_finish sets the flag again once everything is in place)
  h->svi_active = false;
  h->svi_adagrad = false;
  CK(set_device(h));
  CK(wait_side_streams(h));
  CK(drop_auto_status(h));
  if (family != 2) CK(centre_deferred(h));
  const size_t kk = (size_t)K * K;
  {
    // The global step is var_tran <- (1 - rho) var_tran + rho (1 + bA (A_raw + nwin (prior_tran - 1)))
    // with bA nwin ~ T / 2L >> 1 (quirk Q2): only for prior_tran >= 1 is every later entry bounded
    // below by min(var_tran, 1), so that psi(v) - psi(row sum) stays inside the range of the
    // linear-domain recursions for the whole loop (the row sums are < 1e12: psi < 28).  Sparser
    // priors can drive entries towards zero or below mid-loop; the host loop, which picks the
    // recursion per upload, serves those.
    double vmin = INFINITY, pmin = INFINITY;
    for (size_t i = 0; i < kk; ++i) { vmin = std::fmin(vmin, var_tran[i]); pmin = std::fmin(pmin, prior_tran[i]); }
    if (!(vmin >= SVIHMM_SVI_MIN_PSEUDOCOUNT) || !(pmin >= 1.0))
      return fail("svihmm_svi_begin: prior_tran below 1 or var_tran below SVIHMM_SVI_MIN_PSEUDOCOUNT may need the "
                  "log-domain recursion mid-loop: run the loop through svihmm_set_globals + svihmm_estep_minibatch");
    h->svi_f32_ok = vmin > 0.05;      // psi(0.05) - 28 > SVIHMM_LTRAN_F32_MIN
    h->svi_vmin = vmin;
    h->exact_log = false;
    h->f32_ok = h->svi_f32_ok;
  }
  h->svi_K = K; h->svi_D = D; h->svi_maxit = maxit; h->svi_family = family;
  {   // sum_i [lgamma(sum_j p_ij + eps) - sum_j lgamma(p_ij + eps)]: the prior-only part of the rows' energy
    double pc = 0.0;
    for (int i = 0; i < K; ++i) {
      double rsum = 0.0, lg = 0.0;
      for (int j = 0; j < K; ++j) { const double p = prior_tran[(size_t)i * K + j]; rsum += p; lg += std::lgamma(p + 1e-9); }
      pc += std::lgamma(rsum + 1e-9) - lg;
    }
    h->svi_prior_const = pc;
  }
  CK(ensure(h->svi_state, (3 * kk + 6 * (size_t)K + 16) * sizeof(double)));
  h->svi_adagrad = false;
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, 2 * kk * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  std::memcpy(hp, var_tran, kk * 8); std::memcpy(hp + kk, prior_tran, kk * 8);
  CK(pull_small(h, svi_ptr(h, 0), hp, 2 * kk * 8));
  CK(pin_release(h, slot));
  if (h->svi_elbo_cap < maxit) {
    if (h->svi_elbo) hipHostFree(h->svi_elbo);
    h->svi_elbo = nullptr; h->svi_elbo_cap = 0;
    HIPCK(hipHostMalloc((void**)&h->svi_elbo, (size_t)maxit * sizeof(double) + 64, hipHostMallocMapped));
    h->svi_elbo_cap = maxit;
  }
  for (int i = 0; i < maxit; ++i) h->svi_elbo[i] = NAN;
  // device-side dependencies (variant[0] = 1: the stream-event choreography of rounds 2-4 instead)
  h->svi_flags = h->variant[0] != 1;
  h->tgt_step = h->tgt_glob = h->tgt_theta = h->tgt_side = h->tgt_early = 0;
  h->elbo_pending = false;
  h->svi_period_ticks = 0; h->svi_cur_it = -1; h->svi_replaying = false; h->svi_recoveries = 0;
  for (auto& e : h->svi_log) e.it = -1;
  h->svi_it_events.assign(maxit, (char)0);
  if (h->svi_flags) {
    // (begin is cold: the counters are zeroed with every stream idle, so no gate of this loop can see a
    //  previous loop's counts)
    CK(ensure(h->svi_sync, 8 * 64));
    HIPCK(hipStreamSynchronize(h->stream));
    HIPCK(hipMemset(h->svi_sync.p, 0, 8 * 64));
    if (h->svi_ts_cap < 2 * maxit) {
      if (h->svi_ts) hipHostFree(h->svi_ts);
      h->svi_ts = nullptr; h->svi_ts_dev = nullptr; h->svi_ts_cap = 0;
      HIPCK(hipHostMalloc((void**)&h->svi_ts, (size_t)2 * maxit * sizeof(unsigned long long) + 64, hipHostMallocMapped));
      h->svi_ts_cap = 2 * maxit;
      HIPCK(hipHostGetDevicePointer((void**)&h->svi_ts_dev, h->svi_ts, 0));
    }
    std::memset(h->svi_ts, 0, (size_t)2 * maxit * sizeof(unsigned long long));
    if (!h->pin_status) {
      HIPCK(hipHostMalloc((void**)&h->pin_status, 64, hipHostMallocMapped));
      std::memset(h->pin_status, 0, 64);
    }
    h->pin_status[1] = 0;
    h->svi_status_dev = svi_gate_status(h);
    // a device that runs one kernel at a time (a counter-collecting profiler, serialised launches) cannot carry
    // spinning gates: stream events there
    CK(svi_probe_concurrency(h));
    if (h->svi_concurrent != 1) h->svi_flags = false;
    if (h->wall_clock_khz <= 0.0) {
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
      h->wall_clock_khz = (double)khz;
    }
    h->svi_ticks = svi_gate_ticks(h, 0);
  }
  // (event choreography only: the counter loop times its iterations with device stamps)
  while (!h->svi_flags && (int)h->svi_ev.size() < 2 * maxit) {      // [2 it]: first launch of iteration it, [2 it + 1]: its last
    hipEvent_t e;
    HIPCK(hipEventCreate(&e));
    h->svi_ev.push_back(e);
  }
  h->svi_ev_begin.assign(maxit, 0);
  h->svi_last_it = -1;
  return 0;
}
static int svi_begin_finish(svihmm_ctx* h) {
  HIPCK(hipStreamSynchronize(h->stream));          // (the rings below are keyed to this loop's events)
  for (auto& ss : h->svi_starts) ss.used_it = -1;
  CK(svi_refresh_emission(h, -1, 0));   // theta / table of the initial factors (their vlb is not used)
  h->svi_vi_cur = 1;
  CK(svi_globals(h, 0, 0));          // globals of iteration 0
  h->svi_active = true;
  return 0;
}
// upload `n` doubles through a pinned slot into dst (asynchronous, pulled by a kernel)
static int svi_upload(svihmm_ctx* h, void* dst, const double* a, size_t na, const double* b, size_t nb2,
                      int centre_K, int centre_D) {
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, (na + nb2) * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  std::memcpy(hp, a, na * 8);
  if (b) std::memcpy(hp + na, b, nb2 * 8);
  if (centre_K > 0) to_centred(h, hp, centre_K, centre_D);     // leading [K][D] block = means
  CK(pull_small(h, dst, hp, (na + nb2) * 8));
  return pin_release(h, slot);
}

int svihmm_svi_begin(svihmm_ctx* h, int32_t K, int32_t D, const double* prior_tran, const double* var_tran,
                     const double* mu0, const double* sigma0, const double* kappa0, const double* nu0,
                     const double* prior_logpart, const double* mu, const double* sigma,
                     const double* kappa, const double* nu, int32_t maxit, double zsign) {
  if (!h || K <= 0 || D <= 0 || !prior_tran || !var_tran || !mu0 || !sigma0 || !kappa0 || !nu0 ||
      !prior_logpart || !mu || !sigma || !kappa || !nu || maxit <= 0)
    return fail("svihmm_svi_begin: bad arguments");
  CK(svi_begin_common(h, K, D, prior_tran, var_tran, maxit, 0));
  h->svi_zsign = zsign;
  const size_t nmu = (size_t)K * D, nsg = (size_t)K * D * D;
  const size_t nin = nmu + nsg + 2 * (size_t)K;
  CK(ensure(h->svi_prior, (nin + 8) * sizeof(double)));
  CK(ensure(h->niw, nin * sizeof(double) + 64));
  // one staging slot: [prior_logpart | prior block | niw block]
  const size_t tot = K + 2 * nin;
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, tot * sizeof(double), &pin, &slot));
  double* hp = (double*)pin;
  std::memcpy(hp, prior_logpart, K * 8);
  double* pp = hp + K;
  std::memcpy(pp, mu0, nmu * 8); std::memcpy(pp + nmu, sigma0, nsg * 8);
  to_centred(h, pp, K, D);             // the loop's state lives in the resident copy's coordinates
  std::memcpy(pp + nmu + nsg, kappa0, K * 8); std::memcpy(pp + nmu + nsg + K, nu0, K * 8);
  double* np_ = pp + nin;
  std::memcpy(np_, mu, nmu * 8); std::memcpy(np_ + nmu, sigma, nsg * 8);
  to_centred(h, np_, K, D);
  std::memcpy(np_ + nmu + nsg, kappa, K * 8); std::memcpy(np_ + nmu + nsg + K, nu, K * 8);
  CK(pull_small(h, svi_ptr(h, 5), hp, K * 8));
  CK(pull_small(h, h->svi_prior.p, pp, nin * 8));
  CK(pull_small(h, h->niw.p, np_, nin * 8));
  CK(pin_release(h, slot));
  return svi_begin_finish(h);
}

// Diagonal family (distributions.DiagonalGaussian: per dimension a normal-inverse-gamma factor):
// blocks [mu | nus | alphas | betas], each [K][D]; the same loop with the family's natural-parameter
// blend (hmmsgd_metaobs.py:1050-1069 in [nu mu, nu, 2 beta + nu mu^2, 2 alpha]), theta from
// k_diag_to_theta, ELBO term -KL(q || prior) per dimension.
int svihmm_svi_begin_diag(svihmm_ctx* h, int32_t K, int32_t D, const double* prior_tran, const double* var_tran,
                          const double* prior_blk, const double* factor_blk, int32_t maxit) {
  if (!h || K <= 0 || D <= 0 || !prior_tran || !var_tran || !prior_blk || !factor_blk || maxit <= 0)
    return fail("svihmm_svi_begin_diag: bad arguments");
  if (D > SVIHMM_DIAG_MAX_D) return fail("svihmm_svi_begin_diag: D > SVIHMM_DIAG_MAX_D");
  const size_t n4 = 4 * (size_t)K * D;
  for (size_t i = (size_t)K * D; i < n4; ++i)       // (validated before any state is touched)
    if (!(prior_blk[i] > 0.0) || !(factor_blk[i] > 0.0)) return fail("svihmm_svi_begin_diag: nus, alphas, betas must be positive");
  CK(svi_begin_common(h, K, D, prior_tran, var_tran, maxit, 1));
  CK(ensure(h->svi_prior, (n4 + 8) * sizeof(double)));
  CK(ensure(h->niw, n4 * sizeof(double) + 64));
  CK(svi_upload(h, h->svi_prior.p, prior_blk, n4, nullptr, 0, K, D));
  CK(svi_upload(h, h->niw.p, factor_blk, n4, nullptr, 0, K, D));
  return svi_begin_finish(h);
}

// Categorical family (hmmsgd_metaobs.py:907-926, 1071-1084): Dirichlet factors alpha[K][V] over one
// integer-valued observation column; the table E log theta is rebuilt on the device every iteration.
int svihmm_svi_begin_cat(svihmm_ctx* h, int32_t K, int32_t V, const double* prior_tran, const double* var_tran,
                         const double* alpha0, const double* alpha, int32_t maxit) {
  if (!h || K <= 0 || V <= 0 || !prior_tran || !var_tran || !alpha0 || !alpha || maxit <= 0)
    return fail("svihmm_svi_begin_cat: bad arguments");
  if (h->D != 1) return fail("svihmm_svi_begin_cat: the resident observations must be one symbol column (D = 1)");
  const size_t n = (size_t)K * V;
  for (size_t i = 0; i < n; ++i)
    if (!(alpha0[i] > 0.0) || !(alpha[i] > 0.0)) return fail("svihmm_svi_begin_cat: Dirichlet parameters must be positive");
  CK(svi_begin_common(h, K, 1, prior_tran, var_tran, maxit, 2));
  h->V = V;
  CK(cat_uncentre(h));
  CK(ensure(h->svi_prior, (n + 8) * sizeof(double)));
  CK(ensure(h->niw, n * sizeof(double) + 64));
  CK(ensure(h->cat_table, n * sizeof(double)));
  CK(svi_upload(h, h->svi_prior.p, alpha0, n, nullptr, 0, 0, 0));
  CK(svi_upload(h, h->niw.p, alpha, n, nullptr, 0, 0, 0));
  return svi_begin_finish(h);
}

static int svi_iteration_impl(svihmm_ctx* h, int32_t it, const int64_t* starts, int32_t B, int32_t nwin_total,
                              int32_t Lm, int32_t inner_off, int32_t inner_len, uint32_t flags, double rho,
                              double bfactA, double bfactE);
// Leave the counters in the middle of a loop and carry on with the stream-event choreography (round 6).  Two
// callers: a gate of the loop gave up (the status word is set: a tool that serialises kernels attached after
// svihmm_svi_begin's probe, another process time-slicing the device, a partition too small for the loop's
// kernels side by side), or the host decides to (`clean`: debug variant 0 = 2).  When a gate gave up the loop is
// dead on the device: every later gate returned at once, the E-steps whose inputs were missing poisoned their
// iteration and the global steps from that iteration on did not run (device_helpers.h) -- the loop's state is the
// one after iteration p - 1, whole.  The iterations from p on are replayed from the host's log (the host runs at
// most eight iterations ahead: the window-start ring), the last applied iteration's ELBO entry is formed again
// if its kernels were caught by the shutdown.  The event loop computes bit for bit what the counter loop does
// (tests/test_gpu_classes.py::test_svi_loop_on_counters_equals_the_stream_event_loop).
static int svi_recover(svihmm_ctx* h, bool clean) {
  if (!h->svi_flags) return 0;
  if (clean) CK(svi_flush_elbo(h));
  HIPCK(hipStreamSynchronize(h->stream));
  if (h->stream2) HIPCK(hipStreamSynchronize(h->stream2));
  if (h->stream3) HIPCK(hipStreamSynchronize(h->stream3));
  unsigned poison = 0;
  HIPCK(hipMemcpy(&poison, svi_cnt(h, 5), sizeof(unsigned), hipMemcpyDeviceToHost));
  const int submitted = h->svi_last_it;                     // last iteration handed to the device
  const int p = poison ? (int)(SVI_POISON_BASE - poison) : submitted + 1;   // first iteration that did not reach the state
  HIPCK(hipMemset(h->svi_sync.p, 0, 8 * 64));
  h->pin_status[1] = 0;
  h->tgt_step = h->tgt_glob = h->tgt_theta = h->tgt_side = h->tgt_early = 0;
  h->elbo_pending = false; h->vlb_pending = false; h->globals_ev = nullptr;
  h->svi_flags = false;
  ++h->svi_recoveries;
  while ((int)h->svi_ev.size() < 2 * h->svi_maxit) {
    hipEvent_t e;
    HIPCK(hipEventCreate(&e));
    h->svi_ev.push_back(e);
  }
  if (p > submitted + 1 || p < 0) return fail("SVI loop: inconsistent poison word after a gate gave up");
  // the ELBO entry of the last iteration that did reach the state: its kernels may have found the loop dead
  if (p >= 1 && std::isnan(h->svi_elbo[p - 1])) CK(svi_launch_elbo(h, p - 1, (p - 1) & 1, false, nullptr));
  if (p > submitted) return 0;                              // nothing was lost: the next iteration's globals are in place
  // replay iterations p .. submitted
  for (int j = p; j <= submitted; ++j)
    if (h->svi_log[j % 16].it != j) return fail("SVI loop: iteration " + std::to_string(j) + " is no longer in the replay log");
  h->svi_globals_ready = false;                             // (the globals on the device stem from a skipped kernel)
  h->svi_last_it = p - 1;
  h->svi_vmin = h->svi_log[p % 16].vmin_before;
  h->svi_f32_ok = h->svi_log[p % 16].f32_ok_before;
  h->svi_replaying = true;
  int rc = 0;
  for (int j = p; j <= submitted && rc == 0; ++j) {
    const SviLogEntry& e = h->svi_log[j % 16];
    rc = svi_iteration_impl(h, e.it, e.starts.data(), e.B, e.nwin, e.Lm, e.off, e.len, e.flags, e.rho, e.bA, e.bE);
  }
  h->svi_replaying = false;
  return rc;
}
int svihmm_svi_recoveries(svihmm_ctx* h, int32_t* out) {
  if (!h || !out) return fail("svihmm_svi_recoveries: bad arguments");
  *out = h->svi_recoveries;
  return 0;
}

int svihmm_svi_iteration(svihmm_ctx* h, int32_t it, const int64_t* starts, int32_t B, int32_t nwin_total,
                         int32_t Lm, int32_t inner_off, int32_t inner_len, uint32_t flags, double rho,
                         double bfactA, double bfactE) {
  if (!h || !h->svi_active) return fail("svihmm_svi_iteration: call svihmm_svi_begin first");
  if (it < 0 || it >= h->svi_maxit) return fail("svihmm_svi_iteration: iteration index out of range");
  if (B < 0 || (B > 0 && !starts) || nwin_total < B) return fail("svihmm_svi_iteration: bad window batch");
  if (flags & SVIHMM_USE_HOST_LLIKS) return fail("svihmm_svi_iteration: device-side emission families only");
  CK(set_device(h));
  if (h->svi_flags) {
    // a gate of the loop gave up since the last call (mapped status word: a plain host read)
    if (h->pin_status && *(volatile int*)&h->pin_status[1] != 0) CK(svi_recover(h, false));
    else if (h->variant[0] == 2 && it == 3) CK(svi_recover(h, true));     // (debug: "concurrency lost" between two iterations)
  }
  if (h->svi_flags) {
    // the iteration period, from the device stamps of an iteration that is certainly complete (eight back)
    const int j = it - 8;
    if (j >= 0 && (int)h->svi_ev_begin.size() > j && h->svi_ts) {
      const unsigned long long t0 = h->svi_ts[h->svi_ev_begin[j]], t1 = h->svi_ts[2 * j + 1];
      if (t0 && t1 > t0 && t1 - t0 > h->svi_period_ticks) h->svi_period_ticks = t1 - t0;
    }
    h->svi_ticks = svi_gate_ticks(h, (int64_t)B * Lm);
  }
  {
    SviLogEntry& e = h->svi_log[it % 16];
    e.it = it; e.B = B; e.nwin = nwin_total; e.Lm = Lm; e.off = inner_off; e.len = inner_len; e.flags = flags;
    e.rho = rho; e.bA = bfactA; e.bE = bfactE; e.vmin_before = h->svi_vmin; e.f32_ok_before = h->svi_f32_ok;
    e.starts.assign(starts, starts + (B > 0 ? B : 0));
  }
  return svi_iteration_impl(h, it, starts, B, nwin_total, Lm, inner_off, inner_len, flags, rho, bfactA, bfactE);
}
static int svi_iteration_impl(svihmm_ctx* h, int32_t it, const int64_t* starts, int32_t B, int32_t nwin_total,
                              int32_t Lm, int32_t inner_off, int32_t inner_len, uint32_t flags, double rho,
                              double bfactA, double bfactE) {
  const int K = h->svi_K, D = h->svi_D;
  h->svi_cur_it = it;
  if ((int)h->svi_it_events.size() < h->svi_maxit) h->svi_it_events.resize(h->svi_maxit, (char)0);
  h->svi_it_events[it] = h->svi_flags ? (char)0 : (char)1;
  // start of the iteration: when the previous iteration is still running, the stream would execute
  // this marker right behind that iteration's end marker -- the end marker serves as both
  if ((int)h->svi_ev_begin.size() < h->svi_maxit) h->svi_ev_begin.resize(h->svi_maxit, 0);
  h->svi_ev_begin[it] = 2 * it;
  if (h->svi_flags) {
    // (stamps instead of events: the previous iteration's end stamp -- written by its theta builder into mapped
    //  host memory -- is still 0 while that iteration runs)
    if (it > 0 && h->svi_last_it == it - 1 && *(volatile unsigned long long*)&h->svi_ts[2 * it - 1] == 0)
      h->svi_ev_begin[it] = 2 * it - 1;
    else {
      hipLaunchKernelGGL(k_svi_stamp, dim3(1), dim3(64), 0, h->stream, svi_stamp_dev(h, 2 * it));
      HIPCK(hipGetLastError());
    }
  } else if (it > 0 && h->svi_last_it == it - 1 && hipEventQuery(h->svi_ev[2 * it - 1]) == hipErrorNotReady)
    h->svi_ev_begin[it] = 2 * it - 1;
  else
    HIPCK(hipEventRecord(h->svi_ev[2 * it], h->stream));
  (void)hipGetLastError();            // (hipErrorNotReady is not an error here)
  h->svi_last_it = it;
  if (!h->svi_globals_ready) CK(svi_globals(h, h->svi_vi_cur ^ 1, it));   // (a host set_globals came in between)
  h->svi_vi_cur = h->svi_globals_slot;
  h->svi_globals_ready = false;       // consumed by this iteration's sweeps
  const bool keep = (flags & SVIHMM_SVI_KEEP_WINDOW) != 0;
  flags &= ~(uint32_t)SVIHMM_SVI_KEEP_WINDOW;
  if (B > 0) {
    h->svi_upload_it = it;
    h->in_svi_estep = true;
    h->sweep_signalled = false;
    const int rc = estep_core(h, starts, B, Lm, inner_off, inner_len, flags);
    h->in_svi_estep = false;
    h->svi_upload_it = -1;
    // the previous iteration's ELBO kernels: behind this iteration's sweeps when their launch signals its start
    if (rc == 0 && h->elbo_pending) {
      h->elbo_pending = false;
      CK(svi_launch_elbo(h, h->elbo_pend_it, h->elbo_pend_slot, h->sweep_signalled, nullptr));
    }
    if (rc) return rc;
    // the last window's log-domain rows are rebuilt on demand from the CURRENT parameters:
    // do it now, before the global step replaces them
    if (keep && h->lin_mode) CK(materialise(h, B - 1, 1));
  } else {   // empty shard of a multi-GPU minibatch: all-zero statistics
    const size_t nb = (size_t)packed_len(h) * sizeof(double);
    CK(ensure(h->packed, nb));
    HIPCK(hipMemsetAsync(h->packed.p, 0, nb, h->stream));
    h->have_packed = true; h->mirror_valid = false;
  }
  if (h->comm) {
    ProfScope ps(h, KS_ALLREDUCE);
    CK(allreduce_packed_dev(h));
  }
  const bool flushed_now = h->elbo_pending;
  CK(svi_flush_elbo(h));   // (an empty shard, or an E-step that failed over to an ungated path)
  CK(wait_globals(h));     // (an empty shard ran no sweeps; a gated sweep kernel has cleared the event)
  // the previous iteration's ELBO kernels (their own stream) read var_tran / theta / logdet,
  // which this global step and the NIW kernel after it rewrite (normally long finished): flags mode gates the
  // step kernel on their counter, the event mode waits for their event
  // (NIW factors of 17 .. 32 dimensions: the step rides in the theta builder's launch -- k_svi_step_theta32s)
  const bool merged = h->svi_family == 0 && step_theta_ok(h, K, D);
  const unsigned ntran = merged ? (unsigned)((K * K + 63) / 64) : (unsigned)((K * K + 255) / 256);
  const unsigned step_grid = h->svi_family == 0 ? (unsigned)K + ntran
                                                : (unsigned)(((size_t)K * (h->svi_family == 1 ? D : h->V) + 255) / 256) + ntran;
  SviSync ssy = svi_sy(h);
  if (h->svi_flags) {
    ssy.gate = svi_cnt(h, 3); ssy.gate_tgt = (unsigned)h->tgt_side; ssy.arrive = svi_cnt(h, 0);
    ssy.poison = svi_cnt(h, 5);
    if (h->tgt_side == 0) ssy.gate = nullptr;
    // A gated grid must leave room for what it waits for.  The step's workgroups all spin in their gate: a grid
    // that can hold half the device's wave slots (K beyond ~600), or ELBO kernels that were enqueued only just now
    // (svi_flush_elbo above: they may not be resident yet), could keep those kernels off the device -- the stream
    // waits for their event instead (recorded in both modes, svi_launch_elbo).
    const bool big = (size_t)step_grid * (merged ? 1 : 4) > (size_t)h->ncu * h->waves_per_cu / 2;
    if (ssy.gate && (big || flushed_now) && h->vlb_pending) {
      HIPCK(hipStreamWaitEvent(h->stream, h->svi_ed, 0));
      ssy.gate = nullptr;
    }
  } else if (h->vlb_pending) { HIPCK(hipStreamWaitEvent(h->stream, h->svi_ed, 0)); h->vlb_pending = false; }
  double* adag = h->svi_adagrad ? svi_ptr(h, 9) : (double*)nullptr;
  SviStepArgs sargs = {(const double*)h->packed.p, (const double*)svi_ptr(h, 1), svi_ptr(h, 0), (const double*)h->svi_prior.p,
                       rho, bfactA, bfactE, (double)nwin_total, svi_ptr(h, 8) + (it & 1), adag, ssy};
  if (merged) {
    if (h->svi_flags) h->tgt_step += ntran;          // (the merged kernel's transition workgroups: see k_svi_step_theta32s)
  } else {
    ProfScope ps(h, KS_MISC);
    if (h->svi_family == 0)
      hipLaunchKernelGGL(k_svi_global_step, dim3((unsigned)K + ntran), dim3(256), (size_t)3 * D * sizeof(double), h->stream,
                         (const double*)h->packed.p, (const double*)svi_ptr(h, 1), svi_ptr(h, 0),
                         (double*)h->niw.p, (const double*)h->svi_prior.p, K, D, rho, bfactA, bfactE,
                         (double)nwin_total, svi_ptr(h, 8) + (it & 1), adag, ssy);
    else {
      const int W = h->svi_family == 1 ? D : h->V;
      const unsigned nem = (unsigned)(((size_t)K * W + 255) / 256);
      hipLaunchKernelGGL(k_svi_global_step_simple, dim3(nem + ntran), dim3(256), 0, h->stream, h->svi_family,
                         (const double*)h->packed.p, (const double*)svi_ptr(h, 1), svi_ptr(h, 0),
                         (double*)h->niw.p, (const double*)h->svi_prior.p, K, W, rho, bfactA, bfactE,
                         (double)nwin_total, svi_ptr(h, 8) + (it & 1), adag, (int)nem, ssy);
    }
    HIPCK(hipGetLastError());
    if (h->svi_flags) h->tgt_step += step_grid;
  }
  // the next iteration's globals, ahead of time, forked right behind the global step (its own event:
  // forked behind theta together with the ELBO kernels it competes with the next emission GEMM,
  // ends after that GEMM, and a stream wait that really has to block wakes up ~20 us late -- measured
  // 0.272 against 0.25 ms per iteration)
  // Lower bound of var_tran after this step, kept on the host: the plain step is
  // v <- (1 - rho) v + rho (1 + bA (A_raw + nwin (prior_tran - 1))) with prior_tran >= 1 and A_raw >= 0,
  // so every entry is >= (1 - rho) bound + rho.  A loop that started below the fp32 mode's range for
  // E[log A] (a class's default initial var_tran of 1/K) enters it after its first steps (rho_0 = 1 for
  // tau = 1) instead of running fp64 to the end.  AdaGrad's element-wise weights give no such bound.
  if (!h->svi_adagrad && rho >= 0.0 && rho <= 1.0) {
    h->svi_vmin = (1.0 - rho) * h->svi_vmin + rho;
    h->svi_f32_ok = h->svi_vmin > 0.05;
  }
  if (merged) {
    // step + theta in one launch FIRST: the globals kernel's gate must wait for work submitted earlier in host order
    CK(svi_refresh_emission(h, it, it & 1, h->svi_flags ? (hipEvent_t) nullptr : h->svi_ev[2 * it + 1], it + 1 < h->svi_maxit, &sargs));
    if (it + 1 < h->svi_maxit) CK(svi_globals(h, h->svi_vi_cur ^ 1, it + 1));
    return 0;
  }
  if (it + 1 < h->svi_maxit) CK(svi_globals(h, h->svi_vi_cur ^ 1, it + 1));
  CK(svi_refresh_emission(h, it, it & 1, h->svi_flags ? (hipEvent_t) nullptr : h->svi_ev[2 * it + 1], it + 1 < h->svi_maxit));
  return 0;
}

// AdaGrad (hmmsgd_metaobs.py:179-183, 1036-1040): the K x K matrix of accumulated squared natural
// parameters of the transition factor joins the resident state; NULL switches the branch off.
int svihmm_svi_set_adagrad(svihmm_ctx* h, const double* ada_G) {
  if (!h || !h->svi_active) return fail("svihmm_svi_set_adagrad: call svihmm_svi_begin first");
  CK(set_device(h));
  if (!ada_G) { h->svi_adagrad = false; return 0; }
  const size_t kk = (size_t)h->svi_K * h->svi_K;
  // ada_G >= 1: the reference starts it at ones and only adds squares (hmmsgd_metaobs.py:179-183,
  // :1036-1040), and the step's weights 1 / ada_G^.25 must stay <= 1 -- a larger weight would let
  // var_tran leave the range svi_begin_common checked for the linear-domain recursions
  for (size_t i = 0; i < kk; ++i)
    if (!(ada_G[i] >= 1.0 && ada_G[i] < 1.7e308)) return fail("svihmm_svi_set_adagrad: ada_G must be finite and >= 1");
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, kk * sizeof(double), &pin, &slot));
  std::memcpy(pin, ada_G, kk * sizeof(double));
  CK(pull_small(h, svi_ptr(h, 9), pin, kk * sizeof(double)));
  CK(pin_release(h, slot));
  h->svi_adagrad = true;               // (only once the accumulator is on its way)
  return 0;
}
int svihmm_svi_read_adagrad(svihmm_ctx* h, double* ada_G_out) {
  if (!h || !h->svi_active || !ada_G_out) return fail("svihmm_svi_read_adagrad: bad arguments");
  if (!h->svi_adagrad) return fail("svihmm_svi_read_adagrad: the loop runs without AdaGrad");
  CK(set_device(h));
  return d2h_sync_small(h, ada_G_out, svi_ptr(h, 9), (size_t)h->svi_K * h->svi_K * sizeof(double));
}

// Everything the loop has in flight is complete and has reached the loop's state: pending ELBO kernels launched,
// all three streams idle and -- when a gate of the loop gave up on the way -- the lost iterations replayed on stream
// events (svi_recover).  Every read of the loop's results starts here.
static int svi_settle(svihmm_ctx* h) {
  CK(svi_flush_elbo(h));
  for (int pass = 0; pass < 2; ++pass) {
    HIPCK(hipStreamSynchronize(h->stream));
    if (h->stream2) HIPCK(hipStreamSynchronize(h->stream2));
    if (h->stream3) HIPCK(hipStreamSynchronize(h->stream3));
    if (!(h->svi_flags && h->pin_status && h->pin_status[1] != 0)) break;
    CK(svi_recover(h, false));
  }
  return 0;
}
int svihmm_svi_read_elbo(svihmm_ctx* h, int32_t n, double* out_elbo, double* out_ms) {
  if (!h || !h->svi_active || n < 0 || n > h->svi_maxit) return fail("svihmm_svi_read_elbo: bad arguments");
  CK(set_device(h));
  CK(svi_settle(h));
  CK(check_emission_status(h));
  for (int i = 0; i < n; ++i) {
    if (out_elbo) out_elbo[i] = h->svi_elbo[i];
    // (per iteration: a loop that left the counters mid-way timed its later iterations with events)
    if (out_ms && !((int)h->svi_it_events.size() > i && h->svi_it_events[i])) {
      // device wall-clock stamps: begin (k_svi_stamp, or the previous iteration's end) to the theta builder's end
      const int bi = (int)h->svi_ev_begin.size() > i ? h->svi_ev_begin[i] : 2 * i;
      const unsigned long long t0 = h->svi_ts[bi], t1 = h->svi_ts[2 * i + 1];
      out_ms[i] = (t0 && t1 && t1 >= t0) ? (double)(t1 - t0) / h->wall_clock_khz : NAN;
    } else if (out_ms) {
      float ms = 0.f;
      out_ms[i] = hipEventElapsedTime(&ms, h->svi_ev[(int)h->svi_ev_begin.size() > i ? h->svi_ev_begin[i] : 2 * i], h->svi_ev[2 * i + 1]) == hipSuccess ? (double)ms : NAN;
    }
  }
  return 0;
}

int svihmm_svi_read_state(svihmm_ctx* h, double* var_tran, double* var_init, double* mu, double* sigma,
                          double* kappa, double* nu) {
  if (!h || !h->svi_active) return fail("svihmm_svi_read_state: no SVI state on the device");
  if (h->svi_family != 0) return fail("svihmm_svi_read_state: NIW layout; this loop's family reads through svihmm_svi_read_factors");
  CK(set_device(h));
  CK(svi_settle(h));
  const size_t K = h->svi_K, D = h->svi_D, nmu = K * D, nsg = K * D * D;
  const double* nw = (const double*)h->niw.p;
  if (var_tran) CK(d2h(h, var_tran, svi_ptr(h, 0), K * K * 8));
  CK(wait_globals(h));
  if (var_init) CK(d2h(h, var_init, svi_ptr(h, h->svi_vi_cur ? 7 : 2), K * 8));
  if (mu) CK(d2h(h, mu, nw, nmu * 8));
  if (sigma) CK(d2h(h, sigma, nw + nmu, nsg * 8));
  if (kappa) CK(d2h(h, kappa, nw + nmu + nsg, K * 8));
  if (nu) CK(d2h(h, nu, nw + nmu + nsg + K, K * 8));
  HIPCK(hipStreamSynchronize(h->stream));
  if (h->stream3) HIPCK(hipStreamSynchronize(h->stream3));   // ELBO kernels still reading niw / theta
  h->vlb_pending = false;
  if (mu) from_centred(h, mu, (int)K, (int)D);
  CK(check_emission_status(h));
  return 0;
}

// The loop's state in the family's own block layout: NIW [mu | sigma | kappa | nu], diagonal
// [mu | nus | alphas | betas] (each [K][D]), Categorical alpha[K][V]; means in the caller's coordinates.
int svihmm_svi_read_factors(svihmm_ctx* h, double* var_tran, double* var_init, double* factors_out) {
  if (!h || !h->svi_active) return fail("svihmm_svi_read_factors: no SVI state on the device");
  CK(set_device(h));
  CK(svi_settle(h));
  const size_t K = h->svi_K, D = h->svi_D;
  const size_t n = h->svi_family == 0 ? K * D + K * D * D + 2 * K : h->svi_family == 1 ? 4 * K * D : K * (size_t)h->V;
  if (var_tran) CK(d2h(h, var_tran, svi_ptr(h, 0), K * K * 8));
  CK(wait_globals(h));
  if (var_init) CK(d2h(h, var_init, svi_ptr(h, h->svi_vi_cur ? 7 : 2), K * 8));
  if (factors_out) CK(d2h(h, factors_out, h->niw.p, n * 8));
  HIPCK(hipStreamSynchronize(h->stream));
  if (h->stream3) HIPCK(hipStreamSynchronize(h->stream3));   // ELBO kernels still reading the factors
  h->vlb_pending = false;
  if (factors_out && h->svi_family != 2) from_centred(h, factors_out, (int)K, (int)D);
  CK(check_emission_status(h));
  return 0;
}

// The globals the recursions currently hold (log domain): what svihmm_set_globals uploaded last, or
// what the device-resident loop's last iteration computed from var_tran (the reference leaves that
// iteration's psi-expectations on the object: hmmsgd_metaobs.py:502-504).
int svihmm_read_globals(svihmm_ctx* h, double* mod_init_out, double* ltran_out) {
  if (!h || (!mod_init_out && !ltran_out)) return fail("svihmm_read_globals: bad arguments");
  if (!h->have_globals) return fail("svihmm_read_globals: no globals on the device");
  CK(set_device(h));
  CK(wait_globals(h));
  const size_t K = h->K;
  if (mod_init_out) CK(d2h(h, mod_init_out, h->mod_init.p, K * 8));
  if (ltran_out) CK(d2h(h, ltran_out, h->ltran.p, K * K * 8));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_read_packed(svihmm_ctx* h, double* out_packed) {
  if (!h || !out_packed) return fail("svihmm_read_packed: bad arguments");
  if (!h->have_packed) return fail("svihmm_read_packed: no statistics computed yet");
  CK(set_device(h));
  CK(read_packed_host(h, out_packed));
  CK(check_emission_status(h));
  return 0;
}

int svihmm_read_intermediate(svihmm_ctx* h, int32_t what, double* out) {
  if (!h || !out) return fail("svihmm_read_intermediate: bad arguments");
  if (h->lastB <= 0) return fail("svihmm_read_intermediate: nothing computed yet");
  CK(set_device(h));
  if (what < 0 || what > 3) return fail("svihmm_read_intermediate: bad selector");
  const int64_t rows = (int64_t)h->lastB * h->lastLm;
  const double* src = nullptr;
  CK(intermediate_ptr(h, what, 0, rows, &src));
  CK(d2h(h, out, src, (size_t)rows * h->K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_read_rows(svihmm_ctx* h, int32_t what, int64_t row0, int64_t nrows, double* out) {
  if (!h || !out || row0 < 0 || nrows <= 0) return fail("svihmm_read_rows: bad arguments");
  if (h->lastB <= 0) return fail("svihmm_read_rows: nothing computed yet");
  if (what < 0 || what > 3) return fail("svihmm_read_rows: bad selector");
  if (row0 + nrows > (int64_t)h->lastB * h->lastLm) return fail("svihmm_read_rows: out of range");
  CK(set_device(h));
  const double* src = nullptr;
  CK(intermediate_ptr(h, what, row0, nrows, &src));
  CK(d2h(h, out, src, (size_t)nrows * h->K * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

int svihmm_state_argmax(svihmm_ctx* h, const int32_t* true_sts, int32_t* out_z,
                        int64_t* out_conf) {
  if (!h || (!out_z && !out_conf)) return fail("svihmm_state_argmax: bad arguments");
  if (out_conf && !true_sts) return fail("svihmm_state_argmax: the count matrix needs true_sts");
  if (h->lastB <= 0) return fail("svihmm_state_argmax: nothing computed yet");
  CK(set_device(h));
  const int K = h->K;
  const int64_t n = (int64_t)h->lastB * h->lastLm;
  const double* q = nullptr;
  CK(intermediate_ptr(h, 3, 0, n, &q));
  const size_t zb = ((size_t)n * sizeof(int32_t) + 15) & ~(size_t)15;
  const size_t cb = (size_t)K * K * sizeof(unsigned long long);
  CK(ensure(h->scratch, 2 * zb + cb));
  int32_t* dz = (int32_t*)h->scratch.p;
  int32_t* dtrue = (int32_t*)((char*)h->scratch.p + zb);
  unsigned long long* dconf = (unsigned long long*)((char*)h->scratch.p + 2 * zb);
  if (true_sts) {
    HIPCK(hipMemcpyAsync(dtrue, true_sts, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipMemsetAsync(dconf, 0, cb, h->stream));
  }
  {
    ProfScope ps(h, KS_MISC);
    const int nblk = (int)((n + ARGMAX_ROWS_PER_BLOCK - 1) / ARGMAX_ROWS_PER_BLOCK);
    const size_t lds = (true_sts && K <= 64) ? (size_t)K * K * sizeof(unsigned int) : 0;
    hipLaunchKernelGGL(k_state_argmax, dim3(nblk), dim3(256), lds, h->stream, q, n, K,
                       (const int32_t*)(true_sts ? dtrue : nullptr), dz, dconf);
    HIPCK(hipGetLastError());
  }
  if (out_z) CK(d2h(h, out_z, dz, (size_t)n * sizeof(int32_t)));
  if (out_conf) CK(d2h(h, out_conf, dconf, cb));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}


// ---- synthetic sequences generated in HBM (gen_synthetic.py:27-44) --------------------------
int svihmm_generate(svihmm_ctx* h, int64_t T, int32_t K, int32_t D, const double* cdf,
                    const double* means, const double* chols, uint64_t seed) {
  if (!h || T <= 0 || K <= 0 || D <= 0 || !cdf || !means || !chols) return fail("svihmm_generate: bad arguments");
  if (K > 64) return fail("svihmm_generate: K > 64 not supported");
  if (D > 4095) return fail("svihmm_generate: D too large");
  CK(set_device(h));
  h->lin_stale = true;
  CK(ensure(h->obs, (size_t)T * D * sizeof(double)));
  CK(ensure(h->gen_z, (size_t)T * sizeof(int32_t)));
  h->have_mask = false;
  const int KS = K <= 16 ? 16 : K <= 32 ? 32 : 64;
  const int Ls = T >= 65536 ? 512 : 256;
  const int Cs = (int)((T + Ls - 1) / Ls);
  const size_t npar = (size_t)K * K + (size_t)K * D + (size_t)K * D * D;
  const size_t bytes = npar * sizeof(double) + (size_t)T * KS + 2 * (size_t)Cs * KS + (size_t)Cs + 64;
  CK(ensure(h->scratch, bytes));
  double* dcdf = (double*)h->scratch.p;
  double* dmean = dcdf + (size_t)K * K;
  double* dchol = dmean + (size_t)K * D;
  unsigned char* path = (unsigned char*)(dchol + (size_t)K * D * D);
  unsigned char* mA = path + (size_t)T * KS;
  unsigned char* mB = mA + (size_t)Cs * KS;
  unsigned char* entry = mB + (size_t)Cs * KS;
  HIPCK(hipMemcpyAsync(dcdf, cdf, (size_t)K * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(dmean, means, (size_t)K * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(dchol, chols, (size_t)K * D * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, KS_MISC);
#define GPATH(KM)                                                                                        \
  hipLaunchKernelGGL(k_gen_paths<KM>, dim3(Cs), dim3(64), (size_t)K * (KM + 1) * sizeof(double), h->stream, \
                     (const double*)dcdf, (unsigned long long)seed, T, K, Ls, path)
    if (KS == 16) GPATH(16); else if (KS == 32) GPATH(32); else GPATH(64);
#undef GPATH
    hipLaunchKernelGGL(k_gen_compose, dim3(1), dim3(1024), 0, h->stream, (const unsigned char*)path, T, KS, Ls,
                       Cs, mA, mB, entry);
    hipLaunchKernelGGL(k_gen_gather, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, h->stream,
                       (const unsigned char*)path, (const unsigned char*)entry, T, KS, Ls, (int32_t*)h->gen_z.p);
    hipLaunchKernelGGL(k_gen_obs, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, h->stream,
                       (const int32_t*)h->gen_z.p, (const double*)dmean, (const double*)dchol,
                       (unsigned long long)seed, T, D, (double*)h->obs.p);
    HIPCK(hipGetLastError());
  }
  h->T = T; h->D = D; h->gen_T = T;
  h->svi_active = false;
  h->center_pending = false;
  {   // centre: the plain average of the state means (a point inside the data)
    std::vector<double> c((size_t)D, 0.0);
    h->center_deferred = h->have_emission && h->emis_cat;
    if (h->variant[9] != 1 && !(h->have_emission && h->emis_cat)) {
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) c[d] += means[(size_t)k * D + d] / K;
      for (int d = 0; d < D; ++d) if (!(c[d] > -1.7e308 && c[d] < 1.7e308)) c[d] = 0.0;
    }
    CK(reset_shift(h, c, 0, T));
  }
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}
int svihmm_read_generated(svihmm_ctx* h, int32_t* sts_out, double* obs_out) {
  if (!h || (!sts_out && !obs_out)) return fail("svihmm_read_generated: bad arguments");
  if (h->gen_T <= 0 || h->gen_T != h->T) return fail("svihmm_read_generated: no generated sequence is resident");
  CK(set_device(h));
  if (sts_out) CK(d2h(h, sts_out, h->gen_z.p, (size_t)h->T * sizeof(int32_t)));
  if (obs_out) CK(d2h(h, obs_out, h->obs.p, (size_t)h->T * h->D * sizeof(double)));
  HIPCK(hipStreamSynchronize(h->stream));
  if (obs_out && h->shifted) {          // the resident copy is centred: hand out caller coordinates
    const size_t D = (size_t)h->D;
    for (int64_t t = 0; t < h->T; ++t)
      for (size_t d = 0; d < D; ++d) obs_out[(size_t)t * D + d] += h->shift[d];
  }
  return 0;
}

// ---- multi-GPU -------------------------------------------------------------------------
int svihmm_comm_unique_id(char uid_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLCK(ncclGetUniqueId(&id));
  std::memcpy(uid_out, &id, 128);
  return 0;
}

int svihmm_comm_init(svihmm_ctx* h, const char uid[128], int32_t rank, int32_t nranks) {
  if (!h || !uid || nranks <= 0 || rank < 0 || rank >= nranks)
    return fail("svihmm_comm_init: bad arguments");
  CK(set_device(h));
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  ncclUniqueId id;
  std::memcpy(&id, uid, 128);
  NCCLCK(ncclCommInitRank(&h->comm, nranks, id, rank));
  h->rank = rank; h->nranks = nranks;
  return 0;
}

int svihmm_comm_count(svihmm_ctx* h, int32_t* nranks_out) {
  if (!h || !nranks_out) return fail("svihmm_comm_count: bad arguments");
  if (!h->comm) { *nranks_out = 0; return 0; }
  int n = 0;
  NCCLCK(ncclCommCount(h->comm, &n));
  *nranks_out = n;
  return 0;
}

int svihmm_comm_destroy(svihmm_ctx* h) {
  if (!h) return 0;
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  h->nranks = 1; h->rank = 0;
  return 0;
}

int svihmm_allreduce_packed(svihmm_ctx* h) {
  if (!h) return fail("svihmm_allreduce_packed: NULL handle");
  if (!h->have_packed) return fail("svihmm_allreduce_packed: no statistics computed yet");
  if (!h->comm) return fail("svihmm_allreduce_packed: communicator not initialised");
  CK(set_device(h));
  ProfScope ps(h, KS_ALLREDUCE);
  CK(allreduce_packed_dev(h));
  h->mirror_valid = false;
  return launch_mirror(h);   // the host-visible copy follows the reduced statistics
}

// Host-mediated exchange of the same statistics (a communicator other than RCCL -- MPI, gloo -- or
// two handles on one device): export hands out what the all-reduce would put on the wire (caller
// coordinates), import takes the reduced vector back into the handle's centred `packed`, from where
// svihmm_read_packed / the device-side global step continue as after svihmm_allreduce_packed.
int svihmm_export_packed(svihmm_ctx* h, double* out_packed) {
  if (!h || !out_packed) return fail("svihmm_export_packed: bad arguments");
  if (!h->have_packed) return fail("svihmm_export_packed: no statistics computed yet");
  CK(set_device(h));
  double* buf = nullptr;
  CK(packed_to_common(h, true, &buf));
  return d2h_sync_small(h, out_packed, buf, (size_t)packed_len(h) * sizeof(double));
}
int svihmm_import_packed(svihmm_ctx* h, const double* packed_in) {
  if (!h || !packed_in) return fail("svihmm_import_packed: bad arguments");
  if (!h->have_globals || h->D <= 0) return fail("svihmm_import_packed: set obs / globals / emission first");
  CK(set_device(h));
  const size_t nb = (size_t)packed_len(h) * sizeof(double);
  CK(ensure(h->packed, nb));
  const bool sh = common_needs_shift(h, true);
  if (sh) CK(ensure(h->commtmp, nb));
  double* dst = sh ? (double*)h->commtmp.p : (double*)h->packed.p;
  void* pin = nullptr;
  int slot = 0;
  CK(pinned(h, nb, &pin, &slot));
  std::memcpy(pin, packed_in, nb);
  CK(pull_small(h, dst, pin, nb));
  CK(pin_release(h, slot));
  CK(packed_from_common(h, dst));
  h->have_packed = true;
  h->mirror_valid = false;
  return 0;
}

int svihmm_allreduce_host(svihmm_ctx* h, double* buf, int64_t n, int32_t op) {
  if (!h || !buf || n <= 0) return fail("svihmm_allreduce_host: bad arguments");
  if (!h->comm) return fail("svihmm_allreduce_host: communicator not initialised");
  CK(set_device(h));
  // a persistent staging buffer: the barriers of a timed region must not pay hipMalloc / hipFree
  CK(ensure(h->commtmp, (size_t)n * sizeof(double)));
  void* tp = h->commtmp.p;
  HIPCK(hipMemcpyAsync(tp, buf, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  NCCLCK(ncclAllReduce(tp, tp, (size_t)n, ncclDouble, op == 1 ? ncclMax : ncclSum, h->comm, h->stream));
  HIPCK(hipMemcpyAsync(buf, tp, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- measurement --------------------------------------------------------------------------
static void drain(svihmm_ctx* h) {
  for (auto& p : h->pending) {
    hipEventSynchronize(p.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      h->ms[p.slot] += ms;
      h->cnt[p.slot] += 1;
    }
    h->pool.push_back(p.e0);
    h->pool.push_back(p.e1);
  }
  h->pending.clear();
}

int svihmm_profile_enable(svihmm_ctx* h, int32_t on) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  if (!on) drain(h);
  h->prof = on != 0;
  // on = 1: every slot; on = SVIHMM_PROF_SLOTS | (1 << slot) | ...: only those slots (an event pair
  // between two dependent kernels costs a few microseconds of dispatch: a timed region that wants
  // one kernel's duration should not pay for eleven)
  h->prof_mask = (on & SVIHMM_PROF_SLOTS) ? ((uint32_t)on & 0xfffu) : 0xffffffffu;
  return 0;
}
int svihmm_profile_reset(svihmm_ctx* h) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  drain(h);
  for (int i = 0; i < SVIHMM_NKERN; ++i) { h->ms[i] = 0; h->cnt[i] = 0; }
  return 0;
}
int svihmm_profile_read(svihmm_ctx* h, double ms_out[SVIHMM_NKERN], int64_t count_out[SVIHMM_NKERN]) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  HIPCK(hipStreamSynchronize(h->stream));
  drain(h);
  for (int i = 0; i < SVIHMM_NKERN; ++i) { ms_out[i] = h->ms[i]; count_out[i] = h->cnt[i]; }
  return 0;
}
const char* svihmm_last_kernel_name(svihmm_ctx* h, int32_t slot) {
  if (!h || slot < 0 || slot >= SVIHMM_NKERN || !h->last_kernel[slot]) return "";
  return h->last_kernel[slot];
}
int svihmm_set_variant(svihmm_ctx* h, int32_t which, int32_t value) {
  if (!h || which < 0 || which >= 24) return fail("svihmm_set_variant: bad arguments");
#ifndef SVIHMM_MEASURE
  // codes under which a call's results are invalid exist in the measurement build only
  if (which == 7 && value == 9)
    return fail("svihmm_set_variant: measurement-only code (build with -DSVIHMM_MEASURE: make measure)");
#endif
  h->variant[which] = value;
  return 0;
}

// ---- diagnostics ---------------------------------------------------------------------------
int svihmm_selftest_mfma(svihmm_ctx* h, const double* A16x4, const double* B4x16, double* C16x16) {
  if (!h) return fail("NULL handle");
  CK(set_device(h));
  CK(ensure(h->scratch, (64 + 64 + 256) * sizeof(double)));
  double* dA = (double*)h->scratch.p; double* dB = dA + 64; double* dC = dB + 64;
  HIPCK(hipMemcpyAsync(dA, A16x4, 64 * 8, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(dB, B4x16, 64 * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, h->stream, dA, dB, dC);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(C16x16, dC, 256 * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return 0;
}

}  // extern "C"

