// host.h -- host-side declarations shared by the translation units of libsvihmm_hip.so:
// error handling, the handle (svihmm_ctx), buffer helpers, the profiling scope and the launchers
// each kernel family's translation unit exports to the others.
//   svihmm_hip.hip     C ABI, uploads, SVI loop, multi-GPU, misc + SVI kernels
//   tu_emission.hip    emission kernels (kernels_emission.h) and their launchers
//   tu_recursion.hip   sweeps / scans / FFBS (kernels_recursion.h) and their launchers
//   tu_stats.hip       statistics GEMMs + finalize (kernels_stats.h) and their launchers
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/svihmm.h"
#include "../../include/svihmm_debug.h"
#include "svihmm_common.h"

// ------------------------------------------------------------------------------------
//  error handling
// ------------------------------------------------------------------------------------
extern thread_local std::string g_err;
inline int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess)                                                          \
      return fail(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ \
                  ":" + std::to_string(__LINE__) + ")");                           \
  } while (0)
#define NCCLCK(x)                                                                  \
  do {                                                                             \
    ncclResult_t r_ = (x);                                                         \
    if (r_ != ncclSuccess)                                                         \
      return fail(std::string(#x) + ": " + ncclGetErrorString(r_));                \
  } while (0)
#define CK(x)              \
  do {                     \
    if (int r__ = (x)) return r__; \
  } while (0)



// ------------------------------------------------------------------------------------
//  host side
// ------------------------------------------------------------------------------------
#define SVIHMM_INT_ST32 0x10000u   // internal kernel flag: scaled emission output stored as float
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};
inline int ensure(Buf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return 0;
  if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
  // 512 bytes past `cap` always belong to the allocation: kernels that stage padded state
  // columns read up to 63 doubles beyond the last row (K > 64: state groups of 64)
  size_t want = bytes + bytes / 8 + 256;
  HIPCK(hipMalloc(&b.p, want + 512));
  b.cap = want;
  return 0;
}
inline void release(Buf& b) {
  if (b.p) hipFree(b.p);
  b.p = nullptr; b.cap = 0;
}

enum { KS_EMISSION = 0, KS_FB, KS_POSTERIOR, KS_STATS, KS_FINALIZE, KS_FFBS, KS_MISC,
       KS_ALLREDUCE, KS_H2D, KS_D2H, KS_RES0, KS_RES1 };

struct Pending { int slot; hipEvent_t e0, e1; };

// one svihmm_svi_iteration call, kept until the iteration is known to have reached the loop's state (the host runs at
// most eight iterations ahead of the device: the window-start ring): what svi_recover replays on stream events
struct SviLogEntry {
  int it = -1, B = 0, nwin = 0, Lm = 0, off = 0, len = 0;
  uint32_t flags = 0;
  double rho = 0.0, bA = 0.0, bE = 0.0, vmin_before = 0.0;
  bool f32_ok_before = true;
  std::vector<int64_t> starts;
};
struct svihmm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // the device's shape (svihmm_create): launch-shape thresholds tuned on the 256-CU MI355X scale with it (cu_scaled)
  int ncu = 256, waves_per_cu = 32;
  // data
  int64_t T = 0; int D = 0; bool have_mask = false;
  Buf obs, mask;
  // globals
  int K = 0;
  Buf mod_init, ltran, Aexp, AexpT;
  Buf AexpF, AexpTF;        // float copies for the fp32 mode's wide-model sweeps (256 + 16 rows, slack zeroed)
  bool have_globals = false;
  // transition expectations below the range exp() represents with headroom: every recursion goes
  // through the literal log-domain kernel (k_fb_exact); f32_ok: within the range of a float
  bool exact_log = false, f32_ok = true;
  // emission
  int eK = 0, eD = 0, Kp = 0, F = 0, Fp = 0;
  Buf theta, theta_orb, fab, niw, cat_table, partc, prior, vlb_aux, gen_z;
  int64_t gen_T = 0;
  double* vlb_host = nullptr;   // pinned + mapped: [3][K] ELBO terms
  int prior_K = 0, prior_D = 0, vlb_host_K = 0;
  void *slack_a = nullptr, *slack_t = nullptr;   // Aexp / AexpT whose slack rows are zeroed
  int slack_k = 0;
  void* orb_zero_p = nullptr; size_t orb_zero_n = 0;
  const double* chain_kbef = nullptr; int chain_C = 0, chain_L = 0, chain_T = 0;   // of the last launch_fb_chain
  int lb_pending = 0;       // windows whose local_lb sum has not been written to packed yet
  bool orb_valid = false;   // theta_orb matches theta (k_theta_orbit ran since the last parameter upload)
  Buf uwb;                  // fp32 mode: centred factors U_k as bf16 triples + bias (k_emission_bf16x3)
  void* uw_zero_p = nullptr;
  bool uw_valid = false;    // uwb matches the NIW factors in h->niw
  Buf uwd;                  // the same for 32 < D <= 64 / wide models (k_emission_bf16x3d: 32 KB pair records)
  void* uwd_zero_p = nullptr; size_t uwd_zero_n = 0;
  bool uwd_valid = false, emd_attr_set[2] = {false, false};
  bool emb_attr_set = false;
  bool emis_cat = false; int V = 0;          // Categorical emission: table [V][K] = E log theta
  bool emis_diag = false;                    // diagonal Gaussian family: 2 D + 1 features, h->niw = [mu | nus | alphas | betas]
  bool tab_diag = false;                     // feature table currently on the device is the diagonal one
  void* theta_zero_p = nullptr; size_t theta_zero_n = 0;   // what the last theta memset covered
  int tabD = -1;
  // pinned host staging: a ring of slots, each guarded by an event recorded after the copy
  // that uses it, so that parameter uploads and small readbacks never synchronise the stream
  struct PinSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; };
  PinSlot pins[6];
  int pin_next = 0;
  // window starts of the SVI loop's iterations: their own ring, released by the iterations' end
  // events (an event record between two kernels of the chain costs ~7 us of dispatch)
  struct StartSlot { void* p = nullptr; size_t cap = 0; int used_it = -1; };
  StartSlot svi_starts[8];
  const int64_t* starts_pending = nullptr;   // device-visible pinned slot whose pull into h->starts is still owed
  int starts_pending_n = 0;
  // synchronous E-step calls (statistics read back before the call returns): the starts go through ONE mapped slot
  // that the emission kernel reads itself -- free again once the call's final synchronisation has passed
  bool starts_sync_call = false, starts_slot_inflight = false;
  void* starts_slot = nullptr; size_t starts_slot_cap = 0;
  int svi_upload_it = -1;
  int* pin_status = nullptr;                 // pinned: NIW factorisation status (lazy check)
  double* mirror = nullptr; size_t mirror_cap = 0;   // pinned + mapped copy of `packed`
  bool mirror_valid = false;
  bool status_pending = false;
  bool status_auto = false;      // the pending status word stems from an automatic theta rebuild (params_follow_centre)
  bool have_emission = false;
  // work
  Buf starts, ll, la, lb, q, lse_part, local_lb, logz, part, packed, scratch;
  // scaled linear-domain sweeps: per-row binary exponents, (na, k) records, 1/Z factors,
  // Eh of host-supplied lliks; log-domain intermediates materialised on demand (m_*)
  Buf kexp, hx, gx, zfac, llE, m_ll, m_la, m_lb, chain, chain2;
  Buf ll0, a0v, a0e;               // first-row log-likelihoods of the windows; initial messages + exponents (k_lin_init)
  bool lin_mode = false;           // ll/la/lb hold Eh / ah / bh of the last sweep (not logs)
  bool q_valid = false;            // lin_mode: var_x has been formed from ah, bh (k_lin_posterior)
  bool lin_stale = false;          // parameters changed since: logs can no longer be rebuilt
  bool last_host_ll = false;       // the last sweep ran on host-supplied lliks
  bool eh_in_llE = false;          // scaled emission lives in llE (h->ll holds the plain lliks)
  uint32_t last_flags = 0;
  int m_b0 = 0, m_nb = 0;          // window range currently materialised in m_*
  int lastB = 0, lastLm = 0;       // shape of the intermediates currently held
  int curB = 0;                    // windows of the batch being processed
  int hostB = 0, hostLm = 0;       // shape of host-uploaded lliks
  bool have_host_ll = false;
  bool have_packed = false;
  bool have_lb = false;           // lbeta materialised by the last call
  // variants: [0] emission (0 auto,1 outer,2 mfma) [1] stats (0 auto,1 outer,2 mfma,3 pipelined)
  // [2] sweeps (0 auto,1 wave,2 log-MFMA,3 scaled) [3] emission row tiles
  // [4] two-stream E-step pipeline (0 auto,1 off,2 on)
  // [8] row chunks of the statistics GEMM (0 = automatic)
  int variant[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // second stream + events of the pipelined E-step (created on first use)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_em[2] = {nullptr, nullptr}, ev_sw[2] = {nullptr, nullptr};
  // profiling (prof_mask: bit s = record events around launches of slot s)
  bool prof = false;
  uint32_t prof_mask = 0xffffffffu;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  double ms[SVIHMM_NKERN] = {0};
  const char* last_kernel[SVIHMM_NKERN] = {nullptr};   // name of the kernel the slot's last launch dispatched (svihmm_last_kernel_name)
  int64_t cnt[SVIHMM_NKERN] = {0};
  // device-resident SVI loop (svihmm_svi_*): var_tran | prior_tran | var_init | vlb[K] | logdet[K] |
  // prior_logpart[K]; prior block [mu0 | sigma0 | kappa0 | nu0]; GTH scratch; elbo / event ring
  Buf svi_state, svi_prior, svi_work, commtmp;
  int svi_K = 0, svi_D = 0, svi_maxit = 0;
  int svi_family = 0;       // 0 NIW, 1 diagonal Gaussian, 2 Categorical
  double svi_zsign = 1.0, svi_prior_const = 0.0;
  double* svi_elbo = nullptr; int svi_elbo_cap = 0;      // pinned + mapped: elbo_vec
  std::vector<hipEvent_t> svi_ev;                        // iteration boundaries (iter_time)
  std::vector<int> svi_ev_begin;                         // event that marks the start of iteration it
  int svi_last_it = -1;
  bool svi_active = false, svi_f32_ok = true, svi_adagrad = false;
  double svi_vmin = 0.0;    // host-side lower bound of var_tran's entries (plain rho steps with prior_tran >= 1 only raise it)
  hipEvent_t svi_ea = nullptr, svi_eb = nullptr, globals_ev = nullptr;   // side-stream globals kernel
  hipEvent_t svi_ec = nullptr, svi_ed = nullptr;   // theta ready / side-stream ELBO kernels done
  hipStream_t stream3 = nullptr;                   // the ELBO kernels' own stream
  bool vlb_pending = false;
  // device-side dependencies of the loop (round 5, device_helpers.h): counters [step | globals | theta | side] in
  // HBM with the totals the host expects, the status word of the gates, iteration stamps in mapped host memory
  Buf svi_sync;
  // (64-bit on the host, their low 32 bits on the device: the counters wrap, svi_gate compares signed differences)
  unsigned long long tgt_step = 0, tgt_glob = 0, tgt_theta = 0, tgt_side = 0;
  bool svi_flags = false;          // this loop runs on counters instead of stream-order events
  int svi_concurrent = -1;         // -1 not probed; 1: kernels of two streams run side by side (svi_probe_concurrency)
  int* svi_status_dev = nullptr;   // device address of pin_status[1]: a gate that gave up
  // ELBO kernels of iteration it, launched during the host call of iteration it + 1 behind its sweeps' gate
  // (they then run beside the sweeps' 128 waves instead of beside the emission GEMM)
  bool elbo_pending = false; int elbo_pend_it = -1, elbo_pend_slot = 0;
  unsigned long long tgt_early = 0;   // sweep launches that signal their start (counter 4)
  // bounded waits + recovery (round 6): gate bound from the measured iteration period, the iterations in flight,
  // which iterations were timed by stream events (after a mid-loop switch away from the counters)
  unsigned long long svi_period_ticks = 0;   // longest iteration seen so far (device stamps), 0: none yet
  unsigned long long svi_ticks = 0;          // bound the gates launched now carry (svi_gate_ticks)
  SviLogEntry svi_log[16];
  std::vector<char> svi_it_events;
  int svi_cur_it = -1;               // iteration whose E-step is being launched (poison value of its sweeps)
  bool svi_replaying = false;
  int svi_recoveries = 0;            // times this loop left the counters mid-way (svihmm_svi_recoveries)
  // fused sweep + statistics launch (tu_fused.hip): band counters with their running targets, cached readiness orders
  Buf pipe_cnt;
  unsigned pipe_tgt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  struct PipeTab { Buf buf; int Lq = -1, off = -1, Lm = -1, NS = -1, B = -1, nst = -1; bool wrap = false; unsigned long long stamp = 0;
                   int thr[12], em_thr[12], em_n = 0, ntile = 0, nround = 0; };    // (host copies of what the plan needs: a hit costs no sort)
  unsigned em_tgt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // emission rounds inside the fused launch: counter targets (monotonic)
  // An emission launch held back for the fused E-step kernel (launch_emission with em_defer_req set has done everything
  // but the launch: buffers, theta orbit, the pending window starts); launch_emission_deferred sends it after all
  // fp32 mode, minibatch-sized batch bound for the fused sweep + statistics launch: the emission kernel of the mode writes
  // float Eh (eh_float), the messages and the statistics stay fp64 (cur_f32 = false) -- the fused fp64 stages behind a
  // sweep that re-normalises every fourth step end earlier than float messages + the bf16 statistics kernel behind it
  bool f32_fused_req = false, eh_float = false;
  bool em_defer_req = false;
  struct EmDeferred { bool active = false; const int64_t* starts = nullptr; int64_t* starts_copy = nullptr; int nstarts = 0;
                      int B = 0, Lm = 0; uint32_t flags = 0; double* out = nullptr; double* kexp = nullptr; double* ll0 = nullptr; } em_def;
  PipeTab pipe_tabs[8];
  unsigned long long pipe_stamp = 0;
  bool sweep_signalled = false;    // this E-step's sweep launch does
  bool in_svi_estep = false;       // estep_core is running for svihmm_svi_iteration
  unsigned long long* svi_ts = nullptr; unsigned long long* svi_ts_dev = nullptr; int svi_ts_cap = 0;   // pinned + mapped: [2 it] begin, [2 it + 1] end (wall_clock64)
  SviSync theta_sy = {nullptr, 0u, nullptr, nullptr, nullptr};   // what the next theta-builder launch arrives on
  double wall_clock_khz = 0.0;
  bool svi_globals_ready = false;   // Aexp / mod_init / var_init[slot] of the NEXT iteration are computed (or in flight)
  int svi_globals_slot = 0, svi_vi_cur = 0;   // var_init slot the pending globals write / the last iteration used
  // The resident observations are kept centred: obs_dev[t] = obs_caller[t] - shift.  The shift is
  // chosen at upload (a point inside the data) and moved by svihmm_shift_obs; it never shows at the
  // ABI: means come in / go out in the caller's coordinates (set_emission_niw, svi_begin,
  // svi_read_state), statistics are handed out in the caller's coordinates (launch_mirror), the
  // device-side state (h->niw, svi_prior, packed) lives in centred coordinates.
  std::vector<double> shift;     // [D]; empty = no observations yet
  Buf shift_d;                   // the same vector on the device
  bool shifted = false;          // some component is non-zero
  bool center_pending = false;   // svihmm_alloc_obs: centre on the first block that arrives
  bool center_deferred = false;  // uploaded / un-centred under a Categorical table: a Gaussian family that follows centres it
  int shift_epoch = 0, prior_epoch = -1;
  std::vector<double> prior_mu0; // caller-coordinate prior means of svihmm_set_emission_prior
  // precision mode (svihmm_set_precision): 0 fp64, 1 fp32 (see the header); cur_f32: the batch in
  // flight / the intermediates currently held are in the fp32 format
  int prec = 0;
  bool cur_f32 = false;
  // comm
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
};

// length of the packed statistics in the layout of the current emission family:
// NIW  [A_raw K*K | xbar K*D | neff K | S K*D*D | lb],  Categorical  [A_raw K*K | counts K*V | lb]
inline size_t packed_len(const svihmm_ctx* h) {
  if (h->emis_cat) return (size_t)h->K * h->K + (size_t)h->K * h->V + 1;
  const size_t D = h->D > 0 ? h->D : 1;
  if (h->emis_diag) return (size_t)h->K * h->K + 2 * (size_t)h->K * D + h->K + 1;
  return (size_t)h->K * h->K + (size_t)h->K * D + h->K + (size_t)h->K * D * D + 1;
}

struct ProfScope {
  svihmm_ctx* h; int slot; hipEvent_t e0 = nullptr, e1 = nullptr; bool on; hipStream_t st;
  ProfScope(svihmm_ctx* h_, int slot_, hipStream_t st_ = nullptr)
      : h(h_), slot(slot_), on(h_->prof && ((h_->prof_mask >> slot_) & 1u)), st(st_ ? st_ : h_->stream) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!h->pool.empty()) { e = h->pool.back(); h->pool.pop_back(); }
      else hipEventCreate(&e);
      return e;
    };
    e0 = get(); e1 = get();
    hipEventRecord(e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(e1, st);
    h->pending.push_back({slot, e0, e1});
  }
};

inline int set_device(svihmm_ctx* h) {
  HIPCK(hipSetDevice(h->device));
  return 0;
}

// a count tuned on the 256 compute units of one MI355X, for this device (CPX partitions, other parts)
inline int64_t cu_scaled(const svihmm_ctx* h, int64_t v) {
  const int64_t r = (v * h->ncu + 128) / 256;
  return r < 1 ? 1 : r;
}
struct StatsPlan { int64_t rpc, nchunk; };
// the natural-gradient step's arguments when it rides in the theta builder's launch (k_svi_step_theta32s, kernels_svi_theta.h)
struct SviStepArgs {
  const double* packed; const double* prior_tran; double* var_tran; const double* prior;
  double rho, bA, bE, nwin; double* lb_keep; double* ada_G;
  SviSync sy;          // the step's gate / poison / arrival
};
// up to this many windows the wave-per-window scaled sweep beats the MFMA one (which is
// latency-bound at ~0.9 us per step however few windows it gets): 2 x 1024 waves are resident
// at once on 256 CUs (190 VGPRs: two per SIMD); measured (tools/sweep_crossover.py, K = 64, Lm = 257)
// 0.16 ms at 64 .. 0.19 ms at 1024 windows against 0.23 .. 0.25 ms, 0.33 against 0.26 ms at 1399
inline int lin_wave_max(const svihmm_ctx* h) { return 4 * h->ncu + 1; }
// up to this many windows: four waves per (window, direction); 2 x 256 x 4 = 2048 waves, two per SIMD of 256 CUs
inline int lin_wave4_max(const svihmm_ctx* h) { return h->ncu; }
// up to this many windows: the register-resident one-wave kernel (k_wave_linr; round 5)
inline int lin_waver_max(const svihmm_ctx* h) { return h->ncu; }

extern "C" {
int d2h(svihmm_ctx* h, void* dst, const void* src, size_t bytes);
int ensure_fb_lin(svihmm_ctx* h, int B, int Lm);
int ensure_feature_table(svihmm_ctx* h);
int ensure_q(svihmm_ctx* h, int B, int Lm, hipStream_t stream);
int ensure_stats(svihmm_ctx* h, int64_t nchunk_total);
int flush_lb(svihmm_ctx* h, hipStream_t stream);
int launch_diag_to_theta(svihmm_ctx* h, int K, int D);
int launch_emission(svihmm_ctx* h, int B, int Lm, uint32_t flags, bool scaled = false, const int64_t* starts_dev = nullptr, double* out = nullptr, double* kexp_out = nullptr, hipStream_t stream = nullptr, size_t min_lds = 0, double* ll0_out = nullptr);
int launch_fb(svihmm_ctx* h, int B, int Lm, int dir0, int ndir, const double* ll = nullptr, double* la = nullptr, double* lb = nullptr);
int launch_fb_fused(svihmm_ctx* h, int B, int Lm, bool want_lb, bool total);
int launch_fb_lin(svihmm_ctx* h, int B, int Lm, bool total);
int launch_fb_lin_range(svihmm_ctx* h, int b0, int nb, int Lm, hipStream_t stream);
int launch_niw_to_theta(svihmm_ctx* h, int K, int D, double* logdet_out, const SviStepArgs* step = nullptr);
bool step_theta_ok(const svihmm_ctx* h, int K, int D);
int launch_posterior(svihmm_ctx* h, int B, int Lm, bool total);
int launch_scale_ll(svihmm_ctx* h, int B, int Lm);
int launch_scale_ll_f32(svihmm_ctx* h, int B, int Lm);
bool f32_wide_ok(const svihmm_ctx* h, int64_t n);
bool stats_bf16w_shape_ok(const svihmm_ctx* h, int64_t n);
int launch_stats(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags);
int launch_stats_finalize(svihmm_ctx* h, int64_t nchunk, hipStream_t stream);
int launch_stats_range(svihmm_ctx* h, int b0, int nb, int Lq, int off, int Lm, uint32_t flags, StatsPlan plan, int64_t chunk_base, hipStream_t stream);
int launch_sum_lb(svihmm_ctx* h, int B, hipStream_t stream);
int materialise(svihmm_ctx* h, int b0, int nb);
int prepare_ll(svihmm_ctx* h, const int64_t* starts, int B, int Lm, uint32_t flags, bool need_obs_for_stats, bool lin = false);
StatsPlan stats_plan(const svihmm_ctx* h, int64_t n, int forced = 0);
int upload_feature_table(svihmm_ctx* h, int D, int K, bool diag = false);
bool use_chain(const svihmm_ctx* h, int B, int Lm);
int wait_globals(svihmm_ctx* h);
int svi_flush_elbo(svihmm_ctx* h);
int ensure_starts_pulled(svihmm_ctx* h);
int cat_uncentre(svihmm_ctx* h);
int launch_fb_chain(svihmm_ctx* h, int Lm, bool total);
int launch_emission_deferred(svihmm_ctx* h);
bool sweep_emission_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags);
bool sweep_mixed_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags);
bool sweep_stats_ok(const svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags);
int launch_sweep_stats(svihmm_ctx* h, int B, int Lq, int off, int Lm, uint32_t flags);
SviSync sweep_gate(svihmm_ctx* h, hipStream_t stream);
int launch_niw_vlb(svihmm_ctx* h, int K, int D, const double* dmu, const double* dsg, const double* dka,
                   const double* dnu, double* th2, int* dstat, double* ld, const double* p0, double* dout);
}
