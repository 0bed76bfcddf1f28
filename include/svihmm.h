/* svihmm.h -- C ABI of the MI355X-native SVI-HMM E-step engine (libsvihmm_hip.so).
 *
 * This is the drop-in boundary for the hot path of dillonalaird/pysvihmm:
 *   emission expected log-likelihood -> log-domain forward/backward ->
 *   posterior marginals -> expected sufficient statistics (natural-gradient
 *   direction) -> (all-reduce) -> host global step.
 * The reference has a single native boundary on this path, the Cython function
 *   hmm_fast.FFBS(self, var_init, lalpha_init=None)          (hmm_fast.pyx:43-124)
 * bound as VariationalHMMBase.ffbs_fast (hmmbase.py:409-411); everything else on
 * the path is NumPy inside Python methods.  Each entry point below names the
 * reference code (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - Plain C: pointers + sizes, no C++/torch types.  All arrays are C-contiguous
 *     row-major float64 unless stated; "host" pointers are borrowed for the call
 *     only (caller keeps ownership); the library owns all device memory behind
 *     the opaque handle.
 *   - Every function returns 0 on success, non-zero on failure; the message of
 *     the last failure on the calling thread is svihmm_last_error().  The Python
 *     host raises RuntimeError (the reference's only exception type,
 *     hmmbase.py:134, hmmsgd_metaobs.py:76,177,191,344).
 *   - One thread per handle, one handle per device (the reference is
 *     single-threaded and non-re-entrant).  Calls are synchronous with respect to
 *     their host output buffers; device work runs on the handle's own HIP stream.
 *   - A "window" is a meta-observation (hmmsgd_metaobs.py:42-45): Lm consecutive
 *     rows of obs starting at starts[b] (inclusive bounds i1=starts[b],
 *     i2=starts[b]+Lm-1).  A full-chain E-step is B=1, starts={0}, Lm=T.
 */
#ifndef SVIHMM_H
#define SVIHMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: svihmm_export_packed / svihmm_import_packed, svihmm_svi_set_adagrad / _read_adagrad,
 *    svihmm_svi_begin_diag / _begin_cat / _read_factors added, slot mask
 *    of svihmm_profile_enable (round 4; nothing else changed meaning).
 * 2: svihmm_get_shift added, svihmm_shift_obs no longer changes what later calls mean (the shift is
 *    the handle's business), svihmm_set_emission_diag added; 1 was rounds 1-2 (svi_*,
 *    set_precision, shift_obs, ffbs_sample, comm_count were added under it). */
#define SVIHMM_ABI_VERSION 3

typedef struct svihmm_ctx svihmm_ctx;

/* ---- flags ------------------------------------------------------------------ */
/* Masked rows are treated as NaN rows when computing lliks (-> lliks row = 0):
 * full_local_update semantics, hmmsgd_metaobs.py:1166-1168,1176.  Without it
 * masked rows still contribute to lliks (infer() semantics, quirk Q9,
 * hmmsgd_metaobs.py:314-315 commented out). */
#define SVIHMM_MASK_AS_NAN 1u
/* Transition statistic sum_{t=0}^{Lm-1} q[t-1] (x) q[t] with t-1 wrapping to the
 * window's last row (quirk Q1, hmmsgd_metaobs.py:877-878).  Without it the batch
 * form sum_{t=1}^{Lm-1} (hmmbatchcd.py:183-184, hmmbatchsgd.py:224-225). */
#define SVIHMM_TRANS_WRAP 2u
/* Use the lliks previously uploaded with svihmm_set_lliks (generic emission
 * plugin route) instead of evaluating the NIW emission kernel. */
#define SVIHMM_USE_HOST_LLIKS 4u
/* svihmm_estep_minibatch only: also write lbeta to HBM during the sweep (the log-domain fused
 * backward sweep otherwise keeps it in registers).  A hint: without it a later read of lbeta
 * costs one more backward sweep over the lliks still held (and fails once the parameters have
 * changed); the scaled sweeps rebuild all logs on demand and ignore the flag. */
#define SVIHMM_KEEP_LBETA 8u

/* svihmm_svi_iteration only: make the log-domain lliks / lalpha / lbeta of the batch's LAST window
 * readable (svihmm_read_rows) after the call -- they are otherwise rebuilt on demand from the
 * current parameters, which the iteration's global step replaces.  The reference leaves the
 * state of the last meta-observation on the object (hmmsgd_metaobs.py:405-436). */
#define SVIHMM_SVI_KEEP_WINDOW 16u
#define SVIHMM_SVI_MIN_PSEUDOCOUNT 2.5e-3

/* ---- errors / lifecycle ------------------------------------------------------- */
const char* svihmm_last_error(void);
int svihmm_abi_version(void);
int svihmm_device_count(int* n_out);
int svihmm_create(int device_id, svihmm_ctx** out);
int svihmm_destroy(svihmm_ctx* h);
int svihmm_sync(svihmm_ctx* h);

/* ---- precision mode -----------------------------------------------------------------
 * SVIHMM_F64 (default): the reference's own type throughout (hmmbase.py:102-103).
 * SVIHMM_F32: north_star's second tolerance ("posteriors and natural gradients within 1e-3 fp32").
 *   For the E-step fast path it covers -- NIW emissions, K <= 64, window batches (not the
 *   whole-chain scan, not host-supplied lliks) -- the scaled emission likelihoods and the scaled
 *   forward / backward messages are STORED as fp32 (half the HBM traffic of the sweeps) and the
 *   expected-sufficient-statistics GEMM runs on v_mfma_f32_16x16x4_f32 (twice the matrix rate),
 *   each row chunk accumulated in fp32 and the chunks reduced in fp64.  The emission quadratic
 *   form of batches from 8192 rows (D <= 32; from 32768 rows at D <= 64) is evaluated CENTRED, c_k - |U_k x + b_k|^2 with
 *   U_k = sqrt(nu_k/2) L_k^-1, on the bf16 matrix pipe: x and U_k as three bf16 terms each (= the
 *   values to fp32 accuracy), six products into fp32 accumulators (round 3; the expanded feature
 *   form of the fp64 kernels cancels ~1e5 : 1, which fp32 cannot carry -- smaller batches and
 *   D > 32 keep that fp64 GEMM).  The arithmetic of the recursion itself stays fp64 (exact
 *   binary exponents; the launch is HBM-bound, not flop-bound).  Inputs and outputs of the ABI
 *   stay float64.
 *   Round 5: the centred bf16 x 3 emission covers D <= 64 (64-dimension pair records), the statistics GEMM on
 *   the bf16 pipe covers D <= 64 and wide models, and NIW models with 64 < K <= 256, D <= 64 run the whole fast
 *   path in the fp32 format from 32 768 rows and 192 windows per batch (BASELINE configs[4]: 19 instead of 45 ms
 *   per epoch step, statistics within 3.2e-4 of fp64).
 *   Calls outside that fast path run in fp64 regardless (smaller batches of wide models, other families,
 *   whole-chain scans, host-supplied lliks, transition expectations outside a float's range);
 *   svihmm_get_precision reports whether the last E-step batch actually ran in the fp32 format. */
#define SVIHMM_F64 0
#define SVIHMM_F32 1
int svihmm_set_precision(svihmm_ctx* h, int32_t mode);
int svihmm_get_precision(svihmm_ctx* h, int32_t* mode_out, int32_t* last_batch_f32_out);

/* ---- inputs ----------------------------------------------------------------- */
/* obs[T,D], mask[T] (1 = missing, may be NULL): hmmbase.py:60-65,122-123.
 * Copied to HBM once; NaN entries are preserved.
 * Coordinates: the resident copy is kept CENTRED, obs_dev[t] = obs[t] - c, with c chosen by the
 * library at upload (a point inside the data: column means of a row sample; svihmm_generate: the
 * average of the state means; svihmm_alloc_obs: taken from the first block that arrives) because
 * the emission GEMM evaluates the NIW quadratic form expanded around the resident copy's origin
 * and loses ~5e-16 (mu-c)'W(mu-c) to cancellation.  The model is shift-equivariant and c never
 * shows at this boundary: every mean that comes in (svihmm_set_emission_niw / _diag,
 * svihmm_svi_begin, svihmm_set_emission_prior, svihmm_niw_vlb_terms) or goes out
 * (svihmm_svi_read_state) and all statistics (svihmm_estep_minibatch*, svihmm_read_packed) are in
 * the CALLER's coordinates, exactly as the reference evaluates everything in the coordinates of
 * self.obs (hmmbase.py:219-229).  Only svihmm_allreduce_packed touches it internally: every rank
 * has its own c, so the sum is formed in the common caller coordinates. */
int svihmm_set_obs(svihmm_ctx* h, const double* obs, int64_t T, int32_t D,
                   const uint8_t* mask);

/* The same resident copy filled in pieces (gen_synthetic.py:188-191 read_data_mmap streams
 * [size, D] blocks of the on-disk float64 array): svihmm_alloc_obs sizes it (mask zeroed when
 * with_mask), svihmm_set_obs_rows uploads rows [row0, row0+nrows) and returns when the block
 * is on the device.  Rows never written are undefined. */
int svihmm_alloc_obs(svihmm_ctx* h, int64_t T, int32_t D, int32_t with_mask);
/* Moves the library's centre: c += shift[D] (resident copy, device-side NIW state and prior of a
 * running SVI loop follow; NaN rows stay NaN).  Purely a conditioning hint -- no later call
 * changes its meaning, results stay in the caller's coordinates.  Useful when the automatic
 * centre is poor (strongly drifting data) or was switched off.  svihmm_get_shift reads c[D]. */
int svihmm_shift_obs(svihmm_ctx* h, const double* shift);
int svihmm_get_shift(svihmm_ctx* h, double* shift_out);
int svihmm_set_obs_rows(svihmm_ctx* h, int64_t row0, int64_t nrows, const double* obs,
                        const uint8_t* mask);

/* mod_init[K], ltran[K,K] in the log domain: the psi-expectations of
 * hmmbase.py:214-216 / hmmsgd_metaobs.py:502-504 (computed on the host with
 * SciPy's digamma, uploaded once per minibatch).  The FFBS variant passes
 * ltran = log(var_tran + DBL_EPSILON) instead (hmm_fast.pyx:91-93).
 * Dynamic range: the fast recursions carry exp(ltran).  If any entry lies below
 * SVIHMM_LTRAN_LINEAR_MIN (Dirichlet pseudo-counts of ~2e-3 and less: psi(1e-3) = -1000, not
 * representable) every recursion with these globals runs the reference's own
 * np.logaddexp.reduce form instead (hmmbase.py:295, 319; one exp per state PAIR and step --
 * correct, 5-7x slower); the fp32 mode needs entries above
 * SVIHMM_LTRAN_F32_MIN and computes in fp64 otherwise. */
#define SVIHMM_LTRAN_LINEAR_MIN (-600.0)
#define SVIHMM_LTRAN_F32_MIN (-60.0)
int svihmm_set_globals(svihmm_ctx* h, int32_t K, const double* mod_init,
                       const double* ltran);

/* NIW mean-field emission factors mu[K,D], sigma[K,D,D], kappa[K], nu[K]
 * (pybasicbayes Gaussian.{mu_mf,sigma_mf,kappa_mf,nu_mf}); replaces the K calls
 * odist.expected_log_likelihood(obs) at hmmbase.py:219-220,
 * hmmsgd_metaobs.py:508-509,685-686,815-816,1175-1176, hmm_fast.pyx:84-85.
 * The Cholesky factorisation runs on the device and the call does not wait for it: a
 * sigma that is not positive definite is reported by the next synchronising call
 * (svihmm_sync, svihmm_loglik, svihmm_forward_backward, svihmm_estep_minibatch with an
 * output buffer, svihmm_read_packed).  Reported the same way: a factor so far from the library's
 * centre c of the resident observations for its spread ((mu-c)' (nu/2 sigma^-1) (mu-c) > 1e9,
 * i.e. a state mean > 3e4 standard deviations away from the data) that the emission GEMM -- which
 * evaluates the quadratic form expanded around c -- would lose more than 5e-7 in the
 * log-likelihoods; svihmm_shift_obs moves the centre.
 * D <= SVIHMM_NIW_MAX_D; wider observations go the svihmm_set_lliks route (the classes do). */
#define SVIHMM_NIW_MAX_D 96
int svihmm_set_emission_niw(svihmm_ctx* h, int32_t K, int32_t D, const double* mu,
                            const double* sigma, const double* kappa,
                            const double* nu);

/* Diagonal-covariance Gaussian factors (pybasicbayes DiagonalGaussian; BASELINE configs[0]; the
 * plugin objects the reference evaluates at hmmbase.py:219-220 and updates with
 * meanfieldupdate at hmmbatchcd.py:189): per state and dimension a normal-inverse-gamma mean-field
 * factor  sigma_d^2 ~ InvGamma(alphas, betas),  mu_d | sigma_d^2 ~ N(mu, sigma_d^2 / nus);  all four
 * arrays [K,D].  E_q log N(x | mu, diag sigma^2) is linear in the 2 D + 1 features (x_d^2, x_d, 1):
 * the emission and statistics GEMMs run on that feature table instead of the (D+1)(D+2)/2
 * products a full covariance needs (D = 32: 80 instead of 576 feature rows).  Afterwards the
 * packed statistics have the layout
 *     [ A_raw K*K | xbar K*D | neff K | xsq K*D | lb ]      (svihmm_packed_len)
 * with xsq[k,d] = sum over unmasked rows of q[t,k] x[t,d]^2 -- what the conjugate update needs
 * (nus += n, mu = (nus0 mu0 + xbar)/nus, alphas += n/2, betas += (xsq + nus0 mu0^2 - nus mu^2)/2).
 * Same asynchronous status word as svihmm_set_emission_niw (a non-positive nus / alphas / betas
 * entry, or a factor > 3e4 standard deviations from the data).  The device-resident SVI loop
 * (svihmm_svi_*) is NIW only; this family runs through svihmm_estep_minibatch*. */
#define SVIHMM_DIAG_MAX_D 128
int svihmm_set_emission_diag(svihmm_ctx* h, int32_t K, int32_t D, const double* mu,
                             const double* nus, const double* alphas, const double* betas);

/* Generic plugin route: lliks[B,Lm,K] evaluated by an arbitrary emission object on
 * the host (any object with expected_log_likelihood), already nan_to_num'ed. */
int svihmm_set_lliks(svihmm_ctx* h, const double* lliks, int32_t B, int32_t Lm);

/* ---- a3: emission expected log-likelihood ---------------------------------------- */
/* out_lliks[B,Lm,K] = nan_to_num(E_q log p(obs[starts[b]+t] | theta_k)). */
int svihmm_loglik(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                  uint32_t flags, double* out_lliks);

/* ---- a4..a7: messages, posterior, local bound ------------------------------------- */
/* lalpha/lbeta/var_x [B,Lm,K], local_lb[B]; any output pointer may be NULL.
 *   lalpha : hmmbase.py:292-295 == hmmsgd_metaobs.py:800-803
 *   lbeta  : hmmbase.py:316-320 == hmmsgd_metaobs.py:851-855
 *   var_x  : hmmbase.py:226-229 == hmmsgd_metaobs.py:516-519
 *   local_lb[b] = sum_t LSE_k lalpha[b,t,k]  (hmmsgd_metaobs.py:271; quirk Q4) */
int svihmm_forward_backward(svihmm_ctx* h, const int64_t* starts, int32_t B,
                            int32_t Lm, uint32_t flags, double* out_lalpha,
                            double* out_lbeta, double* out_var_x,
                            double* out_local_lb);

/* ---- a3..a9: whole minibatch E-step -> packed expected sufficient statistics --- */
/* packed = [ A_raw(K*K) | xbar(K*D) | neff(K) | S(K*D*D) | lb(1) ], length
 * svihmm_packed_size(K,D):
 *   A_raw = sum_windows sum_t q[t-1] (x) q[t]       (hmmsgd_metaobs.py:876-878;
 *           the caller adds B*(prior_tran-1), quirk Q2)
 *   xbar_k, neff_k, S_k = sum over unmasked rows of q[t,k]*{x, 1, x x'}
 *                                                    (util.py:73-83, :884-904)
 *   lb    = sum_windows local_lb                     (hmmsgd_metaobs.py:436)
 * out_packed may be NULL: the result then stays in HBM for
 * svihmm_allreduce_packed / svihmm_read_packed. */
int64_t svihmm_packed_size(int32_t K, int32_t D);
int svihmm_estep_minibatch(svihmm_ctx* h, const int64_t* starts, int32_t B,
                           int32_t Lm, uint32_t flags, double* out_packed);
int svihmm_read_packed(svihmm_ctx* h, double* out_packed);
/* Buffered meta-observations (growBuffer): the E-step runs on windows of length Lm,
 * the statistics are taken over the inner segment [inner_off, inner_off+inner_len) of
 * each window only, with the wrap-around term inside that segment
 * (intermediate_pars_buffer, hmmsgd_metaobs.py:932-1008); lb still sums over the
 * whole window (hmmsgd_metaobs.py:436). */
int svihmm_estep_minibatch_ex(svihmm_ctx* h, const int64_t* starts, int32_t B,
                              int32_t Lm, int32_t inner_off, int32_t inner_len,
                              uint32_t flags, double* out_packed);

/* ---- the SVI loop with the variational state resident in HBM ----------------------------------
 * hmmsgd_metaobs.VBHMM.infer (:347-445) iterates  { stationary init :413-418, psi-expectations
 * :502-504, E-step over the minibatch :405-436, global natural-gradient step :1010-1069
 * (util.py:28-60), elbo_vec[it] = lb + global_lower_bound() :444-445, :273-296 }.  With these
 * entry points var_tran and the K NIW factors stay on the device between iterations: per
 * iteration the host sends the window starts and three scalars, and nothing has to come back
 * until the caller wants the ELBO trace or the parameters.  Calls are asynchronous (they return
 * when the work is enqueued); svihmm_svi_read_elbo / svihmm_svi_read_state synchronise.
 *
 * svihmm_svi_begin: uploads the state.  prior_tran / var_tran [K,K]; NIW prior (mu0 [K,D],
 *   sigma0 [K,D,D], kappa0 [K], nu0 [K]) and current factors (mu, sigma, kappa, nu);
 *   prior_logpart[k] = invwishart_log_partitionfunction(sigma0[k], nu0[k]) (constant of the
 *   factors' ELBO term, computed once by the host); zsign = +1 / -1: the sign with which that
 *   constant enters get_vlb (pybasicbayes / Bishop 10.74, see distributions.Gaussian.get_vlb).
 *   The observations must be resident (svihmm_set_obs / svihmm_generate).  Fails when an entry
 *   of prior_tran is below 1 or one of var_tran below SVIHMM_SVI_MIN_PSEUDOCOUNT: with
 *   prior_tran >= 1 no entry of var_tran falls below min(var_tran, 1) during the loop (the step adds
 *   rho bA nwin (prior_tran - 1) with bA nwin ~ T / 2L, quirk Q2 -- negative and large for sparser
 *   priors), so psi(var_tran) stays inside the range of the linear-domain recursions; other
 *   Dirichlet models take the per-call route (svihmm_set_globals decides per upload).
 * svihmm_svi_iteration(it, ...): one iteration on windows starts[B] of length Lm (statistics over
 *   the inner segment, as svihmm_estep_minibatch_ex; flags: SVIHMM_TRANS_WRAP | ...).
 *   nwin_total = windows of the whole minibatch (= B on one GPU; with a communicator the ranks
 *   hold shards and the packed statistics are all-reduced before the step): quirk Q2 adds
 *   nwin_total * (prior_tran - 1).  rho = (it + tau)^-kappa; bfactA = (T-2L-1)/(2L*S),
 *   bfactE = (T-2L-1)/((2L+1)*S) from the CONSTRUCTOR's L, S (quirk Q3).
 * svihmm_svi_read_elbo: elbo_vec[0..n) and the device time of each iteration in ms (device clock:
 *   from the iteration's first kernel -- or the end of the previous iteration when the loop runs back
 *   to back -- to the last workgroup of its theta builder; either may be NULL).  The loop orders its
 *   streams with device-side counters; a dependency that is not met within a minute (a kernel of
 *   the loop never ran) makes this call fail and ends the loop.  svihmm_svi_read_state: current var_tran [K,K], var_init [K] (the stationary vector
 *   of the last iteration, quirk Q5), NIW factors; any pointer may be NULL. */
int svihmm_svi_begin(svihmm_ctx* h, int32_t K, int32_t D, const double* prior_tran,
                     const double* var_tran, const double* mu0, const double* sigma0,
                     const double* kappa0, const double* nu0, const double* prior_logpart,
                     const double* mu, const double* sigma, const double* kappa, const double* nu,
                     int32_t maxit, double zsign);
int svihmm_svi_iteration(svihmm_ctx* h, int32_t it, const int64_t* starts, int32_t B,
                         int32_t nwin_total, int32_t Lm, int32_t inner_off, int32_t inner_len,
                         uint32_t flags, double rho, double bfactA, double bfactE);
/* The same loop for the two families whose factors are element-wise (round 4):
 * svihmm_svi_begin_diag: distributions.DiagonalGaussian (per dimension a normal-inverse-gamma factor;
 *   prior_blk / factor_blk = [mu | nus | alphas | betas], each [K][D], means in the caller's
 *   coordinates): the Gaussian branch's blend hmmsgd_metaobs.py:1050-1069 in this family's natural
 *   parameters [nu mu, nu, 2 beta + nu mu^2, 2 alpha], theta by the device builder, ELBO term
 *   -KL(q || prior).  (The reference dispatches on Gaussian / Categorical only: an extension.)
 * svihmm_svi_begin_cat: Categorical emitters over ONE integer-valued observation column (D = 1),
 *   Dirichlet factors alpha[K][V] and prior alpha0[K][V]: global step hmmsgd_metaobs.py:1071-1084 with
 *   every window's alpha_0 + counts - 1 (:907-926), the E log theta table rebuilt on the device.
 * svihmm_svi_iteration / svihmm_svi_read_elbo / svihmm_svi_set_adagrad work for all three;
 * svihmm_svi_read_factors hands back the state in the family's block layout (NIW: [mu | sigma |
 * kappa | nu]; any of the three pointers may be NULL). */
int svihmm_svi_begin_diag(svihmm_ctx* h, int32_t K, int32_t D, const double* prior_tran, const double* var_tran,
                          const double* prior_blk, const double* factor_blk, int32_t maxit);
int svihmm_svi_begin_cat(svihmm_ctx* h, int32_t K, int32_t V, const double* prior_tran, const double* var_tran,
                         const double* alpha0, const double* alpha, int32_t maxit);
int svihmm_svi_read_factors(svihmm_ctx* h, double* var_tran, double* var_init, double* factors_out);
/* AdaGrad-scaled transition step (hmmsgd_metaobs.py:179-183 ada_G = ones, :1036-1040
 * ada_G += nats_old^2; adaMatrix = ada_G^.25; nats_new = (1 - 1/adaMatrix) nats_old + A_up/adaMatrix):
 * after svihmm_svi_begin, svihmm_svi_set_adagrad uploads the K x K accumulator (NULL: the plain rho
 * step again); it stays on the device with var_tran, svihmm_svi_read_adagrad hands it back. */
int svihmm_svi_set_adagrad(svihmm_ctx* h, const double* ada_G);
int svihmm_svi_read_adagrad(svihmm_ctx* h, double* ada_G_out);
int svihmm_svi_read_elbo(svihmm_ctx* h, int32_t n, double* out_elbo, double* out_ms);
/* mod_init[K] / ltran[K,K] as the recursions currently hold them (either may be NULL): the last
 * svihmm_set_globals upload, or -- after svihmm_svi_iteration -- the psi-expectations that
 * iteration computed on the device (hmmsgd_metaobs.py:502-504; the reference leaves them on the
 * object as mod_init / mod_tran). */
int svihmm_read_globals(svihmm_ctx* h, double* mod_init_out, double* ltran_out);
int svihmm_svi_read_state(svihmm_ctx* h, double* var_tran, double* var_init, double* mu,
                          double* sigma, double* kappa, double* nu);

/* Categorical emissions (hmmsgd_metaobs.py:907-926, 1071-1084; pybasicbayes Categorical):
 * obs must be [T][1] holding the symbol index 0..V-1 as a double; logp[k][v] =
 * E_q log theta_k[v] = psi(alpha_mf[k][v]) - psi(sum_v alpha_mf[k][v]) (host, SciPy digamma).
 * Afterwards the E-step entry points use the table instead of the NIW kernel and the packed
 * statistics have the layout [A_raw K*K | counts K*V | lb] (counts[k][v] = sum of var_x[t,k]
 * over unmasked rows with symbol v); svihmm_packed_len gives the current length. */
int svihmm_set_emission_cat(svihmm_ctx* h, int32_t K, int32_t V, const double* logp);
int64_t svihmm_packed_len(svihmm_ctx* h);

/* ELBO bookkeeping of the SVI loop (hmmsgd_metaobs.py:273-296 global_lower_bound,
 * hmmbase.py:183-185: sum_k var_emit[k].get_vlb()): the data-dependent scalars of the NIW
 * factors' term for the given mean-field parameters (D <= SVIHMM_NIW_MAX_D; D <= 64 runs the
 * single-wave factorisation, wider factors the workgroup-per-state one the E-step uses there) --
 * out[k] = log det sigma_mf[k], out[K+k] = tr(sigma_mf[k]^-1 sigma_0[k]),
 * out[2K+k] = (mu_mf[k]-mu_0[k])' sigma_mf[k]^-1 (mu_mf[k]-mu_0[k]); the host adds the
 * closed-form parts (digamma / gammaln of nu, kappa).  The prior (mu_0[K,D], sigma_0[K,D,D])
 * is uploaded once with svihmm_set_emission_prior.  A pure function of its arguments: the
 * E-step's own parameter set and the intermediates of the last E-step are not touched. */
int svihmm_set_emission_prior(svihmm_ctx* h, int32_t K, int32_t D, const double* mu0,
                              const double* sigma0);
int svihmm_niw_vlb_terms(svihmm_ctx* h, int32_t K, int32_t D, const double* mu,
                         const double* sigma, const double* kappa, const double* nu,
                         double* out3K);

/* Mean predictive log-probability of the held-out (masked) rows of the given windows
 * (hmmsgd_metaobs.py:1086-1145 pred_logprob / pred_logprob_full, hmmbase.py:322-340):
 * E-step with `flags` (SVIHMM_MASK_AS_NAN: the masked rows are missing), then the mean over
 * the masked rows of LSE_k( log(var_x[t,k] + 1e-9) + E_q log p(x_t | k) ) with the emission
 * term evaluated on the true observations.  out2[0] = mean (NaN if nothing is masked),
 * out2[1] = number of masked rows.  NIW emission only (a host-supplied lliks batch has no
 * "true observation" term). */
int svihmm_pred_logprob(svihmm_ctx* h, const int64_t* starts, int32_t B, int32_t Lm,
                        uint32_t flags, double out2[2]);

/* State decoding of the last estep/forward_backward call for the evaluation metrics
 * (hmmbase.py:346-355 hamming_dist; util.py:236-277 munkres_match's count matrix):
 * z[r] = argmax_k var_x[r, k] for the n = B*Lm rows (window-major; the first maximum wins like
 * np.argmax) and, when true_sts (host, int32[n], same row order) is given,
 * conf[pred*K + true] = number of rows decoded as `pred` whose label is `true` (labels outside
 * [0, K) are skipped).  out_z (host int32[n]) and out_conf (host int64[K*K]) are optional. */
int svihmm_state_argmax(svihmm_ctx* h, const int32_t* true_sts, int32_t* out_z,
                        int64_t* out_conf);

/* Readback of the intermediates of the last estep/forward_backward call
 * (what 0: lliks, 1: lalpha, 2: lbeta, 3: var_x; each [B,Lm,K]).
 * Large batches (B >= 192, K <= 64) run scaled linear-domain sweeps that never write a
 * logarithm: var_x is formed on first read, and the log-domain lliks / lalpha / lbeta of the
 * requested windows are recomputed by the log-domain kernels (exact, same values as a
 * svihmm_forward_backward call on those windows).  That recomputation uses the CURRENT
 * observations / globals / emission parameters: read log-domain intermediates before the
 * next svihmm_set_* call (a read after one fails with an explicit error; ranges already
 * rebuilt stay readable). */
int svihmm_read_intermediate(svihmm_ctx* h, int32_t what, double* out);
/* nrows rows starting at flattened row row0 of the same [B*Lm, K] arrays. */
int svihmm_read_rows(svihmm_ctx* h, int32_t what, int64_t row0, int64_t nrows,
                     double* out);

/* ---- synthetic sequences generated in HBM (gen_synthetic.py:27-44 generate_data) ---------- */
/* Start in state 0; z_t drawn from row z_{t-1} of the transition matrix by np.random.choice's
 * inverse CDF (cdf[K,K] = cumsum(tran, axis=1) / cumsum[:, -1], searchsorted side='right');
 * x_t = means[z_t] + chols[z_t] n_t with n_t ~ N(0, I) (chols: lower Cholesky factors [K,D,D]).
 * Counter-based randomness (Philox4x32-10 keyed by `seed`; row t: stream 0 = transition uniform,
 * stream 1 + p = Box-Muller pair for components 2p, 2p+1), so a run is reproducible and the
 * state chain needs no sequential pass (composition of per-row maps).  The sequence becomes
 * the resident observation copy (no mask); K <= 64.  svihmm_read_generated copies the states
 * (int32[T]) and / or the observations ([T,D]) to the host. */
int svihmm_generate(svihmm_ctx* h, int64_t T, int32_t K, int32_t D, const double* cdf,
                    const double* means, const double* chols, uint64_t seed);
int svihmm_read_generated(svihmm_ctx* h, int32_t* sts_out, double* obs_out);

/* ---- a12: forward-filter backward-sample (hmm_fast.pyx:43-124) ---------------- */
/* Forward filter over the whole chain with the globals currently set (the host
 * passes the Cython variant's mod_init/ltran), then z[T-1] ~ softmax(lalpha[T-1]),
 * z[t] ~ softmax_k(lalpha[t,k] + log_tran_col[k, z[t+1]]) by inverse CDF
 * (rand_discrete, hmm_fast.pyx:29-36) with uniforms[t] in [0,1) supplied by the
 * caller (libc rand() streams are not reproducible on a device).
 * logA[K,K] = log(var_tran + DBL_EPSILON).  out_z[T] int64, out_lalpha[T,K] or NULL.
 * Long chains (K <= 64, T >= 2048) take no sequential pass over T: the forward filter runs as
 * the blocked scan of the full-chain E-step, the draws z[t] = F_t(z[t+1]) are composed as maps
 * over row chunks (same path as the sequential sampler unless a uniform lies within rounding of
 * a CDF step); out_lalpha is exact also for entries the scaled scan loses to underflow. */
int svihmm_ffbs(svihmm_ctx* h, const double* logA, const double* uniforms,
                uint32_t flags, int64_t* out_z, double* out_lalpha);

/* Backward sampling only, from forward messages the caller supplies: the `lalpha_init` branch of
 * FFBS (hmm_fast.pyx:80-95 skips the likelihoods and the filter, :97-122 samples).  lalpha[T,K]
 * host; logA / uniforms / out_z as svihmm_ffbs.  The resident observations are not used. */
int svihmm_ffbs_sample(svihmm_ctx* h, int64_t T, int32_t K, const double* lalpha,
                       const double* logA, const double* uniforms, int64_t* out_z);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------------- */
/* uid is a 128-byte ncclUniqueId produced on rank 0 and distributed by the host. */
int svihmm_comm_unique_id(char uid_out[128]);
int svihmm_comm_init(svihmm_ctx* h, const char uid[128], int32_t rank,
                     int32_t nranks);
int svihmm_comm_destroy(svihmm_ctx* h);
/* ncclCommCount of the handle's communicator (0 when none): the number of ranks RCCL itself
 * sees, reported by bench.py so that an N-GPU line can be told from N independent replicas. */
int svihmm_comm_count(svihmm_ctx* h, int32_t* nranks_out);
/* In-place ncclAllReduce(sum, double) of the packed statistics in HBM: the
 * "A_inter += A_i ; emit_inter[k] += e_i[k]" accumulation of
 * hmmsgd_metaobs.py:430-436 extended across ranks. */
int svihmm_allreduce_packed(svihmm_ctx* h);
/* The same exchange with the sum formed by the HOST (a communicator other than RCCL -- MPI, gloo --
 * or two handles on one device): svihmm_export_packed hands out exactly what the all-reduce puts on
 * the wire -- this rank's statistics in the callers' common coordinates (every rank's resident copy
 * has its own centre) -- and svihmm_import_packed takes the reduced vector back into the handle, from
 * where svihmm_read_packed continues as after svihmm_allreduce_packed (hmmsgd_metaobs.py:430-436). */
int svihmm_export_packed(svihmm_ctx* h, double* out_packed);
int svihmm_import_packed(svihmm_ctx* h, const double* packed_in);
/* Generic all-reduce of a small host vector (op 0: sum, 1: max) through HBM;
 * used for barriers and max-over-ranks timing. */
int svihmm_allreduce_host(svihmm_ctx* h, double* buf, int64_t n, int32_t op);

/* Measurement hooks (per-kernel HIP-event timing, kernel-generation selection for A/B runs and the
 * test suite, the MFMA operand-layout self-test) are NOT part of this boundary: include/svihmm_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* SVIHMM_H */
