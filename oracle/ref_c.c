/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked or loaded by the product.
 *
 * Plain-C restatement of the reference's E-step (dillonalaird/pysvihmm), single
 * threaded like the reference, same algorithm: pairwise log-add-exp folds with
 * K^2 exp/log1p per time step (np.logaddexp.reduce), one state / one window at a
 * time.  Used (a) as a fast checker for the HIP path at sizes where the NumPy
 * oracle is too slow and (b) as bench.py's cpu_baseline ("port": the reference
 * itself is Python 2 + an absent third-party package and cannot run on the GPU
 * box).  Pinned to the NumPy oracle / the reference's golden vectors by
 * tests/test_oracle_golden.py.  Emission arithmetic: pybasicbayes' published
 * algorithm (parity unpinned, see oracle/ref_numpy.py).
 *
 * Each function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double logaddexp(double a, double b) {
  /* numpy npy_logaddexp */
  if (a == b) return a + 0.6931471805599453;
  double d = a - b;
  if (d > 0) return a + log1p(exp(-d));
  if (d <= 0) return b + log1p(exp(d));
  return a + b; /* NaN */
}

static double digamma_c(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12 + f * (1.0 / 120 + f * (-1.0 / 252 + f * (1.0 / 240 +
             f * (-1.0 / 132 + f * (691.0 / 32760 + f * (-1.0 / 12)))))));
  return r + log(x) - 0.5 / x + t;
}

double orc_digamma(double x) { return digamma_c(x); }

static double nan_to_num(double v) {
  if (v != v) return 0.0;
  if (isinf(v)) return v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
  return v;
}

/* a3: lliks[:,k] = nan_to_num(odist.expected_log_likelihood(obs_rows))
 * hmmbase.py:219-220; arithmetic = pybasicbayes Gaussian.expected_log_likelihood
 * (Cholesky solve of the centred rows).  Returns 0 ok, 1 not PD. */
int orc_lliks_niw(const double* x, int64_t n, int D, int K, const double* mu,
                  const double* sigma, const double* kappa, const double* nu,
                  double* out) {
  double* L = (double*)malloc(sizeof(double) * D * D);
  double* y = (double*)malloc(sizeof(double) * D);
  for (int k = 0; k < K; ++k) {
    const double* S = sigma + (size_t)k * D * D;
    const double* m = mu + (size_t)k * D;
    memcpy(L, S, sizeof(double) * D * D);
    for (int j = 0; j < D; ++j) {
      double d = L[j * D + j];
      for (int c = 0; c < j; ++c) d -= L[j * D + c] * L[j * D + c];
      if (!(d > 0)) { free(L); free(y); return 1; }
      d = sqrt(d);
      L[j * D + j] = d;
      for (int i = j + 1; i < D; ++i) {
        double s = L[i * D + j];
        for (int c = 0; c < j; ++c) s -= L[i * D + c] * L[j * D + c];
        L[i * D + j] = s / d;
      }
    }
    double llt = D * log(2.0);
    for (int i = 0; i < D; ++i) llt += digamma_c(0.5 * (nu[k] - i)) - 2.0 * log(L[i * D + i]);
    const double cst = 0.5 * llt - D / (2.0 * kappa[k]) - 0.5 * D * log(2.0 * M_PI);
    for (int64_t t = 0; t < n; ++t) {
      const double* xt = x + (size_t)t * D;
      double q = 0.0;
      for (int i = 0; i < D; ++i) {
        double s = xt[i] - m[i];
        for (int c = 0; c < i; ++c) s -= L[i * D + c] * y[c];
        y[i] = s / L[i * D + i];
        q += y[i] * y[i];
      }
      out[(size_t)t * K + k] = nan_to_num(cst - 0.5 * nu[k] * q);
    }
  }
  free(L); free(y);
  return 0;
}

/* a4: hmmbase.py:292-295 == hmmsgd_metaobs.py:800-803
 * lalpha[t] = np.logaddexp.reduce(lalpha[t-1] + ltran.T, axis=1) + ll[t] */
void orc_forward(const double* ll, const double* mod_init, const double* ltran, int64_t T,
                 int K, double* la) {
  for (int j = 0; j < K; ++j) la[j] = mod_init[j] + ll[j];
  for (int64_t t = 1; t < T; ++t) {
    const double* p = la + (size_t)(t - 1) * K;
    double* o = la + (size_t)t * K;
    for (int j = 0; j < K; ++j) {
      double acc = p[0] + ltran[j];
      for (int i = 1; i < K; ++i) acc = logaddexp(acc, p[i] + ltran[(size_t)i * K + j]);
      o[j] = acc + ll[(size_t)t * K + j];
    }
  }
}

/* a5: hmmbase.py:316-320 == hmmsgd_metaobs.py:851-855
 * lbeta[t] = np.logaddexp.reduce(ltran + lbeta[t+1] + ll[t+1], axis=1) */
void orc_backward(const double* ll, const double* ltran, int64_t T, int K, double* lb) {
  for (int j = 0; j < K; ++j) lb[(size_t)(T - 1) * K + j] = 0.0;
  for (int64_t t = T - 2; t >= 0; --t) {
    const double* n = lb + (size_t)(t + 1) * K;
    const double* l = ll + (size_t)(t + 1) * K;
    double* o = lb + (size_t)t * K;
    for (int i = 0; i < K; ++i) {
      double acc = (ltran[(size_t)i * K] + n[0]) + l[0];
      for (int j = 1; j < K; ++j) acc = logaddexp(acc, (ltran[(size_t)i * K + j] + n[j]) + l[j]);
      o[i] = acc;
    }
  }
}

/* a6: hmmbase.py:226-229;  a7: hmmsgd_metaobs.py:271 (sum over all t, quirk Q4) */
double orc_posterior(const double* la, const double* lb, int64_t T, int K, double* q) {
  double lbsum = 0.0;
  for (int64_t t = 0; t < T; ++t) {
    const double* a = la + (size_t)t * K;
    const double* b = lb + (size_t)t * K;
    double* o = q + (size_t)t * K;
    double m = -INFINITY;
    for (int k = 0; k < K; ++k) { o[k] = a[k] + b[k]; if (o[k] > m) m = o[k]; }
    double s = 0.0;
    for (int k = 0; k < K; ++k) { o[k] = exp(o[k] - m); s += o[k]; }
    for (int k = 0; k < K; ++k) o[k] /= s;
    double acc = a[0];
    for (int k = 1; k < K; ++k) acc = logaddexp(acc, a[k]);
    lbsum += acc;
  }
  return lbsum;
}

/* a8: hmmsgd_metaobs.py:876-904 + util.py:73-83 (accumulating).  wrap: quirk Q1. */
void orc_suffstats(const double* q, const double* x, const uint8_t* mask, int64_t T, int D,
                   int K, int wrap, double* A, double* xbar, double* neff, double* S) {
  for (int64_t t = 0; t < T; ++t) {
    const double* qp;
    if (t > 0) qp = q + (size_t)(t - 1) * K;
    else if (wrap) qp = q + (size_t)(T - 1) * K;
    else qp = NULL;
    const double* qt = q + (size_t)t * K;
    if (qp)
      for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) A[(size_t)i * K + j] += qp[i] * qt[j];
    if (mask && mask[t]) continue;
    const double* xt = x + (size_t)t * D;
    for (int k = 0; k < K; ++k) {
      const double w = qt[k];
      neff[k] += w;
      double* Sk = S + (size_t)k * D * D;
      for (int a = 0; a < D; ++a) {
        const double wa = w * xt[a];
        xbar[(size_t)k * D + a] += wa;
        for (int b = 0; b < D; ++b) Sk[a * D + b] += xt[b] * wa;
      }
    }
  }
}

/* a3..a9: the minibatch loop hmmsgd_metaobs.py:405-436.
 * packed = [A_raw K*K | xbar K*D | neff K | S K*D*D | lb]; flags as svihmm.h. */
int orc_estep_minibatch(const double* obs, const uint8_t* mask, int64_t T, int D,
                        const int64_t* starts, int B, int Lm, int K, const double* mod_init,
                        const double* ltran, const double* mu, const double* sigma,
                        const double* kappa, const double* nu, unsigned flags,
                        double* packed) {
  const size_t n = (size_t)Lm * K;
  double* ll = (double*)malloc(sizeof(double) * n * 4);
  double* xw = (double*)malloc(sizeof(double) * (size_t)Lm * D);
  if (!ll || !xw) return 2;
  double *la = ll + n, *lb = la + n, *q = lb + n;
  double* A = packed;
  double* xbar = A + (size_t)K * K;
  double* neff = xbar + (size_t)K * D;
  double* S = neff + K;
  double* lbt = S + (size_t)K * D * D;
  memset(packed, 0, sizeof(double) * ((size_t)K * K + (size_t)K * D + K + (size_t)K * D * D + 1));
  for (int b = 0; b < B; ++b) {
    const int64_t s0 = starts[b];
    if (s0 < 0 || s0 + Lm > T) { free(ll); free(xw); return 3; }
    memcpy(xw, obs + (size_t)s0 * D, sizeof(double) * (size_t)Lm * D);
    if ((flags & 1u) && mask)
      for (int t = 0; t < Lm; ++t)
        if (mask[s0 + t]) for (int d = 0; d < D; ++d) xw[(size_t)t * D + d] = NAN;
    if (orc_lliks_niw(xw, Lm, D, K, mu, sigma, kappa, nu, ll)) { free(ll); free(xw); return 1; }
    orc_forward(ll, mod_init, ltran, Lm, K, la);
    orc_backward(ll, ltran, Lm, K, lb);
    *lbt += orc_posterior(la, lb, Lm, K, q);
    orc_suffstats(q, obs + (size_t)s0 * D, mask ? mask + s0 : NULL, Lm, D, K,
                  (flags & 2u) ? 1 : 0, A, xbar, neff, S);
  }
  free(ll); free(xw);
  return 0;
}

/* The same minibatch loop with the windows dealt in contiguous slices to `nthreads` OpenMP
 * threads (each slice is the serial loop above on a private accumulator; the slices are added
 * in slice order).  The reference itself is single-threaded: this exists only as bench.py's
 * "all host cores" baseline (SURVEY.md 8d item ii). */
int orc_estep_minibatch_mt(const double* obs, const uint8_t* mask, int64_t T, int D,
                           const int64_t* starts, int B, int Lm, int K, const double* mod_init,
                           const double* ltran, const double* mu, const double* sigma,
                           const double* kappa, const double* nu, unsigned flags, int nthreads,
                           double* packed) {
  const size_t np_ = (size_t)K * K + (size_t)K * D + K + (size_t)K * D * D + 1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > B) nthreads = B > 0 ? B : 1;
  double* parts = (double*)malloc(sizeof(double) * np_ * (size_t)nthreads);
  int* rcs = (int*)calloc((size_t)nthreads, sizeof(int));
  if (!parts || !rcs) { free(parts); free(rcs); return 2; }
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
  for (int th = 0; th < nthreads; ++th) {
    const int b0 = (int)((int64_t)B * th / nthreads), b1 = (int)((int64_t)B * (th + 1) / nthreads);
    rcs[th] = orc_estep_minibatch(obs, mask, T, D, starts + b0, b1 - b0, Lm, K, mod_init, ltran, mu,
                                  sigma, kappa, nu, flags, parts + np_ * (size_t)th);
  }
  int rc = 0;
  memset(packed, 0, sizeof(double) * np_);
  for (int th = 0; th < nthreads; ++th) {
    if (rcs[th]) rc = rcs[th];
    for (size_t i = 0; i < np_; ++i) packed[i] += parts[np_ * (size_t)th + i];
  }
  free(parts); free(rcs);
  return rc;
}
