// kernels_misc.h -- exp/transpose of ltran, MFMA layout self-test, fp64 peak probes.
// Part of libsvihmm_hip.so; included by svihmm_hip.hip (single translation unit).
#pragma once

// small utility kernels
__global__ void k_exp_transpose(const double* __restrict__ ltran, int K, double* __restrict__ A,
                                double* __restrict__ AT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * K) return;
  const int i = idx / K, j = idx - i * K;
  const double v = exp(ltran[idx]);
  A[idx] = v;
  AT[(size_t)j * K + i] = v;
}

__global__ void k_selftest_mfma(const double* __restrict__ A, const double* __restrict__ Bm,
                                double* __restrict__ C) {
  // A[16][4], B[4][16] row-major -> C[16][16]
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = Bm[(l >> 4) * 16 + (l & 15)];
  double4_t c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

// fp64 throughput micro-benchmarks (peak calibration for the roofline)
__global__ __launch_bounds__(256) void k_peak_mfma_f64(double* out, int iters) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) out[(size_t)gridDim.x * blockDim.x] = (double)(t1 - t0);
}
template <int NACC>
__global__ __launch_bounds__(256) void k_peak_mfma_chain(double* out, int iters) {
  double4_t c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = (double4_t){0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i][i & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[(size_t)gridDim.x * blockDim.x] = (double)(t1 - t0);
}
// MFMA + fp64 VALU overlap probe: per iteration 8 MFMAs and NF*8 independent v_fma_f64
template <int NF, bool MF>
__global__ __launch_bounds__(256) void k_peak_mix(double* out, int iters) {
  double4_t c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = (double4_t){0, 0, 0, 0};
  double f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = i;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MF) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NF; ++k) f[(i + k) & 7] = fma(f[(i + k) & 7], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i][i & 3] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_peak_fma_f64(double* out, int iters) {
  double c[8];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = fma(c[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
