// probe.hip -- tools-only micro-benchmarks (NOT part of libsvihmm_hip.so or its public header):
// fp64 MFMA / VALU throughput calibration for the roofline and the HBM access-pattern probe of
// the sweeps.  Built by tools/probe/Makefile into tools/probe/libsvihmm_probe.so and driven by
// tools/peak_probe.py / tools/pattern_probe.py.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CK(x) do { if (int r__ = (x)) return r__; } while (0)
struct Scratch {
  void* p = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) hipFree(p);
    p = nullptr; cap = 0;
    HIPCK(hipMalloc(&p, bytes));
    cap = bytes;
    return 0;
  }
  ~Scratch() { if (p) hipFree(p); }
};

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


// HBM access-pattern probe for the sweeps: every workgroup walks 16 windows in time, reading
// two [16 x 64] fp64 row sets per step and writing two (the traffic of a forward + backward
// pair), PERM 0: window-major rows (16 pieces of 512 B, Lm * 512 B apart), PERM 1: the 16
// windows' rows of a step adjacent (one 8 KB piece).
template <int PERM>
__global__ __launch_bounds__(256) void k_probe_pattern(const double* __restrict__ src,
                                                       double* __restrict__ dst, int Lm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gbase = (size_t)blockIdx.x * 16 * Lm * 64;
  double acc = 0.0;
  for (int t = 0; t < Lm; ++t) {
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = wave * 4 + r;
      const size_t o = PERM ? gbase + ((size_t)t * 16 + w) * 64 + lane
                            : gbase + ((size_t)w * Lm + t) * 64 + lane;
      v[r] = src[o];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = wave * 4 + r;
      const size_t o = PERM ? gbase + ((size_t)t * 16 + w) * 64 + lane
                            : gbase + ((size_t)w * Lm + t) * 64 + lane;
      dst[o] = v[r] + acc;
      acc += 1e-300;
    }
  }
}


extern "C" {
// which 0: v_mfma_f64_16x16x4_f64, 1: v_fma_f64.  Returns achieved TFLOP/s.
int probe_fp64(int device, int32_t which, double* tflops_out) {
  if (!tflops_out) return 1;
  HIPCK(hipSetDevice(device));
  Scratch scr; hipStream_t stream = nullptr;
  // which: 0 mfma (8 blocks/CU), 1 fma, 2 mfma 1 block/CU (1 wave/SIMD), 3 mfma 2 blocks/CU;
  // +16: return s_memtime ticks per loop iteration of block 0 instead of TFLOP/s
  const bool ticks = (which & 16) != 0;
  if (which >= 400 && which <= 401) {   // sweep access-pattern probe: returns TB/s (read + write)
    const int Lm = 257, groups = 488;
    const size_t n = (size_t)groups * 16 * Lm * 64;
    CK(scr.ensure(2 * n * sizeof(double)));
    double* src = (double*)scr.p;
    double* dst = src + n;
    HIPCK(hipMemsetAsync(src, 0, n * sizeof(double), stream));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      HIPCK(hipEventRecord(e0, stream));
      if (which == 400) hipLaunchKernelGGL(k_probe_pattern<0>, dim3(groups), dim3(256), 0, stream, (const double*)src, dst, Lm);
      else hipLaunchKernelGGL(k_probe_pattern<1>, dim3(groups), dim3(256), 0, stream, (const double*)src, dst, Lm);
      HIPCK(hipEventRecord(e1, stream));
      HIPCK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *tflops_out = 2.0 * n * sizeof(double) / (ms * 1e-3) / 1e12;
    return 0;
  }
  if (which >= 200) {   // 200 + 10*nf + mf : mix probe, 2 blocks/CU; returns ns per loop iteration
    const int nf = (which - 200) / 10, mf = (which - 200) % 10;
    const int blocks = 512, threads = 256, iters = 4000;
    CK(scr.ensure(((size_t)blocks * threads + 8) * sizeof(double)));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      HIPCK(hipEventRecord(e0, stream));
#define PM(N, M) hipLaunchKernelGGL((k_peak_mix<N, M>), dim3(blocks), dim3(threads), 0, stream, (double*)scr.p, iters)
      if (nf == 0) PM(0, true);
      else if (nf == 8) { if (mf) PM(8, true); else PM(8, false); }
      else { if (mf) PM(16, true); else PM(16, false); }
#undef PM
      HIPCK(hipEventRecord(e1, stream));
      HIPCK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *tflops_out = (double)ms * 1e6 / iters;
    return 0;
  }
  if (which >= 100) {   // 100 + nacc: 1 block/CU (1 wave/SIMD), nacc independent accumulators
    const int nacc = which - 100, blocks = 256, threads = 256, iters = 4000;
    CK(scr.ensure(((size_t)blocks * threads + 8) * sizeof(double)));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      HIPCK(hipEventRecord(e0, stream));
#define PK(N) hipLaunchKernelGGL(k_peak_mfma_chain<N>, dim3(blocks), dim3(threads), 0, stream, (double*)scr.p, iters)
      if (nacc == 1) PK(1); else if (nacc == 2) PK(2); else if (nacc == 4) PK(4); else if (nacc == 8) PK(8);
      else if (nacc == 12) PK(12); else PK(16);
#undef PK
      HIPCK(hipEventRecord(e1, stream));
      HIPCK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    // nanoseconds per MFMA per SIMD
    *tflops_out = (double)ms * 1e6 / ((double)iters * (nacc == 1 || nacc == 2 || nacc == 4 || nacc == 8 || nacc == 12 ? nacc : 16));
    return 0;
  }
  which &= 15;
  const int bpc = which == 2 ? 1 : which == 3 ? 2 : 8;
  const int blocks = 256 * bpc, threads = 256, iters = 20000;
  CK(scr.ensure(((size_t)blocks * threads + 8) * sizeof(double)));
  hipEvent_t e0, e1;
  HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCK(hipEventRecord(e0, stream));
    if (which != 1)
      hipLaunchKernelGGL(k_peak_mfma_f64, dim3(blocks), dim3(threads), 0, stream, (double*)scr.p, iters);
    else
      hipLaunchKernelGGL(k_peak_fma_f64, dim3(blocks), dim3(threads), 0, stream, (double*)scr.p, iters);
    HIPCK(hipEventRecord(e1, stream));
    HIPCK(hipEventSynchronize(e1));
  }
  float ms = 0.f;
  HIPCK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  double flops;
  if (which != 1) flops = (double)blocks * (threads / 64) * (double)iters * 4.0 * 2048.0;
  else flops = (double)blocks * threads * (double)iters * 8.0 * 2.0;
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  if (ticks && which != 1) {
    double tk = 0;
    HIPCK(hipMemcpy(&tk, (double*)scr.p + (size_t)blocks * threads, sizeof(double), hipMemcpyDeviceToHost));
    *tflops_out = tk / iters;
  }
  return 0;
}

}  // extern "C"
