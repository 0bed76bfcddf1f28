// issue_probe.hip -- instruction issue rates of ONE wavefront on a SIMD (tools only, round 5):
// what a latency-bound single-wave recursion step (k_wave_linr / k_wave_lin4) can count on.
// Each test runs a block of NI instructions ITERS times in one wave per workgroup (one workgroup per
// CU is launched, 64 threads) and reports ns per instruction from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 20000;

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)

// 64 instructions per iteration, four independent accumulator chains
__global__ __launch_bounds__(64) void t_fmac_f32_dpp(float* o) {
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, p = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_fmac_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f32_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f32_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f32_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p), "v"(c));)
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void t_fmac_f32(float* o) {
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, p = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p), "v"(c));)
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void t_fmac_f64_dpp(double* o) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, p = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p), "v"(c));)
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(64) void t_fmac_f64(double* o) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, p = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p), "v"(c));)
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// eight chains of fp64 FMA
__global__ __launch_bounds__(64) void t_fmac_f64_8(double* o) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, p = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP4(REP4(asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p), "v"(c));
              asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5"
                           : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(c));))
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void t_pk_fma_f32(float* o) {
  f2 a0 = {0, 0}, a1 = a0, a2 = a0, a3 = a0, p = {o[threadIdx.x], o[threadIdx.x + 1]}, c = {o[64 + threadIdx.x], 1.0f};
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p), "v"(c));)
  }
  o[128 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y;
}
__global__ __launch_bounds__(64) void t_mov_dpp(float* o) {
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, p = o[threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p));)
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// v_readlane + v_fma with an SGPR operand (the scalar-broadcast form of a mat-vec term): 2 + 1 per term
__global__ __launch_bounds__(64) void t_readlane_fma_f64(double* o) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, p = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(asm volatile("v_readlane_b32 s20, %4, 3\n\tv_readlane_b32 s21, %5, 3\n\t"
                       "v_fma_f64 %0, s[20:21], %6, %0\n\tv_fma_f64 %1, s[20:21], %6, %1\n\t"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                       : "v"(__double2loint(p)), "v"(__double2hiint(p)), "v"(c) : "s20", "s21");)
  }
  o[128 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// dependent chain of fp64 FMAs (latency)
__global__ __launch_bounds__(64) void t_dep_f64(double* o) {
  double a0 = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(REP4(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(c));))
  }
  o[128 + threadIdx.x] = a0;
}
__global__ __launch_bounds__(64) void t_dep_f32(float* o) {
  float a0 = o[threadIdx.x], c = o[64 + threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    REP16(REP4(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(c));))
  }
  o[128 + threadIdx.x] = a0;
}
// LDS round trip: write, (barrier among 4 waves when NW = 4), read
template <int NW>
__global__ __launch_bounds__(64 * NW) void t_lds_rt(double* o) {
  __shared__ double s[2][256];
  double v = o[threadIdx.x];
  for (int i = 0; i < ITERS; ++i) {
    s[i & 1][threadIdx.x] = v;
    __syncthreads();
    v = s[i & 1][(threadIdx.x + 17) & (64 * NW - 1)] + 1.0;
  }
  o[256 + threadIdx.x] = v;
}
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void t_swap(float* o) {
  unsigned x = threadIdx.x, y = threadIdx.x * 3;
  for (int i = 0; i < ITERS; ++i) {
    REP16(REP4(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));))
  }
  o[128 + threadIdx.x] = (float)(x + y);
}

// grid-wide barrier across one workgroup per CU (what a single cooperative kernel per SVI iteration would
// pay between its phases): sense-reversing counter in HBM/L2, agent-scope atomics, thread 0 of every
// workgroup arrives and spins, the rest wait at s_barrier
__global__ __launch_bounds__(256) void t_grid_barrier(unsigned* ctr, int nwg, int iters, double* o) {
  unsigned target = 0;
  double v = o[threadIdx.x];
  for (int i = 0; i < iters; ++i) {
    target += (unsigned)nwg;
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    v += 1.0;
  }
  o[256 + threadIdx.x] = v;
}
static double run_grid_barrier(int nwg, int nthreads) {
  unsigned* ctr; double* d;
  CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&d, 1 << 16));
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(ctr, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(t_grid_barrier, dim3(nwg), dim3(nthreads), 0, 0, ctr, nwg, iters, d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return 1e6 * best / iters;
}

template <typename F, typename T>
static double run(F kern, T* buf, int nthreads, int ninstr) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(nthreads), 0, 0, buf);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(256), dim3(nthreads), 0, 0, buf);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e6 * ms / ((double)ITERS * ninstr);
}

int main() {
  double* d; CK(hipMalloc(&d, 1 << 16)); CK(hipMemset(d, 0, 1 << 16));
  float* f = (float*)d;
  printf("ns per instruction, one wave per SIMD (one 64-thread workgroup per CU), 4 independent chains unless noted\n");
  printf("v_fmac_f32            %.3f\n", run(t_fmac_f32, f, 64, 64));
  printf("v_fmac_f32_dpp        %.3f\n", run(t_fmac_f32_dpp, f, 64, 64));
  printf("v_fmac_f64            %.3f\n", run(t_fmac_f64, d, 64, 64));
  printf("v_fmac_f64 (8 chains) %.3f\n", run(t_fmac_f64_8, d, 64, 128));
  printf("v_fmac_f64_dpp        %.3f\n", run(t_fmac_f64_dpp, d, 64, 64));
  printf("v_pk_fma_f32          %.3f\n", run(t_pk_fma_f32, f, 64, 64));
  printf("v_mov_b32_dpp         %.3f\n", run(t_mov_dpp, f, 64, 64));
  printf("2 readlane + 2 fma_f64 with SGPR source (per group of 4)  %.3f\n", run(t_readlane_fma_f64, d, 64, 16));
  printf("dependent v_fma_f64   %.3f\n", run(t_dep_f64, d, 64, 64));
  printf("dependent v_fma_f32   %.3f\n", run(t_dep_f32, f, 64, 64));
  printf("v_permlane32_swap     %.3f\n", run(t_swap, f, 64, 64));
  printf("LDS write + barrier + read, 1 wave   %.1f ns per round trip\n", run(t_lds_rt<1>, d, 64, 1));
  printf("LDS write + barrier + read, 4 waves  %.1f ns per round trip\n", run(t_lds_rt<4>, d, 256, 1));
  printf("grid barrier (counter in L2, one 256-thread workgroup per CU):  64 workgroups %.0f ns, 128: %.0f ns, 256: %.0f ns\n",
         run_grid_barrier(64, 256), run_grid_barrier(128, 256), run_grid_barrier(256, 256));
  return 0;
}
