// Stand-alone timing probe of k_emission_bf16x3 (fp32-mode emission): the bench shape (K = 64, D = 32,
// 3891 windows of 257 rows) on full-range random operands, HIP events around 20 launches.
//   make -C tools/probe emb_probe [EMB_KO=2|4|6] && tools/probe/emb_probe [K D]
// (EMB_KO knocks parts of the kernel out for timing: 2 = no record copies, 4 = no step barrier;
// tools/probe/prof_emb.sh runs it under rocprofv3 with the SQ counters.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "../../include/svihmm.h"
#include "../../pysvihmm_amd/csrc/svihmm_common.h"
#include "../../pysvihmm_amd/csrc/device_helpers.h"
#include "../../pysvihmm_amd/csrc/kernels_emission.h"
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 64, D = argc > 2 ? atoi(argv[2]) : 32, Lm = 257;
  const int64_t T = 1000000, B = T / Lm, n = B * Lm;
  std::vector<double> obs((size_t)T * D);
  for (auto& v : obs) v = (rand() / (double)RAND_MAX - 0.5) * 8.0;
  std::vector<int64_t> starts(B);
  for (int64_t b = 0; b < B; ++b) starts[b] = b * Lm;
  const size_t nb = (size_t)EMB_NREC * EMB_REC;
  std::vector<uint16_t> uwh(nb / 2);
  for (auto& v : uwh) v = (uint16_t)(0x3c00 + (rand() & 0x3ff));   // bf16 ~ 0.008 .. 0.03
  double *dobs, *dkexp, *dll0; int64_t* dst; char* duw; float* dEh;
  CKH(hipMalloc(&dobs, obs.size() * 8)); CKH(hipMalloc(&dst, B * 8)); CKH(hipMalloc(&duw, nb));
  CKH(hipMalloc(&dEh, (size_t)n * K * 4)); CKH(hipMalloc(&dkexp, n * 8)); CKH(hipMalloc(&dll0, B * K * 8));
  CKH(hipMemcpy(dobs, obs.data(), obs.size() * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dst, starts.data(), B * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(duw, uwh.data(), nb, hipMemcpyHostToDevice));
  const size_t lds = (size_t)EMB_REC + (size_t)4 * 64 * 64 * 4;
  CKH(hipFuncSetAttribute((const void*)k_emission_bf16x3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    hipLaunchKernelGGL(k_emission_bf16x3<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, 0, dobs, (const uint8_t*)nullptr,
                       dst, n, Lm, D, K, (const char*)duw, 0x10000u, dEh, dkexp, dll0);
  };
  for (int i = 0; i < 3; ++i) launch();
  CKH(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(e1);
  CKH(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)n * K * 6 * 2 * 32 * 24;
  printf("K=%d D=%d: %.4f ms per launch, %.1f TF/s bf16 (six-term products)\n", K, D, ms / 20, fl / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
