// Stand-alone timing probe of k_emission_bf16x3 (fp32-mode emission): the bench shape (K = 64, D = 32,
// 3891 windows of 257 rows) on full-range random operands, HIP events around 20 launches.
//   make -C tools/probe emb_probe [EMB_KO=2|4|6] && tools/probe/emb_probe [K D B form]
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
  const int64_t T = 1000000, B = argc > 3 ? atoi(argv[3]) : T / Lm, n = B * Lm;
  const int form = argc > 4 ? atoi(argv[4]) : 0;      // 0: 256-row workgroups; 1: 128-row; 10 NH + RW: 32 RW rows, NH pair groups
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
  CKH(hipFuncSetAttribute((const void*)k_emission_bf16x3h<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * EMB_REC + 8192)));
  CKH(hipFuncSetAttribute((const void*)k_emission_bf16x3h<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * EMB_REC + 32768)));
  auto launch = [&]() {
    const unsigned g1 = (unsigned)((n + 127) / 128);
    if (form == 0)
      hipLaunchKernelGGL(k_emission_bf16x3<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, 0, dobs, (const uint8_t*)nullptr,
                         dst, n, Lm, D, K, (const char*)duw, 0x10000u, dEh, dkexp, dll0);
    else if (form == 1)
      hipLaunchKernelGGL(k_emission_bf16x3<1>, dim3(g1), dim3(256), (size_t)EMB_REC + 32768, 0, dobs, (const uint8_t*)nullptr,
                         dst, n, Lm, D, K, (const char*)duw, 0x10000u, dEh, dkexp, dll0);
    else {
      // form = 10 NH + RW: 32 RW rows per workgroup, NH pair groups
#define EMH(NHV, RWV) hipLaunchKernelGGL((k_emission_bf16x3h<NHV, RWV>), dim3((unsigned)((n + 32 * RWV - 1) / (32 * RWV))), dim3(64 * NHV * RWV), \
                                         (size_t)NHV * EMB_REC + (size_t)RWV * 8192, 0, dobs, (const uint8_t*)nullptr,  \
                                         dst, n, Lm, D, K, (const char*)duw, 0x10000u, dEh, dkexp, dll0)
      if (form == 21) EMH(2, 1); else if (form == 41) EMH(4, 1); else if (form == 81) EMH(8, 1);
      else if (form == 22) EMH(2, 2); else if (form == 42) EMH(4, 2); else if (form == 24) EMH(2, 4); else EMH(4, 4);
#undef EMH
    }
  };
  const int reps = B < 1000 ? 200 : 20;
  for (int i = 0; i < 3; ++i) launch();
  CKH(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  CKH(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)n * K * 6 * 2 * 32 * 24;
  printf("K=%d D=%d B=%lld form %d: %.4f ms per launch, %.1f TF/s bf16 (six-term products)\n", K, D, (long long)B, form, ms / reps,
         fl / (ms / reps * 1e-3) / 1e12);
  return 0;
}
