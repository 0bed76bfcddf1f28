// Stand-alone timing of the fp64 orbit emission kernels at minibatch sizes (K = 64, D = 32, B windows of 257
// rows): the 64-row form k_emission_orbit<4, 4, 1>, the 128-row form <4, 4, 2> and the k-split 16-row form
// k_emission_orbit_ks<4>.  HIP events around REPS launches; tools only.
//   hipcc ... [-DEMO_KO=<bits>] -o emo_probe emo_probe.hip ; emo_probe [B [LDS KB of the k-split kernel]]
// EMO_KO (k_emission_orbit_ks only): 1 = no k-steps, 2 = no theta loads, 4 = no exp in the epilogue.
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
  const int K = 64, D = 32, Lm = 257, NT = 4;
  const int64_t B = argc > 1 ? atoi(argv[1]) : 64, n = B * Lm, T = 1000000;
  const size_t ldspad = argc > 2 ? (size_t)atoi(argv[2]) * 1024 : 0;     // LDS request of the k-split kernel (occupancy cap)
  std::vector<double> obs((size_t)T * D);
  for (auto& v : obs) v = (rand() / (double)RAND_MAX - 0.5) * 2.0;
  std::vector<int64_t> starts(B);
  for (int64_t b = 0; b < B; ++b) starts[b] = (b * 7919 * Lm) % (T - Lm);
  const int c = D >> 2, nd = (D >> 1) + 1, nks = c * nd + ((nd + 3) >> 2), LEN = D + (D >> 1) + 1;
  std::vector<double> orb((size_t)nks * 4 * K);
  for (auto& v : orb) v = -(rand() / (double)RAND_MAX) * 0.01;
  double *dobs, *dorb, *dll, *dkexp, *dll0; int64_t* dst;
  CKH(hipMalloc(&dobs, obs.size() * 8)); CKH(hipMalloc(&dst, B * 8)); CKH(hipMalloc(&dorb, orb.size() * 8 + 65536));
  CKH(hipMalloc(&dll, (size_t)n * K * 8)); CKH(hipMalloc(&dkexp, n * 8)); CKH(hipMalloc(&dll0, B * K * 8));
  CKH(hipMemcpy(dobs, obs.data(), obs.size() * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dst, starts.data(), B * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dorb, orb.data(), orb.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  for (int which = 0; which < 3; ++which) {
    auto launch = [&]() {
      if (which == 0) {
        const size_t lds = (size_t)64 * LEN * 8 + 64 * 9;
        hipLaunchKernelGGL((k_emission_orbit<4, 4, 1>), dim3((unsigned)((n + 63) / 64)), dim3(256), lds, 0, dobs, (const uint8_t*)nullptr,
                           dst, n, Lm, D, K, dorb, 0u, dll, dkexp, dll0, (int64_t*)nullptr, 0);
      } else if (which == 1) {
        const size_t lds = (size_t)128 * LEN * 8 + 128 * 9;
        hipLaunchKernelGGL((k_emission_orbit<4, 4, 2>), dim3((unsigned)((n + 127) / 128)), dim3(256), lds, 0, dobs, (const uint8_t*)nullptr,
                           dst, n, Lm, D, K, dorb, 0u, dll, dkexp, dll0, (int64_t*)nullptr, 0);
      } else {
        size_t lds = (size_t)(3 * NT * 256 + 16 * NT + 16) * 8 + 16;
        if (ldspad > lds) lds = ldspad;
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_emission_orbit_ks<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_emission_orbit_ks<4>), dim3((unsigned)((n + 15) / 16)), dim3(256), lds, 0, dobs, (const uint8_t*)nullptr,
                           dst, n, Lm, D, K, dorb, 0u, dll, dkexp, dll0, (int64_t*)nullptr, 0);
      }
    };
    for (int i = 0; i < 5; ++i) launch();
    CKH(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    CKH(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)n * K * 2.0 * 4 * nks;
    printf("%s B=%lld (%lld rows): %.2f us per launch, %.1f TF/s (back-to-back launches, KO=%d)\n",
           which == 0 ? "k_emission_orbit<4,4,1>" : which == 1 ? "k_emission_orbit<4,4,2>" : "k_emission_orbit_ks<4>",
           (long long)B, (long long)n, 1e3 * ms / reps, fl / (ms / reps * 1e-3) / 1e12, (int)EMO_KO);
  }
  return 0;
}
