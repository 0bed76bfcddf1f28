// Stand-alone timing of the minibatch sweep kernels (k_wave_linr, k_wave_lin4) at the S = 64 shape:
// B windows of Lm rows, K = 64, random scaled inputs.  HIP events around REPS launches; tools only.
//   hipcc ... [-DWLR_KO=<bits>] -o wlr_probe wlr_probe.hip ; wlr_probe [B Lm which]
// WLR_KO (k_wave_linr only): 1 = one DPP FMA per accumulator instead of 16, 2 = exponent fixed at 0,
// 4 = no message stores, 8 = no Eh loads in the loop; add-ons: 16 / 32 = an agent-scope release + progress store
// every 32 / 16 steps (what publishing the sweep's progress to a consumer kernel would cost).
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
#include "../../pysvihmm_amd/csrc/kernels_recursion.h"
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename ST>
int run(int B, int Lm, int which, int reps) {
  const int K = 64;
  const size_t n = (size_t)B * Lm;
  std::vector<ST> Eh(n * K);
  for (auto& v : Eh) v = (ST)(0.05 + 0.95 * rand() / (double)RAND_MAX);
  std::vector<double> A(K * K), AT(K * K), kexp(n, 0.0), mi(K, -4.0), l0((size_t)B * K, -1.0);
  for (int i = 0; i < K; ++i) { double s = 0; for (int j = 0; j < K; ++j) { A[i * K + j] = rand() / (double)RAND_MAX; s += A[i * K + j]; }
    for (int j = 0; j < K; ++j) { A[i * K + j] /= s; AT[j * K + i] = A[i * K + j]; } }
  ST *dE, *da, *db; double *dA, *dAT, *dk, *dmi, *dl0, *dhx, *dgx, *dlb, *dlz; double2* dzf;
  CKH(hipMalloc(&dE, n * K * sizeof(ST))); CKH(hipMalloc(&da, n * K * sizeof(ST))); CKH(hipMalloc(&db, n * K * sizeof(ST)));
  CKH(hipMalloc(&dA, K * K * 8)); CKH(hipMalloc(&dAT, K * K * 8)); CKH(hipMalloc(&dk, n * 8)); CKH(hipMalloc(&dmi, K * 8));
  CKH(hipMalloc(&dl0, (size_t)B * K * 8)); CKH(hipMalloc(&dhx, n * 8)); CKH(hipMalloc(&dgx, n * 8));
  CKH(hipMalloc(&dlb, B * 8)); CKH(hipMalloc(&dlz, B * 8)); CKH(hipMalloc(&dzf, B * 16));
  CKH(hipMemcpy(dE, Eh.data(), n * K * sizeof(ST), hipMemcpyHostToDevice));
  CKH(hipMemcpy(dA, A.data(), K * K * 8, hipMemcpyHostToDevice)); CKH(hipMemcpy(dAT, AT.data(), K * K * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dk, kexp.data(), n * 8, hipMemcpyHostToDevice)); CKH(hipMemcpy(dmi, mi.data(), K * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dl0, l0.data(), (size_t)B * K * 8, hipMemcpyHostToDevice));
  auto launch = [&]() {
    if (which == 0)
      hipLaunchKernelGGL((k_wave_linr<ST, ST>), dim3(B, 2), dim3(64), 0, 0, dE, dk, dA, dAT, dmi, dl0, (size_t)K, Lm, K, da, db, dhx, dgx, dlb, dlz, dzf);
    else
      hipLaunchKernelGGL((k_wave_lin4<64, ST>), dim3(B, 2), dim3(256), 0, 0, dE, dk, dA, dAT, dmi, dl0, (size_t)K, Lm, K, da, db, dhx, dgx, dlb, dlz, dzf);
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  CKH(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  CKH(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> lb(B);
  CKH(hipMemcpy(lb.data(), dlb, B * 8, hipMemcpyDeviceToHost));
  printf("%s %s B=%d Lm=%d: %.2f us per launch = %.1f ns per step (back-to-back launches; lb[0] = %.6f)\n",
         which == 0 ? "k_wave_linr" : "k_wave_lin4", sizeof(ST) == 4 ? "f32" : "f64", B, Lm, 1e3 * ms / reps,
         1e6 * ms / reps / (Lm - 1), lb[0]);
  return 0;
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, Lm = argc > 2 ? atoi(argv[2]) : 257;
  for (int which = 0; which < 2; ++which) {
    if (run<double>(B, Lm, which, 200)) return 1;
    if (run<float>(B, Lm, which, 200)) return 1;        // (k_wave_linr<float>: fp32 arithmetic, not used by the product)
  }
  return 0;
}
