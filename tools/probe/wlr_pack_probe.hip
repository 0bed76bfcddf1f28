// wlr_pack_probe.hip -- what slows the one-wave sweep down inside the fused sweep + statistics kernel (tools only, round 6).
// The stand-alone k_wave_linr does a step in 296 ns; as the sweep workgroup of k_sweep_stats (four waves of one direction
// per workgroup, 512-thread launch bounds = 256 registers per wave, six Eh rows requested ahead instead of twelve, progress
// published) the same body took 390-410 ns.  Variants: W = active waves per workgroup, LB = launch bounds (threads),
// PUB = progress publication compiled in; the prefetch depth of the PUB variants is the macro PIPE_PD (build twice).
//   hipcc ... [-DPIPE_PD=12] -o wlr_pack_probe wlr_pack_probe.hip ; wlr_pack_probe [B Lm]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/svihmm.h"
#include "../../pysvihmm_amd/csrc/svihmm_common.h"
#include "../../pysvihmm_amd/csrc/device_helpers.h"
#include "../../pysvihmm_amd/csrc/kernels_wave_linr.h"
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int W, int LB, bool PUBV>
__global__ __launch_bounds__(LB) void k_pack(const double* Eh, const double* kexp, const double* A, const double* AT,
                                             const double* mi, const double* l0, int Lm, int K, int B, double* ah, double* bh,
                                             double* hx, double* gx, double* lb, double* lz, double2* zf, WlrPub pub) {
  extern __shared__ double smem[];
  const int wave = threadIdx.x >> 6, j = threadIdx.x & 63, bx = blockIdx.x;
  const int b = W * (bx >> 1) + wave;
  if (wave >= W || b >= B) return;
  WlrRing<double>* rings = reinterpret_cast<WlrRing<double>*>(smem);
  if ((bx & 1) == 0) wave_linr_body<true, true, double, double, PUBV>(Eh, kexp, A, mi, l0, (size_t)K, Lm, K, ah, hx, lb, lz, zf, rings[wave], b, j, &pub);
  else wave_linr_body<false, true, double, double, PUBV>(Eh, kexp, AT, mi, l0, (size_t)K, Lm, K, bh, gx, lb, lz, zf, rings[wave], b, j, &pub);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, Lm = argc > 2 ? atoi(argv[2]) : 257, K = 64, reps = 100;
  const size_t n = (size_t)B * Lm;
  std::vector<double> Eh(n * K);
  for (auto& v : Eh) v = 0.05 + 0.95 * rand() / (double)RAND_MAX;
  std::vector<double> A(K * K), AT(K * K), kexp(n, 0.0), mi(K, -4.0), l0((size_t)B * K, -1.0);
  for (int i = 0; i < K; ++i) { double s = 0; for (int j = 0; j < K; ++j) { A[i * K + j] = rand() / (double)RAND_MAX; s += A[i * K + j]; }
    for (int j = 0; j < K; ++j) { A[i * K + j] /= s; AT[j * K + i] = A[i * K + j]; } }
  double *dE, *da, *db, *dA, *dAT, *dk, *dmi, *dl0, *dhx, *dgx, *dlb, *dlz; double2* dzf; unsigned* dcnt;
  CKH(hipMalloc(&dE, n * K * 8)); CKH(hipMalloc(&da, n * K * 8)); CKH(hipMalloc(&db, n * K * 8));
  CKH(hipMalloc(&dA, K * K * 8)); CKH(hipMalloc(&dAT, K * K * 8)); CKH(hipMalloc(&dk, n * 8)); CKH(hipMalloc(&dmi, K * 8));
  CKH(hipMalloc(&dl0, (size_t)B * K * 8)); CKH(hipMalloc(&dhx, n * 8)); CKH(hipMalloc(&dgx, n * 8));
  CKH(hipMalloc(&dlb, B * 8)); CKH(hipMalloc(&dlz, B * 8)); CKH(hipMalloc(&dzf, B * 16)); CKH(hipMalloc(&dcnt, 12 * 64));
  CKH(hipMemset(dcnt, 0, 12 * 64));
  CKH(hipMemcpy(dE, Eh.data(), n * K * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dA, A.data(), K * K * 8, hipMemcpyHostToDevice)); CKH(hipMemcpy(dAT, AT.data(), K * K * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dk, kexp.data(), n * 8, hipMemcpyHostToDevice)); CKH(hipMemcpy(dmi, mi.data(), K * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dl0, l0.data(), (size_t)B * K * 8, hipMemcpyHostToDevice));
  WlrPub pub = {};
  pub.cnt = dcnt; pub.nb = 5;
  const int thr[5] = {154, 180, 206, 232, 256};
  for (int i = 0; i < 5; ++i) pub.thr[i] = thr[i];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(W, LB, PUBV, NAME)                                                                                          \
  do {                                                                                                                   \
    const size_t lds = (size_t)W * sizeof(WlrRing<double>);                                                              \
    hipFuncSetAttribute((const void*)k_pack<W, LB, PUBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
    const dim3 grid(2 * ((B + W - 1) / W));                                                                              \
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_pack<W, LB, PUBV>), grid, dim3(LB), lds, 0, dE, dk, dA, dAT, dmi, dl0, Lm, K, B, da, db, dhx, dgx, dlb, dlz, dzf, pub); \
    CKH(hipDeviceSynchronize());                                                                                         \
    hipEventRecord(e0);                                                                                                  \
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_pack<W, LB, PUBV>), grid, dim3(LB), lds, 0, dE, dk, dA, dAT, dmi, dl0, Lm, K, B, da, db, dhx, dgx, dlb, dlz, dzf, pub); \
    hipEventRecord(e1);                                                                                                  \
    CKH(hipEventSynchronize(e1));                                                                                        \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                                          \
    printf("%-52s %7.2f us per launch = %6.1f ns per step\n", NAME, 1e3 * ms / reps, 1e6 * ms / reps / (Lm - 1));        \
  } while (0)
  printf("B = %d, Lm = %d, PIPE_PD = %d\n", B, Lm, PIPE_PD);
  RUN(1, 64, false, "1 wave / wg,  64 threads, no publication (PD 12)");
  RUN(1, 64, true, "1 wave / wg,  64 threads, publication (PIPE_PD)");
  RUN(1, 512, true, "1 wave / wg, 512-thread bounds, publication");
  RUN(2, 128, true, "2 waves / wg, 128 threads, publication");
  RUN(4, 256, true, "4 waves / wg, 256 threads, publication");
  RUN(4, 512, true, "4 waves / wg, 512-thread bounds, publication");
  RUN(4, 256, false, "4 waves / wg, 256 threads, no publication (PD 12)");
  return 0;
}
