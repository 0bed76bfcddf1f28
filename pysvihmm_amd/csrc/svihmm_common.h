// svihmm_common.h -- declarations shared by the translation units of libsvihmm_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/svihmm.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
#define ST_RB 32

__device__ __forceinline__ int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

struct StatsLaunch {
  hipStream_t stream;
  const double* obs; const uint8_t* mask; const int64_t* starts;
  int64_t n; int Lm, D, K, Kp, Fp, F;
  const int* fab; const double* q; int64_t rpc; uint32_t flags; int Lq, off;
  double* part; int nchunk;
};
// pipelined != 0: k_stats_mfma2 (K <= 64); else k_stats_mfma.  Returns 0 ok, 2 = LDS too
// small for this shape (caller falls back), 1 = launch error.
int svihmm_launch_stats_mfma(const StatsLaunch* a, int pipelined);
