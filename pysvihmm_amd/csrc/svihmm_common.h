// svihmm_common.h -- declarations shared by the translation units of libsvihmm_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/svihmm.h"
#include "../../include/svihmm_debug.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
#define ST_RB 32

__device__ __forceinline__ int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }


// status word of the NIW / diagonal theta builders: 1 + k = factor k not positive definite,
// NIW_STATUS_RANGE + 1 + k = factor k too far from the data centre (kernels_emission.h)
#define NIW_STATUS_RANGE (1 << 20)

// device-side dependency of the resident SVI loop (device_helpers.h: svi_gate / svi_arrive); passed by value
struct SviSync {
  const unsigned* gate;        // wait until *gate >= gate_tgt before touching the inputs (nullptr: no wait)
  unsigned gate_tgt;
  unsigned* arrive;            // += 1 per workgroup when its outputs are written (nullptr: none)
  int* status;                 // raised to SVI_SYNC_TIMEOUT when a gate gives up
  unsigned long long* stamp;   // the workgroup whose arrival brings *arrive to stamp_at stores wall_clock64() here
  unsigned stamp_at;           //   (the kernel's last workgroup; nullptr: none)
  unsigned* early;             // += 1 by the kernel's first workgroup as soon as it runs ("my predecessors on the
                               //   stream are done": the deferred ELBO kernels' gate; nullptr: none)
  unsigned* dead;              // HBM flag of the loop: non-zero once a gate has given up -- every later gate returns at
                               //   once and its kernel skips its body, so the loop's state stays at the last completed
                               //   global step and the queue drains in microseconds (nullptr: not checked)
  unsigned long long ticks;    // bound of this gate's wait in ticks of the device wall clock (0: SVI_SYNC_TICKS)
  unsigned* poison;            // HBM word of the loop: 0, or SVI_POISON_BASE - it of the EARLIEST iteration `it` whose E-step ran
                               //   on inputs that were not there (its sweeps' gate gave up / its globals kernel was skipped).
                               //   Raised (atomicMax) by those kernels with poison_val; read by the global-step kernels, which
                               //   skip from that iteration on -- they follow the sweeps in stream order, so every workgroup of
                               //   a step sees the same value and the loop's state freezes at a whole iteration
  unsigned poison_val;
};
#define SVI_POISON_BASE 0x7fffffffu
