// Common definitions for the MI355X (gfx950) hot-path library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MO_OK 0
#define MO_EINVAL (-1)      // bad argument (null pointer, non-positive dim, unsupported size)
#define MO_ELAUNCH (-2)     // hipLaunch / runtime error
#define MO_EUNSUPPORTED (-3)
#define MO_ECOMM (-4)       // RCCL error

#define MO_CHECK_ARG(cond)            \
  do {                                \
    if (!(cond)) return MO_EINVAL;    \
  } while (0)

static inline int mo_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MO_OK : MO_ELAUNCH;
}

static inline int mo_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Counter-based dropout mask: keep iff hash(seed, idx) >= thresh (thresh = p * 2^32).
// Same function in the forward epilogue and the backward loaders (mask is never stored).
__host__ __device__ static inline uint32_t mo_hash32(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  x += seed * 0x85ebca6bu;
  x ^= x >> 13; x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}
