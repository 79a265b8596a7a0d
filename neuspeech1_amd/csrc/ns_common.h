// Shared device/host helpers for libneuspeech_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/neuspeech_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (thread-local message, int status; no exceptions cross the ABI)
void ns_set_error(const char* fmt, ...);

#define NS_CHECK_ARG(cond, ...)                     \
  do {                                              \
    if (!(cond)) {                                  \
      ns_set_error(__VA_ARGS__);                    \
      return NS_ERR_BAD_ARG;                        \
    }                                               \
  } while (0)

#define NS_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ns_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__)); \
      return NS_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

// ---- device math
__device__ __forceinline__ float ns_gelu(float x) {
  // exact (erf) GELU, torch.nn.functional.gelu default
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float ns_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// counter-based dropout keep-decision: same (seed,row,col) -> same decision in
// the forward LoRA-down kernel, the dgrad epilogue and the wgrad staging.
__device__ __forceinline__ uint32_t ns_hash3(uint32_t seed, uint32_t a, uint32_t b) {
  uint32_t x = seed ^ (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float ns_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ns_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
