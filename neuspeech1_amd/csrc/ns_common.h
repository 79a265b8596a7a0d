// Shared device/host helpers for libneuspeech_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <initializer_list>
#include "../../include/neuspeech_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ns_u4v __attribute__((ext_vector_type(4)));   // a 128-bit inline-asm operand (HIP's uint4 is a struct)

// few-query one-pass attention backward (ns_attn.hip attn_bwd_fewq_kernel): key blocks of 128 per workgroup = keys per fp32 dQ slab / 128
#ifndef NS_FEWQ_KPW
#define NS_FEWQ_KPW 4
#endif

// ---- error plumbing (thread-local message, int status; no exceptions cross the ABI)
void ns_set_error(const char* fmt, ...);

#define NS_CHECK_ARG(cond, ...)                     \
  do {                                              \
    if (!(cond)) {                                  \
      ns_set_error(__VA_ARGS__);                    \
      return NS_ERR_BAD_ARG;                        \
    }                                               \
  } while (0)

// Kernel attributes (the opt-in to more than 64 KiB of dynamic LDS) belong to the function ON ONE DEVICE: a once-flag per process
// leaves a second device's copy of the kernel without it, and its first launch there fails with an opaque error.  One bit per
// device ordinal; hipFuncSetAttribute is idempotent, so two threads racing through a device's first launch is harmless.  Returns
// false with the error text set when the runtime refuses (callers return NS_ERR_HIP).
struct ns_dev_once { std::atomic<uint64_t> done{0}; };
inline bool ns_dyn_lds_once(ns_dev_once& o, std::initializer_list<const void*> fns, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  if (o.done.load(std::memory_order_acquire) & bit) return true;
  for (const void* f : fns) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      ns_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", what, bytes, dev, hipGetErrorString(e));
      return false;
    }
  }
  o.done.fetch_or(bit, std::memory_order_release);
  return true;
}

#define NS_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ns_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__)); \
      return NS_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

// ---- device math
// exact-erf GELU (torch.nn.functional.gelu default) evaluated with the Abramowitz-Stegun 7.1.26 rational form of
// erf (|error| <= 1.5e-7, far below the fp16 rounding of every value these feed): ~12 VALU + one exp + one rcp
// instead of libm's branchy erff.  The same exp(-x^2/2) serves the pdf term of the derivative.
// 1 / x as ONE v_rcp_f32 (1 ulp).  __frcp_rn compiles to the correctly rounded division sequence (v_div_scale x 2, v_rcp, five fma,
// v_div_fmas, v_div_fixup: 11 instructions per element -- a fifth of the GELU epilogue's VALU stream, found in the ISA of the fc1
// launches in round 4); the argument here is 1 + 0.33 |x| / sqrt2 in [1, 16], and a 6e-8 relative error in t moves erf by less than the
// 1.5e-7 of the rational form itself.
__device__ __forceinline__ float ns_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ void ns_gelu_terms(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = ns_rcp(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);                       // = exp(-x^2/2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;                // erf(|x|/sqrt2)
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float ns_gelu(float x) {
  float cdf, pdf;
  ns_gelu_terms(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ void ns_gelu_both(float x, float& g, float& dg) {
  float cdf, pdf;
  ns_gelu_terms(x, cdf, pdf);
  g = x * cdf;
  dg = cdf + x * pdf;
}
// Two elements at a time with packed fp32 arithmetic (v_pk_mul / v_pk_fma_f32: two lanes' worth of FMA per issue slot on
// gfx950); exp and rcp stay scalar.  Same formula, same constants, same rounding per element as ns_gelu_both.
typedef float ns_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ns_gelu_both2(ns_f2 x, ns_f2& g, ns_f2& dg) {
  const ns_f2 ax = {fabsf(x.x), fabsf(x.y)};
  const ns_f2 z = ax * 0.70710678118654752440f;
  const ns_f2 den = z * 0.3275911f + 1.0f;
  const ns_f2 t = {ns_rcp(den.x), ns_rcp(den.y)};
  const ns_f2 nz2 = -z * z;
  const ns_f2 e = {__expf(nz2.x), __expf(nz2.y)};
  ns_f2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  poly = poly * t;
  const ns_f2 erf_abs = 1.0f - poly * e;
  const ns_f2 erf_s = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
  const ns_f2 cdf = erf_s * 0.5f + 0.5f;
  const ns_f2 pdf = e * 0.39894228040143267794f;
  g = x * cdf;
  dg = x * pdf + cdf;
}

__device__ __forceinline__ float ns_gelu_grad(float x) {
  float cdf, pdf;
  ns_gelu_terms(x, cdf, pdf);
  return cdf + x * pdf;
}

// LoRA-dropout keep decision, identical in the forward down-projection (A operand), the dgrad epilogue mask and
// the weight-gradient staging: one 32-bit hash per group of 4 consecutive columns, one byte per element, keep iff
// byte >= thr8.  The drop probability is therefore quantised to thr8/256 (0.05 -> 13/256 = 0.0508) and the
// survivors are scaled by 1/(1 - thr8/256).
// (the two input products are linear mod 2^32: a caller that walks rows / column groups by constant steps forms a * NS_HASH_A and
//  b * NS_HASH_B once and adds constants -- 32-bit integer multiplies are quarter-rate VALU operations)
#define NS_HASH_A 0x9E3779B1u
#define NS_HASH_B 0x85EBCA77u
__device__ __forceinline__ uint32_t ns_hash3_mix(uint32_t x) {      // x = seed ^ a * NS_HASH_A ^ b * NS_HASH_B
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t ns_hash3(uint32_t seed, uint32_t a, uint32_t b) {
  return ns_hash3_mix(seed ^ (a * NS_HASH_A) ^ (b * NS_HASH_B));
}
// effective dropout seed of a launch: ns_gemm_desc.seed_dev (a device-resident step counter, wave-uniform scalar load) folded in
__device__ __forceinline__ uint32_t ns_eff_seed(uint32_t seed, const uint32_t* seed_dev) {
  return seed_dev ? seed + *seed_dev * 0x9E3779B1u : seed;
}
__device__ __forceinline__ uint32_t ns_drop_thr8(float p) { return (uint32_t)(p * 256.0f + 0.5f); }
__device__ __forceinline__ float ns_drop_inv(float p) { return 256.0f / (256.0f - (float)ns_drop_thr8(p)); }
// The kernels apply the keep MASK only; the survivors' 1/(1 - thr8/256) rides in the caller's `alpha` (forward
// down-projection, and the du GEMM of the backward), so no site multiplies per element.
// Packed form for fp16 operands: bit 7 of byte e of `ge` = keep flag of element e (thr8 <= 128), expanded to two
// dwords of half2 AND-masks.
__device__ __forceinline__ void ns_keep_masks(uint32_t w, uint32_t thr8, uint32_t& m01, uint32_t& m23) {
  const uint32_t ge = (((w & 0x7F7F7F7Fu) + (0x80u - thr8) * 0x01010101u) | w) & 0x80808080u;
  const uint32_t ff = (ge - (ge >> 7)) | ge;   // bytes 0x00 / 0xFF: 0x80 - 0x01 = 0x7F per kept byte (no borrow across bytes), | 0x80
                                                // ((f << 8) - f on the 0 / 1 bytes came out of hipcc as a quarter-rate v_mul_lo_u32 by 0xff)
  m01 = __builtin_amdgcn_perm(ff, ff, 0x01010000u);
  m23 = __builtin_amdgcn_perm(ff, ff, 0x03030202u);
}
__device__ __forceinline__ uint32_t ns_drop_word(uint32_t seed, uint32_t row, uint32_t col4) { return ns_hash3(seed, row, col4); }
__device__ __forceinline__ bool ns_keep(uint32_t word, uint32_t col, uint32_t thr8) { return ((word >> ((col & 3) * 8)) & 0xFFu) >= thr8; }
__device__ __forceinline__ bool ns_keep_el(uint32_t seed, uint32_t row, uint32_t col, uint32_t thr8) {
  return ns_keep(ns_drop_word(seed, row, col >> 2), col, thr8);
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
