// VERDICT r5 item 5, "bit-exact GELU by table gather (one probe, kill quickly)".
// Under autocast GELU is a pure fp16 -> fp16 function of the rounded fc1 output, so (g, g') could be READ from a table indexed by the 16 input
// bits instead of computed (ns_gelu_both2: ~27 VALU instructions per pair of elements, two of them transcendental).  What decides it is what a
// gather costs on this chip against that VALU stream, at the occupancy of the GEMM epilogue (two waves per SIMD, every CU busy).  This probe
// measures exactly that, data in registers, no HBM traffic:
//   valu   ns_gelu_both2 on 8 values per lane and iteration (the epilogue's row piece), results rounded to fp16
//   l2     one global_load_dword per element from a 256 KB table {g16 | g'16 << 16}[65536] (L2-resident; 32-bit entries so ONE gather yields both)
//   lds    one ds_read_u16 per element from a 128 KB LDS table g16[65536] (one output only: both would need 256 KB, more than the CU has)
// The values change every iteration (an xorshift of the previous bits, clamped into [-6, 6]: pre-activations of a trained MLP) so nothing hoists,
// and the fp16 inputs follow the spread of the real operand: normal(0, 1).
// hipcc --offload-arch=gfx950 -O3 -I neuspeech1_amd/csrc -I include -o tools/probe/build/gelu_gather tools/probe/gelu_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
#include "ns_common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ us8 next_bits(us8 b, uint32_t it) {
  // a cheap bijection on 16 bits per element (same cost in every variant); keeps the exponent field out of inf / nan: |x| < 8
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    uint32_t v = b[e];
    v ^= (v << 7) & 0xFFFFu; v ^= v >> 9; v ^= (it * 0x9E37u) & 0x3FFu;     // mantissa-heavy shuffle
    const uint32_t ex = (v >> 10) & 0x1Fu;
    v = (v & 0x83FFu) | ((ex > 17u ? ex - 14u : ex) << 10);                   // exponent 0..17 -> |x| < 8
    b[e] = (unsigned short)v;
  }
  return b;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const us8* __restrict__ src, const uint32_t* __restrict__ tab32, const unsigned short* __restrict__ tab16,
                                         int iters, uint32_t* __restrict__ sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lt[];
  const int tid = threadIdx.x;
  if (MODE == 2) {
    for (int i = tid; i < 65536 / 8; i += 512) ((uint4*)lt)[i] = ((const uint4*)tab16)[i];
    __syncthreads();
  }
  us8 b = src[blockIdx.x * 512 + tid];
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    b = next_bits(b, (uint32_t)it);
    if (MODE == 0) {
      const h8 x = __builtin_bit_cast(h8, b);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        ns_f2 g, dg;
        ns_gelu_both2(ns_f2{(float)x[e], (float)x[e + 1]}, g, dg);
        const _Float16 g0 = (_Float16)g.x, g1 = (_Float16)g.y, d0 = (_Float16)dg.x, d1 = (_Float16)dg.y;
        acc += (uint32_t)__builtin_bit_cast(unsigned short, g0) + ((uint32_t)__builtin_bit_cast(unsigned short, d0) << 16);
        acc ^= (uint32_t)__builtin_bit_cast(unsigned short, g1) + ((uint32_t)__builtin_bit_cast(unsigned short, d1) << 16);
      }
    } else if (MODE == 1) {
      uint32_t v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tab32[b[e]];
#pragma unroll
      for (int e = 0; e < 8; e += 2) { acc += v[e]; acc ^= v[e + 1]; }
    } else if (MODE == 3) {      // the loop's own cost: the bit shuffle and the checksum
#pragma unroll
      for (int e = 0; e < 8; e += 2) { acc += b[e]; acc ^= b[e + 1]; }
    } else {
      uint32_t v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = lt[b[e]];
#pragma unroll
      for (int e = 0; e < 8; e += 2) { acc += v[e]; acc ^= v[e + 1]; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static float gelu_ref(float x, float& dg) {
  const double c = 0.5 * (1.0 + erf((double)x * 0.70710678118654752440));
  dg = (float)(c + (double)x * 0.39894228040143267794 * exp(-0.5 * (double)x * x));
  return (float)(x * c);
}

int main() {
  std::vector<unsigned short> src(256 * 512 * 8);
  std::mt19937 rng(5);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : src) { _Float16 h = (_Float16)nd(rng); v = __builtin_bit_cast(unsigned short, h); }
  std::vector<uint32_t> t32(65536);
  std::vector<unsigned short> t16(65536);
  for (uint32_t i = 0; i < 65536; ++i) {
    unsigned short bits = (unsigned short)i;
    const float x = (float)__builtin_bit_cast(_Float16, bits);
    float dg = 0.f;
    const float g = std::isfinite(x) ? gelu_ref(x, dg) : x;
    const _Float16 g16 = (_Float16)g, d16 = (_Float16)dg;
    t16[i] = __builtin_bit_cast(unsigned short, g16);
    t32[i] = (uint32_t)t16[i] | ((uint32_t)__builtin_bit_cast(unsigned short, d16) << 16);
  }
  us8* dsrc; uint32_t* dt32; unsigned short* dt16; uint32_t* sink; unsigned long long* cyc;
  CK(hipMalloc(&dsrc, src.size() * 2)); CK(hipMalloc(&dt32, 65536 * 4)); CK(hipMalloc(&dt16, 65536 * 2));
  CK(hipMalloc(&sink, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMemcpy(dsrc, src.data(), src.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt32, t32.data(), 65536 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt16, t16.data(), 65536 * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const int iters = 2000;
  const char* names[4] = {"valu (ns_gelu_both2: g and g')", "l2 gather (g | g' in one dword, 256 KB table)", "lds gather (g only, 128 KB table)",
                          "loop only (bit shuffle + checksum)"};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, dsrc, dt32, dt16, iters, sink, cyc);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, dsrc, dt32, dt16, iters, sink, cyc);
      else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, dsrc, dt32, dt16, iters, sink, cyc);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 128 * 1024, 0, dsrc, dt32, dt16, iters, sink, cyc);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> c(256);
      CK(hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
      double mean = 0;
      for (auto v : c) mean += (double)v / 256.0;
      const double elems_per_cu = 512.0 * 8.0 * iters;
      if (rep == 2)
        printf("%-50s %8.3f ms   %7.2f cycles per element and CU   = %6.1f k cycles per 256 x 256 tile   (%.2f Gelem/s per CU)\n", names[mode], ms,
               mean / elems_per_cu, mean / elems_per_cu * 65536.0 / 1e3, elems_per_cu / (ms * 1e-3) / 1e9);
    }
  return 0;
}
