// The clock and the MFMA rate this chip HOLDS under a sustained fp16 matrix load.
// Back-to-back v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 with operands in registers, every CU busy, launched
// back to back for >= 2 s; the in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz around the loop
// (MI355X_MICROARCH.md 'DVFS give-back' item 6), the rate is FLOP / wall time by HIP events.
//   modes: shape {32x32x16, 16x16x32} x waves per SIMD {1, 2} x operands {random, zero} x {registers only, operands re-read
//   from LDS by ds_read_b128 every step (what a GEMM main loop does)}.
// hipcc --offload-arch=gfx950 -O3 -o tools/probe/build/mfma_clock tools/probe/mfma_clock.hip && tools/probe/build/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SHAPE 0: 32x32x16 (4 independent 16-register accumulators), SHAPE 1: 16x16x32 (16 independent 4-register accumulators).
// LDSRD: every step re-reads its 8 operand fragments (8 x ds_read_b128 per wave and 8 (32x32) / 16 (16x16) MFMAs: the
// 128 x 64 wave tile's ratio is 12 reads per 32 MFMAs of 16x16x32, i.e. less than this).
template <int SHAPE, bool LDSRD>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ src, int iters, unsigned long long* stamps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, NT = blockDim.x;
  half8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 4095]; b[i] = src[(tid * 8 + 4 + i) & 4095]; }
  if (LDSRD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { *(half8*)(smem + (i * NT + tid) * 16) = a[i]; *(half8*)(smem + ((4 + i) * NT + tid) * 16) = b[i]; }
    __syncthreads();
  }
  f32x16 c32[4];
  f32x4 c16[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) c32[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) c16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDSRD) {
      // conflict-free: every wave instruction reads 64 consecutive 16-B pieces
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = *(const half8*)(smem + (i * NT + tid) * 16); b[i] = *(const half8*)(smem + ((4 + i) * NT + tid) * 16); }
    }
    if (SHAPE == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + r) & 3], c32[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) c16[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], c16[i], 0, 0, 0);
    }
    if (LDSRD) asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += c32[i][e];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c16[i][0] + c16[i][1] + c16[i][2] + c16[i][3];
  if (s == 12345.678f) sink[0] = s;       // keeps the chains live
  if ((tid & 63) == 0 && stamps) {
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2 + 0] = t1 - t0;
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE, bool LDSRD>
static void run(const char* name, int threads, const half8* src, unsigned long long* stamps, float* sink, int iters, double seconds) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int waves = threads / 64;
  const size_t lds = LDSRD ? (size_t)threads * 8 * 16 : 0;
  // one launch to size the run
  hipLaunchKernelGGL((k<SHAPE, LDSRD>), dim3(256), dim3(threads), lds, 0, src, iters, stamps, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<SHAPE, LDSRD>), dim3(256), dim3(threads), lds, 0, src, iters, stamps, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms1 = 0.f;
  CK(hipEventElapsedTime(&ms1, e0, e1));
  const int n = std::max(4, (int)(seconds * 1e3 / ms1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<SHAPE, LDSRD>), dim3(256), dim3(threads), lds, 0, src, iters, stamps, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)256 * 8 * 2);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < waves; ++w) {
      const double c = (double)h[(b * 8 + w) * 2], r = (double)h[(b * 8 + w) * 2 + 1];
      if (r > 0) { clk.push_back(c / r * 0.1); cyc.push_back(c); }
    }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double mfma_per_wave = (double)iters * (SHAPE == 0 ? 8 : 16);
  const double flop = 256.0 * waves * mfma_per_wave * (SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32) * (double)n;
  printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"launches\": %d, \"seconds\": %.2f, \"tflops\": %.1f, \"clock_ghz_median\": %.3f, "
         "\"clock_ghz_min\": %.3f, \"clock_ghz_max\": %.3f, \"cycles_per_mfma_per_simd\": %.2f}\n",
         name, waves / 4, n, ms * 1e-3, flop / (ms * 1e-3) / 1e12, clk[clk.size() / 2], clk.front(), clk.back(),
         cyc[cyc.size() / 2] / (mfma_per_wave * (waves / 4)));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
  half8* src; unsigned long long* stamps; float* sink;
  CK(hipMalloc(&src, 4096 * 16)); CK(hipMalloc(&stamps, 256 * 8 * 2 * 8)); CK(hipMalloc(&sink, 4));
  std::vector<_Float16> h(4096 * 8);
  srand(7);
  for (int pass = 0; pass < 2; ++pass) {
    const bool zero = pass == 1;
    for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)(((rand() & 0xFFFF) / 32768.0f - 1.0f) * 0.5f);   // uniform [-0.5, 0.5): sums stay finite in fp32
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const char* tag = zero ? "zero" : "random";
    char name[96];
    const int iters = 200000;
    snprintf(name, sizeof name, "mfma_32x32x16_f16 regs %s", tag); run<0, false>(name, 256, src, stamps, sink, iters, seconds);
    snprintf(name, sizeof name, "mfma_16x16x32_f16 regs %s", tag); run<1, false>(name, 256, src, stamps, sink, iters / 2, seconds);
    snprintf(name, sizeof name, "mfma_16x16x32_f16 regs %s", tag); run<1, false>(name, 512, src, stamps, sink, iters / 4, seconds);
    snprintf(name, sizeof name, "mfma_32x32x16_f16 lds_reread %s", tag); run<0, true>(name, 256, src, stamps, sink, iters, seconds);
    snprintf(name, sizeof name, "mfma_16x16x32_f16 lds_reread %s", tag); run<1, true>(name, 256, src, stamps, sink, iters / 2, seconds);
    snprintf(name, sizeof name, "mfma_16x16x32_f16 lds_reread %s", tag); run<1, true>(name, 512, src, stamps, sink, iters / 4, seconds);
  }
  return 0;
}
