// How fast can ONE CU store a 256 x 256 fp16 tile (128 KiB, 16-B lane stores of whole 512-B rows: the GEMM epilogue's pattern), as a function of
// how many CUs store at the same time?  The plain epilogue of ns_gemm_p8s takes ~11 k cycles per tile = 11.5 B per cycle and CU = the chip's
// 6 TB/s divided by 256: is that the CU's own store path, or only its fair share when every CU bursts together?
// 256 workgroups of 512 threads (one per CU); workgroup b stores `reps` tiles iff b % stride == 0, the others leave at once.
// hipcc --offload-arch=gfx950 -O3 -o tools/probe/build/store_burst tools/probe/store_burst.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(half_t* C, int N, int stride, int reps, unsigned long long* cyc) {
  if (blockIdx.x % stride) return;
  const int tid = threadIdx.x, cg = tid & 31, r0 = tid >> 5;
  const int tiles_n = N / 256;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    const int tile = blockIdx.x + 256 * r;            // a different tile per repetition: nothing is rewritten while it is still in flight
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    half_t* base = C + (size_t)tm * 256 * N + tn * 256;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rl = r0 + 16 * i;
      half8 v = {(half_t)rl, (half_t)cg, (half_t)r, 1, 2, 3, 4, 5};
      if (MODE == 0) *(half8*)(base + (size_t)rl * N + cg * 8) = v;
      else if (MODE == 1) __builtin_nontemporal_store(v, (half8*)(base + (size_t)rl * N + cg * 8));
      else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(base + (size_t)rl * N + cg * 8), "v"(v) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the stores have been acknowledged
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1536, reps = 16;
  printf("row stride %d B (N = %d)\n", 2 * N, N);
  const size_t rows = (size_t)(256 * reps / (N / 256) + 1) * 256;
  half_t* C; unsigned long long* cyc;
  CK(hipMalloc(&C, rows * N * 2)); CK(hipMalloc(&cyc, 256 * 8));
  for (int mode = 0; mode < 3; ++mode)
  for (int stride : {1, 2, 4, 8, 16, 32, 64, 256}) {
    if (mode > 0 && stride > 4) continue;
    std::vector<double> best;
    for (int it = 0; it < 5; ++it) {
      CK(hipMemset(cyc, 0, 256 * 8));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, C, N, stride, reps, cyc);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, C, N, stride, reps, cyc);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, C, N, stride, reps, cyc);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256);
      CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
      double s = 0; int n = 0;
      for (int b = 0; b < 256; b += stride) { s += (double)h[b]; ++n; }
      best.push_back(s / n / reps);
    }
    std::sort(best.begin(), best.end());
    printf("%s storing CUs %3d: %8.0f cycles per 128-KiB tile (median of 5) = %5.1f B per cycle and CU\n", mode == 0 ? "plain   " : (mode == 1 ? "nt      " : "sc0 sc1 "), 256 / stride, best[2], 131072.0 / best[2]);
  }
  return 0;
}
