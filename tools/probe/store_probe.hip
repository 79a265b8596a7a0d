// Store-pattern floor for a GEMM epilogue: every 512-thread block writes one 256x256 fp16 tile of a (M x N) matrix.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/probe/store_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half_t;
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef half_t half8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(half_t* C, const float* R, float* H, int M, int N) {
  extern __shared__ char sm[];
  if (N < 0) sm[threadIdx.x] = 1;   // keeps the dynamic LDS allocation attached (occupancy control only)
  const int tiles_n = N / 256;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 256, n0 = tn * 256, tid = threadIdx.x;
  if (MODE == 0) {          // half4: 32 threads x 8 B per half row (two column groups 128 apart), 16 rows per pass
    const int cg = tid & 31, r0 = tid >> 5;
    for (int rl = r0; rl < 256; rl += 16) {
      const int row = m0 + rl; if (row >= M) continue;
      for (int g = 0; g < 2; ++g) {
        half4 v = {(half_t)rl, (half_t)cg, (half_t)g, (half_t)1};
        *(half4*)(C + (size_t)row * N + n0 + cg * 4 + g * 128) = v;
      }
    }
  } else if (MODE == 1) {   // half8: 32 threads x 16 B = one 512-B tile row, 16 rows per pass
    const int cg = tid & 31, r0 = tid >> 5;
    for (int rl = r0; rl < 256; rl += 16) {
      const int row = m0 + rl; if (row >= M) continue;
      half8 v = {(half_t)rl, (half_t)cg, 0, 1, 2, 3, 4, 5};
      *(half8*)(C + (size_t)row * N + n0 + cg * 8) = v;
    }
  } else if (MODE == 2) {   // fp32 residual: float4 load R + float4 store H, 64 threads per 1-KiB tile row
    const int cg = tid & 63, r0 = tid >> 6;
    for (int rl = r0; rl < 256; rl += 8) {
      const int row = m0 + rl; if (row >= M) continue;
      float4 v = *(const float4*)(R + (size_t)row * N + n0 + cg * 4);
      v.x += 1.f; v.y += 2.f;
      *(float4*)(H + (size_t)row * N + n0 + cg * 4) = v;
    }
  } else if (MODE == 3) {   // fp32 residual, loads batched 8 deep before the stores
    const int cg = tid & 63, r0 = tid >> 6;
    for (int rb = 0; rb < 256; rb += 64) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = min(m0 + rb + r0 + 8 * i, M - 1);
        v[i] = *(const float4*)(R + (size_t)row * N + n0 + cg * 4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + rb + r0 + 8 * i;
        v[i].x += 1.f;
        if (row < M) *(float4*)(H + (size_t)row * N + n0 + cg * 4) = v[i];
      }
    }
  } else if (MODE == 5 || MODE == 6 || MODE == 7) {
    // accumulator-shaped direct stores: wave (wm, wn) owns rows wm*128.., cols wn*64..; lane (l15, lg)
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, l15 = lane & 15, lg = lane >> 4;
    if (MODE == 5) {        // 32 x dwordx2: 16 rows x 32 B per instruction
      for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) {
        const int row = m0 + wm * 128 + a * 16 + l15; if (row >= M) continue;
        half4 v = {(half_t)a, (half_t)b, (half_t)lg, (half_t)1};
        *(half4*)(C + (size_t)row * N + n0 + wn * 64 + b * 16 + lg * 4) = v;
      }
    } else if (MODE == 6) { // 16 x dwordx4: 32 rows x 32 B per instruction (after a lane-pair swap)
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
        const int row = m0 + wm * 128 + a * 32 + (lg >> 1) * 16 + l15; if (row >= M) continue;
        half8 v = {(half_t)a, (half_t)b, (half_t)lg, 1, 2, 3, 4, 5};
        *(half8*)(C + (size_t)row * N + n0 + wn * 64 + b * 16 + (lg & 1) * 8) = v;
      }
    } else {                // fp32 residual: 32 x (dwordx4 load + dwordx4 store), 16 rows x 64 B per instruction
      float4 v[32];
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int row = min(m0 + wm * 128 + a * 16 + l15, M - 1);
          v[a * 4 + b] = *(const float4*)(R + (size_t)row * N + n0 + wn * 64 + b * 16 + lg * 4);
        }
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int row = m0 + wm * 128 + a * 16 + l15; if (row >= M) continue;
          v[a * 4 + b].x += 1.f;
          *(float4*)(H + (size_t)row * N + n0 + wn * 64 + b * 16 + lg * 4) = v[a * 4 + b];
        }
    }
  } else if (MODE == 4) {   // half8 stores only, non-temporal
    const int cg = tid & 31, r0 = tid >> 5;
    for (int rl = r0; rl < 256; rl += 16) {
      const int row = m0 + rl; if (row >= M) continue;
      half8 v = {(half_t)rl, (half_t)cg, 0, 1, 2, 3, 4, 5};
      __builtin_nontemporal_store(v, (half8*)(C + (size_t)row * N + n0 + cg * 8));
    }
  }
}

static int g_lds = 0;
template <int MODE>
void run(const char* name, half_t* C, float* R, float* H, int M, int N, double bytes) {
  const int tiles = ((M + 255) / 256) * (N / 256);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(tiles), dim3(512), g_lds, 0, C, R, H, M, N);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(tiles), dim3(512), g_lds, 0, C, R, H, M, N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-44s N=%4d  %.1f us  %.2f TB/s  (%.1f us per tile round)\n", name, N, ms * 1000, bytes / ms / 1e9, ms * 1000 / (tiles / 256.0));
}

int main() {
  const int M = 96000;
  half_t* C; float *R, *H;
  hipMalloc(&C, (size_t)M * 2048 * 2); hipMalloc(&R, (size_t)M * 2048 * 4); hipMalloc(&H, (size_t)M * 2048 * 4);
  hipMemset(R, 0, (size_t)M * 2048 * 4);
  for (int lds : {130 * 1024})
  for (int N : {1536, 512}) {
    g_lds = lds;
    printf("dynamic LDS %d KiB (=> %s)\n", lds / 1024, lds ? "1 block per CU" : "up to 4 blocks per CU");
    run<0>("fp16 tile, half4 stores", C, R, H, M, N, (double)M * N * 2);
    run<1>("fp16 tile, half8 stores", C, R, H, M, N, (double)M * N * 2);
    run<4>("fp16 tile, half8 nontemporal stores", C, R, H, M, N, (double)M * N * 2);
    run<5>("fp16 acc-shaped dwordx2 (16 rows x 32 B)", C, R, H, M, N, (double)M * N * 2);
    run<6>("fp16 acc-shaped dwordx4 (32 rows x 32 B)", C, R, H, M, N, (double)M * N * 2);
    run<7>("fp32 residual acc-shaped (16 rows x 64 B)", C, R, H, M, N, (double)M * N * 8);
    run<2>("fp32 residual load+store float4", C, R, H, M, N, (double)M * N * 8);
    run<3>("fp32 residual, 8 loads in flight", C, R, H, M, N, (double)M * N * 8);
  }
  return 0;
}
