// Probe: NT GEMM main loop on 128 x 128 WAVE tiles, one 512-register wave per SIMD (4-wave workgroup, 256 x 256 tile).
// C16[M][N] = A[M][K] * B[N][K]^T, fp16 operands, fp32 accumulation in 64 v_mfma_f32_16x16x32_f16 accumulators per wave.
//
// LDS ring: 4 stages of one 32-deep K step (A 256 rows x 64 B + B 256 rows x 64 B = 32 KiB), filled by LDS-DMA
// (buffer_load_dwordx4 ... lds, 1-KiB pieces of 16 rows x 64 B, 8 pieces per wave and step) three steps ahead.
// 16-B chunk g of row r sits at chunk g ^ sigma((r >> 2) & 3), sigma = (0, 2, 3, 1): conflict-free ds_read_b128 fragment
// reads for the 16x16x32 operand pattern on 64-B rows (applied to the DMA source address and to the read address).
// One barrier per step, in the MIDDLE of the step's MFMA stream:
//   first half  (accumulator rows 0..63 of the wave: 32 MFMAs)  +  8 DMA pieces of step t+3  +  A fragments 4..7 of step t
//   s_waitcnt lgkmcnt(0), vmcnt(16) [step t+1 landed], s_barrier
//   second half (rows 64..127: 32 MFMAs)  +  B fragments 0..7 and A fragments 0..3 of step t+1
// so a stage is overwritten (step t+3 -> stage (t-1) & 3) only after every wave has passed the barrier of step t-1, behind
// which no wave reads stage t-1 any more, and it is first read one barrier after the wait that retires its DMA.
//
// hipcc --offload-arch=gfx950 -O3 -o /tmp/w4_gemm tools/probe/w4_gemm.hip && /tmp/w4_gemm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 256, BN = 256, BK = 32, NST = 4, NTH = 256;
constexpr int HALF_ST = 256 * 64;          // 16 KiB: the A (or B) rows of one stage
constexpr int STAGE = 2 * HALF_ST;         // 32 KiB
constexpr int LDH = BN * 2 + 16;           // epilogue: bytes per staged fp16 row
constexpr int LDS_BYTES = (BM * LDH > NST * STAGE) ? BM * LDH : NST * STAGE;

#define W4_BARRIER()                            \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

__device__ __forceinline__ int sigma4(int x) { return (0x1320 >> (4 * x)) & 3; }   // (0, 2, 3, 1)

template <bool STAMPS, int MODE>
__global__ __launch_bounds__(NTH) void w4_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C, int M, int N, int K,
                                                 unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tiles_n = N / BN, tiles_m = M / BM;
  const int nwg = tiles_m * tiles_n;
  int wgid;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (STAMPS && tid == 0) stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x80000000u, 0x00020000);
  // DMA sources: wave w fills pieces 4w .. 4w+3 (16 rows each) of the A half and of the B half of a stage
  uint32_t a_off[4], b_off[4];
  {
    const int g = (lane & 3) ^ sigma4((lane >> 4) & 3);       // row-in-piece = lane >> 2, (row >> 2) & 3 = (lane >> 4) & 3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 16 * (4 * wave + j) + (lane >> 2);
      a_off[j] = 2u * ((uint32_t)(m0 + row) * (uint32_t)K + (uint32_t)g * 8u);
      b_off[j] = 2u * ((uint32_t)(n0 + row) * (uint32_t)K + (uint32_t)g * 8u);
    }
  }
  const int nsteps = K / BK;
  auto dma = [&](int t, int j) __attribute__((always_inline)) {       // piece j (0..7) of step t: A pieces 0..3, B pieces 4..7
    const bool isb = j >= 4;
    const uint32_t off = isb ? b_off[j & 3] : a_off[j & 3];
    char* const dst = smem + (t & 3) * STAGE + (isb ? HALF_ST : 0) + (4 * wave + (j & 3)) * 1024;
    // steps past the end fetch zeros (offset beyond num_records)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rsrc_b : rsrc_a, (lds_void*)dst, 16, t < nsteps ? off : 0x80000000u, 2 * BK * t, 0, 0);
  };

  const int fro = l15 * 64 + ((lg ^ sigma4((l15 >> 2) & 3)) << 4);
  const int a_base = wm * 8192 + fro;
  const int b_base = HALF_ST + wn * 8192 + fro;
  half8 af[8], bf[2][8];
  auto read_a = [&](int t, int mt) __attribute__((always_inline)) {
    af[mt] = *(const half8*)(smem + (t & 3) * STAGE + a_base + mt * 1024);
  };
  auto read_b = [&](int t, int set, int nt) __attribute__((always_inline)) {
    bf[set][nt] = *(const half8*)(smem + (t & 3) * STAGE + b_base + nt * 1024);
  };
  auto mma = [&](int set, int mt, int nt) __attribute__((always_inline)) {
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[set][nt], af[mt], acc[mt][nt], 0, 0, 0);
  };
#define SB() __builtin_amdgcn_sched_barrier(0)

  // prologue: steps 0, 1, 2 in flight; fragments of step 0
#pragma unroll
  for (int j = 0; j < 8; ++j) dma(0, j);
#pragma unroll
  for (int j = 0; j < 8; ++j) dma(1, j);
#pragma unroll
  for (int j = 0; j < 8; ++j) dma(2, j);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  W4_BARRIER();
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) read_b(0, 0, nt);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) read_a(0, mt);
  if (STAMPS && tid == 0) stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();

  // MODE 0: all 8 pieces in the first half.  1: no DMA in the loop (bound of the MFMA + ds_read stream).  2: even waves issue in the first
  // half, odd waves in the second (the CU's address path sees 2 waves at a time, not 4).  3: 4 pieces per half.
  const bool odd = wave & 1;
  auto step = [&](int t, int set) __attribute__((always_inline)) {
    // ---- first half: rows 0..63 (mt 0..3); one filler group after every 4 MFMAs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int mt = q >> 1, nb = (q & 1) * 4;
      mma(set, mt, nb + 0); mma(set, mt, nb + 1);
      if (MODE == 0) dma(t + 3, q);
      if (MODE == 2 && !odd) dma(t + 3, q);
      if (MODE == 3 && (q & 1)) dma(t + 3, q >> 1);
      if (q < 4) read_a(t, 4 + q);
      mma(set, mt, nb + 2); mma(set, mt, nb + 3);
      SB();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2) { if (odd) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    W4_BARRIER();
    SB();
    // ---- second half: rows 64..127 (mt 4..7); the next step's B fragments and its first four A fragments
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int mt = 4 + (q >> 1), nb = (q & 1) * 4;
      mma(set, mt, nb + 0); mma(set, mt, nb + 1);
      read_b(t + 1, set ^ 1, q);
      if (q >= 4) read_a(t + 1, q - 4);      // af[0..3] are dead from the first half on
      if (MODE == 2 && odd) dma(t + 3, q);
      if (MODE == 3 && (q & 1)) dma(t + 3, 4 + (q >> 1));
      mma(set, mt, nb + 2); mma(set, mt, nb + 3);
      SB();
    }
  };
  for (int t = 0; t < nsteps; t += 4) {
    step(t, 0);
    step(t + 1, 1);
    step(t + 2, 0);
    step(t + 3, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_BARRIER();
  if (STAMPS && tid == 0) stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: fp16 rounding in registers, whole tile staged once, 16-B row stores
  char* const hs = smem;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int rl = wm * 128 + mt * 16 + l15;
      const int cl = wn * 128 + nt * 16 + 4 * lg;
      const f32x4 a = acc[mt][nt];
      const half4 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
      *(half4*)(hs + rl * LDH + cl * 2) = h;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_BARRIER();
  const int ecg = tid & 31, er0 = tid >> 5;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const int rl = er0 + 8 * i;
    const half8 v = *(const half8*)(hs + rl * LDH + ecg * 16);
    *(half8*)(C + (size_t)(m0 + rl) * N + n0 + ecg * 8) = v;
  }
  if (STAMPS && tid == 0) stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime();
}

static void fill(std::vector<half_t>& v, uint64_t seed, float scale) {
  uint64_t s = seed;
  for (auto& x : v) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    x = (half_t)(((int)((s >> 33) & 0xFFFF) - 32768) / 32768.0f * scale);
  }
}

template <int MODE>
int run(int M, int N, int K, bool check) {
  std::vector<half_t> hA((size_t)M * K), hB((size_t)N * K);
  fill(hA, 1, 1.0f);
  fill(hB, 2, 0.05f);
  half_t *dA, *dB, *dC;
  unsigned long long* dS;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  const int tiles = (M / BM) * (N / BN);
  hipMalloc(&dS, (size_t)tiles * 64);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)w4_kernel<false, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)w4_kernel<true, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL((w4_kernel<false, MODE>), dim3(tiles), dim3(NTH), LDS_BYTES, 0, dA, dB, dC, M, N, K, dS);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((w4_kernel<false, MODE>), dim3(tiles), dim3(NTH), LDS_BYTES, 0, dA, dB, dC, M, N, K, dS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::fmin(best, ms / 10);
  }
  printf("mode %d M %6d N %5d K %5d: %8.1f us  %7.1f TFLOP/s", MODE, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
  // stamps: median cycles of prologue / main loop / epilogue
  hipLaunchKernelGGL((w4_kernel<true, MODE>), dim3(tiles), dim3(NTH), LDS_BYTES, 0, dA, dB, dC, M, N, K, dS);
  hipDeviceSynchronize();
  std::vector<unsigned long long> st((size_t)tiles * 8);
  hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> d0, d1, d2;
  for (int i = 0; i < tiles; ++i) {
    d0.push_back((double)(st[i * 8 + 1] - st[i * 8 + 0]));
    d1.push_back((double)(st[i * 8 + 2] - st[i * 8 + 1]));
    d2.push_back((double)(st[i * 8 + 3] - st[i * 8 + 2]));
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const double ideal = (double)(K / 32) * 1024.0;
  const double m1 = med(d1);
  printf("   cycles: prologue %6.0f  loop %7.0f (MFMA-ideal %7.0f = %4.1f %%)  epilogue %6.0f\n", med(d0), m1, ideal, 100.0 * ideal / m1, med(d2));
  int bad = 0;
  if (check) {
    std::vector<half_t> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    uint64_t s = 99;
    for (int it = 0; it < 4000; ++it) {
      s = s * 6364136223846793005ULL + 1442695040888963407ULL;
      const int r = (int)((s >> 33) % M);
      s = s * 6364136223846793005ULL + 1442695040888963407ULL;
      const int c = (int)((s >> 33) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)r * K + k] * (double)(float)hB[(size_t)c * K + k];
      const double got = (double)(float)hC[(size_t)r * N + c];
      const double err = std::fabs(got - ref) / (std::fabs(ref) + 0.05);
      if (err > worst) worst = err;
      if (err > 2e-2) { if (bad < 5) printf("  MISMATCH r %d c %d got %f ref %f\n", r, c, got, ref); ++bad; }
    }
    printf("   check: worst rel err %.2e, %d bad of 4000\n", worst, bad);
  }
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dS);
  return bad;
}

template <int MODE>
int suite() {
  int bad = 0;
  const bool chk = MODE != 1;
  bad += run<MODE>(2048, 1536, 512, chk);
  bad += run<MODE>(96000, 1536, 512, chk);    // q|k|v
  bad += run<MODE>(96000, 2048, 512, false);   // fc1
  bad += run<MODE>(96000, 512, 2048, chk);    // fc2
  bad += run<MODE>(96000, 512, 1536, false);   // q|k|v dgrad
  bad += run<MODE>(4096, 4096, 4096, false);
  return bad;
}

int main(int argc, char** argv) {
  int bad = 0;
  bad += suite<0>();
  bad += suite<1>();
  bad += suite<2>();
  bad += suite<3>();
  printf(bad ? "FAILED\n" : "OK\n");
  return bad != 0;
}
