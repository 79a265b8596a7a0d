// Probe 2: the 4-wave / 128 x 128 wave-tile main loop of w4_gemm.hip as a PERSISTENT kernel: one workgroup per CU walks its
// XCD's tiles; the LDS-DMA ring runs on across tile boundaries (the last three steps of a tile fetch the first three steps of
// the next one, its first fragments are read in the last step's second half), so a tile has no prologue; the epilogue goes
// through a 32 KiB staging buffer BESIDE the ring in four 64-row passes (accumulators -> fp16 -> swizzled LDS image -> 16-B row stores).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/w4p_gemm tools/probe/w4p_gemm.hip && /tmp/w4p_gemm
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

constexpr int BM = 256, BN = 256, BK = 32, NTH = 256;
constexpr int HALF_ST = 256 * 64;          // 16 KiB: the A (or B) rows of one stage
constexpr int STAGE = 2 * HALF_ST;         // 32 KiB
constexpr int RING = 4 * STAGE;            // 128 KiB
constexpr int STG_BYTES = 64 * 512;        // staging: 64 rows x 256 fp16
constexpr int LDS_BYTES = RING + STG_BYTES;

#define W4_BARRIER()                            \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int sigma4(int x) { return (0x1320 >> (4 * x)) & 3; }   // (0, 2, 3, 1)

template <bool V> struct w4_bool { static constexpr bool value = V; };

template <bool STAMPS, int MODE>
__global__ __launch_bounds__(NTH) void w4p_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C, int M, int N, int K,
                                                  unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tiles_n = N / BN, tiles_m = M / BM;
  const int nwg = tiles_m * tiles_n;
  const int nx = gridDim.x >> 3;
  int xstart, xcount, ti;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    xstart = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xcount = q + (xcd < r ? 1 : 0);
    ti = bid >> 3;
  }
  if (ti >= xcount) return;
  const int S = K / BK;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x80000000u, 0x00020000);
  uint32_t cur_a[4], cur_b[4], nxt_a[4], nxt_b[4];
  auto set_nxt = [&](int wtile, bool valid) __attribute__((always_inline)) {
    int tn_div = tiles_n;
    asm volatile("" : "+s"(tn_div));
    const int tm_ = wtile / tn_div;
    const int m0 = tm_ * BM, n0 = (wtile - tm_ * tn_div) * BN;
    const int g = (lane & 3) ^ sigma4((lane >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 16 * (4 * wave + j) + (lane >> 2);
      nxt_a[j] = valid ? 2u * ((uint32_t)(m0 + row) * (uint32_t)K + (uint32_t)g * 8u) : 0x80000000u;
      nxt_b[j] = valid ? 2u * ((uint32_t)(n0 + row) * (uint32_t)K + (uint32_t)g * 8u) : 0x80000000u;
    }
  };
  // piece j (0..3) of the A half / of the B half of the stage at `stg`
  auto dma_a = [&](const uint32_t (&oa)[4], int j, char* stg, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(stg + (4 * wave + j) * 1024), 16, oa[j], soff, 0, 0);
  };
  auto dma_b = [&](const uint32_t (&ob)[4], int j, char* stg, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(stg + HALF_ST + (4 * wave + j) * 1024), 16, ob[j], soff, 0, 0);
  };

  const int fro = l15 * 64 + ((lg ^ sigma4((l15 >> 2) & 3)) << 4);
  const int a_base = wm * 8192 + fro;
  const int b_base = HALF_ST + wn * 8192 + fro;
  half8 af[8], bf[8];
  f32x4 acc[8][8];

  // one 32-deep step.  g = ring position of this step; its DMA pieces fetch ring position g + 3: step kd of the current tile
  // (NXT = false) or of the next tile (NXT = true).  FIRST: the accumulators start from zero (no clearing pass).
  // MODE bit 0: relaxed waits in the first steps of a tile (the epilogue's stores are younger than the pieces waited for: counting
  // them in lets them drain under the MFMAs).  MODE bit 1: prefetch distance 4 (all 8 pieces of step t+4 in the second half of step t).
  constexpr bool RELAX = MODE & 1, DIST4 = MODE & 2;
  constexpr int NEPI = 32;      // VMEM operations per thread of one epilogue (full tiles)
  auto step = [&](auto first_c, auto nxt_c, int g, int kd, bool relax) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value, NXT = decltype(nxt_c)::value;
    char* const st_cur = smem + ((g & 3) << 15);
    char* const st_nx1 = smem + (((g + 1) & 3) << 15);
    char* const st_dma = smem + (((g + (DIST4 ? 4 : 3)) & 3) << 15);
    const int soff = kd * (BK * 2);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // (a fragment register is reloaded only once the MFMAs that read it are at least four MFMAs old: a ds_read into the source of an MFMA
    //  issued just before it waits for that MFMA -- measured 400 cycles per step when every column's reload followed its own MFMAs)
    SB();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int nb = (q >> 2) * 4, mt = q & 3;      // columns 0..3 of all four row tiles first, then columns 4..7
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[mt][nb + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nb + e], af[mt], FIRST ? z4 : acc[mt][nb + e], 0, 0, 0);
      if (!DIST4 && (q & 1)) dma_a(NXT ? nxt_a : cur_a, q >> 1, st_dma, soff);
      if (q < 4) af[4 + q] = *(const half8*)(st_cur + a_base + (4 + q) * 1024);
#pragma unroll
      for (int e = 2; e < 4; ++e) acc[mt][nb + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nb + e], af[mt], FIRST ? z4 : acc[mt][nb + e], 0, 0, 0);
      SB();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (RELAX && relax) {
      if (DIST4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 + NEPI) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + NEPI) : "memory");
    } else {
      if (DIST4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    W4_BARRIER();
    SB();
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      acc[4][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[4], FIRST ? z4 : acc[4][nt], 0, 0, 0);
      acc[5][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[5], FIRST ? z4 : acc[5][nt], 0, 0, 0);
      if (DIST4) {
        if (nt < 4) dma_a(NXT ? nxt_a : cur_a, nt, st_dma, soff);
        else dma_b(NXT ? nxt_b : cur_b, nt - 4, st_dma, soff);
      } else if (nt & 1) {
        dma_b(NXT ? nxt_b : cur_b, nt >> 1, st_dma, soff);
      }
      if (nt >= 2) bf[nt - 2] = *(const half8*)(st_nx1 + b_base + (nt - 2) * 1024);      // next step's fragment: its last readers are 4+ MFMAs old
      if (nt >= 4) af[nt - 4] = *(const half8*)(st_nx1 + a_base + (nt - 4) * 1024);      // af[0..3] are dead since the first half
      acc[6][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[6], FIRST ? z4 : acc[6][nt], 0, 0, 0);
      acc[7][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[7], FIRST ? z4 : acc[7][nt], 0, 0, 0);
      SB();
    }
    bf[6] = *(const half8*)(st_nx1 + b_base + 6 * 1024);
    SB();
    bf[7] = *(const half8*)(st_nx1 + b_base + 7 * 1024);
    SB();
  };
  typedef w4_bool<true> T;
  typedef w4_bool<false> F;

  // ---- prologue of the first tile: its steps 0, 1, 2 (ring positions 0, 1, 2) and the fragments of step 0
  set_nxt(xstart + ti, true);
#pragma unroll
  for (int s = 0; s < (DIST4 ? 4 : 3); ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(nxt_a, j, smem + (s << 15), s * BK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_b(nxt_b, j, smem + (s << 15), s * BK * 2);
  }
  if (DIST4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  W4_BARRIER();
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) bf[nt] = *(const half8*)(smem + b_base + nt * 1024);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) af[mt] = *(const half8*)(smem + a_base + mt * 1024);

  int g = 0;     // ring position of the current tile's step 0
  int tcount = 0;
  uint32_t sum_loop = 0, sum_epi = 0, sum_drain = 0;
  for (;;) {
    int tid_k = tid;
    asm volatile("" : "+v"(tid_k));
    const int wgid = xstart + ti;
    int tn_div = tiles_n;
    asm volatile("" : "+s"(tn_div));
    const int tm = wgid / tn_div, tn = wgid - tm * tn_div;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint32_t c0 = STAMPS ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) { cur_a[j] = nxt_a[j]; cur_b[j] = nxt_b[j]; }
    const int tnext = ti + nx;
    const bool has_next = tnext < xcount;
    set_nxt(xstart + (has_next ? tnext : ti), has_next);

    constexpr int D = DIST4 ? 4 : 3;
    const bool rl = tcount > 0;
    step(T{}, F{}, g, D, rl);
    for (int t = 1; t < S - D; ++t) step(F{}, F{}, g + t, t + D, rl && t < D - 1);
    for (int v = 0; v < D; ++v) step(F{}, T{}, g + S - D + v, v, false);
    g += S;
    const uint32_t c1 = STAMPS ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;

    // ---- epilogue: four passes of 64 tile rows (wave rows [32 p, 32 p + 32) of both row halves)
    char* const hs = smem + RING;
    const int lane_e = tid_k & 63, l15e = lane_e & 15, lge = lane_e >> 4;
    const int ecg = tid_k & 31, er0 = tid_k >> 5;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p > 0) W4_BARRIER();
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const int sr = wm * 32 + mi * 16 + l15e;
          const int chunk = wn * 16 + nt * 2 + (lge >> 1);
          const f32x4 a = acc[2 * p + mi][nt];
          const half4 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
          *(half4*)(hs + sr * 512 + ((chunk ^ (sr & 15)) << 4) + (lge & 1) * 8) = h;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_BARRIER();
      half8 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sr = er0 + 8 * i;
        v[i] = *(const half8*)(hs + sr * 512 + ((ecg ^ (sr & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sr = er0 + 8 * i;
        const int row = m0 + (sr >> 5) * 128 + 32 * p + (sr & 31);
        *(half8*)(C + (size_t)row * N + n0 + ecg * 8) = v[i];
      }
    }
    const uint32_t c2 = STAMPS ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
    if (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // diagnostic: how long do the epilogue's stores take to drain?
    const uint32_t c3 = STAMPS ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
    if (STAMPS && tcount > 0) { sum_loop += c1 - c0; sum_epi += c2 - c1; sum_drain += c3 - c2; }
    ++tcount;
    if (!has_next) break;
    ti = tnext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (zero-fill) DMA must not land in the next workgroup's LDS
  if (STAMPS && tid == 0) {
    stamps[blockIdx.x * 4 + 0] = sum_loop; stamps[blockIdx.x * 4 + 1] = sum_epi; stamps[blockIdx.x * 4 + 2] = sum_drain;
    stamps[blockIdx.x * 4 + 3] = tcount - 1;
  }
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
  (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dB, hB.size() * 2); (void)hipMalloc(&dC, (size_t)M * N * 2);
  const int tiles = (M / BM) * (N / BN);
  const int grid = 8 * std::min(32, (tiles + 7) / 8);
  (void)hipMalloc(&dS, (size_t)grid * 16 * 4 * 8);
  (void)hipMemset(dS, 0, (size_t)grid * 16 * 4 * 8);
  (void)hipMemset(dC, 0, (size_t)M * N * 2);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)w4p_kernel<false, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)w4p_kernel<true, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL((w4p_kernel<false, MODE>), dim3(grid), dim3(NTH), LDS_BYTES, 0, dA, dB, dC, M, N, K, dS);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((w4p_kernel<false, MODE>), dim3(grid), dim3(NTH), LDS_BYTES, 0, dA, dB, dC, M, N, K, dS);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = std::fmin(best, ms / 10);
  }
  printf("mode %d M %6d N %5d K %5d: %8.1f us  %7.1f TFLOP/s", MODE, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
  hipLaunchKernelGGL((w4p_kernel<true, MODE>), dim3(grid), dim3(NTH), LDS_BYTES, 0, dA, dB, dC, M, N, K, dS);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> st((size_t)grid * 4);
  (void)hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost);
  double sl = 0, se = 0, sd = 0, nt = 0;
  for (int b = 0; b < grid; ++b) { sl += st[b * 4]; se += st[b * 4 + 1]; sd += st[b * 4 + 2]; nt += st[b * 4 + 3]; }
  nt = std::max(nt, 1.0);
  const double ideal = (double)(K / 32) * 1024.0;
  printf("   per tile (mean, cycles): loop %7.0f (ideal %7.0f = %4.1f %%)  epilogue %6.0f  drain %6.0f\n", sl / nt, ideal, 100.0 * ideal * nt / std::max(1.0, sl),
         se / nt, sd / nt);
  int bad = 0;
  if (check) {
    std::vector<half_t> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    uint64_t s = 99;
    for (int it = 0; it < 6000; ++it) {
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
    printf("   check: worst rel err %.2e, %d bad of 6000\n", worst, bad);
  }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dS);
  return bad;
}

template <int MODE>
int suite() {
  int bad = 0;
  bad += run<MODE>(96000, 1536, 512, true);    // q|k|v: A 98 MB streamed from HBM
  bad += run<MODE>(8192, 8192, 512, false);    // 1024 tiles, 8 MB + 8 MB
  bad += run<MODE>(96000, 512, 2048, false);   // fc2
  bad += run<MODE>(96000, 512, 1536, false);   // q|k|v dgrad
  bad += run<MODE>(96000, 3840, 1280, false);  // large-v2 q|k|v
  bad += run<MODE>(96000, 1280, 1280, false);  // large-v2 out_proj
  bad += run<MODE>(96000, 5120, 1280, false);  // large-v2 fc1
  bad += run<MODE>(96000, 1280, 5120, false);  // large-v2 fc2
  bad += run<MODE>(8192, 8192, 8192, false);   // the guide's shape
  return bad;
}

int main() {
  int bad = 0;
  bad += suite<0>();
  bad += suite<4>();
  printf(bad ? "FAILED\n" : "OK\n");
  return bad != 0;
}
