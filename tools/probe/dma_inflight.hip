// What ONE CU can pull through LDS-DMA, as a function of the bytes it keeps in flight and of where the data sits.
// One 512-thread workgroup per CU; every wave keeps D pieces of 1 KiB (global_load_lds_dwordx4: 64 lanes x 16 B) in flight
// behind a counted s_waitcnt vmcnt(D - 1), so a CU has 8 * D KiB outstanding.  Sources:
//   l2    every workgroup of an XCD walks the SAME 2 MiB window again and again (L2 hits after the first pass)
//   mall  all workgroups walk a 128 MiB buffer in a workgroup-private order, several passes (Infinity Cache)
//   hbm   all workgroups stream disjoint slices of a 4 GiB buffer once
// Access shape = the GEMM's: a piece is 8 rows x 128 B of a row-major matrix with 1-KiB rows (K = 512 fp16), or one
// contiguous 1 KiB (ROWS = 0).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_inflight tools/probe/dma_inflight.hip && /tmp/dma_inflight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int D>
__device__ __forceinline__ void wait_d() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory"); }

// base: this workgroup's window; window_bytes: size of the window it cycles through; pieces: pieces per wave
template <int D, int RB>
__global__ __launch_bounds__(512) void k(const char* __restrict__ buf, size_t wg_stride, size_t window_bytes, int pieces, int xcd_shared,
                                         unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const size_t wg = xcd_shared ? (blockIdx.x & 7) : blockIdx.x;        // l2 mode: one window per XCD
  const char* base = buf + wg * wg_stride;
  // piece p of this wave sits at byte (p * 8 + wave) * 1024 of the window (the 8 waves take consecutive KiB)
  constexpr bool ROWS = RB != 0;
  constexpr int NR = RB ? 1024 / RB : 1;                                // rows per piece: 8 (128 B of each) or 16 (64 B of each: the 32-deep
                                                                        // steps of ns_gemm_rowln / ns_gemm_p4)
  constexpr int LPR = RB ? RB / 16 : 64;                                // lanes per row
  const size_t lane_off = ROWS ? (size_t)(lane / LPR) * 1024 + (lane % LPR) * 16 : (size_t)lane * 16;
  size_t pos = (size_t)wave * (ROWS ? NR * 1024 : 1024);
  const size_t step = ROWS ? 8 * NR * 1024 : 8 * 1024;                  // ROWS: a piece spans NR rows x 1 KiB, takes RB bytes of each
  size_t col = 0;                                                       // ROWS: RB-byte column inside the 1-KiB rows
  auto issue = [&](int slot) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (wave * D + slot) * 1024);
    glds16(base + pos + (ROWS ? col : 0) + lane_off, dst);
    if (ROWS) {
      col += RB;
      if (col == 1024) { col = 0; pos += step; }
    } else {
      pos += step;
    }
    if (pos + (ROWS ? NR * 1024 : 1024) > window_bytes) pos = (size_t)wave * (ROWS ? NR * 1024 : 1024);
  };
#pragma unroll
  for (int i = 0; i < D; ++i) issue(i);
  for (int p = D; p < pieces; p += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      wait_d<D>();          // the oldest piece has landed: its slot is free again
      issue(i);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (pieces < 0) sink[tid] = *(unsigned*)(smem + tid * 4);
}

template <int D, int RB>
double run(const char* buf, size_t wg_stride, size_t window, int pieces, int shared, unsigned* sink) {
  hipFuncSetAttribute((const void*)k<D, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * D * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<D, RB>), dim3(256), dim3(512), 8 * D * 1024, 0, buf, wg_stride, window, pieces, shared, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<D, RB>), dim3(256), dim3(512), 8 * D * 1024, 0, buf, wg_stride, window, pieces, shared, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)pieces * 8 * 1024 / (ms * 1e-3) / 1e9;   // GB/s per CU
}

template <int RB>
void sweep(const char* name, const char* buf, size_t wg_stride, size_t window, int pieces, int shared, unsigned* sink) {
  printf("%-22s %s  GB/s per CU at 8/16/32/64/128 KiB in flight:", name, RB == 128 ? "8x128B rows" : (RB == 64 ? "16x64B rows" : "1 KiB contig"));
  printf(" %6.1f", run<1, RB>(buf, wg_stride, window, pieces, shared, sink));
  printf(" %6.1f", run<2, RB>(buf, wg_stride, window, pieces, shared, sink));
  printf(" %6.1f", run<4, RB>(buf, wg_stride, window, pieces, shared, sink));
  printf(" %6.1f", run<8, RB>(buf, wg_stride, window, pieces, shared, sink));
  printf(" %6.1f\n", run<16, RB>(buf, wg_stride, window, pieces, shared, sink));
}

int main() {
  const size_t total = (size_t)4 << 30;
  char* buf;
  unsigned* sink;
  if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4096);
  hipMemset(buf, 1, total);
  hipDeviceSynchronize();
  // l2: 2 MiB window per XCD (8 windows), 8192 pieces per wave = 64 MiB per CU
  sweep<0>("l2 (2 MiB per XCD)", buf, (size_t)2 << 20, (size_t)2 << 20, 8192, 1, sink);
  sweep<128>("l2 (2 MiB per XCD)", buf, (size_t)2 << 20, (size_t)2 << 20, 8192, 1, sink);
  sweep<64>("l2 (2 MiB per XCD)", buf, (size_t)2 << 20, (size_t)2 << 20, 8192, 1, sink);
  // mall: every workgroup cycles through its own 512 KiB window (256 x 512 KiB = 128 MiB in total), many passes
  sweep<0>("mall (128 MiB total)", buf, (size_t)512 << 10, (size_t)512 << 10, 8192, 0, sink);
  sweep<128>("mall (128 MiB total)", buf, (size_t)512 << 10, (size_t)512 << 10, 8192, 0, sink);
  sweep<64>("mall (128 MiB total)", buf, (size_t)512 << 10, (size_t)512 << 10, 8192, 0, sink);
  // hbm: disjoint 16 MiB slices of the 4 GiB buffer, streamed once (2048 pieces per wave = 16 MiB per CU)
  sweep<0>("hbm (4 GiB, once)", buf, (size_t)16 << 20, (size_t)16 << 20, 2048, 0, sink);
  sweep<128>("hbm (4 GiB, once)", buf, (size_t)16 << 20, (size_t)16 << 20, 2048, 0, sink);
  sweep<64>("hbm (4 GiB, once)", buf, (size_t)16 << 20, (size_t)16 << 20, 2048, 0, sink);
  return 0;
}
