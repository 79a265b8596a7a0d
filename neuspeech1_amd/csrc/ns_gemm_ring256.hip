// ns_gemm NT, large-M form: 256x256 tile, 8 waves (2 x 4), wave tile 128x64 = 4x2 v_mfma_f32_32x32x16_f16.
//
// Why 256^2: the 128^2 ring kernel runs at the per-CU operand-delivery bound (rocprofv3 TCC counters: ~10 TB/s of
// L2->LDS traffic chip-wide, 0.54 PFLOP/s at 1 byte per 32 flop).  A 256x256 tile needs 1 byte per 64 flop, and a
// 64-deep K slice makes every LDS-DMA piece 8 rows x 128 B = whole cache lines instead of half lines.
//
// LDS: two 64 KiB stages (A 256x64 | B 256x64, fp16, 128-B rows), filled by global_load_lds_dwordx4 with the
// chunk XOR-swizzle (row>>1)&7 applied on the SOURCE address and on the ds_read_b128 address.  Stage s+1 is in
// flight while stage s is computed (one s_waitcnt vmcnt(0) + one raw s_barrier per 64-deep slice, 32 MFMAs per
// wave between them).  Epilogue: the fp32 tile goes through LDS in four 64-row chunks (the ring is dead by then)
// so all global accesses are row vectors (shared ns_nt_epilogue).
#include "ns_gemm_epi.h"
#include <mutex>

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTH = 512;
constexpr int OP_BYTES = BM * BK * 2;        // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * OP_BYTES;    // 64 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__device__ __forceinline__ void glds16(const half_t* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}

// K tails: a 16-B chunk beyond the valid K range of its segment is fetched from this block of zeros instead, so
// every slice runs all four k-substeps with no branch in the MFMA loop.
__device__ __attribute__((aligned(16))) const uint32_t ns_zero_chunk[4] = {0, 0, 0, 0};

template <bool DROP>
__global__ __launch_bounds__(NTH, 2) void ns_gemm_ring256_kernel(const ns_gemm_desc p) {
  const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);   // wave-uniform: one scalar load at entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int wgid;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA sources: wave w fills rows [32w, 32w+32) of both operand tiles, four 1-KiB pieces (8 rows x 128 B) each
  // element offsets (32-bit) from the operand bases; host code guarantees they fit
  uint32_t a_src[4], b_src[4], a2_src[4] = {0, 0, 0, 0}, b2_src[4] = {0, 0, 0, 0};
  int my_chunk[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int row = 8 * (wave * 4 + jj) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    my_chunk[jj] = chunk;
    a_src[jj] = (uint32_t)(ns_rm_off64(p.am, min(m0 + row, p.M - 1)) + chunk * 8);
    b_src[jj] = (uint32_t)((long long)min(n0 + row, p.N - 1) * p.bm.ld + chunk * 8);
  }
  if (p.K2 > 0) {
    const int goff = p.a2_ngroup > 0 ? (n0 / p.a2_ngroup) * p.K2 : 0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = 8 * (wave * 4 + jj) + (lane >> 3);
      a2_src[jj] = (uint32_t)(ns_rm_off64(p.am2, min(m0 + row, p.M - 1)) + goff + my_chunk[jj] * 8);
      b2_src[jj] = (uint32_t)((long long)min(n0 + row, p.N - 1) * p.ldb2 + my_chunk[jj] * 8);
    }
  }

  const int steps1 = (p.K + BK - 1) / BK;
  const int steps2 = (p.K2 + BK - 1) / BK;
  const int nsteps = steps1 + steps2;
  const bool seg2_first = DROP && steps2 > 0;

  auto step_info = [&](int s, bool& is2, int& k0, int& klen) __attribute__((always_inline)) {
    if (seg2_first) { is2 = s < steps2; k0 = (is2 ? s : s - steps2) * BK; }
    else { is2 = s >= steps1; k0 = (is2 ? s - steps1 : s) * BK; }
    klen = (is2 ? p.K2 : p.K) - k0;
    klen = klen > BK ? BK : klen;
  };
  auto issue = [&](int s) __attribute__((always_inline)) {
    bool is2; int k0, klen; step_info(s, is2, k0, klen);
    char* const dst = smem + (s & 1) * STAGE_BYTES + (wave * 4) * 1024;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const bool ok = my_chunk[jj] * 8 < klen;
      const half_t* pa = (const half_t*)(is2 ? p.A2 : p.A) + (size_t)(is2 ? a2_src[jj] : a_src[jj]) + k0;
      const half_t* pb = (const half_t*)(is2 ? p.B2 : p.B) + (size_t)(is2 ? b2_src[jj] : b_src[jj]) + k0;
      glds16(ok ? pa : (const half_t*)ns_zero_chunk, dst + jj * 1024);
      glds16(ok ? pb : (const half_t*)ns_zero_chunk, dst + OP_BYTES + jj * 1024);
    }
  };
  // fragment address of k-substep s = base ^ (s << 5): the swizzle term ((2s+lh) ^ sw) << 4 splits into a lane
  // constant and (s ^ (sw>>1)) << 5, and rows 32 apart share the swizzle, so tiles i / j are immediate offsets
  const int a_base = lds_off(wm * 128 + lr, lh);
  const int b_base = lds_off(wn * 64 + lr, lh) + OP_BYTES;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE_BYTES;
    half8 af[2][4], bf[2][2];
    auto frags = [&](int s, int b) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[b][i] = *(const half8*)(st + (a_base ^ (s << 5)) + i * 4096);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[b][j] = *(const half8*)(st + (b_base ^ (s << 5)) + j * 4096);
    };
    auto mma = [&](int b) __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[b][i], bf[b][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // straight line: counted lgkmcnt waits, the reads of substep s+1 overlap the MFMAs of substep s
    frags(0, 0);
    frags(1, 1); mma(0);
    frags(2, 0); mma(1);
    frags(3, 1); mma(0);
    mma(1);
  };

  if (nsteps > 0) issue(0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // stage s landed for every wave; every wave is done with stage s-1
    asm volatile("" ::: "memory");
#ifdef NS_P8_STAMPS     // diagnostic build (tools/probe/gemm_ablate.py): flag bit 30 = no loads, bit 29 = no MFMA
    if (s + 1 < nsteps && !(p.flags & (1 << 30))) issue(s + 1);
    if (!(p.flags & (1 << 29))) compute(s & 1);
#else
    if (s + 1 < nsteps) issue(s + 1);
    compute(s & 1);
#endif
    if (DROP && seg2_first && s == steps2 - 1) {
      const uint32_t drop_thr = ns_drop_thr8(p.drop_p);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t row = (uint32_t)(m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
            const uint32_t col = (uint32_t)(n0 + wn * 64 + j * 32 + lr);
            acc[i][j][r] = ns_keep_el(dseed, row, col, drop_thr) ? acc[i][j][r] : 0.f;
          }
    }
  }

  // ---- epilogue: four 64-row chunks through LDS (64 KiB each)
  float* const ct = (float*)smem;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    __syncthreads();
    if (wm == (c >> 1)) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rowl = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = (c & 1) ? acc[2 + ii][j][r] : acc[ii][j][r];
            ct[rowl * BN + wn * 64 + j * 32 + lr] = v;
          }
    }
    __syncthreads();
    ns_nt_epilogue<64, BN, NTH>(p, ct, m0 + c * 64, n0, tid);
  }
}

}  // namespace

int ns_gemm_ring256_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  const size_t lds = 2 * STAGE_BYTES;
  static ns_dev_once attr_once;      // kernel attributes, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)ns_gemm_ring256_kernel<false>, (const void*)ns_gemm_ring256_kernel<true>}, (int)lds,
                       "ns_gemm (ring256)"))
    return NS_ERR_HIP;
  if (d->drop_p > 0.f) hipLaunchKernelGGL(ns_gemm_ring256_kernel<true>, dim3(tiles), dim3(NTH), lds, st, *d);
  else hipLaunchKernelGGL(ns_gemm_ring256_kernel<false>, dim3(tiles), dim3(NTH), lds, st, *d);
  return 0;
}
