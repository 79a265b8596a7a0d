// ns_gemm NT hot path: 128x128 tile, K streamed through a 4-stage LDS ring of 32-deep slices filled
// by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass), three slices in flight
// behind a COUNTED s_waitcnt vmcnt(N) and ONE raw s_barrier per slice (cdna guide §5 "Pipelining
// across barriers").  64 KiB of LDS per block -> 2 blocks per CU, so one block's epilogue overlaps
// the other's main loop; the fp32 epilogue tile reuses the ring.
//
// LDS slice image: [128 rows][32 k] fp16 = 64-B rows, lane-linear per DMA instruction (16 rows x 64 B
// = 1 KiB).  The 16-B chunk index is XOR-swizzled with (row>>2)&3 on the SOURCE address and on the
// ds_read_b128 address (never on the LDS destination), which makes the fragment reads (lane = row,
// same chunk) conflict-free across the 16-lane ds_read_b128 groups.
#include "ns_gemm_epi.h"
#include <mutex>

namespace {

constexpr int BN = 128, BKS = 32, NTH = 256;     // BM = 128 or 64 (template): 64-row tiles when 128-row tiles leave most CUs idle
// ring depth: 4 slices, one barrier per slice (the dropout variants, whose mask sits between two slices; the 128-row tile; 64-row launches
// of >= 1024 workgroups), or 6 slices walked in PAIRS (64-row launches that leave a CU <= ~3 workgroups) -- one barrier, one counted wait
// and one batch of fragment reads per 64-deep pair: a slice of these mid-size launches is 128 MFMA cycles behind ~600 cycles of wait +
// barrier + DMA issue + ds_read latency (880 cycles per slice measured at K = 2048), so halving the number of synchronisation points is
// worth more than the third slice of look-ahead it costs -- as long as the LDS it takes does not cost a resident workgroup that would
// have had work (the rule and its measurements: ns_gemm_ring_launch)
template <bool PAIRS> struct ring_depth { static constexpr int value = PAIRS ? 6 : 4; };
constexpr int B_BYTES = BN * BKS * 2;          // 8 KiB: one B slice

__device__ __forceinline__ int lds_off32(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const half_t* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K tails: chunks beyond the valid K range are fetched from a block of zeros, so the MFMA loop is branch-free
__device__ __attribute__((aligned(16))) const uint32_t ns_zero_chunk128[4] = {0, 0, 0, 0};

template <bool DROP, int BM, bool PAIRS>
__global__ __launch_bounds__(NTH, 2) void ns_gemm_ring_kernel(const ns_gemm_desc p) {
  static_assert(!(DROP && PAIRS), "the dropout mask sits between two single slices");
  constexpr int MI = BM / 64;                  // 32-row MFMA tiles per wave along M (wave grid 2 x 2: wave tile BM/2 x 64)
  constexpr int A_BYTES = BM * BKS * 2;        // one A slice: 8 or 4 KiB
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int AP = BM / 64;                  // 1-KiB A pieces per wave and slice (B: always 2)
  constexpr int NST = ring_depth<PAIRS>::value;
  const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);   // wave-uniform: one scalar load at entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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

  // split-K (gridDim.y > 1; plain fp32 output only): K = 51 968 of the LM-head dgrad has 22 output tiles -- one K range
  // per workgroup row, partial tiles summed by fp32 atomics straight from the accumulators (a 32x32 accumulator register
  // is two 128-B row segments per wave instruction: the full-rate atomic shape)
  int kbeg = 0, Kloc = p.K;
  if (gridDim.y > 1) {
    const int per = ((p.K + (int)gridDim.y - 1) / (int)gridDim.y + BKS - 1) / BKS * BKS;
    kbeg = blockIdx.y * per;
    Kloc = max(0, min(p.K, kbeg + per) - kbeg);
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- DMA source pointers: this wave fills rows [32*wave, 32*wave+32) of the B slice with two 1-KiB instructions and rows
  // [16*AP*wave, +16*AP) of the A slice with AP; lane -> (row = 16*piece + lane/4, LDS chunk' = lane%4) <- global chunk' ^ swz(row)
  const half_t* a_src[2];
  const half_t* b_src[2];
  const half_t* a2_src[2] = {nullptr, nullptr};
  const half_t* b2_src[2] = {nullptr, nullptr};
  int a_chunk[2], b_chunk[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int brow = 16 * (wave * 2 + jj) + (lane >> 2);
    b_chunk[jj] = (lane & 3) ^ ((brow >> 2) & 3);
    b_src[jj] = (const half_t*)p.B + (long long)min(n0 + brow, p.N - 1) * p.bm.ld + b_chunk[jj] * 8 + kbeg;
    const int arow = 16 * (wave * AP + (jj < AP ? jj : 0)) + (lane >> 2);
    a_chunk[jj] = (lane & 3) ^ ((arow >> 2) & 3);
    a_src[jj] = (const half_t*)p.A + ns_rm_off64(p.am, min(m0 + arow, p.M - 1)) + a_chunk[jj] * 8 + kbeg;
  }
  if (p.K2 > 0) {
    const int goff = p.a2_ngroup > 0 ? (n0 / p.a2_ngroup) * p.K2 : 0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int brow = 16 * (wave * 2 + jj) + (lane >> 2);
      const int arow = 16 * (wave * AP + (jj < AP ? jj : 0)) + (lane >> 2);
      a2_src[jj] = (const half_t*)p.A2 + ns_rm_off64(p.am2, min(m0 + arow, p.M - 1)) + goff + a_chunk[jj] * 8;
      b2_src[jj] = (const half_t*)p.B2 + (long long)min(n0 + brow, p.N - 1) * p.ldb2 + b_chunk[jj] * 8;
    }
  }

  const int steps1 = (Kloc + BKS - 1) / BKS;
  const int steps2 = (p.K2 + BKS - 1) / BKS;
  const int nsteps = steps1 + steps2;
  const bool seg2_first = DROP && steps2 > 0;   // dgrad with LoRA dropout: (A2,B2) first, mask, then the main product

  auto step_info = [&](int s, bool& is2, int& k0, int& klen) __attribute__((always_inline)) {
    if (seg2_first) { is2 = s < steps2; k0 = (is2 ? s : s - steps2) * BKS; }
    else { is2 = s >= steps1; k0 = (is2 ? s - steps1 : s) * BKS; }
    klen = (is2 ? p.K2 : Kloc) - k0;
    klen = klen > BKS ? BKS : klen;
  };
  auto issue = [&](int s) __attribute__((always_inline)) {
    bool is2; int k0, klen; step_info(s, is2, k0, klen);
    char* const dst = smem + (s % NST) * STAGE_BYTES;
#pragma unroll
    for (int jj = 0; jj < AP; ++jj) {
      const bool ok = a_chunk[jj] * 8 < klen;
      glds16(ok ? (is2 ? a2_src[jj] : a_src[jj]) + k0 : (const half_t*)ns_zero_chunk128, dst + (wave * AP + jj) * 1024);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const bool ok = b_chunk[jj] * 8 < klen;
      glds16(ok ? (is2 ? b2_src[jj] : b_src[jj]) + k0 : (const half_t*)ns_zero_chunk128, dst + A_BYTES + (wave * 2 + jj) * 1024);
    }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* as = smem + buf * STAGE_BYTES;
    const char* bs = as + A_BYTES;
    half8 af[2][MI], bf[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[s][i] = *(const half8*)(as + lds_off32(wm * (BM / 2) + i * 32 + lr, 2 * s + lh));
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[s][j] = *(const half8*)(bs + lds_off32(wn * 64 + j * 32 + lr, 2 * s + lh));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
  };

  if constexpr (PAIRS) {
    // ---- slices in pairs: pairs p + 1 in flight while pair p is multiplied, pair p + 2 requested into pair p - 1's slots behind the barrier
    // (slices past the end -- an odd count, a one-pair product -- arrive as zeros: issue() fetches the zero chunk for them)
    const int npairs = (nsteps + 1) >> 1;
    issue(0); issue(1); issue(2); issue(3);
    for (int pr = 0; pr < npairs; ++pr) {
      if (pr + 1 < npairs) wait_vmcnt<2 * (AP + 2)>();      // the newer pair's pieces may stay in flight
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // every wave's DMA of pair pr has landed; every wave has finished pair pr - 1
      asm volatile("" ::: "memory");
      if (pr + 2 < npairs) { issue(2 * pr + 4); issue(2 * pr + 5); }
      compute((2 * pr) % NST);
      compute((2 * pr + 1) % NST);
    }
  } else {
  // ---- prologue: NST-1 slices in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nsteps) issue(s);

  for (int s = 0; s < nsteps; ++s) {
    // slices s+1 .. min(s+NST-2, nsteps-1) may stay in flight
    const int ahead = min(NST - 2, nsteps - 1 - s);
    if (ahead >= 2) wait_vmcnt<2 * (AP + 2)>();      // AP + 2 DMA instructions per slice and wave
    else if (ahead == 1) wait_vmcnt<AP + 2>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // every wave's slice-s DMA has landed; every wave finished slice s-1
    asm volatile("" ::: "memory");
    if (s + NST - 1 < nsteps) issue(s + NST - 1);   // refills the buffer slice s-1 lived in
    compute(s % NST);
    if (DROP && seg2_first && s == steps2 - 1) {
      const uint32_t drop_thr = ns_drop_thr8(p.drop_p);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t row = (uint32_t)(m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
            const uint32_t col = (uint32_t)(n0 + wn * 64 + j * 32 + lr);
            acc[i][j][r] = ns_keep_el(dseed, row, col, drop_thr) ? acc[i][j][r] : 0.f;
          }
    }
  }
  }

  if (gridDim.y > 1) {
    const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
    // the bias values FIRST, settled once: with the load next to its use, every atomic sat behind its own s_waitcnt vmcnt(0) (the
    // use is inside a row-predicated block, where hipcc re-waits), i.e. behind the completion of every atomic before it -- a chain of
    // 16 x MI x 2 memory round trips per lane (found in the ISA in round 4: the LM-head dgrad ran at 0.2 of the MFMA peak)
    float bz[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = min(n0 + wn * 64 + j * 32 + lr, p.N - 1);
      bz[j] = (p.bias && blockIdx.y == 0) ? p.bias[col] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(bz[j]));
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + lr;
        if (col >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < p.M) atomicAdd(p.C32 + (long long)row * p.ldc32 + col, acc[i][j][r] * alpha + bz[j]);
        }
      }
    return;
  }
  // ---- epilogue through LDS (the ring is dead: no DMA outstanding, wait for the last readers)
  __syncthreads();
  float* const ct = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        ct[row * BN + wn * 64 + j * 32 + lr] = acc[i][j][r];
      }
  __syncthreads();
  ns_nt_epilogue<BM, BN, NTH>(p, ct, m0, n0, tid);
}

}  // namespace

extern int g_ns_ab_flag;

// called by ns_gemm() for NT descriptors with N > 96 and no A-operand dropout (arguments already validated)
int ns_gemm_ring_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tn = (d->N + BN - 1) / BN;
  const int tiles128 = ((d->M + 127) / 128) * tn;
  const int splits = ((d->flags & NS_GEMM_TN) || d->splits < 1) ? 1 : d->splits;   // > 1 only for the plain-C32 form (ns_gemm checks)
  // 64-row tiles when 128-row tiles cover less than ~half of the CUs (the decoder-side GEMMs of a training step: 2816 rows x 512
  // columns = 88 tiles): such launches are bound by what ONE CU can fetch, and 176 workgroups move 3/4 of the bytes per CU
  const bool small = splits == 1 && tiles128 <= 384 && d->M > 64;
  const bool dropv = d->drop_p > 0.f;
  // Slices in pairs (six-deep ring) halve the synchronisation points of a workgroup but cost LDS: 72 KiB for the 64-row tile (two workgroups per
  // CU instead of three), 96 KiB for the 128-row tile (ONE instead of two).  Measured in round 6 (tools/probe/ring_depth_ab.py,
  // profiles/r6_probe_ring_depth.log): the 128-row tile is 11-26 % FASTER with four single slices on every shape (the split-K
  // LM-head input gradient 307 -> 264 us), the 64-row tile is 8-18 % faster in pairs while a CU holds at most ~3 of its workgroups
  // (2 816 rows x N <= 1024: 8.8 against 9.7 us at N = K = 512, 21.6 / 25.5 at K = 2048) and 7-10 % slower beyond (N = 1536 / 2048:
  // 1 056 / 1 408 workgroups, where the third resident workgroup counts).  A/B flags 1 / 2 invert the choice for the 128- / 64-row tile.
  const int tiles64 = ((d->M + 63) / 64) * tn;
  const bool pairs = !dropv && (small ? ((tiles64 < 1024) != ((g_ns_ab_flag & 2) != 0)) : (g_ns_ab_flag & 1) != 0);
  const size_t lds = (size_t)(pairs ? 6 : 4) * (size_t)((small ? 64 : 128) * BKS * 2 + B_BYTES);
  static ns_dev_once attr_once;      // kernel attributes, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)ns_gemm_ring_kernel<false, 128, true>, (const void*)ns_gemm_ring_kernel<false, 128, false>,
                                   (const void*)ns_gemm_ring_kernel<true, 128, false>, (const void*)ns_gemm_ring_kernel<false, 64, true>,
                                   (const void*)ns_gemm_ring_kernel<false, 64, false>, (const void*)ns_gemm_ring_kernel<true, 64, false>},
                       96 * 1024, "ns_gemm (ring)"))
    return NS_ERR_HIP;
  if (small) {
    const int tiles = ((d->M + 63) / 64) * tn;
    if (dropv) hipLaunchKernelGGL((ns_gemm_ring_kernel<true, 64, false>), dim3(tiles, 1), dim3(NTH), lds, st, *d);
    else if (pairs) hipLaunchKernelGGL((ns_gemm_ring_kernel<false, 64, true>), dim3(tiles, 1), dim3(NTH), lds, st, *d);
    else hipLaunchKernelGGL((ns_gemm_ring_kernel<false, 64, false>), dim3(tiles, 1), dim3(NTH), lds, st, *d);
  } else {
    if (dropv) hipLaunchKernelGGL((ns_gemm_ring_kernel<true, 128, false>), dim3(tiles128, splits), dim3(NTH), lds, st, *d);
    else if (pairs) hipLaunchKernelGGL((ns_gemm_ring_kernel<false, 128, true>), dim3(tiles128, splits), dim3(NTH), lds, st, *d);
    else hipLaunchKernelGGL((ns_gemm_ring_kernel<false, 128, false>), dim3(tiles128, splits), dim3(NTH), lds, st, *d);
  }
  return 0;
}
