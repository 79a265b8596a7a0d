// Weight-gradient GEMM, TN form: C[i][j] (+)= alpha * sum_m A[m][i] * B[m][j]   (dW = dY^T X).
// Both operands are reduction-major in HBM.  They are staged ROW-MAJOR into LDS with 16-B coalesced loads
// (a wave covers 4 rows x 256 B) and the k-contiguous MFMA fragments are produced by the gfx950 transposed LDS
// read ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of a 4-row block), so no register or
// global-side transpose exists.  LDS row stride = 320 B for 128-column tiles / 64 B for 32-column tiles: the four
// rows a half wave touches per transposed read then tile the 64 banks exactly (conflict-free).
// Tiles: <128,128> (conv weight grads), <32,128> (LoRA dA: r x K), <128,32> (LoRA dB: N x r).  The reduction is split
// over gridDim.z and accumulated with fp32 atomics in the full-rate shape (128 contiguous bytes per half wave).
// Pipeline (round 5): the operands of THREE 64-row steps are in flight in registers (three register stages, the loop unrolled by
// three so that they are named statically), two LDS buffers, one bare s_barrier per step behind the wave's own LDS stores
// (__syncthreads() is a fence: hipcc drains the vector-memory queue in front of it, i.e. every step waited for the loads it had
// just issued -- a memory round trip per 64 rows, 3.3 TB/s over the adapters' dA launches).  The store of step t + 1 waits for
// ITS loads only (the queue is in order; hipcc counts).  Same products in the same order: results are bitwise those of round 4.
#include "ns_common.h"
#include <mutex>

namespace {

constexpr int BKM = 64, NTH = 256;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v lds_s4;

template <int BW> struct RowStride { static constexpr int bytes = BW == 128 ? 320 : 64; };

__device__ __forceinline__ half8 tr_frag(const char* tile, int stride, int m0, int c0, int lane) {
  // rows m0..m0+7 (two 4-row blocks), 16 columns starting at c0 + 16*(g&1); returns the 8 k-values of this lane's column
  const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
  const char* a = tile + (m0 + 8 * (g >> 1) + q) * stride + (c0 + 16 * (g & 1) + 4 * p) * 2;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 4 * stride));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8, r);
}

template <int BI, int BJ, bool DROP>
__global__ __launch_bounds__(NTH, 2) void ns_gemm_tn_kernel(const ns_gemm_desc p) {
  const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);   // wave-uniform: one scalar load at entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SA = RowStride<BI>::bytes, SB = RowStride<BJ>::bytes;
  constexpr int A_BYTES = BKM * SA, B_BYTES = BKM * SB;
  // wave layout: 128x128 -> 2x2 waves of 64x64; 32x128 -> 1x4 of 32x32; 128x32 -> 4x1 of 32x32
  constexpr int WI = BI == 128 && BJ == 128 ? 64 : 32, WJ = WI;
  constexpr int TI = WI / 32, TJ = WJ / 32;
  constexpr int CA = BI / 8, CB = BJ / 8;                 // 16-B chunks per row
  constexpr int LA = BKM * CA / NTH, LB = BKM * CB / NTH;  // loads per thread per K-step (4 or 1)
  char* const As = smem;
  char* const Bs = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = (BI == 128 && BJ == 128) ? (wave >> 1) : (BI == 128 ? wave : 0);
  const int wj = (BI == 128 && BJ == 128) ? (wave & 1) : (BI == 128 ? 0 : wave);
  const int lr = lane & 31, lh = lane >> 5;

  const int tiles_j = (p.N + BJ - 1) / BJ;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
  const int i0 = ti * BI, j0 = tj * BJ;

  const int chunk = (((p.K + p.splits - 1) / p.splits) + BKM - 1) / BKM * BKM;
  const int k_begin = blockIdx.z * chunk;
  const int k_end = min(p.K, k_begin + chunk);
  const int nsteps = (max(k_end - k_begin, 0) + BKM - 1) / BKM;
  int seg = 0, within = k_begin;
  if (p.am.seg_rows > 0) { seg = k_begin / p.am.seg_rows; within = k_begin - seg * p.am.seg_rows; }

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const uint32_t drop_thr = DROP ? ns_drop_thr8(p.drop_p) : 0u;

  // column chunk of this thread (clamped into the valid range: redundant but in-bounds, results discarded)
  const int ca = tid % CA, cb = tid % CB;
  const int cola = min(i0 + ca * 8, p.M - 8), colb = min(j0 + cb * 8, p.N - 8);

  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  struct stage { u32x4 a[LA]; u32x4 b[LB]; };
  stage R[3];
  auto row_off = [&](const ns_rowmap& map, int rl) __attribute__((always_inline)) -> long long {
    if (map.seg_rows > 0) {
      int w = within + rl, s = seg;
      if (w >= map.seg_rows) { w -= map.seg_rows; s += 1; }
      return (long long)s * map.seg_stride + (long long)w * map.ld;
    }
    return (long long)(within + rl) * map.ld;
  };
  // load: the 16-B pieces of one step, UNCONDITIONALLY (a row past the end of the range re-reads the range's last row: a
  // conditional load is a branch, and behind branches hipcc waits for an empty vector-memory queue instead of counting);
  // store: those rows become zeros, the dropout mask is applied, the pieces go to LDS
  auto load = [&](int step, stage& rg) __attribute__((always_inline)) {
    const int klen = min(BKM, k_end - k_begin - step * BKM);
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int rl = min(tid / CA + it * (NTH / CA), klen - 1);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rg.a[it]) : "v"((const half_t*)p.A + row_off(p.am, rl) + cola) : "memory");
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int rl = min(tid / CB + it * (NTH / CB), klen - 1);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rg.b[it]) : "v"((const half_t*)p.B + row_off(p.bm, rl) + colb) : "memory");
    }
    within += BKM;
    if (p.am.seg_rows > 0 && within >= p.am.seg_rows) { within -= p.am.seg_rows; seg += 1; }
  };
  // The loads are inline assembly (hipcc's own counting degrades to vmcnt(0) in this control flow: found in the ISA), so the wait in
  // front of a stage's store is written here: the queue is in order, `later` = the stages requested after this one that may stay
  // in flight (block-uniform).  The empty asm statements tie the stage's registers to the wait: no use may be scheduled above it.
  auto landed = [&](stage& rg, int later) __attribute__((always_inline)) {
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (LA + LB)) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LA + LB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < LA; ++it) asm volatile("" : "+v"(rg.a[it]));
#pragma unroll
    for (int it = 0; it < LB; ++it) asm volatile("" : "+v"(rg.b[it]));
  };
  auto store = [&](int buf, stage& rg, int step) __attribute__((always_inline)) {
    landed(rg, min(2, nsteps - 1 - step));
    const int klen = min(BKM, k_end - k_begin - step * BKM);
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int rl = tid / CA + it * (NTH / CA);
      *(u32x4*)(As + buf * A_BYTES + rl * SA + ca * 16) = rl < klen ? rg.a[it] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int rl = tid / CB + it * (NTH / CB);
      u32x4 v = rl < klen ? rg.b[it] : u32x4{0u, 0u, 0u, 0u};
      if (DROP) {
        const uint32_t grow = (uint32_t)(k_begin + step * BKM + rl);
        uint32_t m[4];
        ns_keep_masks(ns_drop_word(dseed, grow, (uint32_t)colb >> 2), drop_thr, m[0], m[1]);
        ns_keep_masks(ns_drop_word(dseed, grow, ((uint32_t)colb >> 2) + 1), drop_thr, m[2], m[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] &= m[e];
      }
      *(u32x4*)(Bs + buf * B_BYTES + rl * SB + cb * 16) = v;
    }
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // NS_GEMM_COLSUM_A: the workgroups of the first column tile also sum the A columns over the reduction (the bias
  // gradient of a conv: column sums of d(pre), which used to be a separate pass over the same bytes): the transposed
  // fragment of lane (column, half) holds 8 reduction rows of that column
  const bool colsum = (p.flags & NS_GEMM_COLSUM_A) && tj == 0 && wj == 0;
  float csum[TI];
#pragma unroll
  for (int a = 0; a < TI; ++a) csum[a] = 0.f;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* as = As + buf * A_BYTES;
    const char* bs = Bs + buf * B_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      half8 af[TI], bf[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) af[a] = tr_frag(as, SA, 16 * s, wi * WI + a * 32, lane);
      if (colsum) {
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[a] += (float)af[a][e];
      }
#pragma unroll
      for (int b = 0; b < TJ; ++b) bf[b] = tr_frag(bs, SB, 16 * s, wj * WJ + b * 32, lane);
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  };

  if (nsteps > 0) {
    load(0, R[0]);
    if (nsteps > 1) load(1, R[1]);
    if (nsteps > 2) load(2, R[2]);
    store(0, R[0], 0);
    barrier();
    int cur = 0;
    for (int s = 0; s < nsteps; s += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = s + u;
        if (t < nsteps) {                                   // (block-uniform)
          if (t + 3 < nsteps) load(t + 3, R[u]);            // R[u] held step t: in LDS since the previous sub-step
          compute(cur);
          if (t + 1 < nsteps) store(cur ^ 1, R[(u + 1) % 3], t + 1);
          barrier();
          cur ^= 1;
        }
      }
    }
  }

  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  const bool atomic32 = p.flags & NS_GEMM_ATOMIC32;
  if (colsum) {
#pragma unroll
    for (int a = 0; a < TI; ++a) {
      const float t = csum[a] + __shfl_xor(csum[a], 32, 64);      // the two k halves of the column
      const int row = i0 + wi * WI + a * 32 + lr;
      if (lh == 0 && row < p.M) atomicAdd(p.H32 + row, t * alpha);
    }
  }
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int col = j0 + wj * WJ + b * 32 + lr;
      if (col < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + wi * WI + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < p.M) {
            float* dst = p.C32 + (long long)row * p.ldc32 + col;
            const float v = acc[a][b][r] * alpha;
            if (atomic32) atomicAdd(dst, v); else *dst = v;
          }
        }
      }
    }
}

template <int BI, int BJ, bool DROP>
int launch_tn(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles = ((d->M + BI - 1) / BI) * ((d->N + BJ - 1) / BJ);
  const size_t lds = 2 * (size_t)BKM * (RowStride<BI>::bytes + RowStride<BJ>::bytes);
  static ns_dev_once attr_once;      // kernel attribute, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)ns_gemm_tn_kernel<BI, BJ, DROP>}, (int)lds, "ns_gemm (tn)")) return NS_ERR_HIP;
  hipLaunchKernelGGL((ns_gemm_tn_kernel<BI, BJ, DROP>), dim3(tiles, 1, d->splits), dim3(NTH), lds, st, *d);
  return 0;
}

}  // namespace

// called by ns_gemm() for TN descriptors whose M, N and row strides are multiples of 8 (arguments already validated)
int ns_gemm_tn_launch(const ns_gemm_desc* d, hipStream_t st) {
  const bool drop = d->drop_p > 0.f;
  // (M in (32, 128] -- the stacked q|k|v bottleneck, 3r = 96 rows -- takes the 128-row tile: B is read, and masked, once)
  if (d->M <= 32) return drop ? launch_tn<32, 128, true>(d, st) : launch_tn<32, 128, false>(d, st);
  if (d->N <= 96) return drop ? launch_tn<128, 32, true>(d, st) : launch_tn<128, 32, false>(d, st);
  return drop ? launch_tn<128, 128, true>(d, st) : launch_tn<128, 128, false>(d, st);
}
