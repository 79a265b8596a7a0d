// ns_gemm_tn256: weight-gradient GEMM (TN form, see ns_gemm_tn.hip) for the conv stem: C[i][j] += alpha * sum_m A[m][i] B[m][j]
// with 512 x 768 / 512 x 1536 outputs and a reduction over 96 000 .. 192 000 rows (dW = d(pre)^T im2col(x)).
//
// The 128 x 128 register-staged tile of ns_gemm_tn_kernel moves 64 flop per operand byte: at ~300 TFLOP/s it is bound
// by the 2.4 GB per launch its 768 workgroups pull through L2.  Here
//   * the tile is 256 x 256 (128 flop per byte), 8 waves as 2 x 4 of 128 x 64, one workgroup per CU;
//   * both operands travel by LDS-DMA (global_load_lds_dwordx4, inline asm: see ns_lora_bwd.hip for why) into a ring of
//     four 32-row stages, issued THREE stages ahead; a stage is 32 reduction rows x 256 columns of each operand,
//     row-major, with the 16-B chunks of a row XOR-swizzled (t_swz of the row index) on the SOURCE side, which makes the
//     transposed fragment reads (ds_read_b64_tr_b16) conflict-free without padding;
//   * one counted s_waitcnt vmcnt + one raw s_barrier per stage; no staging registers;
//   * the (tile, split) pairs are laid out split-major and dealt to the XCDs in contiguous runs, so the workgroups that
//     share an L2 read the same reduction rows: each operand row crosses HBM -> L2 about once;
//   * row maps (the halo-padded activations of the k = 3 convs), the reduction split with fp32 atomics and the
//     NS_GEMM_COLSUM_A bias-gradient side output behave exactly as in ns_gemm_tn_kernel.
#include <mutex>
#include "ns_common.h"

namespace {

constexpr int T_BK = 32, T_NST = 4, T_NTH = 512, T_TILE = 256;
constexpr int T_OP_BYTES = T_BK * T_TILE * 2;      // 16 KiB: 32 rows x 512 B
constexpr int T_STAGE_BYTES = 2 * T_OP_BYTES;      // A | B
constexpr int T_LDS_BYTES = T_NST * T_STAGE_BYTES; // 128 KiB

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_s4;

__device__ __attribute__((aligned(16))) const uint32_t ns_t256_zero_chunk[4] = {0, 0, 0, 0};

// 16-B chunk c of image row r sits at chunk c ^ t_swz(r).  t_swz puts the row's low two bits into chunk bits 2-3: the four rows of one transposed
// read (ds_read_b64_tr_b16: rows m .. m + 3, four neighbouring chunks each) land in four different 64-B windows of the 256-B bank span.  With the
// plain c ^ r of before they permuted INSIDE one window: a 4-way conflict on every fragment read (SQ_LDS_BANK_CONFLICT 6.9e7 cycles per launch,
// a third of the kernel's time).
__device__ __forceinline__ int t_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int t_off(int row, int col) { return row * 512 + ((((col >> 3) ^ t_swz(row)) & 31) << 4) + ((col & 7) << 1); }

// the 8 reduction rows m0 + 8 (lane >> 5) .. of column c0 + (lane & 31): two 4-row transposed reads
__device__ __forceinline__ half8 t_frag(const char* img, int m0, int c0, int lane) {
  const int i = lane & 15, q = i >> 2, pp = i & 3, g = lane >> 4;
  const int row = m0 + 8 * (g >> 1) + q, col = c0 + 16 * (g & 1) + 4 * pp;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + t_off(row, col)));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + t_off(row + 4, col)));
  const short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8, r);
}

__device__ __forceinline__ void t_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#define NS_T256_BARRIER()                                 \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)

__global__ __launch_bounds__(T_NTH) void ns_gemm_tn256_kernel(const ns_gemm_desc p, int splits, int tiles_j, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // split-major position of this workgroup: XCD x (= linear id % 8) owns the contiguous run [x * per, (x + 1) * per)
  const int per = (total + 7) >> 3;
  const int pos = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || pos >= total) return;
  const int tiles = total / splits;
  const int split = pos / tiles, tile = pos - split * tiles;
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  const int i0 = ti * T_TILE, j0 = tj * T_TILE;
  const int wi = wave >> 2, wj = wave & 3;             // 2 x 4 waves of 128 x 64

  const int chunk = (((p.K + splits - 1) / splits) + T_BK - 1) / T_BK * T_BK;
  const int k_begin = split * chunk;
  const int k_end = min(p.K, k_begin + chunk);
  const int nsteps = (max(k_end - k_begin, 0) + T_BK - 1) / T_BK;

  // DMA pieces of this lane: stage rows 4 wave + 2 i + lh (i = 0, 1), LDS chunk lr <- global chunk lr ^ t_swz(row)
  int seg[2], within[2];
  const half_t* srcA[2];
  const half_t* srcB[2];
  bool colB[2];                     // the last column tile may be ragged (N % 8 == 0): chunks past N come from the block of zeros
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = 4 * wave + 2 * i + lh, k = k_begin + rl;
    seg[i] = 0; within[i] = k;
    if (p.am.seg_rows > 0) { seg[i] = k / p.am.seg_rows; within[i] = k - seg[i] * p.am.seg_rows; }
    const int cl = (lr ^ t_swz(rl)) & 31;
    srcA[i] = (const half_t*)p.A + i0 + cl * 8;
    srcB[i] = (const half_t*)p.B + j0 + cl * 8;
    colB[i] = j0 + cl * 8 < p.N;
  }
  auto issue = [&](int step) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (step & (T_NST - 1)) * T_STAGE_BYTES + wave * 2048);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rl = 4 * wave + 2 * i + lh;
      const bool ok = k_begin + step * T_BK + rl < k_end;
      long long oa, ob;
      if (p.am.seg_rows > 0) {
        oa = (long long)seg[i] * p.am.seg_stride + (long long)within[i] * p.am.ld;
        ob = (long long)seg[i] * p.bm.seg_stride + (long long)within[i] * p.bm.ld;
      } else {
        oa = (long long)within[i] * p.am.ld;
        ob = (long long)within[i] * p.bm.ld;
      }
      t_glds16(ok ? (const void*)(srcA[i] + oa) : (const void*)ns_t256_zero_chunk, dst + i * 1024);
      t_glds16(ok && colB[i] ? (const void*)(srcB[i] + ob) : (const void*)ns_t256_zero_chunk, dst + T_OP_BYTES + i * 1024);
      within[i] += T_BK;
      if (p.am.seg_rows > 0 && within[i] >= p.am.seg_rows) { within[i] -= p.am.seg_rows; seg[i] += 1; }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // column sums of A (NS_GEMM_COLSUM_A), in the workgroups of the first column tile: the four waves that share a row
  // range hold the same A fragments, so wave wj sums the 32 columns of fragment a = wj (a quarter of the work each --
  // with one wave doing all of it those workgroups ran ~20 % longer than the rest, and the launch ends with the slowest)
  const bool colsum = (p.flags & NS_GEMM_COLSUM_A) && tj == 0;
  float csum = 0.f;

#pragma unroll
  for (int s = 0; s < T_NST - 1; ++s)
    if (s < nsteps) issue(s);

  for (int s = 0; s < nsteps; ++s) {
    // everything up to stage s has landed when at most the pieces of stages s + 1, s + 2 (4 per stage and lane) are in flight
    if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NS_T256_BARRIER();
    if (s + T_NST - 1 < nsteps) issue(s + T_NST - 1);   // into the slot stage s - 1 was multiplied from
    const char* const as = smem + (s & (T_NST - 1)) * T_STAGE_BYTES;
    const char* const bs = as + T_OP_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      half8 af[4], bf[2];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = t_frag(as, 16 * h, wi * 128 + a * 32, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = t_frag(bs, 16 * h, wj * 64 + b * 32, lane);
      if (colsum) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (a == wj) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum += (float)af[a][e];
          }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }

  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  if (colsum) {
    const float t = csum + __shfl_xor(csum, 32, 64);          // the two k halves of the column
    if (lh == 0) atomicAdd(p.H32 + i0 + wi * 128 + wj * 32 + lr, t * alpha);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (j0 + wj * 64 + b * 32 + lr >= p.N) continue;
      float* const dst = p.C32 + (long long)(i0 + wi * 128 + a * 32 + 4 * lh) * p.ldc32 + j0 + wj * 64 + b * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomicAdd(dst + (long long)((r & 3) + 8 * (r >> 2)) * p.ldc32, acc[a][b][r] * alpha);
    }
}

}  // namespace

// Whole 256-row tiles of the output (M % 256 == 0), column tiles may end ragged (N % 8 == 0), a long reduction, no dropout mask, fp32 atomics into C32 (the caller zeroes it, as for every split TN call)
bool ns_gemm_tn256_ok(const ns_gemm_desc* d) {
  return (d->flags & NS_GEMM_TN) && (d->flags & NS_GEMM_ATOMIC32) && d->drop_p == 0.f && d->M % 256 == 0 && d->N % 8 == 0 && d->N >= 256 &&
         d->K >= 16384 && d->am.ld % 8 == 0 && d->bm.ld % 8 == 0 && d->am.seg_stride % 8 == 0 && d->bm.seg_stride % 8 == 0 &&
         (d->am.seg_rows == 0 || d->am.seg_rows >= 64) && (d->M / 256) * ((d->N + 255) / 256) <= 128;
}

int ns_gemm_tn256_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles_j = (d->N + 255) / 256, tiles = (d->M / 256) * tiles_j;
  // one workgroup per CU (128 KiB of LDS): as many reduction splits as fill the 256 CUs once
  int splits = 256 / tiles;
  if (splits < 1) splits = 1;
  while (splits > 1 && d->K / splits < 512) --splits;
  const int total = tiles * splits;
  static ns_dev_once once;           // kernel attribute, once per device (ns_common.h)
  if (!ns_dyn_lds_once(once, {(const void*)ns_gemm_tn256_kernel}, T_LDS_BYTES, "ns_gemm (tn256)")) return NS_ERR_HIP;
  const int grid = ((total + 7) / 8) * 8;
  hipLaunchKernelGGL(ns_gemm_tn256_kernel, dim3(grid), dim3(T_NTH), T_LDS_BYTES, st, *d, splits, tiles_j, total);
  return 0;
}
