// ns_gemm: fp16 MFMA GEMM (C = A * B^T [+ A2 * B2^T]) with fused epilogues.
//
// Tile 128 x BN x 64, 4 waves.  BN = 128: waves 2x2, wave tile 64x64 = 2x2
// v_mfma_f32_32x32x16_f16.  BN = 32 (skinny N: LoRA down / du): waves 4x1.
// LDS image per operand tile: [rows][64 k] fp16 = 128-B rows, 16-B chunk index
// XOR-swizzled with (row>>1)&7 so the ds_read_b128 fragment reads (lane = row,
// same chunk) are bank-conflict free (64-dword bank row, 16-lane groups).
// Staging is register-staged and software-pipelined: global loads of tile t+1
// are issued before the MFMA phase of tile t and written to the other LDS
// buffer after it (one barrier per K-step).
//
// Two staging front-ends share the MFMA loop:
//   NT: A (M x K) and B (N x K) are k-contiguous -> 16-B global loads.
//   TN: both operands are reduction-major (weight gradients dW = dY^T X); each
//       thread loads 8 reduction rows x 2 columns as dwords and packs them into
//       two k-contiguous 16-B LDS chunks (register transpose).
// Epilogues:
//   NT: the fp32 accumulator tile is staged through LDS (the staging buffers
//       are dead by then) so every global access of the epilogue is a 16-B /
//       8-B vector along the row: bias, gelu, gelu', fp32 residual (+pos).
//   TN: fp32 atomics straight from the accumulator layout (lanes 0-31 = 128
//       contiguous bytes of one row: the full-rate atomic shape).
#include "ns_gemm_epi.h"
#include <stdlib.h>
#include <mutex>

namespace {

constexpr int BM = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per A tile

__device__ __forceinline__ long long rm_off64(const ns_rowmap& m, int row) {
  if (m.seg_rows > 0) {
    const int s = row / m.seg_rows;
    const int w = row - s * m.seg_rows;
    return (long long)s * m.seg_stride + (long long)w * m.ld;
  }
  return (long long)row * m.ld;
}
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }


template <bool TN, int BN, bool DROP>
__global__ __launch_bounds__(NTHREADS, 2) void ns_gemm_kernel(const ns_gemm_desc p) {
  const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);   // wave-uniform: one scalar load at entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BTILE_BYTES = BN * BK * 2;
  constexpr int WT_M = BN == 128 ? 2 : 1, WT_N = BN == 128 ? 2 : 1, WAVE_M = 32 * WT_M, WAVE_N = 32 * WT_N;
  constexpr int B_ITERS = BN / 32;
  char* const As = smem;                   // 2 buffers
  char* const Bs = smem + 2 * TILE_BYTES;  // 2 buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = BN == 128 ? (wave >> 1) : wave, wn = BN == 128 ? (wave & 1) : 0;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- XCD-aware tile mapping (bijective remap, n fastest inside an XCD chunk)
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

  f32x16 acc[WT_M][WT_N];
#pragma unroll
  for (int i = 0; i < WT_M; ++i)
#pragma unroll
    for (int j = 0; j < WT_N; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t drop_thr = DROP ? ns_drop_thr8(p.drop_p) : 0u;

  // ------------------------------------------------------------------ staging state
  const int st_chunk = tid & 7, st_row = tid >> 3;
  const half_t* a_ptr[4];
  const half_t* b_ptr[B_ITERS];
  const half_t* a2_ptr[4];
  const half_t* b2_ptr[B_ITERS];
  int tn_seg = 0, tn_within = 0, k_begin = 0, k_end = p.K;
  const int tq = tid & 63, tg = tid >> 6;

  if constexpr (!TN) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = st_row + 32 * it;
      a_ptr[it] = (const half_t*)p.A + rm_off64(p.am, min(m0 + row, p.M - 1));
      a2_ptr[it] = nullptr;
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int row = st_row + 32 * it;
      b_ptr[it] = (const half_t*)p.B + (long long)min(n0 + row, p.N - 1) * p.bm.ld;
      b2_ptr[it] = nullptr;
    }
    if (p.K2 > 0) {
      const int goff = p.a2_ngroup > 0 ? (n0 / p.a2_ngroup) * p.K2 : 0;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = st_row + 32 * it;
        a2_ptr[it] = (const half_t*)p.A2 + rm_off64(p.am2, min(m0 + row, p.M - 1)) + goff;
      }
#pragma unroll
      for (int it = 0; it < B_ITERS; ++it) {
        const int row = st_row + 32 * it;
        b2_ptr[it] = (const half_t*)p.B2 + (long long)min(n0 + row, p.N - 1) * p.ldb2;
      }
    }
  } else {
    const int chunk = (((p.K + p.splits - 1) / p.splits) + BK - 1) / BK * BK;
    k_begin = blockIdx.z * chunk;
    k_end = min(p.K, k_begin + chunk);
    if (p.am.seg_rows > 0) {
      tn_seg = k_begin / p.am.seg_rows;
      tn_within = k_begin - tn_seg * p.am.seg_rows;
    } else {
      tn_within = k_begin;
    }
  }

  const int steps1 = TN ? (max(k_end - k_begin, 0) + BK - 1) / BK : (p.K + BK - 1) / BK;
  const int steps2 = TN ? 0 : (p.K2 + BK - 1) / BK;
  const int nsteps = steps1 + steps2;
  // NT + DROP + second product + !DROP_A = dgrad with LoRA dropout: (A2,B2) first, mask, then the main product
  const bool drop_a = DROP && (p.flags & NS_GEMM_DROP_A);
  const bool seg2_first = (!TN) && DROP && steps2 > 0 && !drop_a;

#define NS_ST8 na0, na1, na2, na3, nb0, nb1, nb2, nb3

  auto step_info = [&](int s, bool& is2, int& k0, int& klen) __attribute__((always_inline)) {
    if (seg2_first) { is2 = s < steps2; k0 = (is2 ? s : s - steps2) * BK; }
    else { is2 = s >= steps1; k0 = (is2 ? s - steps1 : s) * BK; }
    klen = (is2 ? p.K2 : (TN ? k_end - k_begin : p.K)) - k0;
    klen = klen > BK ? BK : klen;
  };

  auto load_nt = [&](int s, uint4& a0, uint4& a1, uint4& a2, uint4& a3, uint4& b0, uint4& b1, uint4& b2, uint4& b3) __attribute__((always_inline)) {
    struct { uint4 a[4]; uint4 b[4]; } st;
    bool is2; int k0, klen; step_info(s, is2, k0, klen);
    const int ko = k0 + ((st_chunk * 8 < klen) ? st_chunk : 0) * 8;  // clamp: valid memory, never consumed
#pragma unroll
    for (int it = 0; it < 4; ++it) st.a[it] = *(const uint4*)((is2 ? a2_ptr[it] : a_ptr[it]) + ko);
#pragma unroll
    for (int it = 0; it < 4; ++it)
      st.b[it] = it < B_ITERS ? *(const uint4*)((is2 ? b2_ptr[it < B_ITERS ? it : 0] : b_ptr[it < B_ITERS ? it : 0]) + ko) : make_uint4(0, 0, 0, 0);
    if (DROP && drop_a && !is2) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t grow = (uint32_t)(m0 + st_row + 32 * it);
        uint32_t w[4] = {st.a[it].x, st.a[it].y, st.a[it].z, st.a[it].w};
        uint32_t m[4];
        ns_keep_masks(ns_drop_word(dseed, grow, (uint32_t)ko >> 2), drop_thr, m[0], m[1]);
        ns_keep_masks(ns_drop_word(dseed, grow, ((uint32_t)ko >> 2) + 1), drop_thr, m[2], m[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] &= m[e];
        st.a[it] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    a0 = st.a[0]; a1 = st.a[1]; a2 = st.a[2]; a3 = st.a[3];
    b0 = st.b[0]; b1 = st.b[1]; b2 = st.b[2]; b3 = st.b[3];
  };
  auto store_nt = [&](int buf, const uint4& a0, const uint4& a1, const uint4& a2, const uint4& a3, const uint4& b0, const uint4& b1, const uint4& b2, const uint4& b3) __attribute__((always_inline)) {
    const uint4 sa[4] = {a0, a1, a2, a3}, sb[4] = {b0, b1, b2, b3};
#pragma unroll
    for (int it = 0; it < 4; ++it)
      *(uint4*)(As + buf * TILE_BYTES + lds_off(st_row + 32 * it, st_chunk)) = sa[it];
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it)
      *(uint4*)(Bs + buf * BTILE_BYTES + lds_off(st_row + 32 * it, st_chunk)) = sb[it];
  };

  // TN: task it (0,1): reduction rows 8*(tg+4*it) + e of this K-step, columns c0 + 2*tq, +1
  struct Quad { uint4 v[4]; };
  auto load_tn_operand = [&](const half_t* X, const ns_rowmap& map, int c0, int cmax, bool mask_b, int kb,
                             int klen) -> Quad {
    Quad out;
    const int col = min(c0 + 2 * tq, cmax - 2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      uint32_t w[8];
      const int rbase = 8 * (tg + 4 * it);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rl = rbase + e;
        uint32_t v = 0;
        if (rl < klen) {
          long long off;
          if (map.seg_rows > 0) {
            int wi = tn_within + rl, sg = tn_seg;
            if (wi >= map.seg_rows) { wi -= map.seg_rows; sg += 1; }
            off = (long long)sg * map.seg_stride + (long long)wi * map.ld;
          } else {
            off = (long long)(tn_within + rl) * map.ld;
          }
          v = *(const uint32_t*)(X + off + col);
          if (DROP && mask_b) {
            const uint32_t grow = (uint32_t)(kb + rl);
            half2v hv = __builtin_bit_cast(half2v, v);
            const uint32_t dwd = ns_drop_word(dseed, grow, (uint32_t)col >> 2);
            hv[0] = ns_keep(dwd, (uint32_t)col, drop_thr) ? hv[0] : (half_t)0.f;
            hv[1] = ns_keep(dwd, (uint32_t)col + 1, drop_thr) ? hv[1] : (half_t)0.f;
            v = __builtin_bit_cast(uint32_t, hv);
          }
        }
        w[e] = v;
      }
      uint4 lo, hi;  // low halves -> column 2q, high halves -> column 2q+1 (k-contiguous)
      lo.x = (w[0] & 0xFFFFu) | (w[1] << 16);
      lo.y = (w[2] & 0xFFFFu) | (w[3] << 16);
      lo.z = (w[4] & 0xFFFFu) | (w[5] << 16);
      lo.w = (w[6] & 0xFFFFu) | (w[7] << 16);
      hi.x = (w[0] >> 16) | (w[1] & 0xFFFF0000u);
      hi.y = (w[2] >> 16) | (w[3] & 0xFFFF0000u);
      hi.z = (w[4] >> 16) | (w[5] & 0xFFFF0000u);
      hi.w = (w[6] >> 16) | (w[7] & 0xFFFF0000u);
      out.v[2 * it] = lo;
      out.v[2 * it + 1] = hi;
    }
    return out;
  };
  auto load_tn = [&](int s, uint4& a0, uint4& a1, uint4& a2, uint4& a3, uint4& b0, uint4& b1, uint4& b2, uint4& b3) __attribute__((always_inline)) {
    const int k0 = s * BK;
    const int klen = min(BK, k_end - k_begin - k0);
    const Quad qa = load_tn_operand((const half_t*)p.A, p.am, m0, p.M, false, k_begin + k0, klen);
    const Quad qb = load_tn_operand((const half_t*)p.B, p.bm, n0, p.N, true, k_begin + k0, klen);
    a0 = qa.v[0]; a1 = qa.v[1]; a2 = qa.v[2]; a3 = qa.v[3];
    b0 = qb.v[0]; b1 = qb.v[1]; b2 = qb.v[2]; b3 = qb.v[3];
    tn_within += BK;
    if (p.am.seg_rows > 0 && tn_within >= p.am.seg_rows) { tn_within -= p.am.seg_rows; tn_seg += 1; }
  };
  auto store_tn = [&](int buf, const uint4& a0, const uint4& a1, const uint4& a2, const uint4& a3, const uint4& b0, const uint4& b1, const uint4& b2, const uint4& b3) __attribute__((always_inline)) {
    const uint4 sa[4] = {a0, a1, a2, a3}, sb[4] = {b0, b1, b2, b3};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int chunk = tg + 4 * it;
      *(uint4*)(As + buf * TILE_BYTES + lds_off(2 * tq, chunk)) = sa[2 * it];
      *(uint4*)(As + buf * TILE_BYTES + lds_off(2 * tq + 1, chunk)) = sa[2 * it + 1];
      *(uint4*)(Bs + buf * BTILE_BYTES + lds_off(2 * tq, chunk)) = sb[2 * it];
      *(uint4*)(Bs + buf * BTILE_BYTES + lds_off(2 * tq + 1, chunk)) = sb[2 * it + 1];
    }
  };

  auto mma_sub = [&](const char* as, const char* bs, int s) __attribute__((always_inline)) {
    half8 af[WT_M], bf[WT_N];
#pragma unroll
    for (int i = 0; i < WT_M; ++i) af[i] = *(const half8*)(as + lds_off(wm * WAVE_M + i * 32 + lr, 2 * s + lh));
#pragma unroll
    for (int j = 0; j < WT_N; ++j) bf[j] = *(const half8*)(bs + lds_off(wn * WAVE_N + j * 32 + lr, 2 * s + lh));
#pragma unroll
    for (int i = 0; i < WT_M; ++i)
#pragma unroll
      for (int j = 0; j < WT_N; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  auto compute = [&](int buf, int klen) __attribute__((always_inline)) {
    const char* as = As + buf * TILE_BYTES;
    const char* bs = Bs + buf * BTILE_BYTES;
    if (klen == BK) {
#pragma unroll
      for (int s = 0; s < 4; ++s) mma_sub(as, bs, s);
    } else {
      for (int s = 0; s * 16 < klen; ++s) mma_sub(as, bs, s);
    }
  };

  // ---------------------------------------------------------------- main loop
  if (nsteps > 0) {
    uint4 na0, na1, na2, na3, nb0, nb1, nb2, nb3;
    if constexpr (TN) { load_tn(0, NS_ST8); store_tn(0, NS_ST8); } else { load_nt(0, NS_ST8); store_nt(0, NS_ST8); }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      const bool more = (s + 1 < nsteps);
      if constexpr (TN) { if (more) load_tn(s + 1, NS_ST8); }
      else load_nt(more ? s + 1 : s, NS_ST8);   // unconditional (clamped): staging registers stay SSA values
      bool is2; int k0, klen; step_info(s, is2, k0, klen);
      compute(cur, klen);
      if (!TN && DROP && seg2_first && s == steps2 - 1) {
        // acc holds du*A only: apply keep(seed,row,col)/(1-p) before the main product accumulates on top
#pragma unroll
        for (int i = 0; i < WT_M; ++i)
#pragma unroll
          for (int j = 0; j < WT_N; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t row = (uint32_t)(m0 + wm * WAVE_M + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
              const uint32_t col = (uint32_t)(n0 + wn * WAVE_N + j * 32 + lr);
              acc[i][j][r] = ns_keep_el(dseed, row, col, drop_thr) ? acc[i][j][r] : 0.f;
            }
      }
      if (more) { if constexpr (TN) store_tn(cur ^ 1, NS_ST8); else store_nt(cur ^ 1, NS_ST8); }
      __syncthreads();
      cur ^= 1;
    }
  }

  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;

  if constexpr (TN) {
    // ------------------------------------------------------------- TN epilogue: fp32 atomics / stores
    const bool atomic32 = p.flags & NS_GEMM_ATOMIC32;
#pragma unroll
    for (int i = 0; i < WT_M; ++i)
#pragma unroll
      for (int j = 0; j < WT_N; ++j) {
        const int col = n0 + wn * WAVE_N + j * 32 + lr;
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WAVE_M + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < p.M) {
              float* dst = p.C32 + (long long)row * p.ldc32 + col;
              const float v = acc[i][j][r] * alpha;
              if (atomic32) atomicAdd(dst, v); else *dst = v;
            }
          }
        }
      }
    return;
  }

  // --------------------------------------------------------------- NT epilogue through LDS
  // (the last loop iteration ended with a barrier: every wave is done with the staging buffers)
  float* const ct = (float*)smem;  // [BM][BN] fp32
#pragma unroll
  for (int i = 0; i < WT_M; ++i)
#pragma unroll
    for (int j = 0; j < WT_N; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WAVE_M + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        ct[row * BN + wn * WAVE_N + j * 32 + lr] = acc[i][j][r];
      }
  __syncthreads();

  ns_nt_epilogue<BM, BN, NTHREADS>(p, ct, m0, n0, tid);    // shared: loads batched ahead of the stores, settled once
}

template <bool TN, int BN, bool DROP>
int launch(const ns_gemm_desc* d, dim3 grid, size_t lds, hipStream_t st) {
  static ns_dev_once attr_once;      // kernel attribute, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)ns_gemm_kernel<TN, BN, DROP>}, 4 * TILE_BYTES, "ns_gemm")) return NS_ERR_HIP;
  hipLaunchKernelGGL((ns_gemm_kernel<TN, BN, DROP>), grid, dim3(NTHREADS), lds, st, *d);
  return 0;
}

}  // namespace

int ns_gemm_ring_launch(const ns_gemm_desc* d, hipStream_t st);
int ns_gemm_tn_launch(const ns_gemm_desc* d, hipStream_t st);
int ns_gemm_ring256_launch(const ns_gemm_desc* d, hipStream_t st);
int ns_gemm_p8_launch(const ns_gemm_desc* d, hipStream_t st);
int ns_gemm_p8s_launch(const ns_gemm_desc* d, hipStream_t st);
bool ns_gemm_p8s_ok(const ns_gemm_desc* d);
bool ns_gemm_p8_fits(const ns_gemm_desc* d);
int ns_gemm_smallm_launch(const ns_gemm_desc* d, hipStream_t st);
bool ns_gemm_smallm_ok(const ns_gemm_desc* d);
int ns_gemm_tn256_launch(const ns_gemm_desc* d, hipStream_t st);
bool ns_gemm_tn256_ok(const ns_gemm_desc* d);
int ns_gemm_skinny_launch(const ns_gemm_desc* d, hipStream_t st);
bool ns_gemm_skinny_ok(const ns_gemm_desc* d);
// the persistent form from this many 256 x 256 tiles on (A/B builds: -DNS_P8S_MIN_TILES=...).  1024 until round 4; with that round's epilogues the
// 750-tile launches (N = 512 at M = 96 000: the fp32-residual out-projection / fc2 and the K <= 2048 dgrads) gain too: 31.52 - 31.56 against
// 31.66 - 31.75 ms per step, same box
#ifndef NS_P8S_MIN_TILES
#define NS_P8S_MIN_TILES 700
#endif
static int g_use_ring = 1;
// values >= 100 do not change the mode: they set a free-standing A/B flag (flag = value - 100) that probe builds of single kernels read
// (tools/probe/*: two variants of one kernel timed in ONE process, cdna guide rule 24); the shipped kernels ignore it unless a comment says so
int g_ns_ab_flag = 0;
extern "C" void ns_debug_set_ring(int on) { if (on >= 100) g_ns_ab_flag = on - 100; else g_use_ring = on; }

// the shapes the phase-interleaved 256 x 256 kernel takes in automatic mode, with whole 256-column tiles
extern "C" int ns_gemm_side_supported(int M, int N, int K) {
  const long long tiles256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  return N >= 256 && N % 256 == 0 && M >= 2048 && tiles256 >= 192 && K % 16 == 0;
}

namespace {
__global__ __launch_bounds__(256) void side_reduce_kernel(const float* __restrict__ slabs, int tiles, int M, float alpha,
                                                          half_t* __restrict__ u16, int ldu) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;      // float4 index inside a slab
  if (q >= (long long)M * 8) return;
  const long long per = (long long)M * 8;
  const float4* src = (const float4*)slabs + q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = 0; t < tiles; ++t) {
    const float4 x = src[(long long)t * per];
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
  }
  const long long m = q >> 3;
  const int j = (int)(q & 7) * 4;
  const half4 o = {(half_t)(a.x * alpha), (half_t)(a.y * alpha), (half_t)(a.z * alpha), (half_t)(a.w * alpha)};
  *(half4*)(u16 + m * ldu + j) = o;
}
}  // namespace

extern "C" int ns_gemm_side_reduce(const float* slabs, int tiles, int M, float alpha, void* u16, int ldu, void* stream) {
  NS_CHECK_ARG(slabs && u16 && tiles > 0 && M > 0 && ldu >= 32 && ldu % 4 == 0, "ns_gemm_side_reduce: bad arguments");
  const long long n = (long long)M * 8;
  hipLaunchKernelGGL(side_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs, tiles, M,
                     alpha == 0.f ? 1.f : alpha, (half_t*)u16, ldu);
  NS_CHECK_LAUNCH("ns_gemm_side_reduce");
  return NS_OK;
}

extern "C" int ns_gemm(const ns_gemm_desc* d, void* stream) {
  NS_CHECK_ARG(d != nullptr, "ns_gemm: null descriptor");
  NS_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "ns_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  NS_CHECK_ARG(d->A && d->B, "ns_gemm: null operand");
  const bool tn = d->flags & NS_GEMM_TN;
  const ns_rowmap* maps[] = {&d->am, &d->bm, &d->am2, &d->c16m, &d->g16m, &d->p16m, &d->h32m};
  for (const ns_rowmap* m : maps)
    NS_CHECK_ARG(m->seg_rows == 0 || (m->seg_rows % 4 == 0 && m->seg_rows >= 64),
                 "ns_gemm: seg_rows=%d must be 0 or a multiple of 4 >= 64", m->seg_rows);
  if (!tn) {
    NS_CHECK_ARG(d->K % 16 == 0 && d->am.ld % 8 == 0 && d->bm.ld % 8 == 0,
                 "ns_gemm(NT): K=%d must be a multiple of 16 and lda=%d/ldb=%d multiples of 8", d->K, d->am.ld, d->bm.ld);
    NS_CHECK_ARG(d->bm.seg_rows == 0, "ns_gemm(NT): B must be a plain row-major matrix");
    NS_CHECK_ARG(d->N % 4 == 0, "ns_gemm(NT): N=%d must be a multiple of 4", d->N);
    NS_CHECK_ARG((!d->C16 || d->c16m.ld % 4 == 0) && (!d->G16 || d->g16m.ld % 4 == 0) && (!d->P16 || d->p16m.ld % 4 == 0) &&
                     (!d->H32 || d->h32m.ld % 4 == 0) && (!d->C32 || d->ldc32 % 4 == 0),
                 "ns_gemm(NT): output row strides must be multiples of 4");
    if (d->K2 > 0) {
      NS_CHECK_ARG(d->A2 && d->B2, "ns_gemm: K2>0 needs A2/B2");
      NS_CHECK_ARG(d->K2 % 16 == 0 && d->am2.ld % 8 == 0 && d->ldb2 % 8 == 0,
                   "ns_gemm(NT): K2=%d must be a multiple of 16, lda2=%d/ldb2=%d multiples of 8", d->K2, d->am2.ld, d->ldb2);
      NS_CHECK_ARG(d->a2_ngroup == 0 || d->a2_ngroup % 128 == 0, "ns_gemm: a2_ngroup must be a multiple of 128");
      // column group g of the output reads A2 columns [g*K2, (g+1)*K2): the row stride of A2 must hold every group
      NS_CHECK_ARG(d->a2_ngroup == 0 || (long long)((d->N + d->a2_ngroup - 1) / d->a2_ngroup) * d->K2 <= d->am2.ld,
                   "ns_gemm: A2 has %d columns per row but N=%d / a2_ngroup=%d needs %d groups of K2=%d", d->am2.ld, d->N,
                   d->a2_ngroup, (d->N + d->a2_ngroup - 1) / d->a2_ngroup, d->K2);
    }
    NS_CHECK_ARG(!(d->flags & NS_GEMM_ATOMIC32), "ns_gemm(NT): ATOMIC32 is a TN-only epilogue");
    if (d->splits > 1) {
      // split-K NT: fp32 atomics into a caller-zeroed C32, nothing else (the LM-head dgrad: K = padded vocabulary)
      NS_CHECK_ARG(d->C32 && !d->C16 && !d->G16 && !d->H32 && d->K2 == 0 && d->flags == 0 && d->drop_p == 0.f,
                   "ns_gemm(NT): splits > 1 needs a plain C32 output (no fp16 / residual outputs, flags, second product)");
      NS_CHECK_ARG(d->splits <= 64 && d->K / d->splits >= 256, "ns_gemm(NT): splits=%d for K=%d", d->splits, d->K);
    }
  } else {
    NS_CHECK_ARG(d->C32 && (d->flags & NS_GEMM_ATOMIC32 || d->splits <= 1), "ns_gemm(TN): needs C32 (+ATOMIC32 when split)");
    NS_CHECK_ARG(d->am.seg_rows == d->bm.seg_rows, "ns_gemm(TN): A and B must share seg_rows");
    NS_CHECK_ARG(d->M % 2 == 0 && d->N % 2 == 0 && d->am.ld % 2 == 0 && d->bm.ld % 2 == 0,
                 "ns_gemm(TN): M, N and row strides must be even");
    NS_CHECK_ARG(d->K2 == 0, "ns_gemm(TN): second product unsupported");
    NS_CHECK_ARG(d->splits >= 1 && d->splits <= 65535, "ns_gemm(TN): bad splits=%d", d->splits);
    if (d->flags & NS_GEMM_COLSUM_A)
      NS_CHECK_ARG(d->H32 && d->M > 96 && d->N > 96 && d->M % 8 == 0 && d->N % 8 == 0 && d->am.ld % 8 == 0 &&
                       d->bm.ld % 8 == 0 && d->am.seg_stride % 8 == 0 && d->bm.seg_stride % 8 == 0,
                   "ns_gemm(TN): COLSUM_A needs H32 and the 128 x 128 transposed-read kernel (M, N > 96, strides multiples of 8)");
  }
  NS_CHECK_ARG(tn || !(d->flags & NS_GEMM_COLSUM_A), "ns_gemm: COLSUM_A is a TN-only side output");
  if (d->side_B) {
    NS_CHECK_ARG(!tn && d->side_out && d->side_n == 32 && d->side_ldb % 8 == 0 && d->side_ldb >= d->N && (d->flags & NS_GEMM_GELU) &&
                     d->G16 && !d->H32 && !(d->flags & (NS_GEMM_DGELU | NS_GEMM_MUL_P16)) && d->splits <= 1 &&
                     ns_gemm_side_supported(d->M, d->N, d->K) && (g_use_ring == 1 || g_use_ring == 4 || g_use_ring == 9),
                 "ns_gemm: side product needs the large-M GELU form (ns_gemm_side_supported, G16, side_n = 32, side_out)");
    NS_CHECK_ARG(d->side_drop_p >= 0.f && d->side_drop_p <= 0.5f, "ns_gemm: side_drop_p out of range (0 .. 0.5)");
  }
  if (d->flags & (NS_GEMM_DGELU | NS_GEMM_MUL_P16)) NS_CHECK_ARG(d->P16, "ns_gemm: DGELU / MUL_P16 need P16");
  NS_CHECK_ARG(!(d->flags & NS_GEMM_GELU_SAVE_GRAD) || (d->flags & NS_GEMM_GELU), "ns_gemm: GELU_SAVE_GRAD needs GELU");
  NS_CHECK_ARG(d->drop_p >= 0.f && d->drop_p <= 0.5f, "ns_gemm: drop_p out of range (0 .. 0.5)");

  // skinny-N tiles (128x32) also serve small-M decode GEMMs: 4x more workgroups than 128x128 tiles when M <= 1024
  // (large-M N = 96 -- the stacked q|k|v LoRA bottleneck -- goes to the 128-wide tile: one pass over A, and one dropout
  // hash per element, instead of three 32-wide column tiles each re-reading and re-masking it)
  const bool skinny = !tn && ((d->N <= 96 && !(d->M >= 4096 && d->N > 32)) || (d->M <= 1024 && d->N <= 4096));
  const bool drop = d->drop_p > 0.f;
  const int bn = skinny ? 32 : 128;
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + bn - 1) / bn);
  const size_t lds = 2 * TILE_BYTES + 2 * (size_t)bn * BK * 2;   // BN=128: 64 KiB (= the fp32 epilogue tile)
  hipStream_t st = (hipStream_t)stream;
  NS_CHECK_ARG(!d->side_B || !(skinny || ns_gemm_skinny_ok(d) || ns_gemm_smallm_ok(d) || (d->flags & NS_GEMM_DROP_A)),
               "ns_gemm: side product requested but this shape does not dispatch to the phase-interleaved kernel");
  int rc = 0;      // a launcher fails only when the runtime refuses a kernel attribute (error text already set)
  if (tn && g_use_ring != 0 && g_use_ring != 8 && ns_gemm_tn256_ok(d)) {
    rc = ns_gemm_tn256_launch(d, st);   // conv-stem weight gradients: 256 x 256 LDS-DMA tiles (mode 8 = off, for A/B runs)
  } else if (tn && (g_use_ring || (d->flags & NS_GEMM_COLSUM_A)) && d->M % 8 == 0 && d->N % 8 == 0 && d->am.ld % 8 == 0 && d->bm.ld % 8 == 0 &&
      d->am.seg_stride % 8 == 0 && d->bm.seg_stride % 8 == 0 && d->M >= 8 && d->N >= 8) {
    rc = ns_gemm_tn_launch(d, st);   // row-major staging + ds_read_b64_tr_b16 fragments
  } else if (tn) {
    dim3 grid(tiles, 1, d->splits);
    rc = drop ? launch<true, 128, true>(d, grid, lds, st) : launch<true, 128, false>(d, grid, lds, st);
  } else if (!tn && d->splits > 1) {
    rc = ns_gemm_ring_launch(d, st);
  } else if (g_use_ring != 0 && g_use_ring != 7 && ns_gemm_skinny_ok(d)) {
    rc = ns_gemm_skinny_launch(d, st);   // LoRA down-projections at training size: one stream over x (mode 7 = off, for A/B runs)
  } else if (g_use_ring != 6 && g_use_ring != 0 && ns_gemm_smallm_ok(d)) {
    rc = ns_gemm_smallm_launch(d, st);   // decode shapes: 32x32 tiles, K split over the four waves (mode 6 = off, for A/B runs)
  } else if (skinny) {
    rc = drop ? launch<false, 32, true>(d, dim3(tiles), lds, st) : launch<false, 32, false>(d, dim3(tiles), lds, st);
  } else if (g_use_ring && !(d->flags & NS_GEMM_DROP_A)) {
    // 0 = register-staged kernel, 1 = auto, 2 = force the 128^2 ring, 3 = force the one-barrier 256^2 ring,
    // 4 = force the phase-interleaved 256^2 kernel where it applies, 5 = auto without the phase-interleaved kernel,
    // 9 = force its persistent form where it applies
    const long long tiles256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
    const bool big = g_use_ring == 3 || g_use_ring == 4 || g_use_ring == 9 ||
                     ((g_use_ring == 1 || g_use_ring == 5) && d->N >= 256 && tiles256 >= 192 &&
                      (d->M >= 2048 || (d->M >= 128 && d->N >= 8192)));      // (few rows with a very wide N: the LM head of a decode step, 51 968 columns -- 640 rows: 59 us against 89 on the 128^2 ring, 128 rows: 19.1 against 23.8, 256 rows: 21.5 against 44.8; tools/probe/head_gemm.py)
    const bool p8_ok = (!d->C32 || (d->flags & (1 << 27))) && d->N % 8 == 0 && (!d->C16 || d->c16m.ld % 8 == 0) &&
                       (!d->G16 || d->g16m.ld % 8 == 0) && (!d->P16 || d->p16m.ld % 8 == 0) &&
                       (!d->H32 || d->h32m.ld % 8 == 0) && ns_gemm_p8_fits(d);
    // the persistent form (one workgroup per CU walks its tiles, epilogue / next-prologue overlap) where its 32-bit epilogue addressing
    // applies and a CU gets about three or more tiles (NS_P8S_MIN_TILES; below that the one-tile form with its dynamic tile order is as fast or faster); mode 4 forces the one-tile-per-workgroup form (A/B runs)
    const bool pers = big && p8_ok && ((g_use_ring == 1 && tiles256 >= NS_P8S_MIN_TILES && !(d->flags & (1 << 27))) || g_use_ring == 9) && ns_gemm_p8s_ok(d);
    if (pers) rc = ns_gemm_p8s_launch(d, st);
    else if (big && p8_ok && (g_use_ring == 4 || g_use_ring == 1 || g_use_ring == 9)) rc = ns_gemm_p8_launch(d, st);
    else {
      // only ns_gemm_p8_kernel forms the side product: any other kernel would leave side_out unwritten and the caller's
      // ns_gemm_side_reduce would sum garbage -- refuse instead (e.g. a shape past ns_gemm_p8_fits' 2 GiB limit)
      NS_CHECK_ARG(!d->side_B, "ns_gemm: side product requested but this shape does not dispatch to the phase-interleaved kernel");
      if (big) rc = ns_gemm_ring256_launch(d, st);
      else rc = ns_gemm_ring_launch(d, st);
    }
  } else {
    NS_CHECK_ARG(!d->side_B, "ns_gemm: side product requested but this shape does not dispatch to the phase-interleaved kernel");
    rc = drop ? launch<false, 128, true>(d, dim3(tiles), lds, st) : launch<false, 128, false>(d, dim3(tiles), lds, st);
  }
  if (rc != 0) return rc;
  NS_CHECK_LAUNCH("ns_gemm");
  return NS_OK;
}
