// ns_gemm_skinny: NT GEMM for the LoRA down-projections u = alpha * drop(x) A^T at training size
// (M = B*T rows of x, K = 512 / 2048, N = r or 3r columns): one pass over x at HBM rate.
//
// x is 98 .. 393 MB, A^T is 32 .. 128 KB: the product is a stream over x with the whole of A resident.  The 128 x 32
// LDS-staged tile of ns_gemm_kernel walks K with a barrier per 64-deep step and reaches ~2 TB/s on these shapes.  Here
//   * A (the B operand of the NT form) is copied ONCE per workgroup into LDS, laid out [k / 8][n] in 16-B pieces, so
//     a fragment read (lane = column n, 8 consecutive k) is a conflict-free ds_read_b128;
//   * every wave owns whole 16-row blocks of x and never meets a barrier after that copy: its lanes load x straight
//     into MFMA operand layout from global memory, 32 contiguous bytes per lane, eight 64-deep steps (16 loads of
//     16 B) in flight per wave ahead of the step being multiplied;
//   * the reduction index is permuted inside each 64-deep step so that those 32 bytes are contiguous:
//     v_mfma_f32_16x16x32_f16 lane (l & 15, l >> 4) holds 8 k-values per product; product h of a step takes
//     k = 64 s + 16 (l >> 4) + 8 h + [0, 8) from BOTH operands (any bijection of k works when both sides use it),
//     so the four lane groups of a row read one whole 128-B line per step;
//   * the accumulators are kept transposed (A^T on the MFMA A port): a lane owns 4 consecutive columns of one row of
//     u and stores them as one 8-B piece;
//   * NS_GEMM_DROP_A: the keep mask of ns_common.h (one hash word per 4 columns) is applied to the x fragments in
//     registers, exactly as ns_gemm_kernel's staging does.
#include "ns_common.h"
#include <mutex>

namespace {

constexpr int SK_WAVES = 8, SK_NT = 64 * SK_WAVES;   // SK_DEPTH (template): steps of 64 k in flight per wave, 8 (K % 512 == 0) or 4
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

template <int NTILES, bool DROP, int SK_DEPTH>
__global__ __launch_bounds__(SK_NT, 1) void ns_gemm_skinny_kernel(const ns_gemm_desc p) {
  const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);   // wave-uniform: one scalar load at entry
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_lds[];
  constexpr int N = 16 * NTILES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int SS = p.K >> 6;                                  // 64-deep steps per row block (a multiple of SK_DEPTH)
  const int nblk = (p.M + 15) >> 4;
  const int wstride = gridDim.x * SK_WAVES;
  int rb_l = blockIdx.x * SK_WAVES + wave;                  // row block the load cursor is in
  const half_t* const A = (const half_t*)p.A;
  const long long lda = p.am.ld;
  const uint32_t thr = DROP ? ns_drop_thr8(p.drop_p) : 0u;

  auto rowptr = [&](int rb) __attribute__((always_inline)) {
    return A + (long long)min(rb * 16 + lr, p.M - 1) * lda + 16 * lg;
  };
  const half_t* lp = rowptr(min(rb_l, nblk - 1));
  int s_l = 0;
  half8 xa[SK_DEPTH][2];
  auto load = [&](int u) __attribute__((always_inline)) {
    xa[u][0] = *(const half8*)(lp + 64 * s_l);
    xa[u][1] = *(const half8*)(lp + 64 * s_l + 8);
    if (++s_l == SS) {
      s_l = 0;
      rb_l += wstride;
      lp = rowptr(min(rb_l, nblk - 1));                     // past the end: valid memory, never consumed
    }
  };
  // x first: these loads are in flight while A^T is copied to LDS
#pragma unroll
  for (int u = 0; u < SK_DEPTH; ++u) load(u);

  // A^T -> LDS: piece (kc, n) = B[n][8 kc .. 8 kc + 7] at byte 16 * (kc * N + n); consecutive threads take consecutive n
  {
    const int pieces = (p.K >> 3) * N;
    const half_t* const Bm = (const half_t*)p.B;
    // four loads in flight per thread
    for (int q0 = tid; q0 < pieces; q0 += 4 * SK_NT) {
      half8 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = min(q0 + i * SK_NT, pieces - 1), kc = q / N, n = q - kc * N;
        t[i] = *(const half8*)(Bm + (long long)n * p.bm.ld + 8 * kc);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (q0 + i * SK_NT < pieces) *(half8*)(sk_lds + 16 * (q0 + i * SK_NT)) = t[i];
    }
  }
  __syncthreads();

  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  half_t* const C = (half_t*)p.C16;
  for (int rb = blockIdx.x * SK_WAVES + wave; rb < nblk; rb += wstride) {
    sk_f32x4 acc[NTILES];
#pragma unroll
    for (int j = 0; j < NTILES; ++j) acc[j] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t grow = (uint32_t)min(rb * 16 + lr, p.M - 1);
    for (int s0 = 0; s0 < SS; s0 += SK_DEPTH) {
#pragma unroll
      for (int u = 0; u < SK_DEPTH; ++u) {
        half8 x0 = xa[u][0], x1 = xa[u][1];
        load(u);
        const int s = s0 + u;
        if (DROP) {
          const uint32_t c4 = (uint32_t)(64 * s + 16 * lg) >> 2;
          uint32_t m[8];
          ns_keep_masks(ns_drop_word(dseed, grow, c4), thr, m[0], m[1]);
          ns_keep_masks(ns_drop_word(dseed, grow, c4 + 1), thr, m[2], m[3]);
          ns_keep_masks(ns_drop_word(dseed, grow, c4 + 2), thr, m[4], m[5]);
          ns_keep_masks(ns_drop_word(dseed, grow, c4 + 3), thr, m[6], m[7]);
          uint4 w0 = *(uint4*)&x0, w1 = *(uint4*)&x1;
          w0.x &= m[0]; w0.y &= m[1]; w0.z &= m[2]; w0.w &= m[3];
          w1.x &= m[4]; w1.y &= m[5]; w1.z &= m[6]; w1.w &= m[7];
          x0 = *(half8*)&w0; x1 = *(half8*)&w1;
        }
        const unsigned char* const bs = sk_lds + 16 * ((8 * s + 2 * lg) * N + lr);
#pragma unroll
        for (int j = 0; j < NTILES; ++j) {
          const half8 b0 = *(const half8*)(bs + 256 * j);
          const half8 b1 = *(const half8*)(bs + 16 * N + 256 * j);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, x0, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, x1, acc[j], 0, 0, 0);
        }
      }
    }
    // acc[j][e] = u[16 rb + lr][16 j + 4 lg + e]
    const int row = rb * 16 + lr;
    if (row < p.M) {
      half_t* const dst = C + (long long)row * p.c16m.ld + 4 * lg;
#pragma unroll
      for (int j = 0; j < NTILES; ++j) {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)(acc[j][e] * alpha);
        *(half4*)(dst + 16 * j) = h;
      }
    }
  }
}

template <int NTILES, bool DROP, int DEPTH>
int sk_launch3(const ns_gemm_desc* d, hipStream_t st, int grid, size_t lds) {
  static ns_dev_once once;           // kernel attribute, once per device (ns_common.h)
  if (!ns_dyn_lds_once(once, {(const void*)ns_gemm_skinny_kernel<NTILES, DROP, DEPTH>}, 160 * 1024, "ns_gemm (skinny)")) return NS_ERR_HIP;
  hipLaunchKernelGGL((ns_gemm_skinny_kernel<NTILES, DROP, DEPTH>), dim3(grid), dim3(SK_NT), lds, st, *d);
  return 0;
}

template <int NTILES>
int sk_launch(const ns_gemm_desc* d, hipStream_t st) {
  const size_t lds = (size_t)d->K * 16 * NTILES * 2;
  const int nblk = (d->M + 15) / 16;
  int grid = (nblk + SK_WAVES - 1) / SK_WAVES;
  if (grid > 256) grid = 256;                               // one workgroup per CU, row blocks strided over its waves
  const bool drop = d->drop_p > 0.f && (d->flags & NS_GEMM_DROP_A);
  const bool deep = (d->K >> 6) % 8 == 0;
  if (drop) return deep ? sk_launch3<NTILES, true, 8>(d, st, grid, lds) : sk_launch3<NTILES, true, 4>(d, st, grid, lds);
  return deep ? sk_launch3<NTILES, false, 8>(d, st, grid, lds) : sk_launch3<NTILES, false, 4>(d, st, grid, lds);
}

}  // namespace

// The shapes this form is built for: plain row-major x and u, N = 32 or 96, K a multiple of 256 with A^T fitting LDS,
// enough rows that the one-off copy of A^T is amortised; fp16 output only (no bias / residual / second product).
bool ns_gemm_skinny_ok(const ns_gemm_desc* d) {
  if ((d->flags & ~NS_GEMM_DROP_A) || d->K2 != 0 || d->splits > 1 || d->am.seg_rows != 0 || d->c16m.seg_rows != 0) return false;
  if (!d->C16 || d->G16 || d->H32 || d->C32 || d->bias || d->P16) return false;
  if (d->drop_p > 0.f && !(d->flags & NS_GEMM_DROP_A)) return false;
  if (d->N != 32 && d->N != 96) return false;
  if (d->K % 256 != 0 || (size_t)d->K * d->N * 2 > 144 * 1024) return false;
  return d->M >= 8192 && d->am.ld % 8 == 0 && d->bm.ld % 8 == 0 && d->c16m.ld % 4 == 0;
}

int ns_gemm_skinny_launch(const ns_gemm_desc* d, hipStream_t st) {
  return d->N == 32 ? sk_launch<2>(d, st) : sk_launch<6>(d, st);
}
