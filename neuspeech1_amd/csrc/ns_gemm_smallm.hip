// ns_gemm_smallm: NT GEMM for the decode loop's shapes (M = 128 .. 640 rows, N = 512 .. 2048, K = 512 .. 2048).
//
// These products are a few hundred MFLOP: what they cost is the length of the dependent chain inside a workgroup
// and the number of CUs that have work at all.  The 128x32 register-staged tile gives M = 128, N = 512 only 16
// workgroups, each walking all of K through LDS with a barrier per 64-deep step (14 us on average in the decode
// profile).  Here
//   * the tile is 32 x 32, so the same product spreads over 64 .. 384 workgroups;
//   * the four waves of a workgroup SPLIT K between them: every wave owns a quarter of the reduction and the whole
//     32 x 32 tile, loads its operands from global memory straight into MFMA fragment layout
//     (v_mfma_f32_16x16x32_f16: lane (l & 15, l >> 4) holds row l & 15, k = 8 (l >> 4) .. +7, i.e. one 16-B load)
//     and never meets a barrier inside its loop (register double buffering only);
//   * the four partial tiles meet once in LDS, are summed, and leave through the shared vector epilogue
//     (ns_nt_epilogue: bias, GELU, fp32 residual, fp16 / fp32 outputs).
// The accumulators are kept transposed (B fragment on the MFMA A port), so a lane owns 4 consecutive columns of one
// row and writes its partial with one ds_write_b128.
#include "ns_gemm_epi.h"

namespace {

#ifndef NS_SM_DEPTH
#define NS_SM_DEPTH 3      // 1 or 3 (ring of DEPTH + 1 stages, power of two)
#endif
#ifndef NS_SM_MAXM
#define NS_SM_MAXM 1024
#endif
#ifndef NS_SM_MAXTILES
#define NS_SM_MAXTILES 384
#endif
constexpr int SM_NT = 256, SM_DEPTH = NS_SM_DEPTH;   // loads of SM_DEPTH K-steps in flight ahead of the one being multiplied
typedef float sm_f32x4 __attribute__((ext_vector_type(4)));

struct sm_frags {
  half8 a[2][2];   // [row subtile][k32 half]
  half8 b[2][2];   // [col subtile][k32 half]
};

__global__ __launch_bounds__(SM_NT) void ns_gemm_smallm_kernel(const ns_gemm_desc p) {
  constexpr int BM = 32, BN = 32;
  __shared__ __attribute__((aligned(16))) float part[4 * BM * BN];
  __shared__ __attribute__((aligned(16))) float ct[BM * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const half_t* arow[2];
  const half_t* brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = (const half_t*)p.A + (long long)min(m0 + 16 * i + lr, p.M - 1) * p.am.ld + 8 * lg;
    brow[i] = (const half_t*)p.B + (long long)min(n0 + 16 * i + lr, p.N - 1) * p.bm.ld + 8 * lg;
  }
  const int total = (p.K + 63) >> 6, q = (total + 3) >> 2;
  const int s0 = wave * q, s1 = min(total, s0 + q);

  auto load = [&](int s, sm_frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = s * 64 + 32 * h;
      const bool ok = k + 8 * lg < p.K;       // K % 16 == 0: a 16-B fragment is wholly inside or wholly outside
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        half8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (half_t)0.f;
        f.a[i][h] = ok ? *(const half8*)(arow[i] + k) : z;
        f.b[i][h] = ok ? *(const half8*)(brow[i] + k) : z;
      }
    }
  };

  sm_f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = sm_f32x4{0.f, 0.f, 0.f, 0.f};

  sm_frags f[SM_DEPTH + 1];
#pragma unroll
  for (int u = 0; u < SM_DEPTH; ++u)
    if (s0 + u < s1) load(s0 + u, f[u]);
  for (int s = s0; s < s1; s += SM_DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= SM_DEPTH; ++u) {
      if (s + u + SM_DEPTH < s1) load(s + u + SM_DEPTH, f[(u + SM_DEPTH) & SM_DEPTH]);
      if (s + u < s1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[u].b[j][h], f[u].a[i][h], acc[i][j], 0, 0, 0);
      }
    }
  }
  // acc[i][j][r] = C[m0 + 16 i + lr][n0 + 16 j + 4 lg + r]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(sm_f32x4*)(&part[wave * 1024 + (16 * i + lr) * 32 + 16 * j + 4 * lg]) = acc[i][j];
  __syncthreads();
  {
    const sm_f32x4 x0 = *(const sm_f32x4*)(&part[tid * 4]), x1 = *(const sm_f32x4*)(&part[1024 + tid * 4]);
    const sm_f32x4 x2 = *(const sm_f32x4*)(&part[2048 + tid * 4]), x3 = *(const sm_f32x4*)(&part[3072 + tid * 4]);
    *(sm_f32x4*)(&ct[tid * 4]) = (x0 + x1) + (x2 + x3);
  }
  __syncthreads();
  ns_nt_epilogue<BM, BN, SM_NT>(p, ct, m0, n0, tid);
}

}  // namespace

// Beyond ~384 tiles the 32 x 32 form is bound by operand traffic (every tile re-reads 32 rows of each operand at the
// ~42 GB/s a CU can fetch): measured slower than the 128 x 32 LDS-staged tile there (tools/probe/smallm_ab.py).
bool ns_gemm_smallm_ok(const ns_gemm_desc* d) {
  // (N = 32 with M > 1024 -- the du products of the decoder's adapters under --ft_full -- also comes here: 88 tiles of 32 x 32
  // instead of 22 tiles of 128 x 32)
  if (!(d->flags & NS_GEMM_TN) && d->K2 == 0 && d->drop_p == 0.f && d->am.seg_rows == 0 && d->N == 32 && d->M > NS_SM_MAXM &&
      d->M <= 8192 && d->K >= 256 && d->K <= 4096)
    return true;
  if ((d->flags & NS_GEMM_TN) || d->K2 != 0 || d->drop_p != 0.f || d->am.seg_rows != 0 || d->M > NS_SM_MAXM || d->N < 64 ||
      d->N > 4096 || d->K < 256)
    return false;
  return ((d->M + 31) / 32) * ((d->N + 31) / 32) <= NS_SM_MAXTILES;
}

int ns_gemm_smallm_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles = ((d->M + 31) / 32) * ((d->N + 31) / 32);
  hipLaunchKernelGGL(ns_gemm_smallm_kernel, dim3(tiles), dim3(SM_NT), 0, st, *d);
  return 0;
}
