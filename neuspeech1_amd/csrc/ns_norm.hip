// LayerNorm forward / backward over the fp32 residual stream.
// One wave per row; the row lives in registers (d <= 64*NS_LN_MAX_PER_LANE).
// HBM-bound: fwd reads 4d B, writes 2d (+4d) B per row; bwd reads 2d|4d + 4d (+4d), writes 4d + 2d.
#include "ns_common.h"
#include <mutex>

namespace {
constexpr int LN_MAX_PER_LANE = 32;  // d <= 2048

template <int VEC>  // VEC = elements per lane = d / 64
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, half_t* __restrict__ y16,
                                                      float* __restrict__ y32, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * d;
  float v[VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const float4 t = *(const float4*)(xr + (i * 64 + lane) * 4);
    v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    s += (t.x + t.y) + (t.z + t.w);
  }
  const float mean = ns_wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { const float c = v[i] - mean; q += c * c; }
  const float rstd = rsqrtf(ns_wave_sum(q) / d + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 g = *(const float4*)(gamma + c);
    const float4 b = *(const float4*)(beta + c);
    float4 o;
    o.x = (v[4 * i] - mean) * rstd * g.x + b.x;
    o.y = (v[4 * i + 1] - mean) * rstd * g.y + b.y;
    o.z = (v[4 * i + 2] - mean) * rstd * g.z + b.z;
    o.w = (v[4 * i + 3] - mean) * rstd * g.w + b.w;
    if (y32) *(float4*)(y32 + (size_t)row * d + c) = o;
    if (y16) {
      half4 h = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
      *(half4*)(y16 + (size_t)row * d + c) = h;
    }
  }
}

template <int VEC, bool DY32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                      const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                      const float* __restrict__ gamma, const float* __restrict__ dres,
                                                      float* __restrict__ dx32, half_t* __restrict__ dx16, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mean = mean_in[row], rstd = rstd_in[row];
  float g[VEC], xh[VEC];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 xv = *(const float4*)(x + (size_t)row * d + c);
    const float4 gm = *(const float4*)(gamma + c);
    float dyv[4];
    if (DY32) {
      const float4 t = *(const float4*)((const float*)dy_ + (size_t)row * d + c);
      dyv[0] = t.x; dyv[1] = t.y; dyv[2] = t.z; dyv[3] = t.w;
    } else {
      const half4 t = *(const half4*)((const half_t*)dy_ + (size_t)row * d + c);
      dyv[0] = (float)t[0]; dyv[1] = (float)t[1]; dyv[2] = (float)t[2]; dyv[3] = (float)t[3];
    }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const float gs[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[4 * i + e] = dyv[e] * gs[e];
      xh[4 * i + e] = (xs[e] - mean) * rstd;
      s1 += g[4 * i + e];
      s2 += g[4 * i + e] * xh[4 * i + e];
    }
  }
  const float m1 = ns_wave_sum(s1) / d, m2 = ns_wave_sum(s2) / d;
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rstd * (g[4 * i + e] - m1 - xh[4 * i + e] * m2);
    if (dres) {
      const float4 r = *(const float4*)(dres + (size_t)row * d + c);
      o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    if (dx32) *(float4*)(dx32 + (size_t)row * d + c) = make_float4(o[0], o[1], o[2], o[3]);
    if (dx16) {
      half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
      *(half4*)(dx16 + (size_t)row * d + c) = h;
    }
  }
}

// ---- LayerNorm forward + the LoRA down-projection of the Linear it feeds, in ONE pass over the residual stream:
//        y16 = LN(x)  (the rows, statistics and roundings of ln_fwd_kernel, bit for bit)
//        u16 = round16(alpha * drop(y16) A^T)      (peft lora.Linear.forward: lora_A(dropout(x)); finetune.py:205-212)
// The unfused step ran ns_gemm_skinny right behind every such LayerNorm: a second pass over the 98 MB it had just written
// (25-33 us per site at M = 96 000, 12 sites per step).  Here a 512-thread workgroup takes 32 rows per iteration: each wave
// normalises four rows exactly as ln_fwd_kernel does (one wave per row, row in registers, same reduction tree), stores y16, and
// drops a MASKED fp16 copy into a 32-row LDS tile; after one barrier the waves multiply the tile with A^T, which sits in LDS
// for the whole launch in ns_gemm_skinny's [k / 8][n] layout, with ns_gemm_skinny's products in ns_gemm_skinny's order -- u16 is
// bitwise what the unfused pair produces (tests/test_kernels_gpu.py).  The next 32 rows are requested before the barrier.
// Tile: 16-B chunk c of row r sits at chunk c ^ (r & 15): conflict-free for the ds_write_b64 of a row and for the
// ds_read_b128 operand reads (lane = row, the k slice of ns_gemm_skinny's permuted reduction order).
template <int VEC, int NTILES, bool DROP>
__global__ __launch_bounds__(512, 1) void ln_fwd_lora_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, half_t* __restrict__ y16,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
                                                             float eps, const half_t* __restrict__ A16, int lda, half_t* __restrict__ u16,
                                                             int ldu, float alpha, float drop_p, uint32_t drop_seed,
                                                             const uint32_t* __restrict__ seed_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ll_lds[];
  constexpr int d = 64 * VEC, N = 16 * NTILES, SS = d / 64;
  constexpr int AT_BYTES = d * N * 2, ROWB = d * 2;
  unsigned char* const at = ll_lds;
  unsigned char* const tile = ll_lds + AT_BYTES;           // 32 rows x d fp16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const uint32_t dseed = DROP ? ns_eff_seed(drop_seed, seed_dev) : 0u;
  const uint32_t thr = DROP ? ns_drop_thr8(drop_p) : 0u;
  const int nblk = (rows + 31) >> 5;

  float4 xin[4][VEC / 4];
  auto request = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(blk * 32 + 4 * wave + j, rows - 1);
#pragma unroll
      for (int i = 0; i < VEC / 4; ++i) xin[j][i] = *(const float4*)(x + (size_t)row * d + (i * 64 + lane) * 4);
    }
  };
  int blk = blockIdx.x;
  if (blk < nblk) request(blk);
  // A^T -> LDS once: piece (kc, n) = A[n][8 kc .. 8 kc + 7] at byte 16 * (kc * N + n)
  for (int q = tid; q < (d >> 3) * N; q += 512) {
    const int kc = q / N, n = q - kc * N;
    *(half8*)(at + 16 * q) = *(const half8*)(A16 + (size_t)n * lda + 8 * kc);
  }
  float4 gm[VEC / 4], bt[VEC / 4];
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    gm[i] = *(const float4*)(gamma + (i * 64 + lane) * 4);
    bt[i] = *(const float4*)(beta + (i * 64 + lane) * 4);
  }
  for (; blk < nblk; blk += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rl = 4 * wave + j, row = blk * 32 + rl;
      float v[VEC];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < VEC / 4; ++i) {
        const float4 t = xin[j][i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        s += (t.x + t.y) + (t.z + t.w);
      }
      const float mean = ns_wave_sum(s) / d;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float c = v[i] - mean; q += c * c; }
      const float rstd = rsqrtf(ns_wave_sum(q) / d + eps);
      const bool live = row < rows;
      if (lane == 0 && live) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
      }
#pragma unroll
      for (int i = 0; i < VEC / 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        float4 o;
        o.x = (v[4 * i] - mean) * rstd * gm[i].x + bt[i].x;
        o.y = (v[4 * i + 1] - mean) * rstd * gm[i].y + bt[i].y;
        o.z = (v[4 * i + 2] - mean) * rstd * gm[i].z + bt[i].z;
        o.w = (v[4 * i + 3] - mean) * rstd * gm[i].w + bt[i].w;
        const half4 h = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
        if (live) *(half4*)(y16 + (size_t)row * d + c) = h;
        uint2 w = __builtin_bit_cast(uint2, h);
        if (DROP) {
          uint32_t m01, m23;
          ns_keep_masks(ns_drop_word(dseed, (uint32_t)min(row, rows - 1), (uint32_t)c >> 2), thr, m01, m23);
          w.x &= m01; w.y &= m23;
        }
        const int chunk = c >> 3;
        *(uint2*)(tile + rl * ROWB + ((chunk ^ (rl & 15)) << 4) + (c & 4) * 2) = w;
      }
    }
    const int nb = blk + gridDim.x;
    if (nb < nblk) request(nb);                     // the next rows travel while this tile is multiplied
    __syncthreads();
    // (m tile, n tile) pairs over the eight waves: u[32 blk + 16 m + lr][16 j + 4 lg + e]
#pragma unroll
    for (int pp = 0; pp < (2 * NTILES + 7) / 8; ++pp) {
      const int pr = wave + 8 * pp;
      if (pr < 2 * NTILES) {
        const int m = pr & 1, j = pr >> 1;
        const int rl = 16 * m + lr;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SS; ++s) {
          const int c0 = 8 * s + 2 * lg;
          const half8 x0 = *(const half8*)(tile + rl * ROWB + (((c0) ^ (rl & 15)) << 4));
          const half8 x1 = *(const half8*)(tile + rl * ROWB + (((c0 + 1) ^ (rl & 15)) << 4));
          const unsigned char* const bs = at + 16 * ((8 * s + 2 * lg) * N + lr) + 256 * j;
          const half8 b0 = *(const half8*)bs;
          const half8 b1 = *(const half8*)(bs + 16 * N);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, x0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, x1, acc, 0, 0, 0);
        }
        const int row = blk * 32 + rl;
        if (row < rows) {
          half4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (half_t)(acc[e] * alpha);
          *(half4*)(u16 + (size_t)row * ldu + 16 * j + 4 * lg) = h;
        }
      }
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int ns_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y16, float* y32,
                                float* mean, float* rstd, int rows, int d, float eps, void* stream) {
  NS_CHECK_ARG(x && gamma && beta && (y16 || y32), "ns_layernorm_fwd: null pointer");
  NS_CHECK_ARG(rows > 0 && d > 0 && d % 256 == 0 && d / 64 <= LN_MAX_PER_LANE, "ns_layernorm_fwd: d=%d must be a multiple of 256, <= 2048", d);
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LNF(VEC_)                                                                                                     \
  case VEC_:                                                                                                          \
    hipLaunchKernelGGL(ln_fwd_kernel<VEC_>, grid, block, 0, st, x, gamma, beta, (half_t*)y16, y32, mean, rstd, rows, d, eps); \
    break;
  switch (d / 64) {
    LNF(4) LNF(8) LNF(12) LNF(16) LNF(20)
    default:
      ns_set_error("ns_layernorm_fwd: unsupported d=%d (supported 256,512,768,1024,1280)", d);
      return NS_ERR_UNSUPPORTED;
  }
#undef LNF
  NS_CHECK_LAUNCH("ns_layernorm_fwd");
  return NS_OK;
}

extern "C" int ns_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* dx32, void* dx16, int rows, int d,
                                void* stream) {
  NS_CHECK_ARG(dy && x && mean && rstd && gamma && (dx32 || dx16), "ns_layernorm_bwd: null pointer");
  NS_CHECK_ARG(rows > 0 && d > 0 && d % 256 == 0 && d / 64 <= LN_MAX_PER_LANE, "ns_layernorm_bwd: bad d=%d", d);
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LNB(VEC_)                                                                                                       \
  case VEC_:                                                                                                            \
    if (dy_is_f32) hipLaunchKernelGGL((ln_bwd_kernel<VEC_, true>), grid, block, 0, st, dy, x, mean, rstd, gamma, dres, dx32, (half_t*)dx16, rows, d); \
    else hipLaunchKernelGGL((ln_bwd_kernel<VEC_, false>), grid, block, 0, st, dy, x, mean, rstd, gamma, dres, dx32, (half_t*)dx16, rows, d);          \
    break;
  switch (d / 64) {
    LNB(4) LNB(8) LNB(12) LNB(16) LNB(20)
    default:
      ns_set_error("ns_layernorm_bwd: unsupported d=%d", d);
      return NS_ERR_UNSUPPORTED;
  }
#undef LNB
  NS_CHECK_LAUNCH("ns_layernorm_bwd");
  return NS_OK;
}

// fused LayerNorm + LoRA down-projection (see ln_fwd_lora_kernel): d = 256 or 512, n_out = 32 or 96 rows of A16
extern "C" int ns_layernorm_fwd_lora_supported(int rows, int d, int n_out) {
  return rows >= 32 && (d == 256 || d == 512) && (n_out == 32 || n_out == 96) && (size_t)d * n_out * 2 + 32 * (size_t)d * 2 <= 160 * 1024;
}

extern "C" int ns_layernorm_fwd_lora(const float* x, const float* gamma, const float* beta, void* y16, float* mean, float* rstd,
                                     int rows, int d, float eps, const void* A16, int lda, int n_out, void* u16, int ldu, float alpha,
                                     float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream) {
  NS_CHECK_ARG(x && gamma && beta && y16 && A16 && u16, "ns_layernorm_fwd_lora: null pointer");
  NS_CHECK_ARG(ns_layernorm_fwd_lora_supported(rows, d, n_out), "ns_layernorm_fwd_lora: unsupported rows=%d d=%d n_out=%d", rows, d, n_out);
  NS_CHECK_ARG(lda >= d && lda % 8 == 0 && ldu >= n_out && ldu % 4 == 0, "ns_layernorm_fwd_lora: bad strides lda=%d ldu=%d", lda, ldu);
  NS_CHECK_ARG(drop_p >= 0.f && drop_p <= 0.5f, "ns_layernorm_fwd_lora: drop_p out of range (0 .. 0.5)");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (rows + 31) / 32;
  const int grid = nblk < 256 ? nblk : 256;
  const size_t lds = (size_t)d * n_out * 2 + 32 * (size_t)d * 2;
  const float a = alpha == 0.f ? 1.f : alpha;
#define LNL(VEC_, NT_, DR_)                                                                                                              \
  do {                                                                                                                                   \
    static ns_dev_once once;                                                                                                             \
    if (!ns_dyn_lds_once(once, {(const void*)ln_fwd_lora_kernel<VEC_, NT_, DR_>}, 160 * 1024, "ns_layernorm_fwd_lora")) return NS_ERR_HIP; \
    hipLaunchKernelGGL((ln_fwd_lora_kernel<VEC_, NT_, DR_>), dim3(grid), dim3(512), lds, st, x, gamma, beta, (half_t*)y16, mean, rstd, rows, \
                       eps, (const half_t*)A16, lda, (half_t*)u16, ldu, a, drop_p, drop_seed, seed_dev);                                  \
  } while (0)
  const bool drop = drop_p > 0.f;
  if (d == 512) {
    if (n_out == 32) { if (drop) LNL(8, 2, true); else LNL(8, 2, false); }
    else { if (drop) LNL(8, 6, true); else LNL(8, 6, false); }
  } else {
    if (n_out == 32) { if (drop) LNL(4, 2, true); else LNL(4, 2, false); }
    else { if (drop) LNL(4, 6, true); else LNL(4, 6, false); }
  }
#undef LNL
  NS_CHECK_LAUNCH("ns_layernorm_fwd_lora");
  return NS_OK;
}
