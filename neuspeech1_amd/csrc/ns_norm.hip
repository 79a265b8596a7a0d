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
