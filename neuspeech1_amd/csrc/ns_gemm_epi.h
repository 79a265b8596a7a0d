// Shared NT epilogue: the fp32 accumulator tile sits in LDS as ct[BM][BN]; every global access is a
// 16-B (fp32) / 8-B (fp16) vector along the row.  See ns_gemm_desc in include/neuspeech_hip.h.
#pragma once
#include "ns_common.h"

__device__ __forceinline__ long long ns_rm_off64(const ns_rowmap& m, int row) {
  if (m.seg_rows > 0) {
    const int s = row / m.seg_rows;
    const int w = row - s * m.seg_rows;
    return (long long)s * m.seg_stride + (long long)w * m.ld;
  }
  return (long long)row * m.ld;
}

template <int BM_, int BN_, int NTHREADS_>
__device__ __forceinline__ void ns_nt_epilogue(const ns_gemm_desc& p, const float* ct, int m0, int n0, int tid) {
  constexpr int TPR = BN_ == 256 ? 32 : (BN_ == 128 ? 16 : 8);  // threads per tile row
  constexpr int GPT = BN_ >= 128 ? 2 : 1;   // 4-column groups per thread (second one BN/2 columns on)
  constexpr int GOFF = BN_ / 2;
  constexpr int RPP = NTHREADS_ / TPR;      // rows per pass
  const int cg = tid % TPR, r0 = tid / TPR;
  half_t* const C16 = (half_t*)p.C16;
  half_t* const G16 = (half_t*)p.G16;
  const half_t* const P16 = (const half_t*)p.P16;
  const bool do_gelu = p.flags & NS_GEMM_GELU;
  const bool do_dgelu = p.flags & NS_GEMM_DGELU;
  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;

  float4 bias4[GPT];
  bool colok[GPT];
#pragma unroll
  for (int g = 0; g < GPT; ++g) {
    const int col = n0 + cg * 4 + g * GOFF;
    colok[g] = col + 4 <= p.N;
    bias4[g] = (p.bias && colok[g]) ? *(const float4*)(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int rl = r0; rl < BM_; rl += RPP) {
    const int row = m0 + rl;
    if (row >= p.M) break;
    const long long oc = C16 ? ns_rm_off64(p.c16m, row) : 0;
    const long long og = G16 ? ns_rm_off64(p.g16m, row) : 0;
    const long long op = P16 ? ns_rm_off64(p.p16m, row) : 0;
    const long long oh = p.H32 ? ns_rm_off64(p.h32m, row) : 0;
    const long long opos = p.pos ? (long long)(row % p.pos_rows) * p.N : 0;
#pragma unroll
    for (int g = 0; g < GPT; ++g) {
      if (!colok[g]) continue;
      const int cl = cg * 4 + g * GOFF, col = n0 + cl;
      const float4 a = *(const float4*)(ct + rl * BN_ + cl);
      float v[4] = {a.x * alpha + bias4[g].x, a.y * alpha + bias4[g].y, a.z * alpha + bias4[g].z, a.w * alpha + bias4[g].w};
      if (p.C32) *(float4*)(p.C32 + (long long)row * p.ldc32 + col) = make_float4(v[0], v[1], v[2], v[3]);
      half4 v16 = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      if (do_dgelu) {
        const half4 pv = *(const half4*)(P16 + op + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v16[e] = (half_t)((float)v16[e] * ns_gelu_grad((float)pv[e]));
      }
      if (C16) *(half4*)(C16 + oc + col) = v16;
      half4 gv = v16;
      if (do_gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = (half_t)ns_gelu((float)v16[e]);
      }
      if (G16) *(half4*)(G16 + og + col) = gv;
      if (p.H32) {
        float4 h = p.R32 ? *(const float4*)(p.R32 + oh + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        h.x += (float)gv[0]; h.y += (float)gv[1]; h.z += (float)gv[2]; h.w += (float)gv[3];
        if (p.pos) {
          const float4 ps = *(const float4*)(p.pos + opos + col);
          h.x += ps.x; h.y += ps.y; h.z += ps.z; h.w += ps.w;
        }
        *(float4*)(p.H32 + oh + col) = h;
      }
    }
  }
}
