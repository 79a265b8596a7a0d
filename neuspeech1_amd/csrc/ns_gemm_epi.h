// Shared NT epilogue: the fp32 accumulator tile sits in LDS as ct[rows][LDC]; every global access is a
// 16-B (fp32) / 8-B (fp16) vector along the row.  See ns_gemm_desc in include/neuspeech_hip.h.
//
// Memory-op ordering matters more than instruction count here (gfx950 has ONE in-order vmcnt queue for loads and
// stores): a load issued after a store is only "done" once that store has been acknowledged, so an epilogue that
// alternates {load residual row, store row} pays a full store round trip per row.  The epilogue is therefore split:
//   ns_epi_prefetch  issues every global load of this thread's rows (fp32 residual / position rows / fp16 pre-
//                    activations) back to back -- callers run it BEFORE staging the accumulators through LDS;
//   ns_epi_finish    reads the staged tile, applies bias / GELU / dGELU / residual and issues only stores.
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

#define NS_SETTLE_F(x) asm volatile("" : "+v"(x))

enum { NS_EPI_PLAIN = 0, NS_EPI_RES = 1, NS_EPI_DGELU = 2 };

__device__ __forceinline__ int ns_epi_kind(const ns_gemm_desc& p) {
  return p.H32 ? NS_EPI_RES : ((p.flags & (NS_GEMM_DGELU | NS_GEMM_MUL_P16)) ? NS_EPI_DGELU : NS_EPI_PLAIN);
}

template <int BM_, int BN_, int NTHREADS_>
struct ns_epi_geom {
  static constexpr int TPR = BN_ == 256 ? 32 : (BN_ == 128 ? 16 : 8);  // threads per tile row
  static constexpr int GPT = BN_ >= 128 ? 2 : 1;   // 4-column groups per thread (second one BN/2 columns on)
  static constexpr int GOFF = BN_ / 2;
  static constexpr int RPP = NTHREADS_ / TPR;      // rows per pass
  static constexpr int ROWS = (BM_ + RPP - 1) / RPP;
};

template <int BM_, int BN_, int NTHREADS_, int KIND>
struct ns_epi_regs {
  typedef ns_epi_geom<BM_, BN_, NTHREADS_> G;
  float4 res[KIND == NS_EPI_RES ? G::ROWS : 1][G::GPT];
  half4 pre[KIND == NS_EPI_DGELU ? G::ROWS : 1][G::GPT];
};

template <int BM_, int BN_, int NTHREADS_, int KIND, class RowFn>
__device__ __forceinline__ void ns_epi_prefetch(const ns_gemm_desc& p, const RowFn rowfn, int n0, int tid,
                                                ns_epi_regs<BM_, BN_, NTHREADS_, KIND>& rg) {
  typedef ns_epi_geom<BM_, BN_, NTHREADS_> G;
  const int cg = tid % G::TPR, r0 = tid / G::TPR;
  if (KIND == NS_EPI_RES) {
#pragma unroll
    for (int i = 0; i < G::ROWS; ++i) {
      const int row = min(rowfn(r0 + i * G::RPP), p.M - 1);
      const long long oh = ns_rm_off64(p.h32m, row);
#pragma unroll
      for (int g = 0; g < G::GPT; ++g) {
        const int col = min(n0 + cg * 4 + g * G::GOFF, p.N - 4);
        rg.res[i][g] = p.R32 ? *(const float4*)(p.R32 + oh + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (p.pos) {
#pragma unroll
      for (int i = 0; i < G::ROWS; ++i) {
        const int row = min(rowfn(r0 + i * G::RPP), p.M - 1);
        const long long opos = (long long)(row % p.pos_rows) * p.N;
#pragma unroll
        for (int g = 0; g < G::GPT; ++g) {
          const int col = min(n0 + cg * 4 + g * G::GOFF, p.N - 4);
          const float4 ps = *(const float4*)(p.pos + opos + col);
          rg.res[i][g].x += ps.x; rg.res[i][g].y += ps.y; rg.res[i][g].z += ps.z; rg.res[i][g].w += ps.w;
        }
      }
    }
  }
  if (KIND == NS_EPI_DGELU) {
    const half_t* const P16 = (const half_t*)p.P16;
#pragma unroll
    for (int i = 0; i < G::ROWS; ++i) {
      const int row = min(rowfn(r0 + i * G::RPP), p.M - 1);
      const long long op = ns_rm_off64(p.p16m, row);
#pragma unroll
      for (int g = 0; g < G::GPT; ++g) {
        const int col = min(n0 + cg * 4 + g * G::GOFF, p.N - 4);
        rg.pre[i][g] = *(const half4*)(P16 + op + col);
      }
    }
  }
}

// ct row rl (row stride LDC_ floats) is output row rowfn(rl)
template <int BM_, int BN_, int NTHREADS_, int LDC_, int KIND, class RowFn>
__device__ __forceinline__ void ns_epi_finish(const ns_gemm_desc& p, const float* ct, const RowFn rowfn, int n0, int tid,
                                              const ns_epi_regs<BM_, BN_, NTHREADS_, KIND>& rg) {
  typedef ns_epi_geom<BM_, BN_, NTHREADS_> G;
  const int cg = tid % G::TPR, r0 = tid / G::TPR;
  half_t* const C16 = (half_t*)p.C16;
  half_t* const G16 = (half_t*)p.G16;
  const bool do_gelu = p.flags & NS_GEMM_GELU;
  const bool save_grad = p.flags & NS_GEMM_GELU_SAVE_GRAD, mulp = p.flags & NS_GEMM_MUL_P16;
  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;

  float4 bias4[G::GPT];
  bool colok[G::GPT];
#pragma unroll
  for (int g = 0; g < G::GPT; ++g) {
    const int col = n0 + cg * 4 + g * G::GOFF;
    colok[g] = col + 4 <= p.N;
    bias4[g] = (p.bias && colok[g]) ? *(const float4*)(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // Settle every loaded register HERE, once.  Left to the compiler, the first use of a loaded value inside the
  // branchy row loop becomes s_waitcnt vmcnt(0) in EVERY iteration, which also drains the stores issued so far
  // (one store round trip per row: 8.5k cycles per 128-row half instead of ~3k).
#pragma unroll
  for (int g = 0; g < G::GPT; ++g) {
    NS_SETTLE_F(bias4[g].x); NS_SETTLE_F(bias4[g].y); NS_SETTLE_F(bias4[g].z); NS_SETTLE_F(bias4[g].w);
  }
  ns_epi_regs<BM_, BN_, NTHREADS_, KIND> rs = rg;
#pragma unroll
  for (int i = 0; i < (KIND == NS_EPI_RES ? G::ROWS : 0); ++i)
#pragma unroll
    for (int g = 0; g < G::GPT; ++g) {
      NS_SETTLE_F(rs.res[i][g].x); NS_SETTLE_F(rs.res[i][g].y); NS_SETTLE_F(rs.res[i][g].z); NS_SETTLE_F(rs.res[i][g].w);
    }
#pragma unroll
  for (int i = 0; i < (KIND == NS_EPI_DGELU ? G::ROWS : 0); ++i)
#pragma unroll
    for (int g = 0; g < G::GPT; ++g) {
      typedef uint32_t ns_u2 __attribute__((ext_vector_type(2)));
      ns_u2 t = __builtin_bit_cast(ns_u2, rs.pre[i][g]);
      asm volatile("" : "+v"(t));
      rs.pre[i][g] = __builtin_bit_cast(half4, t);
    }
#pragma unroll
  for (int i = 0; i < G::ROWS; ++i) {
    const int rl = r0 + i * G::RPP;
    if (rl >= BM_) continue;
    const int row = rowfn(rl);
    if (row >= p.M) continue;
    const long long oc = C16 ? ns_rm_off64(p.c16m, row) : 0;
    const long long og = G16 ? ns_rm_off64(p.g16m, row) : 0;
    const long long oh = KIND == NS_EPI_RES ? ns_rm_off64(p.h32m, row) : 0;
#pragma unroll
    for (int g = 0; g < G::GPT; ++g) {
      if (!colok[g]) continue;
      const int cl = cg * 4 + g * G::GOFF, col = n0 + cl;
      const float4 a = *(const float4*)(ct + rl * LDC_ + cl);
      float v[4] = {a.x * alpha + bias4[g].x, a.y * alpha + bias4[g].y, a.z * alpha + bias4[g].z, a.w * alpha + bias4[g].w};
      if (p.C32) *(float4*)(p.C32 + (long long)row * p.ldc32 + col) = make_float4(v[0], v[1], v[2], v[3]);
      half4 v16 = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      if (KIND == NS_EPI_DGELU) {
        const half4 pv = rs.pre[i][g];
        if (mulp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v16[e] = (half_t)((float)v16[e] * (float)pv[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v16[e] = (half_t)((float)v16[e] * ns_gelu_grad((float)pv[e]));
        }
      }
      half4 gv = v16, cv = v16;
      if (do_gelu) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          ns_f2 g_, dg_;
          ns_gelu_both2(ns_f2{(float)v16[e], (float)v16[e + 1]}, g_, dg_);
          gv[e] = (half_t)g_.x; gv[e + 1] = (half_t)g_.y;
          if (save_grad) { cv[e] = (half_t)dg_.x; cv[e + 1] = (half_t)dg_.y; }
        }
      }
      if (C16) *(half4*)(C16 + oc + col) = cv;
      if (G16) *(half4*)(G16 + og + col) = gv;
      if (KIND == NS_EPI_RES) {
        float4 h = rs.res[i][g];
        h.x += (float)gv[0]; h.y += (float)gv[1]; h.z += (float)gv[2]; h.w += (float)gv[3];
        *(float4*)(p.H32 + oh + col) = h;
      }
    }
  }
}

template <int BM_, int BN_, int NTHREADS_, int LDC_, int KIND, class RowFn>
__device__ __forceinline__ void ns_epi_both(const ns_gemm_desc& p, const float* ct, const RowFn rowfn, int n0, int tid) {
  ns_epi_regs<BM_, BN_, NTHREADS_, KIND> rg;
  ns_epi_prefetch<BM_, BN_, NTHREADS_, KIND>(p, rowfn, n0, tid, rg);
  ns_epi_finish<BM_, BN_, NTHREADS_, LDC_, KIND>(p, ct, rowfn, n0, tid, rg);
}

// one-call form for kernels that stage first: loads are still issued back to back ahead of every store
template <int BM_, int BN_, int NTHREADS_, int LDC_, class RowFn>
__device__ __forceinline__ void ns_nt_epilogue_map(const ns_gemm_desc& p, const float* ct, const RowFn rowfn, int n0, int tid) {
  const int kind = ns_epi_kind(p);
  if (kind == NS_EPI_RES) ns_epi_both<BM_, BN_, NTHREADS_, LDC_, NS_EPI_RES>(p, ct, rowfn, n0, tid);
  else if (kind == NS_EPI_DGELU) ns_epi_both<BM_, BN_, NTHREADS_, LDC_, NS_EPI_DGELU>(p, ct, rowfn, n0, tid);
  else ns_epi_both<BM_, BN_, NTHREADS_, LDC_, NS_EPI_PLAIN>(p, ct, rowfn, n0, tid);
}

struct ns_row_affine {
  int base;
  __device__ __forceinline__ int operator()(int rl) const { return base + rl; }
};

template <int BM_, int BN_, int NTHREADS_>
__device__ __forceinline__ void ns_nt_epilogue(const ns_gemm_desc& p, const float* ct, int m0, int n0, int tid) {
  ns_nt_epilogue_map<BM_, BN_, NTHREADS_, BN_>(p, ct, ns_row_affine{m0}, n0, tid);
}
