// Decode-time kernels of the generate loop (HBM-bound byte work, no MFMA):
//   ns_attn_decode     one new query per row against a K/V cache (self: beam ancestry indirection,
//                      cross: the encoder K/V of a sequence is read ONCE for all its beams)
//   ns_logits_process  log_softmax + repetition penalty + no-repeat-ngram + suppress lists (+ beam score)
//   ns_topk_groups     top-k over each sequence's (beams x V) accumulated scores
//   ns_beam_update     HF beam bookkeeping (running beams, finished hypotheses, early-stop heuristic)
//   ns_greedy_update   argmax + EOS/pad bookkeeping
// Replaces, for the path evaluation.py:369-386 drives: HF:generation/utils.py:2783-2975 (greedy), :3077-3545
// (beam search), HF:generation/logits_process.py:306-414, :1073-1141, :1816-1906 and the cache reorder
// utils/load_model.py:1353-1360 (here: an int32 ancestry table instead of re-gathering K/V tensors).
#include "ns_common.h"
#include <stdlib.h>

namespace {

constexpr int MAXQ = 8;       // queries (beams) sharing one K/V stream
constexpr int D = 64;

// ---------------------------------------------------------------------------------------------- attention
// grid (groups, H); group = nq consecutive rows.  8-lane subgroups own one key at a time (lane8 = 8 dims).
//
// ONE pass over the keys: every subgroup keeps its own running (max, sum, 8-dim accumulator per lane) and loads the K
// row and the V row of KU keys in the same iteration; the 32 subgroups are merged once at the end (shuffles inside a
// wave, LDS across the four waves).  The earlier three-phase form (scores -> softmax by one wave -> P V, two block
// barriers, fp32 score array in LDS) left the memory pipe idle between the K stream and the V stream: 100 us per
// layer on the 393 MB cross-attention K/V of B = 128 against 62 us at HBM rate.
template <int NQ>
__global__ __launch_bounds__(256) void attn_decode_kernel(const ns_attn_decode_desc p) {
  __shared__ float red[4][NQ][D + 2];                   // per wave and query: 64 accumulators, max, sum
  const int Lk = p.kv_len_dev ? *p.kv_len_dev : p.Lk;
  const int tid = threadIdx.x, sg = tid >> 3, l8 = tid & 7, wave = tid >> 6, lane = tid & 63;
  const int grp = blockIdx.x, h = blockIdx.y;
  const half_t* Kb = (const half_t*)p.K + h * D + l8 * 8;
  const half_t* Vb = (const half_t*)p.V + h * D + l8 * 8;
  const int* anc = p.anc ? p.anc + (long long)grp * p.anc_ld : nullptr;   // only with NQ == 1

  float q[NQ][8], m[NQ], l[NQ], acc[NQ][8];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const half8 qv = *(const half8*)((const half_t*)p.Q + (long long)(grp * NQ + qi) * p.ldq + h * D + l8 * 8);
    m[qi] = -INFINITY;
    l[qi] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { q[qi][e] = (float)qv[e]; acc[qi][e] = 0.f; }
  }
  auto krow = [&](int j) -> long long {
    return anc ? (long long)j * p.kv_pos_stride + anc[j] : (long long)grp * p.kv_group_stride + j;
  };
  // append mode: the newest position comes from this step's projection rows and is stored into the cache on the way
  const half_t* const Kn = p.Knew ? (const half_t*)p.Knew + (long long)grp * p.ldnew + h * D + l8 * 8 : nullptr;
  const half_t* const Vn = p.Knew ? (const half_t*)p.Vnew + (long long)grp * p.ldnew + h * D + l8 * 8 : nullptr;
  if (Kn && sg == 0) {
    const long long r = (long long)(Lk - 1) * p.kv_pos_stride + p.slot0 + grp;
    *(half8*)((half_t*)p.K + h * D + l8 * 8 + r * p.ldk) = *(const half8*)Kn;
    *(half8*)((half_t*)p.V + h * D + l8 * 8 + r * p.ldv) = *(const half8*)Vn;
  }
  // KU keys per subgroup and iteration (2 KU independent 16-B loads in flight per lane).  With K and V in the same pass
  // and 8 workgroups per CU one key is enough: KU = 1 / 2 / 4 / 8 gave 101.0 / 100.2 / 99.4 / 98.6 k tokens/s greedy
  // and 77.5 / 77.5 / 75.3 / 71.1 k beam-5 in a same-box sweep (short self-attention rows waste the wider forms)
#ifndef NS_AD_KU
#define NS_AD_KU 1
#endif
  constexpr int KU = NS_AD_KU;
  for (int j0 = sg * KU; j0 < Lk; j0 += 32 * KU) {
    half8 kv[KU], vv[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int j = min(j0 + u, Lk - 1);
      if (Kn && j == Lk - 1) {
        kv[u] = *(const half8*)Kn;
        vv[u] = *(const half8*)Vn;
      } else {
        const long long r = krow(j);
        kv[u] = *(const half8*)(Kb + r * p.ldk);
        vv[u] = *(const half8*)(Vb + r * p.ldv);
      }
    }
    float s[NQ][KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      float kf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) kf[e] = (float)kv[u][e];
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += q[qi][e] * kf[e];
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        s[qi][u] = j0 + u < Lk ? d : -INFINITY;
      }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float mn = m[qi];
#pragma unroll
      for (int u = 0; u < KU; ++u) mn = fmaxf(mn, s[qi][u]);      // key j0 is always valid: mn is finite
      const float alpha = __expf(m[qi] - mn);
      m[qi] = mn;
      float pw[KU], ps = 0.f;
#pragma unroll
      for (int u = 0; u < KU; ++u) { pw[u] = __expf(s[qi][u] - mn); ps += pw[u]; }
      l[qi] = l[qi] * alpha + ps;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = acc[qi][e] * alpha;
#pragma unroll
        for (int u = 0; u < KU; ++u) a += pw[u] * (float)vv[u][e];
        acc[qi][e] = a;
      }
    }
  }
  // merge the 8 subgroups of the wave (lanes with equal l8), then the 4 waves through LDS
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    float M = m[qi];
    M = fmaxf(M, __shfl_xor(M, 8, 64));
    M = fmaxf(M, __shfl_xor(M, 16, 64));
    M = fmaxf(M, __shfl_xor(M, 32, 64));
    const float sc = m[qi] == -INFINITY ? 0.f : __expf(m[qi] - M);   // a subgroup that saw no key contributes nothing
    float lw = l[qi] * sc;
    lw += __shfl_xor(lw, 8, 64);
    lw += __shfl_xor(lw, 16, 64);
    lw += __shfl_xor(lw, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[qi][e] * sc;
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 8) red[wave][qi][l8 * 8 + e] = v;
    }
    if (lane == 0) { red[wave][qi][D] = M; red[wave][qi][D + 1] = lw; }
  }
  __syncthreads();
  for (int i = tid; i < NQ * D; i += 256) {
    const int qi = i >> 6, dd = i & 63;
    float M = red[0][qi][D];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, red[w][qi][D]);
    float v = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = red[w][qi][D];
      const float sc = mw == -INFINITY ? 0.f : __expf(mw - M);
      v += red[w][qi][dd] * sc;
      L += red[w][qi][D + 1] * sc;
    }
    ((half_t*)p.O)[(long long)(grp * NQ + qi) * p.ldo + h * D + dd] = (half_t)(v / L);
  }
}

// ---------------------------------------------------------------------------------------------- self-attention, one wave per head
// The decode loop's SELF-attention (ancestry layout, one query per row, <= a few hundred keys): the four-wave form above
// gives every (row, head) 32 subgroups, of which a 68-key row fills 2 1/8 rounds -- and every wave runs the whole instruction
// stream whether its subgroups hold a key or not: at 640 rows x 8 heads that is 20 480 waves of ~400 instructions for 350 k
// keys (16.8 us per layer in the beam-5 profile, 7 us at 128 rows).  Here ONE WAVE owns a (row, head): its 8 subgroups walk
// the keys j = sg, sg + 8, ..., AS_KU keys per iteration (the ancestry entries of an iteration are loaded together, then
// its 2 AS_KU rows), the subgroups meet by shuffles, nothing goes through LDS and nothing waits at a barrier.  A workgroup is
// four such waves = four heads of one row.
#ifndef NS_AS_KU
#define NS_AS_KU 4
#endif
__global__ __launch_bounds__(256) void attn_self_kernel(const ns_attn_decode_desc p) {
  constexpr int KU = NS_AS_KU;
  const int Lk = p.kv_len_dev ? *p.kv_len_dev : p.Lk;
  const int lane = threadIdx.x & 63, sg = lane >> 3, l8 = lane & 7;
  const int grp = blockIdx.x, h = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (h >= p.H) return;                                   // (no barrier below)
  const half_t* Kb = (const half_t*)p.K + h * D + l8 * 8;
  const half_t* Vb = (const half_t*)p.V + h * D + l8 * 8;
  const int* anc = p.anc + (long long)grp * p.anc_ld;
  // the ancestry entries of an iteration are requested one iteration ahead (the first ones before anything else)
  int a[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) a[u] = anc[min(sg + 8 * u, Lk - 1)];
  const half_t* const Kn = p.Knew ? (const half_t*)p.Knew + (long long)grp * p.ldnew + h * D + l8 * 8 : nullptr;
  const half_t* const Vn = p.Knew ? (const half_t*)p.Vnew + (long long)grp * p.ldnew + h * D + l8 * 8 : nullptr;
  half8 knew, vnew;
#pragma unroll
  for (int e = 0; e < 8; ++e) { knew[e] = (half_t)0.f; vnew[e] = (half_t)0.f; }
  if (Kn) {                                               // (stored into the cache at the end: a store in front of the loop is waited for)
    knew = *(const half8*)Kn;
    vnew = *(const half8*)Vn;
  }
  float q[8], acc[8], m = -INFINITY, l = 0.f;
  {
    const half8 qv = *(const half8*)((const half_t*)p.Q + (long long)grp * p.ldq + h * D + l8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { q[e] = (float)qv[e]; acc[e] = 0.f; }
  }
  for (int j0 = sg; j0 < Lk; j0 += 8 * KU) {              // key j0 is valid for this subgroup: its running max is finite
    // branch-free: a lane past the end re-reads the last key (its score is masked below), so that the 2 KU rows of an
    // iteration are in flight together.  The newest position comes from this step's projection rows (in registers; its
    // cache row -- slot 0 of that position, whatever it holds -- is loaded and dropped).
    half8 kv[KU], vv[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int j = min(j0 + 8 * u, Lk - 1);
      const bool fresh = Kn && j == Lk - 1;
      const long long r = (long long)j * p.kv_pos_stride + (fresh ? 0 : a[u]);
      const half8 kl = *(const half8*)(Kb + r * p.ldk), vl = *(const half8*)(Vb + r * p.ldv);
      kv[u] = fresh ? knew : kl;
      vv[u] = fresh ? vnew : vl;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) a[u] = anc[min(j0 + 8 * (KU + u), Lk - 1)];
    float s[KU], mn = m;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += q[e] * (float)kv[u][e];
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      s[u] = j0 + 8 * u < Lk ? d : -INFINITY;
      mn = fmaxf(mn, s[u]);
    }
    const float alpha = __expf(m - mn);
    m = mn;
    float pw[KU], ps = 0.f;
#pragma unroll
    for (int u = 0; u < KU; ++u) { pw[u] = __expf(s[u] - mn); ps += pw[u]; }
    l = l * alpha + ps;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = acc[e] * alpha;
#pragma unroll
      for (int u = 0; u < KU; ++u) t += pw[u] * (float)vv[u][e];
      acc[e] = t;
    }
  }
  // the 8 subgroups (lanes with equal l8) meet
  float M = m;
  M = fmaxf(M, __shfl_xor(M, 8, 64));
  M = fmaxf(M, __shfl_xor(M, 16, 64));
  M = fmaxf(M, __shfl_xor(M, 32, 64));
  const float sc = m == -INFINITY ? 0.f : __expf(m - M);   // a subgroup that saw no key contributes nothing
  float L = l * sc;
  L += __shfl_xor(L, 8, 64);
  L += __shfl_xor(L, 16, 64);
  L += __shfl_xor(L, 32, 64);
  half8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e] * sc;
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    o[e] = (half_t)(v / L);
  }
  if (sg == 0) {
    *(half8*)((half_t*)p.O + (long long)grp * p.ldo + h * D + l8 * 8) = o;
    if (Kn) {
      const long long r = (long long)(Lk - 1) * p.kv_pos_stride + p.slot0 + grp;
      *(half8*)((half_t*)p.K + h * D + l8 * 8 + r * p.ldk) = knew;
      *(half8*)((half_t*)p.V + h * D + l8 * 8 + r * p.ldv) = vnew;
    }
  }
}

// ---------------------------------------------------------------------------------------------- few-query attention
// See ns_attn_fewq in the header.  v_mfma_f32_16x16x32_f16(X, Y, C): lane (l & 15, l >> 4) of X holds row l & 15,
// k = 8 (l >> 4) .. +7; of Y column l & 15, same k; C[row 4 (l >> 4) + i][column l & 15].
//   S^T[key][q] = K Q^T      X = K rows (row r of the tile is key k0 + 8 (r >> 2) + 4 t + (r & 3), t = tile 0 / 1),
//                            Y = Q rows  ->  lane (q, g) holds the scores of keys k0 + 8 g + 4 t + i: 8 CONSECUTIVE keys
//   O^T[d][q]   = V^T P^T    X = rows of the transposed value image (one 16-B load: keys k0 + 8 g .. + 7 of dim d),
//                            Y = those 8 probabilities, already in place  ->  the query index of a lane is l & 15 in
//                            both products, so max / sum / rescale are lane-local plus two cross-lane steps.
typedef float fq_f32x4 __attribute__((ext_vector_type(4)));
struct fq_frags { half8 k[2][2]; half8 v[4]; };   // k[tile][k32 half], v[dim tile]

#ifndef NS_FQ_WAVES
#define NS_FQ_WAVES 4
#endif
constexpr int FQ_W = NS_FQ_WAVES;   // waves per workgroup = ways the keys are split
__global__ __launch_bounds__(64 * FQ_W) void attn_fewq_kernel(const ns_attn_fewq_desc p) {
  __shared__ float red[FQ_W][16][D + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int grp = blockIdx.x, h = blockIdx.y, Lk = p.Lk;
  half8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (half_t)0.f;
  half8 qf[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
    qf[hh] = lr < p.nq ? *(const half8*)((const half_t*)p.Q + (long long)(grp * p.nq + lr) * p.ldq + h * D + 32 * hh + 8 * lg)
                       : zero8;
  const half_t* Kg = (const half_t*)p.K + (long long)grp * Lk * p.ldk + h * D + 8 * lg;
  const half_t* Vg = (const half_t*)p.Vt + ((long long)(grp * p.H + h) * D + lr) * p.ldvt + 8 * lg;
  const int krow0 = 8 * (lr >> 2) + (lr & 3);
  const int chunk = (((Lk + FQ_W - 1) / FQ_W) + 31) & ~31;
  const int k_begin = wave * chunk, k_end = min(Lk, k_begin + chunk);

  auto load = [&](int k0, fq_frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long long ro = (long long)min(k0 + krow0 + 4 * t, Lk - 1) * p.ldk;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) f.k[t][hh] = *(const half8*)(Kg + ro + 32 * hh);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) f.v[dt] = *(const half8*)(Vg + (long long)16 * dt * p.ldvt + k0);
  };

  float m = -INFINITY, l = 0.f;
  fq_f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = fq_f32x4{0.f, 0.f, 0.f, 0.f};

  auto tile = [&](int k0, const fq_frags& f) __attribute__((always_inline)) {
    fq_f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = fq_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.k[t][hh], qf[hh], s[t], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (k0 + 8 * lg + 4 * t + i >= Lk) s[t][i] = -INFINITY;
        mx = fmaxf(mx, s[t][i]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));     // key k0 < Lk exists: finite
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    m = mn;
    half8 pf;
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = __expf(s[t][i] - mn);
        ps += e;
        pf[4 * t + i] = (half_t)e;
      }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      o[dt] *= alpha;
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.v[dt], pf, o[dt], 0, 0, 0);
    }
  };

  // three fragment stages, loads two 32-key tiles ahead
  fq_frags f[3];
  if (k_begin < k_end) load(k_begin, f[0]);
  if (k_begin + 32 < k_end) load(k_begin + 32, f[1]);
  for (int k0 = k_begin; k0 < k_end; k0 += 96) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int kk = k0 + 32 * u;
      if (kk + 64 < k_end) load(kk + 64, f[(u + 2) % 3]);
      if (kk < k_end) tile(kk, f[u]);
    }
  }
  // o[dt][i] = O^T[16 dt + 4 lg + i][q = lr] of this wave's keys
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][lr][16 * dt + 4 * lg + i] = o[dt][i];
  if (lg == 0) { red[wave][lr][D] = m; red[wave][lr][D + 1] = l; }
  __syncthreads();
  for (int idx = tid; idx < p.nq * D; idx += 64 * FQ_W) {
    const int qi = idx >> 6, dd = idx & 63;
    float M = red[0][qi][D];
#pragma unroll
    for (int w = 1; w < FQ_W; ++w) M = fmaxf(M, red[w][qi][D]);
    float v = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < FQ_W; ++w) {
      const float mw = red[w][qi][D];
      const float sc = mw == -INFINITY ? 0.f : __expf(mw - M);
      v += red[w][qi][dd] * sc;
      L += red[w][qi][D + 1] * sc;
    }
    ((half_t*)p.O)[(long long)(grp * p.nq + qi) * p.ldo + h * D + dd] = (half_t)(v / L);
  }
}

// V rows (key-major) -> transposed value image; 64 keys x 64 dims per workgroup through an LDS tile
__global__ __launch_bounds__(256) void vt_pack_kernel(const half_t* __restrict__ v, int ldv, half_t* __restrict__ vt, int H,
                                                       int Lk, int ldvt) {
  __shared__ half_t tile[64][D + 2];
  const int j0 = blockIdx.x * 64, h = blockIdx.y, g = blockIdx.z;
  const int c8 = (threadIdx.x & 7) * 8, r = threadIdx.x >> 3;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int j = j0 + r + 32 * it;
    half8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (half_t)0.f;
    if (j < Lk) x = *(const half8*)(v + ((long long)g * Lk + j) * ldv + h * D + c8);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r + 32 * it][c8 + e] = x[e];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int dd = r + 32 * it;
    if (j0 + c8 < ldvt) {
      half8 x;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[c8 + e][dd];
      *(half8*)(vt + ((long long)(g * H + h) * D + dd) * ldvt + j0 + c8) = x;
    }
  }
}

// ---------------------------------------------------------------------------------------------- processors
__device__ __forceinline__ float blk_reduce(float v, bool is_max, float* sh) {
  v = is_max ? ns_wave_max(v) : ns_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

__device__ __forceinline__ int ns_forced_token(const ns_logits_proc_desc& p, int cur) {
  if (!p.forced || cur < 0 || cur >= p.n_forced) return -1;
  const int t = p.forced[cur];
  return t < p.V ? t : -1;
}

__global__ __launch_bounds__(256) void logits_process_kernel(const ns_logits_proc_desc p) {
  __shared__ float sh[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const half_t* lg = (const half_t*)p.logits16 + (long long)row * p.ldv;
  float* out = p.scores32 + (long long)row * p.V;
  const int64_t* ids = p.ids + (long long)row * p.ids_ld;
  const int cur = p.cur_len_dev ? *p.cur_len_dev : p.cur_len;
  float lse = 0.f;
  if (p.log_softmax) {
    float mx = -INFINITY;
    for (int c = tid; c < p.V; c += 256) mx = fmaxf(mx, (float)lg[c]);
    mx = blk_reduce(mx, true, sh);
    float se = 0.f;
    for (int c = tid; c < p.V; c += 256) se += __expf((float)lg[c] - mx);
    se = blk_reduce(se, false, sh);
    lse = mx + __logf(se);
  }
  const bool biased = p.bias1 != nullptr || p.n_seq > 0;
  if (p.bias1) {
    for (int c = tid; c < p.V; c += 256) out[c] = ((float)lg[c] - lse) + p.bias1[c];
  } else {
    for (int c = tid; c < p.V; c += 256) out[c] = (float)lg[c] - lse;
  }
  __syncthreads();
  if (p.n_seq > 0) {
    // sequences of length >= 2: the row's history must end with all tokens but the last (HF adds once per sequence, so
    // sequences that share their last token accumulate)
    for (int sq = tid; sq < p.n_seq; sq += 256) {
      const int o0 = p.seq_off[sq], len = p.seq_off[sq + 1] - o0, pre = len - 1;
      if (pre < 1 || len > cur) continue;      // HF ignores sequences longer than the context (len > cur)
      bool match = true;
      for (int e = 0; e < pre; ++e) match = match && (ids[cur - pre + e] == (int64_t)p.seq_tok[o0 + e]);
      const int last = p.seq_tok[o0 + pre];
      if (match && last >= 0 && last < p.V) atomicAdd(&out[last], p.seq_bias[sq]);
    }
    __syncthreads();
  }
  if (p.repetition_penalty != 1.f) {
    if (!biased) {
      // every occurrence recomputes from the ORIGINAL value, so duplicates write the same number
      for (int t = tid; t < cur; t += 256) {
        const int64_t tok = ids[t];
        if (tok >= 0 && tok < p.V) {
          const float b = (float)lg[tok] - lse;
          out[tok] = b < 0.f ? b * p.repetition_penalty : b / p.repetition_penalty;
        }
      }
    } else {
      // the penalised value is the BIASED score held in `out`: only the first occurrence of a token rewrites it
      for (int t = tid; t < cur; t += 256) {
        const int64_t tok = ids[t];
        if (tok < 0 || tok >= p.V) continue;
        bool first = true;
        for (int e = 0; e < t; ++e) first = first && (ids[e] != tok);
        if (first) {
          const float b = out[tok];
          out[tok] = b < 0.f ? b * p.repetition_penalty : b / p.repetition_penalty;
        }
      }
    }
    __syncthreads();
  }
  const int n = p.no_repeat_ngram;
  if (n > 0 && cur + 1 >= n) {
    for (int s = tid; s + n - 1 < cur; s += 256) {
      bool match = true;
      for (int e = 0; e < n - 1; ++e) match = match && (ids[s + e] == ids[cur - (n - 1) + e]);
      if (match) {
        const int64_t banned = ids[s + n - 1];
        if (banned >= 0 && banned < p.V) out[banned] = -INFINITY;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < p.n_suppress; i += 256) {
    const int c = p.suppress[i];
    if (c >= 0 && c < p.V) out[c] = -INFINITY;
  }
  if (cur == p.begin_index)
    for (int i = tid; i < p.n_begin_suppress; i += 256) {
      const int c = p.begin_suppress[i];
      if (c >= 0 && c < p.V) out[c] = -INFINITY;
    }
  const int ft = ns_forced_token(p, cur);
  if (ft >= 0) {      // ForceTokensLogitsProcessor: the forced token scores 0, everything else -inf
    __syncthreads();
    for (int c = tid; c < p.V; c += 256) out[c] = c == ft ? 0.f : -INFINITY;
  }
  if (p.beam_scores) {
    __syncthreads();
    const float bs = p.beam_scores[row];
    for (int c = tid; c < p.V; c += 256) out[c] += bs;
  }
}

// top-k of n contiguous floats per group; order = (value desc, index asc); k <= 16.  Two stages: each of `nchunks`
// blocks per group selects the top-k of its chunk out of LDS (k cheap passes), then one block per group merges the
// nchunks*k candidates.  Identical result to k lexicographic passes over the whole group.
constexpr int TOPK_CHUNK = 8192;

__device__ __forceinline__ void blk_argmax_after(float& bv, int& bi, float* shv, int* shi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = bv; shi[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = shv[0]; bi = shi[0];
  for (int w = 1; w < 4; ++w)
    if (shv[w] > bv || (shv[w] == bv && shi[w] < bi)) { bv = shv[w]; bi = shi[w]; }
}

__global__ __launch_bounds__(256) void topk_stage1_kernel(const float* __restrict__ x, long long n, int k, int nchunks,
                                                          float* __restrict__ cv, int* __restrict__ ci) {
  __shared__ float buf[TOPK_CHUNK];
  __shared__ float shv[4];
  __shared__ int shi[4];
  const int grp = blockIdx.y, ch = blockIdx.x;
  const long long lo = (long long)ch * TOPK_CHUNK;
  const int len = (int)min((long long)TOPK_CHUNK, n - lo);
  const float* xr = x + (long long)grp * n + lo;
  for (int i = threadIdx.x; i < len; i += 256) buf[i] = xr[i];
  __syncthreads();
  float pv = INFINITY;
  int pi = -1;
  for (int t = 0; t < k; ++t) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < len; i += 256) {
      const float v = buf[i];
      const bool after = (v < pv) || (v == pv && i > pi);
      if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
    }
    blk_argmax_after(bv, bi, shv, shi);
    if (threadIdx.x == 0) {
      cv[((long long)grp * nchunks + ch) * k + t] = bv;
      ci[((long long)grp * nchunks + ch) * k + t] = bi == 0x7fffffff ? 0x7fffffff : (int)(lo + bi);
    }
    pv = bv; pi = bi;
  }
}

__global__ __launch_bounds__(256) void topk_stage2_kernel(const float* __restrict__ cv, const int* __restrict__ ci, int ncand,
                                                          int k, float* __restrict__ vals, int* __restrict__ idx) {
  __shared__ float shv[4];
  __shared__ int shi[4];
  const float* v = cv + (long long)blockIdx.x * ncand;
  const int* id = ci + (long long)blockIdx.x * ncand;
  if (ncand <= 1024 && k <= 256) {
    // short lists (the beam step: num_beams x 2 num_beams candidates): every candidate counts the ones ahead of it (value
    // desc, index asc) and lands at its rank; k rounds of block-wide argmax cost ~1.2 us each
    __shared__ float sv[1024];
    __shared__ int si[1024];
    for (int i = threadIdx.x; i < ncand; i += 256) { sv[i] = v[i]; si[i] = id[i]; }
    if ((int)threadIdx.x < k) { vals[blockIdx.x * k + threadIdx.x] = -INFINITY; idx[blockIdx.x * k + threadIdx.x] = 0x7fffffff; }
    __syncthreads();
    for (int i = threadIdx.x; i < ncand; i += 256) {
      const float x = sv[i];
      const int gi = si[i];
      if (!(x > -INFINITY) && gi == 0x7fffffff) continue;       // padding entries: what the defaults already say
      int rank = 0;
      for (int jj = 0; jj < ncand; ++jj) rank += (sv[jj] > x || (sv[jj] == x && si[jj] < gi)) ? 1 : 0;
      if (rank < k) { vals[blockIdx.x * k + rank] = x; idx[blockIdx.x * k + rank] = gi; }
    }
    return;
  }
  float pv = INFINITY;
  int pi = -1;
  for (int t = 0; t < k; ++t) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < ncand; i += 256) {
      const float x = v[i];
      const int gi = id[i];
      const bool after = (x < pv) || (x == pv && gi > pi);
      if (after && (x > bv || (x == bv && gi < bi))) { bv = x; bi = gi; }
    }
    blk_argmax_after(bv, bi, shv, shi);
    if (threadIdx.x == 0) { vals[blockIdx.x * k + t] = bv; idx[blockIdx.x * k + t] = bi; }
    pv = bv; pi = bi;
  }
}

// ---------------------------------------------------------------------------------------------- fused select
// logits_process + per-row top-k without the fp32 score matrix.  Four streaming passes over the fp16 row with 16-B
// loads (max, sum-exp, per-thread max, candidate collection): the first comes from HBM, the rest from L2 / Infinity
// Cache; nothing but k (value, column) pairs per row is written.  The processors touch few columns, so they go to
// an LDS patch list (+ bitmaps of overridden / banned columns) and every other column keeps its raw logit:
// (raw - lse) + beam_score is monotone in raw, so the selection compares raw values and only the winners are
// transformed -- with exactly the arithmetic of logits_process_kernel, so values and order are bit-identical.
// Selection: T = k-th largest of the 256 per-thread maxima is a lower bound of the row's k-th largest value; the
// candidates are every un-overridden column with raw >= T (typically k..k+2 of them) plus the finite patch entries;
// k block-argmax rounds over that short list give the result (value desc, column asc).
constexpr int SEL_MAX_LDV = 26 * 256 * 8;   // bitmap capacity (53,248 columns)
constexpr int SEL_CAP = 1024;               // candidate / patch list capacity
constexpr int SEL_WORDS = SEL_MAX_LDV / 32;
constexpr int SEL_NV = SEL_MAX_LDV / 8 / 256;   // 16-B vectors of a row per thread (26)
constexpr int SEL_MATCH = 256;                  // matched multi-token sequence-bias entries per row (more: exact slow path)

__global__ __launch_bounds__(256) void logits_select_kernel(const ns_logits_proc_desc p, int k, int group_rows,
                                                            float* __restrict__ cand_vals, int* __restrict__ cand_idx) {
  __shared__ uint32_t ovr[SEL_WORDS], inf[SEL_WORDS];
  __shared__ float pval[SEL_CAP], cval[SEL_CAP];
  __shared__ int pcol[SEL_CAP], ccol[SEL_CAP];
  __shared__ float sh[4], shv[4];
  __shared__ int shi[4];
  __shared__ int npatch, ncand;
  __shared__ float tmax[256];
  __shared__ float Tsh;
  __shared__ uint32_t hist[SEL_WORDS];       // sequence-bias mode: tokens of the row's history (for the repetition penalty)
  __shared__ int mlast[SEL_MATCH];           // ... and the matched multi-token entries (last token, bias)
  __shared__ float mbias[SEL_MATCH];
  __shared__ int nmatch;
  const int row = blockIdx.x, tid = threadIdx.x;
  const half_t* lg = (const half_t*)p.logits16 + (long long)row * p.ldv;
  const int64_t* ids = p.ids + (long long)row * p.ids_ld;
  const int cur = p.cur_len_dev ? *p.cur_len_dev : p.cur_len;
  const int V = p.V, nvec = (V + 7) >> 3;
  {
    const int ft = ns_forced_token(p, cur);
    if (ft >= 0) {
      // forced position: one finite candidate (0 + beam score); the -inf rest in column order, as a top-k over the
      // processed row of ns_logits_process would list them (value desc, column asc)
      if (tid < k) {
        const int gb = (row % group_rows) * V;
        const int c = tid == 0 ? ft : (tid - 1 < ft ? tid - 1 : tid);
        cand_vals[(long long)row * k + tid] = tid == 0 ? (p.beam_scores ? p.beam_scores[row] : 0.f) : -INFINITY;
        cand_idx[(long long)row * k + tid] = c < V ? gb + c : 0x7fffffff;
      }
      return;
    }
  }
  // The row (<= 53,248 fp16 logits) is read ONCE into registers, every load in flight together: each of the four passes
  // below used to walk global memory again with four 16-B loads in flight per thread, i.e. a latency-bound 8-10 us per
  // pass and row (79 us per beam-search step for 640 rows).
  half8 rowv[SEL_NV];
#pragma unroll
  for (int u = 0; u < SEL_NV; ++u) {
    const int vi = tid + u * 256;
    rowv[u] = *(const half8*)(lg + (long long)min(vi, nvec - 1) * 8);
  }
  const bool biased = p.bias1 != nullptr || p.n_seq > 0;
  for (int i = tid; i < SEL_WORDS; i += 256) { ovr[i] = 0u; inf[i] = 0u; if (biased) hist[i] = 0u; }
  if (tid == 0) { npatch = 0; ncand = 0; nmatch = 0; }

  float lse = 0.f;
  if (p.log_softmax) {
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < SEL_NV; ++u) {
      const int vi = tid + u * 256;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (vi * 8 + e < V) mx = fmaxf(mx, (float)rowv[u][e]);
      }
    }
    mx = blk_reduce(mx, true, sh);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < SEL_NV; ++u) {
      const int vi = tid + u * 256;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (vi * 8 + e < V) se += __expf((float)rowv[u][e] - mx);
      }
    }
    se = blk_reduce(se, false, sh);
    lse = mx + __logf(se);
  }
  __syncthreads();
  const float bs = p.beam_scores ? p.beam_scores[row] : 0.f;
  auto finalv = [&](float raw) __attribute__((always_inline)) -> float {
    float o = raw - lse;
    if (p.beam_scores) o += bs;
    return o;
  };
  // ---- processors -> bitmaps + patch list (same order / semantics as logits_process_kernel)
  if (biased) {
    // HF order: sequence bias first, the repetition penalty then works on the BIASED score.  Every column one of the two
    // touches gets ONE patch: ((raw - lse) + bias1[c]) + the matched multi-token entries ending in c, then the penalty if
    // c occurs in the history.  Columns nothing touches keep (raw - lse) + 0.0 = raw - lse, i.e. the unbiased path.
    auto touch = [&](int c) __attribute__((always_inline)) {
      const uint32_t bit = 1u << (c & 31);
      if (!(atomicOr(&ovr[c >> 5], bit) & bit)) {
        const int pos = atomicAdd(&npatch, 1);
        if (pos < SEL_CAP) { pcol[pos] = c; pval[pos] = ((float)lg[c] - lse) + (p.bias1 ? p.bias1[c] : 0.f); }
      }
    };
    if (p.repetition_penalty != 1.f)
      for (int t = tid; t < cur; t += 256) {
        const int64_t tok = ids[t];
        if (tok >= 0 && tok < V) { atomicOr(&hist[tok >> 5], 1u << (tok & 31)); touch((int)tok); }
      }
    if (p.bias1)
      for (int c = tid; c < V; c += 256)
        if (p.bias1[c] != 0.f) touch(c);
    for (int sq = tid; sq < p.n_seq; sq += 256) {
      const int o0 = p.seq_off[sq], len = p.seq_off[sq + 1] - o0, pre = len - 1;
      if (pre < 1 || len > cur) continue;      // HF ignores sequences longer than the context
      bool match = true;
      for (int e = 0; e < pre; ++e) match = match && (ids[cur - pre + e] == (int64_t)p.seq_tok[o0 + e]);
      const int last = p.seq_tok[o0 + pre];
      if (match && last >= 0 && last < V) {
        touch(last);
        const int mp = atomicAdd(&nmatch, 1);
        if (mp < SEL_MATCH) { mlast[mp] = last; mbias[mp] = p.seq_bias[sq]; }
      }
    }
    __syncthreads();
    const int np0 = min(npatch, SEL_CAP), nm = min(nmatch, SEL_MATCH);
    for (int i = tid; i < nm; i += 256)
      for (int q = 0; q < np0; ++q)
        if (pcol[q] == mlast[i]) { atomicAdd(&pval[q], mbias[i]); break; }
    __syncthreads();
    if (p.repetition_penalty != 1.f)
      for (int q = tid; q < np0; q += 256) {
        const int c = pcol[q];
        if ((hist[c >> 5] >> (c & 31)) & 1u) { const float b = pval[q]; pval[q] = b < 0.f ? b * p.repetition_penalty : b / p.repetition_penalty; }
      }
    __syncthreads();
  } else if (p.repetition_penalty != 1.f) {
    for (int t = tid; t < cur; t += 256) {
      const int64_t tok = ids[t];
      if (tok >= 0 && tok < V) {
        const uint32_t bit = 1u << (tok & 31);
        if (!(atomicOr(&ovr[tok >> 5], bit) & bit)) {          // first occurrence only: duplicates carry the same value
          const float b = (float)lg[tok] - lse;
          const int pos = atomicAdd(&npatch, 1);
          if (pos < SEL_CAP) { pcol[pos] = (int)tok; pval[pos] = b < 0.f ? b * p.repetition_penalty : b / p.repetition_penalty; }
        }
      }
    }
  }
  auto ban = [&](long long c) __attribute__((always_inline)) {
    if (c >= 0 && c < V) { atomicOr(&ovr[c >> 5], 1u << (c & 31)); atomicOr(&inf[c >> 5], 1u << (c & 31)); }
  };
  const int n = p.no_repeat_ngram;
  if (n > 0 && cur + 1 >= n) {
    for (int s0 = tid; s0 + n - 1 < cur; s0 += 256) {
      bool match = true;
      for (int e = 0; e < n - 1; ++e) match = match && (ids[s0 + e] == ids[cur - (n - 1) + e]);
      if (match) ban(ids[s0 + n - 1]);
    }
  }
  for (int i = tid; i < p.n_suppress; i += 256) ban(p.suppress[i]);
  if (cur == p.begin_index)
    for (int i = tid; i < p.n_begin_suppress; i += 256) ban(p.begin_suppress[i]);
  __syncthreads();

  // ---- per-thread maximum of the un-overridden columns, T = k-th largest of the 256 maxima
  float mloc = -INFINITY;
#pragma unroll
  for (int u = 0; u < SEL_NV; ++u) {
    const int vi = tid + u * 256;
    if (vi < nvec) {
      const int c0 = vi * 8;
      const uint32_t m8 = (ovr[c0 >> 5] >> (c0 & 31)) & 0xFFu;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < V && !((m8 >> e) & 1u)) mloc = fmaxf(mloc, (float)rowv[u][e]);
    }
  }
  // (by RANK: every thread counts the maxima ahead of its own -- value desc, thread index asc -- and the one at rank k - 1
  // publishes its value; k rounds of block-wide argmax cost ~1.2 us each, 20 of them per row with the final selection)
  float T = -INFINITY;
  {
    tmax[tid] = mloc;
    if (tid == 0) Tsh = -INFINITY;
    __syncthreads();
    int rank = 0;
#pragma unroll 8
    for (int jj = 0; jj < 256; ++jj) {
      const float vj = tmax[jj];
      rank += (vj > mloc || (vj == mloc && jj < tid)) ? 1 : 0;
    }
    if (rank == k - 1) Tsh = mloc;
    __syncthreads();
    T = Tsh;
  }
  // ---- candidates: raw >= T (un-overridden) + finite patch entries, in final-value space
  if (mloc >= T && mloc > -INFINITY) {     // threads whose maximum is below T hold no candidate: skip their re-read
#pragma unroll
    for (int u = 0; u < SEL_NV; ++u) {
      const int vi = tid + u * 256;
      if (vi >= nvec) continue;
      const int c0 = vi * 8;
      const uint32_t m8 = (ovr[c0 >> 5] >> (c0 & 31)) & 0xFFu;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float raw = (float)rowv[u][e];
        if (c0 + e < V && !((m8 >> e) & 1u) && raw >= T) {
          const int pos = atomicAdd(&ncand, 1);
          if (pos < SEL_CAP) { ccol[pos] = c0 + e; cval[pos] = finalv(raw); }
        }
      }
    }
  }
  const int np = min(npatch, SEL_CAP);
  for (int i = tid; i < np; i += 256) {
    const int c = pcol[i];
    if (!((inf[c >> 5] >> (c & 31)) & 1u)) {
      const int pos = atomicAdd(&ncand, 1);
      if (pos < SEL_CAP) { ccol[pos] = c; cval[pos] = p.beam_scores ? pval[i] + bs : pval[i]; }
    }
  }
  __syncthreads();
  const bool overflow = ncand > SEL_CAP || npatch > SEL_CAP || nmatch > SEL_MATCH;
  const int nc = min(ncand, SEL_CAP);
  const int gbase = (row % group_rows) * V;
  if (!overflow) {
    // the short candidate list, by rank as well (value desc, column asc; columns are unique): candidate i lands at its rank
    for (int i = tid; i < nc; i += 256) {
      const float v = cval[i];
      const int c = ccol[i];
      int rank = 0;
      for (int jj = 0; jj < nc; ++jj) {
        const float vj = cval[jj];
        rank += (vj > v || (vj == v && ccol[jj] < c)) ? 1 : 0;
      }
      if (rank < k) {
        cand_vals[(long long)row * k + rank] = v;
        cand_idx[(long long)row * k + rank] = gbase + c;
      }
    }
    if (tid >= nc && tid < k) {          // fewer candidates than k: the rest as an exhausted argmax reports them
      cand_vals[(long long)row * k + tid] = -INFINITY;
      cand_idx[(long long)row * k + tid] = 0x7fffffff;
    }
    return;
  }
  float pv = INFINITY;
  int pi = -1;
  for (int t = 0; t < k; ++t) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    {
      // lists overflowed (pathological rows: hundreds of ties, or a prompt longer than the list): exact slow path,
      // k full passes over the row in final-value space
#pragma unroll 1
      for (int c = tid; c < V; c += 256) {
        float v;
        if ((inf[c >> 5] >> (c & 31)) & 1u) v = -INFINITY;
        else if ((ovr[c >> 5] >> (c & 31)) & 1u) {
          float b = (float)lg[c] - lse;
          bool pen = true;
          if (biased) {     // rebuild the patch value of this column from the tables (order of the adds as in logits_process_kernel)
            b += p.bias1 ? p.bias1[c] : 0.f;
            for (int sq = 0; sq < p.n_seq; ++sq) {
              const int o0 = p.seq_off[sq], len = p.seq_off[sq + 1] - o0, pre = len - 1;
              if (pre < 1 || len > cur || p.seq_tok[o0 + pre] != c) continue;
              bool match = true;
              for (int e = 0; e < pre; ++e) match = match && (ids[cur - pre + e] == (int64_t)p.seq_tok[o0 + e]);
              if (match) b += p.seq_bias[sq];
            }
            pen = (hist[c >> 5] >> (c & 31)) & 1u;
          }
          v = (pen && p.repetition_penalty != 1.f) ? (b < 0.f ? b * p.repetition_penalty : b / p.repetition_penalty) : b;
          if (p.beam_scores) v += bs;
        } else v = finalv((float)lg[c]);
        const bool after = (v < pv) || (v == pv && c > pi);
        if (after && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
      }
    }
    blk_argmax_after(bv, bi, shv, shi);
    if (tid == 0) {
      cand_vals[(long long)row * k + t] = bv;
      cand_idx[(long long)row * k + t] = bi == 0x7fffffff ? 0x7fffffff : gbase + bi;
    }
    pv = bv; pi = bi;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- beam bookkeeping
__global__ __launch_bounds__(256) void beam_update_kernel(const ns_beam_desc p) {
  const int b = blockIdx.x, nb = p.num_beams, K2 = 2 * nb, ML = p.max_len;
  const int cur = p.cur_len_dev ? *p.cur_len_dev : p.cur_len;
  __shared__ int run_src[MAXQ], run_tok[MAXQ], fin_src[MAXQ], c_beam[2 * MAXQ], c_tok[2 * MAXQ];
  __shared__ float run_sc[MAXQ], fin_sc[MAXQ];
  __shared__ unsigned char fin_dn[MAXQ], c_hit[2 * MAXQ];
  // the inputs of the serial section below come in through LDS, fetched by 3 nb lanes at once (thread 0 used to issue its
  // ~25 dependent global loads one after the other: most of the kernel's 25 us)
  __shared__ float runc[2 * MAXQ], finc[3 * MAXQ];     // LDS, not lane-private arrays: dynamically indexed private arrays live in scratch
  __shared__ unsigned char done_m[3 * MAXQ];
  __shared__ float tv[2 * MAXQ], fin_in[MAXQ];
  __shared__ int ti[2 * MAXQ];
  __shared__ unsigned char fdn_in[MAXQ];
  if ((int)threadIdx.x < K2) {
    tv[threadIdx.x] = p.top_vals[(long long)b * K2 + threadIdx.x];
    ti[threadIdx.x] = p.top_idx[(long long)b * K2 + threadIdx.x];
  } else if ((int)threadIdx.x < K2 + nb) {
    const int i = threadIdx.x - K2;
    fin_in[i] = p.fin_scores_in[b * nb + i];
    fdn_in[i] = p.fin_done_in[b * nb + i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool open = p.open[b] != 0;
    const float lenf = powf((float)(cur + 1 - p.prompt_len), p.length_penalty);
    bool all_hit = true;
    for (int c = 0; c < K2; ++c) {
      c_beam[c] = ti[c] / p.V;
      c_tok[c] = ti[c] - c_beam[c] * p.V;
      const bool hit = (c_tok[c] == p.eos_id) || (cur + 1 >= ML);
      c_hit[c] = hit;
      all_hit = all_hit && hit;
      runc[c] = tv[c] + (hit ? -1.0e9f : 0.f);
      const bool just = hit && c < nb;
      float f = tv[c] / lenf;
      if (!open) f += -1.0e9f;
      if (!just) f += -1.0e9f;
      finc[nb + c] = f;
      done_m[nb + c] = just;
    }
    for (int i = 0; i < nb; ++i) { finc[i] = fin_in[i]; done_m[i] = fdn_in[i]; }
    // next running beams: best nb of runc (value desc, index asc)
    unsigned used = 0;
    for (int r = 0; r < nb; ++r) {
      int best = -1;
      for (int c = 0; c < K2; ++c)
        if (!(used >> c & 1) && (best < 0 || runc[c] > runc[best])) best = c;
      used |= 1u << best;
      run_src[r] = best; run_tok[r] = c_tok[best]; run_sc[r] = runc[best];
    }
    // finished set: best nb of the merged 3nb list
    unsigned usedf = 0;
    for (int r = 0; r < nb; ++r) {
      int best = -1;
      for (int c = 0; c < 3 * nb; ++c)
        if (!(usedf >> c & 1) && (best < 0 || finc[c] > finc[best])) best = c;
      usedf |= 1u << best;
      fin_src[r] = best; fin_sc[r] = finc[best]; fin_dn[r] = done_m[best];
    }
    // early-stop heuristic at the new length
    const float best_run = run_sc[0] / powf((float)(cur + 1 - p.prompt_len), p.length_penalty);
    bool any_empty = false;
    float mn = INFINITY;
    for (int r = 0; r < nb; ++r) { any_empty = any_empty || !fin_dn[r]; mn = fminf(mn, fin_sc[r]); }
    const bool still = open && (any_empty || best_run > mn);
    p.open[b] = still;
    if (still) atomicOr(p.any_open, 1);
    if (!all_hit) atomicOr(p.any_continuation, 1);
  }
  __syncthreads();
  // the 2 nb sequence rows of this batch item, all (row, position) pairs in one flat loop: every load is independent (one loop per
  // row waited for a global round trip per row)
  for (int e = threadIdx.x; e < 2 * nb * ML; e += 256) {
    const int rr = e / ML, t = e - rr * ML;
    if (rr < nb) {
      const int r = rr;
      const int64_t* src = p.run_seqs_in + ((long long)b * nb + c_beam[run_src[r]]) * ML;
      p.run_seqs_out[((long long)b * nb + r) * ML + t] = (t == cur) ? (int64_t)run_tok[r] : src[t];
    } else {
      const int r = rr - nb, fs = fin_src[r];
      int64_t v;
      if (fs < nb) v = p.fin_seqs_in[((long long)b * nb + fs) * ML + t];
      else {
        const int c = fs - nb;
        v = (t == cur) ? (int64_t)c_tok[c] : p.run_seqs_in[((long long)b * nb + c_beam[c]) * ML + t];
      }
      p.fin_seqs_out[((long long)b * nb + r) * ML + t] = v;
    }
  }
  if (threadIdx.x < nb) {
    const int r = threadIdx.x;
    p.run_scores_out[b * nb + r] = run_sc[r];
    p.parent_out[b * nb + r] = b * nb + c_beam[run_src[r]];   // flat row index of the parent beam
    p.next_tok_out[b * nb + r] = run_tok[r];
    p.fin_scores_out[b * nb + r] = fin_sc[r];
    p.fin_done_out[b * nb + r] = fin_dn[r];
  }
}

// ancestry reorder: anc_out[row][0..cur) = anc_in[parent[row]][0..cur); anc_out[row][cur] = row
__global__ void anc_update_kernel(const int* __restrict__ anc_in, int* __restrict__ anc_out, const int* __restrict__ parent,
                                  int rows, int ld, int cur, const int* cur_dev) {
  const int row = blockIdx.x;
  const int c = cur_dev ? *cur_dev : cur;
  const int par = parent ? parent[row] : row;
  for (int t = threadIdx.x; t < c; t += blockDim.x) anc_out[(long long)row * ld + t] = anc_in[(long long)par * ld + t];
  if (threadIdx.x == 0) anc_out[(long long)row * ld + c] = row;
}

__global__ __launch_bounds__(256) void greedy_update_kernel(const float* __restrict__ scores, int V, int64_t* __restrict__ seqs,
                                                            int ld, int cur, const int* cur_dev, int eos, int pad,
                                                            unsigned char* __restrict__ done, int* __restrict__ any_open,
                                                            int64_t* __restrict__ next_tok, const int* __restrict__ best_idx) {
  __shared__ float shv[4];
  __shared__ int shi[4];
  const int row = blockIdx.x;
  const float* s = scores + (long long)row * V;
  float bv = -INFINITY;
  int bi = best_idx ? best_idx[row] : 0x7fffffff;
  for (int c = threadIdx.x; c < (best_idx ? 0 : V); c += 256) {
    const float v = s[c];
    if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = bv; shi[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (!best_idx && (shv[w] > bv || (shv[w] == bv && shi[w] < bi))) { bv = shv[w]; bi = shi[w]; }
    const int c = cur_dev ? *cur_dev : cur;
    int tok = done[row] ? pad : bi;
    seqs[(long long)row * ld + c] = tok;
    next_tok[row] = tok;
    const bool d = done[row] || tok == eos;
    done[row] = d;
    if (!d) atomicOr(any_open, 1);
  }
}

}  // namespace

// Which kernel the ancestry-layout launches (the decode loop's self-attention) take: NS_AD_SELF / ns_debug_set_ad_self =
//   1 (default) by size: one wave per (row, head) once the four-wave form no longer fits the chip in one round (8 workgroups
//     per CU = 2048; tools/probe/attn_self_ab.py: 640 rows x 8 heads 8.3 / 14.2 / 21.1 us against 12.0 / 16.2 / 25.2 at
//     8 / 36 / 68 keys; 128 rows 4.7 / 6.5 / 7.8 against 4.1 / 5.9 / 7.4),  0 never,  2 always (tests).
static int g_ad_self = -1;
extern "C" void ns_debug_set_ad_self(int mode) { g_ad_self = mode < 0 ? 1 : mode; }
static bool self_wave_form(const ns_attn_decode_desc* d) {
  if (g_ad_self < 0) {
    const char* e = getenv("NS_AD_SELF");
    g_ad_self = e ? atoi(e) : 1;
  }
  return g_ad_self >= 2 || (g_ad_self == 1 && (long long)d->groups * d->H > 2048);
}

extern "C" int ns_attn_decode(const ns_attn_decode_desc* d, void* stream) {
  NS_CHECK_ARG(d && d->Q && d->K && d->V && d->O, "ns_attn_decode: null pointer");
  NS_CHECK_ARG(d->head_dim == 64, "ns_attn_decode: head_dim must be 64");
  NS_CHECK_ARG(d->nq >= 1 && d->nq <= MAXQ && d->groups > 0 && d->H > 0 && d->Lk > 0 && d->Lk_max >= d->Lk,
               "ns_attn_decode: bad shape nq=%d groups=%d Lk=%d Lk_max=%d", d->nq, d->groups, d->Lk, d->Lk_max);
  NS_CHECK_ARG(!d->anc || d->nq == 1, "ns_attn_decode: ancestry indirection needs nq == 1");
  NS_CHECK_ARG(!d->Knew || (d->Vnew && d->anc && d->nq == 1 && d->ldnew % 8 == 0), "ns_attn_decode: append needs Vnew, the ancestry layout, nq == 1 and ldnew % 8 == 0");
  NS_CHECK_ARG(d->slot0 >= 0 && (d->slot0 == 0 || d->Knew), "ns_attn_decode: slot0 is the append's slot offset (needs Knew; >= 0)");
  NS_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0, "ns_attn_decode: strides must be multiples of 8");
  dim3 grid(d->groups, d->H);
  hipStream_t st = (hipStream_t)stream;
  if (d->anc && d->nq == 1 && d->ldo % 8 == 0 && self_wave_form(d)) {   // self-attention: one wave per (row, head)
    hipLaunchKernelGGL(attn_self_kernel, dim3(d->groups, (d->H + 3) / 4), dim3(256), 0, st, *d);
    NS_CHECK_LAUNCH("ns_attn_decode");
    return NS_OK;
  }
#define NS_AD(NQ_)                                                                                                \
  case NQ_: {                                                                                                     \
    hipLaunchKernelGGL(attn_decode_kernel<NQ_>, grid, dim3(256), 0, st, *d);                                       \
  } break;
  switch (d->nq) {
    NS_AD(1) NS_AD(2) NS_AD(3) NS_AD(4) NS_AD(5) NS_AD(6) NS_AD(7) NS_AD(8)
  }
#undef NS_AD
  NS_CHECK_LAUNCH("ns_attn_decode");
  return NS_OK;
}

extern "C" int ns_attn_fewq(const ns_attn_fewq_desc* d, void* stream) {
  NS_CHECK_ARG(d && d->Q && d->K && d->Vt && d->O, "ns_attn_fewq: null pointer");
  NS_CHECK_ARG(d->nq >= 1 && d->nq <= 16 && d->groups > 0 && d->H > 0 && d->Lk > 0, "ns_attn_fewq: bad shape nq=%d groups=%d Lk=%d",
               d->nq, d->groups, d->Lk);
  NS_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 32 == 0 && d->ldvt >= d->Lk,
               "ns_attn_fewq: ldq / ldk must be multiples of 8, ldvt a multiple of 32 and >= Lk");
  hipLaunchKernelGGL(attn_fewq_kernel, dim3(d->groups, d->H), dim3(64 * FQ_W), 0, (hipStream_t)stream, *d);
  NS_CHECK_LAUNCH("ns_attn_fewq");
  return NS_OK;
}

extern "C" int ns_vt_pack(const void* v16, int ldv, void* vt16, int groups, int H, int Lk, int ldvt, void* stream) {
  NS_CHECK_ARG(v16 && vt16 && groups > 0 && groups <= 65535 && H > 0 && Lk > 0, "ns_vt_pack: bad arguments");
  NS_CHECK_ARG(ldv % 8 == 0 && ldvt % 32 == 0 && ldvt >= Lk, "ns_vt_pack: ldv must be a multiple of 8, ldvt of 32 and >= Lk");
  hipLaunchKernelGGL(vt_pack_kernel, dim3((ldvt + 63) / 64, H, groups), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)v16, ldv, (half_t*)vt16, H, Lk, ldvt);
  NS_CHECK_LAUNCH("ns_vt_pack");
  return NS_OK;
}

extern "C" int ns_logits_process(const ns_logits_proc_desc* d, void* stream) {
  NS_CHECK_ARG(d && d->logits16 && d->scores32 && d->ids, "ns_logits_process: null pointer");
  NS_CHECK_ARG(d->rows > 0 && d->V > 0 && d->ldv >= d->V && d->no_repeat_ngram >= 0, "ns_logits_process: bad shape");
  NS_CHECK_ARG(d->repetition_penalty > 0.f, "ns_logits_process: repetition_penalty must be > 0");
  NS_CHECK_ARG(d->n_seq >= 0 && (d->n_seq == 0 || (d->seq_tok && d->seq_off && d->seq_bias)), "ns_logits_process: sequence-bias tables missing");
  hipLaunchKernelGGL(logits_process_kernel, dim3(d->rows), dim3(256), 0, (hipStream_t)stream, *d);
  NS_CHECK_LAUNCH("ns_logits_process");
  return NS_OK;
}

extern "C" size_t ns_topk_workspace_bytes(int groups, long long n, int k) {
  const long long nchunks = (n + TOPK_CHUNK - 1) / TOPK_CHUNK;
  return (size_t)groups * nchunks * k * (sizeof(float) + sizeof(int));
}

extern "C" int ns_topk_groups(const float* x, int groups, long long n, int k, float* vals, int* idx, void* workspace,
                              void* stream) {
  NS_CHECK_ARG(x && vals && idx && workspace && groups > 0 && n > 0 && k > 0 && k <= 16 && n < 0x7fffffffLL,
               "ns_topk_groups: bad arguments");
  const int nchunks = (int)((n + TOPK_CHUNK - 1) / TOPK_CHUNK);
  float* cv = (float*)workspace;
  int* ci = (int*)(cv + (size_t)groups * nchunks * k);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(topk_stage1_kernel, dim3(nchunks, groups), dim3(256), 0, st, x, n, k, nchunks, cv, ci);
  hipLaunchKernelGGL(topk_stage2_kernel, dim3(groups), dim3(256), 0, st, cv, ci, nchunks * k, k, vals, idx);
  NS_CHECK_LAUNCH("ns_topk_groups");
  return NS_OK;
}

extern "C" int ns_logits_select(const ns_logits_proc_desc* d, int k, int group_rows, float* cand_vals, int* cand_idx,
                                void* stream) {
  NS_CHECK_ARG(d && d->logits16 && d->ids && cand_vals && cand_idx, "ns_logits_select: null pointer");
  NS_CHECK_ARG(d->rows > 0 && d->V > 0 && d->ldv >= d->V && d->ldv % 8 == 0 && d->ldv <= SEL_MAX_LDV,
               "ns_logits_select: ldv=%d must be a multiple of 8 and <= %d (use ns_logits_process + ns_topk_groups beyond)",
               d->ldv, SEL_MAX_LDV);
  NS_CHECK_ARG(k >= 1 && k <= 16 && group_rows >= 1 && d->cur_len >= 0 && d->cur_len <= d->ids_ld, "ns_logits_select: bad shape");
  NS_CHECK_ARG(d->n_seq == 0 || (d->seq_tok && d->seq_off && d->seq_bias), "ns_logits_select: sequence-bias tables missing");
  NS_CHECK_ARG(d->n_suppress == 0 || d->suppress, "ns_logits_select: suppress list missing");
  NS_CHECK_ARG(d->n_begin_suppress == 0 || d->begin_suppress, "ns_logits_select: begin_suppress list missing");
  hipLaunchKernelGGL(logits_select_kernel, dim3(d->rows), dim3(256), 0, (hipStream_t)stream, *d, k, group_rows, cand_vals, cand_idx);
  NS_CHECK_LAUNCH("ns_logits_select");
  return NS_OK;
}

extern "C" int ns_topk_merge(const float* cand_vals, const int* cand_idx, int groups, int ncand, int k, float* vals, int* idx,
                             void* stream) {
  NS_CHECK_ARG(cand_vals && cand_idx && vals && idx && groups > 0 && ncand > 0 && k >= 1 && k <= 16, "ns_topk_merge: bad arguments");
  hipLaunchKernelGGL(topk_stage2_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, cand_vals, cand_idx, ncand, k, vals, idx);
  NS_CHECK_LAUNCH("ns_topk_merge");
  return NS_OK;
}

extern "C" int ns_beam_update(const ns_beam_desc* d, void* stream) {
  NS_CHECK_ARG(d && d->top_vals && d->top_idx && d->run_seqs_in && d->run_seqs_out && d->fin_seqs_in && d->fin_seqs_out &&
                   d->open && d->any_open && d->any_continuation, "ns_beam_update: null pointer");
  NS_CHECK_ARG(d->num_beams >= 1 && d->num_beams <= MAXQ && d->batch > 0 && d->max_len > 0, "ns_beam_update: bad shape");
  hipLaunchKernelGGL(beam_update_kernel, dim3(d->batch), dim3(256), 0, (hipStream_t)stream, *d);
  NS_CHECK_LAUNCH("ns_beam_update");
  return NS_OK;
}

extern "C" int ns_anc_update(const int* anc_in, int* anc_out, const int* parent, int rows, int ld, int cur, const int* cur_dev,
                             void* stream) {
  NS_CHECK_ARG(anc_in && anc_out && rows > 0 && ld > 0, "ns_anc_update: bad arguments");
  hipLaunchKernelGGL(anc_update_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, anc_in, anc_out, parent, rows, ld, cur,
                     cur_dev);
  NS_CHECK_LAUNCH("ns_anc_update");
  return NS_OK;
}

extern "C" int ns_greedy_update(const float* scores, int rows, int V, int64_t* seqs, int ld, int cur, const int* cur_dev, int eos,
                                int pad, unsigned char* done, int* any_open, int64_t* next_tok, const int* best_idx,
                                void* stream) {
  NS_CHECK_ARG((scores || best_idx) && seqs && done && any_open && next_tok && rows > 0 && V > 0, "ns_greedy_update: bad arguments");
  hipLaunchKernelGGL(greedy_update_kernel, dim3(rows), dim3(best_idx ? 64 : 256), 0, (hipStream_t)stream, scores, V, seqs, ld, cur,
                     cur_dev, eos, pad, done, any_open, next_tok, best_idx);
  NS_CHECK_LAUNCH("ns_greedy_update");
  return NS_OK;
}
