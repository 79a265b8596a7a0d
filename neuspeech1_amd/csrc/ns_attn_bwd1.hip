// Attention backward in ONE pass (head_dim 64, no mask): dQ, dK, dV from a single sweep structure that forms the five
// products S, dP, dV^T, dK^T, dQ once each, where ns_attn.hip's two kernels recompute S and dP (seven products, two
// exponentials per score).  Replaces the autograd backward of HF:modeling_whisper.py:215-238 for the encoder's
// self-attention (Lq = Lk = 1500), where the backward is ~22 % of a training step.
//
// Structure.  One workgroup (8 waves) per (batch, head).  It walks the keys in SWEEPS of 256 (32 keys per wave, the key
// on the MFMA lane as in attn_bwd_dkv_kernel: K / V fragments and the dK^T / dV^T accumulators live in registers for the
// whole sweep) and, inside a sweep, the queries in steps of 64:
//     S^T-orientation:  st[q][key] = Q K^T - lse,  dp[q][key] = dO V^T - delta      (row constants ride in as the
//                       initial accumulators), P = exp2(log2e st) with the scale applied in fp32, dS = P dp
//     dV^T += dO^T P,  dK^T += Q^T dS            (the accumulators are the B operands in place)
//     dQ[64 q][64 d] = dS[64 q][256 keys] K[256 keys][64 d]  sums over the LANE index of dS, so dS crosses LDS once:
//                       every lane stores its 4-query pieces into a [key][query] image and the eight waves read it back
//                       transposed (ds_read_b64_tr_b16) as the A operand of v_mfma_f32_16x16x32_f16, each wave owning a
//                       (32 query x 16 d) part of the tile with its K^T fragments in registers for the whole sweep.
// The dQ partial of a sweep is ADDED to the previous sweeps' sum in a workgroup-private fp32 scratch (same lane, same
// address in every sweep: a plain load / add / store, no atomics, no cross-workgroup reduction, bitwise reproducible); the
// last sweep rounds the sum to fp16 straight into dQ.  Per (batch, head) the scratch is 384 KiB (Lq x 64 floats), written
// and re-read sweep after sweep through L2 / Infinity Cache -- against 1.18 GB of fp32 atomics or slabs per layer for a
// grid of 256-key blocks.
//
// One barrier per step: tiles, dS images and row constants are double-buffered; iteration t forms the scores of step t and
// the dQ product of step t - 1.  Every vector-memory operation of the loop sits in the tail of a half-step, ordered
// consume -> store -> request (gfx950 has ONE in-order counter for loads and stores and hipcc waits vmcnt(0) around
// branches): a request is never waited for in the half-step that issued it.
//
// Measured (B 64, H 8, S 1500, random data, tools/probe/attn_bwd_ab.py, same box): 1.06 ms against 1.27 ms for the two
// passes (0.83x), and 1.6x (|S| ~ 10) to 6x (|S| ~ 40) closer to fp32 autograd than the round-2 two-pass kernels, which
// folded log2(e) into an fp16 operand.  In-kernel stamps (-DNS_AB1_STAMPS): 11 % of a wave's cycles at the step barrier,
// 74 % in the two half-steps, 15 % in the sweeps' prologues / epilogues (K / V fragments, the K image, first tile, dK / dV
// stores).  Waves 4-7 (the second-dispatched half, the loser of every issue arbitration against its SIMD partner) need 1.6x
// the time of waves 0-3 for the first half-step after the barrier re-aligns the two; waves 0-3 then sit at the next barrier.
// What cost the most on the way here, all found in the ISA rather than in the source: 64 two-byte K^T loads per lane that
// hipcc serialised behind a vmcnt(0) each; global addresses hoisted out of the sweep loop as 64-bit pairs, spilled, and
// reloaded one by one; a loaded value COPIED (register rotation) right after its request.
#include "ns_common.h"

namespace {

constexpr int D = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int KB = 256;      // keys per sweep (32 per wave)
constexpr int QT = 64;       // queries per step
constexpr int NT = 512;

// [rows 64][64 halfs] tile image shared by the row reads and the transposed reads (same as ns_attn.hip)
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return 1024 * (row >> 3) + 512 * (chunk >> 2) + 64 * (row & 7) + 16 * ((chunk & 3) ^ ((row >> 2) & 3));
}
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_s4;

__device__ __forceinline__ half8 tr_frag8(const char* tile, int rbase, int cb, int lane) {
  const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
  const int row = rbase + 4 * (g >> 1) + q;
  const int ch = (cb >> 3) + 2 * (g & 1) + (p >> 1);
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + lds_off(row, ch) + 8 * (p & 1)));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + lds_off(row + 8, ch) + 8 * (p & 1)));
  const short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8, r);
}
__device__ __forceinline__ half8 cvt8(const f32x16& x, int base) {
  half8 h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = (half_t)x[base + j];
  return h;
}

// dS image [key 0..255][query 0..63] fp16, 128-B rows, addressed in 8-byte granules (4 queries): granule qg of key row k
// sits at 128 k + 8 (qg ^ s(k)).  s is a bijection on k & 15, so the 16 lanes of a ds_write_b64 group (16 consecutive keys,
// same qg) hit 16 different bank pairs; its bits 2..3 are key bits 1 and 3, so the eight 4-key row pieces a 32-lane half
// reads transposed (keys {0..3} and {8..11} of a 16-key group, or {4..7} and {12..15}) fall into disjoint bank windows.
__device__ __forceinline__ int ds_swz(int k) { return ((k >> 1) & 1) << 2 | ((k >> 3) & 1) << 3 | (k & 1) | ((k >> 2) & 1) << 1; }
__device__ __forceinline__ int ds_off(int k, int qg) { return 128 * k + 8 * (qg ^ ds_swz(k)); }

// delta[b][h][q] = sum_d dO[q][d] O[q][d]   (8 lanes per (row, head)).  Only launched with -DNS_AB1_DELTA_KERNEL: the first sweep of
// attn_bwd1_kernel forms delta itself from the dO chunks its staging threads hold (same arithmetic, same order: bitwise equal).
__global__ __launch_bounds__(256) void attn_delta_kernel(const ns_attn_desc p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long item = t >> 3;                 // (row, head)
  const int part = (int)(t & 7);
  const long long rows = (long long)p.B * p.Lq;
  const bool ok = item < rows * p.H;
  const long long row = ok ? item / p.H : 0;
  const int h = ok ? (int)(item % p.H) : 0;
  float s = 0.f;
  if (ok) {
    const half8 a = *(const half8*)((const half_t*)p.dO + row * p.lddo + h * D + 8 * part);
    const half8 o = *(const half8*)((const half_t*)p.O + row * p.ldo + h * D + 8 * part);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)o[j];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (ok && part == 0) {
    const long long b = row / p.Lq, q = row % p.Lq;
    p.Delta[(b * p.H + h) * p.Lq + q] = s;
  }
}


// global accesses as (wave-uniform base pointer) + (32-bit byte offset of the lane): one VGPR per address instead of a 64-bit
// pair (hipcc spilled those pairs around the sweep loop and reloaded each behind its own s_waitcnt vmcnt(0): 25 serialised
// round trips, ~50 k cycles per sweep).  Every offset stays far below 4 GiB: they are relative to the (batch, head) base.
template <class T>
__device__ __forceinline__ T ns_ld(const void* base, uint32_t byte_off) { return *(const T*)((const char*)base + byte_off); }
template <class T>
__device__ __forceinline__ void ns_st(void* base, uint32_t byte_off, const T& v) { *(T*)((char*)base + byte_off) = v; }

struct I0_ { static constexpr int value = 0; };
struct I1_ { static constexpr int value = 1; };
struct T_ { static constexpr bool value = true; };
struct F_ { static constexpr bool value = false; };

// One sweep (256 keys) of one (batch, head).  first: no earlier partial dQ to add; last: the sum leaves as fp16 dQ.
#ifdef NS_AB1_STAMPS
#define NS_AB1_T() __builtin_amdgcn_s_memtime()
#else
#define NS_AB1_T() 0ull
#endif
__device__ __forceinline__ void sweep(const ns_attn_desc& p, char* smem, float* scr, int kb0, int b, int h, int nsteps,
                                      const bool first, const bool last, unsigned long long* tacc) {
  char* const DS0 = smem + 32768;
  float* const rc0 = (float*)(smem + 32768 + 2 * KB * 128);
  // (the thread index is laundered once per sweep: everything derived from it -- LDS and global lane offsets -- would
  // otherwise be hoisted out of the sweep loop as loop-invariant, kept live across it and SPILLED, each reload behind
  // its own s_waitcnt vmcnt(0); recomputing them per sweep is a handful of VALU instructions)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
  const int l15 = lane & 15, lg = lane >> 4;
  // wave-uniform bases of this (batch, head)
  const half_t* const Q = (const half_t*)p.Q + (long long)b * p.Lq * p.ldq + h * D;
  const half_t* const K = (const half_t*)p.K + (long long)b * p.Lk * p.ldk + h * D;
  const half_t* const V = (const half_t*)p.V + (long long)b * p.Lk * p.ldv + h * D;
  const half_t* const dO = (const half_t*)p.dO + (long long)b * p.Lq * p.lddo + h * D;
  half_t* const dQ = (half_t*)p.dQ + (long long)b * p.Lq * p.lddq + h * D;
  half_t* const dK = (half_t*)p.dK + (long long)b * p.Lk * p.lddk + h * D;
  half_t* const dV = (half_t*)p.dV + (long long)b * p.Lk * p.lddv + h * D;
  const float* const LSE = p.LSE + ((long long)b * p.H + h) * p.Lq;
  const float* const Delta = p.Delta + ((long long)b * p.H + h) * p.Lq;
  const int qh = wave & 1, dq = wave >> 1;
  const uint32_t scr_lane = (uint32_t)(wave * 128 + lane) * 16u;     // + (step * 16 + mt) * 1024 bytes
  const int srow = tid >> 3, sch = tid & 7;
  const int s_off = lds_off(srow, sch);

  const int key = kb0 + wave * 32 + lr;
  const int krow = min(key, p.Lk - 1);
  const bool keyok = key < p.Lk;
  half8 kf[4], vf[4];
#ifdef NS_AB1_DIRECT_LOAD
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = ns_ld<half8>(K, 2u * ((uint32_t)krow * p.ldk + 16 * s + 8 * lh));
    vf[s] = ns_ld<half8>(V, 2u * ((uint32_t)krow * p.ldv + 16 * s + 8 * lh));
  }
#else
  (void)krow;
#endif
  // K^T fragments of the dQ product (B operand of 16x16x32: lane n = d column, 8 consecutive keys per k-group): the sweep's
  // K rows go through LDS once (coalesced 16-B loads into the second dS image, which no step has touched yet) and come
  // back transposed.  Keys past Lk (last sweep) enter as ZERO rows: whatever their dS holds, they add nothing to dQ.
  // (Fetched straight from global memory these are 64 two-byte loads per lane, which hipcc serialised behind one
  // s_waitcnt vmcnt(0) each: ~0.4 ms per launch.)
  half8 ktf[8];
  const half_t* const Oin = (const half_t*)p.O + (long long)b * p.Lq * p.ldo + h * D;
  uint4 qreg, dreg, oreg = {0u, 0u, 0u, 0u};
  float lse_r = 0.f, del_r = 0.f;
  auto load_tile = [&](int q0) __attribute__((always_inline)) {
    const uint32_t rr = (uint32_t)min(q0 + srow, p.Lq - 1);
    qreg = ns_ld<uint4>(Q, 2u * (rr * p.ldq + sch * 8));
    dreg = ns_ld<uint4>(dO, 2u * (rr * p.lddo + sch * 8));
    if (first) oreg = ns_ld<uint4>(Oin, 2u * (rr * p.ldo + sch * 8));      // first sweep: delta is formed here (see store_tile)
    if (tid < 64) {
      const uint32_t qq = (uint32_t)min(q0 + tid, p.Lq - 1);
      lse_r = ns_ld<float>(LSE, 4u * qq);
      if (!first) del_r = ns_ld<float>(Delta, 4u * qq);
    }
  };
  int q_first = 0;
  asm volatile("" : "+s"(q_first));
  {
    char* const KT = DS0 + KB * 128;
    // every request of the sweep's prologue goes out before anything waits: the K and V rows, then the first tile.  All as 128-B row
    // segments (fragment loads straight into registers -- lane = key row, 32 rows x 32 B per instruction -- are bound by the CU's request
    // rate): the rows go to LDS images and the fragments are read from there.  K twice: the plain image for the transposed reads of the
    // K^T fragments (zero rows past Lk), and a swizzled one (first dS image, idle until step 0 writes it) for the row fragments; V into
    // the two tile buffers (idle until the first tile is stored, one barrier later).
    char* const KS = DS0;
    char* const VS = smem;
    uint4 kv[4], vv[4];
    const int l16p = tid & 15;
    const int prow = 2 * (tid >> 4) + ((l16p >> 2) & 1), pch = (l16p & 3) + 4 * (l16p >> 3);     // + 64 i rows (see rm_lane in ns_attn.hip)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t kk = (uint32_t)min(kb0 + 64 * i + prow, p.Lk - 1);
      kv[i] = ns_ld<uint4>(K, 2u * (kk * p.ldk + pch * 8));
#ifndef NS_AB1_DIRECT_LOAD
      vv[i] = ns_ld<uint4>(V, 2u * (kk * p.ldv + pch * 8));
#endif
    }
    load_tile(q_first);
    __syncthreads();            // the previous sweep's last reads of the images and tile buffers are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 64 * i + prow;
      const bool ok = kb0 + row < p.Lk;
      uint4 v = kv[i];
#ifndef NS_AB1_DIRECT_LOAD
      *(uint4*)(KS + 8192 * i + lds_off(prow, pch)) = v;
      *(uint4*)(VS + 8192 * i + lds_off(prow, pch)) = vv[i];
#endif
      v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
      *(uint4*)(KT + (row * 8 + pch) * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = 32 * ks + 8 * lg + (l15 >> 2);
      const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(KT + 128 * k0 + 32 * dq + 8 * (l15 & 3)));
      const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(KT + 128 * (k0 + 4) + 32 * dq + 8 * (l15 & 3)));
      const short8v a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      ktf[ks] = __builtin_bit_cast(half8, a);
    }
#ifndef NS_AB1_DIRECT_LOAD
    {
      const int fr = wave * 32 + lr;        // this lane's key row of the sweep
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf[s] = *(const half8*)(KS + 8192 * (fr >> 6) + lds_off(fr & 63, 2 * s + lh));
        vf[s] = *(const half8*)(VS + 8192 * (fr >> 6) + lds_off(fr & 63, 2 * s + lh));
      }
    }
    __syncthreads();            // the V image lies in the tile buffers: every wave has its fragments before the first tile is stored
#endif
  }
  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[t][r] = 0.f; dvt[t][r] = 0.f; }

  auto store_tile = [&](int buf, int q0) __attribute__((always_inline)) {
    *(uint4*)(smem + buf * 16384 + s_off) = qreg;
    *(uint4*)(smem + buf * 16384 + 8192 + s_off) = dreg;
    if (first) {
      // delta[q] = sum_d dO[q][d] O[q][d] of the tile's rows, from the chunks the staging threads hold anyway (eight lanes per
      // row): the separate delta launch (33 us, 196 MB) is gone.  The first sweep stores it for the later ones.
      const half8 a = __builtin_bit_cast(half8, dreg), o = __builtin_bit_cast(half8, oreg);
      float sdot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sdot += (float)a[j] * (float)o[j];
      sdot += __shfl_xor(sdot, 1, 64);
      sdot += __shfl_xor(sdot, 2, 64);
      sdot += __shfl_xor(sdot, 4, 64);
      if (sch == 0) {
        rc0[buf * 128 + 64 + srow] = -sdot;
        if (q0 + srow < p.Lq) ns_st<float>(p.Delta + ((long long)b * p.H + h) * p.Lq, 4u * (uint32_t)(q0 + srow), sdot);
      }
    }
    if (tid < 64) {
      // (wave 0: the lane id is recomputed with v_mbcnt here -- derived from the long-lived thread index the address was
      // spilled, and its reload sat behind an s_waitcnt vmcnt(0) once per step)
      const int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const bool qok = q0 + l < p.Lq;
      rc0[buf * 128 + l] = qok ? -lse_r : -INFINITY;        // queries past Lq: P = exp2(-inf) = 0, so dS = 0 too
      if (!first) rc0[buf * 128 + 64 + l] = -del_r;
    }
  };
  float4 nrun0 = {0.f, 0.f, 0.f, 0.f}, nrun1 = {0.f, 0.f, 0.f, 0.f};    // running dQ sums in flight (see the step's tail)
  // iteration t: scores of step t (HB) and the dQ product of step t - 1 (HA)
  auto step = [&](auto HA_, auto HB_, int t) __attribute__((always_inline)) {
    constexpr bool HA = decltype(HA_)::value, HB = decltype(HB_)::value;
    const int q0 = t * QT, cur = t & 1;
    const unsigned long long t_b = NS_AB1_T();
    __syncthreads();
    const unsigned long long t_a = NS_AB1_T();
    tacc[0] += t_a - t_b;                 // parked at the barrier
    const char* const Qs = smem + cur * 16384;
    const char* const dOs = Qs + 8192;
    char* const DSw = DS0 + cur * (KB * 128);
    const char* const DSr = DS0 + (cur ^ 1) * (KB * 128);
    const float* const lse_s = rc0 + cur * 128;
    const float* const del_s = lse_s + 64;
    // the two 32-query halves of the step, as two straight-line copies (static registers for the values in flight, exact
    // waits) kept apart by a scheduling fence: interleaved by hipcc they need both halves' scores live at once and spill
    auto half = [&](auto QT_) __attribute__((always_inline)) {
      constexpr int qt = decltype(QT_)::value;
      f32x16 st, dp;
      if constexpr (HB) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 nl = *(const float4*)(lse_s + qt * 32 + 8 * g + 4 * lh);
          const float4 nd = *(const float4*)(del_s + qt * 32 + 8 * g + 4 * lh);
          st[4 * g + 0] = nl.x; st[4 * g + 1] = nl.y; st[4 * g + 2] = nl.z; st[4 * g + 3] = nl.w;
          dp[4 * g + 0] = nd.x; dp[4 * g + 1] = nd.y; dp[4 * g + 2] = nd.z; dp[4 * g + 3] = nd.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const half8 aq = *(const half8*)(Qs + lds_off(qt * 32 + lr, 2 * s + lh));
          const half8 ad = *(const half8*)(dOs + lds_off(qt * 32 + lr, 2 * s + lh));
          st = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, kf[s], st, 0, 0, 0);   // S[q][key] - lse[q]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ad, vf[s], dp, 0, 0, 0);   // dP[q][key] - delta[q]
        }
      }
      // ---- dQ of the PREVIOUS step, 16-row tile mt = qt of this wave's part: rows 32 qh + 16 mt + (0..15), columns
      // 16 dq + (0..15), summed over the sweep's 256 keys; added to the earlier sweeps' sum (same lane, same address)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HA) {
        const int qg = 4 * (2 * qh + qt) + (l15 & 3);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int k0 = 32 * ks + 8 * lg + (l15 >> 2);
          const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(DSr + ds_off(k0, qg)));
          const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(DSr + ds_off(k0 + 4, qg)));
          const short8v a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          // K^T on the A port, dS on the B port: the lane then holds 4 consecutive d of ONE query row (dQ^T[d = 4 lg + e][q = l15]) -- one 8-byte
          // store per lane in the last sweep, where the other orientation gave four 2-byte stores to four rows
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktf[ks], __builtin_bit_cast(half8, a), acc, 0, 0, 0);
        }
      }
      if constexpr (HB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(st[r] * LOG2E);   // the scale stays in fp32: P is the forward's P
          st[r] = pv;
          dp[r] = pv * dp[r];       // dS
        }
  #pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half8 pb = cvt8(st, 8 * s2);
          const half8 dsb = cvt8(dp, 8 * s2);
          const int sg = qt * 2 + s2;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const half8 a1 = tr_frag8(dOs, 16 * sg, dt * 32, lane);
            const half8 a2 = tr_frag8(Qs, 16 * sg, dt * 32, lane);
            dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pb, dvt[dt], 0, 0, 0);
            dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, dsb, dkt[dt], 0, 0, 0);
          }
        }
          // dS -> the [key][query] image: register 4g .. 4g+3 = queries 32 qt + 8 g + 4 lh + (0..3) of this lane's key
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const half4 v = {(half_t)dp[4 * g], (half_t)dp[4 * g + 1], (half_t)dp[4 * g + 2], (half_t)dp[4 * g + 3]};
          *(half4*)(DSw + ds_off(wave * 32 + lr, 8 * qt + 2 * g + lh)) = v;
        }
      }
      // ---- tail of the half-step: every vector-memory operation of the loop lives here, in an order that never waits on
      // something just issued (gfx950 counts loads and stores in ONE in-order counter, and around branches hipcc waits
      // vmcnt(0)): first CONSUME what was requested half a step or a step ago, then issue the stores, then the new loads.
      if (qt == 1 && t + 1 < nsteps) store_tile(cur ^ 1, q0 + QT);       // tile t + 1: requested at the end of step t - 1
      float4 v = {acc[0], acc[1], acc[2], acc[3]};
      if constexpr (HA) {       // + the earlier sweeps' sum of the same lane at the same address (requested one step ago)
        const float4 run = qt ? nrun1 : nrun0;      // compile-time choice
        v.x = first ? v.x : v.x + run.x; v.y = first ? v.y : v.y + run.y;
        v.z = first ? v.z : v.z + run.z; v.w = first ? v.w : v.w + run.w;
      }
      if constexpr (HA) {
#ifdef NS_AB1_NOSCR   /* diagnostic: no scratch traffic (wrong dQ) */
        if (false) {
#else
        if (!last) {
#endif
          ns_st<float4>(scr, scr_lane + (uint32_t)((t - 1) * 16 + qt) * 1024u, v);
        } else {
          const int qr = q0 - QT + 32 * qh + 16 * qt + l15;
          const half4 o = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
          if (qr < p.Lq) ns_st<half4>(dQ, 2u * ((uint32_t)qr * p.lddq + 16 * dq + 4 * lg), o);
        }
      }
      if constexpr (HB) {       // used by the next iteration's half qt (a register pair per half: no copy of a value in flight)
#ifdef NS_AB1_NOSCR
        if (false) {
#else
        if (!first) {
#endif
          if (qt == 0) nrun0 = ns_ld<float4>(scr, scr_lane + (uint32_t)(t * 16 + 0) * 1024u);
          else nrun1 = ns_ld<float4>(scr, scr_lane + (uint32_t)(t * 16 + 1) * 1024u);
        }
      }
      if (qt == 1 && t + 2 < nsteps) load_tile(q0 + 2 * QT);
    };
    half(I0_{});
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t_m = NS_AB1_T();
    half(I1_{});
    const unsigned long long t_e = NS_AB1_T();
    tacc[1] += t_m - t_a;                 // first half
    tacc[2] += t_e - t_m;                 // second half
  };
  store_tile(0, q_first);
  if (nsteps > 1) load_tile(q_first + QT);
  // (the step index of the two peeled iterations is laundered through an SGPR: as compile-time constants their global
  // addresses are loop-invariant 64-bit pairs that hipcc hoists out of the sweep loop, spills, and reloads one by one)
  int t_first = 0, t_last = nsteps;
  asm volatile("" : "+s"(t_first));
  asm volatile("" : "+s"(t_last));
  step(F_{}, T_{}, t_first);
  for (int t = 1; t < nsteps; ++t) step(T_{}, T_{}, t);
  step(T_{}, F_{}, t_last);

  // dK, dV of the sweep's 256 keys: a lane holds 64 d-values of ITS key; written straight from there they are 8-byte pieces (16 B contiguous
  // per row and instruction), and the CU retires such pieces at its request rate, not at HBM's -- the next sweep's loads queue behind them
  // (one in-order vmcnt).  Through the wave's own 32 rows of the (now idle) tile buffers they leave as 128-B row segments.
#ifdef NS_AB1_DIRECT_STORE
  if (keyok) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 ok_, ov_;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ok_[e] = (half_t)dkt[dt][4 * g + e]; ov_[e] = (half_t)dvt[dt][4 * g + e]; }
        ns_st<half4>(dK, 2u * ((uint32_t)key * p.lddk + dt * 32 + 8 * g + 4 * lh), ok_);
        ns_st<half4>(dV, 2u * ((uint32_t)key * p.lddv + dt * 32 + 8 * g + 4 * lh), ov_);
      }
  }
#else
  (void)keyok;
  __syncthreads();            // every wave is through its last step: tile buffers and images are idle
  {
    char* const stg = smem + wave * 4096;          // 32 rows x 128 B of this wave, lds_off layout (8 waves: the two tile buffers)
    const int l16 = lane & 15;
    const int srow2 = 2 * (lane >> 4) + ((l16 >> 2) & 1), schunk2 = (l16 & 3) + 4 * (l16 >> 3);    // + 8 i rows: see rm_lane in ns_attn.hip
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(which ? dvt[dt][4 * g + e] : dkt[dt][4 * g + e]);
          *(half4*)(stg + lds_off(lr, 4 * dt + g) + 8 * lh) = o;
        }
      half_t* const dst = which ? dV : dK;
      const uint32_t ldd = (uint32_t)(which ? p.lddv : p.lddk);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 v = *(const uint4*)(stg + lds_off(8 * i + srow2, schunk2));
        const int krow2 = kb0 + wave * 32 + 8 * i + srow2;
        if (krow2 < p.Lk) ns_st<uint4>(dst, 2u * ((uint32_t)krow2 * ldd + schunk2 * 8), v);
      }
    }
  }
#endif
}

__device__ __forceinline__ size_t ns_attn_bwd_workspace_bytes_dev(const ns_attn_desc& p) {
  return (size_t)p.B * p.H * ((p.Lq + QT - 1) / QT) * QT * D * sizeof(float);
}

__global__ __launch_bounds__(NT, 2) void attn_bwd1_kernel(const ns_attn_desc p, float* __restrict__ scratch) {
  // LDS: two Q / dO tile pairs, two dS images, two sets of row constants.  Everything is double-buffered so that a step needs
  // ONE barrier: step t computes its scores from tile buffer t & 1 into dS image t & 1 while the same waves form the dQ
  // product of step t - 1 from image (t - 1) & 1 -- two independent instruction streams in one basic block (the small
  // dQ MFMAs and their transposed reads fill the exponentials' issue slots), and the tile of step t + 1 is stored meanwhile.
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + 2 * KB * 128 + 1024];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int nsteps = (p.Lq + QT - 1) / QT;
  float* const scr = scratch + (long long)bh * nsteps * (QT * D);   // this (batch, head)'s part: [step][wave][mt][lane] float4
  unsigned long long tacc[3] = {0ull, 0ull, 0ull};
  const unsigned long long t_start = NS_AB1_T();
  (void)t_start;
  for (int kb0 = 0; kb0 < p.Lk; kb0 += KB) {
    const bool first = kb0 == 0, last = kb0 + KB >= p.Lk;
    sweep(p, smem, scr, kb0, b, h, nsteps, first, last, tacc);
  }
#ifdef NS_AB1_STAMPS
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* out = (unsigned long long*)((char*)scratch + ns_attn_bwd_workspace_bytes_dev(p)) + ((long long)bh * 8 + (threadIdx.x >> 6)) * 4;
    out[0] = tacc[0]; out[1] = tacc[1]; out[2] = tacc[2]; out[3] = NS_AB1_T() - t_start;
  }
#endif
}

}  // namespace

// scratch floats: B * H * ceil(Lq / 64) * 64 * 64
extern "C" size_t ns_attn_bwd_workspace_bytes(int B, int H, int Lq, int Lk, int causal) {
  // unmasked attention with many queries: the encoder's self-attention.  The decoder's cross-attention (45 queries, 1500 keys:
  // one step per sweep) was measured 0.200 ms through THIS kernel against 0.166 ms for the two passes -- six serial sweep
  // prologues per workgroup and only B x H workgroups -- and has a one-pass kernel of its own; causal / short-key shapes stay with
  // the two-pass kernels.
  if (causal || Lk < 256) return 0;
  // few queries (the decoder's cross-attention): attn_bwd_fewq_kernel in ns_attn.hip, fp32 dQ slabs, one per group of 128 x NS_FEWQ_KPW keys
  if (Lq <= 64) return (size_t)((Lk + 128 * NS_FEWQ_KPW - 1) / (128 * NS_FEWQ_KPW)) * B * H * Lq * D * sizeof(float);
  if (Lq < 256) return 0;
  return (size_t)B * H * ((Lq + QT - 1) / QT) * QT * D * sizeof(float);
}

int ns_attn_bwd1_launch(const ns_attn_desc* d, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const size_t need = ns_attn_bwd_workspace_bytes(d->B, d->H, d->Lq, d->Lk, d->causal);
  NS_CHECK_ARG(need > 0 && workspace && workspace_bytes >= need, "ns_attn_bwd (one pass): workspace of %zu bytes needed, %zu given",
               need, workspace_bytes);
  const long long items = (long long)d->B * d->Lq * d->H * 8;
#ifdef NS_AB1_DELTA_KERNEL
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, *d);   // (the kernel's first sweep forms delta itself)
#else
  (void)items;
#endif
  hipLaunchKernelGGL(attn_bwd1_kernel, dim3(d->B * d->H), dim3(NT), 0, st, *d, (float*)workspace);
  NS_CHECK_LAUNCH("ns_attn_bwd (one pass)");
  return NS_OK;
}
