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
// One barrier per step: tiles (three buffers), dS images and row constants (two) are multi-buffered; iteration t forms the scores of
// step t and the dQ product of step t - 1.  Every vector-memory operation of the loop sits in the tail of a half-step, ordered
// consume -> store -> request (gfx950 has ONE in-order counter for loads and stores and hipcc waits vmcnt(0) around
// branches): a request is never waited for in the half-step that issued it.
//
// Measured (B 64, H 8, S 1500, random data, tools/probe/attn_bwd_ab.py, same box).  Round 3: 1.06 ms against 1.27 ms for the two
// passes (0.83x), and 1.6x (|S| ~ 10) to 6x (|S| ~ 40) closer to fp32 autograd than the round-2 two-pass kernels, which
// folded log2(e) into an fp16 operand.  What cost the most on the way there, all found in the ISA rather than in the source: 64
// two-byte K^T loads per lane that hipcc serialised behind a vmcnt(0) each; global addresses hoisted out of the sweep loop as 64-bit
// pairs, spilled, and reloaded one by one; a loaded value COPIED (register rotation) right after its request.
//   Round 4: 0.954 - 0.962 ms against 0.996 - 1.000 for round 3's kernel (three boxes, interleaved).  The Q / dO tiles go from global
// memory straight into their LDS images (buffer_load ... lds) instead of through 12 registers per thread, delta is formed by each wave
// from its own pieces, the step barrier is a bare s_barrier, and waves 4 - 7 run the first half-step at raised priority.  In-kernel
// stamps (-DNS_AB1_STAMPS), round 3 -> round 4: 902 k -> 849 k cycles per workgroup, at the step barrier 11 % -> 4.7 % (round 3: waves
// 4 - 7, the second-dispatched half and the loser of every issue arbitration against their SIMD partners, needed 1.5x the partners'
// time for the first half-step, and the partners sat at the barrier 16 - 20 % of the time), in the half-steps 74 % -> 84 %, in the
// sweeps' prologues / epilogues 15 % -> 11 %.  The counters (round 3's kernel): MFMA busy 41 %, 41 % of the wave cycles parked at an
// s_waitcnt or the barrier, 29 % stalled at issue.
//   Tried in round 4 and dropped (kept: tools/probe/attic/ns_attn_bwd1_stagger.hip.txt): the two waves of a SIMD run the same program
// from the same barrier and want the matrix pipe and the vector pipe in the same moments, so waves 4 - 7 were given a program whose
// dV^T / dK^T products run one half-step late (P and dS of a half kept in 16 registers, their K^T fragments re-read from a kept LDS
// image to pay for them, four tile buffers).  Bit-identical, no spills after some persuasion -- and 1.07 - 1.12 ms: the late waves
// became the critical path (54 % of the kernel in their first half-steps, their partners 35 % at the barrier), with or without
// raised priority.  A plain delay of waves 4 - 7 behind the barrier (s_sleep 4 / 8 / 14) bought 1 %, 1 %, -1 %.
#include "ns_common.h"
#include <mutex>

namespace {

constexpr int D = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int KB = 256;      // keys per sweep (32 per wave)
constexpr int QT = 64;       // queries per step
constexpr int NT = 512;
constexpr int TILES = 3 * 16384;                         // three Q / dO tile pairs (see the tile transport in sweep())
constexpr int O_OFF = TILES + 2 * KB * 128 + 1024;       // first sweep: the O rows of the tile in flight (8 waves x 1 KiB), for delta
constexpr int LDS_TOTAL = O_OFF + 8192;                  // 121 KiB
typedef __attribute__((address_space(3))) void lds_void;

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
// attn_bwd1_kernel forms delta itself from the dO and O rows each wave fetches (same arithmetic, same order: bitwise equal).
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

// the sweep's 256 K and V rows as 16-B chunks, four of each per thread: rows 64 i + 2 (tid >> 4) + .., chunk .. (see rm_lane in ns_attn.hip)
__device__ __forceinline__ void ns_attn_bwd1_request_kv(const ns_attn_desc& p, const half_t* K, const half_t* V, int kb0, int tid,
                                                        uint4 (&kv)[4], uint4 (&vv)[4]) {
  const int l16p = tid & 15;
  const int prow = 2 * (tid >> 4) + ((l16p >> 2) & 1), pch = (l16p & 3) + 4 * (l16p >> 3);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t kk = (uint32_t)min(kb0 + 64 * i + prow, p.Lk - 1);
    kv[i] = ns_ld<uint4>(K, 2u * (kk * p.ldk + pch * 8));
    vv[i] = ns_ld<uint4>(V, 2u * (kk * p.ldv + pch * 8));
  }
}

// One sweep (256 keys) of one (batch, head).  first: no earlier partial dQ to add; last: the sum leaves as fp16 dQ.
#ifdef NS_AB1_STAMPS
#define NS_AB1_T() __builtin_amdgcn_s_memtime()
#else
#define NS_AB1_T() 0ull
#endif
// kv / vv: this sweep's K / V row chunks, requested by the caller (first sweep) or by the previous sweep before its dK / dV left --
// the requests' latency then runs beside that epilogue instead of at the head of this prologue; the sweep leaves the next sweep's in them
__device__ __forceinline__ void sweep(const ns_attn_desc& p, char* smem, float* scr, int kb0, int b, int h, int nsteps,
                                      const bool first, const bool last, unsigned long long* tacc, uint4 (&kv)[4], uint4 (&vv)[4]) {
  char* const DS0 = smem + TILES;
  float* const rc0 = (float*)(smem + TILES + 2 * KB * 128);
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
  // ---- tile transport.  The 64 x 64 Q and dO tiles of a step go from global memory STRAIGHT into their LDS image (buffer_load ... lds,
  // 1 KiB per wave and tile: rows 8 w .. 8 w + 7, the image's own 1-KiB unit; lane l fetches the 16-B chunk that belongs at byte 16 l of
  // it), requested a full step before the step that reads them.  Through registers (round 3) a tile cost every thread three loads, two
  // LDS stores and twelve registers that lived a whole step with nothing to do but wait.  Tile i lives in buffer (i + 2) % 3: at the tail
  // of step t the pieces of tile t + 2 go to the buffer of tile t - 1, whose last readers are a barrier away.
  //   First sweep: the wave also fetches ITS eight rows of O (a ninth KiB, one slot per wave), and forms delta = rowsum(dO o O) of those
  // rows itself once its own pieces have landed -- its own transfers need no barrier to be seen -- eight lanes per row as round 3 did
  // from registers: same arithmetic in the same order.  The row constants (-lse, and -delta in the later sweeps: 64 + 64 floats a step)
  // still pass through two registers of wave 0, which negates them.
  //   Issued as inline assembly: behind the builtin hipcc orders LDS reads after every piece it knows to be in flight (an s_waitcnt
  // vmcnt(0) in front of the first read after the step's barrier, i.e. a wait for the pieces requested a moment ago).  The waits are
  // placed by hand; hipcc's own counted waits for its register loads only get stricter from requests it cannot see.
  const ns_u4v rsrc_q = {(uint32_t)(uintptr_t)Q, (uint32_t)((uintptr_t)Q >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
  const ns_u4v rsrc_do = {(uint32_t)(uintptr_t)dO, (uint32_t)((uintptr_t)dO >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
  const ns_u4v rsrc_o = {(uint32_t)(uintptr_t)Oin, (uint32_t)((uintptr_t)Oin >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;
  auto dma16 = [&](const ns_u4v& rsrc, uint32_t lds_byte, uint32_t voff) __attribute__((always_inline)) {
    // (m0 is saved and restored around the transfer, as in ns_gemm_tn256.hip: the register is reserved -- it cannot be named as a clobber -- and
    // hipcc may keep a value of its own in it)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
  };
  // lane l of a piece holds row 8 w + ((l >> 2) & 7), chunk 4 (l >> 5) + ((l & 3) ^ swizzle(row)): byte 16 l of the unit in lds_off's layout
  const int prow_l = 8 * wave + ((lane >> 2) & 7);
  const int pchunk_l = 4 * (lane >> 5) + ((lane & 3) ^ ((prow_l >> 2) & 3));
  auto dma_tile = [&](int tbuf, int q0) __attribute__((always_inline)) {
    const uint32_t rr = (uint32_t)min(q0 + prow_l, p.Lq - 1);
    const uint32_t dst = smem_base + (uint32_t)(tbuf * 16384 + wave * 1024);
    dma16(rsrc_q, dst, 2u * (rr * p.ldq + pchunk_l * 8));
    dma16(rsrc_do, dst + 8192u, 2u * (rr * p.lddo + pchunk_l * 8));
    if (first) dma16(rsrc_o, smem_base + (uint32_t)(O_OFF + wave * 1024), 2u * (rr * p.ldo + pchunk_l * 8));
  };
  float lse_r = 0.f, del_r = 0.f;
  auto load_rc = [&](int q0) __attribute__((always_inline)) {
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
    // tile buffers 0 and 1 (tile 0 goes to buffer 2 meanwhile, tile 1 to buffer 0 once the V fragments are out).
    char* const KS = DS0;
    char* const VS = smem;
    const int l16p = tid & 15;
    const int prow = 2 * (tid >> 4) + ((l16p >> 2) & 1), pch = (l16p & 3) + 4 * (l16p >> 3);     // + 64 i rows (see rm_lane in ns_attn.hip)
    load_rc(q_first);
    __syncthreads();            // the previous sweep's last reads of the images and tile buffers are done
    dma_tile(2, q_first);
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
    __syncthreads();            // the V image lies in tile buffers 0 and 1: every wave has its fragments before tile 1 is requested
#endif
  }
  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[t][r] = 0.f; dvt[t][r] = 0.f; }

  // row constants of the tile in buffer tbuf (its pieces have landed) -> set `buf`
  auto store_rc = [&](int tbuf, int buf, int q0) __attribute__((always_inline)) {
    if (first) {
      // delta[q] = sum_d dO[q][d] O[q][d] of this wave's eight rows, eight lanes per row, from its own pieces.  The first sweep stores
      // it for the later ones.
      const half8 a = *(const half8*)(smem + tbuf * 16384 + 8192 + wave * 1024 + 16 * lane);
      const half8 o = *(const half8*)(smem + O_OFF + wave * 1024 + 16 * lane);
      float sdot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sdot += (float)a[j] * (float)o[j];
      // the row's eight chunks sit in lanes l, l ^ 1, l ^ 2, l ^ 3 (chunks 0 .. 3, swizzled) and the same four + 32 (chunks 4 .. 7).  Round 3
      // summed chunk c with c ^ 1, then ^ 2, then ^ 4: the image's swizzle permutes chunks 0 .. 3 among themselves by an XOR, which
      // maps the pairs {c, c ^ 1} and {c, c ^ 2} onto each other -- the same three levels of the same tree.
      sdot += __shfl_xor(sdot, 1, 64);
      sdot += __shfl_xor(sdot, 2, 64);
      sdot += __shfl_xor(sdot, 32, 64);
      if ((lane & 35) == 0) {
        rc0[buf * 128 + 64 + prow_l] = -sdot;
        if (q0 + prow_l < p.Lq) ns_st<float>(p.Delta + ((long long)b * p.H + h) * p.Lq, 4u * (uint32_t)(q0 + prow_l), sdot);
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
  auto step = [&](auto HA_, auto HB_, int t, int tb) __attribute__((always_inline)) {      // tb = (t + 2) % 3: the buffer of tile t
    constexpr bool HA = decltype(HA_)::value, HB = decltype(HB_)::value;
    const int q0 = t * QT, cur = t & 1;
    const int tb_next = tb == 2 ? 0 : tb + 1, tb_prev = tb == 0 ? 2 : tb - 1;
    const unsigned long long t_b = NS_AB1_T();
    // (a bare s_barrier behind the wave's own LDS operations: __syncthreads() is a fence too, and with memory clobbers in flight hipcc
    // turns that fence into s_waitcnt vmcnt(0) -- a memory round trip per step.  What the step reads behind this barrier was waited for
    // explicitly: the tile pieces in the previous step's tail, LDS stores here.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned long long t_a = NS_AB1_T();
    tacc[0] += t_a - t_b;                 // parked at the barrier
    // (waves 4 - 7, the second-dispatched half, lose every issue arbitration against their SIMD partners and need 1.5x their time
    // for the first half-step; the partners then wait at the barrier.  Raised priority for that half-step: -1.7 % per launch)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const char* const Qs = smem + tb * 16384;
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
      if (qt == 1) {
        // tile t + 1 (requested at the end of step t - 1) and everything older: landed, visible to all behind the next barrier.  The youngest
        // requests before this point are half a step old (the first half's tail).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < nsteps) store_rc(tb_next, cur ^ 1, q0 + QT);
      }
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
      if (qt == 1 && t + 2 < nsteps) {
        if (first) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's O rows of tile t + 1 are read: the slot takes tile t + 2's)
        dma_tile(tb_prev, q0 + 2 * QT);
        load_rc(q0 + 2 * QT);
      }
    };
    half(I0_{});
    __builtin_amdgcn_sched_barrier(0);
    if (wave >= 4) __builtin_amdgcn_s_setprio(0);
    const unsigned long long t_m = NS_AB1_T();
    half(I1_{});
    const unsigned long long t_e = NS_AB1_T();
    tacc[1] += t_m - t_a;                 // first half
    tacc[2] += t_e - t_m;                 // second half
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile 0's pieces (requested at the top of the prologue)
  store_rc(2, 0, q_first);
  if (first) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the O slot is read: it takes tile 1's rows)
  dma_tile(0, q_first + QT);       // (behind the prologue's last barrier: buffer 0 held V rows)
  load_rc(q_first + QT);
  // (the step index of the two peeled iterations is laundered through an SGPR: as compile-time constants their global
  // addresses are loop-invariant 64-bit pairs that hipcc hoists out of the sweep loop, spills, and reloads one by one)
  int t_first = 0, t_last = nsteps;
  asm volatile("" : "+s"(t_first));
  asm volatile("" : "+s"(t_last));
  step(F_{}, T_{}, t_first, 2);
  int tb = 0;
  for (int t = 1; t < nsteps; ++t) {
    step(T_{}, T_{}, t, tb);
    tb = tb == 2 ? 0 : tb + 1;
  }
  step(T_{}, F_{}, t_last, tb);

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
  // the next sweep's rows: in flight while dK / dV leave.  (Unconditional -- past the last sweep the row index clamps to Lk - 1 and the
  // values are dropped: requested under `if (!last)`, the OLD chunks stay live through the whole step loop for the other path, 32 registers)
  ns_attn_bwd1_request_kv(p, K, V, kb0 + KB, tid, kv, vv);
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
  // LDS: three Q / dO tile pairs, two dS images, two sets of row constants, one slot of O rows.  Everything is multi-buffered so that a step needs
  // ONE barrier: step t computes its scores from tile buffer t & 1 into dS image t & 1 while the same waves form the dQ
  // product of step t - 1 from image (t - 1) & 1 -- two independent instruction streams in one basic block (the small
  // dQ MFMAs and their transposed reads fill the exponentials' issue slots), and the tile of step t + 1 is stored meanwhile.
  extern __shared__ __attribute__((aligned(16))) char smem[];      // LDS_TOTAL
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int nsteps = (p.Lq + QT - 1) / QT;
  float* const scr = scratch + (long long)bh * nsteps * (QT * D);   // this (batch, head)'s part: [step][wave][mt][lane] float4
  unsigned long long tacc[3] = {0ull, 0ull, 0ull};
  const unsigned long long t_start = NS_AB1_T();
  (void)t_start;
  uint4 kv[4], vv[4];
  ns_attn_bwd1_request_kv(p, (const half_t*)p.K + (long long)b * p.Lk * p.ldk + h * D, (const half_t*)p.V + (long long)b * p.Lk * p.ldv + h * D, 0,
                          threadIdx.x, kv, vv);
  for (int kb0 = 0; kb0 < p.Lk; kb0 += KB) {
    const bool first = kb0 == 0, last = kb0 + KB >= p.Lk;
    sweep(p, smem, scr, kb0, b, h, nsteps, first, last, tacc, kv, vv);
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
  static ns_dev_once attr_once;
  if (!ns_dyn_lds_once(attr_once, {(const void*)attn_bwd1_kernel}, LDS_TOTAL, "ns_attn_bwd (one pass)")) return NS_ERR_HIP;
  hipLaunchKernelGGL(attn_bwd1_kernel, dim3(d->B * d->H), dim3(NT), LDS_TOTAL, st, *d, (float*)workspace);
  NS_CHECK_LAUNCH("ns_attn_bwd (one pass)");
  return NS_OK;
}
