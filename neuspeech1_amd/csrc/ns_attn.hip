// Fused (flash-style) multi-head attention for head_dim 64, fp16 in / fp32 softmax.
// Replaces HF:modeling_whisper.py:215-238 (eager_attention_forward: QK^T, softmax,
// PV) and its autograd backward for encoder self-attention (S=1500, no mask),
// decoder causal self-attention and decoder->encoder cross-attention.
//
// Orientation: every product is formed TRANSPOSED so that the softmax row (one
// query) lives on ONE lane pair (l, l^32) of v_mfma_f32_32x32x16_f16's C/D layout:
//   S^T[key][q] = K * Q^T      A = K rows from LDS,   B = Q rows in registers
//   O^T[d][q]  += V^T * P^T    A = V^T from a transposed LDS image, B = the S^T
//                               accumulator itself (converted to fp16 in place:
//                               no LDS round trip, no cross-lane traffic)
// Row max / sum / rescale are therefore per-lane scalars.
//
// q is expected PRE-SCALED (head_dim^-0.5 is folded into the q projection, as
// HF does at modeling_whisper.py:309), so no scale is applied here.
//
// Layout: Q/K/V/O/dO are token-major matrices; row (b*L + i), head h at column
// h*64; row strides are arguments so the fused (q|k|v) projection buffer is
// consumed in place.
#include "ns_common.h"

namespace {

constexpr int D = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_TH = 8.f;

// ONE LDS image per 64x64 tile serves the row reads (ds_read_b128, lane = row) AND the transposed reads
// (ds_read_b64_tr_b16): 8-row x 32-column subtiles of 512 B, chunk XOR (row>>2)&3 (cdna guide §5.5 T10 image (a)
// cut down to 64 columns).  Both read kinds are bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return 1024 * (row >> 3) + 512 * (chunk >> 2) + 64 * (row & 7) + 16 * ((chunk & 3) ^ ((row >> 2) & 3));
}

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_s4;
// Transposed fragment: A[row = cb + (lane&31)][k] of v_mfma_f32_32x32x16 where the tile is stored [k][col] and the
// OTHER operand is a 32x32 accumulator converted in place: element j of lane half h is tile row
//   rbase + 8*(j>>2) + 4*h + (j&3)       (cdna guide §3 "accumulator as operand")
// = two 4-row transposed reads at rows rbase+4h and rbase+8+4h.
__device__ __forceinline__ half8 tr_frag8(const char* tile, int rbase, int cb, int lane) {
  const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
  const int row = rbase + 4 * (g >> 1) + q;
  const int ch = (cb >> 3) + 2 * (g & 1) + (p >> 1);
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + lds_off(row, ch) + 8 * (p & 1)));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + lds_off(row + 8, ch) + 8 * (p & 1)));
  const short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8, r);
}

// ---- staging helpers (256 threads, tile = 64 rows x 64 halfs) ------------------
// row-major image: thread handles 2 x 16 B.  Lane -> (row, chunk) inside a group of 16 lanes: lanes 0-3 row 2k chunks 0-3, lanes 4-7 row 2k+1
// chunks 0-3, lanes 8-11 row 2k chunks 4-7, lanes 12-15 row 2k+1 chunks 4-7 -- ds_write_b128 is serviced in groups of 8 consecutive lanes
// over 32 banks, and chunks c and c + 4 of one row sit 512 B apart (the same banks): one row per 8 lanes was a 2-way conflict on every store
// (32 LDS cycles per wave and tile in SQ_LDS_BANK_CONFLICT, all of that counter's total); two half-rows per 8 lanes are conflict-free.
// The global side still covers whole 128-B head rows inside each wave instruction.
__device__ __forceinline__ void rm_lane(int id, int& row, int& chunk) {
  const int l16 = id & 15;
  row = 2 * (id >> 4) + ((l16 >> 2) & 1);
  chunk = (l16 & 3) + 4 * (l16 >> 3);
}
// (wave-uniform base + 32-bit byte offset of the lane, recomputed per tile: 64-bit per-lane address pairs are loop invariants that hipcc keeps live
// through the tile loop -- or spills, each reload behind its own s_waitcnt vmcnt(0); a (batch, head)'s rows stay far below 4 GiB from its base)
__device__ __forceinline__ void load_rm(const half_t* __restrict__ base, long long ld, int row0, int L, uint4& r0, uint4& r1) {
  int row, chunk;
  rm_lane(threadIdx.x, row, chunk);
  const uint32_t o0 = 2u * ((uint32_t)min(row0 + row, L - 1) * (uint32_t)ld + (uint32_t)chunk * 8u);
  const uint32_t o1 = 2u * ((uint32_t)min(row0 + 32 + row, L - 1) * (uint32_t)ld + (uint32_t)chunk * 8u);
  r0 = *(const uint4*)((const char*)base + o0);
  r1 = *(const uint4*)((const char*)base + o1);
}
__device__ __forceinline__ void store_rm(char* lds, const uint4& r0, const uint4& r1) {
  int row, chunk;
  rm_lane(threadIdx.x, row, chunk);
  *(uint4*)(lds + lds_off(row, chunk)) = r0;
  *(uint4*)(lds + lds_off(32 + row, chunk)) = r1;
}
__device__ __forceinline__ half8 cvt8(const f32x16& x, int base) {
  half8 h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = (half_t)x[base + j];
  return h;
}
// row index inside a 32x32 C/D tile held by register r of this lane
__device__ __forceinline__ int crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs in linear-id order (x fastest), so the
// query blocks of ONE (batch, head) would land on 8 different L2s and each would fetch that head's K / V (or Q / dO)
// again from the fabric: rocprofv3 FETCH_SIZE showed 1.85 GB per forward launch against 0.3 GB of operands.  The remap
// gives each XCD a contiguous range of work items, x fastest, so a head's blocks share one L2 (bijective for any grid).
__device__ __forceinline__ void xcd_block_ids(int& bx, int& by, int& bz) {
  const int gx = gridDim.x, gy = gridDim.y;
  const int n = gx * gy * gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int xcd = lin & 7, idx = lin >> 3, q = n >> 3, r = n & 7;
  const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bx = w % gx;
  by = (w / gx) % gy;
  bz = w / (gx * gy);
}

// =============================================================== forward
template <bool CAUSAL>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 4) void attn_fwd_kernel(const ns_attn_desc p) {
  // two K / V tile pairs: tile t + 1 travels from global memory straight into its LDS image (buffer_load ... lds, 1 KiB per piece of eight
  // rows, four pieces per wave and tile) while tile t is consumed -- ONE barrier per tile.  (Round 3: one tile pair, staged through 16
  // registers per thread with two barriers per tile.)
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 31, lh = lane >> 5;
  int bx_, h, b;
  xcd_block_ids(bx_, h, b);
  const int q0 = bx_ * 128;
  const half_t* Q = (const half_t*)p.Q + (long long)b * p.Lq * p.ldq + h * D;
  const half_t* K = (const half_t*)p.K + (long long)b * p.Lk * p.ldk + h * D;
  const half_t* V = (const half_t*)p.V + (long long)b * p.Lk * p.ldv + h * D;
  const int qi = q0 + wave * 32 + lr;           // this lane's query
  const int qrow = min(qi, p.Lq - 1);
  const int coff = p.Lk - p.Lq;                   // causal: key j visible iff j <= qi + coff

  int kend = p.Lk;
  if (CAUSAL) kend = min(p.Lk, q0 + 128 + coff);  // keys beyond the block's last query are never visible
  const int ntiles = (max(kend, 0) + 63) / 64;

  // tile transport (inline assembly: behind the builtin hipcc orders every LDS read after the pieces it knows to be in flight, an
  // s_waitcnt vmcnt(0) per tile in the wrong place; see ns_attn_bwd1.hip).  Lane l of a piece holds row 8 piece + ((l >> 2) & 7), chunk
  // 4 (l >> 5) + ((l & 3) ^ swizzle(row)): byte 16 l of the image's 1-KiB unit.  Keys past Lk repeat row Lk - 1 (masked below).
  const ns_u4v rsrc_k = {(uint32_t)(uintptr_t)K, (uint32_t)((uintptr_t)K >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
  const ns_u4v rsrc_v = {(uint32_t)(uintptr_t)V, (uint32_t)((uintptr_t)V >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  // The forward is bound by VALU ISSUE (round 6 counters, profiles/r6_pmc_attn_fwd.log: 87 % of the SIMDs' vector-issue cycles, the matrix
  // pipe 47 % busy), so everything a tile can do WITHOUT a vector instruction leaves the vector stream: the per-lane part of a piece's source
  // offset is computed once (row inside the tile, swizzled chunk), the tile's position rides in the buffer load's SCALAR offset.  Only the last
  // tile of a sequence whose length is not a multiple of 64 clamps rows (keys past Lk repeat row Lk - 1, masked below) and pays the four 32-bit
  // multiplies (quarter rate) + adds per tile that every tile paid before.
  uint32_t voff_k[2], voff_v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = 2 * wave + j;
    const int row = 8 * piece + ((lane >> 2) & 7);
    const int chunk = 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));
    voff_k[j] = 2u * ((uint32_t)row * (uint32_t)p.ldk + chunk * 8);
    voff_v[j] = 2u * ((uint32_t)row * (uint32_t)p.ldv + chunk * 8);
  }
  auto dma_tile = [&](int buf, int k0) __attribute__((always_inline)) {
    if (k0 + 64 <= p.Lk) {
      const uint32_t so_k = 2u * (uint32_t)k0 * (uint32_t)p.ldk, so_v = 2u * (uint32_t)k0 * (uint32_t)p.ldv;     // wave-uniform: scalar multiplies
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t dst = smem_base + (uint32_t)(buf * 16384 + (2 * wave + j) * 1024);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %7 offen lds\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %8 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(dst), "s"(dst + 8192u), "v"(voff_k[j]), "v"(voff_v[j]), "s"(rsrc_k), "s"(rsrc_v), "s"(so_k), "s"(so_v)
                     : "memory");
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = 2 * wave + j;
      const int row = 8 * piece + ((lane >> 2) & 7);
      const int chunk = 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));
      const uint32_t kr = (uint32_t)min(k0 + row, p.Lk - 1);
      const uint32_t dst = smem_base + (uint32_t)(buf * 16384 + piece * 1024);
      // (m0 saved and restored around the transfers: a reserved register cannot be named as a clobber)
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, 0 offen lds\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "s"(dst), "s"(dst + 8192u), "v"(2u * (kr * (uint32_t)p.ldk + chunk * 8)), "v"(2u * (kr * (uint32_t)p.ldv + chunk * 8)), "s"(rsrc_k), "s"(rsrc_v)
                   : "memory");
    }
  };
  if (ntiles > 0) dma_tile(0, 0);

  // Global rows as 128-B segments through LDS, never as per-lane fragment pieces (lane = row: 32 rows x 32 B per load instruction, 16 B per
  // row and store instruction -- the CU retires such pieces at its request rate, well below HBM's): this wave's 32 query rows go through
  // its own 4 KiB of the second (still idle) tile pair, and O leaves the same way.
  half8 qf[4];
  (void)qrow;
  {
    const int l16 = lane & 15;
    const int srow = 2 * (lane >> 4) + ((l16 >> 2) & 1), schunk = (l16 & 3) + 4 * (l16 >> 3);   // + 8 i rows: see rm_lane
    char* const wst = smem + 16384 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qr = min(q0 + wave * 32 + 8 * i + srow, p.Lq - 1);
      *(uint4*)(wst + lds_off(8 * i + srow, schunk)) = *(const uint4*)(Q + (long long)qr * p.ldq + schunk * 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const half8*)(wst + lds_off(lr, 2 * s + lh));
  }

  f32x16 ot[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // K fragment addresses (same reason): lds_off(kt * 32 + lr, 2 s + lh) = [1024 (lr >> 3) + 64 (lr & 7)] + 16 ((2 (s & 1) + lh) ^ ((lr >> 2) & 3))
  // + 4096 kt + 512 (s >> 1): the swizzle term takes TWO values per lane (s even / odd: x and x ^ 32), the rest are immediates of the read --
  // two address registers and one add per tile for the buffer instead of an address computation per read
  const int kf_base = 1024 * (lr >> 3) + 64 * (lr & 7);
  const int kf_x = 16 * (lh ^ ((lr >> 2) & 3));
  const int kf_off[2] = {kf_base + kf_x, kf_base + (kf_x ^ 32)};
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * 64;
    const char* const Ks = smem + (t & 1) * 16384;
    const char* const Vs = Ks + 8192;
    const char* const kf0 = Ks + kf_off[0];
    const char* const kf1 = Ks + kf_off[1];
    // this wave's pieces of tile t are in; behind the barrier everyone's are, and every wave is through tile t - 1 (the other pair is free)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < ntiles) dma_tile((t + 1) & 1, k0 + 64);

    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 a = *(const half8*)(((s & 1) ? kf1 : kf0) + 4096 * kt + 512 * (s >> 1));
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], st[kt], 0, 0, 0);
      }
    }
    // Element-wise work is what bounds this kernel at head_dim 64 (32 scores per lane and tile against 16 MFMAs), so
    // it is kept to max / fma+exp2 / add / cvt per score: masking only on tiles that need it (block-uniform branch),
    // log2(e) folded into one fma, and the accumulator rescale only when the row maximum moved by more than
    // RESCALE_TH (the exponent then stays below 2^12, far inside fp16 / fp32 range).
    const bool need_mask = CAUSAL || (k0 + 64 > p.Lk);
    if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kt * 32 + crow(r, lh);
          bool ok = key < p.Lk;
          if (CAUSAL) ok = ok && (key <= qi + coff);
          st[kt][r] = ok ? st[kt][r] : -INFINITY;
        }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[kt][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const bool moved = mt > m_run + RESCALE_TH;          // also true on the first visible tile (m_run = -inf)
    if (__builtin_amdgcn_ballot_w64(moved)) {
      const float m_new = moved ? mt : m_run;
      const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
    }
    const float mneg = (m_run == -INFINITY) ? 0.f : -m_run * LOG2E;
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(st[kt][r], LOG2E, mneg));
        st[kt][r] = e;
        ls += e;
      }
    l_run += ls;
    // O^T += V^T P^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const half8 pb = cvt8(st[kt], 8 * s2);
        const int sg = kt * 2 + s2;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const half8 a = tr_frag8(Vs, 16 * sg, dt * 32, lane);
          ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb, ot[dt], 0, 0, 0);
        }
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  __syncthreads();      // every wave is through the last tile: the tile buffers are idle
  // (lane-derived staging offsets are re-derived here from a laundered thread index: kept live through the tile loop they cost the
  // registers that hold this kernel at four waves per SIMD)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, l16 = lane_e & 15, lr_e = lane_e & 31, lh_e = lane_e >> 5;
  const int srow = 2 * (lane_e >> 4) + ((l16 >> 2) & 1), schunk = (l16 & 3) + 4 * (l16 >> 3);
  char* const wst = smem + (tid_e >> 6) * 4096;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[dt][4 * g + e] * inv);
      *(half4*)(wst + lds_off(lr_e, 4 * dt + g) + 8 * lh_e) = o;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint4 v = *(const uint4*)(wst + lds_off(8 * i + srow, schunk));
    const int qr = q0 + wave * 32 + 8 * i + srow;
    if (qr < p.Lq) *(uint4*)((half_t*)p.O + ((long long)b * p.Lq + qr) * p.ldo + h * D + schunk * 8) = v;
  }
  if (qi < p.Lq && p.LSE && lh == 0) p.LSE[((long long)b * p.H + h) * p.Lq + qi] = m_run + __logf(l_tot);
}

// =============================================================== backward: dQ (+ delta)
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const ns_attn_desc p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8192];
  char* const Ks = smem;
  char* const Vs = smem + 8192;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
  int bx_, h, b;
  xcd_block_ids(bx_, h, b);
  const int q0 = bx_ * 128;
  const half_t* Q = (const half_t*)p.Q + (long long)b * p.Lq * p.ldq + h * D;
  const half_t* K = (const half_t*)p.K + (long long)b * p.Lk * p.ldk + h * D;
  const half_t* V = (const half_t*)p.V + (long long)b * p.Lk * p.ldv + h * D;
  const half_t* O = (const half_t*)p.O + (long long)b * p.Lq * p.ldo + h * D;
  const half_t* dO = (const half_t*)p.dO + (long long)b * p.Lq * p.lddo + h * D;
  const int qi = q0 + wave * 32 + lr;
  const int qrow = min(qi, p.Lq - 1);
  const int coff = p.Lk - p.Lq;

  half8 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *(const half8*)(Q + (long long)qrow * p.ldq + 16 * s + 8 * lh);
#ifdef NS_ATTN_LOG2_ON_OPERAND
    // log2(e) rides on the query fragment (used for S only; dQ = dS K reads K): S' = S log2(e) comes out of the MFMA and the
    // probability is exp2(S' - lse log2(e)) with no multiply per score (32 of the ~190 VALU issue slots per tile and lane)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[s][j] = (half_t)((float)qf[s][j] * LOG2E);
#endif
    dof[s] = *(const half8*)(dO + (long long)qrow * p.lddo + 16 * s + 8 * lh);
    const half8 of = *(const half8*)(O + (long long)qrow * p.ldo + 16 * s + 8 * lh);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)of[j];
  }
  const float delta = dl + __shfl_xor(dl, 32, 64);
  const long long stat = ((long long)b * p.H + h) * p.Lq + qrow;
  const float lse = p.LSE[stat];
  if (qi < p.Lq && lh == 0) p.Delta[stat] = delta;

  f32x16 dqt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[t][r] = 0.f;

  int kend = p.Lk;
  if (CAUSAL) kend = min(p.Lk, q0 + 128 + coff);
  const int ntiles = (max(kend, 0) + 63) / 64;

  uint4 kr0, kr1, vr0, vr1;
  if (ntiles > 0) { load_rm(K, p.ldk, 0, p.Lk, kr0, kr1); load_rm(V, p.ldv, 0, p.Lk, vr0, vr1); }
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * 64;
    __syncthreads();
    store_rm(Ks, kr0, kr1); store_rm(Vs, vr0, vr1);
    __syncthreads();
    if (t + 1 < ntiles) { load_rm(K, p.ldk, k0 + 64, p.Lk, kr0, kr1); load_rm(V, p.ldv, k0 + 64, p.Lk, vr0, vr1); }

    // the row constants ride in as the MFMA's initial accumulator: st = S - lse, dp = dP - delta (query on the lane)
    f32x16 st[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
#ifdef NS_ATTN_LOG2_ON_OPERAND
      for (int r = 0; r < 16; ++r) { st[kt][r] = -lse * LOG2E; dp[kt][r] = -delta; }
#else
      for (int r = 0; r < 16; ++r) { st[kt][r] = -lse; dp[kt][r] = -delta; }
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 ak = *(const half8*)(Ks + lds_off(kt * 32 + lr, 2 * s + lh));
        const half8 av = *(const half8*)(Vs + lds_off(kt * 32 + lr, 2 * s + lh));
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ak, qf[s], st[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, dof[s], dp[kt], 0, 0, 0);
      }
    }
    if (CAUSAL || k0 + 64 > p.Lk) {      // block-uniform: only tiles that hold invisible keys pay for the mask
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kt * 32 + crow(r, lh);
          bool ok = key < p.Lk;
          if (CAUSAL) ok = ok && (key <= qi + coff);
          st[kt][r] = ok ? st[kt][r] : -INFINITY;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
#ifdef NS_ATTN_LOG2_ON_OPERAND
      for (int r = 0; r < 16; ++r) st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r]) * dp[kt][r];   // dS^T
#else
      for (int r = 0; r < 16; ++r) st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r] * LOG2E) * dp[kt][r];   // dS^T
#endif
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const half8 dsb = cvt8(st[kt], 8 * s2);
        const int sg = kt * 2 + s2;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const half8 a = tr_frag8(Ks, 16 * sg, dt * 32, lane);
          dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, dsb, dqt[dt], 0, 0, 0);
        }
      }
  }
  if (qi < p.Lq) {
    half_t* dQ = (half_t*)p.dQ + ((long long)b * p.Lq + qi) * p.lddq + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)dqt[dt][4 * g + e];
        *(half4*)(dQ + dt * 32 + 8 * g + 4 * lh) = o;
      }
  }
}

// =============================================================== backward: dK, dV
// One wave owns 32 keys (on the lanes); the block sweeps 64-query tiles.
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const ns_attn_desc p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8192 + 512];
  char* const Qs = smem;
  char* const dOs = smem + 8192;
  float* const lse_s = (float*)(smem + 16384);
  float* const del_s = lse_s + 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
  int bx_, h, b;
  xcd_block_ids(bx_, h, b);
  const int kb0 = bx_ * 128;
  const half_t* Q = (const half_t*)p.Q + (long long)b * p.Lq * p.ldq + h * D;
  const half_t* K = (const half_t*)p.K + (long long)b * p.Lk * p.ldk + h * D;
  const half_t* V = (const half_t*)p.V + (long long)b * p.Lk * p.ldv + h * D;
  const half_t* dO = (const half_t*)p.dO + (long long)b * p.Lq * p.lddo + h * D;
  const float* LSE = p.LSE + ((long long)b * p.H + h) * p.Lq;
  const float* Delta = p.Delta + ((long long)b * p.H + h) * p.Lq;
  const int key = kb0 + wave * 32 + lr;
  const int krow = min(key, p.Lk - 1);
  const int coff = p.Lk - p.Lq;

  half8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = *(const half8*)(K + (long long)krow * p.ldk + 16 * s + 8 * lh);
    vf[s] = *(const half8*)(V + (long long)krow * p.ldv + 16 * s + 8 * lh);
#ifdef NS_ATTN_LOG2_ON_OPERAND
    // log2(e) rides on the key fragment (used for S only; dK = dS^T Q reads Q from LDS), see attn_bwd_dq_kernel
#pragma unroll
    for (int j = 0; j < 8; ++j) kf[s][j] = (half_t)((float)kf[s][j] * LOG2E);
#endif
  }
  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[t][r] = 0.f; dvt[t][r] = 0.f; }

  int qstart = 0;
  if (CAUSAL) qstart = max(0, kb0 - coff) / 64 * 64;  // queries before the block's first key see none of its keys
  const int ntiles = (p.Lq - qstart + 63) / 64;

  uint4 qr0, qr1, dr0, dr1;
  float lse_r = 0.f, del_r = 0.f;
  auto load_all = [&](int q0) {
    load_rm(Q, p.ldq, q0, p.Lq, qr0, qr1); load_rm(dO, p.lddo, q0, p.Lq, dr0, dr1);
    if (threadIdx.x < 64) { const int qq = min(q0 + (int)threadIdx.x, p.Lq - 1); lse_r = LSE[qq]; del_r = Delta[qq]; }
  };
  if (ntiles > 0) load_all(qstart);
  for (int t = 0; t < ntiles; ++t) {
    const int q0 = qstart + t * 64;
    __syncthreads();
    store_rm(Qs, qr0, qr1); store_rm(dOs, dr0, dr1);
    // -lse / -delta of the tile's 64 queries; queries past Lq get -inf so their probabilities vanish without a mask
    if (threadIdx.x < 64) {
      const bool qok = q0 + (int)threadIdx.x < p.Lq;
#ifdef NS_ATTN_LOG2_ON_OPERAND
      lse_s[threadIdx.x] = qok ? -lse_r * LOG2E : -INFINITY;
#else
      lse_s[threadIdx.x] = qok ? -lse_r : -INFINITY;
#endif
      del_s[threadIdx.x] = -del_r;
    }
    __syncthreads();
    if (t + 1 < ntiles) load_all(q0 + 64);

    // the row constants ride in as the initial accumulators (query = accumulator row): st = S - lse, dp = dP - delta
    f32x16 st[2], dp[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 nl = *(const float4*)(lse_s + qt * 32 + 8 * g + 4 * lh);
        const float4 nd = *(const float4*)(del_s + qt * 32 + 8 * g + 4 * lh);
        st[qt][4 * g + 0] = nl.x; st[qt][4 * g + 1] = nl.y; st[qt][4 * g + 2] = nl.z; st[qt][4 * g + 3] = nl.w;
        dp[qt][4 * g + 0] = nd.x; dp[qt][4 * g + 1] = nd.y; dp[qt][4 * g + 2] = nd.z; dp[qt][4 * g + 3] = nd.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 aq = *(const half8*)(Qs + lds_off(qt * 32 + lr, 2 * s + lh));
        const half8 ad = *(const half8*)(dOs + lds_off(qt * 32 + lr, 2 * s + lh));
        st[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, kf[s], st[qt], 0, 0, 0);   // S[q][key] - lse[q]
        dp[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ad, vf[s], dp[qt], 0, 0, 0);   // dP[q][key] - delta[q]
      }
    }
    if (CAUSAL) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qi = q0 + qt * 32 + crow(r, lh);
          if (key > qi + coff) st[qt][r] = -INFINITY;
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef NS_ATTN_LOG2_ON_OPERAND
        const float pv = __builtin_amdgcn_exp2f(st[qt][r]);
#else
        const float pv = __builtin_amdgcn_exp2f(st[qt][r] * LOG2E);
#endif
        st[qt][r] = pv;                 // P
        dp[qt][r] = pv * dp[qt][r];     // dS
      }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const half8 pb = cvt8(st[qt], 8 * s2);
        const half8 dsb = cvt8(dp[qt], 8 * s2);
        const int sg = qt * 2 + s2;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const half8 a1 = tr_frag8(dOs, 16 * sg, dt * 32, lane);
          const half8 a2 = tr_frag8(Qs, 16 * sg, dt * 32, lane);
          dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pb, dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, dsb, dkt[dt], 0, 0, 0);
        }
      }
  }
  if (key < p.Lk) {
    half_t* dK = (half_t*)p.dK + ((long long)b * p.Lk + key) * p.lddk + h * D;
    half_t* dV = (half_t*)p.dV + ((long long)b * p.Lk + key) * p.lddv + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 ok_, ov_;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ok_[e] = (half_t)dkt[dt][4 * g + e]; ov_[e] = (half_t)dvt[dt][4 * g + e]; }
        *(half4*)(dK + dt * 32 + 8 * g + 4 * lh) = ok_;
        *(half4*)(dV + dt * 32 + 8 * g + 4 * lh) = ov_;
      }
  }
}

// =============================================================== backward, few queries (<= 64), no mask, ONE pass: the decoder's cross-attention
// (Lq = label length, Lk = 1500).  Such a launch is bound by streaming K / V in and dK / dV out (392 MB per layer at B = 64 against 17 GFLOP); the
// two-pass form reads K and V a second time for dQ (59 us of 157 per layer).  Here a workgroup owns KPW blocks of 128 keys (one wave = 32 keys,
// fragments in registers), holds the (batch, head)'s single 64-query tile of Q and dO in LDS, and per key block
//   * forms S / dP with the query on the accumulator rows (as attn_bwd_dkv_kernel) for dK^T, dV^T, stored at once;
//   * forms S^T / dP^T again with the operands swapped (key on the accumulator rows, query on the lane: 16 more MFMAs, nothing next to the streaming),
//     rounds dS^T to fp16 and writes it to a [query][key] LDS image -- lane half h's accumulator rows 0..7 / 8..15 are exactly the 8 + 8 keys that the
//     accumulator-as-operand order of tr_frag8 pairs with one k-step, so each is ONE 16-byte store and a plain row read gives the MFMA's B operand;
//   * after one barrier, wave w adds K^T dS^T of all 128 keys into ITS 32 x 32 piece of dQ^T (d half w & 1, query half w >> 1; K^T through the
//     transposed read of a [key][d] LDS image written from the key fragments).
// dQ partial sums of the key-block groups leave as fp32 slabs [group][b][h][q][d] (6 x 5.8 MB at B = 64) and attn_fewq_dq_reduce_kernel adds them
// in a fixed order and rounds to fp16: bitwise reproducible.  delta = rowsum(dO o O) is formed here too (the dQ pass used to).
constexpr int FEWQ_KPW = NS_FEWQ_KPW;       // key blocks of 128 per workgroup (ns_common.h)
__device__ __forceinline__ int dst_off(int q, int chunk) { return q * 256 + ((chunk ^ (q & 15)) << 4); }   // [64 q][128 keys] fp16, conflict-free row reads

__global__ __launch_bounds__(256, 2) void attn_bwd_fewq_kernel(const ns_attn_desc p, float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8192 + 512 + 3 * 16384];
  char* const Qs = smem;
  char* const dOs = smem + 8192;
  float* const lse_s = (float*)(smem + 16384);
  float* const del_s = lse_s + 64;
  char* const Kimg = smem + 16384 + 512;     // [128 keys][64 d] of the current block (row reads: fragments; transposed reads: K^T for dQ)
  char* const Vimg = Kimg + 16384;           // [128 keys][64 d]; each wave's 32 rows double as its dK / dV store staging
  char* const dST = Vimg + 16384;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
  int bx_, h, b;
  xcd_block_ids(bx_, h, b);
  const half_t* Q = (const half_t*)p.Q + (long long)b * p.Lq * p.ldq + h * D;
  const half_t* K = (const half_t*)p.K + (long long)b * p.Lk * p.ldk + h * D;
  const half_t* V = (const half_t*)p.V + (long long)b * p.Lk * p.ldv + h * D;
  const half_t* O = (const half_t*)p.O + (long long)b * p.Lq * p.ldo + h * D;
  const half_t* dO = (const half_t*)p.dO + (long long)b * p.Lq * p.lddo + h * D;

  {   // the one query tile: Q, dO -> LDS; -lse, -delta (queries past Lq: -inf, their probabilities vanish)
    uint4 qr0, qr1, dr0, dr1;
    load_rm(Q, p.ldq, 0, p.Lq, qr0, qr1); load_rm(dO, p.lddo, 0, p.Lq, dr0, dr1);
    store_rm(Qs, qr0, qr1); store_rm(dOs, dr0, dr1);
    // delta[q] = sum_d dO[q][d] O[q][d]: 4 threads per query, 16 columns each
    const int q = threadIdx.x >> 2, part = threadIdx.x & 3, qq = min(q, p.Lq - 1);
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const half8 a = *(const half8*)(dO + (long long)qq * p.lddo + part * 16 + c * 8);
      const half8 o = *(const half8*)(O + (long long)qq * p.ldo + part * 16 + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)a[j] * (float)o[j];
    }
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    if (part == 0) {
      const long long stat = ((long long)b * p.H + h) * p.Lq + qq;
      const bool qok = q < p.Lq;
      lse_s[q] = qok ? -p.LSE[stat] : -INFINITY;
      del_s[q] = -dl;
      if (qok && bx_ == 0) p.Delta[stat] = dl;
    }
  }
  __syncthreads();

  const int dt_w = wave & 1, qt_w = wave >> 1;     // this wave's piece of dQ^T
  f32x16 dqa;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqa[r] = 0.f;

  // Global side of K / V / dK / dV: a wave's 32 key rows as four instructions of 8 rows x 128 B (lanes 0-3 | 8-11 one row's two halves,
  // 4-7 | 12-15 the next row's: whole 128-B lines per instruction, and conflict-free 16-byte LDS stores -- see rm_lane).  Fragment loads
  // straight into registers (lane = key row, 32 rows x 32 B per instruction) are bound by the CU's request rate, not by HBM: 3.0 TB/s.
  const int l16 = lane & 15;
  const int srow = 2 * (lane >> 4) + ((l16 >> 2) & 1), schunk = (l16 & 3) + 4 * (l16 >> 3);   // + 8 i rows for instruction i
  // the next key block, requested one block ahead (two workgroups per CU do not cover an HBM fetch otherwise).  Named registers, not an array:
  // hipcc keeps an array that a lambda fills inside this runtime loop in scratch memory
  uint4 kn0, kn1, kn2, kn3, vn0, vn1, vn2, vn3;
#define NS_FEWQ_LD(i, kr, vr)                                                              \
  do {                                                                                     \
    const int krow_ = min(kb_ + wave * 32 + 8 * (i) + srow, p.Lk - 1);                     \
    kr = *(const uint4*)(K + (long long)krow_ * p.ldk + schunk * 8);                       \
    vr = *(const uint4*)(V + (long long)krow_ * p.ldv + schunk * 8);                       \
  } while (0)
#define NS_FEWQ_LOAD_KV(kb)                                                                \
  do {                                                                                     \
    const int kb_ = (kb);                                                                  \
    NS_FEWQ_LD(0, kn0, vn0); NS_FEWQ_LD(1, kn1, vn1); NS_FEWQ_LD(2, kn2, vn2); NS_FEWQ_LD(3, kn3, vn3); \
  } while (0)
#define NS_FEWQ_ST(i, kr, vr)                                                              \
  do {                                                                                     \
    *(uint4*)(Kimg + lds_off(wave * 32 + 8 * (i) + srow, schunk)) = kr;                    \
    *(uint4*)(Vimg + lds_off(wave * 32 + 8 * (i) + srow, schunk)) = vr;                    \
  } while (0)
  NS_FEWQ_LOAD_KV(bx_ * FEWQ_KPW * 128);
  for (int kbi = 0; kbi < FEWQ_KPW; ++kbi) {
    const int kb0 = (bx_ * FEWQ_KPW + kbi) * 128;
    if (kb0 >= p.Lk) break;       // block-uniform
    const int key = kb0 + wave * 32 + lr;
    NS_FEWQ_ST(0, kn0, vn0); NS_FEWQ_ST(1, kn1, vn1); NS_FEWQ_ST(2, kn2, vn2); NS_FEWQ_ST(3, kn3, vn3);
    if (kbi + 1 < FEWQ_KPW && kb0 + 128 < p.Lk) NS_FEWQ_LOAD_KV(kb0 + 128);
    half8 kf[4], vf[4];      // this wave's rows only: written and read by the same wave, in order
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *(const half8*)(Kimg + lds_off(wave * 32 + lr, 2 * s + lh));
      vf[s] = *(const half8*)(Vimg + lds_off(wave * 32 + lr, 2 * s + lh));
    }
    // ---- dK^T, dV^T (query on the accumulator rows; the row constants ride in as the initial accumulators)
    {
      f32x16 dkt[2], dvt[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[t][r] = 0.f; dvt[t][r] = 0.f; }
      f32x16 st[2], dp[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 nl = *(const float4*)(lse_s + qt * 32 + 8 * g + 4 * lh);
          const float4 nd = *(const float4*)(del_s + qt * 32 + 8 * g + 4 * lh);
          st[qt][4 * g + 0] = nl.x; st[qt][4 * g + 1] = nl.y; st[qt][4 * g + 2] = nl.z; st[qt][4 * g + 3] = nl.w;
          dp[qt][4 * g + 0] = nd.x; dp[qt][4 * g + 1] = nd.y; dp[qt][4 * g + 2] = nd.z; dp[qt][4 * g + 3] = nd.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const half8 aq = *(const half8*)(Qs + lds_off(qt * 32 + lr, 2 * s + lh));
          const half8 ad = *(const half8*)(dOs + lds_off(qt * 32 + lr, 2 * s + lh));
          st[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, kf[s], st[qt], 0, 0, 0);   // S[q][key] - lse[q]
          dp[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ad, vf[s], dp[qt], 0, 0, 0);   // dP[q][key] - delta[q]
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(st[qt][r] * LOG2E);
          st[qt][r] = pv;                 // P
          dp[qt][r] = pv * dp[qt][r];     // dS
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half8 pb = cvt8(st[qt], 8 * s2);
          const half8 dsb = cvt8(dp[qt], 8 * s2);
          const int sg = qt * 2 + s2;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const half8 a1 = tr_frag8(dOs, 16 * sg, dt * 32, lane);
            const half8 a2 = tr_frag8(Qs, 16 * sg, dt * 32, lane);
            dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pb, dvt[dt], 0, 0, 0);
            dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, dsb, dkt[dt], 0, 0, 0);
          }
        }
      // dK, dV: a lane holds 64 d-values of ITS key; through the wave's own 32 rows of the V image (its V fragments are in registers) they
      // leave as 128-B row segments
#ifdef NS_FEWQ_DIRECT_STORE
      if (key < p.Lk) {
        half_t* dK = (half_t*)p.dK + ((long long)b * p.Lk + key) * p.lddk + h * D;
        half_t* dV = (half_t*)p.dV + ((long long)b * p.Lk + key) * p.lddv + h * D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4 ok_, ov_;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ok_[e] = (half_t)dkt[dt][4 * g + e]; ov_[e] = (half_t)dvt[dt][4 * g + e]; }
            *(half4*)(dK + dt * 32 + 8 * g + 4 * lh) = ok_;
            *(half4*)(dV + dt * 32 + 8 * g + 4 * lh) = ov_;
          }
      }
#else
      (void)key;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(which ? dvt[dt][4 * g + e] : dkt[dt][4 * g + e]);
            *(half4*)(Vimg + lds_off(wave * 32 + lr, 4 * dt + g) + 8 * lh) = o;
          }
        half_t* const dst = which ? (half_t*)p.dV + h * D : (half_t*)p.dK + h * D;
        const long long ldd = which ? p.lddv : p.lddk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint4 v = *(const uint4*)(Vimg + lds_off(wave * 32 + 8 * i + srow, schunk));
          const int krow = kb0 + wave * 32 + 8 * i + srow;
          if (krow < p.Lk) *(uint4*)(dst + ((long long)b * p.Lk + krow) * ldd + schunk * 8) = v;
        }
      }
#endif
    }
    // ---- dS^T (key on the accumulator rows, query on the lane) -> [query][key] image; the [key][d] image is the staged K block itself
    {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float nl = lse_s[qt * 32 + lr], nd = del_s[qt * 32 + lr];
        f32x16 sT, dpT;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] = nl; dpT[r] = nd; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const half8 bq = *(const half8*)(Qs + lds_off(qt * 32 + lr, 2 * s + lh));
          const half8 bd = *(const half8*)(dOs + lds_off(qt * 32 + lr, 2 * s + lh));
          sT = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], bq, sT, 0, 0, 0);    // S^T[key][q] - lse[q]
          dpT = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s], bd, dpT, 0, 0, 0);  // dP^T[key][q] - delta[q]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool kok = kb0 + wave * 32 + crow(r, lh) < p.Lk;      // rows past Lk hold copies of the last key: no contribution to dQ
          sT[r] = kok ? __builtin_amdgcn_exp2f(sT[r] * LOG2E) * dpT[r] : 0.f;
        }
        // accumulator rows 0..7 of lane half h = keys {4h..4h+3, 8+4h..8+4h+3} of the first 16, rows 8..15 the same of the second 16: the k order
        // tr_frag8 gives the OTHER operand -- stored as positions [8h, 8h+8) of each 16-key group, a row read returns them as the B operand
        *(half8*)(dST + dst_off(qt * 32 + lr, 4 * wave + lh)) = cvt8(sT, 0);
        *(half8*)(dST + dst_off(qt * 32 + lr, 4 * wave + 2 + lh)) = cvt8(sT, 8);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const half8 a = tr_frag8(Kimg, 16 * ks, dt_w * 32, lane);                                   // K^T[d][16 keys]
      const half8 bfr = *(const half8*)(dST + dst_off(qt_w * 32 + lr, 2 * ks + lh));            // dS^T[16 keys][q]
      dqa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr, dqa, 0, 0, 0);
    }
    __syncthreads();      // the images are rewritten by the next key block
  }
  // dQ^T[d = 32 dt + crow(r, lh)][q = 32 qt + lr] of this group of key blocks
  const int q = qt_w * 32 + lr;
  if (q < p.Lq) {
    float* const dst = slabs + ((((long long)bx_ * p.B + b) * p.H + h) * p.Lq + q) * D + dt_w * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(float4*)(dst + 8 * g + 4 * lh) = make_float4(dqa[4 * g], dqa[4 * g + 1], dqa[4 * g + 2], dqa[4 * g + 3]);
  }
}

// dQ[b][q][h*64 + d] = fp16(sum over the key-block groups, in order): one thread per 4 columns
__global__ __launch_bounds__(256) void attn_fewq_dq_reduce_kernel(const ns_attn_desc p, const float* __restrict__ slabs, int nslab) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // float4 index inside a slab: ((b H + h) Lq + q) 16 + c
  const long long per = (long long)p.B * p.H * p.Lq * (D / 4);
  if (i >= per) return;
  float4 a = *(const float4*)(slabs + 4 * i);
  for (int s = 1; s < nslab; ++s) {
    const float4 v = *(const float4*)(slabs + 4 * (i + s * per));
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int c = (int)(i % (D / 4));
  const long long row = i / (D / 4);            // (b H + h) Lq + q
  const int q = (int)(row % p.Lq);
  const long long bh = row / p.Lq;
  const int h = (int)(bh % p.H);
  const long long b = bh / p.H;
  const half4 o = {(half_t)a.x, (half_t)a.y, (half_t)a.z, (half_t)a.w};
  *(half4*)((half_t*)p.dQ + (b * p.Lq + q) * p.lddq + h * D + 4 * c) = o;
}

int check_desc(const ns_attn_desc* d, bool bwd) {
  NS_CHECK_ARG(d, "ns_attn: null descriptor");
  NS_CHECK_ARG(d->head_dim == 64, "ns_attn: head_dim=%d unsupported (only 64)", d->head_dim);
  NS_CHECK_ARG(d->B > 0 && d->H > 0 && d->Lq > 0 && d->Lk > 0, "ns_attn: bad shape");
  NS_CHECK_ARG(d->Q && d->K && d->V && d->O, "ns_attn: null tensor");
  NS_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 8 == 0, "ns_attn: row strides must be multiples of 8");
  if (bwd) {
    NS_CHECK_ARG(d->dO && d->dQ && d->dK && d->dV && d->LSE && d->Delta, "ns_attn_bwd: null tensor");
    NS_CHECK_ARG(d->lddo % 8 == 0 && d->lddq % 8 == 0 && d->lddk % 8 == 0 && d->lddv % 8 == 0, "ns_attn_bwd: row strides must be multiples of 8");
  }
  return NS_OK;
}

}  // namespace

extern "C" int ns_attn_fwd(const ns_attn_desc* d, void* stream) {
  int rc = check_desc(d, false);
  if (rc) return rc;
  dim3 grid((d->Lq + 127) / 128, d->H, d->B);
  hipStream_t st = (hipStream_t)stream;
  if (d->causal) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, st, *d);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, st, *d);
  NS_CHECK_LAUNCH("ns_attn_fwd");
  return NS_OK;
}

int ns_attn_bwd1_launch(const ns_attn_desc* d, void* workspace, size_t workspace_bytes, hipStream_t st);

extern "C" int ns_attn_bwd(const ns_attn_desc* d, void* stream) {
  int rc = check_desc(d, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t ws_need = d->workspace ? ns_attn_bwd_workspace_bytes(d->B, d->H, d->Lq, d->Lk, d->causal) : 0;
  if (ws_need > 0 && d->Lq <= 64) {      // few queries (the decoder's cross-attention): one pass, dQ through fp32 slabs
    NS_CHECK_ARG(d->workspace_bytes >= ws_need, "ns_attn_bwd (few queries): workspace of %zu bytes needed, %zu given", ws_need, d->workspace_bytes);
    const int nslab = (d->Lk + 128 * FEWQ_KPW - 1) / (128 * FEWQ_KPW);
    hipLaunchKernelGGL(attn_bwd_fewq_kernel, dim3(nslab, d->H, d->B), dim3(256), 0, st, *d, (float*)d->workspace);
    const long long per = (long long)d->B * d->H * d->Lq * (D / 4);
    hipLaunchKernelGGL(attn_fewq_dq_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, *d, (const float*)d->workspace, nslab);
    NS_CHECK_LAUNCH("ns_attn_bwd (few queries)");
    return NS_OK;
  }
  if (ws_need > 0)
    return ns_attn_bwd1_launch(d, d->workspace, d->workspace_bytes, st);   // one pass: csrc/ns_attn_bwd1.hip
  dim3 gq((d->Lq + 127) / 128, d->H, d->B), gk((d->Lk + 127) / 128, d->H, d->B);
  if (d->causal) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, gq, dim3(256), 0, st, *d);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, gk, dim3(256), 0, st, *d);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, gq, dim3(256), 0, st, *d);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, gk, dim3(256), 0, st, *d);
  }
  NS_CHECK_LAUNCH("ns_attn_bwd");
  return NS_OK;
}
