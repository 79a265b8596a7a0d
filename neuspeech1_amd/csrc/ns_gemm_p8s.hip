// ns_gemm NT, large-M form, PERSISTENT phase-interleaved kernel: the tile arithmetic of ns_gemm_p8.hip (256 x 256 tile, 8 waves,
// four 16-MFMA phases per K tile, LDS-DMA ring of two 64 KiB buffers -- see that file for the ring, its swizzle and the
// second (LoRA) product), with one workgroup per CU walking several output tiles so that one tile's epilogue carries the next
// tile's prologue.  Same products in the same order as ns_gemm_p8_kernel: outputs are bit-identical (tests/test_kernels_gpu.py).
//   * The epilogue stages the tile in two 128-row halves (the accumulator halves ah = 0 / 1 of every wave) through the 66 KiB
//     that start at ring buffer 1, so ring buffer 0 is free from the moment the main loop ends: the four regions of the NEXT
//     tile's first K tile (8 DMA pieces per wave) and its 256 bias values (a ninth piece, into a 1-KiB LDS slot) are requested
//     between the conversions of the first half and land under the stores.  The remaining two regions (K tile 1 -> buffer 1)
//     follow the last store, and with them the second product's first round: the tile's 256 x 32 slices of A2 / B2 as LDS-DMA pieces
//     into ring buffer 1's two idle regions (stage_k2; fragment loads straight into registers cost 18 % of a K = 512 tile).
//   * A tile's set-up, its prologue latency (6.1 k of 41 k cycles in the one-tile form at K = 512) and the workgroup launch are paid
//     once per CU; what the half-wise epilogue costs more than the whole-tile one (two more barriers, 4.8 k + 5.9 k against 3.0 k +
//     5.9 k cycles) takes most of it back: 36-37 k cycles per K = 512 tile against 41 k, 0-8 % per launch on the step's shapes.
//   * Per-tile address arithmetic is re-derived from laundered copies of tid and of the row-map divisors: hipcc otherwise hoists
//     the tile loop's invariants (lane offsets, division reciprocals) above the loop and spills them -- a scratch reload in the
//     epilogue is a VMEM operation that waits for every store before it.  Epilogue accesses are scalar base + 32-bit offset.
//   * No position-row add (the conv2 epilogue stays on ns_gemm_p8_kernel), destinations below 4 GiB (ns_gemm_p8s_ok).
// Tiles of one XCD's range are dealt round-robin to that XCD's workgroups (same L2 sharing as the one-tile-per-workgroup map).
// ns_gemm dispatches here at >= 700 tiles (NS_P8S_MIN_TILES in ns_gemm.hip: about three or more per CU); below that the one-tile form, whose tiles the hardware deals dynamically,
// is as fast or faster (same-box A/B of whole steps: 32.7 ms against 33.3 with the one-tile form everywhere and 33.6 with this one everywhere).
#include "ns_gemm_epi.h"
#include <mutex>

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTH = 512;
constexpr int REGION = 128 * 128;            // 16 KiB
constexpr int BUF = 4 * REGION;              // 64 KiB
constexpr int RA0 = 0, RA1 = 1, RB0 = 2, RB1 = 3;
constexpr int LDH = BN * 2 + 16;             // epilogue: bytes per staged fp16 row
constexpr int HALF_BYTES = 128 * LDH;        // 66 KiB: one staged half (128 rows)
constexpr int SIDE_OFF = BUF + HALF_BYTES + 2048;   // 132 KiB
constexpr int SIDE_BYTES = 32 * 256 * 2;     // side product: the 32 x 256 slice of side_B of this tile's columns
constexpr int BIAS_OFF = SIDE_OFF + SIDE_BYTES;     // two 2-KiB slots (tile parity): this tile's 256 bias values, fetched by LDS-DMA one tile ahead
constexpr int LDS_BYTES = BIAS_OFF + 2 * 2048;      // 152 KiB

typedef __attribute__((address_space(3))) void lds_void;

#define NS_P8_BARRIER()                         \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

template <int V> struct p8_int_c { static constexpr int value = V; };

// epilogue addressing in 32 bits: element offset of a row (< 2^30 elements, checked by ns_gemm_p8s_fits), and accesses as
// scalar base + 32-bit byte offset (one address register per access instead of two)
__device__ __forceinline__ uint32_t rm_off32(const ns_rowmap& m, int row) {
  if (m.seg_rows > 0) {
    const int s = row / m.seg_rows;
    const int w = row - s * m.seg_rows;
    return (uint32_t)s * (uint32_t)m.seg_stride + (uint32_t)w * (uint32_t)m.ld;
  }
  return (uint32_t)row * (uint32_t)m.ld;
}
// Element offsets of the tile rows row0 + c (0 <= c < 256, c a compile-time constant) of a row map WITHOUT a multiply or a division
// per row: rm_off32 per row and output was two quarter-rate v_mul_lo_u32 for a plain map and a 35-instruction integer division for a
// segmented one (the conv stem's halo layouts) -- a quarter of a GELU epilogue row on the conv launches (round 4, ISA).  `base` is the
// offset of row0; c * ld is a scalar product; a tile crosses at most four segment ends (seg_rows >= 64), each adds `jump`.
struct rm_walk {
  uint32_t base, w0, jump;
  int seg_rows, ld;
  __device__ __forceinline__ void init(const ns_rowmap& m, int row0) {
    ld = m.ld;
    seg_rows = m.seg_rows;
    if (m.seg_rows > 0) {
      const int s = row0 / m.seg_rows;
      w0 = (uint32_t)(row0 - s * m.seg_rows);
      base = (uint32_t)s * (uint32_t)m.seg_stride + w0 * (uint32_t)m.ld;
      jump = (uint32_t)m.seg_stride - (uint32_t)m.seg_rows * (uint32_t)m.ld;
    } else {
      w0 = 0u;
      base = (uint32_t)row0 * (uint32_t)m.ld;
      jump = 0u;
    }
  }
  __device__ __forceinline__ uint32_t at(int c) const {
    uint32_t o = base + (uint32_t)(c * ld);
    if (seg_rows > 0) {
      const uint32_t w = w0 + (uint32_t)c;
#pragma unroll
      for (int k = 1; k <= 4; ++k) o += w >= (uint32_t)(k * seg_rows) ? jump : 0u;
    }
    return o;
  }
};
template <class T>
__device__ __forceinline__ T ld32(const void* base, uint32_t byte_off) { return *(const T*)((const char*)base + byte_off); }
template <class T>
__device__ __forceinline__ void st32(void* base, uint32_t byte_off, const T& v) { *(T*)((char*)base + byte_off) = v; }

#ifdef NS_P8_STAMPS
#define NS_STAMP(i) do { if (stamps && tid == 0) { stamps[(xstart + ti) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define NS_STAMP(i) do { } while (0)
#endif

// WP ("wave-private" epilogue, round 6; PLAIN launches without a side product): every wave reads back and stores exactly the 64 x 64 block of
// the staged half that IT wrote (8 rows x 128 B per wave instruction instead of 2 rows x 512 B), so staging and storing need no workgroup
// barrier between them -- a wave's own LDS traffic is ordered by s_waitcnt lgkmcnt(0) -- and none between the halves.  In-kernel stamps of the
// barrier form at K = 512 (profiles/r6_probe_p8s_stamps.log): stage 3.7 k + finish 3.6 k + stage 1.5 k + finish 3.2 k = 12 k of a 37 k-cycle tile,
// the two finish phases paced by the rate the CU's store path accepts 1-KiB store instructions (~19 B per cycle: tools/probe/store_burst.hip),
// every wave in the same phase.  Without the barriers the waves drift apart and one wave's conversions run under another's stores: measured
// (tools/probe/wp_epilogue_ab.py, profiles/r6_probe_wp_epilogue.log, outputs bitwise equal) -4 % on the stacked cross K|V launch, the conv GELU
// launches and the K = 512 dgrads, -1 % on q|k|v, level on the K >= 1536 dgrads and on the gelu'-multiply form (which therefore keeps the
// barrier form: no instantiation is built for it).  What is left of the finish phases is the store path itself: a CU's 128 KiB take 6.5 k
// cycles when every CU stores at once and 3.7 k when half of them do (tools/probe/store_burst.hip), and inside this kernel the storing CU
// shares its L2 with every other CU's operand stream.
template <bool DROP, int KIND, bool K2LDS, bool WP>
__global__ __launch_bounds__(NTH) void ns_gemm_p8s_kernel(const ns_gemm_desc p_in) {
  static_assert(!WP || KIND == NS_EPI_PLAIN, "the fp32-residual (HBM-bound) and gelu'-multiply (no gain measured) epilogues keep the barrier form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef NS_P8_STAMPS
  ns_gemm_desc p = p_in;
  unsigned long long* const stamps = (p_in.flags & (1 << 27)) ? (unsigned long long*)p_in.C32 : nullptr;
  if (stamps) p.C32 = nullptr;
#else
  const ns_gemm_desc& p = p_in;
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  // XCD x owns the contiguous tile range [xstart, xstart + xcount); its workgroups (blockIdx % 8 == x) take every nx-th tile of it
  const int nx = gridDim.x >> 3;
  int xstart, xcount, ti;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    xstart = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xcount = q + (xcd < r ? 1 : 0);
    ti = bid >> 3;
  }
  if (ti >= xcount) return;
  // Start-up stagger (round 6, the gelu'-multiply launches of >= 8 tiles per CU only).  One workgroup per CU walks equal tiles from a common start, so
  // the first epilogues -- here 128 KiB of pre-activation loads + 128 KiB of stores per tile -- hit the fabric together (later ones drift apart by
  // the main loops' own spread): a CU's 128 KiB
  // of stores take 6.5 k cycles when every CU stores at once and 3.7 k when half of them do (tools/probe/store_burst.hip).  Workgroup group
  // g = (id >> 3) & 3 starts g * splits / 4 cycles late (`splits`, unused by this form otherwise, carries the spread: the launcher).  Measured
  // (tools/probe/p8s_stagger_ab.py, profiles/r6_probe_p8s_stagger.log; spread 0 / 8 k / 16 k / 24 k / 40 k cycles): fc2 dgrad x gelu' 280.7 /
  // 275.1 / 268.5 / 266.5 / 266.1 us; the other epilogue kinds do not gain (q|k|v level, GELU +2 %, fp32 residual +3..6 %), so only this kind
  // does it.  Timing only: results are bit-identical.
  if (KIND == NS_EPI_DGELU && p.splits > 1) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dl = (unsigned long long)((blockIdx.x >> 3) & 3) * (unsigned long long)p.splits / 4ull;
    while (__builtin_amdgcn_s_memtime() - t0 < dl) __builtin_amdgcn_s_sleep(8);
  }

  const int nsteps = (p.K + BK - 1) / BK;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x80000000u, 0x00020000);
  // bias of a tile's 256 columns -> LDS, one 4-B element per lane: waves 0..3 fetch the four 64-column groups, waves 4..7 a second
  // (unread) copy so that every wave's vmcnt sees the same number of pieces.  Columns past N and a null bias read as zeros (range check).
  const __amdgpu_buffer_rsrc_t rsrc_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.A), 0, p.bias ? 4u * (uint32_t)p.N : 0u, 0x00020000);
  auto stage_bias = [&](int n0, int slot, int lane) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_bias, (lds_void*)(smem + BIAS_OFF + slot * 2048 + wave * 256), 4,
                                             4u * (uint32_t)(n0 + 64 * (wave & 3) + lane), 0, 0, 0);
  };

  // DMA sources of the current / next tile.  Wave w fills region rows [16w, 16w+16) as two 1-KiB pieces (8 rows x 128 B).
  uint32_t a_src[2][2], b_src[2][2];
  int my_chunk[2];
  int src_n0 = 0;
  // Second (LoRA) product, first 32-deep round (K2LDS instantiations: K2 > 0 and one column group per tile, a2_ngroup % 256 == 0): the tile's
  // 256 x 32 slices of A2 and B2 (64-B rows) come in as 1-KiB LDS-DMA pieces, like every other operand, into the two regions of ring buffer 1
  // that are idle until phase 1 / 2 of the main loop refill them (RA1 <- A2 rows, RB1 <- B2 rows).  Fragment loads straight into registers
  // (16 rows x 64 B per instruction) are bound by the CU's request rate: 1 536 pieces per tile took 7.5 k cycles, 18 % of a K = 512 tile for
  // 6 % of its FLOPs.  16-B chunk c of image row r sits at chunk c ^ ((r >> 2) & 2): conflict-free ds_read_b128 fragment reads at a 64-B row
  // stride.  (A template parameter, not a run-time flag: with the code merely present, hipcc's allocation of the OTHER launches' epilogues
  // shifted and they lost 2-5 % in a same-box A/B of the two builds.)
  const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(K2LDS ? p.A2 : p.A), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b2 = __builtin_amdgcn_make_buffer_rsrc((void*)(K2LDS ? p.B2 : p.B), 0, 0x80000000u, 0x00020000);
  auto stage_k2 = [&](int wtile, int lane) __attribute__((always_inline)) {
    ns_rowmap am2 = p.am2;
    int tn_div = tiles_n, a2g = p.a2_ngroup;
    asm volatile("" : "+s"(am2.seg_rows), "+s"(tn_div), "+s"(a2g));
    const int tm_ = wtile / tn_div;
    const int m0 = tm_ * BM, n0 = (wtile - tm_ * tn_div) * BN;
    const uint32_t goff = a2g > 0 ? (uint32_t)((n0 / a2g) * p.K2) : 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * (2 * wave + j) + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 2);
      const bool ok = c * 8 < p.K2;
      const uint32_t ao = 2u * (rm_off32(am2, min(m0 + row, p.M - 1)) + goff + (uint32_t)c * 8u);
      const uint32_t bo = 2u * ((uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldb2 + (uint32_t)c * 8u);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a2, (lds_void*)(smem + BUF + RA1 * REGION + (2 * wave + j) * 1024), 16, ok ? ao : 0x80000000u, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b2, (lds_void*)(smem + BUF + RB1 * REGION + (2 * wave + j) * 1024), 16, ok ? bo : 0x80000000u, 0, 0, 0);
    }
  };
  auto set_src = [&](int wtile, int lane) __attribute__((always_inline)) {
    // divisors laundered: their reciprocals are formed here, per tile, not above the tile loop (where they would be spilled)
    ns_rowmap am = p.am;
    int tn_div = tiles_n;
    asm volatile("" : "+s"(am.seg_rows), "+s"(tn_div));
    const int tm_ = wtile / tn_div;
    const int m0 = tm_ * BM, n0 = (wtile - tm_ * tn_div) * BN;
    src_n0 = n0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = 16 * wave + 8 * j + (lane >> 3);
      my_chunk[j] = (lane & 7) ^ ((rr >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int arow = ((rr >> 6) << 7) + h * 64 + (rr & 63);
        const int brow = ((rr >> 5) << 6) + h * 32 + (rr & 31);
        a_src[h][j] = rm_off32(am, min(m0 + arow, p.M - 1)) + (uint32_t)(my_chunk[j] * 8);      // element offsets < 2^30 (ns_gemm_p8_fits)
        b_src[h][j] = (uint32_t)min(n0 + brow, p.N - 1) * (uint32_t)p.bm.ld + (uint32_t)(my_chunk[j] * 8);
      }
    }
  };
  auto stage1 = [&](int region, int j) __attribute__((always_inline)) {     // one piece of K tile 0 -> ring buffer 0
    const int h = region & 1;
    const bool isb = region >= 2;
    const uint32_t off = isb ? b_src[h][j] : a_src[h][j];
    const bool ok = my_chunk[j] * 8 < p.K;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rsrc_b : rsrc_a, (lds_void*)(smem + region * REGION + wave * 2048 + j * 1024), 16,
                                             ok ? 2u * off : 0x80000000u, 0, 0, 0);
  };
  auto stage = [&](int tt, int region, int buf) __attribute__((always_inline)) {
    const int k0 = tt * BK;
    const int klen = p.K - k0;              // <= 0 past the end (padding tiles): fetched as zeros
    const int h = region & 1;
    const bool isb = region >= 2;
    char* const dst = smem + buf * BUF + region * REGION + wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t off = isb ? b_src[h][j] : a_src[h][j];
      const bool ok = my_chunk[j] * 8 < klen;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rsrc_b : rsrc_a, (lds_void*)(dst + j * 1024), 16, ok ? 2u * off : 0x80000000u,
                                               2 * k0, 0, 0);
    }
  };

  const int fsw = ((lg ^ (l15 >> 1)) & 7) << 4;
  const int a_base = (wm * 64 + l15) * 128 + fsw;
  const int b_base = (wn * 32 + l15) * 128 + fsw;

  f32x4 acc[2][4][2][2];
  half8 af[4][2], bq[2][2][2];
  auto read_a = [&](int buf, int ah) __attribute__((always_inline)) {
    const char* st = smem + buf * BUF + (ah ? RA1 : RA0) * REGION;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const half8*)(st + ((a_base ^ (ks << 6)) + mt * 2048));
  };
  auto read_b = [&](int buf, int bh) __attribute__((always_inline)) {
    const char* st = smem + buf * BUF + (bh ? RB1 : RB0) * REGION;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bq[bh][nt][ks] = *(const half8*)(st + ((b_base ^ (ks << 6)) + nt * 2048));
  };
  auto mma = [&](int ah, int bh) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[ah][mt][bh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bh][nt][ks], af[mt][ks], acc[ah][mt][bh][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define NS_P8S_RUN(AH, BH)                                  \
  do {                                                      \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        \
    NS_P8_BARRIER();                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);                      \
    mma(AH, BH);                                            \
    __builtin_amdgcn_sched_barrier(0);                      \
    NS_P8_BARRIER();                                        \
  } while (0)
  auto tile = [&](int t, int b) __attribute__((always_inline)) {
    read_b(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(b, 0);
    stage(t + 1, RB1, b ^ 1);
    NS_P8S_RUN(0, 0);
    read_b(b, 1);
    stage(t + 1, RA1, b ^ 1);
    NS_P8S_RUN(0, 1);
    read_a(b, 1);
    stage(t + 2, RA0, b);
    NS_P8S_RUN(1, 1);
    stage(t + 2, RB0, b);
    NS_P8S_RUN(1, 0);
  };

  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  char* const hs = smem + BUF;          // staged half tile
  char* const sbs = smem + SIDE_OFF;    // side_B slice

  // ---- first tile: full prologue
  set_src(xstart + ti, lane);
  stage_bias(src_n0, 0, lane);
  stage(0, RA0, 0); stage(0, RB0, 0); stage(0, RB1, 0); stage(0, RA1, 0);
  if (K2LDS) stage_k2(xstart + ti, lane);
  stage(1, RA0, 1); stage(1, RB0, 1);
  int par = 0;    // bias slot of the current tile

  for (;;) {
    // lane-derived values are re-derived per tile from a laundered copy of tid: hipcc would otherwise hoist this tile loop's
    // invariant address arithmetic (second product, epilogue) above the loop and carry it through the main loop in registers
    int tid_k = tid;
    asm volatile("" : "+v"(tid_k));
    const int lane = tid_k & 63, l15 = lane & 15, lg = lane >> 4;
    NS_STAMP(0);
    const int wgid = xstart + ti;
    int tn_div = tiles_n, a2g = p.a2_ngroup;
    ns_rowmap am2 = p.am2;
    asm volatile("" : "+s"(tn_div), "+s"(a2g), "+s"(am2.seg_rows));
    const int tm = wgid / tn_div, tn = wgid - tm * tn_div;
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- second product (LoRA up-projection, K2 = 16 .. 96): the first 32-deep round from the LDS images (stage_k2), further rounds
    // (the stacked q|k|v bottleneck of a dgrad, 3r) and column groups narrower than a tile as fragments straight from global memory
    if (p.K2 > 0) {
      half8 a2f[2][4], b2f[2][2];
      uint32_t a2o[2][4], b2o[2][2];     // byte offsets from A2 / B2 (< 2^32: ns_gemm_p8s_ok)
      const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
      const int goff = a2g > 0 ? ((n0 + wn * 64) / a2g) * p.K2 : 0;
#pragma unroll
      for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row = min(m0 + wm * 128 + ah * 64 + mt * 16 + l15, p.M - 1);
          a2o[ah][mt] = 2u * (rm_off32(am2, row) + (uint32_t)goff);
        }
#pragma unroll
      for (int bh = 0; bh < 2; ++bh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int col = min(n0 + wn * 64 + bh * 32 + nt * 16 + l15, p.N - 1);
          b2o[bh][nt] = 2u * ((uint32_t)col * (uint32_t)p.ldb2);
        }
      if (K2LDS) {
        // round 0 from the LDS images: the four K2 pieces of this wave are older than the four pieces of K tile 1 behind them
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        NS_P8_BARRIER();
        const char* const a2img = smem + BUF + RA1 * REGION;
        const char* const b2img = smem + BUF + RB1 * REGION;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int row = wm * 128 + ah * 64 + mt * 16 + l15;
            a2f[ah][mt] = *(const half8*)(a2img + row * 64 + ((lg ^ ((row >> 2) & 2)) << 4));
          }
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int row = wn * 64 + bh * 32 + nt * 16 + l15;
            b2f[bh][nt] = *(const half8*)(b2img + row * 64 + ((lg ^ ((row >> 2) & 2)) << 4));
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[ah][mt][bh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2f[bh][nt], a2f[ah][mt], acc[ah][mt][bh][nt], 0, 0, 0);
      }
      for (int k0 = K2LDS ? 32 : 0; k0 < p.K2; k0 += 32) {
        const bool ok = k0 + 8 * lg < p.K2;
        const int ko = ok ? k0 + 8 * lg : 0;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a2f[ah][mt]) : "v"(a2o[ah][mt] + 2u * (uint32_t)ko), "s"(p.A2) : "memory");
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b2f[bh][nt]) : "v"(b2o[bh][nt] + 2u * (uint32_t)ko), "s"(p.B2) : "memory");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(a2f[0][0]), "+v"(a2f[0][1]), "+v"(a2f[0][2]), "+v"(a2f[0][3]), "+v"(a2f[1][0]), "+v"(a2f[1][1]),
                       "+v"(a2f[1][2]), "+v"(a2f[1][3]), "+v"(b2f[0][0]), "+v"(b2f[0][1]), "+v"(b2f[1][0]), "+v"(b2f[1][1])
                     :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!ok) {
#pragma unroll
          for (int ah = 0; ah < 2; ++ah)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a2f[ah][mt] = hz;
        }
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[ah][mt][bh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2f[bh][nt], a2f[ah][mt], acc[ah][mt][bh][nt], 0, 0, 0);
      }
      if (DROP) {
        // LoRA-dropout mask on the (A2, B2) product, before the main product accumulates on top
        const uint32_t drop_thr = ns_drop_thr8(p.drop_p);
        const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const uint32_t row = (uint32_t)(m0 + wm * 128 + a * 64 + i * 16 + l15);
                const uint32_t col = (uint32_t)(n0 + wn * 64 + bb * 32 + j * 16 + 4 * lg);
                const uint32_t w = ns_drop_word(dseed, row, col >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][i][bb][j][e] = ns_keep(w, e, drop_thr) ? acc[a][i][bb][j][e] : 0.f;
              }
      }
    }
    // only RA0 / RB0 of K tile 0 must have landed: phase 1 reads nothing else, and the in-loop vmcnt(8) of phases 1 and 2
    // retires RB1 / RA1 one phase before they are read
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    NS_P8_BARRIER();
    NS_STAMP(1);
    if (wm == 1) NS_P8_BARRIER();     // group 1 runs one barrier interval behind group 0
    for (int t = 0; t < nsteps; t += 2) {
      tile(t, 0);
      tile(t + 1, 1);
    }
    if (wm == 0) NS_P8_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-chunk DMA must not land on the staged tile / the next prologue
    NS_P8_BARRIER();
    NS_STAMP(2);

    // ---- epilogue in two 128-row halves (half hh = accumulator half ah of every wave; staged row rl -> tile row (rl>>6)*128 + hh*64 + (rl&63))
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, l15e = lane_e & 15, lge = lane_e >> 4;
    // barrier form: thread = (16-B column chunk ecg of the 512-B row, row er0 + 16 i); WP: the wave's own block, lane = (row (lane >> 3) + 8 i of
    // its 64 rows, chunk wn * 8 + (lane & 7))
    const int ecg = WP ? wn * 8 + (lane_e & 7) : (tid_e & 31), er0 = WP ? wm * 64 + (lane_e >> 3) : (tid_e >> 5);
    ns_rowmap c16m = p.c16m, g16m = p.g16m, p16m = p.p16m, h32m = p.h32m;
    int pos_rows = p.pos_rows;
    asm volatile("" : "+s"(c16m.seg_rows), "+s"(g16m.seg_rows), "+s"(p16m.seg_rows), "+s"(h32m.seg_rows), "+s"(pos_rows));
    const int tnext = ti + nx;
    const bool has_next = tnext < xcount;
    const int ecol = n0 + ecg * 8;
    const bool ecolok = ecol + 8 <= p.N;
    const int ecolc = min(ecol, p.N - 8);
    // residual epilogue: a thread owns columns {4 ecg .. +3} and {128 + 4 ecg .. +3} of its rows instead of 8 consecutive ones: the fp32
    // accesses of a wave instruction are then 16 B per lane at 16-B pitch (512 contiguous bytes per row), not every other 16-B chunk
    const int rcolA = n0 + ecg * 4, rcolB = rcolA + 128;
    const bool rokA = rcolA + 4 <= p.N, rokB = rcolB + 4 <= p.N;
    const int rcolAc = min(rcolA, p.N - 4), rcolBc = min(rcolB, p.N - 4);
    half_t* const C16 = (half_t*)p.C16;
    half_t* const G16 = (half_t*)p.G16;
    const half_t* const P16 = (const half_t*)p.P16;
    const bool do_gelu = p.flags & NS_GEMM_GELU;
    const bool save_grad = p.flags & NS_GEMM_GELU_SAVE_GRAD, mulp = p.flags & NS_GEMM_MUL_P16;
    const bool side = !WP && KIND == NS_EPI_PLAIN && p.side_B != nullptr;
    const uint32_t side_thr = (side && p.side_drop_p > 0.f) ? ns_drop_thr8(p.side_drop_p) : 0u;
    const uint32_t side_dseed = side_thr ? ns_eff_seed(p.side_drop_seed, p.seed_dev) : 0u;
    f32x4 res[KIND == NS_EPI_RES ? 8 : 1][2];     // one half at a time (the half-1 rows are requested when half 0's are consumed)
    half8 pre[KIND == NS_EPI_DGELU ? 16 : 1];
    float4 bz[2][2];
#pragma unroll
    for (int bh = 0; bh < 2; ++bh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bz[bh][nt] = *(const float4*)(smem + BIAS_OFF + par * 2048 + (wn * 64 + bh * 32 + nt * 16 + 4 * lge) * 4);
      }
    // this thread's rows of half hh: staged row er0 + 16 i (er0 < 16) = tile row hh * 64 + er0 + rowc(i), rowc(i) = (i >> 2) * 128 + 16 * (i & 3)
    // a compile-time constant: offsets walk from the first row's (rm_walk) instead of being re-derived row by row
    // (WP: staged row er0 + 8 i, er0 = wm * 64 + (lane >> 3) = tile row wm * 128 + hh * 64 + (lane >> 3) + 8 i)
    auto rowc = [](int i) __attribute__((always_inline)) { return WP ? 8 * i : (i >> 2) * 128 + 16 * (i & 3); };
    auto grow = [&](int hh, int rl) __attribute__((always_inline)) { return m0 + ((rl >> 6) << 7) + hh * 64 + (rl & 63); };
    const int erow0 = WP ? wm * 128 + (lane_e >> 3) : er0;      // tile row (within a half) of this thread's first row
    constexpr int RSTEP = WP ? 8 : 16;                          // staged rows between this thread's consecutive rows
    auto prefetch = [&](int hh) __attribute__((always_inline)) {
      const int row0 = m0 + hh * 64 + erow0;
      rm_walk wk;
      wk.init(KIND == NS_EPI_RES ? h32m : p16m, min(row0, p.M - 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t o = row0 + rowc(i) < p.M ? wk.at(rowc(i)) : 0u;      // rows past M: any valid address (never consumed)
        if (KIND == NS_EPI_RES) {
          res[i][0] = p.R32 ? ld32<f32x4>(p.R32, 4u * (o + (uint32_t)rcolAc)) : f32x4{0.f, 0.f, 0.f, 0.f};
          res[i][1] = p.R32 ? ld32<f32x4>(p.R32, 4u * (o + (uint32_t)rcolBc)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (KIND == NS_EPI_DGELU) pre[hh * 8 + i] = ld32<half8>(P16, 2u * (o + (uint32_t)ecolc));
      }
    };
    // half 0 carries the next tile's first K tile (-> ring buffer 0, dead since the main loop ended) and its bias values: one piece after
    // every second accumulator group, so the requests drain through the address pipe under the conversions instead of in one burst
    auto stage_half = [&](int hh) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            if (KIND != NS_EPI_RES && hh == 0 && has_next && nt == 0) {
              const int pc = mt * 2 + bh;     // 0 .. 7: RA0, RB0, RB1, RA1 (the order phase 1 .. 3 read them), two pieces each
              stage1(pc < 2 ? RA0 : (pc < 4 ? RB0 : (pc < 6 ? RB1 : RA1)), pc & 1);
              if (pc == 7) stage_bias(src_n0, par ^ 1, lane_e);
            }
            const int rl = wm * 64 + mt * 16 + l15e;
            const int cl = wn * 64 + bh * 32 + nt * 16 + 4 * lge;
            const f32x4 a = acc[hh][mt][bh][nt];
            const half4 h = {(half_t)(a[0] * alpha + bz[bh][nt].x), (half_t)(a[1] * alpha + bz[bh][nt].y),
                             (half_t)(a[2] * alpha + bz[bh][nt].z), (half_t)(a[3] * alpha + bz[bh][nt].w)};
            *(half4*)(hs + rl * LDH + cl * 2) = h;
          }
    };
    auto finish_half = [&](int hh) __attribute__((always_inline)) {
      // this thread's 8 staged rows first (one LDS round trip, not eight)
      half8 vst[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == NS_EPI_RES) {
          const half4 va = *(const half4*)(hs + (er0 + 16 * i) * LDH + ecg * 8), vb = *(const half4*)(hs + (er0 + 16 * i) * LDH + 256 + ecg * 8);
          vst[i] = half8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        } else {
          vst[i] = *(const half8*)(hs + (er0 + RSTEP * i) * LDH + ecg * 16);
        }
      }
      // offsets of this thread's first row of the half in every destination, walked row by row below
      const int frow0 = m0 + hh * 64 + erow0;
      rm_walk wc, wg, wh;
      if (C16) wc.init(c16m, min(frow0, p.M - 1));
      if (G16) wg.init(g16m, min(frow0, p.M - 1));
      if (KIND == NS_EPI_RES) wh.init(h32m, min(frow0, p.M - 1));
      const uint32_t side_ha0 = (uint32_t)frow0 * NS_HASH_A;
      const uint32_t side_hb0 = side_dseed ^ (((uint32_t)ecol >> 2) * NS_HASH_B), side_hb1 = side_dseed ^ ((((uint32_t)ecol >> 2) + 1) * NS_HASH_B);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        {
          // the two waves of a SIMD run the same VALU-bound row loop (GELU: ~37 instructions per element pair); at equal priority the
          // older wave wins every issue arbitration, finishes its rows at 11.7 k cycles and leaves the younger one to issue alone at
          // half rate until 17.5 k (in-kernel stamps, fc1 shape).  Alternating the priority row by row keeps both in the loop together.
          if (KIND != NS_EPI_RES && do_gelu) {
            if ((i & 1) == wm) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
          }
          const int rl = er0 + RSTEP * i, row = frow0 + rowc(i);
          if (!(KIND == NS_EPI_RES ? rokA : ecolok) || row >= p.M) continue;
          half8 v = vst[i];
          if (KIND == NS_EPI_DGELU) {
            if (mulp) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * (float)pre[hh * 8 + i][e]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * ns_gelu_grad((float)pre[hh * 8 + i][e]));
            }
          }
          half8 gv = v, cv = v;
          if (do_gelu) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              ns_f2 g_, dg_;
              ns_gelu_both2(ns_f2{(float)v[e], (float)v[e + 1]}, g_, dg_);
              gv[e] = (half_t)g_.x; gv[e + 1] = (half_t)g_.y;
              if (save_grad) { cv[e] = (half_t)dg_.x; cv[e + 1] = (half_t)dg_.y; }
            }
          }
          if (side) {
            // side product (see ns_gemm_desc): the GELU values go back to this thread's own place in the staged half, LoRA-dropout
            // mask applied, for the MFMA pass below.  BEFORE the row's global stores: placed behind them, the masks were ANDed in
            // place onto the registers the stores had just been given, and hipcc protected that write-after-read with
            // s_waitcnt vmcnt(0) -- a full store round trip per row, sixteen per tile of every fc1 launch.
            uint4 w = __builtin_bit_cast(uint4, gv);
            if (side_thr) {
              uint32_t mk[4];
              // ns_drop_word(seed, row, col4) with its two input products walked instead of multiplied (ns_common.h)
              const uint32_t ha = side_ha0 + (uint32_t)rowc(i) * NS_HASH_A;
              ns_keep_masks(ns_hash3_mix(side_hb0 ^ ha), side_thr, mk[0], mk[1]);
              ns_keep_masks(ns_hash3_mix(side_hb1 ^ ha), side_thr, mk[2], mk[3]);
              w.x &= mk[0]; w.y &= mk[1]; w.z &= mk[2]; w.w &= mk[3];
            }
            // an asm store: behind a C++ store to LDS hipcc waits vmcnt(0) for the next tile's LDS-DMA pieces in flight (it models them as LDS
            // stores that might alias) AND for this row's global stores -- a full round trip per row, sixteen per tile of every fc1 launch.
            // The pieces land in ring buffer 0 / the bias slot, never in the staged tile; the MFMA pass below reads it behind lgkmcnt(0) + barrier.
            asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(hs + rl * LDH + ecg * 16)), "v"(ns_u4v{w.x, w.y, w.z, w.w}) : "memory");
          }
          if (KIND == NS_EPI_RES) {
            if (C16) {
              const uint32_t oc = wc.at(rowc(i));
              st32<half4>(C16, 2u * (oc + (uint32_t)rcolA), half4{cv[0], cv[1], cv[2], cv[3]});
              if (rokB) st32<half4>(C16, 2u * (oc + (uint32_t)rcolB), half4{cv[4], cv[5], cv[6], cv[7]});
            }
            if (G16) {
              const uint32_t og = wg.at(rowc(i));
              st32<half4>(G16, 2u * (og + (uint32_t)rcolA), half4{gv[0], gv[1], gv[2], gv[3]});
              if (rokB) st32<half4>(G16, 2u * (og + (uint32_t)rcolB), half4{gv[4], gv[5], gv[6], gv[7]});
            }
          } else {
            if (C16) st32<half8>(C16, 2u * (wc.at(rowc(i)) + (uint32_t)ecol), cv);
            if (G16) st32<half8>(G16, 2u * (wg.at(rowc(i)) + (uint32_t)ecol), gv);
          }
          if (KIND == NS_EPI_RES) {
            f32x4 h0 = res[i][0], h1 = res[i][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] += (float)gv[e]; h1[e] += (float)gv[4 + e]; }
            const uint32_t oh = wh.at(rowc(i));
            st32<f32x4>(p.H32, 4u * (oh + (uint32_t)rcolA), h0);
            if (rokB) st32<f32x4>(p.H32, 4u * (oh + (uint32_t)rcolB), h1);
          }
        }
      }
      if (KIND != NS_EPI_RES && do_gelu) __builtin_amdgcn_s_setprio(0);
      if (side) {
        // side_out[tn][m][j] = sum_n gm[m][n] side_B[j][n0 + n] over this tile's 256 columns: wave w takes staged rows 16 w .. 16 w + 15
        // x 32 adapter rows (two 16-row tiles) x 8 steps of 32 columns; side_B on the MFMA A port (a lane owns 4 consecutive j of a row)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        NS_P8_BARRIER();
        float* const slab = p.side_out + ((long long)tn * p.M) * 32;
        const int rl = 16 * wave + l15e;
        f32x4 su[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ss = 0; ss < 8; ++ss) {
          const half8 gm = *(const half8*)(hs + rl * LDH + (32 * ss + 8 * lge) * 2);
          const half8 s0 = *(const half8*)(sbs + l15e * 512 + (((4 * ss + lge) ^ l15e) << 4));
          const half8 s1 = *(const half8*)(sbs + (16 + l15e) * 512 + (((4 * ss + lge) ^ l15e) << 4));
          su[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s0, gm, su[0], 0, 0, 0);
          su[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1, gm, su[1], 0, 0, 0);
        }
        const int row = grow(hh, rl);
        if (row < p.M) {
          *(f32x4*)(slab + (long long)row * 32 + 4 * lge) = su[0];
          *(f32x4*)(slab + (long long)row * 32 + 16 + 4 * lge) = su[1];
        }
      }
    };

    if (side) {
      // this tile's 32 x 256 slice of side_B, once per tile; 16-B chunk c of row j sits at chunk c ^ (j & 15)
      const half_t* const SB = (const half_t*)p.side_B + n0 + (tid & 31) * 8;
      const uint4 sb0 = *(const uint4*)(SB + (long long)(tid >> 5) * p.side_ldb);
      const uint4 sb1 = *(const uint4*)(SB + (long long)(16 + (tid >> 5)) * p.side_ldb);
      const int j = tid >> 5, c = tid & 31;
      *(uint4*)(sbs + j * 512 + ((c ^ (j & 15)) << 4)) = sb0;
      *(uint4*)(sbs + (16 + j) * 512 + ((c ^ (j & 15)) << 4)) = sb1;
    }
    prefetch(0);
    if (KIND != NS_EPI_RES && has_next) set_src(xstart + tnext, lane_e);
    NS_STAMP(7);
    stage_half(0);
    NS_STAMP(10);
    if (KIND == NS_EPI_RES && has_next) {     // (the residual form has no registers to spare for the next tile's addresses before half 0 is staged)
      set_src(xstart + tnext, lane_e);
      stage(0, RA0, 0); stage(0, RB0, 0); stage(0, RB1, 0); stage(0, RA1, 0);
      stage_bias(src_n0, par ^ 1, lane_e);
    }
    if (KIND != NS_EPI_RES) prefetch(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!WP) NS_P8_BARRIER();       // (WP: the wave reads back only what it wrote itself)
    NS_STAMP(3);
    // half 0's loads are older than the next tile's pieces and half 1's loads: settle them by count
    if (KIND == NS_EPI_RES) {
      if (has_next) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(res[i][0])); asm volatile("" : "+v"(res[i][1])); }
    }
    if (KIND == NS_EPI_DGELU) {
      if (has_next) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(pre[i]));
    }
    finish_half(0);
    NS_STAMP(4);
    if (KIND == NS_EPI_RES) prefetch(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (WP: this wave's reads of half 0 are done before it overwrites its block)
    if (!WP) NS_P8_BARRIER();
    stage_half(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!WP) NS_P8_BARRIER();
    NS_STAMP(5);
    if (KIND == NS_EPI_RES) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(res[i][0])); asm volatile("" : "+v"(res[i][1])); }
    }
    if (KIND == NS_EPI_DGELU) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 8; i < 16; ++i) asm volatile("" : "+v"(pre[i]));
    }
    finish_half(1);
    NS_STAMP(6);
    if (!has_next) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NS_P8_BARRIER();          // every wave is done with the staged half: ring buffer 1 may be refilled
    if (K2LDS) stage_k2(xstart + tnext, lane_e);
    stage(1, RA0, 1); stage(1, RB0, 1);
    ti = tnext;
    par ^= 1;
  }
}

}  // namespace
extern int g_ns_ab_flag;      // A/B flag 4: the barrier form of the epilogue everywhere (tools/probe)
namespace {

template <bool DROP, bool K2LDS>
void launch_kind(const ns_gemm_desc* d, int grid, hipStream_t st) {
  const int kind = d->H32 ? NS_EPI_RES : ((d->flags & (NS_GEMM_DGELU | NS_GEMM_MUL_P16)) ? NS_EPI_DGELU : NS_EPI_PLAIN);
  const bool wp = !(g_ns_ab_flag & 4) && !d->side_B;
  if (kind == NS_EPI_RES) hipLaunchKernelGGL((ns_gemm_p8s_kernel<DROP, NS_EPI_RES, K2LDS, false>), dim3(grid), dim3(NTH), LDS_BYTES, st, *d);
  else if (kind == NS_EPI_DGELU) hipLaunchKernelGGL((ns_gemm_p8s_kernel<DROP, NS_EPI_DGELU, K2LDS, false>), dim3(grid), dim3(NTH), LDS_BYTES, st, *d);
  else {
    if (wp) hipLaunchKernelGGL((ns_gemm_p8s_kernel<DROP, NS_EPI_PLAIN, K2LDS, true>), dim3(grid), dim3(NTH), LDS_BYTES, st, *d);
    else hipLaunchKernelGGL((ns_gemm_p8s_kernel<DROP, NS_EPI_PLAIN, K2LDS, false>), dim3(grid), dim3(NTH), LDS_BYTES, st, *d);
  }
}

template <bool DROP, int KIND, bool K2LDS, bool WP = false>
const void* kfn() { return (const void*)ns_gemm_p8s_kernel<DROP, KIND, K2LDS, WP>; }

}  // namespace

// the epilogue addresses its fp32 / fp16 destinations with 32-bit byte offsets, and has no position-row add
bool ns_gemm_p8s_ok(const ns_gemm_desc* d) {
  const auto extent = [](const ns_rowmap& m, int rows, int n) -> long long {
    const long long last = m.seg_rows > 0 ? (long long)((rows - 1) / m.seg_rows) * m.seg_stride + (long long)((rows - 1) % m.seg_rows) * m.ld
                                          : (long long)(rows - 1) * m.ld;
    return last + n + 16;
  };
  if (d->pos) return false;
  if (d->K2 > 0 && (2 * extent(d->am2, d->M, d->K2 * 3) >= 0xFFFF0000LL || 2LL * d->N * d->ldb2 >= 0xFFFF0000LL)) return false;
  if (d->H32 && 4 * extent(d->h32m, d->M, d->N) >= 0xFFFF0000LL) return false;
  if (d->C16 && 2 * extent(d->c16m, d->M, d->N) >= 0xFFFF0000LL) return false;
  if (d->G16 && 2 * extent(d->g16m, d->M, d->N) >= 0xFFFF0000LL) return false;
  if (d->P16 && 2 * extent(d->p16m, d->M, d->N) >= 0xFFFF0000LL) return false;
  return true;
}

int ns_gemm_p8s_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  static ns_dev_once attr_once;      // kernel attributes, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once,
                       {kfn<false, NS_EPI_PLAIN, false>(), kfn<false, NS_EPI_RES, false>(), kfn<false, NS_EPI_DGELU, false>(),
                        kfn<true, NS_EPI_PLAIN, false>(), kfn<true, NS_EPI_RES, false>(), kfn<true, NS_EPI_DGELU, false>(),
                        kfn<false, NS_EPI_PLAIN, true>(), kfn<false, NS_EPI_RES, true>(), kfn<false, NS_EPI_DGELU, true>(),
                        kfn<true, NS_EPI_PLAIN, true>(), kfn<true, NS_EPI_RES, true>(), kfn<true, NS_EPI_DGELU, true>(),
                        kfn<false, NS_EPI_PLAIN, false, true>(), kfn<true, NS_EPI_PLAIN, false, true>(), kfn<false, NS_EPI_PLAIN, true, true>(),
                        kfn<true, NS_EPI_PLAIN, true, true>()},
                       LDS_BYTES, "ns_gemm (p8s)"))
    return NS_ERR_HIP;
  static std::once_flag cu_once;     // every device of a node is the same part: the CU count is read once
  static int cus_per_xcd = 32;
  std::call_once(cu_once, [&] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8)
      cus_per_xcd = cus / 8;
  });
  // one workgroup per CU (148 KiB of LDS, 512 threads x 256 registers: nothing else fits beside it), 8 XCDs round-robin
  const int per_xcd = std::min(cus_per_xcd, (tiles + 7) / 8);
  const bool k2lds = d->K2 > 0 && (d->a2_ngroup == 0 || d->a2_ngroup % BN == 0);   // one column group per tile: the LDS form of the second product
  ns_gemm_desc dd = *d;
  dd.splits = 1;
  if ((d->flags & (NS_GEMM_DGELU | NS_GEMM_MUL_P16)) && !d->H32 && tiles >= 8 * 8 * cus_per_xcd && !(g_ns_ab_flag & 8))
    dd.splits = 24000;      // the start-up stagger of the gelu'-multiply form (see the kernel); A/B flag 8 = off
  d = &dd;
  if (d->drop_p > 0.f) { if (k2lds) launch_kind<true, true>(d, 8 * per_xcd, st); else launch_kind<true, false>(d, 8 * per_xcd, st); }
  else { if (k2lds) launch_kind<false, true>(d, 8 * per_xcd, st); else launch_kind<false, false>(d, 8 * per_xcd, st); }
  return 0;
}
