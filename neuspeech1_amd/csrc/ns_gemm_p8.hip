// ns_gemm NT, large-M form, phase-interleaved: 256x256 tile, 8 waves (2 x 4), wave tile 128 x 64 computed as four
// 64 x 32 quadrants of v_mfma_f32_16x16x32_f16, one quadrant (16 MFMAs) per phase.
//
// The two wave groups (wm = 0 / 1, one wave of each per SIMD) run one barrier interval apart: while one group issues
// its LDS fragment reads and its LDS-DMA pieces, the other group's MFMA cluster owns the matrix pipe, so operand
// delivery and MFMA overlap instead of adding up (the one-barrier-per-slice kernel ns_gemm_ring256 issues both
// groups' DMA in the same interval: its loads-only and compute-only times add).
//
// LDS (one array): two 64 KiB buffers, each four 16 KiB regions of 128 rows x 128 B (fp16, K = 64):
//   RA0 = rows {wm*128 + 0..63}, RA1 = rows {wm*128 + 64..127} of the A tile; RB0 = rows {wn*64 + 0..31},
//   RB1 = rows {wn*64 + 32..63} of the B tile -- i.e. each region is exactly what ONE phase reads, so a region is dead
//   one phase after it was read and is refilled two phases later.  Per K tile t (buffer b = t & 1):
//     phase 1: read B0, A0   stage RB1 of tile t+1 -> b^1     MFMA A0 x B0
//     phase 2: read B1       stage RA1 of tile t+1 -> b^1     MFMA A0 x B1
//     phase 3: read A1       stage RA0 of tile t+2 -> b       MFMA A1 x B1
//     phase 4: -             stage RB0 of tile t+2 -> b       MFMA A1 x B0
//   Every phase ends its load part with s_waitcnt vmcnt(8): the region staged four phases earlier has landed, it is
//   first read five phases after it was issued, with a barrier of the issuing wave in between (never vmcnt(0) inside
//   the loop).  K tails and tiles past the end are fetched from a block of zeros, so the loop has no branch.
// Swizzle: 16-B chunk c of region row r sits at chunk c ^ ((r >> 1) & 7) (applied to the DMA SOURCE address and to
// the ds_read_b128 address): conflict-free for the 16x16x32 fragment pattern.
// Epilogue: accumulators are kept TRANSPOSED (the B rows feed the MFMA's A port), so a lane owns 4 consecutive
// columns of one output row: the fp32 tile goes to LDS with ds_write_b128 (row stride 260 floats) in two 128-row
// halves, and the shared vector epilogue streams it out.
#include "ns_gemm_epi.h"
#include <mutex>

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTH = 512;
constexpr int REGION = 128 * 128;            // 16 KiB
constexpr int BUF = 4 * REGION;              // 64 KiB
constexpr int RA0 = 0, RA1 = 1, RB0 = 2, RB1 = 3;
constexpr int LDH = BN * 2 + 16;             // epilogue: bytes per staged fp16 row
constexpr int EPI_BYTES = BM * LDH;          // 132 KiB
constexpr int SIDE_BYTES = 32 * 256 * 2;     // side product: the 32 x 256 slice of side_B of this tile's columns (16 KiB, behind the staged tile)
constexpr int LDS_BYTES = (EPI_BYTES > 2 * BUF ? EPI_BYTES : 2 * BUF) + SIDE_BYTES;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__device__ __forceinline__ void glds16(const half_t* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}

__device__ __attribute__((aligned(16))) const uint32_t ns_p8_zero_chunk[4] = {0, 0, 0, 0};

#define NS_P8_BARRIER()                         \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

template <int V> struct p8_int_c { static constexpr int value = V; };


#ifdef NS_P8_STAMPS
#define NS_STAMP(i) do { if (stamps && tid == 0) { stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define NS_STAMP(i) do { } while (0)
#endif

template <bool DROP>
__global__ __launch_bounds__(NTH) void ns_gemm_p8_kernel(const ns_gemm_desc p_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef NS_P8_STAMPS
  ns_gemm_desc p = p_in;
  unsigned long long* const stamps = (p_in.flags & (1 << 27)) ? (unsigned long long*)p_in.C32 : nullptr;
  if (stamps) { p.C32 = nullptr; if (tid == 0) { stamps[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime(); stamps[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg(6164); } }
  NS_STAMP(0);
#else
  const ns_gemm_desc& p = p_in;
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int wgid;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // acc[ah][mt][bh][nt]: rows m = wm*128 + ah*64 + mt*16 + l15, columns n = wn*64 + bh*32 + nt*16 + 4*lg + reg
  f32x4 acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA sources.  Wave w fills region rows [16w, 16w+16) as two 1-KiB pieces (8 rows x 128 B, lane-linear).
  uint32_t a_src[2][2], b_src[2][2];
  int my_chunk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rr = 16 * wave + 8 * j + (lane >> 3);
    my_chunk[j] = (lane & 7) ^ ((rr >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int arow = ((rr >> 6) << 7) + h * 64 + (rr & 63);
      const int brow = ((rr >> 5) << 6) + h * 32 + (rr & 31);
      a_src[h][j] = (uint32_t)(ns_rm_off64(p.am, min(m0 + arow, p.M - 1)) + my_chunk[j] * 8);
      b_src[h][j] = (uint32_t)((long long)min(n0 + brow, p.N - 1) * p.bm.ld + my_chunk[j] * 8);
    }
  }

  // The (A2, B2) product (the LoRA up-projection: K2 = r or 3r, i.e. 16 .. 96) does NOT go through the LDS ring: a
  // 64-deep ring tile for a 32-deep operand cost a whole K tile of operand delivery (+30 % on the q|k|v shape, two
  // padded tiles on the dropout variant).  Its fragments are loaded from global memory (L2 / Infinity Cache: u was just
  // written) straight into MFMA operand registers and multiplied while the prologue's LDS-DMA pieces are in flight.
  const int nsteps = (p.K + BK - 1) / BK;

#ifndef NS_P8_GLOBAL_DMA
  // LDS-DMA pieces as buffer loads: the per-lane part of the address is a 32-bit byte offset computed once per tile, the
  // K position rides in the scalar offset, and a lane whose chunk lies past K (K tails, padding tiles) gets an offset
  // beyond num_records, which the hardware range check turns into zeros -- one compare + one select per piece where the
  // global_load form needed a 64-bit add and a 64-bit select against a block of zeros (the main loop is bound by each
  // wave's own instruction stream: DESIGN.md §6).  Operands are < 2 GiB (checked by the launcher).
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x80000000u, 0x00020000);
#endif
  auto stage = [&](int tt, int region, int buf) __attribute__((always_inline)) {
    const int k0 = tt * BK;
    int klen = p.K - k0;                    // <= 0 past the end (padding tiles): fetched as zeros
#ifdef NS_P8_STAMPS
    if (p.flags & (1 << 30)) return;   // diagnostic build: no loads
#endif
    const int h = region & 1;
    const bool isb = region >= 2;
    char* const dst = smem + buf * BUF + region * REGION + wave * 2048;
#ifdef NS_P8_GLOBAL_DMA
    const half_t* const base = (const half_t*)(isb ? p.B : p.A);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t off = isb ? b_src[h][j] : a_src[h][j];
      const bool ok = my_chunk[j] * 8 < klen;
      glds16(ok ? base + (size_t)off + k0 : (const half_t*)ns_p8_zero_chunk, dst + j * 1024);
    }
#else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t off = isb ? b_src[h][j] : a_src[h][j];
      const bool ok = my_chunk[j] * 8 < klen;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rsrc_b : rsrc_a, (lds_void*)(dst + j * 1024), 16, ok ? 2u * off : 0x80000000u,
                                               2 * k0, 0, 0);
    }
#endif
  };

  // fragment addresses: row (16-row tile base + l15), chunk (4*ks + lg) ^ (l15 >> 1)  =>  k-step 1 = address ^ 64
  const int fsw = ((lg ^ (l15 >> 1)) & 7) << 4;
  const int a_base = (wm * 64 + l15) * 128 + fsw;
  const int b_base = (wn * 32 + l15) * 128 + fsw;

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
#ifdef NS_P8_STAMPS
#define NS_P8_MMA(AH, BH) do { if (!(p.flags & (1 << 29))) mma(AH, BH); } while (0)   /* diagnostic build: no MFMA */
#else
#define NS_P8_MMA(AH, BH) mma(AH, BH)
#endif
#ifdef NS_P8_SETPRIO
#define NS_P8_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define NS_P8_PRIO(x) do {} while (0)
#endif
  // load part done -> barrier -> fragments landed -> MFMA cluster -> barrier
#define NS_P8_RUN(AH, BH)                                   \
  do {                                                      \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        \
    NS_P8_BARRIER();                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);                      \
    NS_P8_PRIO(1);                                          \
    NS_P8_MMA(AH, BH);                                      \
    NS_P8_PRIO(0);                                          \
    __builtin_amdgcn_sched_barrier(0);                      \
    NS_P8_BARRIER();                                        \
  } while (0)

  auto tile = [&](int t, int b) __attribute__((always_inline)) {
    read_b(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(b, 0);
    stage(t + 1, RB1, b ^ 1);
    NS_P8_RUN(0, 0);
    read_b(b, 1);
    stage(t + 1, RA1, b ^ 1);
    NS_P8_RUN(0, 1);
    read_a(b, 1);
    stage(t + 2, RA0, b);
    NS_P8_RUN(1, 1);
    stage(t + 2, RB0, b);
    NS_P8_RUN(1, 0);
  };

  // second product, round 0: the fragment loads go out FIRST (12 global_load_dwordx4 per lane), the 12 prologue DMA
  // pieces behind them; vmcnt(12) then says "fragments landed" without waiting for a single DMA piece.  The loads are
  // inline asm (hipcc would wait vmcnt(0) for an ordinary load while LDS-DMA is in flight, draining the prologue).
  half8 a2f[2][4], b2f[2][2];
  const half_t* a2p[2][4];
  const half_t* b2p[2][2];
  const bool k2ok = p.K2 > 0 && 8 * lg < p.K2;
  if (p.K2 > 0) {
    // a 64-column wave strip lies inside ONE column group (a2_ngroup % 64 == 0)
    const int goff = p.a2_ngroup > 0 ? ((n0 + wn * 64) / p.a2_ngroup) * p.K2 : 0;
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = min(m0 + wm * 128 + ah * 64 + mt * 16 + l15, p.M - 1);
        a2p[ah][mt] = (const half_t*)p.A2 + ns_rm_off64(p.am2, row) + goff + (k2ok ? 8 * lg : 0);
      }
#pragma unroll
    for (int bh = 0; bh < 2; ++bh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = min(n0 + wn * 64 + bh * 32 + nt * 16 + l15, p.N - 1);
        b2p[bh][nt] = (const half_t*)p.B2 + (long long)col * p.ldb2 + (k2ok ? 8 * lg : 0);
      }
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a2f[ah][mt]) : "v"(a2p[ah][mt]) : "memory");
#pragma unroll
    for (int bh = 0; bh < 2; ++bh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b2f[bh][nt]) : "v"(b2p[bh][nt]) : "memory");
  }
  // prologue: tile 0 complete + the first two regions of tile 1
  NS_STAMP(1);
  stage(0, RA0, 0); stage(0, RB0, 0); stage(0, RB1, 0); stage(0, RA1, 0);
  stage(1, RA0, 1); stage(1, RB0, 1);
  if (p.K2 > 0) {
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mma2 = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[ah][mt][bh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2f[bh][nt], a2f[ah][mt], acc[ah][mt][bh][nt], 0, 0, 0);
    };
    // round 0: everything issued after the 12 fragment loads is the 12 DMA pieces
    asm volatile("s_waitcnt vmcnt(12)"
                 : "+v"(a2f[0][0]), "+v"(a2f[0][1]), "+v"(a2f[0][2]), "+v"(a2f[0][3]), "+v"(a2f[1][0]), "+v"(a2f[1][1]),
                   "+v"(a2f[1][2]), "+v"(a2f[1][3]), "+v"(b2f[0][0]), "+v"(b2f[0][1]), "+v"(b2f[1][0]), "+v"(b2f[1][1])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!k2ok) {      // lanes whose 8 k-values lie past K2 (K2 = 16: lanes 32..63) contribute zeros
#pragma unroll
      for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a2f[ah][mt] = hz;
    }
    mma2();
    // further rounds (K2 > 32: the stacked q|k|v bottleneck of a dgrad, 3r): loaded behind the DMA pieces, so their
    // wait also covers the prologue (which phase 1 needs anyway)
    for (int k0 = 32; k0 < p.K2; k0 += 32) {
      const bool ok = k0 + 8 * lg < p.K2;
      const int ko = ok ? k0 : 0;
#pragma unroll
      for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a2f[ah][mt]) : "v"(a2p[ah][mt] + ko) : "memory");
#pragma unroll
      for (int bh = 0; bh < 2; ++bh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b2f[bh][nt]) : "v"(b2p[bh][nt] + ko) : "memory");
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
      mma2();
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
  // only RA0 / RB0 of tile 0 (the first four pieces) must have landed: phase 1 reads nothing else, and the in-loop
  // vmcnt(8) of phases 1 and 2 retires RB1 / RA1 one phase before they are read
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  NS_P8_BARRIER();
  NS_STAMP(2);
  if (wm == 1) NS_P8_BARRIER();     // group 1 runs one barrier interval behind group 0
  int t = 0;
  for (; t < nsteps; t += 2) {
    tile(t, 0);
    tile(t + 1, 1);
  }
  if (wm == 0) NS_P8_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-chunk DMA must not land on the staged tile
  NS_P8_BARRIER();

  NS_STAMP(3);
  // ---- epilogue.  Each lane rounds its accumulators (x alpha + bias) to fp16 -- the rounding point of a Linear
  // output -- and the WHOLE 256 x 256 fp16 tile is staged once (row stride 528 B); every thread then owns 8
  // consecutive columns of 16 rows: one ds_read_b128 and one 16-B store per row and fp16 destination.  Global loads
  // (fp32 residual / position rows / fp16 pre-activations) are all issued before the first store, in two batches
  // around the staging pass, and settled once (see ns_gemm_epi.h on why that matters).
  char* const hs = smem;
  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  const int ecg = tid & 31, er0 = tid >> 5;
  const int ecol = n0 + ecg * 8;
  const bool ecolok = ecol + 8 <= p.N;
  const int ecolc = min(ecol, p.N - 8);
  // residual epilogue: a thread owns columns {4 ecg .. +3} and {128 + 4 ecg .. +3} of its rows instead of 8 consecutive ones, so that the
  // fp32 accesses of a wave instruction are 16 B per lane at 16-B pitch (512 contiguous bytes per row) and not every other 16-B chunk
  const int rcolA = n0 + ecg * 4, rcolB = rcolA + 128;
  const bool rokA = rcolA + 4 <= p.N, rokB = rcolB + 4 <= p.N;
  const int rcolAc = min(rcolA, p.N - 4), rcolBc = min(rcolB, p.N - 4);
  auto stage_tile = [&]() __attribute__((always_inline)) {
    float4 bz[2][2];
#pragma unroll
    for (int bh = 0; bh < 2; ++bh)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = min(n0 + wn * 64 + bh * 32 + nt * 16 + 4 * lg, p.N - 4);
        bz[bh][nt] = p.bias ? *(const float4*)(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int rl = wm * 128 + ah * 64 + mt * 16 + l15;
            const int cl = wn * 64 + bh * 32 + nt * 16 + 4 * lg;
            const f32x4 a = acc[ah][mt][bh][nt];
            const half4 h = {(half_t)(a[0] * alpha + bz[bh][nt].x), (half_t)(a[1] * alpha + bz[bh][nt].y),
                             (half_t)(a[2] * alpha + bz[bh][nt].z), (half_t)(a[3] * alpha + bz[bh][nt].w)};
            *(half4*)(hs + rl * LDH + cl * 2) = h;
          }
  };
  auto epilogue = [&](auto kind_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    half_t* const C16 = (half_t*)p.C16;
    half_t* const G16 = (half_t*)p.G16;
    const half_t* const P16 = (const half_t*)p.P16;
    const bool do_gelu = p.flags & NS_GEMM_GELU;
    const bool save_grad = p.flags & NS_GEMM_GELU_SAVE_GRAD, mulp = p.flags & NS_GEMM_MUL_P16;
    const uint32_t side_thr = (KIND == NS_EPI_PLAIN && p.side_B && p.side_drop_p > 0.f) ? ns_drop_thr8(p.side_drop_p) : 0u;
    const uint32_t side_dseed = side_thr ? ns_eff_seed(p.side_drop_seed, p.seed_dev) : 0u;
    f32x4 res[KIND == NS_EPI_RES ? 16 : 1][2];
    half8 pre[KIND == NS_EPI_DGELU ? 16 : 1];
    auto prefetch = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int i = i0; i < i0 + 8; ++i) {
        const int row = min(m0 + er0 + 16 * i, p.M - 1);
        if (KIND == NS_EPI_RES) {
          const long long oh = ns_rm_off64(p.h32m, row);
          res[i][0] = p.R32 ? *(const f32x4*)(p.R32 + oh + rcolAc) : f32x4{0.f, 0.f, 0.f, 0.f};
          res[i][1] = p.R32 ? *(const f32x4*)(p.R32 + oh + rcolBc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (KIND == NS_EPI_DGELU) pre[i] = *(const half8*)(P16 + ns_rm_off64(p.p16m, row) + ecolc);
      }
      if (KIND == NS_EPI_RES && p.pos) {
#pragma unroll
        for (int i = i0; i < i0 + 8; ++i) {
          const int row = min(m0 + er0 + 16 * i, p.M - 1);
          const float* ps = p.pos + (long long)(row % p.pos_rows) * p.N;
          res[i][0] += *(const f32x4*)(ps + rcolAc);
          res[i][1] += *(const f32x4*)(ps + rcolBc);
        }
      }
    };
    // side product: this tile's 32 x 256 slice of side_B goes to LDS ONCE per workgroup (each wave fetching its own fragments
    // from global memory cost 2048 64-B requests per tile, ~12 k cycles); 16-B chunk c of row j sits at chunk c ^ (j & 15)
    char* const sbs = smem + (EPI_BYTES > 2 * BUF ? EPI_BYTES : 2 * BUF);
    uint4 sb0 = make_uint4(0, 0, 0, 0), sb1 = sb0;
    const bool side = KIND == NS_EPI_PLAIN && p.side_B != nullptr;
    if (side) {
      const half_t* const SB = (const half_t*)p.side_B + n0 + (tid & 31) * 8;
      sb0 = *(const uint4*)(SB + (long long)(tid >> 5) * p.side_ldb);
      sb1 = *(const uint4*)(SB + (long long)(16 + (tid >> 5)) * p.side_ldb);
    }
    prefetch(0);
    stage_tile();
    prefetch(8);
    if (side) {
      const int j = tid >> 5, c = tid & 31;
      *(uint4*)(sbs + j * 512 + ((c ^ (j & 15)) << 4)) = sb0;
      *(uint4*)(sbs + (16 + j) * 512 + ((c ^ (j & 15)) << 4)) = sb1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NS_P8_BARRIER();
    NS_STAMP(4);
#pragma unroll
    for (int i = 0; i < (KIND == NS_EPI_RES ? 16 : 0); ++i) { asm volatile("" : "+v"(res[i][0])); asm volatile("" : "+v"(res[i][1])); }
#pragma unroll
    for (int i = 0; i < (KIND == NS_EPI_DGELU ? 16 : 0); ++i) asm volatile("" : "+v"(pre[i]));
    NS_STAMP(5);
    if (KIND == NS_EPI_RES ? rokA : ecolok) {      // (N % 8 == 0 and tiles start at multiples of 256: group A valid whenever anything of the thread is)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = er0 + 16 * i, row = m0 + rl;
        if (row >= p.M) continue;
        half8 v;
        if (KIND == NS_EPI_RES) {
          const half4 va = *(const half4*)(hs + rl * LDH + ecg * 8), vb = *(const half4*)(hs + rl * LDH + 256 + ecg * 8);
          v = half8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        } else {
          v = *(const half8*)(hs + rl * LDH + ecg * 16);
        }
        if (KIND == NS_EPI_DGELU) {
          if (mulp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * (float)pre[i][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * ns_gelu_grad((float)pre[i][e]));
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
        if (KIND == NS_EPI_RES) {
          if (C16) {
            half_t* const c = C16 + ns_rm_off64(p.c16m, row);
            *(half4*)(c + rcolA) = half4{cv[0], cv[1], cv[2], cv[3]};
            if (rokB) *(half4*)(c + rcolB) = half4{cv[4], cv[5], cv[6], cv[7]};
          }
          if (G16) {
            half_t* const g = G16 + ns_rm_off64(p.g16m, row);
            *(half4*)(g + rcolA) = half4{gv[0], gv[1], gv[2], gv[3]};
            if (rokB) *(half4*)(g + rcolB) = half4{gv[4], gv[5], gv[6], gv[7]};
          }
        } else {
          if (C16) *(half8*)(C16 + ns_rm_off64(p.c16m, row) + ecol) = cv;
          if (G16) *(half8*)(G16 + ns_rm_off64(p.g16m, row) + ecol) = gv;
        }
        if (KIND == NS_EPI_PLAIN && p.side_B) {
          // side product (see ns_gemm_desc): the GELU values go back to this thread's own place in the staged tile,
          // LoRA-dropout mask applied, for the MFMA pass below
          uint4 w = __builtin_bit_cast(uint4, gv);
          if (side_thr) {
            uint32_t mk[4];
            ns_keep_masks(ns_drop_word(side_dseed, (uint32_t)row, (uint32_t)ecol >> 2), side_thr, mk[0], mk[1]);
            ns_keep_masks(ns_drop_word(side_dseed, (uint32_t)row, ((uint32_t)ecol >> 2) + 1), side_thr, mk[2], mk[3]);
            w.x &= mk[0]; w.y &= mk[1]; w.z &= mk[2]; w.w &= mk[3];
          }
          *(uint4*)(hs + rl * LDH + ecg * 16) = w;
        }
        if (KIND == NS_EPI_RES) {
          f32x4 h0 = res[i][0], h1 = res[i][1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { h0[e] += (float)gv[e]; h1[e] += (float)gv[4 + e]; }
          float* const hp = p.H32 + ns_rm_off64(p.h32m, row);
          *(f32x4*)(hp + rcolA) = h0;
          if (rokB) *(f32x4*)(hp + rcolB) = h1;
        }
      }
    }
    NS_STAMP(6);
    if (KIND == NS_EPI_PLAIN && p.side_B) {
      // side_out[tn][m][j] = sum_n gm[m][n] side_B[j][n0 + n] over this tile's 256 columns: wave w takes rows 32 w .. 32 w + 31
      // (two 16-row tiles) x 32 adapter rows (two 16-row tiles) x 8 steps of 32 columns; side_B on the MFMA A port, so a lane
      // owns 4 consecutive j of one output row (one 16-B store)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      NS_P8_BARRIER();
      half8 sbf[2][8];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ss = 0; ss < 8; ++ss)
          sbf[jt][ss] = *(const half8*)(sbs + (16 * jt + l15) * 512 + (((4 * ss + lg) ^ l15) << 4));
      float* const slab = p.side_out + ((long long)tn * p.M) * 32;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rl = 32 * wave + 16 * mt + l15;
        f32x4 su[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ss = 0; ss < 8; ++ss) {
          const half8 gm = *(const half8*)(hs + rl * LDH + (32 * ss + 8 * lg) * 2);
          su[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sbf[0][ss], gm, su[0], 0, 0, 0);
          su[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sbf[1][ss], gm, su[1], 0, 0, 0);
        }
        const int row = m0 + rl;
        if (row < p.M) {
          *(f32x4*)(slab + (long long)row * 32 + 4 * lg) = su[0];
          *(f32x4*)(slab + (long long)row * 32 + 16 + 4 * lg) = su[1];
        }
      }
    }
    NS_STAMP(7);
  };
  const int kind = ns_epi_kind(p);
  if (kind == NS_EPI_RES) epilogue(p8_int_c<NS_EPI_RES>{});
  else if (kind == NS_EPI_DGELU) epilogue(p8_int_c<NS_EPI_DGELU>{});
  else epilogue(p8_int_c<NS_EPI_PLAIN>{});
#ifdef NS_P8_STAMPS
  if (stamps && tid == 0) stamps[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

// operands addressed with 32-bit byte offsets (buffer loads): the last row's end must stay below 2 GiB
bool ns_gemm_p8_fits(const ns_gemm_desc* d) {
  const auto extent = [](const ns_rowmap& m, int rows, int k) -> long long {
    const long long last = m.seg_rows > 0 ? (long long)((rows - 1) / m.seg_rows) * m.seg_stride + (long long)((rows - 1) % m.seg_rows) * m.ld
                                          : (long long)(rows - 1) * m.ld;
    return 2 * (last + k + 64);
  };
  return extent(d->am, d->M, d->K) < 0x7FFF0000LL && extent(d->bm, d->N, d->K) < 0x7FFF0000LL;
}

int ns_gemm_p8_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  static ns_dev_once attr_once;      // kernel attributes, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)ns_gemm_p8_kernel<false>, (const void*)ns_gemm_p8_kernel<true>}, LDS_BYTES, "ns_gemm (p8)"))
    return NS_ERR_HIP;
  if (d->drop_p > 0.f) hipLaunchKernelGGL(ns_gemm_p8_kernel<true>, dim3(tiles), dim3(NTH), LDS_BYTES, st, *d);
  else hipLaunchKernelGGL(ns_gemm_p8_kernel<false>, dim3(tiles), dim3(NTH), LDS_BYTES, st, *d);
  return 0;
}
