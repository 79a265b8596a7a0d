// ns_gemm NT, large-M form with TWO workgroups per CU: 128 x 256 tile, 4 waves (2 x 2), wave tile 64 x 128 -- the accumulator
// arithmetic of ns_gemm_p8's 128 x 64 wave tile transposed, the same v_mfma_f32_16x16x32_f16 products in the same order (second
// product first, then K ascending), so outputs are bit-identical to ns_gemm_p8 / ns_gemm_p8s (tests/test_kernels_gpu.py).
//
// Why: ns_gemm_p8* holds a CU with ONE 8-wave workgroup (152 KiB of LDS, 2 x 238 registers per SIMD), so a tile's epilogue --
// 23 k cycles of erf / exp VALU for a GELU tile, 16 k more for the side product, or the wait for 128 KiB of fp16 pre-activations /
// fp32 residual rows from HBM -- runs with the matrix pipe idle, and its main loop runs with HBM idle: at K = 512 the fc1 launch
// (GELU + saved gelu' + side product) takes 76 k cycles per 256 x 256 tile of which 24 k are main loop, the fc2 dgrad (x gelu',
// dropout mask, adapter product) 56 k.  With half the tile per workgroup (80 KiB of LDS, 4 waves) two workgroups share a CU and
// the hardware interleaves one's epilogue with the other's main loop: the two waves of a SIMD belong to different workgroups that
// drift apart by themselves.  The price is operand traffic: 24 KiB of LDS-DMA per 128 x 256 x 32 step against 32 KiB per
// 256 x 256 x 32 (1.5 x per FLOP; 47 B per cycle and CU with both workgroups in their main loops, under the ~ 67 B the L2 -> LDS path
// of a CU delivers); LDS fragment reads per MFMA are ns_gemm_p8's.
//
// K advances in 32-deep steps through two LDS-DMA rings (csrc/ns_gemm_rowln.hip has the same schedule): B 3 stages of 16 KiB
// requested two and a half steps ahead, A 4 stages of 8 KiB requested three and a half steps ahead, ONE s_barrier per step in the
// middle of the step's 32 MFMAs per wave; counted s_waitcnt vmcnt(8), never 0 inside the loop; K tails and padding steps arrive as
// zeros from the buffer range check.  64-B LDS rows, 16-B chunk g of row r at g ^ sigma((r >> 2) & 3), sigma = (0, 2, 3, 1).
// Epilogue: ns_gemm_p8's (fp16 rounding in registers, the whole tile staged once, 16-B row accesses, every global load issued before
// the first store and settled once), on a staged tile WITHOUT row padding -- 16-B chunk c of row r at c ^ (r & 7) -- so that the
// tile (64 KiB) and the side product's 32 x 256 slice of side_B (16 KiB) fit the 80 KiB the rings leave.
#include "ns_gemm_epi.h"

namespace {

constexpr int BM = 128, BN = 256, BK = 32, NTH = 256;
constexpr int A_ST = BM * 64;            // 8 KiB
constexpr int NA = 4;
constexpr int B_ST = BN * 64;            // 16 KiB
constexpr int NB = 3;
constexpr int A_OFF = NB * B_ST;
constexpr int LDS_BYTES = NB * B_ST + NA * A_ST;     // 80 KiB: two workgroups per CU
constexpr int LDH = BN * 2;              // staged fp16 row, unpadded (chunk-swizzled)
constexpr int SIDE_OFF = BM * LDH;       // 64 KiB
static_assert(SIDE_OFF + 32 * 256 * 2 <= LDS_BYTES, "staged tile + side_B slice");

typedef __attribute__((address_space(3))) void lds_void;

#define P4_BARRIER()                            \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)
#define P4_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int sigma4(int x) { return (0x1320 >> (4 * x)) & 3; }   // (0, 2, 3, 1)
// byte offset of (row rl, byte b of the row) in the staged tile
__device__ __forceinline__ int hs_off(int rl, int b) { return rl * LDH + ((((b >> 4) ^ (rl & 7)) << 4) | (b & 15)); }

template <int V> struct p4_int_c { static constexpr int value = V; };

template <bool DROP>
__global__ __launch_bounds__(NTH, 2) void ns_gemm_p4_kernel(const ns_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int wgid;
  {   // XCD-aware bijective remap: consecutive tiles (the column tiles of a row block: one A tile) run on one XCD, sharing its L2
    const int bid = blockIdx.x;
    const int qd = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wgid = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // acc[mt][nt]: row m0 + 64 wm + 16 mt + l15, columns n0 + 128 wn + 16 nt + 4 lg + e
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x80000000u, 0x00020000);
  // DMA sources (byte offsets; a piece = 16 rows x 64 B, lane-linear in LDS): wave w fills A pieces 2 w, 2 w + 1 and B pieces 4 w .. 4 w + 3
  uint32_t a_off[2], b_off[4];
  const int g = (lane & 3) ^ sigma4((lane >> 4) & 3);       // row in piece = lane >> 2, so (row >> 2) & 3 = (lane >> 4) & 3
  {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int arow = min(m0 + 16 * (2 * wave + j) + (lane >> 2), p.M - 1);
      a_off[j] = 2u * (uint32_t)(ns_rm_off64(p.am, arow) + g * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int brow = min(n0 + 16 * (4 * wave + j) + (lane >> 2), p.N - 1);
      b_off[j] = 2u * ((uint32_t)brow * (uint32_t)p.bm.ld + (uint32_t)g * 8u);
    }
  }
  const int nsteps = (p.K + BK - 1) / BK;
  // a lane whose 8 k-values lie past K (K tails: the first conv's 3 x 208 = 624, padding steps) asks beyond num_records: zeros
  auto dma_a = [&](int t, int slot, int j) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(smem + A_OFF + slot * A_ST + (2 * wave + j) * 1024), 16,
                                             BK * t + 8 * g < p.K ? a_off[j] : 0x80000000u, 2 * BK * t, 0, 0);
  };
  auto dma_b = [&](int t, int slot, int j) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(smem + slot * B_ST + (4 * wave + j) * 1024), 16,
                                             BK * t + 8 * g < p.K ? b_off[j] : 0x80000000u, 2 * BK * t, 0, 0);
  };

  // ---- second product (the LoRA up-projection, K2 = r or 3 r: 16 .. 96), formed FIRST as in ns_gemm_p8: fragments straight from
  // global memory, requested ahead of the prologue's DMA pieces so that a counted wait retires them alone
  half8 a2f[4], b2f[8];
  const half_t* a2p[4];
  const half_t* b2p[8];
  const bool k2ok = p.K2 > 0 && 8 * lg < p.K2;
  if (p.K2 > 0) {
    // a 128-column wave strip lies inside ONE column group (a2_ngroup % 128 == 0, checked by the launcher)
    const int goff = p.a2_ngroup > 0 ? ((n0 + wn * 128) / p.a2_ngroup) * p.K2 : 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = min(m0 + 64 * wm + mt * 16 + l15, p.M - 1);
      a2p[mt] = (const half_t*)p.A2 + ns_rm_off64(p.am2, row) + goff + (k2ok ? 8 * lg : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a2f[mt]) : "v"(a2p[mt]) : "memory");
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int col = min(n0 + 128 * wn + nt * 16 + l15, p.N - 1);
      b2p[nt] = (const half_t*)p.B2 + (long long)col * p.ldb2 + (k2ok ? 8 * lg : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b2f[nt]) : "v"(b2p[nt]) : "memory");
    }
  }
  // ---- prologue, in the loop's own request order (step s requests B(s + 3) x 4, then A(s + 4) x 2):
  //      A(0) | B(0), A(1) | B(1), A(2) | B(2), A(3)        2 + 3 x 6 = 20 pieces
  dma_a(0, 0, 0); dma_a(0, 0, 1);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_b(s, s, j);
    dma_a(s + 1, s + 1, 0); dma_a(s + 1, s + 1, 1);
  }
  if (p.K2 > 0) {
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mma2 = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2f[nt], a2f[mt], acc[mt][nt], 0, 0, 0);
    };
    asm volatile("s_waitcnt vmcnt(20)"
                 : "+v"(a2f[0]), "+v"(a2f[1]), "+v"(a2f[2]), "+v"(a2f[3]), "+v"(b2f[0]), "+v"(b2f[1]), "+v"(b2f[2]), "+v"(b2f[3]),
                   "+v"(b2f[4]), "+v"(b2f[5]), "+v"(b2f[6]), "+v"(b2f[7])
                 :: "memory");
    P4_SB();
    if (!k2ok) {      // lanes whose 8 k-values lie past K2 (K2 = 16: lanes 32..63) contribute zeros
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a2f[mt] = hz;
    }
    mma2();
    // further rounds (K2 > 32: the stacked q|k|v bottleneck of a dgrad, 3 r): loaded behind the DMA pieces, so their wait also covers the
    // prologue (which the first step needs anyway)
    for (int k0 = 32; k0 < p.K2; k0 += 32) {
      const bool ok = k0 + 8 * lg < p.K2;
      const int ko = ok ? k0 : 0;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a2f[mt]) : "v"(a2p[mt] + ko) : "memory");
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b2f[nt]) : "v"(b2p[nt] + ko) : "memory");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(a2f[0]), "+v"(a2f[1]), "+v"(a2f[2]), "+v"(a2f[3]), "+v"(b2f[0]), "+v"(b2f[1]), "+v"(b2f[2]), "+v"(b2f[3]),
                     "+v"(b2f[4]), "+v"(b2f[5]), "+v"(b2f[6]), "+v"(b2f[7])
                   :: "memory");
      P4_SB();
      if (!ok) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a2f[mt] = hz;
      }
      mma2();
    }
    if (DROP) {
      // LoRA-dropout mask on the (A2, B2) product, before the main product accumulates on top
      const uint32_t drop_thr = ns_drop_thr8(p.drop_p);
      const uint32_t dseed = ns_eff_seed(p.drop_seed, p.seed_dev);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const uint32_t row = (uint32_t)(m0 + 64 * wm + mt * 16 + l15);
          const uint32_t col = (uint32_t)(n0 + 128 * wn + nt * 16 + 4 * lg);
          const uint32_t w = ns_drop_word(dseed, row, col >> 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][nt][e] = ns_keep(w, e, drop_thr) ? acc[mt][nt][e] : 0.f;
        }
    }
  }

  const int fro = l15 * 64 + ((lg ^ sigma4((l15 >> 2) & 3)) << 4);
  half8 af[4], bf[2][8];
  auto read_a = [&](int slot, int mt) __attribute__((always_inline)) {
    af[mt] = *(const half8*)(smem + A_OFF + slot * A_ST + (4 * wm + mt) * 1024 + fro);
  };
  auto read_b = [&](int slot, int set, int nt) __attribute__((always_inline)) {
    bf[set][nt] = *(const half8*)(smem + slot * B_ST + (8 * wn + nt) * 1024 + fro);
  };
  auto mma = [&](int set, int mt, int nt) __attribute__((always_inline)) {
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[set][nt], af[mt], acc[mt][nt], 0, 0, 0);
  };
  asm volatile("s_waitcnt vmcnt(14)" ::: "memory");      // A(0) and B(0) have landed (14 newer pieces may still travel)
  P4_BARRIER();
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) read_b(0, 0, nt);
  read_a(0, 0);
  read_a(0, 1);

  // One step = 32 MFMAs per wave around ONE barrier.  Before it: rows 0..31 of the wave tile and the step's last two A fragments.
  // Behind it -- every wave has finished step t - 1, whose second half read the fragments of step t, so the stages of B(t) and A(t)
  // are free --: rows 32..63, the fragments of step t + 1, the requests B(t + 3) -> B(t)'s stage and A(t + 4) -> A(t)'s stage.
  // The counted wait in front of the barrier lets the eight newest pieces travel on -- A(t + 2), B(t + 2), A(t + 3) -- so B(t + 1),
  // requested a step and a half ago, and A(t + 1), requested two and a half steps ago, have landed.
  auto step = [&](int t, int set, int sa, int sa1, int sb, int sb1) __attribute__((always_inline)) {   // slots of A(t), A(t+1), B(t), B(t+1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P4_SB();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int mt = h >> 1, nb = (h & 1) * 4;
      mma(set, mt, nb + 0); mma(set, mt, nb + 1);
      if (h < 2) read_a(sa, 2 + h);
      mma(set, mt, nb + 2); mma(set, mt, nb + 3);
      P4_SB();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    P4_BARRIER();
    P4_SB();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int mt = 2 + (h >> 1), nb = (h & 1) * 4;
      mma(set, mt, nb + 0); mma(set, mt, nb + 1);
      read_b(sb1, set ^ 1, 2 * h);
      read_b(sb1, set ^ 1, 2 * h + 1);
      if (h >= 2) read_a(sa1, h - 2);            // af[0..1] are dead from the first half on
      dma_b(t + 3, sb, h);
      if (h == 3) { dma_a(t + 4, sa, 0); dma_a(t + 4, sa, 1); }      // strictly behind the four B pieces: the counted wait relies on the order
      mma(set, mt, nb + 2); mma(set, mt, nb + 3);
      P4_SB();
    }
  };
  constexpr int UN = 12;       // lcm(2 fragment sets, NA = 4, NB = 3)
  for (int t = 0; t < nsteps; t += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (t + u < nsteps) step(t + u, u & 1, u % NA, (u + 1) % NA, u % NB, (u + 1) % NB);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing (zero) pieces must not land on the staged tile
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  P4_BARRIER();

  // ---- epilogue (ns_gemm_p8's, on a 128 x 256 tile: a thread owns 8 consecutive columns of 16 rows)
  char* const hs = smem;
  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  const int ecg = tid & 31, er0 = tid >> 5;
  const int ecol = n0 + ecg * 8;
  const bool ecolok = ecol + 8 <= p.N;
  const int ecolc = min(ecol, p.N - 8);
  // residual epilogue: a thread owns columns {4 ecg .. +3} and {128 + 4 ecg .. +3} of its rows instead of 8 consecutive ones, so that the
  // fp32 accesses of a wave instruction are 16 B per lane at 16-B pitch (512 contiguous bytes per row)
  const int rcolA = n0 + ecg * 4, rcolB = rcolA + 128;
  const bool rokA = rcolA + 4 <= p.N, rokB = rcolB + 4 <= p.N;
  const int rcolAc = min(rcolA, p.N - 4), rcolBc = min(rcolB, p.N - 4);
  auto stage_tile = [&]() __attribute__((always_inline)) {
    float4 bz[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int col = min(n0 + 128 * wn + nt * 16 + 4 * lg, p.N - 4);
      bz[nt] = p.bias ? *(const float4*)(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int rl = 64 * wm + mt * 16 + l15;
        const int cl = 128 * wn + nt * 16 + 4 * lg;
        const f32x4 a = acc[mt][nt];
        const half4 h = {(half_t)(a[0] * alpha + bz[nt].x), (half_t)(a[1] * alpha + bz[nt].y), (half_t)(a[2] * alpha + bz[nt].z),
                         (half_t)(a[3] * alpha + bz[nt].w)};
        *(half4*)(hs + hs_off(rl, cl * 2)) = h;
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
        const int row = min(m0 + er0 + 8 * i, p.M - 1);
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
          const int row = min(m0 + er0 + 8 * i, p.M - 1);
          const float* ps = p.pos + (long long)(row % p.pos_rows) * p.N;
          res[i][0] += *(const f32x4*)(ps + rcolAc);
          res[i][1] += *(const f32x4*)(ps + rcolBc);
        }
      }
    };
    // side product: this tile's 32 x 256 slice of side_B goes to LDS once per workgroup, by LDS-DMA (no staging registers: sixteen more
    // live registers beside the 128 accumulators spilled): 16 pieces of 2 rows x 512 B, four per wave; 16-B chunk c of row j at chunk
    // c ^ (j & 15), applied on the source side.  The rings are dead (the main loop's last barrier has passed); the pieces land under the
    // staging pass and are waited for in front of the barrier below (a PLAIN epilogue has no other load in flight there).
    char* const sbs = smem + SIDE_OFF;
    const bool side = KIND == NS_EPI_PLAIN && p.side_B != nullptr;
    if (side) {
      const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.side_B, 0, 0x80000000u, 0x00020000);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = 2 * (4 * wave + k) + (lane >> 5);
        const int c = (lane & 31) ^ (j & 15);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lds_void*)(sbs + (4 * wave + k) * 1024), 16,
                                                 2u * ((uint32_t)j * (uint32_t)p.side_ldb + (uint32_t)(n0 + c * 8)), 0, 0, 0);
      }
    }
    prefetch(0);
    stage_tile();
    prefetch(8);
    if (side) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P4_BARRIER();
#pragma unroll
    for (int i = 0; i < (KIND == NS_EPI_RES ? 16 : 0); ++i) { asm volatile("" : "+v"(res[i][0])); asm volatile("" : "+v"(res[i][1])); }
#pragma unroll
    for (int i = 0; i < (KIND == NS_EPI_DGELU ? 16 : 0); ++i) asm volatile("" : "+v"(pre[i]));
    if (KIND == NS_EPI_RES ? rokA : ecolok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = er0 + 8 * i, row = m0 + rl;
        if (row >= p.M) continue;
        half8 v;
        if (KIND == NS_EPI_RES) {
          const half4 va = *(const half4*)(hs + hs_off(rl, ecg * 8)), vb = *(const half4*)(hs + hs_off(rl, 256 + ecg * 8));
          v = half8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        } else {
          v = *(const half8*)(hs + hs_off(rl, ecg * 16));
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
            half_t* const gq = G16 + ns_rm_off64(p.g16m, row);
            *(half4*)(gq + rcolA) = half4{gv[0], gv[1], gv[2], gv[3]};
            if (rokB) *(half4*)(gq + rcolB) = half4{gv[4], gv[5], gv[6], gv[7]};
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
          *(uint4*)(hs + hs_off(rl, ecg * 16)) = w;       // (no LDS-DMA piece is in flight here: a plain store carries no extra wait)
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
    if (KIND == NS_EPI_PLAIN && p.side_B) {
      // side_out[tn][m][j] = sum_n gm[m][n] side_B[j][n0 + n] over this tile's 256 columns: wave w takes rows 32 w .. 32 w + 31
      // (two 16-row tiles) x 32 adapter rows (two 16-row tiles) x 8 steps of 32 columns; side_B on the MFMA A port, so a lane
      // owns 4 consecutive j of one output row (one 16-B store)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      P4_BARRIER();
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
          const half8 gm = *(const half8*)(hs + hs_off(rl, (32 * ss + 8 * lg) * 2));
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
  };
  const int kind = ns_epi_kind(p);
  if (kind == NS_EPI_RES) epilogue(p4_int_c<NS_EPI_RES>{});
  else if (kind == NS_EPI_DGELU) epilogue(p4_int_c<NS_EPI_DGELU>{});
  else epilogue(p4_int_c<NS_EPI_PLAIN>{});
}

}  // namespace

// what ns_gemm_p8 takes, with whole-strip column groups for the second product
bool ns_gemm_p4_ok(const ns_gemm_desc* d) {
  if (d->K % 8 != 0) return false;
  if (d->K2 > 0 && d->a2_ngroup > 0 && d->a2_ngroup % 128 != 0) return false;
  return true;
}

int ns_gemm_p4_launch(const ns_gemm_desc* d, hipStream_t st) {
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  static ns_dev_once attr_once;      // kernel attributes, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)ns_gemm_p4_kernel<false>, (const void*)ns_gemm_p4_kernel<true>}, LDS_BYTES, "ns_gemm (p4)"))
    return NS_ERR_HIP;
  if (d->drop_p > 0.f) hipLaunchKernelGGL(ns_gemm_p4_kernel<true>, dim3(tiles), dim3(NTH), LDS_BYTES, st, *d);
  else hipLaunchKernelGGL(ns_gemm_p4_kernel<false>, dim3(tiles), dim3(NTH), LDS_BYTES, st, *d);
  return 0;
}
