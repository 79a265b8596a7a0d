// ns_gemm_ln: a residual Linear with N = 512 output columns AND the LayerNorm that reads its result, in one launch:
//     g   = round16(A W^T (+ A2 B2^T) + bias)          the Linear's fp16 output (HF:modeling_whisper.py:354 out_proj, :405 fc2;
//                                                        the second product is the LoRA up-projection, finetune.py:205-212)
//     H32 = R32 + g                                      the fp32 residual stream (HF :394, :407)
//     x16 = LN(H32) * gamma + beta, mean, rstd           the NEXT sub-layer's LayerNorm (HF :402 final_layer_norm, :392
//                                                        self_attn_layer_norm of the next layer, utils/load_model.py:468 encoder.layer_norm)
// Unfused, the step ran ns_gemm (two 256 x 256 column tiles per row block) and then ns_layernorm_fwd, which read the 196 MB
// of H32 straight back (M = 96 000): 118 + 50 us for the K = 512 projection, 230 + 50 us for fc2, twelve such pairs per step,
// each pair bound by HBM (A once, R32 in, H32 out, x16 out = 588 MB = 107 us at 5.5 TB/s for K = 512).  A workgroup that owns
// COMPLETE rows normalises them before they leave: no second pass over H32, and A is read once instead of once per column tile.
//
// Shape: 128 x 512 tile per 8-wave workgroup (2 x 4 waves, wave tile 64 x 128: the accumulator arithmetic of ns_gemm_p8's 128 x 64
// wave tile transposed), one workgroup per CU.  (A first version ran 64 x 512 tiles, two workgroups per CU, so that one workgroup's
// HBM-bound epilogue would run under the other's main loop: 176-185 us for K = 512 and 364 us for K = 2048 against 165 / 282 us for the
// two launches -- every 64 rows re-stream the whole of W through LDS-DMA, 57 B per cycle and CU, which IS the L2 -> LDS limit of a CU.)
// K advances in 32-deep steps through two LDS-DMA rings: B (the weights: L2 resident) 3 stages of 32 KiB, requested two and a
// half steps ahead; A (the activations: straight from HBM) 4 stages of 8 KiB, requested three and a half steps ahead; one
// s_barrier per step, in the middle of the step's 32 MFMAs per wave (tools/probe/w4_gemm.hip's schedule).
// 64-B LDS rows, 16-B chunk g of row r at g ^ sigma((r >> 2) & 3), sigma = (0, 2, 3, 1): conflict-free ds_read_b128 fragments.
//
// Arithmetic order = ns_gemm_p8's (second product first, then K ascending in 32-deep v_mfma_f32_16x16x32_f16 products, the B rows on
// the MFMA's A port; x alpha + bias in one fma; round16; R32 + g) and ns_layernorm_fwd's (one wave per row, lane l holds columns
// 4 l .. 4 l + 3 and 256 + 4 l .. + 3, the same sums and shuffle tree): H32, x16, mean and rstd are BITWISE what the two launches
// produce (tests/test_kernels_gpu.py::test_gemm_ln_is_bitwise_the_two_launches).
#include "ns_gemm_epi.h"

namespace {

constexpr int BM = 128, BN = 512, BK = 32, NTH = 512;
constexpr int A_ST = BM * 64;            // 8 KiB: one 32-deep step of the 128 A rows
constexpr int NA = 4;
constexpr int B_ST = BN * 64;            // 32 KiB: one 32-deep step of the 512 B rows
constexpr int NB = 3;
constexpr int A_OFF = NB * B_ST;
constexpr int LDH = BN * 2 + 16;         // epilogue: bytes per staged fp16 row
constexpr int RING_BYTES = NB * B_ST + NA * A_ST;                         // 128 KiB
constexpr int LDS_BYTES = BM * LDH > RING_BYTES ? BM * LDH : RING_BYTES;  // the staged fp16 tile (130 KiB) reuses the rings

typedef __attribute__((address_space(3))) void lds_void;

#define RL_BARRIER()                            \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)
#define RL_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int sigma4(int x) { return (0x1320 >> (4 * x)) & 3; }   // (0, 2, 3, 1)

__global__ __launch_bounds__(NTH) void gemm_ln_kernel(const ns_gemm_ln_desc q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ns_gemm_desc& p = q.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * BM;

  // acc[mt][nt]: row m0 + 64 wm + 16 mt + l15, columns 128 wn + 16 nt + 4 lg + e
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x80000000u, 0x00020000);
  // DMA sources (byte offsets; a piece = 16 rows x 64 B, lane-linear in LDS): wave w fills A piece w and B pieces 4 w .. 4 w + 3
  uint32_t a_off, b_off[4];
  {
    const int g = (lane & 3) ^ sigma4((lane >> 4) & 3);       // row in piece = lane >> 2, so (row >> 2) & 3 = (lane >> 4) & 3
    const int arow = min(m0 + 16 * wave + (lane >> 2), p.M - 1);
    a_off = 2u * (uint32_t)(ns_rm_off64(p.am, arow) + g * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int brow = 64 * wave + 16 * j + (lane >> 2);
      b_off[j] = 2u * ((uint32_t)brow * (uint32_t)p.bm.ld + (uint32_t)g * 8u);
    }
  }
  const int nsteps = p.K / BK;
  // (ring slots are passed in: the loop below is unrolled over lcm(NA, NB) steps, so they are compile-time constants there)
  auto dma_a = [&](int t, int slot) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(smem + A_OFF + slot * A_ST + wave * 1024), 16,
                                             t < nsteps ? a_off : 0x80000000u, 2 * BK * t, 0, 0);
  };
  auto dma_b = [&](int t, int slot, int j) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(smem + slot * B_ST + (4 * wave + j) * 1024), 16,
                                             t < nsteps ? b_off[j] : 0x80000000u, 2 * BK * t, 0, 0);
  };

  // ---- second product (LoRA up-projection, K2 = 16 or 32), formed FIRST as in ns_gemm_p8: fragments straight from global memory
  // (u was just written: L2 / Infinity Cache), requested ahead of the prologue's DMA pieces so that a counted wait retires them alone
  half8 a2f[4], b2f[8];
  const bool k2ok = p.K2 > 0 && 8 * lg < p.K2;
  if (p.K2 > 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = min(m0 + 64 * wm + mt * 16 + l15, p.M - 1);
      const half_t* ap = (const half_t*)p.A2 + ns_rm_off64(p.am2, row) + (k2ok ? 8 * lg : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a2f[mt]) : "v"(ap) : "memory");
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const half_t* bp = (const half_t*)p.B2 + (long long)(128 * wn + nt * 16 + l15) * p.ldb2 + (k2ok ? 8 * lg : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b2f[nt]) : "v"(bp) : "memory");
    }
  }
  // ---- prologue, in the loop's own request order (step s requests B(s + 3), then A(s + 4)):
  //      A(0) | B(0), A(1) | B(1), A(2) | B(2), A(3)        1 + 3 x 5 = 16 pieces
  dma_a(0, 0);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_b(s, s, j);
    dma_a(s + 1, s + 1);
  }
  if (p.K2 > 0) {
    asm volatile("s_waitcnt vmcnt(16)"
                 : "+v"(a2f[0]), "+v"(a2f[1]), "+v"(a2f[2]), "+v"(a2f[3]), "+v"(b2f[0]), "+v"(b2f[1]), "+v"(b2f[2]), "+v"(b2f[3]),
                   "+v"(b2f[4]), "+v"(b2f[5]), "+v"(b2f[6]), "+v"(b2f[7])
                 :: "memory");
    RL_SB();
    if (!k2ok) {      // lanes whose 8 k-values lie past K2 (K2 = 16: lanes 32..63) contribute zeros
      const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a2f[mt] = hz;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2f[nt], a2f[mt], acc[mt][nt], 0, 0, 0);
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
  asm volatile("s_waitcnt vmcnt(11)" ::: "memory");      // A(0) and B(0) have landed (11 newer pieces may still travel)
  RL_BARRIER();
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) read_b(0, 0, nt);
  read_a(0, 0);
  read_a(0, 1);

  // One step = 32 MFMAs per wave around ONE barrier.  Before it: rows 0..31 of the wave tile and the step's last two A fragments.
  // Behind it -- every wave has finished step t - 1, whose second half read the fragments of step t, so the stages of B(t) and A(t)
  // are free --: rows 32..63, the fragments of step t + 1, the requests B(t + 3) -> B(t)'s stage and A(t + 4) -> A(t)'s stage.
  // The counted wait in front of the barrier lets the six newest pieces travel on -- A(t + 2), B(t + 2), A(t + 3) -- so B(t + 1),
  // requested a step and a half ago, and A(t + 1), requested two and a half steps ago, have landed.
  auto step = [&](int t, int set, int sa, int sa1, int sb, int sb1) __attribute__((always_inline)) {   // slots of A(t), A(t+1), B(t), B(t+1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RL_SB();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int mt = h >> 1, nb = (h & 1) * 4;
      mma(set, mt, nb + 0); mma(set, mt, nb + 1);
      if (h < 2) read_a(sa, 2 + h);
      mma(set, mt, nb + 2); mma(set, mt, nb + 3);
      RL_SB();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    RL_BARRIER();
    RL_SB();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int mt = 2 + (h >> 1), nb = (h & 1) * 4;
      mma(set, mt, nb + 0); mma(set, mt, nb + 1);
      read_b(sb1, set ^ 1, 2 * h);
      read_b(sb1, set ^ 1, 2 * h + 1);
      if (h >= 2) read_a(sa1, h - 2);            // af[0..1] are dead from the first half on
      dma_b(t + 3, sb, h);
      if (h == 3) dma_a(t + 4, sa);
      mma(set, mt, nb + 2); mma(set, mt, nb + 3);
      RL_SB();
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
  RL_BARRIER();

  // ---- epilogue, part 1: x alpha + bias, round to fp16 (the rounding point of the Linear's output), stage the 128 x 512 tile
  char* const hs = smem;
  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;
  {
    float4 bz[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      bz[nt] = p.bias ? *(const float4*)(p.bias + 128 * wn + nt * 16 + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int rl = 64 * wm + mt * 16 + l15, cl = 128 * wn + nt * 16 + 4 * lg;
        const f32x4 a = acc[mt][nt];
        const half4 h = {(half_t)(a[0] * alpha + bz[nt].x), (half_t)(a[1] * alpha + bz[nt].y), (half_t)(a[2] * alpha + bz[nt].z),
                         (half_t)(a[3] * alpha + bz[nt].w)};
        *(half4*)(hs + rl * LDH + cl * 2) = h;
      }
  }
  // part 2: wave w owns rows 16 w .. 16 w + 15 as ns_layernorm_fwd's wave owns a row (lane l: columns 4 l .. + 3 and 256 + 4 l .. + 3).
  // Every global load (the fp32 residual rows, gamma, beta) is issued before the first store (gfx950 orders loads and stores in ONE
  // counter: a load behind a store waits for the store's acknowledgement) and settled once.
  f32x4 res[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = min(m0 + 16 * wave + i, p.M - 1);
    const float* rp = p.R32 + ns_rm_off64(p.h32m, row);
    res[i][0] = p.R32 ? *(const f32x4*)(rp + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
    res[i][1] = p.R32 ? *(const f32x4*)(rp + 256 + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 gm[2], bt[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    gm[i] = *(const f32x4*)(q.gamma + i * 256 + 4 * lane);
    bt[i] = *(const f32x4*)(q.beta + i * 256 + 4 * lane);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  RL_BARRIER();
#pragma unroll
  for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(res[i][0])); asm volatile("" : "+v"(res[i][1])); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { asm volatile("" : "+v"(gm[i])); asm volatile("" : "+v"(bt[i])); }
  half_t* const X16 = (half_t*)q.x16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rl = 16 * wave + i, row = m0 + rl;
    const half4 g0 = *(const half4*)(hs + rl * LDH + 8 * lane);
    const half4 g1 = *(const half4*)(hs + rl * LDH + 512 + 8 * lane);
    f32x4 h0 = res[i][0], h1 = res[i][1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { h0[e] += (float)g0[e]; h1[e] += (float)g1[e]; }
    // ns_layernorm_fwd's sums, term for term
    float s = 0.f;
    s += (h0[0] + h0[1]) + (h0[2] + h0[3]);
    s += (h1[0] + h1[1]) + (h1[2] + h1[3]);
    const float mean = ns_wave_sum(s) / 512;
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float c = h0[e] - mean; qq += c * c; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float c = h1[e] - mean; qq += c * c; }
    const float rstd = rsqrtf(ns_wave_sum(qq) / 512 + q.eps);
    if (row < p.M) {
      float* const hp = p.H32 + ns_rm_off64(p.h32m, row);
      *(f32x4*)(hp + 4 * lane) = h0;
      *(f32x4*)(hp + 256 + 4 * lane) = h1;
      if (lane == 0) {
        if (q.mean) q.mean[row] = mean;
        if (q.rstd) q.rstd[row] = rstd;
      }
      half4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = (half_t)((h0[e] - mean) * rstd * gm[0][e] + bt[0][e]);
        o1[e] = (half_t)((h1[e] - mean) * rstd * gm[1][e] + bt[1][e]);
      }
      *(half4*)(X16 + (long long)row * q.ldx + 4 * lane) = o0;
      *(half4*)(X16 + (long long)row * q.ldx + 256 + 4 * lane) = o1;
    }
  }
}

}  // namespace

extern "C" int ns_gemm_ln_supported(int M, int N, int K, int K2) {
  return M >= 1024 && N == BN && K >= 2 * BK && K % (2 * BK) == 0 && (K2 == 0 || K2 == 16 || K2 == 32);
}

extern "C" int ns_gemm_ln(const ns_gemm_ln_desc* q, void* stream) {
  NS_CHECK_ARG(q, "ns_gemm_ln: null descriptor");
  const ns_gemm_desc* d = &q->g;
  NS_CHECK_ARG(d->A && d->B && d->H32 && q->gamma && q->beta && q->x16, "ns_gemm_ln: null pointer");
  NS_CHECK_ARG(ns_gemm_ln_supported(d->M, d->N, d->K, d->K2), "ns_gemm_ln: unsupported shape M=%d N=%d K=%d K2=%d (N = 512, K %% 64 == 0, M >= 1024)",
               d->M, d->N, d->K, d->K2);
  NS_CHECK_ARG(d->flags == 0 && !d->C16 && !d->G16 && !d->P16 && !d->C32 && !d->pos && !d->side_B && d->drop_p == 0.f && d->splits <= 1 &&
                   d->a2_ngroup == 0,
               "ns_gemm_ln: only the plain residual Linear (+ second product) is built: no flags, fp16 / fp32 side outputs, positions, dropout");
  NS_CHECK_ARG(d->bm.seg_rows == 0 && d->bm.ld % 8 == 0 && d->am.ld % 8 == 0 && d->am.seg_stride % 8 == 0 && d->h32m.ld % 4 == 0 &&
                   d->h32m.seg_rows == 0 && q->ldx >= BN && q->ldx % 4 == 0,
               "ns_gemm_ln: bad strides");
  NS_CHECK_ARG(d->K2 == 0 || (d->A2 && d->B2 && d->am2.ld % 8 == 0 && d->ldb2 % 8 == 0), "ns_gemm_ln: second product operands");
  {   // 32-bit byte offsets (buffer loads): the operands' last rows must end below 2 GiB
    const ns_rowmap& m = d->am;
    const long long last = m.seg_rows > 0 ? (long long)((d->M - 1) / m.seg_rows) * m.seg_stride + (long long)((d->M - 1) % m.seg_rows) * m.ld
                                          : (long long)(d->M - 1) * m.ld;
    NS_CHECK_ARG(2 * (last + d->K + 64) < 0x7FFF0000LL && 2LL * BN * d->bm.ld < 0x7FFF0000LL, "ns_gemm_ln: operand beyond 2 GiB");
  }
  static ns_dev_once attr_once;      // kernel attribute, once per device (ns_common.h)
  if (!ns_dyn_lds_once(attr_once, {(const void*)gemm_ln_kernel}, LDS_BYTES, "ns_gemm_ln")) return NS_ERR_HIP;
  hipLaunchKernelGGL(gemm_ln_kernel, dim3((d->M + BM - 1) / BM), dim3(NTH), LDS_BYTES, (hipStream_t)stream, *q);
  NS_CHECK_LAUNCH("ns_gemm_ln");
  return NS_OK;
}
