// ns_gemm: fp16 MFMA GEMM (C = A * B^T [+ A2 * B2^T]) with fused epilogues.
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_f16.
// LDS image per operand tile: [128 rows][64 k] fp16 = 128-B rows, 16-B chunk
// index XOR-swizzled with (row>>1)&7 so the ds_read_b128 fragment reads
// (lane = row, same chunk) are bank-conflict free (64-dword bank row, 16-lane
// groups).  Staging is register-staged and software-pipelined: global loads of
// tile t+1 are issued before the MFMA phase of tile t and written to the other
// LDS buffer after it (one barrier per K-step).
//
// Two staging front-ends share the MFMA loop and the epilogue:
//   NT: A (M x K) and B (N x K) are k-contiguous -> 16-B global loads.
//   TN: both operands are reduction-major (weight gradients dW = dY^T X);
//       each thread loads 8 reduction rows x 2 columns as dwords and packs
//       them into two k-contiguous 16-B LDS chunks (register transpose).
#include "ns_common.h"

namespace {

constexpr int BM = 128, BK = 64, NT_THREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per A tile

__device__ __forceinline__ uint32_t rm_off(const ns_rowmap& m, int row) {
  if (m.seg_rows > 0) {
    const int s = row / m.seg_rows;
    const int w = row - s * m.seg_rows;
    return (uint32_t)((long long)s * m.seg_stride + (long long)w * m.ld);
  }
  return (uint32_t)row * (uint32_t)m.ld;
}
__device__ __forceinline__ long long rm_off64(const ns_rowmap& m, int row) {
  if (m.seg_rows > 0) {
    const int s = row / m.seg_rows;
    const int w = row - s * m.seg_rows;
    return (long long)s * m.seg_stride + (long long)w * m.ld;
  }
  return (long long)row * m.ld;
}

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// BN = 128: waves 2x2, wave tile 64x64.  BN = 32 (skinny N: LoRA down / du): waves 4x1, wave tile 32x32.
template <bool TN, int BN>
__global__ __launch_bounds__(NT_THREADS, 2) void ns_gemm_kernel(const ns_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BTILE_BYTES = BN * BK * 2;
  constexpr int WT_M = BN == 128 ? 2 : 1, WT_N = BN == 128 ? 2 : 1, WAVE_M = 32 * WT_M, WAVE_N = 32 * WT_N;
  constexpr int B_ITERS = BN / 32;
  char* const As = smem;                   // 2 buffers
  char* const Bs = smem + 2 * TILE_BYTES;  // 2 buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = BN == 128 ? (wave >> 1) : wave, wn = BN == 128 ? (wave & 1) : 0;

  // ---- XCD-aware tile mapping (bijective remap, n fastest inside an XCD chunk)
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

  const half_t* __restrict__ A = (const half_t*)p.A;
  const half_t* __restrict__ B = (const half_t*)p.B;
  const half_t* __restrict__ A2 = (const half_t*)p.A2;
  const half_t* __restrict__ B2 = (const half_t*)p.B2;

  f32x16 acc[WT_M][WT_N];
#pragma unroll
  for (int i = 0; i < WT_M; ++i)
#pragma unroll
    for (int j = 0; j < WT_N; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ------------------------------------------------------------------ staging
  // NT state
  uint32_t a_off[4], b_off[4], a2_off[4], b2_off[4];
  const int st_chunk = tid & 7, st_row = tid >> 3;
  // TN state
  int tn_seg = 0, tn_within = 0, k_begin = 0, k_end = p.K;
  const int tq = tid & 63, tg = tid >> 6;  // column pair, reduction group (it adds 4)

  if (!TN) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = st_row + 32 * it;
      a_off[it] = rm_off(p.am, min(m0 + row, p.M - 1));
      b_off[it] = (uint32_t)min(n0 + min(row, BN - 1), p.N - 1) * (uint32_t)p.bm.ld;
      a2_off[it] = 0; b2_off[it] = 0;
    }
    if (p.K2 > 0) {
      const int goff = p.a2_ngroup > 0 ? (n0 / p.a2_ngroup) * p.K2 : 0;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = st_row + 32 * it;
        a2_off[it] = rm_off(p.am2, min(m0 + row, p.M - 1)) + goff;
        b2_off[it] = (uint32_t)min(n0 + min(row, BN - 1), p.N - 1) * (uint32_t)p.ldb2;
      }
    }
  } else {
    const int chunk = (((p.K + p.splits - 1) / p.splits) + BK - 1) / BK * BK;
    k_begin = blockIdx.z * chunk;
    k_end = min(p.K, k_begin + chunk);
    if (p.am.seg_rows > 0) {
      tn_seg = k_begin / p.am.seg_rows;
      tn_within = k_begin - tn_seg * p.am.seg_rows;
    } else {
      tn_within = k_begin;
    }
  }

  const int steps1 = TN ? (max(k_end - k_begin, 0) + BK - 1) / BK : (p.K + BK - 1) / BK;
  const int steps2 = TN ? 0 : (p.K2 + BK - 1) / BK;
  const int nsteps = steps1 + steps2;
  const bool seg2_first = (!TN) && steps2 > 0 && p.drop_p > 0.f && !(p.flags & NS_GEMM_DROP_A);

  uint4 ra[4], rb[4];  // staging registers (NT: 4x16B per operand; TN: packed 2 tasks x 2 chunks)

  auto step_info = [&](int s, bool& is2, int& k0, int& klen) {
    if (seg2_first) { is2 = s < steps2; k0 = (is2 ? s : s - steps2) * BK; }
    else { is2 = s >= steps1; k0 = (is2 ? s - steps1 : s) * BK; }
    klen = (is2 ? p.K2 : (TN ? k_end - k_begin : p.K)) - k0;
    klen = klen > BK ? BK : klen;
  };

  auto load_nt = [&](int s) {
    bool is2; int k0, klen; step_info(s, is2, k0, klen);
    const int ch = (st_chunk * 8 < klen) ? st_chunk : 0;  // clamp: valid memory, never consumed
    const half_t* a = is2 ? A2 : A;
    const half_t* b = is2 ? B2 : B;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint32_t ao = is2 ? a2_off[it] : a_off[it];
      const uint32_t bo = is2 ? b2_off[it] : b_off[it];
      ra[it] = *(const uint4*)(a + (size_t)ao + k0 + ch * 8);
      if (it < B_ITERS) rb[it] = *(const uint4*)(b + (size_t)bo + k0 + ch * 8);
    }
    if ((p.flags & NS_GEMM_DROP_A) && !is2 && p.drop_p > 0.f) {
      // LoRA dropout on the A operand (forward down-projection): keep(seed,row,col)/(1-p)
      const float inv = 1.f / (1.f - p.drop_p);
      const uint32_t thr = (uint32_t)(p.drop_p * 4294967296.f);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        half8 h = __builtin_bit_cast(half8, ra[it]);
        const uint32_t grow = (uint32_t)(m0 + st_row + 32 * it);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool keep = ns_hash3(p.drop_seed, grow, (uint32_t)(k0 + ch * 8 + e)) >= thr;
          h[e] = keep ? (half_t)((float)h[e] * inv) : (half_t)0.f;
        }
        ra[it] = __builtin_bit_cast(uint4, h);
      }
    }
  };
  auto store_nt = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = st_row + 32 * it;
      *(uint4*)(As + buf * TILE_BYTES + lds_off(row, st_chunk)) = ra[it];
      if (it < B_ITERS) *(uint4*)(Bs + buf * BTILE_BYTES + lds_off(row, st_chunk)) = rb[it];
    }
  };

  // TN: task it (0,1): reduction rows kb + 8*(tg+4*it) + e, columns c0 + 2*tq, +1
  auto load_tn_operand = [&](const half_t* X, const ns_rowmap& map, int c0, int cmax, bool mask_b,
                             int kb, int klen, uint4* out) {
    const int col = min(c0 + 2 * tq, cmax - 2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      uint32_t w[8];
      const int rbase = 8 * (tg + 4 * it);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rl = rbase + e;  // local reduction row in this K-step
        uint32_t v = 0;
        if (rl < klen) {
          long long off;
          if (map.seg_rows > 0) {
            int wi = tn_within + rl, sg = tn_seg;
            if (wi >= map.seg_rows) { wi -= map.seg_rows; sg += 1; }
            off = (long long)sg * map.seg_stride + (long long)wi * map.ld;
          } else {
            off = (long long)(tn_within + rl) * map.ld;
          }
          v = *(const uint32_t*)(X + off + col);
          if (mask_b && p.drop_p > 0.f) {
            const uint32_t grow = (uint32_t)(kb + rl);
            const float inv = 1.f / (1.f - p.drop_p);
            const uint32_t thr = (uint32_t)(p.drop_p * 4294967296.f);
            half2v hv = __builtin_bit_cast(half2v, v);
            const bool k0 = ns_hash3(p.drop_seed, grow, (uint32_t)col) >= thr;
            const bool k1 = ns_hash3(p.drop_seed, grow, (uint32_t)col + 1) >= thr;
            hv[0] = k0 ? (half_t)((float)hv[0] * inv) : (half_t)0.f;
            hv[1] = k1 ? (half_t)((float)hv[1] * inv) : (half_t)0.f;
            v = __builtin_bit_cast(uint32_t, hv);
          }
        }
        w[e] = v;
      }
      // pack: low halves -> column 2q, high halves -> column 2q+1 (k-contiguous)
      uint4 lo, hi;
      lo.x = (w[0] & 0xFFFFu) | (w[1] << 16);
      lo.y = (w[2] & 0xFFFFu) | (w[3] << 16);
      lo.z = (w[4] & 0xFFFFu) | (w[5] << 16);
      lo.w = (w[6] & 0xFFFFu) | (w[7] << 16);
      hi.x = (w[0] >> 16) | (w[1] & 0xFFFF0000u);
      hi.y = (w[2] >> 16) | (w[3] & 0xFFFF0000u);
      hi.z = (w[4] >> 16) | (w[5] & 0xFFFF0000u);
      hi.w = (w[6] >> 16) | (w[7] & 0xFFFF0000u);
      out[2 * it] = lo;
      out[2 * it + 1] = hi;
    }
  };
  auto load_tn = [&](int s) {
    const int k0 = s * BK;
    const int klen = min(BK, k_end - k_begin - k0);
    load_tn_operand(A, p.am, m0, p.M, false, k_begin + k0, klen, ra);
    load_tn_operand(B, p.bm, n0, p.N, true, k_begin + k0, klen, rb);
    // advance the running (segment, within) position by one K-step
    tn_within += BK;
    if (p.am.seg_rows > 0 && tn_within >= p.am.seg_rows) { tn_within -= p.am.seg_rows; tn_seg += 1; }
  };
  auto store_tn = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int chunk = tg + 4 * it;
      *(uint4*)(As + buf * TILE_BYTES + lds_off(2 * tq, chunk)) = ra[2 * it];
      *(uint4*)(As + buf * TILE_BYTES + lds_off(2 * tq + 1, chunk)) = ra[2 * it + 1];
      *(uint4*)(Bs + buf * BTILE_BYTES + lds_off(2 * tq, chunk)) = rb[2 * it];
      *(uint4*)(Bs + buf * BTILE_BYTES + lds_off(2 * tq + 1, chunk)) = rb[2 * it + 1];
    }
  };

  auto compute = [&](int buf, int nsub) {
    const char* as = As + buf * TILE_BYTES;
    const char* bs = Bs + buf * BTILE_BYTES;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nsub) {
        half8 af[WT_M], bf[WT_N];
#pragma unroll
        for (int i = 0; i < WT_M; ++i) af[i] = *(const half8*)(as + lds_off(wm * WAVE_M + i * 32 + lr, 2 * s + lh));
#pragma unroll
        for (int j = 0; j < WT_N; ++j) bf[j] = *(const half8*)(bs + lds_off(wn * WAVE_N + j * 32 + lr, 2 * s + lh));
#pragma unroll
        for (int i = 0; i < WT_M; ++i)
#pragma unroll
          for (int j = 0; j < WT_N; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // LoRA-dropout on the dgrad path: acc currently holds du*A only.
  auto apply_mask = [&]() {
    const float inv = 1.f / (1.f - p.drop_p);
    const uint32_t thr = (uint32_t)(p.drop_p * 4294967296.f);
#pragma unroll
    for (int i = 0; i < WT_M; ++i)
#pragma unroll
      for (int j = 0; j < WT_N; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WAVE_M + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = n0 + wn * WAVE_N + j * 32 + (lane & 31);
          const bool keep = ns_hash3(p.drop_seed, (uint32_t)row, (uint32_t)col) >= thr;
          acc[i][j][r] = keep ? acc[i][j][r] * inv : 0.f;
        }
  };

  // ---------------------------------------------------------------- main loop
  if (nsteps > 0) {
    if (TN) { load_tn(0); store_tn(0); } else { load_nt(0); store_nt(0); }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      const bool more = (s + 1 < nsteps);
      if (more) { if (TN) load_tn(s + 1); else load_nt(s + 1); }
      bool is2; int k0, klen; step_info(s, is2, k0, klen);
      compute(cur, (klen + 15) >> 4);
      if (seg2_first && s == steps2 - 1) apply_mask();
      if (more) { if (TN) store_tn(cur ^ 1); else store_nt(cur ^ 1); }
      __syncthreads();
      cur ^= 1;
    }
  }

  // ----------------------------------------------------------------- epilogue
  half_t* C16 = (half_t*)p.C16;
  half_t* G16 = (half_t*)p.G16;
  const half_t* P16 = (const half_t*)p.P16;
  const bool do_gelu = p.flags & NS_GEMM_GELU;
  const bool do_dgelu = p.flags & NS_GEMM_DGELU;
  const bool atomic32 = p.flags & NS_GEMM_ATOMIC32;
  const float alpha = p.alpha == 0.f ? 1.f : p.alpha;

#pragma unroll
  for (int i = 0; i < WT_M; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rowb = m0 + wm * WAVE_M + i * 32 + 8 * g + 4 * (lane >> 5);  // multiple of 4
      if (rowb >= p.M) continue;
      long long oc = 0, og = 0, op = 0, oh = 0;
      if (C16) oc = rm_off64(p.c16m, rowb);
      if (G16) og = rm_off64(p.g16m, rowb);
      if (P16) op = rm_off64(p.p16m, rowb);
      if (p.H32) oh = rm_off64(p.h32m, rowb);
#pragma unroll
      for (int j = 0; j < WT_N; ++j) {
        const int col = n0 + wn * WAVE_N + j * 32 + (lane & 31);
        if (col >= p.N) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = rowb + e;
          if (row >= p.M) break;
          const float v = acc[i][j][4 * g + e] * alpha + bias;
          if (p.C32) {
            float* dst = p.C32 + (long long)row * p.ldc32 + col;
            if (atomic32) atomicAdd(dst, v); else *dst = v;
          }
          half_t v16 = (half_t)v;
          if (do_dgelu) v16 = (half_t)((float)v16 * ns_gelu_grad((float)P16[op + (long long)e * p.p16m.ld + col]));
          if (C16) C16[oc + (long long)e * p.c16m.ld + col] = v16;
          half_t gv = v16;
          if (do_gelu) gv = (half_t)ns_gelu((float)v16);
          if (G16) G16[og + (long long)e * p.g16m.ld + col] = gv;
          if (p.H32) {
            const long long o = oh + (long long)e * p.h32m.ld + col;
            float h = (p.R32 ? p.R32[o] : 0.f) + (float)gv;
            if (p.pos) h += p.pos[(long long)(row % p.pos_rows) * p.N + col];
            p.H32[o] = h;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int ns_gemm(const ns_gemm_desc* d, void* stream) {
  NS_CHECK_ARG(d != nullptr, "ns_gemm: null descriptor");
  NS_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "ns_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  NS_CHECK_ARG(d->A && d->B, "ns_gemm: null operand");
  const bool tn = d->flags & NS_GEMM_TN;
  const ns_rowmap* maps[] = {&d->am, &d->bm, &d->am2, &d->c16m, &d->g16m, &d->p16m, &d->h32m};
  for (const ns_rowmap* m : maps)
    NS_CHECK_ARG(m->seg_rows == 0 || (m->seg_rows % 4 == 0 && m->seg_rows >= 64),
                 "ns_gemm: seg_rows=%d must be 0 or a multiple of 4 >= 64", m->seg_rows);
  if (!tn) {
    NS_CHECK_ARG(d->K % 16 == 0 && d->am.ld % 8 == 0 && d->bm.ld % 8 == 0,
                 "ns_gemm(NT): K=%d must be a multiple of 16 and lda=%d/ldb=%d multiples of 8", d->K, d->am.ld, d->bm.ld);
    NS_CHECK_ARG(d->bm.seg_rows == 0, "ns_gemm(NT): B must be a plain row-major matrix");
    if (d->K2 > 0) {
      NS_CHECK_ARG(d->A2 && d->B2, "ns_gemm: K2>0 needs A2/B2");
      NS_CHECK_ARG(d->K2 % 16 == 0 && d->am2.ld % 8 == 0 && d->ldb2 % 8 == 0,
                   "ns_gemm(NT): K2=%d must be a multiple of 16, lda2=%d/ldb2=%d multiples of 8", d->K2, d->am2.ld, d->ldb2);
      NS_CHECK_ARG(d->a2_ngroup == 0 || d->a2_ngroup % 128 == 0, "ns_gemm: a2_ngroup must be a multiple of 128");
    }
  } else {
    NS_CHECK_ARG(d->C32 && (d->flags & NS_GEMM_ATOMIC32 || d->splits <= 1), "ns_gemm(TN): needs C32 (+ATOMIC32 when split)");
    NS_CHECK_ARG(d->am.seg_rows == d->bm.seg_rows, "ns_gemm(TN): A and B must share seg_rows");
    NS_CHECK_ARG(d->M % 2 == 0 && d->N % 2 == 0 && d->am.ld % 2 == 0 && d->bm.ld % 2 == 0,
                 "ns_gemm(TN): M, N and row strides must be even");
    NS_CHECK_ARG(d->K2 == 0, "ns_gemm(TN): second product unsupported");
    NS_CHECK_ARG(d->splits >= 1 && d->splits <= 65535, "ns_gemm(TN): bad splits=%d", d->splits);
  }
  if (d->flags & NS_GEMM_DGELU) NS_CHECK_ARG(d->P16, "ns_gemm: DGELU needs P16");
  NS_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f, "ns_gemm: drop_p out of range");

  const bool skinny = !tn && d->N <= 96;
  const int bn = skinny ? 32 : 128;
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + bn - 1) / bn);
  const size_t lds = 2 * TILE_BYTES + 2 * (size_t)bn * BK * 2;
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    const int big = 4 * TILE_BYTES;
    hipFuncSetAttribute((const void*)ns_gemm_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    hipFuncSetAttribute((const void*)ns_gemm_kernel<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    hipFuncSetAttribute((const void*)ns_gemm_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    attr_set = true;
  }
  if (tn) {
    dim3 grid(tiles, 1, d->splits);
    hipLaunchKernelGGL((ns_gemm_kernel<true, 128>), grid, dim3(NT_THREADS), lds, st, *d);
  } else if (skinny) {
    hipLaunchKernelGGL((ns_gemm_kernel<false, 32>), dim3(tiles), dim3(NT_THREADS), lds, st, *d);
  } else {
    hipLaunchKernelGGL((ns_gemm_kernel<false, 128>), dim3(tiles), dim3(NT_THREADS), lds, st, *d);
  }
  NS_CHECK_LAUNCH("ns_gemm");
  return NS_OK;
}
