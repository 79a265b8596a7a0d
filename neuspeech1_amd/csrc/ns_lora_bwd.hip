// LoRA backward, the two products that stream dy:   du_g = alpha_du * dy_g sB_g      (M x r)
//                                                    dB_g += alpha_db[g] * dy_g^T u_g (N x r, reduction over M)
// in ONE pass over dy (peft lora.Linear backward of `y += scale * B (A drop(x))`, finetune.py:187-212; the engine used
// to run them as a skinny NT GEMM and a TN weight-gradient GEMM that each read dy).  HBM-bound: dy is read once
// (M x G*N x 2 B), everything else is small.
//
// A workgroup (512 threads, 8 waves, one per CU) owns a contiguous range of 64-row chunks and walks the items
// (chunk, group, 256-column slice of dy): the slice (64 x 256 fp16 = 32 KiB) is staged row-major into LDS by 16-B
// register loads one item ahead (XOR-swizzled 16-B chunks: conflict-free for the row reads AND the transposed reads), and
//   du:  wave w multiplies rows 32*(w&1).. by the 64 columns 64*(w>>1).. of the slice against sB^T fragments loaded from
//        global memory (L2-resident) -- partial sums over the four column quarters and over the slices of a group stay in
//        registers and are summed through LDS once per (chunk, group);
//   dB:  wave w owns the 32 dy columns 32*w.. of every slice: dy^T fragments by ds_read_b64_tr_b16 from the same image,
//        u^T fragments by transposed reads of the chunk's u rows; its (N/256) x 32 x 32 accumulators persist over the whole
//        row range and leave by fp32 atomics (128 contiguous bytes per half wave) at the end.
#include <mutex>
#include "ns_common.h"

namespace {

constexpr int RC = 64, SC = 256, NTH = 512;
constexpr int DY_BYTES = RC * SC * 2;        // 32 KiB
constexpr int U_STRIDE = 64;                 // bytes per u row image (32 halfs)
constexpr int U_BYTES = RC * U_STRIDE;       // 4 KiB
constexpr int RED_BYTES = 8 * 32 * 32 * 4;   // du partials of the 8 waves
constexpr int LDS_BYTES = 2 * DY_BYTES + 2 * U_BYTES + RED_BYTES;

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_s4;

__device__ __forceinline__ int dy_off(int row, int chunk) { return row * 512 + ((chunk ^ (row & 31)) << 4); }

// A[row = c0 + (lane & 31)][k = m0 + 8*(lane>>5) + i] of a 32x32x16 MFMA from a row-major [m][c] image: two 4-row
// transposed reads (see ns_gemm_tn.hip tr_frag); `off(row, col)` is the byte offset of element (row, col)
template <class Off>
__device__ __forceinline__ half8 tr_frag(const char* tile, Off off, int m0, int c0, int lane) {
  const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
  const int row = m0 + 8 * (g >> 1) + q, col = c0 + 16 * (g & 1) + 4 * p;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + off(row, col)));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + off(row + 4, col)));
  const short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8, r);
}

template <int G, int NSUB>
__global__ __launch_bounds__(NTH) void lora_bwd_dudb_kernel(const ns_lora_bwd_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dyb = smem;
  char* const ub = smem + 2 * DY_BYTES;
  float* const red = (float*)(smem + 2 * DY_BYTES + 2 * U_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  const int nchunks = (p.M + RC - 1) / RC;
  const int per = (nchunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = blockIdx.x * per, c_hi = min(nchunks, c_lo + per);

  f32x16 accB[G][NSUB];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[g][s][r] = 0.f;
  f32x16 accU;
#pragma unroll
  for (int r = 0; r < 16; ++r) accU[r] = 0.f;

  // staging registers of the NEXT item
  uint4 dyr[4];
  half8 sbr[4];
  uint4 ur;
  const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
  const int mt = wave & 1, kq = wave >> 1;

  auto load_item = [&](int c, int g, int s) __attribute__((always_inline)) {
    const int row0 = c * RC;
    const half_t* dy = (const half_t*)p.dy + (long long)g * p.N + s * SC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + NTH * i, row = id >> 5, ch = id & 31;
      dyr[i] = row0 + row < p.M ? *(const uint4*)(dy + (long long)(row0 + row) * p.ldy + ch * 8) : make_uint4(0, 0, 0, 0);
    }
    if (s == 0 && tid < 256) {      // the chunk's u rows of this group (64 x r), one 16-B piece per thread
      const int row = tid >> 2, cc = tid & 3;
      const bool ok = row0 + row < p.M && cc * 8 < p.r;
      ur = ok ? *(const uint4*)((const half_t*)p.u + (long long)(row0 + row) * p.ldu + g * p.r + cc * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  // sB^T fragments of this wave's 64 slice columns: B[k = column][j]: lane (j = lr, half lh) holds 8 consecutive columns.
  // Loaded for the NEXT item right after the current item's du MFMAs have consumed the registers (L2-resident operand).
  auto load_sb = [&](int g, int s) __attribute__((always_inline)) {
    const half_t* sbt = (const half_t*)p.sBT[g] + (long long)lr * p.N + s * SC + 64 * kq + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sbr[ks] = lr < p.r ? *(const half8*)(sbt + 16 * ks) : hz;
  };
  auto store_item = [&](int it, int c, int g, int s) __attribute__((always_inline)) {
    char* const buf = dyb + (it & 1) * DY_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + NTH * i, row = id >> 5, ch = id & 31;
      *(uint4*)(buf + dy_off(row, ch)) = dyr[i];
    }
    if (s == 0 && tid < 256) {
      const int slot = (c * G + g) & 1;
      *(uint4*)(ub + slot * U_BYTES + (tid >> 2) * U_STRIDE + (tid & 3) * 16) = ur;
    }
  };
  if (c_lo < c_hi) {
    load_item(c_lo, 0, 0);
    load_sb(0, 0);
    store_item(0, c_lo, 0, 0);
  }
  __syncthreads();
  int it = 0;
  for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s = 0; s < NSUB; ++s, ++it) {
        // the next item: (c, g, s + 1) -> (c, g + 1, 0) -> (c + 1, 0, 0)
        const int ns = s + 1 < NSUB ? s + 1 : 0;
        const int ng = s + 1 < NSUB ? g : (g + 1 < G ? g + 1 : 0);
        const int nc = (s + 1 < NSUB || g + 1 < G) ? c : c + 1;
        const bool more = nc < c_hi;
        if (more) load_item(nc, ng, ns);
        const char* const buf = dyb + (it & 1) * DY_BYTES;
        const char* const uimg = ub + ((c * G + g) & 1) * U_BYTES;
        // ---- du partial: rows 32*mt.., slice columns 64*kq.. (4 k-steps of 16)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const half8 a = *(const half8*)(buf + dy_off(32 * mt + lr, 8 * kq + 2 * ks + lh));
          accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, sbr[ks], accU, 0, 0, 0);
        }
        if (more) load_sb(ng, ns);
        // ---- dB: dy columns 32*wave.. of this slice, reduction over the chunk's 64 rows
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const half8 a = tr_frag(buf, [](int row, int col) { return dy_off(row, col >> 3) + (col & 7) * 2; }, 16 * ks, 32 * wave, lane);
          const half8 b = tr_frag(uimg, [](int row, int col) { return row * U_STRIDE + col * 2; }, 16 * ks, 0, lane);
          accB[g][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accB[g][s], 0, 0, 0);
        }
        if (s == NSUB - 1) {
          // ---- du of (chunk c, group g) is complete: sum the four column-quarter partials through LDS, scale, store
          float* const mine = red + wave * 1024;
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = accU[r];
#pragma unroll
          for (int r = 0; r < 16; ++r) accU[r] = 0.f;
          __syncthreads();
          const int row0 = c * RC;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int id = tid + NTH * i, m = id >> 5, j = id & 31;     // m: row of the chunk, j: bottleneck column
            const int mtile = m >> 5, ml = m & 31;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += red[(mtile + 2 * q) * 1024 + ml * 32 + j];
            if (row0 + m < p.M && j < p.r)
              ((half_t*)p.du)[(long long)(row0 + m) * p.lddu + g * p.r + j] = (half_t)(v * p.alpha_du);
          }
        }
        if (more) store_item(it + 1, nc, ng, ns);
        __syncthreads();
      }
  }

  // ---- dB leaves by fp32 atomics: accumulator tile (g, s): rows n = 256*s + 32*wave + crow(r), column j = lr
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      if (lr >= p.r) continue;
      float* const dst = p.dB[g] + (long long)(s * SC + 32 * wave) * p.lddb + lr;
      const float al = p.alpha_db[g];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;
        atomicAdd(dst + (long long)n * p.lddb, accB[g][s][r] * al);
      }
    }
}

template <int G, int NSUB>
void launch(const ns_lora_bwd_desc* d, int grid, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [&] {
    hipFuncSetAttribute((const void*)lora_bwd_dudb_kernel<G, NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  hipLaunchKernelGGL((lora_bwd_dudb_kernel<G, NSUB>), dim3(grid), dim3(NTH), LDS_BYTES, st, *d);
}

}  // namespace

extern "C" int ns_lora_bwd_supported(int N, int r, int G) {
  if (N <= 0 || N % SC != 0 || (r != 16 && r != 32) || (G != 1 && G != 3)) return 0;
  const int nsub = N / SC;
  if (G == 1) return nsub == 1 || nsub == 2 || nsub == 5 || nsub == 8;
  return nsub == 1 || nsub == 2;
}

extern "C" int ns_lora_bwd_dudb(const ns_lora_bwd_desc* d, void* stream) {
  NS_CHECK_ARG(d && d->dy && d->u && d->du, "ns_lora_bwd_dudb: null pointer");
  NS_CHECK_ARG(d->M > 0 && ns_lora_bwd_supported(d->N, d->r, d->G),
               "ns_lora_bwd_dudb: unsupported shape M=%d N=%d r=%d G=%d (N %% 256 == 0, r in {16, 32}, G in {1, 3})", d->M, d->N,
               d->r, d->G);
  for (int g = 0; g < d->G; ++g) NS_CHECK_ARG(d->sBT[g] && d->dB[g], "ns_lora_bwd_dudb: group %d operands missing", g);
  NS_CHECK_ARG(d->ldy % 8 == 0 && d->ldu % 8 == 0 && d->ldy >= d->G * d->N && d->ldu >= d->G * d->r && d->lddu >= d->G * d->r &&
                   d->lddb >= d->r,
               "ns_lora_bwd_dudb: bad strides");
  const int nchunks = (d->M + RC - 1) / RC;
  int grid = d->splits > 0 ? d->splits : 256;
  if (grid > nchunks) grid = nchunks;
  hipStream_t st = (hipStream_t)stream;
  const int nsub = d->N / SC;
#define NS_LB(G_, S_) if (d->G == G_ && nsub == S_) { launch<G_, S_>(d, grid, st); }
  NS_LB(1, 1) else NS_LB(1, 2) else NS_LB(1, 5) else NS_LB(1, 8) else NS_LB(3, 1) else NS_LB(3, 2)
#undef NS_LB
  NS_CHECK_LAUNCH("ns_lora_bwd_dudb");
  return NS_OK;
}
