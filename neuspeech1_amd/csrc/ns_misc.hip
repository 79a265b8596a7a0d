// Byte-moving kernels of the path: MEG signal pack, token embedding, operand
// refresh (cast / transpose jobs), plus the ABI bookkeeping entry points.
#include "ns_common.h"
#include <string.h>

static thread_local char g_err[512] = "";
void ns_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" int ns_version(void) { return 3; }   // 2: ns_gemm_desc.seed_dev; 3: ns_zero_spans / ns_add_i32
extern "C" const char* ns_last_error(void) { return g_err; }

namespace {

// (B, ch, T) fp32 channel-major  ->  (B, T+2, Cp) fp16 token-major with a zero halo row at both ends and zero channel
// padding.  The only full-size read of the batch: 16-B lane loads along T (a wave instruction covers 4 channel rows x
// 256 B; T % 4 == 0 keeps every row 16-B aligned, else the scalar form below), transposed through an LDS tile stored
// [time][channel] so that each lane leaves with ONE ds_read_b128 and ONE 16-B store along the channel axis.
// Block = 256 threads handles 64 time steps x 64 channels.
template <bool VEC4>
__global__ __launch_bounds__(256) void signal_pack_kernel(const float* __restrict__ x, half_t* __restrict__ out, int ch,
                                                           int T, int Cp) {
  // tile [time 64][channel 64] fp16, 128-B rows; 16-B chunk c of time row t sits at chunk c ^ ((t >> 2) & 7).  Round 3 stored single
  // halfs into 144-B rows: the 16 time groups of a wave instruction fell on two banks (SQ_LDS_BANK_CONFLICT 6x the LDS instruction
  // cycles).  Now a thread owns channel PAIRS and writes dwords: 16 banks per 32 lanes (2-way, free for ds_write_b32), and the
  // ds_read_b128 of the way out is conflict-free (the four rows of a lane group differ in bit 5 of the bank or in the chunk).
  __shared__ __attribute__((aligned(16))) unsigned char tile[64 * 128];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const float* xb = x + (size_t)b * ch * T;
  if (VEC4) {
    const int cl = threadIdx.x >> 4, k = threadIdx.x & 15, tl4 = 4 * k;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + 2 * cl + (i & 1) + 32 * (i >> 1), t = t0 + tl4;
      v[i] = (c < ch && t < T) ? *(const float4*)(xb + (size_t)c * T + t) : make_float4(0.f, 0.f, 0.f, 0.f);   // T % 4 == 0: t + 3 < T
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float lo[4] = {v[2 * pr].x, v[2 * pr].y, v[2 * pr].z, v[2 * pr].w};
      const float hi[4] = {v[2 * pr + 1].x, v[2 * pr + 1].y, v[2 * pr + 1].z, v[2 * pr + 1].w};
      const int chunk = (cl >> 2) + 4 * pr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2v h = {(half_t)lo[e], (half_t)hi[e]};
        *(half2v*)(tile + (tl4 + e) * 128 + ((chunk ^ (k & 7)) << 4) + (cl & 3) * 4) = h;
      }
    }
  } else {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int cc_ = w * 16 + i, c = c0 + cc_, t = t0 + lane;
      float v = 0.f;
      if (c < ch && t < T) v = xb[(size_t)c * T + t];
      *(half_t*)(tile + lane * 128 + (((cc_ >> 3) ^ ((lane >> 2) & 7)) << 4) + (cc_ & 7) * 2) = (half_t)v;
    }
  }
  __syncthreads();
  half_t* ob = out + (size_t)b * (T + 2) * Cp;
  const int cc = (threadIdx.x & 7) * 8;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int tl = (threadIdx.x >> 3) + 32 * it;
    const int t = t0 + tl;
    // (Cp is a multiple of 8, not of the tile's 64: the last channel block writes only its pieces below Cp)
    if (t < T && c0 + cc < Cp) *(half8*)(ob + (size_t)(t + 1) * Cp + c0 + cc) = *(const half8*)(tile + tl * 128 + (((cc >> 3) ^ ((tl >> 2) & 7)) << 4));
  }
  // halo rows
  if (blockIdx.x == 0 && threadIdx.x < 64 && c0 + (int)threadIdx.x < Cp) {
    ob[c0 + threadIdx.x] = (half_t)0.f;
    ob[(size_t)(T + 1) * Cp + c0 + threadIdx.x] = (half_t)0.f;
  }
}

// Raw recordings -> the same (B, T+2, Cp) fp16 layout, with the reader's channel / time zero padding and the
// collator's f64 -> f32 rounding applied on the way (see ns_feed_pack in the header).  Same tiling as
// signal_pack_kernel; a block that lies wholly in a recording's padding issues no loads at all.
template <typename SRC>
__device__ __forceinline__ float feed_load(const void* src, long long idx) { return (float)((const SRC*)src)[idx]; }

__global__ __launch_bounds__(256) void feed_pack_kernel(const ns_feed_item* __restrict__ items, half_t* __restrict__ out,
                                                         float* __restrict__ x32, int ch, int T, int Cp) {
  __shared__ half_t tile[64][66];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const ns_feed_item it = items[b];
  const int rows = min(it.rows, ch), n = min(it.n, T);
  const bool live = c0 < rows && t0 < n;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + w * 16 + i, t = t0 + lane;
    float v = 0.f;
    if (live && c < rows && t < n) {
      const long long idx = (long long)c * it.ld + t;
      v = it.dtype == NS_FEED_F64 ? feed_load<double>(it.src, idx)
          : it.dtype == NS_FEED_F32 ? feed_load<float>(it.src, idx) : feed_load<half_t>(it.src, idx);
    }
    tile[w * 16 + i][lane] = (half_t)v;
    if (x32 && c < ch && t < T) x32[((size_t)b * ch + c) * T + t] = v;
  }
  __syncthreads();
  half_t* ob = out + (size_t)b * (T + 2) * Cp;
  const int cc = (threadIdx.x & 7) * 8;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int tl = (threadIdx.x >> 3) + 32 * k;
    const int t = t0 + tl;
    if (t < T && c0 + cc < Cp) {
      half8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = tile[cc + e][tl];
      *(half8*)(ob + (size_t)(t + 1) * Cp + c0 + cc) = h;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64 && c0 + (int)threadIdx.x < Cp) {
    ob[c0 + threadIdx.x] = (half_t)0.f;
    ob[(size_t)(T + 1) * Cp + c0 + threadIdx.x] = (half_t)0.f;
  }
}

// h32[row] = E32[id[row]] + P32[pos0 + (row % L)]   (utils/load_model.py:645,668-673)
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E,
                                                     const float* __restrict__ P, float* __restrict__ h, int rows, int L,
                                                     int d, int pos0, const int* __restrict__ pos0_dev) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t id = ids[row];
  const int p0 = pos0_dev ? *pos0_dev : pos0;
  const float* e = E + (size_t)id * d;
  const float* p = P + (size_t)(p0 + row % L) * d;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 a = *(const float4*)(e + c);
    const float4 b = *(const float4*)(p + c);
    *(float4*)(h + (size_t)row * d + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// batched operand refresh: dst16[r][c] = scale * src32[...]; transpose optional
__global__ __launch_bounds__(256) void cast_jobs_kernel(const ns_cast_job* __restrict__ jobs) {
  const ns_cast_job j = jobs[blockIdx.y];
  __shared__ float tile[32][33];
  const int tiles_c = (j.cols + 31) / 32;
  const int tiles_r = (j.rows + 31) / 32;
  const float* src = (const float*)j.src;
  half_t* dst = (half_t*)j.dst;
  for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    if (!j.transpose) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < j.rows && c < j.cols)
          dst[(size_t)r * j.ld_dst + c] = (half_t)(j.scale * (j.colscale ? j.colscale[c] : 1.f) * src[(size_t)r * j.ld_src + c]);
      }
    } else {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < j.rows && c < j.cols) ? src[(size_t)r * j.ld_src + c] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;  // dst is (cols x rows)
        if (r < j.rows && c < j.cols) dst[(size_t)c * j.ld_dst + r] = (half_t)(j.scale * (j.colscale ? j.colscale[c] : 1.f) * tile[tx][ty + 8 * i]);
      }
    }
  }
}


// out16[map(row)][c] = round16(a16[row][c] * gelu'(pre16[row][c]))   (conv-stem backward seam)
__global__ __launch_bounds__(256) void dgelu_mul_kernel(const half_t* __restrict__ a, const half_t* __restrict__ pre,
                                                         half_t* __restrict__ out, ns_rowmap om, int rows, int cols,
                                                         int pre_is_grad) {
  const int cpr = cols / 8;
  const long long total = (long long)rows * cpr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int row = (int)(i / cpr), c = (int)(i - (long long)row * cpr) * 8;
    const half8 av = *(const half8*)(a + (long long)row * cols + c);
    const half8 pv = *(const half8*)(pre + (long long)row * cols + c);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)av[e] * (pre_is_grad ? (float)pv[e] : ns_gelu_grad((float)pv[e])));
    long long off;
    if (om.seg_rows > 0) { const int sg = row / om.seg_rows; off = (long long)sg * om.seg_stride + (long long)(row - sg * om.seg_rows) * om.ld; }
    else off = (long long)row * om.ld;
    *(half8*)(out + off + c) = o;
  }
}

// out32[c] += alpha * sum_rows a16[row][c]  (bias gradients).  512 blocks (every block adds into the same `cols` addresses: more blocks only add atomic contention --
// 3000 blocks ran 3x slower than 750); a thread owns 8 consecutive columns
// (16-B loads, 4 rows in flight), the block's row lanes are reduced through LDS, one fp32 atomic per column and block.
template <bool VEC8>
__global__ __launch_bounds__(256) void colsum_kernel(const half_t* __restrict__ a, float* __restrict__ out, int rows,
                                                      int cols, int ld, float alpha) {
  if (VEC8) {
    __shared__ float red[256 * 8];
    const int rpb = ((rows + gridDim.x - 1) / gridDim.x + 3) & ~3;     // rows per block
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    for (int cb = 0; cb < cols; cb += 512) {           // 64 threads x 8 columns per pass
      const int cgrp = threadIdx.x & 63, rl = threadIdx.x >> 6, c = cb + cgrp * 8;
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (c < cols) {
        int r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
          half8 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *(const half8*)(a + (long long)(r + 4 * u) * ld + c);
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
        }
        for (; r < r1; r += 4) {
          const half8 v = *(const half8*)(a + (long long)r * ld + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
      __syncthreads();
      if (rl == 0 && c < cols) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = red[cgrp * 8 + e] + red[(64 + cgrp) * 8 + e] + red[(128 + cgrp) * 8 + e] + red[(192 + cgrp) * 8 + e];
          atomicAdd(out + c + e, t * alpha);
        }
      }
    }
  } else {
    const int r0 = blockIdx.x * 256, r1 = min(rows, r0 + 256);
    for (int c = threadIdx.x * 2; c < cols; c += 512) {
      float s0 = 0.f, s1 = 0.f;
      for (int r = r0; r < r1; ++r) {
        const half2v v = *(const half2v*)(a + (long long)r * ld + c);
        s0 += (float)v[0]; s1 += (float)v[1];
      }
      atomicAdd(out + c, s0 * alpha);
      atomicAdd(out + c + 1, s1 * alpha);
    }
  }
}

// ---- AdaLoRA (finetune.py:205-208, peft AdaLoraLayer): Delta W = B diag(E) A * alpha/(r+1e-5)
// gradient fold: the weight-gradient GEMM produces dBf = dY^T u for the folded operand Bf = s * B * diag(E);
//   dB[n][k] += s * E[k] * dBf[n][k],   dE[k] += s * sum_n dBf[n][k] * B[n][k]
__global__ __launch_bounds__(256) void adalora_fold_kernel(const float* __restrict__ dBf, const float* __restrict__ Bm,
                                                           const float* __restrict__ E, float* __restrict__ dB,
                                                           float* __restrict__ dE, int N, int r, float s) {
  __shared__ float acc[32];
  if (threadIdx.x < 32) acc[threadIdx.x] = 0.f;
  __syncthreads();
  const int total = N * r;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i % r;
    const float g = dBf[i];
    dB[i] += s * E[k] * g;
    atomicAdd(&acc[k], s * g * Bm[i]);
  }
  __syncthreads();
  if (threadIdx.x < r) atomicAdd(dE + threadIdx.x, acc[threadIdx.x]);
}

// the same fold for a table of adapters in ONE launch (a layer's six projections: grid = (blocks per job, jobs)); the scratch
// gradient is zeroed behind the read, so the next backward finds it clear without a fill launch of its own
__global__ __launch_bounds__(256) void adalora_fold_jobs_kernel(const ns_adalora_fold_job* __restrict__ jobs) {
  const ns_adalora_fold_job j = jobs[blockIdx.y];
  __shared__ float acc[32];
  if (threadIdx.x < 32) acc[threadIdx.x] = 0.f;
  __syncthreads();
  const int total = j.N * j.r;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i % j.r;
    const float g = j.dBf[i];
    j.dBf[i] = 0.f;
    j.dB[i] += j.s * j.E[k] * g;
    atomicAdd(&acc[k], j.s * g * j.B[i]);
  }
  __syncthreads();
  if ((int)threadIdx.x < j.r) atomicAdd(j.dE + threadIdx.x, acc[threadIdx.x]);
}

// orthogonality regulariser of AdaLoRA: loss += w/num * ||P P^T - I||_F (lora_A, r x in) or ||P^T P - I||_F (lora_B,
// out x r); adds loss_scale * w/num * 2 (cov - I) P / ||cov - I||_F to the gradient; r <= 32.
// P is walked in 128-column chunks staged through LDS with coalesced loads (either storage order) and read back with 16-B LDS
// reads.  ORTH_SPLIT blocks share a matrix: the first launch adds each block's partial Gram matrix (four entries per thread)
// into a zeroed workspace, the second reads the complete matrix, forms the norm and gives each thread 16 gradient rows of one
// column.  (The first version let each thread walk two strided rows of P in global memory per Gram entry, one block per matrix:
// 2.5 ms per step for the 72 matrices of whisper-base; one block per matrix with LDS staging: 0.35 ms.)
constexpr int ORTH_CH = 128, ORTH_LD = ORTH_CH + 4, ORTH_SPLIT = 4;

__device__ __forceinline__ void orth_stage(const ns_orth_job& j, int r, int t0, float (*tile)[ORTH_LD]) {
  const int tid = threadIdx.x;
  if (j.is_b) {        // (len x r), element (k, t) at P[t * ld + k]
    const int k = tid & 31;
    for (int tt = tid >> 5; tt < ORTH_CH; tt += 8)
      tile[k][tt] = (k < r && t0 + tt < j.len) ? j.P[(size_t)(t0 + tt) * j.ld + k] : 0.f;
  } else {             // (r x len), element (k, t) at P[k * ld + t]
    const int tt = tid & (ORTH_CH - 1);
    for (int k = tid >> 7; k < NS_ORTH_MAX_R; k += 2)
      tile[k][tt] = (k < r && t0 + tt < j.len) ? j.P[(size_t)k * j.ld + t0 + tt] : 0.f;
  }
}

// partial Gram matrices: block (job, s) takes chunks s, s + ORTH_SPLIT, ...; thread (a, b0) owns entries (a, b0 .. b0 + 3)
__global__ __launch_bounds__(256) void orth_gram_kernel(const ns_orth_job* __restrict__ jobs, float* __restrict__ gram) {
  const ns_orth_job j = jobs[blockIdx.x];
  __shared__ __attribute__((aligned(16))) float tile[NS_ORTH_MAX_R][ORTH_LD];
  const int tid = threadIdx.x, r = min(j.r, NS_ORTH_MAX_R);
  const int a = tid >> 3, b0 = (tid & 7) * 4;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = blockIdx.y * ORTH_CH; t0 < j.len; t0 += ORTH_SPLIT * ORTH_CH) {
    __syncthreads();
    orth_stage(j, r, t0, tile);
    __syncthreads();
#pragma unroll 4
    for (int tt = 0; tt < ORTH_CH; tt += 4) {
      const float4 av = *(const float4*)&tile[a][tt];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = *(const float4*)&tile[b0 + q][tt];
        c[q] += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
      }
    }
  }
  float* g = gram + (size_t)blockIdx.x * NS_ORTH_MAX_R * NS_ORTH_MAX_R + a * NS_ORTH_MAX_R + b0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (a < r && b0 + q < r) atomicAdd(g + q, c[q]);
}

__global__ __launch_bounds__(256) void orth_grad_kernel(const ns_orth_job* __restrict__ jobs, const float* __restrict__ gram,
                                                        float weight_over_num, const float* __restrict__ loss_scale,
                                                        float* __restrict__ reg_out) {
  const ns_orth_job j = jobs[blockIdx.x];
  __shared__ __attribute__((aligned(16))) float cov[NS_ORTH_MAX_R][NS_ORTH_MAX_R + 4];
  __shared__ __attribute__((aligned(16))) float tile[NS_ORTH_MAX_R][ORTH_LD];
  __shared__ float red[4];
  const int tid = threadIdx.x, r = min(j.r, NS_ORTH_MAX_R);
  float sq = 0.f;
  for (int i = tid; i < NS_ORTH_MAX_R * NS_ORTH_MAX_R; i += 256) {
    const int a = i / NS_ORTH_MAX_R, b = i % NS_ORTH_MAX_R;
    const float v = (a < r && b < r) ? gram[(size_t)blockIdx.x * NS_ORTH_MAX_R * NS_ORTH_MAX_R + i] - (a == b ? 1.f : 0.f) : 0.f;
    cov[a][b] = v;
    sq += v * v;
  }
  sq = ns_wave_sum(sq);
  if ((tid & 63) == 0) red[tid >> 6] = sq;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  if (tid == 0 && blockIdx.y == 0) atomicAdd(reg_out, weight_over_num * nrm);
  if (nrm <= 0.f) return;
  const float coef = (loss_scale ? *loss_scale : 1.f) * weight_over_num * 2.f / nrm;
  // G(k, t) += coef * sum_b cov[k][b] P(b, t); thread (column tt, rows 16 kh .. 16 kh + 15)
  const int tt = tid & (ORTH_CH - 1), kh = tid >> 7;
  for (int t0 = blockIdx.y * ORTH_CH; t0 < j.len; t0 += ORTH_SPLIT * ORTH_CH) {
    __syncthreads();
    orth_stage(j, r, t0, tile);
    __syncthreads();
    if (t0 + tt >= j.len) continue;
    float pc[NS_ORTH_MAX_R];
#pragma unroll
    for (int b = 0; b < NS_ORTH_MAX_R; ++b) pc[b] = tile[b][tt];
#pragma unroll 2
    for (int kk = 0; kk < 16; ++kk) {
      const int k = 16 * kh + kk;
      if (k >= r) break;
      float sacc = 0.f;
#pragma unroll
      for (int b = 0; b < NS_ORTH_MAX_R; b += 4) {
        const float4 cv = *(const float4*)&cov[k][b];
        sacc += cv.x * pc[b] + cv.y * pc[b + 1] + cv.z * pc[b + 2] + cv.w * pc[b + 3];
      }
      float* g = j.is_b ? j.G + (size_t)(t0 + tt) * j.ld + k : j.G + (size_t)k * j.ld + t0 + tt;
      *g += coef * sacc;
    }
  }
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// up to NS_ZERO_MAX_SPANS buffers cleared by ONE launch (blockIdx.y = span): 16-B lane stores, a dword tail
struct zero_spans_arg { ns_span s[NS_ZERO_MAX_SPANS]; };
__global__ __launch_bounds__(256) void zero_spans_kernel(const zero_spans_arg a) {
  const ns_span sp = a.s[blockIdx.y];
  const size_t n16 = sp.bytes >> 4;
  uint4* const p16 = (uint4*)sp.p;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = z;
  if (blockIdx.x == 0) {
    uint32_t* const p4 = (uint32_t*)sp.p;
    for (size_t i = (n16 << 2) + threadIdx.x; i < (sp.bytes >> 2); i += blockDim.x) p4[i] = 0u;
  }
}
__global__ void add_i32_kernel(int32_t* p, int32_t n, int32_t v) { if ((int)threadIdx.x < n) p[threadIdx.x] += v; }

}  // namespace

extern "C" int ns_signal_pack(const float* x, void* out16, int B, int ch, int T, int Cp, void* stream) {
  NS_CHECK_ARG(x && out16, "ns_signal_pack: null pointer");
  NS_CHECK_ARG(B > 0 && ch > 0 && T > 0 && Cp >= ch && Cp % 8 == 0, "ns_signal_pack: bad shape B=%d ch=%d T=%d Cp=%d", B, ch, T, Cp);
  dim3 grid((T + 63) / 64, (Cp + 63) / 64, B);
  if (T % 4 == 0 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(signal_pack_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (half_t*)out16, ch, T, Cp);
  else
    hipLaunchKernelGGL(signal_pack_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (half_t*)out16, ch, T, Cp);
  NS_CHECK_LAUNCH("ns_signal_pack");
  return NS_OK;
}

extern "C" int ns_feed_pack(const ns_feed_item* items_dev, int B, int ch, int T, int Cp, void* out16, float* x32,
                            void* stream) {
  NS_CHECK_ARG(items_dev && out16, "ns_feed_pack: null pointer");
  NS_CHECK_ARG(B > 0 && B <= 65535 && ch > 0 && T > 0 && Cp >= ch && Cp % 8 == 0,
               "ns_feed_pack: bad shape B=%d ch=%d T=%d Cp=%d", B, ch, T, Cp);
  dim3 grid((T + 63) / 64, (Cp + 63) / 64, B);
  hipLaunchKernelGGL(feed_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, items_dev, (half_t*)out16, x32, ch, T, Cp);
  NS_CHECK_LAUNCH("ns_feed_pack");
  return NS_OK;
}

extern "C" int ns_embed_pos(const int64_t* ids, const float* E32, const float* P32, float* h32, int rows, int L, int d,
                            int pos0, const int* pos0_dev, void* stream) {
  NS_CHECK_ARG(ids && E32 && P32 && h32, "ns_embed_pos: null pointer");
  NS_CHECK_ARG(rows > 0 && L > 0 && d % 4 == 0, "ns_embed_pos: bad shape");
  hipLaunchKernelGGL(embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, E32, P32, h32, rows, L,
                     d, pos0, pos0_dev);
  NS_CHECK_LAUNCH("ns_embed_pos");
  return NS_OK;
}

extern "C" int ns_cast_jobs(const ns_cast_job* jobs_dev, int njobs, void* stream) {
  NS_CHECK_ARG(jobs_dev && njobs > 0, "ns_cast_jobs: bad arguments");
  hipLaunchKernelGGL(cast_jobs_kernel, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  NS_CHECK_LAUNCH("ns_cast_jobs");
  return NS_OK;
}

extern "C" int ns_dgelu_mul(const void* a16, const void* pre16, void* out16, const ns_rowmap* out_map, int rows, int cols,
                            int pre_is_grad, void* stream) {
  NS_CHECK_ARG(a16 && pre16 && out16 && out_map && rows > 0 && cols > 0 && cols % 8 == 0 && out_map->ld % 8 == 0,
               "ns_dgelu_mul: bad arguments");
  long long total = (long long)rows * (cols / 8);
  int nb = (int)((total + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(dgelu_mul_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const half_t*)a16,
                     (const half_t*)pre16, (half_t*)out16, *out_map, rows, cols, pre_is_grad);
  NS_CHECK_LAUNCH("ns_dgelu_mul");
  return NS_OK;
}

extern "C" int ns_colsum(const void* a16, float* out32, int rows, int cols, int ld, float alpha, void* stream) {
  NS_CHECK_ARG(a16 && out32 && rows > 0 && cols > 0 && cols % 2 == 0 && ld % 2 == 0, "ns_colsum: bad arguments");
  if (cols % 8 == 0 && ld % 8 == 0)
    hipLaunchKernelGGL(colsum_kernel<true>, dim3(rows >= 8192 ? 512 : (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const half_t*)a16,
                       out32, rows, cols, ld, alpha);
  else
    hipLaunchKernelGGL(colsum_kernel<false>, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const half_t*)a16,
                       out32, rows, cols, ld, alpha);
  NS_CHECK_LAUNCH("ns_colsum");
  return NS_OK;
}

extern "C" int ns_adalora_fold_grads(const float* dBf, const float* B, const float* E, float* dB, float* dE, int N, int r,
                                     float s, void* stream) {
  NS_CHECK_ARG(dBf && B && E && dB && dE && N > 0 && r > 0 && r <= 32, "ns_adalora_fold_grads: bad arguments");
  int nb = (N * r + 255) / 256;
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(adalora_fold_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dBf, B, E, dB, dE, N, r, s);
  NS_CHECK_LAUNCH("ns_adalora_fold_grads");
  return NS_OK;
}

extern "C" int ns_adalora_fold_jobs(const ns_adalora_fold_job* jobs_dev, int njobs, void* stream) {
  NS_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535, "ns_adalora_fold_jobs: bad arguments");
  hipLaunchKernelGGL(adalora_fold_jobs_kernel, dim3(16, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  NS_CHECK_LAUNCH("ns_adalora_fold_jobs");
  return NS_OK;
}

extern "C" size_t ns_orth_reg_workspace_bytes(int njobs) { return (size_t)njobs * NS_ORTH_MAX_R * NS_ORTH_MAX_R * sizeof(float); }

extern "C" int ns_orth_reg(const ns_orth_job* jobs_dev, int njobs, float weight_over_num, const float* loss_scale_dev,
                           float* reg_out_dev, void* workspace, size_t workspace_bytes, void* stream) {
  NS_CHECK_ARG(jobs_dev && njobs > 0 && reg_out_dev, "ns_orth_reg: bad arguments");
  NS_CHECK_ARG(workspace && workspace_bytes >= ns_orth_reg_workspace_bytes(njobs),
               "ns_orth_reg: workspace of %zu bytes is smaller than ns_orth_reg_workspace_bytes(%d)", workspace_bytes, njobs);
  hipStream_t st = (hipStream_t)stream;
  // cleared by a KERNEL, not by hipMemsetAsync: captured in a hipGraph (engine.train_step) the memset node was not reliably
  // ordered against the kernel nodes around it under back-to-back replays -- the AdaLoRA step reported a garbage / inf
  // regulariser value every few dozen replays (tools/soak_train.py ... adalora), never with eager launches
  const size_t nwords = ns_orth_reg_workspace_bytes(njobs) / 4;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((nwords + 1023) / 1024)), dim3(256), 0, st, (uint32_t*)workspace, 0u, nwords);
  hipLaunchKernelGGL(orth_gram_kernel, dim3(njobs, ORTH_SPLIT), dim3(256), 0, st, jobs_dev, (float*)workspace);
  hipLaunchKernelGGL(orth_grad_kernel, dim3(njobs, ORTH_SPLIT), dim3(256), 0, st, jobs_dev, (const float*)workspace, weight_over_num,
                     loss_scale_dev, reg_out_dev);
  NS_CHECK_LAUNCH("ns_orth_reg");
  return NS_OK;
}

extern "C" int ns_zero_spans(const ns_span* spans, int n, void* stream) {
  NS_CHECK_ARG(spans && n > 0 && n <= NS_ZERO_MAX_SPANS, "ns_zero_spans: 1..%d spans (got %d)", NS_ZERO_MAX_SPANS, n);
  zero_spans_arg a;
  size_t most = 0;
  for (int i = 0; i < NS_ZERO_MAX_SPANS; ++i) {
    a.s[i] = i < n ? spans[i] : ns_span{nullptr, 0};
    if (i < n) {
      NS_CHECK_ARG(spans[i].p && ((uintptr_t)spans[i].p & 15) == 0 && (spans[i].bytes & 3) == 0,
                   "ns_zero_spans: span %d must be 16-byte aligned with a multiple of 4 bytes", i);
      most = spans[i].bytes > most ? spans[i].bytes : most;
    }
  }
  if (most == 0) return NS_OK;
  const size_t blocks = (most / 16 + 1023) / 1024;      // four 16-B stores per thread and pass
  hipLaunchKernelGGL(zero_spans_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks)), n), dim3(256), 0,
                     (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ns_zero_spans");
  return NS_OK;
}

extern "C" int ns_add_i32(int32_t* counter_dev, int32_t n, int32_t v, void* stream) {
  NS_CHECK_ARG(counter_dev && n >= 1 && n <= 64, "ns_add_i32: null pointer or n=%d outside 1 .. 64", n);
  hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter_dev, n, v);
  NS_CHECK_LAUNCH("ns_add_i32");
  return NS_OK;
}
