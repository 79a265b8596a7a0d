// Fused cross-entropy over fp16 logits: loss (fp32, mean over labels != -100)
// and dlogits (fp16) in one pass structure.  HBM-bound: one row (V*2 B) is read
// three times (second and third from L2) and written once.
// Replaces utils/load_model.py:1049-1054 (CrossEntropyLoss on proj_out logits)
// and its autograd backward.
#include "ns_common.h"

namespace {

// a label outside [0, V) other than the ignore index -100 is an error in the reference (torch's CrossEntropyLoss
// asserts; the reference collator only warns, utils/data_utils.py:201): here it is ignored like -100, never read
__global__ void count_valid_kernel(const int64_t* __restrict__ labels, int rows, int V, int* __restrict__ nvalid) {
  __shared__ int sh[4];
  int c = 0;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) c += labels[i] >= 0 && labels[i] < V;
  c = (int)ns_wave_sum((float)c);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) *nvalid = sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? ns_wave_max(v) : ns_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

__global__ __launch_bounds__(256) void ce_kernel(const half_t* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  int V, int ldv, float* __restrict__ row_loss,
                                                  half_t* __restrict__ dlogits, const int* __restrict__ nvalid,
                                                  const float* __restrict__ loss_scale) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  const int64_t lab = labels[row];
  const half_t* lr = logits + (size_t)row * ldv;
  half_t* dr = dlogits ? dlogits + (size_t)row * ldv : nullptr;
  const int nchunks = ldv / 8;
  if (lab < 0 || lab >= V) {
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    if (dr) {
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = threadIdx.x; c < nchunks; c += 256) *(half8*)(dr + c * 8) = z;
    }
    return;
  }
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < nchunks; c += 256) {
    const half8 h = *(const half8*)(lr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c * 8 + e < V) mx = fmaxf(mx, (float)h[e]);
  }
  mx = block_reduce(mx, true, sh);
  float se = 0.f;
  for (int c = threadIdx.x; c < nchunks; c += 256) {
    const half8 h = *(const half8*)(lr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c * 8 + e < V) se += __expf((float)h[e] - mx);
  }
  se = block_reduce(se, false, sh);
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0) row_loss[row] = lse - (float)lr[lab];
  __syncthreads();  // dlogits may alias logits: lr[lab] must be read before any overwrite
  if (dr) {
    const float gs = (loss_scale ? *loss_scale : 1.f) / (float)max(*nvalid, 1);
    const float inv = 1.f / se;
    for (int c = threadIdx.x; c < nchunks; c += 256) {
      const half8 h = *(const half8*)(lr + c * 8);
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = c * 8 + e;
        float p = col < V ? __expf((float)h[e] - mx) * inv : 0.f;
        if (col == lab) p -= 1.f;
        o[e] = (half_t)(p * gs);
      }
      *(half8*)(dr + c * 8) = o;
    }
  }
}

__global__ void loss_reduce_kernel(const float* __restrict__ row_loss, int rows, const int* __restrict__ nvalid,
                                   float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) s += row_loss[i];
  s = block_reduce(s, false, sh);
  if (threadIdx.x == 0) *loss = s / (float)max(*nvalid, 1);
}

// argmax over the first V columns of each row (teacher-forced eval, evaluation.py:394-399; greedy step)
__global__ __launch_bounds__(256) void argmax_kernel(const half_t* __restrict__ logits, int V, int ldv,
                                                      int64_t* __restrict__ out) {
  __shared__ float shv[4];
  __shared__ int shi[4];
  const half_t* lr = logits + (size_t)blockIdx.x * ldv;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += 256) {
    const float v = (float)lr[c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = best; shi[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (shv[i] > best || (shv[i] == best && shi[i] < bi)) { best = shv[i]; bi = shi[i]; }
    out[blockIdx.x] = bi;
  }
}

}  // namespace

extern "C" int ns_cross_entropy(const void* logits16, const int64_t* labels, int rows, int V, int ldv, float* row_loss,
                                void* dlogits16, int* nvalid_dev, const float* loss_scale_dev, float* loss_dev,
                                void* stream) {
  NS_CHECK_ARG(logits16 && labels && row_loss && nvalid_dev && loss_dev, "ns_cross_entropy: null pointer");
  NS_CHECK_ARG(rows > 0 && V > 0 && ldv >= V && ldv % 8 == 0, "ns_cross_entropy: bad shape rows=%d V=%d ldv=%d", rows, V, ldv);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, st, labels, rows, V, nvalid_dev);
  hipLaunchKernelGGL(ce_kernel, dim3(rows), dim3(256), 0, st, (const half_t*)logits16, labels, V, ldv, row_loss,
                     (half_t*)dlogits16, nvalid_dev, loss_scale_dev);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, rows, nvalid_dev, loss_dev);
  NS_CHECK_LAUNCH("ns_cross_entropy");
  return NS_OK;
}

extern "C" int ns_argmax_rows(const void* logits16, int rows, int V, int ldv, int64_t* out, void* stream) {
  NS_CHECK_ARG(logits16 && out && rows > 0 && V > 0 && ldv >= V, "ns_argmax_rows: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const half_t*)logits16, V, ldv, out);
  NS_CHECK_LAUNCH("ns_argmax_rows");
  return NS_OK;
}
