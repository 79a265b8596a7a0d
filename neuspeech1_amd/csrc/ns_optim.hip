// Fused trainable-parameter update: grad-norm + inf check, GradScaler unscale,
// global-norm clip, AdamW and the linear warmup/decay schedule, all on-device
// and sync-free (graph-capturable).  Replaces the HF Trainer inner-loop tail
// configured at finetune.py:231-253 (fp16 GradScaler, adamw_torch, max_grad_norm
// 1.0, linear schedule).  Trainables live in ONE flat fp32 buffer (LoRA A/B and
// the conv weights/biases), which is also what the RCCL all-reduce moves.
#include "ns_common.h"

namespace {

__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ g, size_t n,
                                                            float* __restrict__ partial, int* __restrict__ bad) {
  __shared__ float sh[4];
  __shared__ int shb[4];
  float s = 0.f;
  int b = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = g[i];
    s += v * v;
    b |= !isfinite(v);
  }
  s = ns_wave_sum(s);
  b = __any(b);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shb[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    bad[blockIdx.x] = shb[0] | shb[1] | shb[2] | shb[3];
  }
}

__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, const int* __restrict__ bad,
                                                          int nblocks, float* __restrict__ norm2,
                                                          int* __restrict__ found_inf) {
  __shared__ float sh[4];
  __shared__ int shb[4];
  float s = 0.f;
  int b = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s += partial[i]; b |= bad[i]; }
  s = ns_wave_sum(s);
  b = __any(b);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shb[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const int fb = shb[0] | shb[1] | shb[2] | shb[3];
    *norm2 = t;
    *found_inf = fb | !isfinite(t);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, size_t n,
                                                     ns_adamw_cfg c, const int* __restrict__ step_dev,
                                                     const float* __restrict__ norm2, const int* __restrict__ found_inf,
                                                     const float* __restrict__ loss_scale) {
  if (*found_inf) return;  // GradScaler semantics: skip the whole step
  const int t = *step_dev;  // completed optimizer steps so far
  // linear warmup then linear decay (HF get_linear_schedule_with_warmup), lr for step t
  float lam;
  if (c.total_steps <= 0) lam = 1.f;
  else if (t < c.warmup_steps) lam = (float)t / (float)max(1, c.warmup_steps);
  else lam = fmaxf(0.f, (float)(c.total_steps - t) / (float)max(1, c.total_steps - c.warmup_steps));
  const float lr = c.lr * lam;
  const float inv_scale = 1.f / (loss_scale ? *loss_scale : 1.f);
  const float gnorm = sqrtf(*norm2) * inv_scale;
  const float clip = c.max_grad_norm > 0.f ? fminf(1.f, c.max_grad_norm / (gnorm + 1e-6f)) : 1.f;
  const float gmul = inv_scale * clip;
  const float tt = (float)(t + 1);
  const float bc1 = 1.f - powf(c.beta1, tt);
  const float bc2s = sqrtf(1.f - powf(c.beta2, tt));
  const float step_size = lr / bc1;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * gmul;
    float pi = p[i] * (1.f - lr * c.weight_decay);
    const float mi = c.beta1 * m[i] + (1.f - c.beta1) * gi;
    const float vi = c.beta2 * v[i] + (1.f - c.beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2s + c.eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// after adamw: bump step (if not skipped) and update the dynamic loss scale
__global__ void scaler_update_kernel(int* step_dev, float* loss_scale, int* growth_tracker, const int* found_inf,
                                     float growth, float backoff, int interval) {
  if (*found_inf) {
    if (loss_scale) *loss_scale *= backoff;
    if (growth_tracker) *growth_tracker = 0;
  } else {
    *step_dev += 1;
    if (loss_scale && growth_tracker) {
      const int gt = *growth_tracker + 1;
      if (gt >= interval) { *loss_scale *= growth; *growth_tracker = 0; }
      else *growth_tracker = gt;
    }
  }
}

}  // namespace

extern "C" size_t ns_grad_norm_workspace_bytes(void) { return 1024 * (sizeof(float) + sizeof(int)); }

extern "C" int ns_grad_norm(const float* g, size_t n, void* workspace, float* norm2_dev, int* found_inf_dev,
                            void* stream) {
  NS_CHECK_ARG(g && workspace && norm2_dev && found_inf_dev && n > 0, "ns_grad_norm: bad arguments");
  int nb = (int)((n + 4095) / 4096);
  if (nb > 1024) nb = 1024;
  float* partial = (float*)workspace;
  int* bad = (int*)(partial + 1024);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(norm_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial, bad);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, partial, bad, nb, norm2_dev, found_inf_dev);
  NS_CHECK_LAUNCH("ns_grad_norm");
  return NS_OK;
}

extern "C" int ns_adamw_step(float* p, const float* g, float* m, float* v, size_t n, const ns_adamw_cfg* cfg,
                             int* step_dev, const float* norm2_dev, const int* found_inf_dev, float* loss_scale_dev,
                             int* growth_tracker_dev, void* stream) {
  NS_CHECK_ARG(p && g && m && v && cfg && step_dev && norm2_dev && found_inf_dev && n > 0, "ns_adamw_step: bad arguments");
  int nb = (int)((n + 1023) / 1024);
  if (nb > 2048) nb = 2048;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, st, p, g, m, v, n, *cfg, step_dev, norm2_dev, found_inf_dev,
                     (const float*)loss_scale_dev);
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, st, step_dev, loss_scale_dev, growth_tracker_dev,
                     found_inf_dev, cfg->scale_growth, cfg->scale_backoff, cfg->scale_interval);
  NS_CHECK_LAUNCH("ns_adamw_step");
  return NS_OK;
}
