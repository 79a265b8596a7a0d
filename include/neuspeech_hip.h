/*
 * libneuspeech_hip — C ABI of the MI355X (gfx950) hot path of NeuSpeech's
 * Whisper-based MEG->text training / decoding.
 *
 * The reference (NeuSpeech/NeuSpeech1, pure Python) has no FFI: its hot path is
 * torch.nn / HuggingFace / PEFT module calls.  Each entry point below cites the
 * reference call site (file:line, relative to the reference tree, or HF: for
 * transformers' modeling_whisper.py) whose arithmetic it replaces.  A
 * maintainer binds these with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless named host_*; the caller owns all
 *    memory (the library never allocates device memory nor retains pointers).
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream).  All
 *    functions only enqueue work; none synchronises.
 *  - return value: 0 = ok, <0 = ns_status; text via ns_last_error()
 *    (thread-local).
 *  - fp16 tensors are IEEE binary16 ("f16"), accumulations are fp32.
 *  - activations are TOKEN-MAJOR: (rows = batch*time, cols = channels).
 */
#ifndef NEUSPEECH_HIP_H
#define NEUSPEECH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NS_OK = 0,
  NS_ERR_BAD_ARG = -1,
  NS_ERR_UNSUPPORTED = -2,
  NS_ERR_HIP = -3
} ns_status;

int ns_version(void);                 /* ABI version, currently 3 (bumped with every descriptor-layout / signature change) */
const char* ns_last_error(void);      /* thread-local, never NULL */

/* ------------------------------------------------------------------------
 * Row maps.  A logical row index m of a matrix is located at element offset
 *     (m / seg_rows) * seg_stride + (m % seg_rows) * ld          seg_rows > 0
 *     m * ld                                                      seg_rows == 0
 * This lets one GEMM walk a (B, T+2, C) halo-padded activation as the
 * overlapping-row im2col view of a k=3 Conv1d (ld = stride*C, K = 3*C) with no
 * im2col buffer.  seg_rows must be a multiple of 4.
 * ---------------------------------------------------------------------- */
typedef struct {
  int64_t seg_stride;
  int32_t seg_rows;
  int32_t ld;
} ns_rowmap;

enum {
  NS_GEMM_GELU = 1,      /* G16 / H32 take gelu(round16(acc+bias)) */
  NS_GEMM_DGELU = 2,     /* C16 = round16(round16(acc) * gelu'(P16)) */
  NS_GEMM_TN = 4,        /* operands are reduction-major: A is (Kred x M) as X[m_red][m], see ns_gemm */
  NS_GEMM_ATOMIC32 = 8,  /* C32 += acc (fp32 atomics), for split reductions */
  NS_GEMM_DROP_A = 16,   /* NT: A is multiplied by the LoRA-dropout keep mask (forward down-projection; mask only) */
  NS_GEMM_GELU_SAVE_GRAD = 32, /* with NS_GEMM_GELU: C16 = round16(gelu'(x)) instead of x = round16(acc+bias): the backward then
                                  multiplies (NS_GEMM_MUL_P16) instead of re-evaluating erf / exp per element */
  NS_GEMM_MUL_P16 = 64,  /* C16 = round16(round16(acc) * P16) */
  NS_GEMM_COLSUM_A = 128 /* TN: additionally H32[i] += alpha * sum_m A[m][i] (fp32 atomics; H32 = M floats): the bias gradient of
                            a conv rides on its weight-gradient GEMM instead of a second pass over d(pre) */
};

/*
 * ns_gemm: C = A * B^T (+ A2 * B2^T) with fused epilogues, fp16 in / fp32 acc.
 *
 * NT form (default):  A (M x K) rows via `am`, B (N x K) row-major ldb.
 *   Replaces torch.nn.Linear / Conv1d forward+dgrad at
 *   utils/model_utils.py:13-15, utils/load_model.py:410-411 (conv stem),
 *   HF:modeling_whisper.py:279-282,309,332-333,354 (q/k/v/out), :403-405
 *   (fc1/fc2), utils/load_model.py:1047 (proj_out) and their autograd
 *   backward; the optional second product is the LoRA side path
 *   (finetune.py:205-212, peft lora.Linear.forward).
 *   If a2_ngroup > 0, A2's column window starts at ((n0 / a2_ngroup) * K2)
 *   for the output-column tile starting at n0 (fused q|k|v with separate
 *   LoRA-A outputs side by side).
 *
 * TN form (NS_GEMM_TN): C[n][k] = sum_m A[m][n] * B[m][k], i.e. both operands
 *   are stored reduction-major (A = dY (Mred x N) rows via `am`, B = X
 *   (Mred x K) rows via `bm`); M in the descriptor is the number of OUTPUT
 *   rows (N of dY), N the number of output columns, K the reduction length.
 *   The reduction is split over gridDim.z = `splits` chunks and accumulated
 *   with fp32 atomics into C32 (which the caller zeroes).  Replaces the weight
 *   gradients autograd produces for the trainable Conv1d / LoRA tensors
 *   (finetune.py:202,205-212).
 *
 * Epilogue, per element, v = acc (+ bias[n]):
 *   v16 = round16(v);  DGELU: v16 = round16(v16 * gelu'(P16))
 *   C16 <- v16 (if C16);  g = GELU ? round16(gelu(v16)) : v16;  G16 <- g
 *   H32 <- R32 + g (+ pos[m % pos_rows][n])  (if H32)
 *   C32 (+)= v (if C32)
 */
typedef struct {
  const void* A;  ns_rowmap am;  int32_t K;
  const void* B;  ns_rowmap bm;              /* NT: bm.ld = ldb, seg_rows = 0 */
  const void* A2; ns_rowmap am2; int32_t K2; /* optional second product */
  const void* B2; int32_t ldb2;
  int32_t a2_ngroup;
  int32_t M, N;
  const float* bias;
  void* C16;       ns_rowmap c16m;
  void* G16;       ns_rowmap g16m;
  const void* P16; ns_rowmap p16m;
  const float* R32; float* H32; ns_rowmap h32m;
  const float* pos; int32_t pos_rows;
  float* C32; int32_t ldc32;
  int32_t flags;
  int32_t splits;           /* TN only: reduction split count (>=1) */
  /* LoRA-dropout hook (NT dgrad): if drop_p > 0 (<= 0.5) the (A2,B2) product is formed FIRST, multiplied
     element-wise by the keep MASK keep(seed,row,col) (p quantised to thr8/256), and the main product accumulates on
     top.  For TN the mask multiplies operand B, with NS_GEMM_DROP_A operand A.  The kernels never scale: the
     survivors' 256/(256-thr8) belongs in the caller's alpha (of the down-projection / of the GEMM producing A2). */
  float drop_p; uint32_t drop_seed;
  float alpha;              /* acc is multiplied by alpha first (0 is read as 1) */
  /* Side product of a GELU epilogue (the LoRA down-projection of the NEXT Linear, peft lora.Linear.forward of fc2:
     u = A drop(gelu(fc1 x)), finetune.py:205-212): if side_B != NULL the kernel also forms, per 256-column output tile t,
       side_out[t][m][j] = sum_{n in tile t} mask(side_drop_seed, m, n) * G16[m][n] * side_B[j][n]      (fp32, j < side_n = 32)
     from the fp16 GELU values it has just staged, so the (M x N) GELU output is not read again for it.  side_out holds
     (N / 256) slabs of (M x 32) floats; ns_gemm_side_reduce sums them, scales and rounds to fp16.  Only on the large-M path
     (ns_gemm_side_supported), with NS_GEMM_GELU and G16; N % 256 == 0.  The mask is the LoRA-dropout keep mask (mask only). */
  const void* side_B; int32_t side_ldb, side_n;
  float* side_out;
  float side_drop_p; uint32_t side_drop_seed;
  /* Device-resident step counter for the dropout masks (ABI 2): when non-NULL every mask of this launch uses
     seed + (*seed_dev) * 0x9E3779B1 in place of seed (drop_seed and side_drop_seed alike).  The training step then has
     no launch argument that changes from step to step, which is what lets it be captured in a hipGraph: the counter is
     advanced on the device (by the optimizer step), the graph is replayed unchanged. */
  const uint32_t* seed_dev;
} ns_gemm_desc;

int ns_gemm(const ns_gemm_desc* d, void* stream);
int ns_gemm_side_supported(int M, int N, int K);
/* u16[m][j] = round16(alpha * sum_t slabs[t][m][j]),  t < tiles, j < 32 */
int ns_gemm_side_reduce(const float* slabs, int tiles, int M, float alpha, void* u16, int ldu, void* stream);
/* ------------------------------------------------------------------------
 * ns_gemm_ln: a residual Linear with N = 512 output columns and the LayerNorm that reads its result, in one launch
 * (HF:modeling_whisper.py:354 out_proj / :405 fc2 -> :394,:407 residual add -> :402 final_layer_norm / :392 the next layer's
 * self_attn_layer_norm / utils/load_model.py:468 encoder.layer_norm):
 *     H32 = R32 + round16(A W^T (+ A2 B2^T) + bias);   x16 = LN(H32) * gamma + beta;   mean / rstd saved for the backward.
 * `g` describes the Linear exactly as for ns_gemm (NT form; A, am, K, B, bias, the optional second product, R32, H32, h32m, alpha)
 * with N = 512 and nothing else set; H32, x16, mean and rstd are bitwise what ns_gemm followed by ns_layernorm_fwd produce.
 * A workgroup owns complete rows, so H32 is not read back and A is read once (csrc/ns_gemm_rowln.hip).
 * ns_gemm_ln_supported: N == 512, K % 64 == 0, K2 in {0, 16, 32}, M >= 1024.
 * ---------------------------------------------------------------------- */
typedef struct {
  ns_gemm_desc g;
  const float* gamma; const float* beta;
  float eps; int32_t ldx;                 /* ldx: row stride of x16 (elements) */
  void* x16; float* mean; float* rstd;    /* mean / rstd may be NULL (inference) */
} ns_gemm_ln_desc;
int ns_gemm_ln_supported(int M, int N, int K, int K2);
int ns_gemm_ln(const ns_gemm_ln_desc* d, void* stream);

/* A/B knob for benchmarks: 1 (default) = automatic kernel choice, 0 = register-staged kernel only; 2..5 force one of
 * the wide NT kernels (see ns_gemm.hip), 6 = automatic without the small-M split-K kernel */
void ns_debug_set_ring(int on);
/* A/B knob (NS_AD_SELF sets the same): which kernel ns_attn_decode's ancestry-layout launches (the decode loop's self-attention) take:
 * 1 (default) = one wave per (row, head) when groups * H > 2048, else the four-wave kernel of the cross-attention layout; 0 = never the
 * wave-per-head form; 2 = always.  Outputs agree up to fp32 summation order. */
void ns_debug_set_ad_self(int mode);

/* ------------------------------------------------------------------------
 * LayerNorm over the fp32 residual stream (eps 1e-5, affine), one row = d
 * channels.  Replaces nn.LayerNorm at HF:modeling_whisper.py:392,402,
 * utils/load_model.py:468 (encoder final) and the decoder's three per layer.
 * fwd: y16 (and/or y32) = LN(x); mean/rstd (fp32, rows) are saved for bwd.
 * bwd: dx = LN'(dy) with frozen gamma/beta (no parameter grads: the reference
 *      freezes them, finetune.py:176); dx32 = dres + dx (residual-gradient
 *      accumulate, dres may be NULL), dx16 = round16(dx32) for the next GEMM.
 * d must be a multiple of 256 and <= 1280.
 * ---------------------------------------------------------------------- */
int ns_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y16, float* y32,
                     float* mean, float* rstd, int rows, int d, float eps, void* stream);
int ns_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* mean, const float* rstd,
                     const float* gamma, const float* dres, float* dx32, void* dx16, int rows, int d,
                     void* stream);
/* ------------------------------------------------------------------------
 * ns_signal_pack: the MEG batch (B, ch, T) fp32 exactly as the collator emits
 * it (utils/data_utils.py:191-193) -> (B, T+2, Cp) fp16 token-major with zero
 * halo rows 0 and T+1 and zero channels [ch, Cp); Cp % 8 == 0.  This is the
 * only full-size read of the signal tensor (coalesced along T).
 * ---------------------------------------------------------------------- */
int ns_signal_pack(const float* x, void* out16, int B, int ch, int T, int Cp, void* stream);

/* ------------------------------------------------------------------------
 * ns_feed_pack: the reader's crop / pad / cast rules as ONE device pass over
 * the raw recordings.  Item b describes recording b as
 * it sits in a device staging buffer: the channel rows the reader keeps
 * (utils/reader.py:282-290: schoffelen [28:301], gwilliams [:208], else
 * [:modal_ch]), `n` samples each, in the file's own dtype.  The kernel
 *   - zero-fills channels [rows, ch)            (pad_sample_ch, reader.py:508-516)
 *   - crops at T and zero-fills samples [n, T)  (padding_sample, reader.py:496-506)
 *   - rounds to fp32 exactly like the collator  (utils/data_utils.py:191-193)
 *   - and emits what ns_signal_pack emits: (B, T+2, Cp) fp16 with zero halo rows;
 *     x32 (optional) receives the collator's own (B, ch, T) fp32 tensor.
 * rows <= ch is required (the reader asserts the same shape).
 * ---------------------------------------------------------------------- */
enum { NS_FEED_F64 = 0, NS_FEED_F32 = 1, NS_FEED_F16 = 2 };
typedef struct ns_feed_item {
  const void* src;   /* device pointer to element [row 0][sample 0] of the kept rows, aligned to its dtype */
  long long ld;      /* elements between consecutive channel rows */
  int rows;          /* channel rows present, 0 .. ch */
  int n;             /* samples present per row, >= 0 (cropped at T) */
  int dtype;         /* NS_FEED_* */
  int reserved;
} ns_feed_item;
int ns_feed_pack(const ns_feed_item* items_dev, int B, int ch, int T, int Cp, void* out16, float* x32,
                 void* stream);

/* h32[row] = E32[ids[row]] + P32[pos0 + row % L]; pos0_dev (optional) overrides
 * pos0 from device memory (decode step counter).  utils/load_model.py:645,668-673 */
int ns_embed_pos(const int64_t* ids, const float* E32, const float* P32, float* h32, int rows, int L, int d,
                 int pos0, const int* pos0_dev, void* stream);

/* conv-stem backward seams (the GELUs at utils/model_utils.py:14 and
 * utils/load_model.py:410-411): out16[map(row)] = round16(a16[row] * gelu'(pre16[row])), or, with pre_is_grad,
 * round16(a16[row] * pre16[row]) (pre16 already holds gelu', see NS_GEMM_GELU_SAVE_GRAD);
 * bias gradient out32[c] += alpha * sum_rows a16[row][c] (fp32 atomics). */
int ns_dgelu_mul(const void* a16, const void* pre16, void* out16, const ns_rowmap* out_map, int rows, int cols,
                 int pre_is_grad, void* stream);
int ns_colsum(const void* a16, float* out32, int rows, int cols, int ld, float alpha, void* stream);

/* Buffer clears and the step counter of a captured training step as KERNELS: the step (finetune.py:231-253,281 as the HF
 * Trainer runs it: zero_grad, forward, backward, optimizer) is replayed from hipGraphs, and a captured memset NODE was not
 * reliably ordered against the kernel nodes around it under back-to-back replays (see ns_orth_reg).  ns_zero_spans clears
 * up to NS_ZERO_MAX_SPANS buffers in one launch (`spans` is a HOST array, copied into the launch arguments; each span
 * 16-byte aligned, a multiple of 4 bytes); ns_add_i32 adds v to n <= 64 consecutive device counters (n = 1: the LoRA-dropout
 * step counter that ns_gemm_desc.seed_dev points at; n = 2: the decode loop's position / length pair, one launch per step). */
#define NS_ZERO_MAX_SPANS 8
typedef struct { void* p; size_t bytes; } ns_span;
int ns_zero_spans(const ns_span* spans, int n, void* stream);
int ns_add_i32(int32_t* counter_dev, int32_t n, int32_t v, void* stream);

/* batched fp32 -> fp16 operand refresh after an optimizer step:
 * dst[r][c] = scale*src[r][c]  or (transpose) dst[c][r] = scale*src[r][c] */
typedef struct {
  const void* src; void* dst;
  int32_t rows, cols, ld_src, ld_dst;
  float scale; int32_t transpose;
  const float* colscale;   /* optional: src[r][c] is additionally multiplied by colscale[c] (AdaLoRA's diag(E)) */
} ns_cast_job;
int ns_cast_jobs(const ns_cast_job* jobs_dev, int njobs, void* stream);

/* ------------------------------------------------------------------------
 * AdaLoRA (the reference's default adapter, finetune.py:205-208; arithmetic in peft's AdaLoraLayer / AdaLoraModel):
 *   y = W x + b + B((A x) * E) * alpha / (r + 1e-5),  loss += orth_reg_weight * mean_P || P P^T - I ||_F.
 * The engine folds diag(E) and the scale into the fp16 B operand (ns_cast_job.colscale); these two entry points
 * turn the folded gradient back into dB / dE and add the regulariser's value and gradient.
 * ---------------------------------------------------------------------- */
int ns_adalora_fold_grads(const float* dBf, const float* B, const float* E, float* dB, float* dE, int N, int r,
                          float s, void* stream);
/* the same fold for a table of adapters in one launch; dBf (the scratch gradient of the folded operand) is read, folded
 * and ZEROED, so a scratch buffer that starts clear stays clear from step to step; r <= 32 */
typedef struct {
  float* dBf; const float* B; const float* E; float* dB; float* dE;
  int32_t N, r; float s;
} ns_adalora_fold_job;
int ns_adalora_fold_jobs(const ns_adalora_fold_job* jobs_dev, int njobs, void* stream);
#define NS_ORTH_MAX_R 32            /* largest live rank ns_orth_reg handles (the reference's init_r is 12) */
typedef struct {
  const float* P; float* G;       /* parameter and its gradient (same layout) */
  int32_t r, len, ld, is_b;       /* lora_A: (r x len) rows; lora_B: (len x r) with is_b = 1; ld = row stride; r <= NS_ORTH_MAX_R */
} ns_orth_job;
/* workspace: ns_orth_reg_workspace_bytes(njobs) bytes (the Gram matrices, summed over the blocks that share a matrix) */
size_t ns_orth_reg_workspace_bytes(int njobs);
int ns_orth_reg(const ns_orth_job* jobs_dev, int njobs, float weight_over_num, const float* loss_scale_dev,
                float* reg_out_dev, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * LoRA backward, the two products that read dy, in one pass over it (peft lora.Linear backward of
 * y += scale * B (A drop(x)), finetune.py:187-212): for each of the G column groups of dy (q | k | v: G = 3, else 1)
 *   du[:, g*r : (g+1)*r]  = alpha_du * dy_g sB_g           (M x r fp16; sBT[g] = (scale * B_g)^T, r x N fp16, ld = N)
 *   dB[g] (N x lddb fp32) += alpha_db[g] * dy_g^T u_g       (u = the forward bottleneck, M x G*r fp16; summed over the
 *                                                            workgroups through `workspace` slabs, or by fp32 atomics)
 * dy_g = columns [g*N, (g+1)*N) of dy.  N % 256 == 0, r in {16, 32} (the padded rank), G in {1, 3};
 * ns_lora_bwd_supported() says whether a shape is built (callers keep the two-GEMM path otherwise).
 * `splits` = workgroups (row ranges), 0 = default.
 * ---------------------------------------------------------------------- */
typedef struct {
  const void* dy; const void* u; void* du;
  const void* sBT[3]; float* dB[3];
  int32_t M, N, r, G;
  int32_t ldy, ldu, lddu, lddb;
  float alpha_du; float alpha_db[3];
  int32_t splits;
  /* optional: with a workspace of ns_lora_bwd_workspace_bytes(M, N, G, splits) bytes the per-workgroup dB partials go
   * through plain stores + a small reduce launch instead of fp32 atomics (the result is then also bitwise reproducible) */
  void* workspace; size_t workspace_bytes;
} ns_lora_bwd_desc;
size_t ns_lora_bwd_workspace_bytes(int M, int N, int G, int splits);
int ns_lora_bwd_supported(int N, int r, int G);
int ns_lora_bwd_dudb(const ns_lora_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------
 * Fused attention, head_dim 64.  Row (b*L + i) of each token-major matrix,
 * head h at column h*64.  q must be pre-scaled by head_dim^-0.5 (HF folds it
 * into q_proj's output, modeling_whisper.py:309).  causal: key j is visible to
 * query i iff j <= i + (Lk - Lq).  LSE/Delta are (B, H, Lq) fp32.
 * Replaces HF:modeling_whisper.py:215-238 and its backward.
 * ---------------------------------------------------------------------- */
typedef struct {
  const void *Q, *K, *V; void* O;
  const void* dO; void *dQ, *dK, *dV;
  float* LSE; float* Delta;
  int32_t B, H, Lq, Lk, head_dim;
  int32_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int32_t causal;
  /* backward only (ABI 2): with a workspace of ns_attn_bwd_workspace_bytes(...) bytes (> 0 for unmasked attention over
     >= 256 queries and keys) the backward runs in ONE pass -- S and dP formed once, dQ summed over the key sweeps of a
     (batch, head) in this caller-owned fp32 scratch, no atomics (csrc/ns_attn_bwd1.hip) -- instead of the dQ pass +
     dK/dV pass, which each recompute S and dP.  Also > 0 for unmasked attention with <= 64 queries over >= 256 keys (the
     decoder's cross-attention): K / V are streamed once, dQ leaves through fp32 slabs in the workspace (one per group of
     key blocks) that a second launch adds in a fixed order (csrc/ns_attn.hip, attn_bwd_fewq_kernel).  NULL / 0 keeps
     the two-pass kernels. */
  void* workspace; size_t workspace_bytes;
} ns_attn_desc;
int ns_attn_fwd(const ns_attn_desc* d, void* stream);
int ns_attn_bwd(const ns_attn_desc* d, void* stream);   /* writes dQ, dK, dV and Delta */
size_t ns_attn_bwd_workspace_bytes(int B, int H, int Lq, int Lk, int causal);   /* 0: the one-pass form does not apply */

/* ------------------------------------------------------------------------
 * Cross-entropy on fp16 logits (rows x ldv, first V columns valid), labels
 * int64 with -100 = ignore: loss = mean over valid rows (fp32, *loss_dev),
 * dlogits16 = round16((softmax - onehot) * loss_scale / n_valid) (may alias
 * logits16; NULL = forward only).  utils/load_model.py:1049-1054.
 * ---------------------------------------------------------------------- */
int ns_cross_entropy(const void* logits16, const int64_t* labels, int rows, int V, int ldv, float* row_loss,
                     void* dlogits16, int* nvalid_dev, const float* loss_scale_dev, float* loss_dev,
                     void* stream);
/* first-index argmax of each row's first V columns (evaluation.py:394-399, greedy step) */
int ns_argmax_rows(const void* logits16, int rows, int V, int ldv, int64_t* out, void* stream);

/* ------------------------------------------------------------------------
 * Optimizer tail of one training step on the flat fp32 trainable buffer
 * (HF Trainer defaults configured at finetune.py:231-253): grad-norm + inf
 * check -> [skip if inf] unscale, clip to max_grad_norm, AdamW, LR schedule,
 * GradScaler update; all state on device, no host sync.
 * ---------------------------------------------------------------------- */
typedef struct {
  float lr, beta1, beta2, eps, weight_decay, max_grad_norm;
  int32_t warmup_steps, total_steps;       /* total_steps <= 0: constant lr */
  float scale_growth, scale_backoff; int32_t scale_interval;
} ns_adamw_cfg;
size_t ns_grad_norm_workspace_bytes(void);
int ns_grad_norm(const float* g, size_t n, void* workspace, float* norm2_dev, int* found_inf_dev, void* stream);
int ns_adamw_step(float* p, const float* g, float* m, float* v, size_t n, const ns_adamw_cfg* cfg,
                  int* step_dev, const float* norm2_dev, const int* found_inf_dev, float* loss_scale_dev,
                  int* growth_tracker_dev, void* stream);

/* ------------------------------------------------------------------------
 * Decode loop (evaluation.py:369-386 -> HF GenerationMixin greedy / beam search).
 * ---------------------------------------------------------------------- */
/* One new query per row against a K/V cache, head_dim 64.  Rows are grouped: group g = rows
 * [g*nq, (g+1)*nq) share ONE K/V stream (cross-attention: the encoder K/V of a sequence is read once
 * for all its beams; key row = g*kv_group_stride + j).  With `anc` (nq must be 1) key j of row r lives at
 * row j*kv_pos_stride + anc[r*anc_ld + j] (self-attention cache laid out [position][slot], beams reordered by
 * rewriting the small int32 ancestry table instead of the K/V tensors, cf. utils/load_model.py:1353-1360).
 * kv_len_dev (optional) overrides Lk from device memory. */
typedef struct {
  const void *Q, *K, *V; void* O;
  const int32_t* anc; const int32_t* kv_len_dev;
  int32_t groups, nq, H, Lk, Lk_max, head_dim;
  int32_t ldq, ldk, ldv, ldo, anc_ld;
  int64_t kv_group_stride, kv_pos_stride;
  /* optional append (anc layout, nq == 1): Knew / Vnew = this step's key / value rows (row r at r*ldnew, head h at
   * h*64) -- position Lk-1 is read from them and also written into the cache row (Lk-1)*kv_pos_stride + r, so the
   * decode loop needs no separate cache-update launch (utils/load_model.py:1332-1351 keeps past_key_values).
   * slot0 (ABI 3; the field was `reserved`, always 0): the append writes cache slot slot0 + r.  A caller that runs a ROW RANGE
   * [slot0, slot0 + groups) of the batch as a launch of its own (the decode loop's half-batches on two streams) offsets Q / O /
   * anc / Knew / Vnew itself and leaves K / V -- which `anc` indexes by global slot -- at their base. */
  const void *Knew, *Vnew; int32_t ldnew, slot0;
} ns_attn_decode_desc;
int ns_attn_decode(const ns_attn_decode_desc* d, void* stream);

/* Cross-attention of a FEW query rows (nq <= 16: the beams of one sequence, or the single greedy row) against the
 * sequence's long encoder K/V, on the matrix cores and without LDS staging: the four waves of a workgroup split the
 * keys, load K rows and rows of the TRANSPOSED value image straight into MFMA fragments, keep their own online-softmax
 * state and meet once at the end.  The K/V of a sequence is written once per utterance and read at every decode step
 * (evaluation.py:369-386 -> utils/load_model.py:1332-1351 keeps the cross K/V in the cache), so the value image is
 * stored transposed once by ns_vt_pack: element (group g, head h, dim d, key j) at ((g*H + h)*64 + d)*ldvt + j,
 * ldvt % 32 == 0, ldvt >= Lk, columns [Lk, ldvt) zero.  Key row of K: g*Lk + j (row stride ldk).  Queries pre-scaled. */
typedef struct {
  const void *Q, *K, *Vt; void* O;
  int32_t groups, nq, H, Lk;
  int32_t ldq, ldk, ldvt, ldo;
} ns_attn_fewq_desc;
int ns_attn_fewq(const ns_attn_fewq_desc* d, void* stream);
/* V (fp16 rows g*Lk + j, row stride ldv, head h at columns h*64..) -> the transposed image described above */
int ns_vt_pack(const void* v16, int ldv, void* vt16, int groups, int H, int Lk, int ldvt, void* stream);

/* Last-position logits (fp16, rows x ldv) -> processed fp32 scores (rows x V):
 * [log_softmax] -> repetition penalty (s<0 ? s*p : s/p on tokens already in ids[row][0:cur_len]) ->
 * no-repeat-ngram (-inf) -> suppress / begin-suppress lists (-inf) -> + beam_scores[row].
 * HF:generation/logits_process.py:306-414, :1073-1141, :1816-1906; order as HF:generation/utils.py:3388-3407. */
typedef struct {
  const void* logits16; float* scores32; const int64_t* ids; const float* beam_scores;
  const int32_t* suppress; const int32_t* begin_suppress; const int32_t* cur_len_dev;
  int32_t rows, V, ldv, ids_ld, cur_len, begin_index;
  int32_t n_suppress, n_begin_suppress, no_repeat_ngram, log_softmax;
  float repetition_penalty;
  /* sequence bias (HF SequenceBiasLogitsProcessor, generation/logits_process.py; FIRST in HF's processor order, i.e.
   * added right after the log-softmax and before the repetition penalty): bias1 = dense per-token bias of the
   * length-1 sequences (V floats) or NULL; the n_seq longer sequences are seq_tok[seq_off[s] .. seq_off[s+1]) and add
   * seq_bias[s] to their LAST token in every row whose history ends with the tokens before it.  ns_logits_process only
   * (ns_logits_select refuses a descriptor that carries a bias). */
  const float* bias1; const int32_t* seq_tok; const int32_t* seq_off; const float* seq_bias;
  int32_t n_seq;
  /* forced decoder ids (generation_config.forced_decoder_ids as the reference's generate wrapper hands them to
   * super().generate, utils/load_model.py:1210-1256,1314-1322; HF ForceTokensLogitsProcessor, LAST in HF's processor
   * order): forced[pos] = token forced at sequence position pos (cur_len), or -1 = free; positions >= n_forced are free.
   * At a forced position every score becomes -inf except the forced token's, which becomes 0 (+ beam score). */
  int32_t n_forced;
  const int32_t* forced;
} ns_logits_proc_desc;
int ns_logits_process(const ns_logits_proc_desc* d, void* stream);

/* ns_logits_process + per-row top-k fused, without materialising the fp32 score matrix (d->scores32 is ignored):
 * cand_vals / cand_idx (rows x k) receive each row's k best processed scores, ordered (value desc, column asc), with
 * cand_idx = (row % group_rows) * V + column -- the flat index HF's beam search takes its top-2*beams over
 * (HF:generation/utils.py:3409-3417).  Values and order are bit-identical to ns_logits_process followed by a top-k.
 * ns_topk_merge then reduces each group's ncand = group_rows * k candidates to its k best (same ordering). */
int ns_logits_select(const ns_logits_proc_desc* d, int k, int group_rows, float* cand_vals, int* cand_idx, void* stream);
int ns_topk_merge(const float* cand_vals, const int* cand_idx, int groups, int ncand, int k, float* vals, int* idx,
                  void* stream);

/* top-k (k <= 16) of each group of n contiguous floats, ordered (value desc, index asc); two-stage (per-chunk
 * candidates in `workspace`, then a merge) */
size_t ns_topk_workspace_bytes(int groups, long long n, int k);
int ns_topk_groups(const float* x, int groups, long long n, int k, float* vals, int* idx, void* workspace,
                   void* stream);

/* HF beam-search bookkeeping for one step (HF:generation/utils.py:3077-3204, :3008-3075): from the top-2*beams
 * candidates pick the next running beams, merge just-finished hypotheses (score / (cur+1-prompt)^lp) into the
 * finished set, update the per-row early-stop flag; ORs into *any_open / *any_continuation (caller zeroes). */
typedef struct {
  const float* top_vals; const int32_t* top_idx;
  const int64_t* run_seqs_in; int64_t* run_seqs_out; float* run_scores_out;
  const int64_t* fin_seqs_in; int64_t* fin_seqs_out;
  const float* fin_scores_in; float* fin_scores_out;
  const uint8_t* fin_done_in; uint8_t* fin_done_out;
  uint8_t* open; int32_t* parent_out; int64_t* next_tok_out;
  int32_t* any_open; int32_t* any_continuation; const int32_t* cur_len_dev;
  int32_t batch, num_beams, V, max_len, cur_len, prompt_len, eos_id;
  float length_penalty;
} ns_beam_desc;
int ns_beam_update(const ns_beam_desc* d, void* stream);
int ns_anc_update(const int* anc_in, int* anc_out, const int* parent, int rows, int ld, int cur, const int* cur_dev,
                  void* stream);
/* greedy: argmax of processed scores (or best_idx[row] from ns_logits_select with k = 1, scores may then be NULL),
 * finished rows emit pad (HF:generation/utils.py:2897-2960) */
int ns_greedy_update(const float* scores, int rows, int V, int64_t* seqs, int ld, int cur, const int* cur_dev, int eos,
                     int pad, unsigned char* done, int* any_open, int64_t* next_tok, const int* best_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUSPEECH_HIP_H */
