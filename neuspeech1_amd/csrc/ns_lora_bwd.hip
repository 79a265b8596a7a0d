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

constexpr int RC = 64, SC = 256, NTH = 512, NBUF = 3;
constexpr int DY_BYTES = RC * SC * 2;        // 32 KiB: dy slice, 64 rows x 512 B
constexpr int SB_BYTES = 32 * SC * 2;        // 16 KiB: sB^T slice, 32 bottleneck rows x 512 B
constexpr int SLOT_BYTES = DY_BYTES + SB_BYTES;
constexpr int U_STRIDE = 64;                 // bytes per u row image (32 halfs)
constexpr int U_BYTES = RC * U_STRIDE;       // 4 KiB
constexpr int LDS_BYTES = NBUF * SLOT_BYTES + 3 * U_BYTES;   // 156 KiB

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_s4;

__device__ __attribute__((aligned(16))) const uint32_t ns_lb_zero_chunk[4] = {0, 0, 0, 0};

// 16-B chunk c of image row r sits at chunk c ^ lb_swz(r): the row's low two bits go to chunk bits 2-3, so the four rows of a transposed read fall into
// four different 64-B windows of the 256-B bank span (c ^ r permuted them inside ONE window: 4-way conflicts on every dy^T fragment), and the sixteen
// rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27} ...) still get sixteen different chunk positions.
__device__ __forceinline__ int lb_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int dy_off(int row, int chunk) { return row * 512 + ((chunk ^ lb_swz(row)) << 4); }

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

// LDS-DMA as inline asm.  With the builtin hipcc knows that an LDS write is pending on the vm counter and, the ring
// slots being runtime-indexed, fences the loop's ds_read / ds_write with s_waitcnt vmcnt(0) -- which drains the items in
// flight.  Hidden in asm, the transfers are ordered by this kernel's own counted waits and raw barriers only.  M0 carries
// the wave-uniform LDS byte address and is saved / restored inside the statement (cdna guide §5.7).
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#define NS_LB_BARRIER()                                   \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)

// Every operand of an item -- the dy slice (64 x 256), the sB^T slice (32 x 256) and, at a group's first slice, the chunk's
// u rows -- travels by LDS-DMA into a ring of three slots, issued TWO items ahead of the one being multiplied: no staging
// registers, ~100 KiB in flight per CU, and no register load in the loop for hipcc to fence.  The swizzle sits on the
// per-lane SOURCE address (LDS chunk c' of row r <- global chunk c' ^ lb_swz(r)).  A wave issues 6 pieces per item (+ 2
// four-byte pieces of u at a group's first slice); the counted wait at the top of an iteration leaves exactly the pieces of
// the NEXT-BUT-ONE item in flight.
template <int G, int NSUB>
__global__ __launch_bounds__(NTH) void lora_bwd_dudb_kernel(const ns_lora_bwd_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const ub = smem + NBUF * SLOT_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int nchunks = (p.M + RC - 1) / RC;
  const int per = (nchunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = blockIdx.x * per, c_hi = min(nchunks, c_lo + per);
  constexpr int IPC = G * NSUB;                  // items per chunk
  const int nitems = max(c_hi - c_lo, 0) * IPC;

  f32x16 accB[G][NSUB];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[g][s][r] = 0.f;
  f32x16 accU;                                   // waves 0 / 1: du of chunk rows 0..31 / 32..63, all columns of the group
#pragma unroll
  for (int r = 0; r < 16; ++r) accU[r] = 0.f;

  // 6 (+2) pieces per wave: dy rows [8w, 8w+8) (4 x 1 KiB), sB^T rows [4w, 4w+4) (2 x 1 KiB), u rows [8w, 8w+8) (2 x 256 B)
  auto issue = [&](int c, int g, int s, int slot) __attribute__((always_inline)) {
    const int row0 = c * RC;
    int col0 = g * p.N + s * SC, scol0 = s * SC;
    asm volatile("" : "+s"(col0), "+s"(scol0));      // opaque: keeps hipcc from hoisting one pointer set per unrolled position
    const half_t* dy = (const half_t*)p.dy + col0;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * SLOT_BYTES + wave * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * wave + 2 * i + lh, cl = lr ^ lb_swz(row);
      const void* src = row0 + row < p.M ? (const void*)(dy + (long long)(row0 + row) * p.ldy + cl * 8) : (const void*)ns_lb_zero_chunk;
      glds16_asm(src, dst + i * 1024);
    }
    const half_t* sbt = (const half_t*)p.sBT[g] + scol0;
    const unsigned sdst = __builtin_amdgcn_readfirstlane(lds_base + slot * SLOT_BYTES + DY_BYTES + wave * 2048);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 4 * wave + 2 * i + lh, cl = lr ^ lb_swz(row);       // bottleneck row j
      const void* src = row < p.r ? (const void*)(sbt + (long long)row * p.N + cl * 8) : (const void*)ns_lb_zero_chunk;
      glds16_asm(src, sdst + i * 1024);
    }
    if (s == 0) {
      const unsigned udst = __builtin_amdgcn_readfirstlane(lds_base + NBUF * SLOT_BYTES + ((c * G + g) % 3) * U_BYTES + wave * 512);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 8 * wave + 4 * i + (lane >> 4), c2 = 2 * (lane & 15);   // 4 rows x 16 pieces of 2 halfs
        const void* src = (row0 + row < p.M && c2 < p.r) ? (const void*)((const half_t*)p.u + (long long)(row0 + row) * p.ldu + g * p.r + c2)
                                                          : (const void*)ns_lb_zero_chunk;
        glds4_asm(src, udst + i * 256);
      }
    }
  };

  // ---- prologue: items 0 and 1
  if (nitems > 0) {
    issue(c_lo, 0, 0, 0);
    if (nitems > 1) {
      constexpr int g1 = (1 % IPC) / NSUB, s1 = (1 % IPC) % NSUB;
      issue(c_lo + 1 / IPC, g1, s1, 1);
    }
  }
  int it = 0;
  for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s = 0; s < NSUB; ++s, ++it) {
        // successors: linear positions l + 1, l + 2 within / past the chunk
        constexpr int dummy = 0;
        (void)dummy;
        const int l1 = g * NSUB + s + 1, l2 = g * NSUB + s + 2;
        const int s1 = (l1 % IPC) % NSUB;
        const int c2 = c + l2 / IPC, g2 = (l2 % IPC) / NSUB, s2 = (l2 % IPC) % NSUB;
        const bool more1 = it + 1 < nitems, more2 = it + 2 < nitems;
        // in flight behind item `it`: exactly the pieces of item it+1 (6, or 8 when it opens a group), issued last in the previous iteration
        if (more1) {
          if (s1 == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        NS_LB_BARRIER();      // every wave's pieces of item `it` have landed; the slot of item it-1 is free
        const char* const buf = ring + (it % NBUF) * SLOT_BYTES;
        const char* const sbuf = buf + DY_BYTES;
        const char* const uimg = ub + ((c * G + g) % 3) * U_BYTES;
        // ---- du: waves 0 and 1 own the chunk's two 32-row halves and walk all 256 slice columns (16 k-steps)
        if (wave < 2) {
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            const half8 a = *(const half8*)(buf + dy_off(32 * wave + lr, 2 * ks + lh));
            const half8 bfr = *(const half8*)(sbuf + dy_off(lr, 2 * ks + lh));
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr, accU, 0, 0, 0);
          }
        }
        // ---- dB: dy columns 32*wave.. of this slice, reduction over the chunk's 64 rows
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const half8 a = tr_frag(buf, [](int row, int col) { return dy_off(row, col >> 3) + (col & 7) * 2; }, 16 * ks, 32 * wave, lane);
          const half8 b = tr_frag(uimg, [](int row, int col) { return row * U_STRIDE + col * 2; }, 16 * ks, 0, lane);
          accB[g][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accB[g][s], 0, 0, 0);
        }
        if (s == NSUB - 1 && wave < 2) {
          // ---- du of (chunk c, group g) is complete in waves 0 / 1: accumulator (row = crow(r), column j = lr)
          const int row0 = c * RC + 32 * wave;
          if (lr < p.r) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (m < p.M) ((half_t*)p.du)[(long long)m * p.lddu + g * p.r + lr] = (half_t)(accU[r] * p.alpha_du);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) accU[r] = 0.f;
        }
        // the next-but-one item goes out LAST in the iteration: at the next counted wait the youngest operations are then
        // exactly its pieces (the du stores above are older).  Its ring slot held item it-1, which every wave left before
        // this iteration's barrier; the u image slot (three of them) held the group three back.
        __builtin_amdgcn_sched_barrier(0);
        if (more2) issue(c2, g2, s2, (it + 2) % NBUF);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- dB partials leave the workgroup: accumulator tile (g, s): rows n = 256*s + 32*wave + crow(r), column j = lr.
  // With a workspace: plain stores into this workgroup's slab (G x N x 32 floats), summed by lora_bwd_reduce_kernel -- the
  // atomics form moves N*r*4 B per workgroup through the ~1.3 TB/s atomic path (52 us of the fc1 site's 143 us).
  if (p.workspace) {
    float* const slab = (float*)p.workspace + (size_t)blockIdx.x * G * p.N * 32;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        float* const dst = slab + ((size_t)g * p.N + s * SC + 32 * wave) * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32] = accB[g][s][r];
      }
    return;
  }
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

// dB[g][n][j] += alpha_db[g] * sum over the workgroups' slabs.  A workgroup owns 16 consecutive float4 of a slab (256 B
// per slab row); its 16 thread groups each sum every 16th slab and meet in LDS -- G*N/2 workgroups, so the d-wide site
// (16.8 MB of slabs) still spreads over the whole chip.
__global__ __launch_bounds__(256) void lora_bwd_reduce_kernel(const ns_lora_bwd_desc p, int nslabs) {
  __shared__ float4 red[16][16];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int per = p.G * p.N * 8;                          // float4 per slab
  const int q = blockIdx.x * 16 + c;
  const float4* src = (const float4*)p.workspace + q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int w = grp; w < nslabs; w += 16) {
    const float4 x = src[(size_t)w * per];
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
  }
  red[grp][c] = a;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  // thread (c, e): element e of float4 c
  const int cc = threadIdx.x >> 2, e = threadIdx.x & 3;
  float v = 0.f;
#pragma unroll
  for (int gq = 0; gq < 16; ++gq) v += ((const float*)&red[gq][cc])[e];
  const int qq = blockIdx.x * 16 + cc;
  const int g = qq / (p.N * 8), rem = qq - g * p.N * 8, n = rem >> 3, j = (rem & 7) * 4 + e;
  if (j < p.r) p.dB[g][(long long)n * p.lddb + j] += v * p.alpha_db[g];
}

template <int G, int NSUB>
int launch(const ns_lora_bwd_desc* d, int grid, hipStream_t st) {
  static ns_dev_once once;           // kernel attribute, once per device (ns_common.h)
  if (!ns_dyn_lds_once(once, {(const void*)lora_bwd_dudb_kernel<G, NSUB>}, LDS_BYTES, "ns_lora_bwd_dudb")) return NS_ERR_HIP;
  hipLaunchKernelGGL((lora_bwd_dudb_kernel<G, NSUB>), dim3(grid), dim3(NTH), LDS_BYTES, st, *d);
  return 0;
}

}  // namespace

extern "C" int ns_lora_bwd_supported(int N, int r, int G) {
  if (N <= 0 || N % SC != 0 || (r != 16 && r != 32) || (G != 1 && G != 3)) return 0;
  const int nsub = N / SC;
  if (G == 1) return nsub == 1 || nsub == 2 || nsub == 5 || nsub == 8;
  return nsub == 1 || nsub == 2;
}

extern "C" size_t ns_lora_bwd_workspace_bytes(int M, int N, int G, int splits) {
  const int nchunks = (M + RC - 1) / RC;
  int grid = splits > 0 ? splits : 256;
  if (grid > nchunks) grid = nchunks;
  return (size_t)grid * G * N * 32 * sizeof(float);
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
  NS_CHECK_ARG(!d->workspace || d->workspace_bytes >= (size_t)grid * d->G * d->N * 32 * sizeof(float),
               "ns_lora_bwd_dudb: workspace of %zu bytes is smaller than ns_lora_bwd_workspace_bytes()", (size_t)d->workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int nsub = d->N / SC;
  int rc = 0;
#define NS_LB(G_, S_) if (d->G == G_ && nsub == S_) { rc = launch<G_, S_>(d, grid, st); }
  NS_LB(1, 1) else NS_LB(1, 2) else NS_LB(1, 5) else NS_LB(1, 8) else NS_LB(3, 1) else NS_LB(3, 2)
#undef NS_LB
  if (rc != 0) return rc;
  if (d->workspace) {
    hipLaunchKernelGGL(lora_bwd_reduce_kernel, dim3(d->G * d->N / 2), dim3(256), 0, st, *d, grid);
  }
  NS_CHECK_LAUNCH("ns_lora_bwd_dudb");
  return NS_OK;
}
