// Probe of ds_read_b64_tr_b16 semantics (gfx950): prints, for every lane, the 4 elements it receives when
// lane 4q+p of each 16-lane group supplies the address of row q, cols 4p..4p+3 of a 4x16 block.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short tile[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = (short)i;   // tile[r][c] = r*64 + c
  __syncthreads();
  const int l = threadIdx.x;
  const int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const short* addr = tile + (4 * g + q) * 64 + 4 * p;
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
    printf("\n");
  }
  return 0;
}
