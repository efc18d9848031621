// What does ds_read_b64_tr_b16 return?  LDS holds 16-bit values equal to their own index; every lane passes an address and gets
// four values back.  Prints, per lane, the four indices for two address patterns.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/ds_read_tr_probe.hip -o tools/ubench/_trp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int off;                                   // in 16-bit elements
  if (mode == 0) off = l * 4;                // contiguous: lane l -> elements 4l .. 4l+3
  else if (mode == 1) off = (l & 15) * 64 + (l >> 4) * 4;   // [row = l & 15][64 columns], column block l >> 4
  else off = (l >> 2) * 64 + (l & 3) * 4;    // rows of 64 elements: lane l -> row l >> 2, columns 4 (l & 3) ..
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
