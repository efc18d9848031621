// Bare fp32 MFMA issue-rate microbenchmark: waves/SIMD x accumulators.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 0.001f;
  float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocksPerCU, int iters) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  int grid = 256 * blocksPerCU;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double flops = (double)grid * 4 * iters * 16.0 * NACC * 4096.0;
  printf("acc=%d blocks/CU=%d (waves/SIMD=%d): %.3f ms  %.1f TFLOP/s\n", NACC, blocksPerCU, blocksPerCU, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(1, 4000); run<4>(1, 1000); run<1>(2, 2000); run<4>(2, 500); run<4>(4, 250); run<2>(2, 1000);
  }
  return 0;
}
