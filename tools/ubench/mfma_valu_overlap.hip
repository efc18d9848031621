// Does vector-ALU work overlap a bf16 MFMA stream on gfx950?  (it does not overlap an fp32 MFMA stream: DESIGN.md section 3)
//   mode 0: MFMA only      mode 1: VALU only      mode 2: both in the same wave, interleaved 1 MFMA : R VALU
//   mode 3: two waves per SIMD, one issues only MFMAs, the other only VALU
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int R>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  const bool doM = MODE == 0 || MODE == 2 || (MODE == 3 && (wave >> 2) == 0);   // waves 0-3 / 4-7: one of each per SIMD
  const bool doV = MODE == 1 || MODE == 2 || (MODE == 3 && (wave >> 2) == 1);
  if (MODE == 3) {                       // roles by wave, each role its own loop
    if ((wave >> 2) == 0) {
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
    } else {
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) v[(u + r) & 7] = __builtin_fmaf(v[(u + r) & 7], seed, 1.0f);
    }
  } else {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (doM) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
      if (doV) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[(u + r) & 7] = __builtin_fmaf(v[(u + r) & 7], seed, 1.0f);
      }
    }
  }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int R>
float run(int threads, int iters) {
  static float* out = nullptr;
  if (!out) hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  return ms;
}
int main() {
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    // one wave per SIMD (256 threads)
    const float m = run<0, 2>(256, iters), v2 = run<1, 2>(256, iters), v4 = run<1, 4>(256, iters);
    const float b2 = run<2, 2>(256, iters), b4 = run<2, 4>(256, iters), b1 = run<2, 1>(256, iters), v1 = run<1, 1>(256, iters);
    printf("1 wave/SIMD, per iteration 8 MFMA 16x16x32 bf16 [+ 8R VALU]:  MFMA %.3f ms | VALU R=1 %.3f R=2 %.3f R=4 %.3f | same wave R=1 %.3f R=2 %.3f R=4 %.3f\n",
           m, v1, v2, v4, b1, b2, b4);
    // two waves per SIMD (512 threads): one MFMA-only, one VALU-only
    const float mm = run<0, 2>(512, iters), c2 = run<3, 2>(512, iters), c4 = run<3, 4>(512, iters), vv4 = run<1, 4>(512, iters);
    printf("2 waves/SIMD: both MFMA %.3f ms | both VALU R=4 %.3f | split roles R=2 %.3f R=4 %.3f\n", mm, vv4, c2, c4);
    const double cyc = m * 1e-3 * 2.4e9 / (iters * 8.0);
    printf("   (MFMA: %.1f cycles each at 2.4 GHz)\n", cyc);
  }
  return 0;
}
