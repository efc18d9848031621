// HBM-bound kernels of the step: losses, bias gradients, optimiser, small utilities.
// All are grid-stride, 16 B per lane where the layout allows it.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) { return group_sum_valu<64>(v); }      // (the xor butterfly 32, 16, .., 1 on the vector ALU: common.h)

// fo_selftest_lane_moves: lane_xor<K> (common.h) against __shfl_xor for every K, on one wave of distinct values: bad[lane] = bit mask of the K that differ
__global__ __launch_bounds__(64) void lane_moves_selftest_kernel(int* __restrict__ bad) {
  const float v = 1.5f + 0.37f * (float)threadIdx.x;
  const int i = 1000 + 17 * (int)threadIdx.x;
  int b = 0;
  b |= (lane_xor<1>(v) != __shfl_xor(v, 1) || lane_xor<1>(i) != __shfl_xor(i, 1)) << 0;
  b |= (lane_xor<2>(v) != __shfl_xor(v, 2) || lane_xor<2>(i) != __shfl_xor(i, 2)) << 1;
  b |= (lane_xor<4>(v) != __shfl_xor(v, 4) || lane_xor<4>(i) != __shfl_xor(i, 4)) << 2;
  b |= (lane_xor<8>(v) != __shfl_xor(v, 8) || lane_xor<8>(i) != __shfl_xor(i, 8)) << 3;
  b |= (lane_xor<16>(v) != __shfl_xor(v, 16) || lane_xor<16>(i) != __shfl_xor(i, 16)) << 4;
  b |= (lane_xor<32>(v) != __shfl_xor(v, 32) || lane_xor<32>(i) != __shfl_xor(i, 32)) << 5;
  float t = v, u = v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  b |= (group_sum_valu<64>(u) != t) << 6;
  t = v;
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) t += __shfl_xor(t, o);
  b |= (group_sum_valu<8>(u) != t) << 7;
  bad[threadIdx.x] = b;
}

// A workgroup's share of a loss sum without float atomics: wave sums (shuffle tree), added in wave order by thread 0, written to part[blockIdx.x];
// ordered_sum_kernel then adds the partials in a fixed order.  The printed loss is the same bits run after run.
__device__ __forceinline__ void block_partial(float s, float* __restrict__ part) {
  __shared__ float wsum[16];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
    part[blockIdx.x] = t;
  }
}

// out[0] = part[0] + ... + part[n-1]: lane i adds part[i], part[i + 64], ... in order, then a shuffle tree.  One wave.
__global__ __launch_bounds__(64) void ordered_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) t += part[i];
  t = wave_sum(t);
  if (threadIdx.x == 0) out[0] = t;
}

// criterion = nn.MSELoss() on out[:, :3] (train_faceoff_perceptual.py:21,37-39): dec is NHWC (ld floats
// per pixel), gt is the loader's NCHW tensor.  One partial sum of squares per workgroup (block_partial).
__global__ void mse_slice_fwd_kernel(const float* __restrict__ dec, int ldd, const float* __restrict__ gt, int HW,
                                     long long npix, int C3, float* __restrict__ part) {
  float s = 0.f;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW;
    const int hw = (int)(p - n * HW);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dec + p * ldd);
    const float* g = gt + (n * C3) * (long long)HW + hw;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < C3) { const float e = d[c] - g[(long long)c * HW]; s = fmaf(e, e, s); }
  }
  block_partial(s, part);
}

__global__ void mse_slice_bwd_kernel(const float* __restrict__ dec, int ldd, const float* __restrict__ gt, int HW,
                                     long long npix, int C3, const float* __restrict__ gscale, float inv_numel,
                                     float* __restrict__ gdec, int ldg) {
  const float k = 2.f * inv_numel * gscale[0];
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW;
    const int hw = (int)(p - n * HW);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dec + p * ldd);
    const float* g = gt + (n * C3) * (long long)HW + hw;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < C3) o[c] = k * (d[c] - g[(long long)c * HW]);
    *reinterpret_cast<f32x4*>(gdec + p * ldg) = o;
    for (int c = 4; c < ldg; c += 4) *reinterpret_cast<f32x4*>(gdec + p * ldg + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Both at once (the training step needs the loss value AND its gradient; neither depends on the other): one pass over dec and gt.
__global__ void mse_slice_fwd_bwd_kernel(const float* __restrict__ dec, int ldd, const float* __restrict__ gt, int HW, long long npix, int C3,
                                         const float* __restrict__ gscale, float inv_numel, float* __restrict__ gdec, int ldg, float* __restrict__ part) {
  const float k = 2.f * inv_numel * gscale[0];
  float s = 0.f;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW;
    const int hw = (int)(p - n * HW);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dec + p * ldd);
    const float* g = gt + (n * C3) * (long long)HW + hw;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < C3) { const float e = d[c] - g[(long long)c * HW]; s = fmaf(e, e, s); o[c] = k * e; }
    *reinterpret_cast<f32x4*>(gdec + p * ldg) = o;
    for (int c = 4; c < ldg; c += 4) *reinterpret_cast<f32x4*>(gdec + p * ldg + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  block_partial(s, part);
}

// column sums of g[M][ld] (first C channels): stage 1 -> ws[block][C], stage 2 sums blocks in order
__global__ void colsum_stage1(const float* __restrict__ g, float* __restrict__ ws, long long M, int C, int ld) {
  __shared__ f32x4 red[256];
  const int cg = C / 4;  // float4 column groups
  const int col = threadIdx.x % cg, rl = threadIdx.x / cg, nrl = blockDim.x / cg;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (rl < nrl)
    for (long long m = (long long)blockIdx.x * nrl + rl; m < M; m += (long long)gridDim.x * nrl)
      s += *reinterpret_cast<const f32x4*>(g + m * ld + col * 4);
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < cg) {
    f32x4 t = red[threadIdx.x];
    for (int r = 1; r < nrl; ++r) t += red[threadIdx.x + r * cg];
    *reinterpret_cast<f32x4*>(ws + (size_t)blockIdx.x * C + threadIdx.x * 4) = t;
  }
}
__global__ void colsum_stage2(const float* __restrict__ ws, float* __restrict__ out, int nblk, int C, int Creal) {
  __shared__ float red[1024];   // 64 channels x 16 partial-sum lanes (four running sums each: four loads in flight), combined in a fixed order
  const int cl = threadIdx.x >> 6, li = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + li;
  float s = 0.f;
  if (c < Creal) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = cl;
    for (; b + 48 < nblk; b += 64) {
      s += ws[(size_t)b * C + c]; s1 += ws[(size_t)(b + 16) * C + c]; s2 += ws[(size_t)(b + 32) * C + c]; s3 += ws[(size_t)(b + 48) * C + c];
    }
    for (; b < nblk; b += 16) s += ws[(size_t)b * C + c];
    s = (s + s1) + (s2 + s3);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (cl == 0 && c < Creal) {
    float t = red[li];
    for (int k = 1; k < 16; ++k) t += red[k * 64 + li];
    out[c] = t;
  }
}

// torch.optim.Adam (defaults; train_faceoff_perceptual.py:190) over a flat arena.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n4, float lr_over_bc1, float b1, float b2, float eps, float inv_sqrt_bc2, float gscale) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[e] * gscale;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[e], vv = reinterpret_cast<f32x4*>(v)[e], pv = reinterpret_cast<f32x4*>(p)[e];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mv[c] = mv[c] * b1 + gv[c] * (1.f - b1);
      vv[c] = vv[c] * b2 + gv[c] * gv[c] * (1.f - b2);
      const float denom = sqrtf(vv[c]) * inv_sqrt_bc2 + eps;
      pv[c] = pv[c] - lr_over_bc1 * (mv[c] / denom);
    }
    reinterpret_cast<f32x4*>(m)[e] = mv;
    reinterpret_cast<f32x4*>(v)[e] = vv;
    reinterpret_cast<f32x4*>(p)[e] = pv;
  }
}

__global__ void zero_kernel(float* p, long long n) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) p[e] = 0.f;
}

__global__ void relu_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long rows, int C4) {
  const long long total = rows * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C4;
    const int c = (int)(e % C4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<f32x4*>(y + r * ldy + c) = v;
  }
}

__global__ void add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ y,
                           int ldy, long long rows, int C4) {
  const long long total = rows * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C4;
    const int c = (int)(e % C4) * 4;
    *reinterpret_cast<f32x4*>(y + r * ldy + c) =
        *reinterpret_cast<const f32x4*>(a + r * lda + c) + *reinterpret_cast<const f32x4*>(b + r * ldb + c);
  }
}

inline int grid_for(long long total, int cap = 4096) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, cap));
}

}  // namespace

// out[0] = the sum of n partials in a fixed order (internal: the finish launch of every loss sum in this library)
int fo_ordered_sum(const float* part, int n, float* out, void* stream) {
  hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, n, out);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

extern "C" {

int fo_mse_slice_fwd(const float* dec, int ldd, const float* gt_nchw, int N, int H, int W, int C3, float* sum, float* ws, void* stream) {
  FO_REQUIRE(C3 <= 3 && ldd % 4 == 0 && ws && sum, FO_E_SHAPE, "mse: at most 3 channels, ld %% 4 == 0, a workspace of FO_LOSS_WS_BYTES");
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix, 2048);
  hipLaunchKernelGGL(mse_slice_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dec, ldd, gt_nchw, H * W, npix, C3, ws);
  FO_CHECK_LAUNCH();
  return fo_ordered_sum(ws, grid, sum, stream);
}

int fo_mse_slice_bwd(const float* dec, int ldd, const float* gt_nchw, int N, int H, int W, int C3, const float* gscale,
                     float inv_numel, float* gdec, int ldg, void* stream) {
  FO_REQUIRE(C3 <= 3 && ldd % 4 == 0 && ldg % 4 == 0, FO_E_SHAPE, "mse: at most 3 channels, ld %% 4 == 0");
  const long long npix = (long long)N * H * W;
  hipLaunchKernelGGL(mse_slice_bwd_kernel, dim3(grid_for(npix, 4096)), dim3(256), 0, (hipStream_t)stream, dec, ldd, gt_nchw,
                     H * W, npix, C3, gscale, inv_numel, gdec, ldg);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_mse_slice_fwd_bwd(const float* dec, int ldd, const float* gt_nchw, int N, int H, int W, int C3, const float* gscale,
                         float inv_numel, float* gdec, int ldg, float* sum, float* ws, void* stream) {
  FO_REQUIRE(C3 <= 3 && ldd % 4 == 0 && ldg % 4 == 0 && ws && sum, FO_E_SHAPE, "mse: at most 3 channels, ld %% 4 == 0, a workspace of FO_LOSS_WS_BYTES");
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix, 4096);
  hipLaunchKernelGGL(mse_slice_fwd_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dec, ldd, gt_nchw, H * W, npix,
                     C3, gscale, inv_numel, gdec, ldg, ws);
  FO_CHECK_LAUNCH();
  return fo_ordered_sum(ws, grid, sum, stream);
}

int fo_bias_grad(const float* g, float* dbias, int64_t M, int C, int Creal, int ld, float* ws, void* stream) {
  FO_REQUIRE(C % 4 == 0 && C / 4 <= 256 && Creal <= C && ld % 4 == 0, FO_E_SHAPE, "bias_grad: C %% 4 == 0, C <= 1024");
  const int nrl = 256 / (C / 4);
  const int nblk = (int)std::max<long long>(1, std::min<long long>(1024, (M + nrl * 8 - 1) / (nrl * 8)));
  hipLaunchKernelGGL(colsum_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, ws, (long long)M, C, ld);
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_stage2, dim3((Creal + 63) / 64), dim3(1024), 0, (hipStream_t)stream, ws, dbias, nblk, C, Creal);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

/* diagnostic: bad64[lane] = 0 where the vector-ALU lane moves of common.h (lane_xor<1..32>, group_sum_valu) return what __shfl_xor returns */
int fo_selftest_lane_moves(int32_t* bad64, void* stream) {
  FO_REQUIRE(bad64, FO_E_SHAPE, "selftest_lane_moves: null output");
  hipLaunchKernelGGL(lane_moves_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bad64);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

}  // extern "C"

// out[c] = sum over the nblk partial rows ws[b][C] (c < Creal): the second stage of every column-sum in this library (internal)
int fo_colsum_finish(const float* ws, float* out, int nblk, int C, int Creal, void* stream) {
  hipLaunchKernelGGL(colsum_stage2, dim3((Creal + 63) / 64), dim3(1024), 0, (hipStream_t)stream, ws, out, nblk, C, Creal);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

extern "C" {
int fo_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float bias_corr1, float bias_corr2, float grad_scale, void* stream) {
  FO_REQUIRE(n % 4 == 0, FO_E_ALIGN, "adam: arena length must be a multiple of 4");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)(n / 4), lr / bias_corr1, beta1, beta2, eps, 1.f / sqrtf(bias_corr2), grad_scale);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_zero(float* p, int64_t n, void* stream) {
  hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, p, (long long)n);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_relu(const float* x, int ldx, float* y, int ldy, int64_t rows, int C, void* stream) {
  FO_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, FO_E_ALIGN, "relu: C/ld %% 4");
  hipLaunchKernelGGL(relu_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                     (long long)rows, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int64_t rows, int C, void* stream) {
  FO_REQUIRE(C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldy % 4 == 0, FO_E_ALIGN, "add: C/ld %% 4");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, y, ldy,
                     (long long)rows, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
