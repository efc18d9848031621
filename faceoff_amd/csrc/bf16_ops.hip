// Element-wise companions of the bf16-operand VQ-VAE step (conv_bf16.hip, wgrad_bf16.hip): conversions between the fp32 tensors
// that stay fp32 (quantiser inputs / outputs, decoder output and its gradient, packed filters) and the bf16 tensors the convolution
// kernels read, the input layout kernel, and the straight-through backward of Quantize with bf16 gradients.  All HBM-bound, 16 bytes
// per lane.
#include <algorithm>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// y[r][c] = bf16(x[r][c]), c < C (a multiple of 8); row strides in elements
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, long long ldx, __bf16* __restrict__ y, long long ldy, long long rows, int C8) {
  const long long total = rows * C8;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C8;
    const int c = (int)(e - r * C8) * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4);
    const bf16x8 o = {(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
    *reinterpret_cast<bf16x8*>(y + r * ldy + c) = o;
  }
}

__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ x, long long ldx, float* __restrict__ y, long long ldy, long long rows, int C8) {
  const long long total = rows * C8;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C8;
    const int c = (int)(e - r * C8) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ldx + c);
    *reinterpret_cast<f32x4*>(y + r * ldy + c) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<f32x4*>(y + r * ldy + c + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  }
}

// process_data's torch.cat([source, background], channels) (reference utils.py:32) fused with the layout change and the rounding of
// the network input: y[n][p][0..8) = bf16 of a[n][:, p], b[n][:, p], zeros -- one 16-byte pixel per lane
__global__ void nchw2_to_nhwc8_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, __bf16* __restrict__ y, int Ca, int Cb, int HW,
                                           long long npix) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW;
    const int hw = (int)(p - n * HW);
    bf16x8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float f = 0.f;
      if (c < Ca) f = a[(n * Ca + c) * HW + hw];
      else if (c < Ca + Cb) f = b[(n * Cb + (c - Ca)) * HW + hw];
      v[c] = (__bf16)f;
    }
    *reinterpret_cast<bf16x8*>(y + p * 8) = v;
  }
}

// Quantize backward with the straight-through gradient stored as bf16 (reference :77-78 under autograd):
//   gx = bf16( gq + gdiff * scale * (x - q) ),  gq bf16, x and q fp32 (the quantiser runs in fp32), 64 channels per vector
__global__ void vq_bwd_bf16_kernel(const __bf16* __restrict__ gq, int ldg, const float* __restrict__ x, int ldx, const float* __restrict__ q, int ldq,
                                   const float* __restrict__ gdiff, float scale, __bf16* __restrict__ gx, int ldgx, long long nvec) {
  const float gs = gdiff[0] * scale;
  const long long total = nvec * 8;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long v = e >> 3;
    const int c = (int)(e & 7) * 8;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gq + v * ldg + c);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + v * ldx + c), x1 = *reinterpret_cast<const f32x4*>(x + v * ldx + c + 4);
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(q + v * ldq + c), q1 = *reinterpret_cast<const f32x4*>(q + v * ldq + c + 4);
    const f32x4 d0 = gs * (x0 - q0), d1 = gs * (x1 - q1);
    const bf16x8 o = {(__bf16)((float)g[0] + d0.x), (__bf16)((float)g[1] + d0.y), (__bf16)((float)g[2] + d0.z), (__bf16)((float)g[3] + d0.w),
                      (__bf16)((float)g[4] + d1.x), (__bf16)((float)g[5] + d1.y), (__bf16)((float)g[6] + d1.z), (__bf16)((float)g[7] + d1.w)};
    *reinterpret_cast<bf16x8*>(gx + v * ldgx + c) = o;
  }
}

inline int grid_for(long long total) { return (int)std::min<long long>((total + 255) / 256, 16384); }

}  // namespace

extern "C" {

int fo_f32_to_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, void* stream) {
  FO_REQUIRE(x && y && rows > 0 && C > 0 && C % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0 && fo_aligned16(x) && fo_aligned16(y), FO_E_ALIGN,
             "f32_to_bf16: C %% 8 == 0 and 16-byte aligned rows");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, reinterpret_cast<__bf16*>(y),
                     (long long)ldy, (long long)rows, C / 8);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_bf16_to_f32(const void* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, void* stream) {
  FO_REQUIRE(x && y && rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 4 == 0 && fo_aligned16(x) && fo_aligned16(y), FO_E_ALIGN,
             "bf16_to_f32: C %% 8 == 0 and 16-byte aligned rows");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(x), (long long)ldx, y,
                     (long long)ldy, (long long)rows, C / 8);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_nchw2_to_nhwc8_bf16(const float* a, int Ca, const float* b, int Cb, void* y, int N, int H, int W, void* stream) {
  FO_REQUIRE(a && y && Ca >= 1 && Cb >= 0 && Ca + Cb <= 8 && (Cb == 0 || b) && N > 0 && H > 0 && W > 0 && fo_aligned16(y), FO_E_SHAPE,
             "nchw2_to_nhwc8_bf16: at most 8 channels in all");
  const long long npix = (long long)N * H * W;
  hipLaunchKernelGGL(nchw2_to_nhwc8_bf16_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, a, b, reinterpret_cast<__bf16*>(y), Ca, Cb, H * W, npix);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_vq_bwd_bf16(const void* gq, int ldg, const float* x, int ldx, const float* q, int ldq, const float* gdiff, float scale, void* gx, int ldgx,
                   int64_t nvec, void* stream) {
  FO_REQUIRE(gq && x && q && gdiff && gx && nvec > 0 && ldg % 8 == 0 && ldgx % 8 == 0 && ldx % 4 == 0 && ldq % 4 == 0, FO_E_ALIGN, "vq_bwd_bf16: alignment");
  hipLaunchKernelGGL(vq_bwd_bf16_kernel, dim3(grid_for(nvec * 8)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(gq), ldg, x, ldx, q, ldq,
                     gdiff, scale, reinterpret_cast<__bf16*>(gx), ldgx, (long long)nvec);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
