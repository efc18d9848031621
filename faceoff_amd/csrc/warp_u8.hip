// The reference's image perturbations on 8-bit frames, with OpenCV's own fixed-point arithmetic.
//
// TemporalAlignment/perturbations.py perturbs uint8 HxWx3 images on the loader's CPU workers with cv2.warpAffine (:51 :63 :80
// :117), cv2.resize INTER_CUBIC (:88) and cv2.flip (:125); opencv-python is pinned at 4.6.0.66 (environment.yml:70).  These
// kernels do the same on whole stacks of frames [N][H][W][C] in HBM and follow OpenCV 4.6.0's 8-bit algorithm step by step
// (imgwarp.cpp: warpAffine / WarpAffineInvoker / remapBilinear; resize.cpp: HResizeCubic / VResizeCubic / VResizeCubicVec_32s8u),
// so that a frame perturbed here is the frame cv2 would have produced:
//   warp    source coordinates in 1/32 pixel from 10-bit fixed-point rows and columns, four neighbours blended with 15-bit
//           integer weights, neighbours outside the image are 0 (BORDER_CONSTANT)
//   resize  destination size cvRound(w m), scale 1 / m, float cubic weights (A = -0.75) rounded to 11 bits, edge taps
//           clamped, horizontal pass in int; vertical pass in float for the elements of a row OpenCV's 8-lane SIMD loop
//           covers and in int for the tail; followed by the centre crop (m >= 1) or centre paste onto zeros (m < 1) of
//           perturbations.py:89-103 in the same launch
// Every floating-point step is a separate IEEE operation, as in the SSE3-baseline build of the wheel: no contraction in this file.
// The checker is oracle/cv2_oracle.py (parity unpinned: cv2 itself is not installed here).
#pragma clang fp contract(off)
#include <algorithm>
#include <cmath>
#include "common.h"

namespace {

constexpr int kFramesPerLaunch = 64;      // 3 KB of kernel arguments

struct WarpFrames { double m[kFramesPerLaunch][6]; };            // inverse maps (destination -> source)
struct ResizeFrame { double scale; int dw, dh, offx, offy; };    // resized size; resized pixel = output pixel + off
struct ResizeFrames { ResizeFrame f[kFramesPerLaunch]; };

__device__ __forceinline__ int sat_short(int v) { return min(max(v, -32768), 32767); }

// one thread per destination pixel, blockIdx.y = frame of this launch
__global__ void warp_affine_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int C, WarpFrames A) {
  const int f = blockIdx.y;
  const double* m = A.m[f];
  const long long fo = (long long)f * H * W * C;
  const uint8_t* s = src + fo;
  uint8_t* d = dst + fo;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < H * W; p += gridDim.x * blockDim.x) {
    const int x = p % W, y = p / W;
    const int adelta = __double2int_rn(m[0] * x * 1024.0), bdelta = __double2int_rn(m[3] * x * 1024.0);
    const int X0 = __double2int_rn((m[1] * y + m[2]) * 1024.0) + 16, Y0 = __double2int_rn((m[4] * y + m[5]) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5), fx = X & 31, fy = Y & 31;
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }          // BilinearTab_i[0]: saturate_cast<short>(32768), then the sum fix-up
    const bool x0 = (unsigned)sx < (unsigned)W, x1 = (unsigned)(sx + 1) < (unsigned)W;
    const bool y0 = (unsigned)sy < (unsigned)H, y1 = (unsigned)(sy + 1) < (unsigned)H;
    const long long r0 = (long long)sy * W, r1 = r0 + W;
    for (int c = 0; c < C; ++c) {
      const int v00 = (y0 && x0) ? s[(r0 + sx) * C + c] : 0, v01 = (y0 && x1) ? s[(r0 + sx + 1) * C + c] : 0;
      const int v10 = (y1 && x0) ? s[(r1 + sx) * C + c] : 0, v11 = (y1 && x1) ? s[(r1 + sx + 1) * C + c] : 0;
      const int acc = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
      d[(long long)p * C + c] = (uint8_t)min(max((acc + (1 << 14)) >> 15, 0), 255);
    }
  }
}

// interpolateCubic (float) then saturate_cast<short>(c * 2048)
__device__ __forceinline__ void cubic_coef(float x, int* q) {
  const float A = -0.75f;
  const float x1 = x + 1.f;
  float c[4];
  c[0] = ((A * x1 - 5.f * A) * x1 + 8.f * A) * x1 - 4.f * A;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  const float xm = 1.f - x;
  c[2] = ((A + 2.f) * xm - (A + 3.f)) * xm * xm + 1.f;
  c[3] = 1.f - c[0] - c[1] - c[2];
  for (int k = 0; k < 4; ++k) q[k] = sat_short(__float2int_rn(c[k] * 2048.f));
}

__global__ void resize_center_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int C, ResizeFrames A) {
  const int f = blockIdx.y;
  const ResizeFrame R = A.f[f];
  const long long fo = (long long)f * H * W * C;
  const uint8_t* s = src + fo;
  uint8_t* d = dst + fo;
  const int nvec = (R.dw * C) & ~7;                          // elements of a resized row the 8-lane SIMD loop covers
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < H * W; p += gridDim.x * blockDim.x) {
    const int x = p % W, y = p / W;
    const int rx = x + R.offx, ry = y + R.offy;
    if ((unsigned)rx >= (unsigned)R.dw || (unsigned)ry >= (unsigned)R.dh) {
      for (int c = 0; c < C; ++c) d[(long long)p * C + c] = 0;
      continue;
    }
    float fx = (float)((rx + 0.5) * R.scale - 0.5), fy = (float)((ry + 0.5) * R.scale - 0.5);
    const int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx;
    fy -= (float)sy;
    int al[4], be[4];
    cubic_coef(fx, al);
    cubic_coef(fy, be);
    int xs[4];
    long long rows[4];
    for (int j = 0; j < 4; ++j) {
      xs[j] = min(max(sx - 1 + j, 0), W - 1);
      rows[j] = (long long)min(max(sy - 1 + j, 0), H - 1) * W;
    }
    const float sc = 1.f / (2048.f * 2048.f);
    for (int c = 0; c < C; ++c) {
      int hsum[4];
      for (int k = 0; k < 4; ++k) {
        int v = 0;
        for (int j = 0; j < 4; ++j) v += (int)s[(rows[k] + xs[j]) * C + c] * al[j];
        hsum[k] = v;
      }
      int out;
      if (rx * C + c < nvec) {
        float r = (float)hsum[3] * ((float)be[3] * sc);
        r = (float)hsum[2] * ((float)be[2] * sc) + r;
        r = (float)hsum[1] * ((float)be[1] * sc) + r;
        r = (float)hsum[0] * ((float)be[0] * sc) + r;
        out = __float2int_rn(r);
      } else {
        out = (hsum[0] * be[0] + hsum[1] * be[1] + hsum[2] * be[2] + hsum[3] * be[3] + (1 << 21)) >> 22;
      }
      d[(long long)p * C + c] = (uint8_t)min(max(out, 0), 255);
    }
  }
}

__global__ void flip_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int N, int H, int W, int C, int fx, int fy) {
  const long long total = (long long)N * H * W;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const long long n = e / ((long long)W * H);
    const long long q = (n * H + (fy ? H - 1 - y : y)) * W + (fx ? W - 1 - x : x);
    for (int c = 0; c < C; ++c) dst[e * C + c] = src[q * C + c];
  }
}

// transforms.ToTensor() + transforms.Normalize(mean, std): float(v) / 255, then (t - mean) / std, channels-last bytes -> NCHW floats
__global__ void u8_to_norm_nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C, int reverse,
                                       float mean, float stdv) {
  const long long HW = (long long)H * W, total = (long long)N * HW;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long n = e / HW, pix = e % HW;
    for (int c = 0; c < C; ++c) {
      const float t = (float)src[e * C + (reverse ? C - 1 - c : c)] / 255.f;
      dst[(n * C + c) * HW + pix] = (t - mean) / stdv;
    }
  }
}

inline int blocks_for(long long total) { return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 4096)); }

// cv::warpAffine's inversion of a forward map, in double (imgwarp.cpp)
inline void invert_like_cv(const double* Min, double* M) {
  for (int i = 0; i < 6; ++i) M[i] = Min[i];
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11;
  M[1] *= -D;
  M[3] *= -D;
  M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1;
  M[5] = b2;
}

}  // namespace

extern "C" {

int fo_warp_affine_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int C, const double* M_fwd, int per_frame, void* stream) {
  FO_REQUIRE(src && dst && src != dst && M_fwd && N > 0 && H > 0 && W > 0 && C > 0 && C <= 4 && H < 32768 && W < 32768, FO_E_SHAPE,
             "warp_affine_u8: bad arguments");
  const long long frame = (long long)H * W * C;
  for (int n0 = 0; n0 < N; n0 += kFramesPerLaunch) {
    const int nf = std::min(kFramesPerLaunch, N - n0);
    WarpFrames A;
    for (int i = 0; i < nf; ++i) invert_like_cv(M_fwd + (per_frame ? (size_t)(n0 + i) * 6 : 0), A.m[i]);
    hipLaunchKernelGGL(warp_affine_u8_kernel, dim3(blocks_for((long long)H * W), nf), dim3(256), 0, (hipStream_t)stream, src + n0 * frame,
                       dst + n0 * frame, H, W, C, A);
    FO_CHECK_LAUNCH();
  }
  return FO_OK;
}

int fo_resize_center_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int C, const double* magnification, int per_frame, void* stream) {
  FO_REQUIRE(src && dst && src != dst && magnification && N > 0 && H > 0 && W > 0 && C > 0 && C <= 4, FO_E_SHAPE, "resize_center_u8: bad arguments");
  const long long frame = (long long)H * W * C;
  for (int n0 = 0; n0 < N; n0 += kFramesPerLaunch) {
    const int nf = std::min(kFramesPerLaunch, N - n0);
    ResizeFrames A;
    for (int i = 0; i < nf; ++i) {
      const double m = magnification[per_frame ? n0 + i : 0];
      FO_REQUIRE(m > 0, FO_E_SHAPE, "resize_center_u8: magnification must be positive");
      ResizeFrame& R = A.f[i];
      R.dw = (int)std::nearbyint(W * m);                     // saturate_cast<int>(ssize.width * inv_scale_x): round half to even
      R.dh = (int)std::nearbyint(H * m);
      FO_REQUIRE(R.dw > 0 && R.dh > 0 && (long long)R.dw * C < (1 << 30), FO_E_SHAPE, "resize_center_u8: resized image is empty or too wide");
      R.scale = 1. / m;
      if (m >= 1) {                                          // perturbations.py:91-95: crop of size (h, w) about the resized centre
        R.offx = R.dw / 2 - W / 2;
        R.offy = R.dh / 2 - H / 2;
      } else {                                               // :96-103: pasted at ((h - hs) // 2, (w - ws) // 2) onto zeros
        R.offx = -((W - R.dw) / 2);
        R.offy = -((H - R.dh) / 2);
      }
    }
    hipLaunchKernelGGL(resize_center_u8_kernel, dim3(blocks_for((long long)H * W), nf), dim3(256), 0, (hipStream_t)stream, src + n0 * frame,
                       dst + n0 * frame, H, W, C, A);
    FO_CHECK_LAUNCH();
  }
  return FO_OK;
}

int fo_flip_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int C, int flip_code, void* stream) {
  FO_REQUIRE(src && dst && src != dst && N > 0 && H > 0 && W > 0 && C > 0, FO_E_SHAPE, "flip_u8: bad arguments");
  hipLaunchKernelGGL(flip_u8_kernel, dim3(blocks_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, N, H, W, C,
                     flip_code != 0, flip_code <= 0);       // cv2.flip: 0 rows reversed, > 0 columns reversed, < 0 both
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_u8_to_norm_nchw(const uint8_t* src, float* dst, int N, int H, int W, int C, int reverse_channels, float mean, float stdv, void* stream) {
  FO_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && stdv != 0.f, FO_E_SHAPE, "u8_to_norm_nchw: bad arguments");
  hipLaunchKernelGGL(u8_to_norm_nchw_kernel, dim3(blocks_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, N, H, W, C,
                     reverse_channels, mean, stdv);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
