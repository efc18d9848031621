// LPIPS / VGG-16 support kernels (reference models/lpips.py:80-161, loss.py:27-33).  The thirteen 3x3
// convolutions run on the implicit-GEMM conv kernel (conv_igemm.hip); this file holds what sits between
// them: input scaling + layout, 2x2 max-pool forward/backward, and the fused per-tap LPIPS head
// (channel-L2 normalise both feature maps, squared difference, 1x1 `lin` weights, spatial mean) with
// its backward.  All HBM-bound, channels-last, 16 B per lane.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) { return group_sum_valu<64>(v); }

// ScalingLayer (lpips.py:96-103): y[n][h][w][c] = (x_c - shift_c) / scale_c for c < 3, 0 for c in 3..7.
// src is either NCHW [N,3,H,W] (the loader's ground truth) or NHWC with pixel stride ld (the decoder output).
__global__ void lpips_prep_kernel(const float* __restrict__ src, int src_is_nhwc, int ld, float* __restrict__ y, int HW,
                                  long long npix, f32x4 shift, f32x4 inv_scale) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    f32x4 v;
    if (src_is_nhwc) {
      v = *reinterpret_cast<const f32x4*>(src + p * ld);
    } else {
      const long long n = p / HW;
      const int hw = (int)(p - n * HW);
      const float* s = src + n * 3 * (long long)HW + hw;
      v = f32x4{s[0], s[(long long)HW], s[2 * (long long)HW], 0.f};
    }
    f32x4 o = (v - shift) * inv_scale;
    o.w = 0.f;
    *reinterpret_cast<f32x4*>(y + p * 8) = o;
    *reinterpret_cast<f32x4*>(y + p * 8 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// backward of the scaling into the decoder-output gradient: gdec[p][c] += w * g[p][c] / scale_c  (c < 3)
__global__ void lpips_prep_bwd_kernel(const float* __restrict__ g, int ldg, float* __restrict__ gdec, int ldd, long long npix,
                                      f32x4 inv_scale, const float* __restrict__ gscale, float weight) {
  const float k = weight * gscale[0];
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + p * ldg);
    f32x4 d = *reinterpret_cast<f32x4*>(gdec + p * ldd);
    d.x += k * gv.x * inv_scale.x; d.y += k * gv.y * inv_scale.y; d.z += k * gv.z * inv_scale.z;
    *reinterpret_cast<f32x4*>(gdec + p * ldd) = d;
  }
}

// MaxPool2d(2,2) on [N,H,W,C] -> [N,H/2,W/2,C] (torchvision vgg16.features[4,9,16,23]).
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Ho, int Wo, int C4) {
  const long long total = (long long)N * Ho * Wo * C4;
  const int W = Wo * 2;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const f32x4* base = reinterpret_cast<const f32x4*>(x) + ((n * (2 * Ho) + 2 * yo) * W + 2 * xo) * C4 + c;
    const f32x4 a = base[0], b = base[C4], cc = base[(long long)W * C4], d = base[(long long)W * C4 + C4];
    f32x4 m;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(cc[k], d[k]));
    reinterpret_cast<f32x4*>(y)[e] = m;
  }
}

// Backward of pool + the ReLU in front of it, with an optional extra gradient at the pre-pool tensor
// (a LPIPS tap):  gx = relu'(x) * ( [x is the FIRST maximum of its window] * gy + add ).
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ add,
                                    float* __restrict__ gx, int N, int Ho, int Wo, int C4) {
  const long long total = (long long)N * Ho * Wo * C4;
  const int W = Wo * 2;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const long long i00 = ((n * (2 * Ho) + 2 * yo) * W + 2 * xo) * C4 + c;
    const long long idx[4] = {i00, i00 + C4, i00 + (long long)W * C4, i00 + (long long)W * C4 + C4};
    f32x4 v[4], o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = reinterpret_cast<const f32x4*>(x)[idx[t]];
    const f32x4 g = reinterpret_cast<const f32x4*>(gy)[e];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float m = fmaxf(fmaxf(v[0][k], v[1][k]), fmaxf(v[2][k], v[3][k]));
      bool taken = false;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool is = !taken && v[t][k] == m;
        taken = taken || is;
        o[t][k] = is ? g[k] : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 r = o[t];
      if (add) r += reinterpret_cast<const f32x4*>(add)[idx[t]];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = v[t][k] > 0.f ? r[k] : 0.f;
      reinterpret_cast<f32x4*>(gx)[idx[t]] = r;
    }
  }
}

// LPIPS head for one tap (lpips.py:85-89,155-161): one wave per pixel, lane holds CPL = C/64 channels.
//   n0 = f0 / (||f0|| + eps), n1 = f1 / (||f1|| + eps);  pix[p] = sum_c lin_c (n0_c - n1_c)^2; lpips_pix_finish_kernel: val[n] += mean over the
//   frame's pixels, added in pixel order (no float atomics: the printed loss is reproducible bit for bit)
template <int CPL>
__global__ void lpips_head_fwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ lin,
                                      float* __restrict__ pix, long long npix) {
  constexpr int C = CPL * 64;
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  float w[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) w[k] = lin[lane * CPL + k];
  for (long long p = wave; p < npix; p += nwaves) {
    float a[CPL], b[CPL], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      a[k] = f0[p * C + lane * CPL + k];
      b[k] = f1[p * C + lane * CPL + k];
      sa = fmaf(a[k], a[k], sa);
      sb = fmaf(b[k], b[k], sb);
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) { const float r = a[k] * ia - b[k] * ib; acc = fmaf(w[k] * r, r, acc); }
    acc = wave_sum(acc);
    if (lane == 0) pix[p] = acc;
  }
}

__global__ __launch_bounds__(64) void lpips_pix_finish_kernel(const float* __restrict__ pix, float* __restrict__ val, int HW, float inv_hw) {
  const long long n = blockIdx.x;
  float t = 0.f;
  for (int i = threadIdx.x; i < HW; i += 64) t += pix[n * HW + i];
  t = wave_sum(t);
  if (threadIdx.x == 0) val[n] += t * inv_hw;
}

// Gradient wrt f1 (the reconstruction branch), through the normalisation and through f1's own ReLU:
//   r = n1 - n0, gn_c = gscale * 2 lin_c r_c / (H*W),  gf_j = relu'(f1_j) * ( gn_j / (s+eps) - f1_j (gn . f1) / (s (s+eps)^2) )
// (the reference's autograd yields NaN where a pixel's feature vector is all zero -- sqrt'(0); we emit 0).
template <int CPL>
__global__ void lpips_head_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ lin,
                                      const float* __restrict__ gscale, float* __restrict__ gf1, long long npix, float k_scale) {
  constexpr int C = CPL * 64;
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const float gk = gscale[0] * k_scale;
  float w[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) w[k] = lin[lane * CPL + k];
  for (long long p = wave; p < npix; p += nwaves) {
    float a[CPL], b[CPL], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      a[k] = f0[p * C + lane * CPL + k];
      b[k] = f1[p * C + lane * CPL + k];
      sa = fmaf(a[k], a[k], sa);
      sb = fmaf(b[k], b[k], sb);
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float na = sqrtf(sa), nb = sqrtf(sb);
    const float ia = 1.f / (na + 1e-10f), ib = 1.f / (nb + 1e-10f);
    float gn[CPL], dot = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      gn[k] = gk * 2.f * w[k] * (b[k] * ib - a[k] * ia);
      dot = fmaf(gn[k], b[k], dot);
    }
    dot = wave_sum(dot);
    const float c2 = nb > 0.f ? dot * ib * ib / nb : 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const float g = gn[k] * ib - b[k] * c2;
      gf1[p * C + lane * CPL + k] = b[k] > 0.f ? g : 0.f;
    }
  }
}

inline int grid_for(long long total, int cap = 4096) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, cap));
}

}  // namespace

extern "C" {

int fo_lpips_prep(const float* src, int src_is_nhwc, int ld, float* y, int N, int H, int W, const float* shift3,
                  const float* scale3, void* stream) {
  FO_REQUIRE(!src_is_nhwc || ld % 4 == 0, FO_E_ALIGN, "lpips_prep: ld %% 4");
  const f32x4 sh = {shift3[0], shift3[1], shift3[2], 0.f};
  const f32x4 is = {1.f / scale3[0], 1.f / scale3[1], 1.f / scale3[2], 0.f};
  const long long npix = (long long)N * H * W;
  hipLaunchKernelGGL(lpips_prep_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, src_is_nhwc, ld, y, H * W,
                     npix, sh, is);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_lpips_prep_bwd(const float* g, int ldg, float* gdec, int ldd, int64_t npix, const float* scale3, const float* gscale,
                      float weight, void* stream) {
  FO_REQUIRE(ldg % 4 == 0 && ldd % 4 == 0, FO_E_ALIGN, "lpips_prep_bwd: ld %% 4");
  const f32x4 is = {1.f / scale3[0], 1.f / scale3[1], 1.f / scale3[2], 0.f};
  hipLaunchKernelGGL(lpips_prep_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, g, ldg, gdec, ldd,
                     (long long)npix, is, gscale, weight);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, FO_E_SHAPE, "maxpool2: even H, W and C %% 4 == 0");
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 4), 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, y, N, H / 2, W / 2, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_maxpool2_bwd(const float* x, const float* gy, const float* add, float* gx, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, FO_E_SHAPE, "maxpool2: even H, W and C %% 4 == 0");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 4), 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, gy, add, gx, N, H / 2, W / 2, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int64_t fo_lpips_tap_ws_bytes(int N, int H, int W) { return (int64_t)N * H * W * 4 + 1024; }      // one float per pixel

int fo_lpips_tap_fwd(const float* f0, const float* f1, const float* lin, float* val, int N, int H, int W, int C, float* ws, void* stream) {
  FO_REQUIRE(ws && (C == 64 || C == 128 || C == 256 || C == 512), FO_E_SHAPE, "lpips_tap: C must be 64/128/256/512 (got %d), and a workspace", C);
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix * 64, 4096);
#define FO_HEAD_FWD(CPL_) hipLaunchKernelGGL(lpips_head_fwd_kernel<CPL_>, dim3(grid), dim3(256), 0, (hipStream_t)stream, f0, f1, lin, ws, npix)
  if (C == 64) FO_HEAD_FWD(1);
  else if (C == 128) FO_HEAD_FWD(2);
  else if (C == 256) FO_HEAD_FWD(4);
  else FO_HEAD_FWD(8);
#undef FO_HEAD_FWD
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(lpips_pix_finish_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, ws, val, H * W, 1.f / (float)(H * W));
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_lpips_tap_bwd(const float* f0, const float* f1, const float* lin, const float* gscale, float* gf1, int N, int H, int W,
                     int C, void* stream) {
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix * 64, 4096);
  const float ks = 1.f / ((float)(H * W) * (float)N);   // spatial mean (lpips.py:160-161) and VQLPIPS .mean() over N (loss.py:33)
#define FO_HEAD_BWD(CPL_) \
  hipLaunchKernelGGL(lpips_head_bwd_kernel<CPL_>, dim3(grid), dim3(256), 0, (hipStream_t)stream, f0, f1, lin, gscale, gf1, npix, ks)
  if (C == 64) FO_HEAD_BWD(1);
  else if (C == 128) FO_HEAD_BWD(2);
  else if (C == 256) FO_HEAD_BWD(4);
  else if (C == 512) FO_HEAD_BWD(8);
  else FO_REQUIRE(false, FO_E_SHAPE, "lpips_tap: C must be 64/128/256/512 (got %d)", C);
#undef FO_HEAD_BWD
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
