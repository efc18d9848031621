// Input-pipeline and validation helpers on the device:
//  * affine warps of NCHW frame stacks -- the cheap motion perturbations of the reference's data pipeline
//    (TemporalAlignment/perturbations.py:45-105: translate_horizontal / translate_vertical / rotate_image via cv2.warpAffine,
//    resize_image via cv2.resize INTER_CUBIC), which the reference runs per frame on two CPU loader workers;
//  * de-normalisation of model outputs to 8-bit RGB frames for the validation videos
//    (train_faceoff_perceptual.py:71-77 `(x.clamp(-1, 1) + 1) / 2`, utils.py:9-17 `(frame * 255).astype(np.uint8)`).
#include <algorithm>
#include "common.h"

namespace {

struct Affine { float m[6]; };   // source = M * (x_dst, y_dst, 1): the INVERSE map, like cv2.warpAffine applies it

__device__ __forceinline__ float cubic_w(float t, int k) {   // OpenCV's bicubic kernel (A = -0.75), tap k = -1, 0, 1, 2
  const float A = -0.75f;
  switch (k) {
    case -1: return ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
    case 0: return ((A + 2) * t - (A + 3)) * t * t + 1;
    case 1: return ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
    default: return 1.f - (((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A) - (((A + 2) * t - (A + 3)) * t * t + 1) -
                    (((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1);
  }
}

// dst[n][c][y][x] = interpolate(src[n][c], M (x, y, 1)), zero outside (BORDER_CONSTANT 0).  mode 0 bilinear, 1 bicubic.
__global__ void affine_warp_kernel(const float* __restrict__ src, float* __restrict__ dst, int NC, int H, int W, Affine M, int mode) {
  const long long total = (long long)NC * H * W;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const long long plane = e / ((long long)W * H);
    const float* s = src + plane * H * W;
    const float sx = M.m[0] * x + M.m[1] * y + M.m[2], sy = M.m[3] * x + M.m[4] * y + M.m[5];
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = sx - fx, ty = sy - fy;
    auto at = [&](int yy, int xx) -> float { return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? s[(long long)yy * W + xx] : 0.f; };
    float v;
    if (mode == 0) {
      v = (1 - ty) * ((1 - tx) * at(y0, x0) + tx * at(y0, x0 + 1)) + ty * ((1 - tx) * at(y0 + 1, x0) + tx * at(y0 + 1, x0 + 1));
    } else {
      v = 0.f;
      for (int j = -1; j <= 2; ++j) {
        float row = 0.f;
        for (int i = -1; i <= 2; ++i) row += cubic_w(tx, i) * at(y0 + j, x0 + i);
        v += cubic_w(ty, j) * row;
      }
    }
    dst[e] = v;
  }
}

// out[n][y][x][3] (uint8 RGB, or BGR with bgr != 0) = uint8(255 * (clamp(v, -1, 1) + 1) / 2), v from NCHW (ld == 0) or
// channels-last frames with pixel stride ld
__global__ void denorm_u8_kernel(const float* __restrict__ src, int ld, int C, unsigned char* __restrict__ out, int N, int H, int W, int c0, int bgr) {
  const long long HW = (long long)H * W, total = (long long)N * HW;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long pix = e % HW, n = e / HW;
    for (int c = 0; c < 3; ++c) {
      const float v = ld ? src[(n * HW + pix) * ld + c0 + c] : src[(n * C + c0 + c) * HW + pix];
      const float u = (fminf(fmaxf(v, -1.f), 1.f) + 1.f) * 0.5f * 255.f;
      out[e * 3 + (bgr ? 2 - c : c)] = (unsigned char)u;       // truncation, as ndarray.astype(np.uint8)
    }
  }
}

inline int grid_for(long long total) { return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 16384)); }

}  // namespace

extern "C" {

int fo_affine_warp(const float* src, float* dst, int N, int C, int H, int W, const float* M_dst_to_src, int mode, void* stream) {
  FO_REQUIRE(src && dst && src != dst && M_dst_to_src && N > 0 && C > 0 && H > 0 && W > 0 && (mode == 0 || mode == 1), FO_E_SHAPE,
             "affine_warp: bad arguments");
  Affine M;
  for (int i = 0; i < 6; ++i) M.m[i] = M_dst_to_src[i];
  hipLaunchKernelGGL(affine_warp_kernel, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, N * C, H, W, M, mode);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_denorm_u8(const float* src, int ld, int C, int c0, uint8_t* out, int N, int H, int W, int bgr, void* stream) {
  FO_REQUIRE(src && out && N > 0 && H > 0 && W > 0 && c0 >= 0 && (ld > 0 ? c0 + 3 <= ld : c0 + 3 <= C), FO_E_SHAPE, "denorm_u8: bad arguments");
  hipLaunchKernelGGL(denorm_u8_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, ld, C, out, N, H, W, c0, bgr);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
