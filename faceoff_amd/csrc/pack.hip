// Filter packing (checkpoint layout -> K-contiguous GEMM-B rows) and NCHW <-> NHWC transforms.
// The checkpoint layouts are the reference's state_dict shapes.
#include <algorithm>
#include "common.h"

namespace {

// wp[o][t][i] = w[o][i][t]                     (o < O, i < I; zero elsewhere)
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ wp, int O, int I, int taps,
                                 int Opad, int Ipad) {
  const size_t total = (size_t)Opad * taps * Ipad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = e % Ipad;
    const int t = (e / Ipad) % taps;
    const int o = e / ((size_t)Ipad * taps);
    wp[e] = (o < O && i < I) ? w[((size_t)o * I + i) * taps + t] : 0.f;
  }
}

// wp[i][t][o] = w[o][i][taps-1-t]              (stride-1 dgrad: flipped taps, swapped channels)
__global__ void pack_conv_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int O, int I, int taps,
                                       int Opad, int Ipad) {
  const size_t total = (size_t)Ipad * taps * Opad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int o = e % Opad;
    const int t = (e / Opad) % taps;
    const int i = e / ((size_t)Opad * taps);
    wp[e] = (o < O && i < I) ? w[((size_t)o * I + i) * taps + (taps - 1 - t)] : 0.f;
  }
}

// ConvTranspose2d k4 s2 p1, weight w[ci][co][kh][kw].  Output row oy = 2*iy - 1 + kh.
// Phase py = oy & 1 uses two input rows; with tap ty in {0,1} at input row m + ty - padp,
//   py = 0: padp = 1, kh = 3 - 2*ty   (rows m-1, m)
//   py = 1: padp = 0, kh = 2 - 2*ty   (rows m, m+1)
// wp[ph][co][ty*2+tx][ci]
__global__ void pack_convT_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co, int Cipad,
                                  int Copad) {
  const size_t total = (size_t)4 * Copad * 4 * Cipad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = e % Cipad;
    const int t = (e / Cipad) % 4;
    const int co = (e / ((size_t)Cipad * 4)) % Copad;
    const int ph = e / ((size_t)Cipad * 4 * Copad);
    const int py = ph >> 1, px = ph & 1, ty = t >> 1, tx = t & 1;
    const int kh = (py == 0 ? 3 : 2) - 2 * ty;
    const int kw = (px == 0 ? 3 : 2) - 2 * tx;
    wp[e] = (ci < Ci && co < Co) ? w[(((size_t)ci * Co + co) * 4 + kh) * 4 + kw] : 0.f;
  }
}

// Fused 4-phase form of the same transposed conv for Co <= 8: wp[ph*8 + co][(dy+1)*3 + (dx+1)][ci], input offsets
// dy, dx in {-1,0,1}.  Phase py uses dy = -1 (kh 3), 0 (kh 1) when py = 0 and dy = 0 (kh 2), +1 (kh 0) when py = 1.
__global__ void pack_convT_fused_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co, int Cipad) {
  const size_t total = (size_t)32 * 9 * Cipad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = e % Cipad;
    const int t = (e / Cipad) % 9;
    const int col = e / ((size_t)Cipad * 9);
    const int ph = col >> 3, co = col & 7, py = ph >> 1, px = ph & 1;
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const int kh = py == 0 ? (dy == -1 ? 3 : (dy == 0 ? 1 : -1)) : (dy == 0 ? 2 : (dy == 1 ? 0 : -1));
    const int kw = px == 0 ? (dx == -1 ? 3 : (dx == 0 ? 1 : -1)) : (dx == 0 ? 2 : (dx == 1 ? 0 : -1));
    wp[e] = (ci < Ci && co < Co && kh >= 0 && kw >= 0) ? w[(((size_t)ci * Co + co) * 4 + kh) * 4 + kw] : 0.f;
  }
}

// Cell form of the same transposed conv for Co <= 8 (FO_DEPTH2SPACE with ophH = 1): a k2 full correlation over the input grid
// producing 2x2-pixel cells -- cell (i, j), phase (a, b), is output pixel (2i + a - 1, 2j + b - 1) = sum over ei, ej in {0,1} of
// x[i + ei - 1][j + ej - 1] . w[ci][co][2 (1 - ei) + a][2 (1 - ej) + b]:  wp[(a*2+b)*8 + co][ei*2 + ej][ci].  K = 4 Ci instead
// of the 9 Ci of the 3x3 form above (whose filter is 5/9 zeros).
__global__ void pack_convT_cells_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co, int Cpp, int Cipad) {      // Cpp: columns per phase
  const size_t total = (size_t)4 * Cpp * 4 * Cipad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = e % Cipad;
    const int t = (e / Cipad) & 3;
    const int col = e / ((size_t)Cipad * 4);
    const int ph = col / Cpp, co = col - ph * Cpp, a = ph >> 1, b = ph & 1, ei = t >> 1, ej = t & 1;
    wp[e] = (ci < Ci && co < Co) ? w[(((size_t)ci * Co + co) * 4 + 2 * (1 - ei) + a) * 4 + 2 * (1 - ej) + b] : 0.f;
  }
}

// [N,C,H,W] -> [N,H,W,ld] through an LDS tile so both sides are coalesced.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int Cpad, int ld) {
  __shared__ float tile[64][33];  // [pixel][channel] up to 32 channels per pass
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  for (int c0 = 0; c0 < Cpad; c0 += 32) {
    for (int idx = threadIdx.x; idx < 32 * 64; idx += blockDim.x) {
      const int c = idx / 64, p = idx % 64;
      float v = 0.f;
      if (c0 + c < C && p0 + p < HW) v = x[((size_t)n * C + c0 + c) * HW + p0 + p];
      tile[p][c] = v;
    }
    __syncthreads();
    const int cw = min(32, Cpad - c0);
    for (int idx = threadIdx.x; idx < cw * 64; idx += blockDim.x) {
      const int p = idx / cw, c = idx % cw;
      if (p0 + p < HW) y[((size_t)n * HW + p0 + p) * ld + c0 + c] = tile[p][c];
    }
    __syncthreads();
  }
}

// process_data's torch.cat([source, background], axis=channels) (reference utils.py:32) fused with the layout change:
// y[n][p][c] = a[n][c][p] (c < Ca), b[n][c-Ca][p] (Ca <= c < Ca+Cb), 0 (padding).  One pass, 8-channel (32-B) pixels.
__global__ void nchw2_to_nhwc8_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int Ca,
                                      int Cb, int HW, long long npix) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW;
    const int hw = (int)(p - n * HW);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v[c] = 0.f;
      if (c < Ca) v[c] = a[(n * Ca + c) * HW + hw];
      else if (c < Ca + Cb) v[c] = b[(n * Cb + (c - Ca)) * HW + hw];
    }
    *reinterpret_cast<f32x4*>(y + p * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(y + p * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int ld, int accumulate) {
  __shared__ float tile[64][33];
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int cw = min(32, C - c0);
    for (int idx = threadIdx.x; idx < cw * 64; idx += blockDim.x) {
      const int p = idx / cw, c = idx % cw;
      tile[p][c] = (p0 + p < HW) ? x[((size_t)n * HW + p0 + p) * ld + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < cw * 64; idx += blockDim.x) {
      const int c = idx / 64, p = idx % 64;
      if (p0 + p < HW) {
        float* dst = y + ((size_t)n * C + c0 + c) * HW + p0 + p;
        *dst = accumulate ? *dst + tile[p][c] : tile[p][c];
      }
    }
    __syncthreads();
  }
}

inline int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 8192); }

}  // namespace

extern "C" {

int fo_pack_conv(const float* w, float* wp, int O, int I, int taps, int Opad, int Ipad, void* stream) {
  FO_REQUIRE(Opad >= O && Ipad >= I && taps > 0, FO_E_SHAPE, "pack_conv: bad padding");
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid_for((size_t)Opad * taps * Ipad)), dim3(256), 0, (hipStream_t)stream, w, wp, O,
                     I, taps, Opad, Ipad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_pack_conv_dgrad(const float* w, float* wp, int O, int I, int taps, int Opad, int Ipad, void* stream) {
  FO_REQUIRE(Opad >= O && Ipad >= I && taps > 0, FO_E_SHAPE, "pack_conv_dgrad: bad padding");
  hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3(grid_for((size_t)Opad * taps * Ipad)), dim3(256), 0, (hipStream_t)stream, w,
                     wp, O, I, taps, Opad, Ipad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_pack_convT_k4s2(const float* w, float* wp, int Ci, int Co, int Cipad, int Copad, void* stream) {
  FO_REQUIRE(Cipad >= Ci && Copad >= Co, FO_E_SHAPE, "pack_convT: bad padding");
  hipLaunchKernelGGL(pack_convT_kernel, dim3(grid_for((size_t)16 * Copad * Cipad)), dim3(256), 0, (hipStream_t)stream, w, wp,
                     Ci, Co, Cipad, Copad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_pack_convT_k4s2_fused(const float* w, float* wp, int Ci, int Co, int Cipad, void* stream) {
  FO_REQUIRE(Cipad >= Ci && Co <= 8, FO_E_SHAPE, "pack_convT_fused: Co <= 8");
  hipLaunchKernelGGL(pack_convT_fused_kernel, dim3(grid_for((size_t)32 * 9 * Cipad)), dim3(256), 0, (hipStream_t)stream, w, wp,
                     Ci, Co, Cipad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_pack_convT_k4s2_cells(const float* w, float* wp, int Ci, int Co, int Cipad, void* stream) {
  FO_REQUIRE(Cipad >= Ci && Co <= 8, FO_E_SHAPE, "pack_convT_cells: Co <= 8");
  hipLaunchKernelGGL(pack_convT_cells_kernel, dim3(grid_for((size_t)32 * 4 * Cipad)), dim3(256), 0, (hipStream_t)stream, w, wp, Ci, Co,
                     8, Cipad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_pack_convT_k4s2_cells_n(const float* w, float* wp, int Ci, int Co, int Cpp, int Cipad, void* stream) {
  FO_REQUIRE(Cipad >= Ci && Cpp >= Co && Cpp % 8 == 0, FO_E_SHAPE, "pack_convT_cells_n: Cpp >= Co, a multiple of 8");
  hipLaunchKernelGGL(pack_convT_cells_kernel, dim3(grid_for((size_t)4 * Cpp * 4 * Cipad)), dim3(256), 0, (hipStream_t)stream, w, wp, Ci, Co,
                     Cpp, Cipad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, int ldy, void* stream) {
  FO_REQUIRE(Cpad >= C && ldy >= Cpad, FO_E_SHAPE, "nchw_to_nhwc: bad channel padding");
  const int HW = H * W;
  if (Cpad == 8 && ldy == 8 && C <= 8 && fo_aligned16(y)) {   // image tensors: one pixel per lane, no LDS (every access coalesced)
    const long long npix = (long long)N * HW;
    hipLaunchKernelGGL(nchw2_to_nhwc8_kernel, dim3(grid_for((size_t)npix)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y, C,
                       0, HW, npix);
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, Cpad, ldy);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_nchw2_to_nhwc8(const float* a, int Ca, const float* b, int Cb, float* y, int N, int H, int W, void* stream) {
  FO_REQUIRE(Ca > 0 && Cb >= 0 && Ca + Cb <= 8 && fo_aligned16(y), FO_E_SHAPE, "nchw2_to_nhwc8: Ca + Cb <= 8, y 16-byte aligned");
  const long long npix = (long long)N * H * W;
  hipLaunchKernelGGL(nchw2_to_nhwc8_kernel, dim3(grid_for((size_t)npix)), dim3(256), 0, (hipStream_t)stream, a, b, y, Ca, Cb, H * W,
                     npix);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int ldx, int accumulate, void* stream) {
  const int HW = H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, ldx,
                     accumulate);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
