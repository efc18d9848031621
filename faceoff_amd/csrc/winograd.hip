// Winograd F(2x2, 3x3) for the Conv3d k3 p1 s1 layers (reference models/vqvae_conv3d_latent.py:181,185 and their data
// gradients): the two spatial dimensions of every depth tap are transformed, the depth taps and the channels stay an
// ordinary contraction:
//
//   V[xi][n][tile][ci]   = (B^T d B)[xi]                      d = 4x4 input patch of output tile `tile` of frame n
//   U[xi][co][kd][ci]    = (G g_kd G^T)[xi]                    g_kd = 3x3 slice of the filter
//   M[xi][n][tile][co]   = sum_{kd,ci} U[xi][co][kd][ci] * V[xi][n + kd - 1][tile][ci]      (16 independent GEMMs)
//   out[n][2ty+a][2tx+b] = (A^T M A)[a][b] + bias, mask, residual, ReLU
//
// 16 multiplies per 2x2 output pixels instead of 36: 2.25x fewer MFMA FLOP.  The 16 GEMMs are exactly a Conv3d with a
// (3,1,1) filter over the plane stack V -- the implicit-GEMM kernel of conv_igemm.hip runs them (clip-padding taps
// skipped as usual), picking the filter bank from the frame index (fo_conv_igemm_banked).  This file holds the three
// HBM-bound transforms; all are channels-last, 16 B per lane, exact +-1 / 0.5 arithmetic in fp32.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// U[xi][o][kd][i] (rows padded to Opad / Ipad with zeros), xi = 4*r + c.
// dgrad = 1: the filter of the data gradient, w'[i][o][kd][p][q] = w[o][i][KD-1-kd][2-p][2-q] (roles of o and i swapped).
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int KD, int Opad, int Ipad,
                                   int dgrad) {
  const int rowsOut = dgrad ? I : O, colsIn = dgrad ? O : I;     // GEMM rows (output channels) / K columns of this bank
  const size_t per = (size_t)Opad * KD * Ipad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = e % Ipad;
    const int kd = (e / Ipad) % KD;
    const int co = e / ((size_t)Ipad * KD);
    float g[3][3];
    const bool real = co < rowsOut && ci < colsIn;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float v = 0.f;
        if (real) {
          if (!dgrad) v = w[((((size_t)co * I + ci) * KD + kd) * 3 + p) * 3 + q];
          else v = w[((((size_t)ci * I + co) * KD + (KD - 1 - kd)) * 3 + (2 - p)) * 3 + (2 - q)];
        }
        g[p][q] = v;
      }
    // t = G g  (4x3), u = t G^T (4x4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      t[0][q] = g[0][q];
      t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
      t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
      t[3][q] = g[2][q];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
      U[(size_t)(4 * r + 0) * per + e] = u0;
      U[(size_t)(4 * r + 1) * per + e] = u1;
      U[(size_t)(4 * r + 2) * per + e] = u2;
      U[(size_t)(4 * r + 3) * per + e] = u3;
    }
  }
}

// V[xi][n][ty][tx][c] = (B^T d B)[xi];  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].  One thread: one tile, 4 channels.
__global__ void wino_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, int N, int H, int W, int C4) {
  const int Ht = H >> 1, Wt = W >> 1;
  const long long total = (long long)N * Ht * Wt * C4;
  const size_t plane = (size_t)N * Ht * Wt * C4 * 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 2 * ty - 1 + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int xx = 2 * tx - 1 + s;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        d[r][s] = ok ? ld4(x + ((n * H + y) * (long long)W + xx) * ldx + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 t[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {       // rows: t = B^T d
      t[0][s] = d[0][s] - d[2][s];
      t[1][s] = d[1][s] + d[2][s];
      t[2][s] = d[2][s] - d[1][s];
      t[3][s] = d[1][s] - d[3][s];
    }
    float* dst = V + (size_t)e * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {       // columns: v = t B
      st4(dst + (size_t)(4 * r + 0) * plane, t[r][0] - t[r][2]);
      st4(dst + (size_t)(4 * r + 1) * plane, t[r][1] + t[r][2]);
      st4(dst + (size_t)(4 * r + 2) * plane, t[r][2] - t[r][1]);
      st4(dst + (size_t)(4 * r + 3) * plane, t[r][1] - t[r][3]);
    }
  }
}

// out = epilogue(A^T M A);  A^T = [[1,1,1,0],[0,1,-1,-1]].  One thread: one tile (2x2 output pixels), 4 channels.
// epilogue order as in the conv kernels: (+ bias) -> ReLU mask -> + residual -> ReLU.
__global__ void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ mask,
                                   int ldMask, const float* __restrict__ add, int ldAdd, float* __restrict__ out, int ldOut, int N,
                                   int H, int W, int C4, int flags) {
  const int Ht = H >> 1, Wt = W >> 1;
  const long long total = (long long)N * Ht * Wt * C4;
  const size_t plane = (size_t)N * Ht * Wt * C4 * 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    const float* src = M + (size_t)e * 4;
    f32x4 t[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {       // rows: t = A^T m
      const f32x4 m0 = ld4(src + (size_t)(0 + s) * plane), m1 = ld4(src + (size_t)(4 + s) * plane),
                  m2 = ld4(src + (size_t)(8 + s) * plane), m3 = ld4(src + (size_t)(12 + s) * plane);
      t[0][s] = m0 + m1 + m2;
      t[1][s] = m1 - m2 - m3;
    }
    const f32x4 bv = (flags & FO_BIAS) ? ld4(bias + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 v = (b == 0 ? t[a][0] + t[a][1] + t[a][2] : t[a][1] - t[a][2] - t[a][3]) + bv;
        const long long pix = (n * H + 2 * ty + a) * (long long)W + 2 * tx + b;
        if (flags & FO_MASK) {
          const f32x4 mk = ld4(mask + pix * ldMask + c * 4);
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        if (flags & FO_ADD) v += ld4(add + pix * ldAdd + c * 4);
        if (flags & FO_OUT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        st4(out + pix * ldOut + c * 4, v);
      }
  }
}

// Filter gradient in the transformed domain.  With Y = A^T [U . V] A the gradient of U is (A dY A^T) . V summed over tiles:
//   dM[xi][n][tile][co] = (A dY A^T)[xi]                              A = [[1,0],[1,1],[1,-1],[0,-1]]
//   dU[xi][co][ci][kd]  = sum_{n,tile} dM[xi][n][tile][co] * V[xi][n + kd - 1][tile][ci]     (16 wgrad GEMMs, banked)
//   dW[co][ci][kd]      = G^T dU G                                    (4x4 -> 3x3)
__global__ void wino_gradout_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dM, int N, int H, int W, int C4) {
  const int Ht = H >> 1, Wt = W >> 1;
  const long long total = (long long)N * Ht * Wt * C4;
  const size_t plane = (size_t)N * Ht * Wt * C4 * 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    const float* base = g + ((n * H + 2 * ty) * (long long)W + 2 * tx) * ldg + c * 4;
    const f32x4 y00 = ld4(base), y01 = ld4(base + ldg), y10 = ld4(base + (long long)W * ldg), y11 = ld4(base + (long long)W * ldg + ldg);
    // rows: r = A y  (4x2)
    const f32x4 r[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {-y10, -y11}};
    float* dst = dM + (size_t)e * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // columns: m = r A^T
      st4(dst + (size_t)(4 * i + 0) * plane, r[i][0]);
      st4(dst + (size_t)(4 * i + 1) * plane, r[i][0] + r[i][1]);
      st4(dst + (size_t)(4 * i + 2) * plane, r[i][0] - r[i][1]);
      st4(dst + (size_t)(4 * i + 3) * plane, -r[i][1]);
    }
  }
}

// dW[o][i][kd][3][3] = G^T dU[.][o][i][kd] G ;  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
__global__ void wino_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dW, long long per /* O*I*KD */) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < per; e += (long long)gridDim.x * blockDim.x) {
    float u[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) u[r][c] = dU[(size_t)(4 * r + c) * per + e];
    float t[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      t[0][c] = u[0][c] + 0.5f * (u[1][c] + u[2][c]);
      t[1][c] = 0.5f * (u[1][c] - u[2][c]);
      t[2][c] = 0.5f * (u[1][c] + u[2][c]) + u[3][c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      dW[e * 9 + r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      dW[e * 9 + r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
      dW[e * 9 + r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
  }
}

inline int grid_for(long long total, int cap = 16384) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, cap));
}

}  // namespace

extern "C" {

int fo_wino_filter(const float* w, float* U, int O, int I, int KD, int Opad, int Ipad, int dgrad, void* stream) {
  const int rows = dgrad ? I : O, cols = dgrad ? O : I;
  FO_REQUIRE(KD >= 1 && Opad >= rows && Ipad >= cols, FO_E_SHAPE, "wino_filter: bad padding");
  hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for((long long)Opad * KD * Ipad)), dim3(256), 0, (hipStream_t)stream, w, U, O,
                     I, KD, Opad, Ipad, dgrad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_input(const float* x, int ldx, float* V, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ldx % 4 == 0 && fo_aligned16(x) && fo_aligned16(V), FO_E_SHAPE,
             "wino_input: even H, W; C, ld %% 4 == 0; 16-byte alignment");
  hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, V, N, H, W, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_output(const float* M, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd, float* out,
                   int ldOut, int N, int H, int W, int C, int flags, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ldOut % 4 == 0 && fo_aligned16(M) && fo_aligned16(out), FO_E_SHAPE,
             "wino_output: even H, W; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(!(flags & FO_BIAS) || bias, FO_E_SHAPE, "wino_output: FO_BIAS without bias");
  FO_REQUIRE(!(flags & FO_MASK) || (mask && ldMask % 4 == 0 && fo_aligned16(mask)), FO_E_ALIGN, "wino_output: mask");
  FO_REQUIRE(!(flags & FO_ADD) || (add && ldAdd % 4 == 0 && fo_aligned16(add)), FO_E_ALIGN, "wino_output: add");
  FO_REQUIRE(!(flags & ~(FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU)), FO_E_SHAPE, "wino_output: unsupported flag");
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, M, bias, mask, ldMask, add, ldAdd, out, ldOut, N, H, W, C / 4, flags);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_gradout(const float* g, int ldg, float* dM, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ldg % 4 == 0 && fo_aligned16(g) && fo_aligned16(dM), FO_E_SHAPE,
             "wino_gradout: even H, W; C, ld %% 4 == 0; 16-byte alignment");
  hipLaunchKernelGGL(wino_gradout_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, g, ldg, dM, N, H, W, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_wgrad_out(const float* dU, float* dW, int O, int I, int KD, void* stream) {
  const long long per = (long long)O * I * KD;
  hipLaunchKernelGGL(wino_wgrad_out_kernel, dim3(grid_for(per)), dim3(256), 0, (hipStream_t)stream, dU, dW, per);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
