// Winograd F(m x m, 3x3), m = 2 or 4, for the Conv3d k3 p1 s1 layers (reference models/vqvae_conv3d_latent.py:181,185 and their data
// gradients): the two spatial dimensions of every depth tap are transformed, the depth taps and the channels stay an
// ordinary contraction:
//
//   V[xi][n][tile][ci]   = (B^T d B)[xi]                      d = 4x4 input patch of output tile `tile` of frame n
//   U[xi][co][kd][ci]    = (G g_kd G^T)[xi]                    g_kd = 3x3 slice of the filter
//   M[xi][n][tile][co]   = sum_{kd,ci} U[xi][co][kd][ci] * V[xi][n + kd - 1][tile][ci]      (16 independent GEMMs)
//   out[n][2ty+a][2tx+b] = (A^T M A)[a][b] + bias, mask, residual, ReLU
//
// (m+2)^2 multiplies per m x m output pixels instead of 9 m^2: 2.25x (m = 2) or 4x (m = 4) fewer MFMA FLOP.  The GEMMs
// (one per transform position xi) are exactly a Conv3d with a
// (3,1,1) filter over the plane stack V -- the implicit-GEMM kernel of conv_igemm.hip runs them (clip-padding taps
// skipped as usual), picking the filter bank from the frame index (fo_conv_igemm_banked).  This file holds the three
// HBM-bound transforms; all are channels-last, 16 B per lane, exact +-1 / 0.5 arithmetic in fp32.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Transform matrices of F(m x m, 3 x 3), m = 2 (points 0, 1, -1, inf) and m = 4 (0, +-1, +-2, inf); a = m + 2.
//   Y = A^T [ (G g G^T) . (B^T d B) ] A.   m = 2: fp32 error = the direct convolution's; m = 4: ~10x that (3e-6 of scale).
template <int MT> struct Wino;
template <> struct Wino<2> {
  static constexpr int A = 4;
  static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  static constexpr float G[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
  static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
};
template <> struct Wino<4> {
  static constexpr int A = 6;
  static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                     {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float G[6][3] = {{1.f / 4, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                    {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
};

// acc += c * v with the trivial coefficients folded at compile time (the loops below are fully unrolled)
template <typename T> __device__ __forceinline__ void axpy(T& acc, float c, const T& v, bool& first) {
  if (c == 0.f) return;
  if (first) { acc = c == 1.f ? v : (c == -1.f ? -v : v * c); first = false; }
  else if (c == 1.f) acc += v;
  else if (c == -1.f) acc -= v;
  else acc += v * c;
}

// U[xi][o][kd][i] (rows padded to Opad / Ipad with zeros), xi = A*r + c.
// dgrad = 1: the filter of the data gradient, w'[i][o][kd][p][q] = w[o][i][KD-1-kd][2-p][2-q] (roles of o and i swapped).
template <int MT>
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int KD, int Opad, int Ipad,
                                   int dgrad) {
  using Wn = Wino<MT>;
  constexpr int A = Wn::A;
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
    float t[A][3];   // t = G g
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float acc = 0.f; bool first = true;
#pragma unroll
        for (int p = 0; p < 3; ++p) axpy(acc, Wn::G[r][p], g[p][q], first);
        t[r][q] = acc;
      }
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int c = 0; c < A; ++c) {   // u = t G^T
        float acc = 0.f; bool first = true;
#pragma unroll
        for (int q = 0; q < 3; ++q) axpy(acc, Wn::G[c][q], t[r][q], first);
        U[(size_t)(A * r + c) * per + e] = acc;
      }
  }
}

// V[xi][n][ty][tx][c] = (B^T d B)[xi].  One thread: one tile, 4 channels.
template <int MT>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, int N, int H, int W, int C4) {
  using Wn = Wino<MT>;
  constexpr int A = Wn::A;
  const int Ht = H / MT, Wt = W / MT;
  const long long total = (long long)N * Ht * Wt * C4;
  const size_t plane = (size_t)N * Ht * Wt * C4 * 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    f32x4 d[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r) {
      const int y = MT * ty - 1 + r;
#pragma unroll
      for (int s = 0; s < A; ++s) {
        const int xx = MT * tx - 1 + s;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        d[r][s] = ok ? ld4(x + ((n * H + y) * (long long)W + xx) * ldx + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 t[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int s = 0; s < A; ++s) {   // rows: t = B^T d
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, Wn::BT[r][k], d[k][s], first);
        t[r][s] = acc;
      }
    float* dst = V + (size_t)e * 4;
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int s = 0; s < A; ++s) {   // columns: v = t B
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, Wn::BT[s][k], t[r][k], first);
        st4(dst + (size_t)(A * r + s) * plane, acc);
      }
  }
}

// out = epilogue(A^T M A).  One thread: one tile (m x m output pixels), 4 channels.
// epilogue order as in the conv kernels: (+ bias) -> ReLU mask -> + residual -> ReLU.
template <int MT>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ mask,
                                   int ldMask, const float* __restrict__ add, int ldAdd, float* __restrict__ out, int ldOut, int N,
                                   int H, int W, int C4, int flags) {
  using Wn = Wino<MT>;
  constexpr int A = Wn::A;
  const int Ht = H / MT, Wt = W / MT;
  const long long total = (long long)N * Ht * Wt * C4;
  const size_t plane = (size_t)N * Ht * Wt * C4 * 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    const float* src = M + (size_t)e * 4;
    f32x4 t[MT][A];
#pragma unroll
    for (int s = 0; s < A; ++s) {       // rows: t = A^T m, one column of m at a time (keeps 6 planes live, not 36)
      f32x4 mcol[A];
#pragma unroll
      for (int k = 0; k < A; ++k) mcol[k] = ld4(src + (size_t)(A * k + s) * plane);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(acc, Wn::AT[a][k], mcol[k], first);
        t[a][s] = acc;
      }
    }
    const f32x4 bv = (flags & FO_BIAS) ? ld4(bias + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(v, Wn::AT[b][k], t[a][k], first);
        v += bv;
        const long long pix = (n * H + MT * ty + a) * (long long)W + MT * tx + b;
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
//   dM[xi][n][tile][co] = (A dY A^T)[xi]
//   dU[xi][co][ci][kd]  = sum_{n,tile} dM[xi][n][tile][co] * V[xi][n + kd - 1][tile][ci]     (a*a wgrad GEMMs, banked)
//   dW[co][ci][kd]      = G^T dU G                                    (a x a -> 3 x 3)
// BIAS: the kernel reads every pixel of g exactly once, so the bias gradient (column sums of g) rides along instead of being a pass of its
// own over g: a thread's channel group c is the same for every element it visits (C4 divides the block size), its running sum meets the
// block's other threads of that group in LDS (fixed order) and the block's partial goes to bws[block][C] for colsum_stage2.
template <int MT, bool BIAS>
__global__ __launch_bounds__(256) void wino_gradout_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dM, int N, int H, int W, int C4,
                                                           float* __restrict__ bws) {
  using Wn = Wino<MT>;
  constexpr int A = Wn::A;
  const int Ht = H / MT, Wt = W / MT;
  const long long total = (long long)N * Ht * Wt * C4;
  const size_t plane = (size_t)N * Ht * Wt * C4 * 4;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    f32x4 y[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) y[a][b] = ld4(g + ((n * H + MT * ty + a) * (long long)W + MT * tx + b) * ldg + c * 4);
    if (BIAS) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) csum += y[a][b];
    }
    f32x4 r[A][MT];                      // r = A y   (A = (A^T)^T: r[i][b] = sum_a AT[a][i] y[a][b])
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int a = 0; a < MT; ++a) axpy(acc, Wn::AT[a][i], y[a][b], first);
        r[i][b] = acc;
      }
    float* dst = dM + (size_t)e * 4;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {      // m = r A^T: m[i][j] = sum_b r[i][b] AT[b][j]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int b = 0; b < MT; ++b) axpy(acc, Wn::AT[b][j], r[i][b], first);
        st4(dst + (size_t)(A * i + j) * plane, acc);
      }
  }
  if (BIAS) {
    __shared__ f32x4 red[256];
    red[threadIdx.x] = csum;
    __syncthreads();
    if ((int)threadIdx.x < C4) {
      f32x4 t = red[threadIdx.x];
      for (int k = threadIdx.x + C4; k < 256; k += C4) t += red[k];
      st4(bws + ((size_t)blockIdx.x * C4 + threadIdx.x) * 4, t);
    }
  }
}

// dW[o][i][kd][3][3] = G^T dU[.][o][i][kd] G
template <int MT>
__global__ void wino_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dW, long long per /* O*I*KD */) {
  using Wn = Wino<MT>;
  constexpr int A = Wn::A;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < per; e += (long long)gridDim.x * blockDim.x) {
    float t[3][A];                        // t = G^T u : t[p][c] = sum_r G[r][p] u[r][c]
#pragma unroll
    for (int c = 0; c < A; ++c) {
      float ucol[A];
#pragma unroll
      for (int r = 0; r < A; ++r) ucol[r] = dU[(size_t)(A * r + c) * per + e];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float acc = 0.f; bool first = true;
#pragma unroll
        for (int r = 0; r < A; ++r) axpy(acc, Wn::G[r][p], ucol[r], first);
        t[p][c] = acc;
      }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) {       // dW = t G : dW[p][q] = sum_c t[p][c] G[c][q]
        float acc = 0.f; bool first = true;
#pragma unroll
        for (int c = 0; c < A; ++c) axpy(acc, Wn::G[c][q], t[p][c], first);
        dW[e * 9 + p * 3 + q] = acc;
      }
  }
}

inline int grid_for(long long total, int cap = 16384) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, cap));
}

}  // namespace

extern "C" {

#define FO_WINO_M(m_) FO_REQUIRE((m_) == 2 || (m_) == 4, FO_E_SHAPE, "winograd: output tile size m must be 2 or 4 (got %d)", (m_))

int fo_wino_filter(const float* w, float* U, int O, int I, int KD, int Opad, int Ipad, int dgrad, int m, void* stream) {
  const int rows = dgrad ? I : O, cols = dgrad ? O : I;
  FO_WINO_M(m);
  FO_REQUIRE(KD >= 1 && Opad >= rows && Ipad >= cols, FO_E_SHAPE, "wino_filter: bad padding");
  const dim3 grid(grid_for((long long)Opad * KD * Ipad));
  if (m == 2) hipLaunchKernelGGL(wino_filter_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, w, U, O, I, KD, Opad, Ipad, dgrad);
  else hipLaunchKernelGGL(wino_filter_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, w, U, O, I, KD, Opad, Ipad, dgrad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_input(const float* x, int ldx, float* V, int N, int H, int W, int C, int m, void* stream) {
  FO_WINO_M(m);
  FO_REQUIRE(H % m == 0 && W % m == 0 && C % 4 == 0 && ldx % 4 == 0 && fo_aligned16(x) && fo_aligned16(V), FO_E_SHAPE,
             "wino_input: H, W multiples of m; C, ld %% 4 == 0; 16-byte alignment");
  const dim3 grid(grid_for((long long)N * (H / m) * (W / m) * (C / 4)));
  if (m == 2) hipLaunchKernelGGL(wino_input_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, V, N, H, W, C / 4);
  else hipLaunchKernelGGL(wino_input_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, V, N, H, W, C / 4);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_output(const float* M, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd, float* out,
                   int ldOut, int N, int H, int W, int C, int flags, int m, void* stream) {
  FO_WINO_M(m);
  FO_REQUIRE(H % m == 0 && W % m == 0 && C % 4 == 0 && ldOut % 4 == 0 && fo_aligned16(M) && fo_aligned16(out), FO_E_SHAPE,
             "wino_output: H, W multiples of m; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(!(flags & FO_BIAS) || bias, FO_E_SHAPE, "wino_output: FO_BIAS without bias");
  FO_REQUIRE(!(flags & FO_MASK) || (mask && ldMask % 4 == 0 && fo_aligned16(mask)), FO_E_ALIGN, "wino_output: mask");
  FO_REQUIRE(!(flags & FO_ADD) || (add && ldAdd % 4 == 0 && fo_aligned16(add)), FO_E_ALIGN, "wino_output: add");
  FO_REQUIRE(!(flags & ~(FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU)), FO_E_SHAPE, "wino_output: unsupported flag");
  const dim3 grid(grid_for((long long)N * (H / m) * (W / m) * (C / 4)));
  if (m == 2)
    hipLaunchKernelGGL(wino_output_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, M, bias, mask, ldMask, add, ldAdd, out, ldOut,
                       N, H, W, C / 4, flags);
  else
    hipLaunchKernelGGL(wino_output_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, M, bias, mask, ldMask, add, ldAdd, out, ldOut,
                       N, H, W, C / 4, flags);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_wino_gradout(const float* g, int ldg, float* dM, int N, int H, int W, int C, int m, void* stream) {
  FO_WINO_M(m);
  FO_REQUIRE(H % m == 0 && W % m == 0 && C % 4 == 0 && ldg % 4 == 0 && fo_aligned16(g) && fo_aligned16(dM), FO_E_SHAPE,
             "wino_gradout: H, W multiples of m; C, ld %% 4 == 0; 16-byte alignment");
  const dim3 grid(grid_for((long long)N * (H / m) * (W / m) * (C / 4)));
  if (m == 2) hipLaunchKernelGGL((wino_gradout_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, g, ldg, dM, N, H, W, C / 4, (float*)nullptr);
  else hipLaunchKernelGGL((wino_gradout_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, g, ldg, dM, N, H, W, C / 4, (float*)nullptr);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// the same transform, with dbias[c] = column sums of g (c < C) riding along
static int gradout_bias_grid(int N, int H, int W, int C, int m) { return grid_for((long long)N * (H / m) * (W / m) * (C / 4), 2048); }
int64_t fo_wino_gradout_bias_ws_bytes(int N, int H, int W, int C, int m) {
  if ((m != 2 && m != 4) || N <= 0 || C <= 0 || C % 4 != 0 || H % m != 0 || W % m != 0) return -1;
  return (int64_t)gradout_bias_grid(N, H, W, C, m) * C * 4;
}
int fo_wino_gradout_bias(const float* g, int ldg, float* dM, int N, int H, int W, int C, int m, float* dbias, float* ws, int64_t ws_bytes, void* stream) {
  FO_WINO_M(m);
  FO_REQUIRE(H % m == 0 && W % m == 0 && C % 4 == 0 && ldg % 4 == 0 && fo_aligned16(g) && fo_aligned16(dM), FO_E_SHAPE,
             "wino_gradout: H, W multiples of m; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(256 % (C / 4) == 0, FO_E_SHAPE, "wino_gradout_bias: C / 4 must divide 256 (got C = %d)", C);
  FO_REQUIRE(dbias && ws && ws_bytes >= fo_wino_gradout_bias_ws_bytes(N, H, W, C, m), FO_E_SHAPE, "wino_gradout_bias: dbias / workspace");
  const int nblk = gradout_bias_grid(N, H, W, C, m);
  if (m == 2) hipLaunchKernelGGL((wino_gradout_kernel<2, true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, ldg, dM, N, H, W, C / 4, ws);
  else hipLaunchKernelGGL((wino_gradout_kernel<4, true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, ldg, dM, N, H, W, C / 4, ws);
  FO_CHECK_LAUNCH();
  return fo_colsum_finish(ws, dbias, nblk, C, C, stream);
}

int fo_wino_wgrad_out(const float* dU, float* dW, int O, int I, int KD, int m, void* stream) {
  FO_WINO_M(m);
  const long long per = (long long)O * I * KD;
  if (m == 2) hipLaunchKernelGGL(wino_wgrad_out_kernel<2>, dim3(grid_for(per)), dim3(256), 0, (hipStream_t)stream, dU, dW, per);
  else hipLaunchKernelGGL(wino_wgrad_out_kernel<4>, dim3(grid_for(per)), dim3(256), 0, (hipStream_t)stream, dU, dW, per);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
