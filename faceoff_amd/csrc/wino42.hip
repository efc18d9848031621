// Winograd F(4 x 4, 2 x 2) for the stride-2 stems: Conv2d k4 s2 p1 (reference models/vqvae_conv3d_latent.py:108,110,117) and
// ConvTranspose2d k4 s2 p1 (:157,160 and upsample_t :215), forward, data gradient and filter gradient.
//
// A k4 s2 p1 convolution is a k2 s1 convolution over 2 x 2 pixel CELLS of the zero-padded input (space-to-depth):
//   cell(i, j)[(a, b, ci)] = x[2i + a - 1][2j + b - 1][ci]          i, j = 0 .. H/2,  a, b = 0, 1
//   y[oy][ox][co] = sum_{di, dj = 0, 1} sum_{a, b, ci} cell(oy + di, ox + dj)[(a, b, ci)] * w[co][ci][2 di + a][2 dj + b]
// and its adjoint (the data gradient; equally the forward of the transposed convolution) is a FULL k2 correlation over the
// gradient grid producing cells (depth-to-space + crop):
//   dcell(i, j)[(a, b, ci)] = sum_{ei, ej = 0, 1} dy[i + ei - 1][j + ej - 1][co] * w[co][ci][2 (1 - ei) + a][2 (1 - ej) + b]
// Both are k2 convolutions on a grid, so F(4 x 4, 2 x 2) applies: 5 x 5 = 25 multiplies per 4 x 4 outputs instead of 64
// (2.56x fewer MFMA FLOP), with the channel contraction K = 4 Cin (conv) or Cout (transposed):
//   V[xi][row][k]  = (B^T d B)[xi]            d = 5 x 5 window of cells / gradient pixels of output tile `row`
//   U[xi][n][k]    = (G g G^T)[xi]            g = the 2 x 2 cell filter
//   M[xi][row][n]  = sum_k V[xi][row][k] U[xi][n][k]                 25 GEMMs: fo_wino_gemm with KD = 1
//   out            = A^T M A  (+ bias, ReLU mask, residual, ReLU; for the transposed form scattered cell -> 2 x 2 pixels)
// Filter gradient: dU[xi] = sum_rows (A dY A^T)[xi] (x) V[xi]  (fo_conv_wgrad_banked), dW = G^T dU G.
// Points 0, 1, -1, 2, inf: |B^T| <= 3, |A^T| <= 8 -- better conditioned than the F(4 x 4, 3 x 3) of winograd.hip.
// All kernels: channels-last, 16 B per lane, one thread per (tile, 4 channels), HBM-bound.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

constexpr int A = 5, MT = 4;
struct W42 {
  static constexpr float BT[5][5] = {{2, -1, -2, 1, 0}, {0, -2, -1, 1, 0}, {0, 2, -3, 1, 0}, {0, -1, 0, 1, 0}, {0, 2, -1, -2, 1}};
  static constexpr float G[5][2] = {{.5f, 0}, {-.5f, -.5f}, {-1.f / 6, 1.f / 6}, {1.f / 6, 1.f / 3}, {0, 1}};
  static constexpr float AT[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 0}, {0, 1, -1, 8, 1}};
};

template <typename T> __device__ __forceinline__ void axpy(T& acc, float c, const T& v, bool& first) {
  if (c == 0.f) return;
  if (first) { acc = c == 1.f ? v : (c == -1.f ? -v : v * c); first = false; }
  else if (c == 1.f) acc += v;
  else if (c == -1.f) acc -= v;
  else acc += v * c;
}

// d (5 x 5 of f32x4) -> B^T d B, stored to dst + xi * plane
__device__ __forceinline__ void input_transform_store(f32x4 (&d)[A][A], float* dst, size_t plane) {
  f32x4 t[A][A];
#pragma unroll
  for (int r = 0; r < A; ++r)
#pragma unroll
    for (int s = 0; s < A; ++s) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
      for (int k = 0; k < A; ++k) axpy(acc, W42::BT[r][k], d[k][s], first);
      t[r][s] = acc;
    }
#pragma unroll
  for (int r = 0; r < A; ++r)
#pragma unroll
    for (int s = 0; s < A; ++s) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
      for (int k = 0; k < A; ++k) axpy(acc, W42::BT[s][k], t[r][k], first);
      st4(dst + (size_t)(A * r + s) * plane, acc);
    }
}

// conv form: V[xi][(n, ty, tx)][(a, b, c)] from x[N][H][W] (C channels, row pitch ldx).  Thread: (tile, ab, 4 channels).
__global__ __launch_bounds__(256) void w42_input_cells_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, int N, int H, int W,
                                                              int C4, size_t plane) {
  const int Ht = H / 8, Wt = W / 8;
  const long long total = (long long)N * Ht * Wt * 4 * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int ab = (int)(q & 3); q >>= 2;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    const int a = ab >> 1, b = ab & 1;
    f32x4 d[A][A];
#pragma unroll
    for (int r = 0; r < A; ++r) {
      const int y = 2 * (MT * ty + r) + a - 1;
#pragma unroll
      for (int s = 0; s < A; ++s) {
        const int xx = 2 * (MT * tx + s) + b - 1;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        d[r][s] = ok ? ld4(x + ((n * H + y) * (long long)W + xx) * ldx + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    input_transform_store(d, V + (size_t)e * 4, plane);
  }
}

// transposed form: V[xi][(n, ty, tx)][c] from g[N][h][w] (C channels); output cells (h + 1) x (w + 1), window rows 4 ty - 1 + r.
__global__ __launch_bounds__(256) void w42_input_full_kernel(const float* __restrict__ g, int ldg, float* __restrict__ V, int N, int h, int w,
                                                             int C4, size_t plane) {
  const int Ht = (h + 1 + MT - 1) / MT, Wt = (w + 1 + MT - 1) / MT;
  const long long total = (long long)N * Ht * Wt * C4;
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
        const bool ok = (unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)w;
        d[r][s] = ok ? ld4(g + ((n * h + y) * (long long)w + xx) * ldg + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    input_transform_store(d, V + (size_t)e * 4, plane);
  }
}

// U from w[O][I][4][4] (the same memory layout serves Conv2d [O][I] and ConvTranspose2d [I_T][O_T] with O := I_T, I := O_T).
//   transposed = 0:  U[xi][o][(a, b, ci)],   g[di][dj] = w[o][ci][2 di + a][2 dj + b]                    (K = 4 I, rows O)
//   transposed = 1:  U[xi][(a, b, ci)][o],   g[ei][ej] = w[o][ci][2 (1 - ei) + a][2 (1 - ej) + b]        (K = O, rows 4 I)
__global__ void w42_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int transposed) {
  const size_t per = (size_t)O * 4 * I;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
    int o, ab, ci;
    if (!transposed) { ci = e % I; ab = (e / I) & 3; o = e / ((size_t)4 * I); }
    else { o = e % O; ci = (e / O) % I; ab = e / ((size_t)O * I); }
    const int a = ab >> 1, b = ab & 1;
    float g[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kh = transposed ? 2 * (1 - p) + a : 2 * p + a, kw = transposed ? 2 * (1 - q) + b : 2 * q + b;
        g[p][q] = w[(((size_t)o * I + ci) * 4 + kh) * 4 + kw];
      }
    float t[A][2];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) t[r][q] = W42::G[r][0] * g[0][q] + W42::G[r][1] * g[1][q];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int c = 0; c < A; ++c) U[(size_t)(A * r + c) * per + e] = W42::G[c][0] * t[r][0] + W42::G[c][1] * t[r][1];
  }
}

__device__ __forceinline__ f32x4 epilogue(f32x4 v, const f32x4& bv, const float* mask, int ldMask, const float* add, int ldAdd, long long pix,
                                          int c, int flags) {
  v += bv;
  if (flags & FO_MASK) {
    const f32x4 mk = ld4(mask + pix * ldMask + c * 4);
    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
  }
  if (flags & FO_ADD) v += ld4(add + pix * ldAdd + c * 4);
  if (flags & FO_OUT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  return v;
}

// t = A^T m (4 x 5), one column of m at a time
__device__ __forceinline__ void output_rows(const float* src, size_t plane, f32x4 (&t)[MT][A]) {
#pragma unroll
  for (int s = 0; s < A; ++s) {
    f32x4 mcol[A];
#pragma unroll
    for (int k = 0; k < A; ++k) mcol[k] = ld4(src + (size_t)(A * k + s) * plane);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
      for (int k = 0; k < A; ++k) axpy(acc, W42::AT[a][k], mcol[k], first);
      t[a][s] = acc;
    }
  }
}

// conv form: out[n][4 ty + oi][4 tx + oj][c] = (A^T M A)[oi][oj] -> epilogue.  h, w = output size (multiples of 4).
__global__ __launch_bounds__(256) void w42_output_kernel(const float* __restrict__ M, size_t plane, const float* __restrict__ bias,
                                                         const float* __restrict__ mask, int ldMask, const float* __restrict__ add, int ldAdd,
                                                         float* __restrict__ out, int ldOut, int N, int h, int w, int C4, int flags) {
  const int Ht = h / MT, Wt = w / MT;
  const long long total = (long long)N * Ht * Wt * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    f32x4 t[MT][A];
    output_rows(M + (size_t)e * 4, plane, t);
    const f32x4 bv = (flags & FO_BIAS) ? ld4(bias + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(v, W42::AT[b][k], t[a][k], first);
        const long long pix = (n * h + MT * ty + a) * (long long)w + MT * tx + b;
        st4(out + pix * ldOut + c * 4, epilogue(v, bv, mask, ldMask, add, ldAdd, pix, c, flags));
      }
  }
}

// transposed form: M[xi][(n, ty, tx)][(a, b, c)] -> cell (4 ty + oi, 4 tx + oj) -> pixel (2 i + a - 1, 2 j + b - 1) of out[N][2h][2w].
__global__ __launch_bounds__(256) void w42_output_cells_kernel(const float* __restrict__ M, size_t plane, const float* __restrict__ bias,
                                                               const float* __restrict__ mask, int ldMask, const float* __restrict__ add,
                                                               int ldAdd, float* __restrict__ out, int ldOut, int N, int h, int w, int C4,
                                                               int flags) {
  const int Ht = (h + 1 + MT - 1) / MT, Wt = (w + 1 + MT - 1) / MT;
  const int H = 2 * h, W = 2 * w;
  const long long total = (long long)N * Ht * Wt * 4 * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int ab = (int)(q & 3); q >>= 2;
    const int tx = (int)(q % Wt); q /= Wt;
    const int ty = (int)(q % Ht);
    const long long n = q / Ht;
    const int a = ab >> 1, b = ab & 1;
    f32x4 t[MT][A];
    output_rows(M + (size_t)e * 4, plane, t);
    const f32x4 bv = (flags & FO_BIAS) ? ld4(bias + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int oi = 0; oi < MT; ++oi) {
      const int y = 2 * (MT * ty + oi) + a - 1;
#pragma unroll
      for (int oj = 0; oj < MT; ++oj) {
        const int xx = 2 * (MT * tx + oj) + b - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy(v, W42::AT[oj][k], t[oi][k], first);
        if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) {
          const long long pix = (n * H + y) * (long long)W + xx;
          st4(out + pix * ldOut + c * 4, epilogue(v, bv, mask, ldMask, add, ldAdd, pix, c, flags));
        }
      }
    }
  }
}

// dM[xi][(n, ty, tx)][c] = (A dY A^T)[xi], dY = 4 x 4 tile of g[N][h][w]
// BIAS: the bias gradient (column sums of g) rides along as in wino_gradout_kernel (winograd.hip): per-block partials to bws[block][4 C4]
template <bool BIAS>
__global__ __launch_bounds__(256) void w42_gradout_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dM, int N, int h, int w,
                                                          int C4, size_t plane, float* __restrict__ bws) {
  const int Ht = h / MT, Wt = w / MT;
  const long long total = (long long)N * Ht * Wt * C4;
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
      for (int b = 0; b < MT; ++b) y[a][b] = ld4(g + ((n * h + MT * ty + a) * (long long)w + MT * tx + b) * ldg + c * 4);
    if (BIAS) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) csum += y[a][b];
    }
    f32x4 r[A][MT];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int a = 0; a < MT; ++a) axpy(acc, W42::AT[a][i], y[a][b], first);
        r[i][b] = acc;
      }
    float* dst = dM + (size_t)e * 4;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}; bool first = true;
#pragma unroll
        for (int b = 0; b < MT; ++b) axpy(acc, W42::AT[b][j], r[i][b], first);
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

// dW[o][ci][2 di + a][2 dj + b] = (G^T dU[.][o][(a, b, ci)] G)[di][dj]
__global__ void w42_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dW, int O, int I) {
  const long long per = (long long)O * 4 * I;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < per; e += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(e % I), ab = (int)((e / I) & 3), o = (int)(e / ((long long)4 * I));
    const int a = ab >> 1, b = ab & 1;
    float t[2][A];
#pragma unroll
    for (int c = 0; c < A; ++c) {
      float ucol[A];
#pragma unroll
      for (int r = 0; r < A; ++r) ucol[r] = dU[(size_t)(A * r + c) * per + e];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float acc = 0.f; bool first = true;
#pragma unroll
        for (int r = 0; r < A; ++r) axpy(acc, W42::G[r][p], ucol[r], first);
        t[p][c] = acc;
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float acc = 0.f; bool first = true;
#pragma unroll
        for (int c = 0; c < A; ++c) axpy(acc, W42::G[c][q], t[p][c], first);
        dW[(((size_t)o * I + ci) * 4 + 2 * p + a) * 4 + 2 * q + b] = acc;
      }
  }
}

inline int grid_for(long long total, int cap = 16384) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, cap));
}

}  // namespace

extern "C" {

int fo_w42_filter(const float* w, float* U, int O, int I, int transposed, void* stream) {
  FO_REQUIRE(w && U && O > 0 && I > 0, FO_E_SHAPE, "w42_filter: bad sizes");
  hipLaunchKernelGGL(w42_filter_kernel, dim3(grid_for((long long)O * 4 * I)), dim3(256), 0, (hipStream_t)stream, w, U, O, I, transposed);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_w42_input_cells(const float* x, int ldx, float* V, int N, int H, int W, int C, long long planeRows, void* stream) {
  FO_REQUIRE(H % 8 == 0 && W % 8 == 0 && C % 4 == 0 && ldx % 4 == 0 && fo_aligned16(x) && fo_aligned16(V), FO_E_SHAPE,
             "w42_input_cells: H, W multiples of 8; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(planeRows >= (long long)N * (H / 8) * (W / 8), FO_E_SHAPE, "w42_input_cells: plane shorter than the tile count");
  hipLaunchKernelGGL(w42_input_cells_kernel, dim3(grid_for((long long)N * (H / 8) * (W / 8) * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, V,
                     N, H, W, C / 4, (size_t)planeRows * 4 * C);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_w42_input_full(const float* g, int ldg, float* V, int N, int h, int w, int C, long long planeRows, void* stream) {
  FO_REQUIRE(h > 0 && w > 0 && C % 4 == 0 && ldg % 4 == 0 && fo_aligned16(g) && fo_aligned16(V), FO_E_SHAPE,
             "w42_input_full: C, ld %% 4 == 0; 16-byte alignment");
  const long long tiles = (long long)N * ((h + 4) / 4) * ((w + 4) / 4);
  FO_REQUIRE(planeRows >= tiles, FO_E_SHAPE, "w42_input_full: plane shorter than the tile count");
  hipLaunchKernelGGL(w42_input_full_kernel, dim3(grid_for(tiles * (C / 4))), dim3(256), 0, (hipStream_t)stream, g, ldg, V, N, h, w, C / 4,
                     (size_t)planeRows * C);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

static int check_epilogue(const char* who, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd, int flags) {
  FO_REQUIRE(!(flags & FO_BIAS) || bias, FO_E_SHAPE, "%s: FO_BIAS without bias", who);
  FO_REQUIRE(!(flags & FO_MASK) || (mask && ldMask % 4 == 0 && fo_aligned16(mask)), FO_E_ALIGN, "%s: mask", who);
  FO_REQUIRE(!(flags & FO_ADD) || (add && ldAdd % 4 == 0 && fo_aligned16(add)), FO_E_ALIGN, "%s: add", who);
  FO_REQUIRE(!(flags & ~(FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU)), FO_E_SHAPE, "%s: unsupported flag", who);
  return FO_OK;
}

int fo_w42_output(const float* M, long long planeRows, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd,
                  float* out, int ldOut, int N, int h, int w, int C, int flags, void* stream) {
  FO_REQUIRE(h % 4 == 0 && w % 4 == 0 && C % 4 == 0 && ldOut % 4 == 0 && fo_aligned16(M) && fo_aligned16(out), FO_E_SHAPE,
             "w42_output: h, w multiples of 4; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(planeRows >= (long long)N * (h / 4) * (w / 4), FO_E_SHAPE, "w42_output: plane shorter than the tile count");
  if (int rc = check_epilogue("w42_output", bias, mask, ldMask, add, ldAdd, flags)) return rc;
  hipLaunchKernelGGL(w42_output_kernel, dim3(grid_for((long long)N * (h / 4) * (w / 4) * (C / 4))), dim3(256), 0, (hipStream_t)stream, M,
                     (size_t)planeRows * C, bias, mask, ldMask, add, ldAdd, out, ldOut, N, h, w, C / 4, flags);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_w42_output_cells(const float* M, long long planeRows, const float* bias, const float* mask, int ldMask, const float* add, int ldAdd,
                        float* out, int ldOut, int N, int h, int w, int C, int flags, void* stream) {
  FO_REQUIRE(h > 0 && w > 0 && C % 4 == 0 && ldOut % 4 == 0 && fo_aligned16(M) && fo_aligned16(out), FO_E_SHAPE,
             "w42_output_cells: C, ld %% 4 == 0; 16-byte alignment");
  const long long tiles = (long long)N * ((h + 4) / 4) * ((w + 4) / 4);
  FO_REQUIRE(planeRows >= tiles, FO_E_SHAPE, "w42_output_cells: plane shorter than the tile count");
  if (int rc = check_epilogue("w42_output_cells", bias, mask, ldMask, add, ldAdd, flags)) return rc;
  hipLaunchKernelGGL(w42_output_cells_kernel, dim3(grid_for(tiles * C)), dim3(256), 0, (hipStream_t)stream, M, (size_t)planeRows * 4 * C, bias,
                     mask, ldMask, add, ldAdd, out, ldOut, N, h, w, C / 4, flags);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_w42_gradout(const float* g, int ldg, float* dM, int N, int h, int w, int C, long long planeRows, void* stream) {
  FO_REQUIRE(h % 4 == 0 && w % 4 == 0 && C % 4 == 0 && ldg % 4 == 0 && fo_aligned16(g) && fo_aligned16(dM), FO_E_SHAPE,
             "w42_gradout: h, w multiples of 4; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(planeRows >= (long long)N * (h / 4) * (w / 4), FO_E_SHAPE, "w42_gradout: plane shorter than the tile count");
  hipLaunchKernelGGL(w42_gradout_kernel<false>, dim3(grid_for((long long)N * (h / 4) * (w / 4) * (C / 4))), dim3(256), 0, (hipStream_t)stream, g, ldg,
                     dM, N, h, w, C / 4, (size_t)planeRows * C, (float*)nullptr);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// the same transform with dbias[c] = column sums of g (c < C) riding along
static int w42_gradout_bias_grid(int N, int h, int w, int C) {
  return (int)std::max<long long>(1, std::min<long long>(((long long)N * (h / 4) * (w / 4) * (C / 4) + 255) / 256, 2048));
}
int64_t fo_w42_gradout_bias_ws_bytes(int N, int h, int w, int C) {
  if (N <= 0 || C <= 0 || C % 4 != 0 || h % 4 != 0 || w % 4 != 0) return -1;
  return (int64_t)w42_gradout_bias_grid(N, h, w, C) * C * 4;
}
int fo_w42_gradout_bias(const float* g, int ldg, float* dM, int N, int h, int w, int C, long long planeRows, float* dbias, float* ws, int64_t ws_bytes,
                        void* stream) {
  FO_REQUIRE(h % 4 == 0 && w % 4 == 0 && C % 4 == 0 && ldg % 4 == 0 && fo_aligned16(g) && fo_aligned16(dM), FO_E_SHAPE,
             "w42_gradout: h, w multiples of 4; C, ld %% 4 == 0; 16-byte alignment");
  FO_REQUIRE(planeRows >= (long long)N * (h / 4) * (w / 4), FO_E_SHAPE, "w42_gradout: plane shorter than the tile count");
  FO_REQUIRE(256 % (C / 4) == 0, FO_E_SHAPE, "w42_gradout_bias: C / 4 must divide 256 (got C = %d)", C);
  FO_REQUIRE(dbias && ws && ws_bytes >= fo_w42_gradout_bias_ws_bytes(N, h, w, C), FO_E_SHAPE, "w42_gradout_bias: dbias / workspace");
  const int nblk = w42_gradout_bias_grid(N, h, w, C);
  hipLaunchKernelGGL(w42_gradout_kernel<true>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, ldg, dM, N, h, w, C / 4, (size_t)planeRows * C, ws);
  FO_CHECK_LAUNCH();
  return fo_colsum_finish(ws, dbias, nblk, C, C, stream);
}

int fo_w42_wgrad_out(const float* dU, float* dW, int O, int I, void* stream) {
  FO_REQUIRE(dU && dW && O > 0 && I > 0, FO_E_SHAPE, "w42_wgrad_out: bad sizes");
  hipLaunchKernelGGL(w42_wgrad_out_kernel, dim3(grid_for((long long)O * 4 * I)), dim3(256), 0, (hipStream_t)stream, dU, dW, O, I);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
