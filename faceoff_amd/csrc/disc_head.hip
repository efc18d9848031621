// The discriminators' patch head (reference TemporalAlignment/models/mocoganhd_video_disc.py:150-158, mocoganhd_content_disc.py: the last
// Conv3d / Conv2d(512, 1, kernel 4, stride 1, padding 2)) as what it is -- one dot product of 64 x 512 (16 x 512) numbers per output position --
// instead of a 64-column implicit-GEMM tile of which one column is real (conv_gen_kernel: 0.51 / 0.23 / 0.63 ms forward / data gradient /
// filter gradient at scale 0 = 1.6 TFLOP/s nominal, each of them a 19 MB read):
//   forward        y[o]      = b + sum_{tap, c} x[o + tap - p][c] * w[tap][c]
//   data gradient  gx[i][c]  = sum_{tap}  g[i - tap + p] * w[tap][c]
//   filter grad.   dw[c][tap] = sum_{i}   g[i - tap + p] * x[i][c]
// Forward and data gradient: the filter (128 KB for the 3-D head) is staged in LDS once per workgroup, a wave owns one position at a time, a
// lane 8 of the 512 channels (two 16-byte loads per tap), one wave reduction per output.  Filter gradient: a workgroup walks INPUT positions
// (x is read once; the <= 64 output gradients that meet a position are scalar loads), a thread keeps its two channels' 64 taps in 128
// accumulators; one slab per workgroup, summed in slab order.  All three: fixed orders, bit-reproducible (the tile kernel's K-slices met in
// float atomics).
#include <algorithm>
#include "common.h"

namespace {

struct HeadArgs {
  const float* x;       // [N][Ds][Hs][Ws][ldX], C channels (fwd, wgrad) | unused (dgrad)
  const float* g;       // [N][Dd][Hd][Wd][ldG], channel 0 (dgrad, wgrad) | unused (fwd)
  const float* w;       // [taps][C]  (row 0 of the forward pack)
  const float* bias;    // [1] or null (fwd)
  float* out;           // fwd: y [..][ldG] channel 0; dgrad: gx [..][ldX]; wgrad: slabs [grid][taps][C]
  int N, Ds, Hs, Ws, Dd, Hd, Wd, C, ldX, ldG;
  int KD, KH, KW, pD, pH, pW;
};

constexpr int MAXCPL = 8;                 // channels per lane (C = 64 * CPL, CPL in {4, 8})

// LDS image of the filter: [tap][CPL / 4][64 lanes][4] -- a lane's two 16-byte reads come from two contiguous 1 KB planes (conflict-free)
__device__ __forceinline__ void stage_filter(const HeadArgs& a, float* wl, int taps, int cpl) {
  const int total = taps * a.C;
  for (int e = threadIdx.x * 4; e < total; e += blockDim.x * 4) {
    const int tap = e / a.C, c = e - tap * a.C;            // 4 consecutive channels c .. c + 3 of lane c / cpl
    const int lane = c / cpl, j = c - lane * cpl;
    *reinterpret_cast<f32x4*>(wl + ((tap * (cpl >> 2) + (j >> 2)) * 64 + lane) * 4) = *reinterpret_cast<const f32x4*>(a.w + e);
  }
}

__device__ __forceinline__ float wave_sum64(float v) { return group_sum_valu<64>(v); }     // (xor butterfly on the vector ALU: common.h)

template <int CPL>
__global__ __launch_bounds__(256) void disc_head_fwd_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int taps = a.KD * a.KH * a.KW;
  stage_filter(a, wl, taps, CPL);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long M = (long long)a.N * a.Dd * a.Hd * a.Wd;
  const float b = a.bias ? a.bias[0] : 0.f;
  for (long long o = (long long)blockIdx.x * 4 + wave; o < M; o += (long long)gridDim.x * 4) {
    int ow = (int)(o % a.Wd); long long q = o / a.Wd;
    const int oh = (int)(q % a.Hd); q /= a.Hd;
    const int od = (int)(q % a.Dd), n = (int)(q / a.Dd);
    f32x4 acc[CPL / 4];
#pragma unroll
    for (int v = 0; v < CPL / 4; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kd = 0; kd < a.KD; ++kd) {
      const int id = od + kd - a.pD;
      if ((unsigned)id >= (unsigned)a.Ds) continue;
      for (int kh = 0; kh < a.KH; ++kh) {
        const int ih = oh + kh - a.pH;
        if ((unsigned)ih >= (unsigned)a.Hs) continue;
        const float* xrow = a.x + ((((size_t)n * a.Ds + id) * a.Hs + ih) * a.Ws) * a.ldX + lane * CPL;
        const float* wrow = wl + (size_t)((kd * a.KH + kh) * a.KW) * (CPL >> 2) * 256 + lane * 4;
        for (int kw = 0; kw < a.KW; ++kw) {
          const int iw = ow + kw - a.pW;
          if ((unsigned)iw >= (unsigned)a.Ws) continue;
#pragma unroll
          for (int v = 0; v < CPL / 4; ++v) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + (size_t)iw * a.ldX + v * 4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + (kw * (CPL >> 2) + v) * 256);
            acc[v] += xv * wv;
          }
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < CPL / 4; ++v) s += (acc[v].x + acc[v].y) + (acc[v].z + acc[v].w);
    s = wave_sum64(s);
    if (lane == 0) a.out[(size_t)o * a.ldG] = s + b;
  }
}

template <int CPL>
__global__ __launch_bounds__(256) void disc_head_dgrad_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int taps = a.KD * a.KH * a.KW;
  stage_filter(a, wl, taps, CPL);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long M = (long long)a.N * a.Ds * a.Hs * a.Ws;
  for (long long i = (long long)blockIdx.x * 4 + wave; i < M; i += (long long)gridDim.x * 4) {
    int iw = (int)(i % a.Ws); long long q = i / a.Ws;
    const int ih = (int)(q % a.Hs); q /= a.Hs;
    const int id = (int)(q % a.Ds), n = (int)(q / a.Ds);
    f32x4 acc[CPL / 4];
#pragma unroll
    for (int v = 0; v < CPL / 4; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kd = 0; kd < a.KD; ++kd) {
      const int od = id - kd + a.pD;
      if ((unsigned)od >= (unsigned)a.Dd) continue;
      for (int kh = 0; kh < a.KH; ++kh) {
        const int oh = ih - kh + a.pH;
        if ((unsigned)oh >= (unsigned)a.Hd) continue;
        const float* grow = a.g + ((((size_t)n * a.Dd + od) * a.Hd + oh) * a.Wd) * a.ldG;
        const float* wrow = wl + (size_t)((kd * a.KH + kh) * a.KW) * (CPL >> 2) * 256 + lane * 4;
        for (int kw = 0; kw < a.KW; ++kw) {
          const int ow = iw - kw + a.pW;
          if ((unsigned)ow >= (unsigned)a.Wd) continue;
          const float gv = grow[(size_t)ow * a.ldG];         // (the same address in every lane)
#pragma unroll
          for (int v = 0; v < CPL / 4; ++v) acc[v] += *reinterpret_cast<const f32x4*>(wrow + (kw * (CPL >> 2) + v) * 256) * gv;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < CPL / 4; ++v) *reinterpret_cast<f32x4*>(a.out + (size_t)i * a.ldX + lane * CPL + v * 4) = acc[v];
  }
}

// thread t owns channels CPT t .. CPT t + CPT - 1 (CPT = C / 256) and all taps; TAPS is a compile-time bound on KD * KH * KW
template <int TAPS, int CPT>
__global__ __launch_bounds__(256) void disc_head_wgrad_kernel(const HeadArgs a) {
  // The output-gradient value of every tap of a position is fetched ONCE by the workgroup (thread t < taps loads tap t's, zero where the tap
  // falls outside the output grid) into LDS, two positions ahead of its use; round 3's form loaded it per thread and tap -- 64 dependent
  // same-address loads per position: 0.34 ms for 2 000 positions, latency-bound.
  __shared__ float gsh[2][TAPS];
  const int taps = a.KD * a.KH * a.KW;
  float acc[TAPS][CPT];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[t][c] = 0.f;
  const long long M = (long long)a.N * a.Ds * a.Hs * a.Ws;
  const int t = threadIdx.x;
  const int tkw = t % a.KW, tkh = (t / a.KW) % a.KH, tkd = t / (a.KW * a.KH);
  auto fetch_g = [&](long long i) -> float {           // thread t: the gradient value tap t of input position i multiplies
    if (t >= taps || i >= M) return 0.f;
    int iw = (int)(i % a.Ws); long long q = i / a.Ws;
    const int ih = (int)(q % a.Hs); q /= a.Hs;
    const int id = (int)(q % a.Ds), n = (int)(q / a.Ds);
    const int od = id - tkd + a.pD, oh = ih - tkh + a.pH, ow = iw - tkw + a.pW;
    if ((unsigned)od < (unsigned)a.Dd && (unsigned)oh < (unsigned)a.Hd && (unsigned)ow < (unsigned)a.Wd)
      return a.g[((((size_t)n * a.Dd + od) * a.Hd + oh) * a.Wd + ow) * a.ldG];
    return 0.f;
  };
  auto fetch_x = [&](long long i, float (&xv)[CPT]) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) xv[c] = i < M ? a.x[(size_t)i * a.ldX + threadIdx.x * CPT + c] : 0.f;
  };
  long long i = blockIdx.x;
  float gnext = fetch_g(i), xnext[CPT];
  fetch_x(i, xnext);
  int buf = 0;
  for (; i < M; i += gridDim.x) {
    if (t < TAPS) gsh[buf][t] = gnext;
    float xv[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) xv[c] = xnext[c];
    gnext = fetch_g(i + gridDim.x);                     // the next position's loads fly during this position's FMAs
    fetch_x(i + gridDim.x, xnext);
    __syncthreads();                                    // (two buffers: the previous position's readers are past their loop before gsh[buf] is rewritten two positions on)
#pragma unroll
    for (int tt = 0; tt < TAPS; ++tt) {
      if (tt < taps) {
        const float gv = gsh[buf][tt];
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[tt][c] = fmaf(gv, xv[c], acc[tt][c]);
      }
    }
    buf ^= 1;
  }
  float* slab = a.out + (size_t)blockIdx.x * taps * a.C;
#pragma unroll
  for (int tt = 0; tt < TAPS; ++tt)
    if (tt < taps)
#pragma unroll
      for (int c = 0; c < CPT; ++c) slab[(size_t)tt * a.C + threadIdx.x * CPT + c] = acc[tt][c];
}

// dw[c][tap] = sum over the slabs [b][tap][c], in slab order (four running sums)
__global__ void disc_head_wgrad_reduce_kernel(const float* __restrict__ ws, int nslabs, int taps, int Cc, int Creal, float* __restrict__ dw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;      // e = tap * C + c
  if (e >= taps * Cc) return;
  const size_t st = (size_t)taps * Cc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 3 < nslabs; b += 4) { s0 += ws[b * st + e]; s1 += ws[(b + 1) * st + e]; s2 += ws[(b + 2) * st + e]; s3 += ws[(b + 3) * st + e]; }
  for (; b < nslabs; ++b) s0 += ws[b * st + e];
  const int tap = e / Cc, c = e - tap * Cc;
  if (c < Creal) dw[(size_t)c * taps + tap] = (s0 + s1) + (s2 + s3);
}

int fill(const fo_convnd_desc* d, HeadArgs& a, const char* what) {
  FO_REQUIRE(d->sD == 1 && d->sH == 1 && d->sW == 1 && (d->Cs == 256 || d->Cs == 512) && d->Cd == 1 && d->ldS % 4 == 0 && d->KD * d->KH * d->KW <= 64 &&
                 d->KD * d->KH * d->KW * d->Cs * 4 <= 160 * 1024 - 1024,
             FO_E_SHAPE, "%s: a stride-1 convolution of 256 or 512 channels to ONE channel with <= 64 taps is required", what);
  a.N = d->N; a.Ds = d->Ds; a.Hs = d->Hs; a.Ws = d->Ws; a.Dd = d->Dd; a.Hd = d->Hd; a.Wd = d->Wd; a.C = d->Cs; a.ldX = d->ldS; a.ldG = d->ldD;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.pD = d->pD; a.pH = d->pH; a.pW = d->pW;
  return FO_OK;
}

template <typename K>
int set_lds(K kern, int bytes) {          // set at every launch (a dozen launches per iteration): correct on any device of the process
  if (bytes > 64 * 1024)
    FO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess, FO_E_HIP,
               "disc_head: cannot reserve %d bytes of LDS", bytes);
  return FO_OK;
}

}  // namespace

extern "C" {

// y[position][0] = bias + conv(x)[position]; d as for fo_convnd's forward (Cd == 1; the other ldD - 1 floats of a pixel are not touched)
int fo_disc_head_fwd(const fo_convnd_desc* d, const float* x, const float* wp, const float* bias, float* y, void* stream) {
  HeadArgs a;
  if (int rc = fill(d, a, "disc_head_fwd")) return rc;
  a.x = x; a.g = nullptr; a.w = wp; a.bias = bias; a.out = y;
  const int lds = a.KD * a.KH * a.KW * a.C * 4;
  const long long M = (long long)a.N * a.Dd * a.Hd * a.Wd;
  const int grid = (int)std::min<long long>((M + 3) / 4, fo_cu_count());
  if (a.C == 512) { if (int rc = set_lds(disc_head_fwd_kernel<8>, lds)) return rc; FO_NOTE_T("disc_head_fwd_kernel", 8); hipLaunchKernelGGL(disc_head_fwd_kernel<8>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a); }
  else { if (int rc = set_lds(disc_head_fwd_kernel<4>, lds)) return rc; FO_NOTE_T("disc_head_fwd_kernel", 4); hipLaunchKernelGGL(disc_head_fwd_kernel<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a); }
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// gx[input position][c] = data gradient of that convolution for the output gradient g[position][0]; d as for the FORWARD (Cs = channels of gx)
int fo_disc_head_dgrad(const fo_convnd_desc* d, const float* g, const float* wp, float* gx, void* stream) {
  HeadArgs a;
  if (int rc = fill(d, a, "disc_head_dgrad")) return rc;
  a.x = nullptr; a.g = g; a.w = wp; a.bias = nullptr; a.out = gx;
  const int lds = a.KD * a.KH * a.KW * a.C * 4;
  const long long M = (long long)a.N * a.Ds * a.Hs * a.Ws;
  const int grid = (int)std::min<long long>((M + 3) / 4, fo_cu_count());
  if (a.C == 512) { if (int rc = set_lds(disc_head_dgrad_kernel<8>, lds)) return rc; FO_NOTE_T("disc_head_dgrad_kernel", 8); hipLaunchKernelGGL(disc_head_dgrad_kernel<8>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a); }
  else { if (int rc = set_lds(disc_head_dgrad_kernel<4>, lds)) return rc; FO_NOTE_T("disc_head_dgrad_kernel", 4); hipLaunchKernelGGL(disc_head_dgrad_kernel<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a); }
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int64_t fo_disc_head_wgrad_ws_bytes(const fo_convnd_desc* d) {
  return (int64_t)fo_cu_count() * d->KD * d->KH * d->KW * d->Cs * 4;
}

// dw[0][c][tap] (c < CsReal) = filter gradient; ws: fo_disc_head_wgrad_ws_bytes(d)
int fo_disc_head_wgrad(const fo_convnd_desc* d, const float* g, const float* x, float* dw, int CsReal, float* ws, int64_t ws_bytes, void* stream) {
  HeadArgs a;
  if (int rc = fill(d, a, "disc_head_wgrad")) return rc;
  FO_REQUIRE(ws && ws_bytes >= fo_disc_head_wgrad_ws_bytes(d) && CsReal > 0 && CsReal <= d->Cs, FO_E_SHAPE, "disc_head_wgrad: workspace / CsReal");
  a.x = x; a.g = g; a.w = nullptr; a.bias = nullptr; a.out = ws;
  const int taps = a.KD * a.KH * a.KW;
  const long long M = (long long)a.N * a.Ds * a.Hs * a.Ws;
  const int grid = (int)std::min<long long>(M, fo_cu_count());
  if (a.C == 512) {
    if (taps > 16) { FO_NOTE_T("disc_head_wgrad_kernel", 64, 2); hipLaunchKernelGGL((disc_head_wgrad_kernel<64, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
    else { FO_NOTE_T("disc_head_wgrad_kernel", 16, 2); hipLaunchKernelGGL((disc_head_wgrad_kernel<16, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
  } else {
    if (taps > 16) { FO_NOTE_T("disc_head_wgrad_kernel", 64, 1); hipLaunchKernelGGL((disc_head_wgrad_kernel<64, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
    else { FO_NOTE_T("disc_head_wgrad_kernel", 16, 1); hipLaunchKernelGGL((disc_head_wgrad_kernel<16, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
  }
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(disc_head_wgrad_reduce_kernel, dim3((taps * a.C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, grid, taps, a.C, CsReal, dw);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

}  // extern "C"
