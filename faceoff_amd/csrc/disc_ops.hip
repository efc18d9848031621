// HBM-bound pieces of the MoCoGAN-HD discriminators (reference TemporalAlignment/models/mocoganhd_video_disc.py,
// mocoganhd_content_disc.py, mocoganhd_losses.py:108-126, disc_trainers/train_vqvae_mocoganhd_disc.py:303-432):
// InstanceNorm + LeakyReLU forward / backward, AvgPool(3, count_include_pad=False) forward / backward, the frame pairing
// that builds the discriminator inputs (and its gradient), the relativistic average LSGAN loss.  Channels-last, 16 B per lane.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

constexpr int IN_G = 2;                 // float4 channel groups per workgroup (8 channels)
constexpr int IN_SLOTS = 256 / IN_G;    // row slots
// sum over the row slots of each channel group of a 256-thread workgroup (thread = slot * IN_G + group)
__device__ __forceinline__ f32x4 reduce_slots(f32x4 v, f32x4* sh, int tid) {
  sh[tid] = v;
  __syncthreads();
  for (int s = 128; s >= IN_G; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const f32x4 r = sh[tid % IN_G];
  __syncthreads();
  return r;
}

// One workgroup = 8 channels x all rows of one sample (blockIdx.y): the normed layers have <= 21k positions x 128..512 channels.
__global__ __launch_bounds__(256) void instnorm_lrelu_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                                 long long rows, int C, float eps, float slope, float* __restrict__ stats,
                                                                 float* __restrict__ running, float momentum, int use_running) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x, grp = tid % IN_G, slot = tid / IN_G;
  const int c = blockIdx.x * (4 * IN_G) + grp * 4;
  x += (long long)blockIdx.y * rows * ldx;            // samples are consecutive [rows][ld] blocks
  y += (long long)blockIdx.y * rows * ldy;
  stats += (long long)blockIdx.y * 2 * C;
  f32x4 mean, rstd;
  if (use_running) {
    mean = ld4(running + c);
    const f32x4 v = ld4(running + C + c);
    for (int e = 0; e < 4; ++e) rstd[e] = 1.f / sqrtf(v[e] + eps);
  } else {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (long long r = slot; r < rows; r += IN_SLOTS) s += ld4(x + r * ldx + c);
    mean = reduce_slots(s, sh, tid) * (1.f / (float)rows);
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    for (long long r = slot; r < rows; r += IN_SLOTS) { const f32x4 dlt = ld4(x + r * ldx + c) - mean; q += dlt * dlt; }
    const f32x4 var = reduce_slots(q, sh, tid) * (1.f / (float)rows);       // biased, as F.instance_norm normalises
    for (int e = 0; e < 4; ++e) rstd[e] = 1.f / sqrtf(var[e] + eps);
    if (running && slot == 0) {          // running statistics: momentum update with the UNBIASED variance
      const float unb = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
      st4(running + c, ld4(running + c) * (1.f - momentum) + mean * momentum);
      st4(running + C + c, ld4(running + C + c) * (1.f - momentum) + var * (unb * momentum));
    }
  }
  if (slot == 0) { st4(stats + c, mean); st4(stats + C + c, rstd); }
  for (long long r = slot; r < rows; r += IN_SLOTS) {
    f32x4 z = (ld4(x + r * ldx + c) - mean) * rstd;
    for (int e = 0; e < 4; ++e) z[e] = z[e] > 0.f ? z[e] : z[e] * slope;
    st4(y + r * ldy + c, z);
  }
}

// z = normalised value (recovered from y), gz = gy * lrelu'(z);  gx = rstd * (gz - mean(gz) - z * mean(gz * z))
__global__ __launch_bounds__(256) void instnorm_lrelu_bwd_kernel(const float* __restrict__ gy, int ldg, const float* __restrict__ y, int ldy,
                                                                 const float* __restrict__ stats, float* __restrict__ gx, int ldgx,
                                                                 long long rows, int C, float slope) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x, grp = tid % IN_G, slot = tid / IN_G;
  const int c = blockIdx.x * (4 * IN_G) + grp * 4;
  gy += (long long)blockIdx.y * rows * ldg;
  y += (long long)blockIdx.y * rows * ldy;
  gx += (long long)blockIdx.y * rows * ldgx;
  stats += (long long)blockIdx.y * 2 * C;
  const float inv_slope = 1.f / slope;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (long long r = slot; r < rows; r += IN_SLOTS) {
    const f32x4 yv = ld4(y + r * ldy + c);
    f32x4 g = ld4(gy + r * ldg + c);
    for (int e = 0; e < 4; ++e) {
      const bool pos = yv[e] > 0.f;
      const float z = pos ? yv[e] : yv[e] * inv_slope;
      g[e] = pos ? g[e] : g[e] * slope;
      s1[e] += g[e];
      s2[e] += g[e] * z;
    }
  }
  const float inv = 1.f / (float)rows;
  const f32x4 m1 = reduce_slots(s1, sh, tid) * inv, m2 = reduce_slots(s2, sh, tid) * inv;
  const f32x4 rstd = ld4(stats + C + c);
  for (long long r = slot; r < rows; r += IN_SLOTS) {
    const f32x4 yv = ld4(y + r * ldy + c);
    f32x4 g = ld4(gy + r * ldg + c), o;
    for (int e = 0; e < 4; ++e) {
      const bool pos = yv[e] > 0.f;
      const float z = pos ? yv[e] : yv[e] * inv_slope;
      const float gz = pos ? g[e] : g[e] * slope;
      o[e] = rstd[e] * (gz - m1[e] - z * m2[e]);
    }
    st4(gx + r * ldgx + c, o);
  }
}

// ---------------------------------------------------------------------------------------------------- chunked InstanceNorm (round 6)
// The one-workgroup-per-8-channels kernels above put 16 .. 128 workgroups on 256 CUs, read 32 bytes of every 512 .. 2048-byte row per
// instruction and pass over x three times: 37 us per launch on tensors that move in 5 (config 5: 24 launches per generator + discriminator
// pair).  Here the statistics are made in three launches that each fill the chip with whole-row accesses:
//   (A) partial:  a workgroup takes a chunk of R = 16 * 1024 / C consecutive rows (every thread 16 rows of one float4 channel group, the
//                 values stay in registers) and writes the chunk's (mean, M2 = sum of squared deviations from ITS mean) -- two-pass inside
//                 the chunk, so no E[x^2] - mean^2 cancellation -- or, backward, its (sum gz, sum gz z);
//   (B) merge:    one workgroup per 32 channels folds the chunks in a FIXED order (eight interleaved chains, then the eight in order; Chan's
//                 pairwise update for (count, mean, M2)): bit-reproducible, no atomics; writes stats = [mean | rstd] and moves the running
//                 statistics sample by sample in `order`;
//   (C) apply:    a grid-stride elementwise pass.
// Taken for C / 4 a power of two between 8 and 256 (the discriminators: 128, 256, 512 channels) in training mode; anything else runs the kernels above.
constexpr int IN_RPT = 16;              // rows per thread and chunk

__device__ __forceinline__ f32x4 reduce_cg(f32x4 v, f32x4* sh, int tid, int cg) {      // thread = slot * cg + group: sum over the slots
  sh[tid] = v;
  __syncthreads();
  for (int s = 128; s >= cg; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const f32x4 r = sh[tid & (cg - 1)];
  __syncthreads();
  return r;
}

template <bool BWD>
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy, long long rows,
                                                               int C, int cg, float slope, float* __restrict__ part, int nchunk) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x, g = tid & (cg - 1), slot = tid / cg, slots = 256 / cg, R = IN_RPT * slots;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const long long r0 = (long long)chunk * R;
  x += (long long)n * rows * ldx;
  if (BWD) y += (long long)n * rows * ldy;
  f32x4 v[IN_RPT], z[BWD ? IN_RPT : 1];
  const float inv_slope = 1.f / slope;
#pragma unroll
  for (int i = 0; i < IN_RPT; ++i) {
    const long long r = r0 + slot + i * slots;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (BWD) z[i] = v[i];
    if (r < rows) {
      v[i] = ld4(x + r * ldx + g * 4);
      if (BWD) {                              // v = gz = gy * lrelu'(z), z recovered from y
        const f32x4 yv = ld4(y + r * ldy + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool pos = yv[e] > 0.f;
          z[i][e] = pos ? yv[e] : yv[e] * inv_slope;
          v[i][e] = pos ? v[i][e] : v[i][e] * slope;
        }
      }
    }
  }
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (BWD) {
#pragma unroll
    for (int i = 0; i < IN_RPT; ++i) { a += v[i]; b += v[i] * z[i]; }
    a = reduce_cg(a, sh, tid, cg);
    b = reduce_cg(b, sh, tid, cg);
  } else {
    const float cnt = (float)min((long long)R, rows - r0);
#pragma unroll
    for (int i = 0; i < IN_RPT; ++i) a += v[i];
    a = reduce_cg(a, sh, tid, cg) * (1.f / cnt);                      // the chunk's mean
#pragma unroll
    for (int i = 0; i < IN_RPT; ++i)
      if (r0 + slot + i * slots < rows) { const f32x4 dlt = v[i] - a; b += dlt * dlt; }
    b = reduce_cg(b, sh, tid, cg);                                   // its M2
  }
  if (slot == 0) {
    float* o = part + ((long long)n * nchunk + chunk) * 2 * C + g * 4;
    st4(o, a);
    st4(o + C, b);
  }
}

// grid = C / 32; thread = (chain tid / 32, channel tid % 32)
template <bool BWD>
__global__ __launch_bounds__(256) void instnorm_merge_kernel(const float* __restrict__ part, int nchunk, int N, long long rows, int R, int C, float eps,
                                                             float* __restrict__ stats, float* __restrict__ running, const int* __restrict__ order,
                                                             float momentum) {
  __shared__ float shc[8][32], shm[8][32], shq[8][32];
  const int tid = threadIdx.x, ch = tid & 31, chain = tid >> 5, c = blockIdx.x * 32 + ch;
  for (int n = 0; n < N; ++n) {
    const float* p = part + (long long)n * nchunk * 2 * C + c;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll 4
    for (int k = chain; k < nchunk; k += 8) {
      const float a = p[(long long)k * 2 * C], b = p[(long long)k * 2 * C + C];
      if (BWD) { mean += a; m2 += b; continue; }
      const float cb = (float)min((long long)R, rows - (long long)k * R), tot = cnt + cb, dlt = a - mean;
      mean += dlt * (cb / tot);
      m2 += b + dlt * dlt * (cnt * cb / tot);
      cnt = tot;
    }
    shc[chain][ch] = cnt; shm[chain][ch] = mean; shq[chain][ch] = m2;
    __syncthreads();
    if (chain == 0) {
      for (int j = 1; j < 8; ++j) {
        const float cb = shc[j][ch], a = shm[j][ch], b = shq[j][ch];
        if (BWD) { mean += a; m2 += b; continue; }
        if (cb == 0.f) continue;
        const float tot = cnt + cb, dlt = a - mean;
        mean += dlt * (cb / tot);
        m2 += b + dlt * dlt * (cnt * cb / tot);
        cnt = tot;
      }
      float* st = stats + (long long)n * 2 * C;
      if (BWD) { st[c] = mean / (float)rows; st[C + c] = m2 / (float)rows; }                 // m1 = mean(gz), m2 = mean(gz z)
      else { st[c] = mean; st[C + c] = 1.f / sqrtf(m2 / (float)rows + eps); }                // biased variance, as F.instance_norm normalises
    }
    __syncthreads();
  }
  if (!BWD && running && chain == 0) {       // as instnorm_running_kernel: sample by sample in `order`, the variance recovered from rstd
    float rm = running[c], rv = running[C + c];
    const float unb = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    for (int i = 0; i < N; ++i) {
      const float* st = stats + (long long)order[i] * 2 * C;
      const float rstd = st[C + c];
      rm = rm * (1.f - momentum) + st[c] * momentum;
      rv = rv * (1.f - momentum) + (1.f / (rstd * rstd) - eps) * unb * momentum;
    }
    running[c] = rm;
    running[C + c] = rv;
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ yin, int ldyin,
                                                             float* __restrict__ out, int ldo, long long rows, int N, int C, int cg, float slope,
                                                             const float* __restrict__ stats, const float* __restrict__ m12) {
  const long long total = (long long)N * rows * cg;
  const float inv_slope = 1.f / slope;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(e & (cg - 1));
    const long long r = e / cg;                       // row over all samples
    const int n = (int)(r / rows);
    const float* st = stats + (long long)n * 2 * C + g * 4;
    if (BWD) {
      const f32x4 rstd = ld4(st + C), m1 = ld4(m12 + (long long)n * 2 * C + g * 4), m2 = ld4(m12 + (long long)n * 2 * C + C + g * 4);
      const f32x4 yv = ld4(yin + r * ldyin + g * 4);
      f32x4 gv = ld4(x + r * ldx + g * 4), o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool pos = yv[k] > 0.f;
        const float z = pos ? yv[k] : yv[k] * inv_slope;
        const float gz = pos ? gv[k] : gv[k] * slope;
        o[k] = rstd[k] * (gz - m1[k] - z * m2[k]);
      }
      st4(out + r * ldo + g * 4, o);
    } else {
      f32x4 z = (ld4(x + r * ldx + g * 4) - ld4(st)) * ld4(st + C);
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] = z[k] > 0.f ? z[k] : z[k] * slope;
      st4(out + r * ldo + g * 4, z);
    }
  }
}

inline bool instnorm_chunked_ok(int C) {
  const int cg = C / 4;
  return C % 4 == 0 && cg >= 8 && cg <= 256 && (cg & (cg - 1)) == 0;
}
inline int instnorm_chunk_rows(int C) { return IN_RPT * (256 / (C / 4)); }

// running statistics of a batch of samples normalised in one launch: the reference calls the module once per sample, in
// `order`, and each call moves running_mean / running_var by `momentum` (unbiased variance) -- applied here in that order
__global__ void instnorm_running_kernel(const float* __restrict__ stats, int N, const int* __restrict__ order, int C, float rows, float eps,
                                        float momentum, float* __restrict__ running) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = running[c], rv = running[C + c];
  const float unb = rows > 1.f ? rows / (rows - 1.f) : 1.f;
  for (int i = 0; i < N; ++i) {
    const float* st = stats + (long long)order[i] * 2 * C;
    const float rstd = st[C + c];
    const float var = 1.f / (rstd * rstd) - eps;
    rm = rm * (1.f - momentum) + st[c] * momentum;
    rv = rv * (1.f - momentum) + var * unb * momentum;
  }
  running[c] = rm;
  running[C + c] = rv;
}

// window of output o along one dimension: inputs [o*s - 1, o*s + 1] clipped to [0, n) (k = 3, pad = 1); k = 1: just o*s
__device__ __forceinline__ void win(int o, int s, int n, int k, int& lo, int& hi) {
  if (k == 1) { lo = hi = o * s; return; }
  lo = max(0, o * s - 1); hi = min(n - 1, o * s + 1);
}

__global__ void avgpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, int C4, int ld, int kD,
                                    int sD, int sH, int sW, int Do, int Ho, int Wo) {
  const long long total = (long long)Do * Ho * Wo * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int ow = (int)(q % Wo); q /= Wo;
    const int oh = (int)(q % Ho);
    const int od = (int)(q / Ho);
    int d0, d1, h0, h1, w0, w1;
    win(od, sD, D, kD, d0, d1); win(oh, sH, H, 3, h0, h1); win(ow, sW, W, 3, w0, w1);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int d = d0; d <= d1; ++d)
      for (int h = h0; h <= h1; ++h)
        for (int w = w0; w <= w1; ++w) s += ld4(x + (((long long)d * H + h) * W + w) * ld + c * 4);
    const float inv = 1.f / (float)((d1 - d0 + 1) * (h1 - h0 + 1) * (w1 - w0 + 1));    // count_include_pad = False
    st4(y + (((long long)od * Ho + oh) * Wo + ow) * ld + c * 4, s * inv);
  }
}

// gather form of the backward: input position i collects gy[o] / count(o) from every output whose window holds it
__global__ void avgpool3_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int D, int H, int W, int C4, int ld, int kD,
                                    int sD, int sH, int sW, int Do, int Ho, int Wo) {
  const long long total = (long long)D * H * W * C4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    long long q = e / C4;
    const int iw = (int)(q % W); q /= W;
    const int ih = (int)(q % H);
    const int id = (int)(q / H);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // candidate outputs along a dimension: o with |o*s - i| <= 1 (k = 3) or o*s == i (k = 1)
    const int od_lo = kD == 1 ? (id % sD == 0 ? id / sD : Do) : max(0, (id - 1 + sD - 1) / sD), od_hi = kD == 1 ? id / sD : min(Do - 1, (id + 1) / sD);
    const int oh_lo = max(0, (ih - 1 + sH - 1) / sH), oh_hi = min(Ho - 1, (ih + 1) / sH);
    const int ow_lo = max(0, (iw - 1 + sW - 1) / sW), ow_hi = min(Wo - 1, (iw + 1) / sW);
    for (int od = od_lo; od <= od_hi; ++od)
      for (int oh = oh_lo; oh <= oh_hi; ++oh)
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
          int d0, d1, h0, h1, w0, w1;
          win(od, sD, D, kD, d0, d1); win(oh, sH, H, 3, h0, h1); win(ow, sW, W, 3, w0, w1);
          const float inv = 1.f / (float)((d1 - d0 + 1) * (h1 - h0 + 1) * (w1 - w0 + 1));
          s += ld4(gy + (((long long)od * Ho + oh) * Wo + ow) * ld + c * 4) * inv;
        }
    float* o = gx + (((long long)id * H + ih) * W + iw) * ld + c * 4;
    st4(o, ld4(o) + s);
  }
}

__global__ void disc_pairs_kernel(const float* __restrict__ src, int nchw, int ldSrc, int H, int W, int f0, int first, int step, int n,
                                  float* __restrict__ out, int ldOut) {
  const int G = ldOut / 4;
  const long long HW = (long long)H * W, total = (long long)n * HW * G;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(e % G);
    const long long q = e / G;
    const long long pix = q % HW;
    const int j = (int)(q / HW);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g < 2) {
      const int fk = first + j * step;
      for (int k = 0; k < 4; ++k) {
        const int c = g * 4 + k;                  // channels 0..2 = frame f0, 3..5 = frame fk
        if (c >= 6) break;
        const int f = c < 3 ? f0 : fk, cc = c < 3 ? c : c - 3;
        v[k] = nchw ? src[((long long)f * 3 + cc) * HW + pix] : src[((long long)f * HW + pix) * ldSrc + cc];
      }
    }
    st4(out + ((long long)j * HW + pix) * ldOut + g * 4, v);
  }
}

// one thread per (frame, pixel): a single writer, so the accumulation order is fixed
__global__ void disc_pairs_bwd_kernel(const float* __restrict__ gout, int ldOut, int H, int W, int F, int f0, int first, int step, int n,
                                      float* __restrict__ gsrc, int ldG, float scale) {
  const long long HW = (long long)H * W, total = (long long)F * HW;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long pix = e % HW;
    const int f = (int)(e / HW);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (f == f0)
      for (int j = 0; j < n; ++j) {
        const float* p = gout + ((long long)j * HW + pix) * ldOut;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
      }
    const int dj = f - first;
    if (dj % step == 0) {
      const int j = dj / step;
      if (j >= 0 && j < n) {
        const float* p = gout + ((long long)j * HW + pix) * ldOut;
        a0 += p[3]; a1 += p[4]; a2 += p[5];
      }
    }
    float* o = gsrc + ((long long)f * HW + pix) * ldG;
    o[0] += a0 * scale; o[1] += a1 * scale; o[2] += a2 * scale;
  }
}

// Space-to-depth by 2 in (depth,) height, width: out[n][d'][h'][w'][((bd*2+bh)*2+bw)*C + c] = x[n][2d'+bd][2h'+bh][2w'+bw][c]
// (zero past the end of an odd axis and in the channel padding).  inverse: the gradient's way back (gx is WRITTEN).
// A k4 s2 p2 convolution of x is a k2 s1 p1 convolution of the result: tap k = 2a + b reads block (o + a - 1), phase b.
__global__ void s2d_kernel(const float* __restrict__ x, int ldx, float* __restrict__ xs, int ldxs, int N, int D, int H, int W, int C,
                           int pd, int inverse) {
  const int Dp = (D + pd - 1) / pd, Hp = (H + 1) / 2, Wp = (W + 1) / 2;
  const long long total = (long long)N * Dp * Hp * Wp * ldxs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(e % ldxs);
    long long q = e / ldxs;
    const int wp = (int)(q % Wp); q /= Wp;
    const int hp = (int)(q % Hp); q /= Hp;
    const int dp = (int)(q % Dp);
    const long long n = q / Dp;
    const int c = ch % C, ph = ch / C;
    const int bw = ph & 1, bh = (ph >> 1) & 1, bd = ph >> 2;
    const bool real = ph < pd * 4;
    const int d = dp * pd + bd, h = hp * 2 + bh, w = wp * 2 + bw;
    const bool in = real && d < D && h < H && w < W;
    if (!inverse) xs[e] = in ? x[(((n * D + d) * H + h) * (long long)W + w) * ldx + c] : 0.f;
    else if (in) const_cast<float*>(x)[(((n * D + d) * H + h) * (long long)W + w) * ldx + c] = xs[e];
  }
}

// w[O][C][KD][4][4] (KD = 4 or 1) <-> w2[O][phases*C][taps2]: w2[o][((bd*2+bh)*2+bw)*C + c][(ad*2+ah)*2+aw] = w[o][c][2ad+bd][2ah+bh][2aw+bw]
__global__ void s2d_filter_kernel(float* __restrict__ w, float* __restrict__ w2, int O, int C, int KD, int inverse) {
  const int taps = KD * 16, total = O * C * taps;
  const int pd = KD == 4 ? 2 : 1, taps2 = pd * 4, C2 = pd * 4 * C;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int kw = e % 4, kh = (e / 4) % 4, kd = (e / 16) % KD;
    const int c = (e / taps) % C, o = e / (taps * C);
    const int ad = kd >> 1, bd = kd & 1, ah = kh >> 1, bh = kh & 1, aw = kw >> 1, bw = kw & 1;
    const int ch = ((bd * 2 + bh) * 2 + bw) * C + c, t2 = (ad * 2 + ah) * 2 + aw;
    const int e2 = (o * C2 + ch) * taps2 + (KD == 4 ? t2 : (ah * 2 + aw));
    if (!inverse) w2[e2] = w[e];
    else w[e] = w2[e2];
  }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(1024) void ralsgan_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, int ld,
                                                       float ta, float tb, float w, float* loss_acc, const float* gscale,
                                                       float* __restrict__ ga, float* __restrict__ gb) {
  __shared__ float sh[1024];
  const int tid = threadIdx.x;
  float sa = 0.f, sb = 0.f;
  for (int i = tid; i < na; i += 1024) sa += a[(long long)i * ld];
  for (int i = tid; i < nb; i += 1024) sb += b[(long long)i * ld];
  const float ma = block_sum(sa, sh) / (float)na, mb = block_sum(sb, sh) / (float)nb;
  float l1 = 0.f, l2 = 0.f;
  for (int i = tid; i < na; i += 1024) { const float e = a[(long long)i * ld] - mb - ta; l1 += e * e; }
  for (int i = tid; i < nb; i += 1024) { const float e = b[(long long)i * ld] - ma - tb; l2 += e * e; }
  l1 = block_sum(l1, sh) / (float)na;
  l2 = block_sum(l2, sh) / (float)nb;
  if (tid == 0) loss_acc[0] += w * (l1 + l2);      // one thread of the launch's only workgroup; launches are ordered on their stream: no atomic needed
  const float gs = w * (gscale ? gscale[0] : 1.f);
  // d/da_i = 2 (a_i - mb - ta) / na  (own term)  -  2 (mb - ma - tb) / na  (through mean(a) in the other term); b likewise
  if (ga)
    for (int i = tid; i < na; i += 1024)
      ga[(long long)i * ld] = gs * 2.f / (float)na * ((a[(long long)i * ld] - mb - ta) - (mb - ma - tb));
  if (gb)
    for (int i = tid; i < nb; i += 1024)
      gb[(long long)i * ld] = gs * 2.f / (float)nb * ((b[(long long)i * ld] - ma - tb) - (ma - mb - ta));
}

inline int grid_for(long long total) { return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 16384)); }

}  // namespace

extern "C" {

int fo_instnorm_lrelu_fwd(const float* x, int ldx, float* y, int ldy, int64_t rows, int C, float eps, float slope, float* stats,
                          float* running, float momentum, int use_running, void* stream) {
  FO_REQUIRE(x && y && stats && rows > 0 && C > 0 && C % 8 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && fo_aligned16(x) && fo_aligned16(y),
             FO_E_SHAPE, "instnorm: C %% 8 == 0, ld %% 4 == 0, 16-byte alignment");
  FO_REQUIRE(!use_running || running, FO_E_SHAPE, "instnorm: eval mode needs the running statistics");
  hipLaunchKernelGGL(instnorm_lrelu_fwd_kernel, dim3(C / 8, 1), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (long long)rows, C, eps,
                     slope, stats, running, momentum, use_running);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int64_t fo_instnorm_ws_bytes(int N, int64_t rows, int C) {
  if (N <= 0 || rows <= 0 || !instnorm_chunked_ok(C)) return 0;
  const int64_t nchunk = (rows + instnorm_chunk_rows(C) - 1) / instnorm_chunk_rows(C);
  return ((int64_t)N * nchunk * 2 * C + (int64_t)N * 2 * C) * 4;
}

int fo_instnorm_lrelu_fwd_batch(const float* x, int ldx, float* y, int ldy, int N, int64_t rows, int C, float eps, float slope,
                                float* stats, float* running, const int32_t* order, float momentum, int use_running, float* ws, int64_t ws_bytes,
                                void* stream) {
  FO_REQUIRE(x && y && stats && N > 0 && rows > 0 && C > 0 && C % 8 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && fo_aligned16(x) && fo_aligned16(y),
             FO_E_SHAPE, "instnorm_batch: C %% 8 == 0, ld %% 4 == 0, 16-byte alignment");
  FO_REQUIRE(!use_running || running, FO_E_SHAPE, "instnorm_batch: eval mode needs the running statistics");
  FO_REQUIRE(use_running || !running || order, FO_E_SHAPE, "instnorm_batch: the sample order of the running-statistics updates is required");
  const int64_t need = fo_instnorm_ws_bytes(N, rows, C);
  if (!use_running && need > 0 && ws && ws_bytes >= need) {          // the chunked form (see the kernels)
    const int R = instnorm_chunk_rows(C), nchunk = (int)((rows + R - 1) / R), cg = C / 4;
    hipLaunchKernelGGL(instnorm_partial_kernel<false>, dim3(nchunk, N), dim3(256), 0, (hipStream_t)stream, x, ldx, (const float*)nullptr, 0, (long long)rows, C,
                       cg, slope, ws, nchunk);
    FO_CHECK_LAUNCH();
    hipLaunchKernelGGL(instnorm_merge_kernel<false>, dim3(C / 32), dim3(256), 0, (hipStream_t)stream, ws, nchunk, N, (long long)rows, R, C, eps, stats,
                       running, order, momentum);
    FO_CHECK_LAUNCH();
    hipLaunchKernelGGL(instnorm_apply_kernel<false>, dim3(grid_for((long long)N * rows * cg)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (const float*)nullptr, 0, y, ldy, (long long)rows, N, C, cg, slope, stats, (const float*)nullptr);
    FO_CHECK_LAUNCH();
    FO_NOTE_T("instnorm_partial_kernel", false);
    return FO_OK;
  }
  hipLaunchKernelGGL(instnorm_lrelu_fwd_kernel, dim3(C / 8, N), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (long long)rows, C, eps,
                     slope, stats, use_running ? running : nullptr, momentum, use_running);
  FO_CHECK_LAUNCH();
  if (!use_running && running) {
    hipLaunchKernelGGL(instnorm_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, N, order, C, (float)rows, eps,
                       momentum, running);
    FO_CHECK_LAUNCH();
  }
  return FO_OK;
}

int fo_instnorm_lrelu_bwd_batch(const float* gy, int ldg, const float* y, int ldy, const float* stats, float* gx, int ldgx, int N,
                                int64_t rows, int C, float slope, float* ws, int64_t ws_bytes, void* stream) {
  FO_REQUIRE(gy && y && stats && gx && N > 0 && rows > 0 && C % 8 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && ldgx % 4 == 0 && slope != 0.f, FO_E_SHAPE,
             "instnorm_bwd_batch: C %% 8 == 0, ld %% 4 == 0, slope != 0");
  const int64_t need = fo_instnorm_ws_bytes(N, rows, C);
  if (need > 0 && ws && ws_bytes >= need) {
    const int R = instnorm_chunk_rows(C), nchunk = (int)((rows + R - 1) / R), cg = C / 4;
    float* m12 = ws + (int64_t)N * nchunk * 2 * C;                  // [N][mean(gz) | mean(gz z)]
    hipLaunchKernelGGL(instnorm_partial_kernel<true>, dim3(nchunk, N), dim3(256), 0, (hipStream_t)stream, gy, ldg, y, ldy, (long long)rows, C, cg, slope,
                       ws, nchunk);
    FO_CHECK_LAUNCH();
    hipLaunchKernelGGL(instnorm_merge_kernel<true>, dim3(C / 32), dim3(256), 0, (hipStream_t)stream, ws, nchunk, N, (long long)rows, R, C, 0.f, m12,
                       (float*)nullptr, (const int*)nullptr, 0.f);
    FO_CHECK_LAUNCH();
    hipLaunchKernelGGL(instnorm_apply_kernel<true>, dim3(grid_for((long long)N * rows * cg)), dim3(256), 0, (hipStream_t)stream, gy, ldg, y, ldy, gx, ldgx,
                       (long long)rows, N, C, cg, slope, stats, m12);
    FO_CHECK_LAUNCH();
    FO_NOTE_T("instnorm_partial_kernel", true);
    return FO_OK;
  }
  hipLaunchKernelGGL(instnorm_lrelu_bwd_kernel, dim3(C / 8, N), dim3(256), 0, (hipStream_t)stream, gy, ldg, y, ldy, stats, gx, ldgx,
                     (long long)rows, C, slope);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_instnorm_lrelu_bwd(const float* gy, int ldg, const float* y, int ldy, const float* stats, float* gx, int ldgx, int64_t rows,
                          int C, float slope, void* stream) {
  FO_REQUIRE(gy && y && stats && gx && rows > 0 && C % 8 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && ldgx % 4 == 0 && slope != 0.f, FO_E_SHAPE,
             "instnorm_bwd: C %% 8 == 0, ld %% 4 == 0, slope != 0");
  hipLaunchKernelGGL(instnorm_lrelu_bwd_kernel, dim3(C / 8, 1), dim3(256), 0, (hipStream_t)stream, gy, ldg, y, ldy, stats, gx, ldgx,
                     (long long)rows, C, slope);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

static int pool_out(int n, int k, int s) { return k == 1 ? (n - 1) / s + 1 : (n + 2 - 3) / s + 1; }

int fo_avgpool3_fwd(const float* x, float* y, int D, int H, int W, int C, int ld, int kD, int sD, int sH, int sW, void* stream) {
  FO_REQUIRE(x && y && (kD == 1 || kD == 3) && C % 4 == 0 && ld % 4 == 0 && ld >= C && sD >= 1 && sH >= 1 && sW >= 1, FO_E_SHAPE, "avgpool3: bad arguments");
  const int Do = pool_out(D, kD, sD), Ho = pool_out(H, 3, sH), Wo = pool_out(W, 3, sW);
  hipLaunchKernelGGL(avgpool3_fwd_kernel, dim3(grid_for((long long)Do * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, D, H, W,
                     C / 4, ld, kD, sD, sH, sW, Do, Ho, Wo);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_avgpool3_bwd(const float* gy, float* gx, int D, int H, int W, int C, int ld, int kD, int sD, int sH, int sW, void* stream) {
  FO_REQUIRE(gy && gx && (kD == 1 || kD == 3) && C % 4 == 0 && ld % 4 == 0 && ld >= C, FO_E_SHAPE, "avgpool3_bwd: bad arguments");
  const int Do = pool_out(D, kD, sD), Ho = pool_out(H, 3, sH), Wo = pool_out(W, 3, sW);
  hipLaunchKernelGGL(avgpool3_bwd_kernel, dim3(grid_for((long long)D * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, gy, gx, D, H, W,
                     C / 4, ld, kD, sD, sH, sW, Do, Ho, Wo);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_disc_pairs(const float* src, int nchw, int ldSrc, int H, int W, int f0, int first, int step, int n, float* out, int ldOut,
                  void* stream) {
  FO_REQUIRE(src && out && n > 0 && step != 0 && ldOut >= 8 && ldOut % 4 == 0 && fo_aligned16(out), FO_E_SHAPE, "disc_pairs: bad arguments");
  hipLaunchKernelGGL(disc_pairs_kernel, dim3(grid_for((long long)n * H * W * (ldOut / 4))), dim3(256), 0, (hipStream_t)stream, src, nchw,
                     ldSrc, H, W, f0, first, step, n, out, ldOut);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_disc_pairs_bwd(const float* gout, int ldOut, int H, int W, int f0, int first, int step, int n, float* gsrc, int ldG, float scale,
                      void* stream) {
  FO_REQUIRE(gout && gsrc && n > 0 && step != 0 && ldOut >= 6 && ldG >= 3, FO_E_SHAPE, "disc_pairs_bwd: bad arguments");
  const int last = first + (n - 1) * step;
  const int F = std::max(std::max(first, last), f0) + 1;
  hipLaunchKernelGGL(disc_pairs_bwd_kernel, dim3(grid_for((long long)F * H * W)), dim3(256), 0, (hipStream_t)stream, gout, ldOut, H, W, F,
                     f0, first, step, n, gsrc, ldG, scale);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_space_to_depth2(float* x, int ldx, float* xs, int ldxs, int N, int D, int H, int W, int C, int depth_too, int inverse, void* stream) {
  const int pd = depth_too ? 2 : 1;
  FO_REQUIRE(x && xs && N > 0 && C > 0 && ldx >= C && ldxs >= pd * 4 * C, FO_E_SHAPE, "space_to_depth2: bad arguments");
  const long long total = (long long)N * ((D + pd - 1) / pd) * ((H + 1) / 2) * ((W + 1) / 2) * ldxs;
  hipLaunchKernelGGL(s2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, xs, ldxs, N, D, H, W, C, pd, inverse);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_s2d_filter(float* w, float* w2, int O, int C, int KD, int inverse, void* stream) {
  FO_REQUIRE(w && w2 && O > 0 && C > 0 && (KD == 4 || KD == 1), FO_E_SHAPE, "s2d_filter: bad arguments");
  hipLaunchKernelGGL(s2d_filter_kernel, dim3(grid_for((long long)O * C * KD * 16)), dim3(256), 0, (hipStream_t)stream, w, w2, O, C, KD, inverse);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_ralsgan(const float* a, int na, const float* b, int nb, int ld, float ta, float tb, float w, float* loss_acc,
               const float* gscale, float* ga, float* gb, void* stream) {
  FO_REQUIRE(a && b && loss_acc && na > 0 && nb > 0 && ld >= 1, FO_E_SHAPE, "ralsgan: bad arguments");
  hipLaunchKernelGGL(ralsgan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, na, b, nb, ld, ta, tb, w, loss_acc, gscale, ga, gb);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
