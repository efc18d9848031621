// bf16 forms of the LPIPS support kernels (see lpips.hip for the fp32 ones and the reference citations:
// models/lpips.py:80-161, loss.py:27-33).  Activations and their gradients are stored as bf16, channels-last,
// 8 channels = 16 B per lane; all arithmetic is fp32 and each result is rounded to bf16 once (RNE).
#include <algorithm>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int W>
__device__ __forceinline__ float group_sum(float v) { return group_sum_valu<W>(v); }   // sum over aligned groups of W lanes (xor butterfly on the vector ALU: common.h)

// ScalingLayer (lpips.py:96-103) + layout + rounding: y[p][c] = bf16((x_c - shift_c) / scale_c), c < 3; 0 for c in 3..7
__global__ void lpips_prep_bf16_kernel(const float* __restrict__ src, int src_is_nhwc, int ld, bf16x8* __restrict__ y, int HW,
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
    const f32x4 o = (v - shift) * inv_scale;
    bf16x8 r;
    r[0] = (__bf16)o.x; r[1] = (__bf16)o.y; r[2] = (__bf16)o.z;
#pragma unroll
    for (int e = 3; e < 8; ++e) r[e] = (__bf16)0.f;
    y[p] = r;
  }
}

// gdec[p][c] += weight * gscale * g[p][c] / scale_c  (c < 3); g is the bf16 [p][8] image gradient
__global__ void lpips_prep_bwd_bf16_kernel(const bf16x8* __restrict__ g, float* __restrict__ gdec, int ldd, long long npix,
                                           f32x4 inv_scale, const float* __restrict__ gscale, float weight) {
  const float k = weight * gscale[0];
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const bf16x8 gv = g[p];
    f32x4 d = *reinterpret_cast<f32x4*>(gdec + p * ldd);
    d.x += k * (float)gv[0] * inv_scale.x; d.y += k * (float)gv[1] * inv_scale.y; d.z += k * (float)gv[2] * inv_scale.z;
    *reinterpret_cast<f32x4*>(gdec + p * ldd) = d;
  }
}

__global__ void maxpool2_fwd_bf16_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y, int N, int Ho, int Wo, int C8) {
  const long long total = (long long)N * Ho * Wo * C8;
  const int W = Wo * 2;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C8);
    long long q = e / C8;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const bf16x8* base = x + ((n * (2 * Ho) + 2 * yo) * W + 2 * xo) * C8 + c;
    const bf16x8 a = base[0], b = base[C8], cc = base[(long long)W * C8], d = base[(long long)W * C8 + C8];
    bf16x8 m;
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = (__bf16)fmaxf(fmaxf((float)a[k], (float)b[k]), fmaxf((float)cc[k], (float)d[k]));
    y[e] = m;
  }
}

// y = max over the 2x2 window, idx = WHICH of its four pixels (scan order (0,0) (0,1) (1,0) (1,1); the FIRST maximum, as torch's backward) in 2 bits
// per channel: 8 channels per lane = one uint16, [N][Ho][Wo][C/8] (= bytes [..][C/4], channel c in byte c / 4, bits 2 (c % 4)).  The backward
// then reads 1/16 of the input's bytes instead of the input (round 4: the pool backwards were 1.43 ms of config 3, all of it HBM time).
__global__ void maxpool2_fwd_idx_bf16_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y, unsigned short* __restrict__ idx,
                                             unsigned char* __restrict__ ybits, int N, int Ho, int Wo, int C8) {
  const long long total = (long long)N * Ho * Wo * C8;
  const int W = Wo * 2;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C8);
    long long q = e / C8;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const bf16x8* base = x + ((n * (2 * Ho) + 2 * yo) * W + 2 * xo) * C8 + c;
    const bf16x8 a = base[0], b = base[C8], cc = base[(long long)W * C8], d = base[(long long)W * C8 + C8];
    bf16x8 m;
    unsigned code = 0, pos = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float fa = (float)a[k], fb = (float)b[k], fc = (float)cc[k], fd = (float)d[k];
      const float mx = fmaxf(fmaxf(fa, fb), fmaxf(fc, fd));
      m[k] = (__bf16)mx;
      code |= (fa == mx ? 0u : fb == mx ? 1u : fc == mx ? 2u : 3u) << (2 * k);
      pos |= (mx > 0.f ? 1u : 0u) << k;
    }
    y[e] = m;
    idx[e] = (unsigned short)code;
    if (ybits) ybits[e] = (unsigned char)pos;              // the pooled tensor's ReLU-mask bit plane: [pixel][C/8] bytes
  }
}

// gx[t] = [t is the window's recorded maximum] * gy + add[t], rounded once.  The ReLU masks are already in: `add` (a tap's head gradient) is zero where
// the input is, and gy was masked by the POOLED tensor in the data gradient that made it (max <= 0 <=> every input of the window is 0).
__global__ void maxpool2_bwd_idx_bf16_kernel(const unsigned short* __restrict__ idx, const bf16x8* __restrict__ gy, const bf16x8* __restrict__ add,
                                             bf16x8* __restrict__ gx, int N, int Ho, int Wo, int C8) {
  const long long total = (long long)N * Ho * Wo * C8;
  const int W = Wo * 2;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C8);
    long long q = e / C8;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const long long i00 = ((n * (2 * Ho) + 2 * yo) * W + 2 * xo) * C8 + c;
    const long long at[4] = {i00, i00 + C8, i00 + (long long)W * C8, i00 + (long long)W * C8 + C8};
    const unsigned code = idx[e];
    const bf16x8 g = gy[e];
    bf16x8 ad[4];
    if (add) {
#pragma unroll
      for (int t = 0; t < 4; ++t) ad[t] = add[at[t]];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = ((code >> (2 * k)) & 3u) == (unsigned)t ? (float)g[k] : 0.f;
        r[k] = (__bf16)(add ? s + (float)ad[t][k] : s);
      }
      gx[at[t]] = r;
    }
  }
}

// gx = relu'(x) * ( [x is the FIRST maximum of its window] * gy + add ), rounded once
__global__ void maxpool2_bwd_bf16_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ gy, const bf16x8* __restrict__ add,
                                         bf16x8* __restrict__ gx, int N, int Ho, int Wo, int C8) {
  const long long total = (long long)N * Ho * Wo * C8;
  const int W = Wo * 2;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C8);
    long long q = e / C8;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho);
    const long long n = q / Ho;
    const long long i00 = ((n * (2 * Ho) + 2 * yo) * W + 2 * xo) * C8 + c;
    const long long idx[4] = {i00, i00 + C8, i00 + (long long)W * C8, i00 + (long long)W * C8 + C8};
    bf16x8 v[4];
    float o[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = x[idx[t]];
    const bf16x8 g = gy[e];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float m = fmaxf(fmaxf((float)v[0][k], (float)v[1][k]), fmaxf((float)v[2][k], (float)v[3][k]));
      bool taken = false;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool is = !taken && (float)v[t][k] == m;
        taken = taken || is;
        o[t][k] = is ? (float)g[k] : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bf16x8 r;
      bf16x8 ad;
      if (add) ad = add[idx[t]];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = add ? o[t][k] + (float)ad[k] : o[t][k];
        r[k] = (__bf16)((float)v[t][k] > 0.f ? s : 0.f);
      }
      gx[idx[t]] = r;
    }
  }
}

// LPIPS head for one tap: LPP = C/8 lanes per pixel (a lane holds 8 channels = 16 B), 64/LPP pixels per wave pass.
// Every wave walks a contiguous range of pixels and keeps a running sum per frame.  No float atomics: wave w writes the sum it holds for frame n
// to slot[w * S + (n - first frame of w)] (S = the most frames a range can touch); lpips_val_finish_kernel adds a frame's slots in wave order.
// Tiny maps (H W < 64, where a pass may span frames) write one value per PIXEL instead, summed per frame in pixel order.
template <int LPP>
__global__ void lpips_head_fwd_bf16_kernel(const bf16x8* __restrict__ f0, const bf16x8* __restrict__ f1, const float* __restrict__ lin,
                                           float* __restrict__ slot, int HW, long long npix, int S) {
  constexpr int PPW = 64 / LPP;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPP, pl = lane / LPP;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long per = ((npix + nwaves - 1) / nwaves + PPW - 1) / PPW * PPW;
  const long long p_begin = wave * per, p_end = min(npix, p_begin + per);
  float* const myslot = slot + wave * S - p_begin / HW;       // myslot[n]: this wave's sum for frame n
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = lin[sub * 8 + k];
  float run = 0.f;
  long long run_n = -1;
  for (long long p0 = p_begin; p0 < p_end; p0 += PPW) {
    const long long p = p0 + pl;
    const bool ok = p < p_end;
    const long long pp = ok ? p : p_begin;
    const bf16x8 a = f0[pp * LPP + sub], b = f1[pp * LPP + sub];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sa = fmaf((float)a[k], (float)a[k], sa); sb = fmaf((float)b[k], (float)b[k], sb); }
    sa = group_sum<LPP>(sa); sb = group_sum<LPP>(sb);
    const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float r = (float)a[k] * ia - (float)b[k] * ib; acc = fmaf(w[k] * r, r, acc); }
    acc = ok ? acc : 0.f;
    if (HW < 64) {   // tiny maps: a pass may span several frames -- one value per pixel (there are few)
      acc = group_sum<LPP>(acc);
      if (ok && sub == 0) slot[p] = acc;
      continue;
    }
    // frames of the first and last pixel of this pass (wave-uniform); a pass straddles at most two frames when PPW <= HW
    const long long n_first = p0 / HW;
    const long long last = min(p0 + PPW, p_end) - 1;
    const long long n_last = last / HW;
    if (n_first != run_n) {
      if (run_n >= 0) { const float t = group_sum<64>(run); if (lane == 0) myslot[run_n] = t; }
      run = 0.f; run_n = n_first;
    }
    if (n_last == n_first) {
      run += acc;
    } else {   // split: pixels of frame n_first stay in run, the rest open the next frame
      const bool mine = (pp / HW) == n_first;
      run += mine ? acc : 0.f;
      const float t = group_sum<64>(run);
      if (lane == 0) myslot[run_n] = t;
      run = mine ? 0.f : acc;
      run_n = n_last;
    }
  }
  if (run_n >= 0) { const float t = group_sum<64>(run); if (lane == 0) myslot[run_n] = t; }
}

// gradient wrt f1 through the normalisation and f1's own ReLU (see lpips.hip), rounded to bf16 once
template <int LPP>
__global__ void lpips_head_bwd_bf16_kernel(const bf16x8* __restrict__ f0, const bf16x8* __restrict__ f1, const float* __restrict__ lin,
                                           const float* __restrict__ gscale, bf16x8* __restrict__ gf1, long long npix, float k_scale) {
  constexpr int PPW = 64 / LPP;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPP, pl = lane / LPP;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const float gk = gscale[0] * k_scale;
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = lin[sub * 8 + k];
  for (long long p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
    const long long p = p0 + pl;
    const bool ok = p < npix;
    const long long pp = ok ? p : 0;
    const bf16x8 a = f0[pp * LPP + sub], b = f1[pp * LPP + sub];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sa = fmaf((float)a[k], (float)a[k], sa); sb = fmaf((float)b[k], (float)b[k], sb); }
    sa = group_sum<LPP>(sa); sb = group_sum<LPP>(sb);
    const float na = sqrtf(sa), nb = sqrtf(sb);
    const float ia = 1.f / (na + 1e-10f), ib = 1.f / (nb + 1e-10f);
    float gn[8], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gn[k] = gk * 2.f * w[k] * ((float)b[k] * ib - (float)a[k] * ia);
      dot = fmaf(gn[k], (float)b[k], dot);
    }
    dot = group_sum<LPP>(dot);
    const float c2 = nb > 0.f ? dot * ib * ib / nb : 0.f;
    bf16x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (__bf16)((float)b[k] > 0.f ? gn[k] * ib - (float)b[k] * c2 : 0.f);
    if (ok) gf1[p * LPP + sub] = r;
  }
}

// Forward AND backward of one tap's head in ONE pass over the two feature maps (the upstream gradient of a tap's value is a known constant,
// gscale * weight / (H W N): nothing of the backward waits for the loss): the per-frame sums of lpips_head_fwd_bf16_kernel and the gradient of
// lpips_head_bwd_bf16_kernel from the same loads.  Saves one read of both maps per tap (2.7 GB at relu1_2).
// UNPOOL (round 6, `fo_lpips_tap_fwd_bwd_unpool_bf16`): the tap also feeds a 2x2 max-pool, and the launch runs in the BACKWARD, when the gradient of the
// pooled tensor is known: gf1 = head gradient + [this pixel is its window's recorded maximum] * gpool, summed in fp32 and rounded ONCE (what autograd
// does for a tensor with two consumers) -- the pool backward's pass (read head gradient + pooled gradient, write the sum) and the head gradient's own
// round trip through memory disappear: 2 x the tap's size less traffic per tap.
template <int LPP, bool UNPOOL = false>
__global__ void lpips_head_fwd_bwd_bf16_kernel(const bf16x8* __restrict__ f0, const bf16x8* __restrict__ f1, const float* __restrict__ lin,
                                               float* __restrict__ slot, const float* __restrict__ gscale, bf16x8* __restrict__ gf1, int HW,
                                               long long npix, int S, float k_scale, const bf16x8* __restrict__ gpool = nullptr,
                                               const unsigned short* __restrict__ codes = nullptr, int W = 0) {
  constexpr int PPW = 64 / LPP;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPP, pl = lane / LPP;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long per = ((npix + nwaves - 1) / nwaves + PPW - 1) / PPW * PPW;
  const long long p_begin = wave * per, p_end = min(npix, p_begin + per);
  float* const myslot = slot + wave * S - p_begin / HW;       // myslot[n]: this wave's sum for frame n
  const float gk = gscale[0] * k_scale;
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = lin[sub * 8 + k];
  float run = 0.f;
  long long run_n = -1;
  for (long long p0 = p_begin; p0 < p_end; p0 += PPW) {
    const long long p = p0 + pl;
    const bool ok = p < p_end;
    const long long pp = ok ? p : p_begin;
    const bf16x8 a = f0[pp * LPP + sub], b = f1[pp * LPP + sub];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sa = fmaf((float)a[k], (float)a[k], sa); sb = fmaf((float)b[k], (float)b[k], sb); }
    sa = group_sum<LPP>(sa); sb = group_sum<LPP>(sb);
    const float na = sqrtf(sa), nb = sqrtf(sb);
    const float ia = 1.f / (na + 1e-10f), ib = 1.f / (nb + 1e-10f);
    float acc = 0.f, gn[8], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float r = (float)a[k] * ia - (float)b[k] * ib;
      acc = fmaf(w[k] * r, r, acc);
      gn[k] = gk * 2.f * w[k] * ((float)b[k] * ib - (float)a[k] * ia);
      dot = fmaf(gn[k], (float)b[k], dot);
    }
    dot = group_sum<LPP>(dot);
    const float c2 = nb > 0.f ? dot * ib * ib / nb : 0.f;
    bf16x8 rr;
    if (UNPOOL) {
      const long long row = pp / W;                        // n * H + y (H is even: row >> 1 = n * H / 2 + y / 2)
      const int x = (int)(pp - row * W);
      const long long pq = (row >> 1) * (W >> 1) + (x >> 1);
      const unsigned pos = (unsigned)((row & 1) * 2 + (x & 1));
      const bf16x8 gq = gpool[pq * LPP + sub];
      const unsigned code = codes[pq * LPP + sub];         // 2 bits per channel: which pixel of the window is its first maximum
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float hg = (float)b[k] > 0.f ? gn[k] * ib - (float)b[k] * c2 : 0.f;
        rr[k] = (__bf16)(hg + (((code >> (2 * k)) & 3u) == pos ? (float)gq[k] : 0.f));
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) rr[k] = (__bf16)((float)b[k] > 0.f ? gn[k] * ib - (float)b[k] * c2 : 0.f);
    }
    if (ok) gf1[p * LPP + sub] = rr;
    acc = ok ? acc : 0.f;
    if (HW < 64) {   // tiny maps: a pass may span several frames -- one value per pixel (there are few)
      acc = group_sum<LPP>(acc);
      if (ok && sub == 0) slot[p] = acc;
      continue;
    }
    const long long n_first = p0 / HW;
    const long long last = min(p0 + PPW, p_end) - 1;
    const long long n_last = last / HW;
    if (n_first != run_n) {
      if (run_n >= 0) { const float t = group_sum<64>(run); if (lane == 0) myslot[run_n] = t; }
      run = 0.f; run_n = n_first;
    }
    if (n_last == n_first) {
      run += acc;
    } else {
      const bool mine = (pp / HW) == n_first;
      run += mine ? acc : 0.f;
      const float t = group_sum<64>(run);
      if (lane == 0) myslot[run_n] = t;
      run = mine ? 0.f : acc;
      run_n = n_last;
    }
  }
  if (run_n >= 0) { const float t = group_sum<64>(run); if (lane == 0) myslot[run_n] = t; }
}

inline int grid_for(long long total, int cap = 4096) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, cap));
}

// val[n] += inv_hw * (the head kernels' sums for frame n, in a fixed order): one wave per frame; lane i adds every 64th slot (pixel) in order,
// then a shuffle tree.  `per` / S as the head kernel computed them (head_plan).
__global__ __launch_bounds__(64) void lpips_val_finish_kernel(const float* __restrict__ slot, float* __restrict__ val, int HW, long long npix,
                                                              long long per, int S, float inv_hw) {
  const long long n = blockIdx.x;
  const int lane = threadIdx.x;
  float t = 0.f;
  if (HW < 64) {
    for (int i = lane; i < HW; i += 64) t += slot[n * HW + i];
  } else {
    const long long w0 = (n * HW) / per, w1 = (min((n + 1) * HW, npix) - 1) / per;
    for (long long w = w0 + lane; w <= w1; w += 64) t += slot[w * S + (n - (w * per) / HW)];
  }
  t = group_sum<64>(t);
  if (lane == 0) val[n] += t * inv_hw;
}

struct HeadPlan { long long per; int S; long long floats; };
// the pixel range per wave and the slot count of a head launch of `grid` workgroups of 4 waves, PPW pixels per pass
inline HeadPlan head_plan(long long npix, int HW, int grid, int PPW) {
  const long long nwaves = (long long)grid * 4;
  HeadPlan h;
  h.per = ((npix + nwaves - 1) / nwaves + PPW - 1) / PPW * PPW;
  h.S = (int)((h.per + HW - 1) / HW) + 1;
  h.floats = HW < 64 ? npix : nwaves * h.S;
  return h;
}

}  // namespace

extern "C" {

int fo_lpips_prep_bf16(const float* src, int src_is_nhwc, int ld, void* y, int N, int H, int W, const float* shift3,
                       const float* scale3, void* stream) {
  FO_REQUIRE(!src_is_nhwc || ld % 4 == 0, FO_E_ALIGN, "lpips_prep_bf16: ld %% 4");
  FO_REQUIRE(fo_aligned16(y), FO_E_ALIGN, "lpips_prep_bf16: y alignment");
  const f32x4 sh = {shift3[0], shift3[1], shift3[2], 0.f};
  const f32x4 is = {1.f / scale3[0], 1.f / scale3[1], 1.f / scale3[2], 0.f};
  const long long npix = (long long)N * H * W;
  hipLaunchKernelGGL(lpips_prep_bf16_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, src_is_nhwc, ld,
                     reinterpret_cast<bf16x8*>(y), H * W, npix, sh, is);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_lpips_prep_bwd_bf16(const void* g, float* gdec, int ldd, int64_t npix, const float* scale3, const float* gscale,
                           float weight, void* stream) {
  FO_REQUIRE(ldd % 4 == 0 && fo_aligned16(g), FO_E_ALIGN, "lpips_prep_bwd_bf16: alignment");
  const f32x4 is = {1.f / scale3[0], 1.f / scale3[1], 1.f / scale3[2], 0.f};
  hipLaunchKernelGGL(lpips_prep_bwd_bf16_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16x8*>(g), gdec, ldd, (long long)npix, is, gscale, weight);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_maxpool2_fwd_bf16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, FO_E_SHAPE, "maxpool2_bf16: even H, W and C %% 8 == 0");
  hipLaunchKernelGGL(maxpool2_fwd_bf16_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 8), 8192)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const bf16x8*>(x), reinterpret_cast<bf16x8*>(y), N, H / 2, W / 2, C / 8);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_maxpool2_fwd_idx_bf16(const void* x, void* y, void* idx, void* ybits, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, FO_E_SHAPE, "maxpool2_bf16: even H, W and C %% 8 == 0");
  hipLaunchKernelGGL(maxpool2_fwd_idx_bf16_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 8), 8192)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const bf16x8*>(x), reinterpret_cast<bf16x8*>(y), reinterpret_cast<unsigned short*>(idx),
                     reinterpret_cast<unsigned char*>(ybits), N, H / 2, W / 2, C / 8);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_maxpool2_bwd_idx_bf16(const void* idx, const void* gy, const void* add, void* gx, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, FO_E_SHAPE, "maxpool2_bf16: even H, W and C %% 8 == 0");
  hipLaunchKernelGGL(maxpool2_bwd_idx_bf16_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 8), 8192)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const unsigned short*>(idx), reinterpret_cast<const bf16x8*>(gy),
                     reinterpret_cast<const bf16x8*>(add), reinterpret_cast<bf16x8*>(gx), N, H / 2, W / 2, C / 8);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_maxpool2_bwd_bf16(const void* x, const void* gy, const void* add, void* gx, int N, int H, int W, int C, void* stream) {
  FO_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, FO_E_SHAPE, "maxpool2_bf16: even H, W and C %% 8 == 0");
  hipLaunchKernelGGL(maxpool2_bwd_bf16_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 8), 8192)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const bf16x8*>(x), reinterpret_cast<const bf16x8*>(gy),
                     reinterpret_cast<const bf16x8*>(add), reinterpret_cast<bf16x8*>(gx), N, H / 2, W / 2, C / 8);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int64_t fo_lpips_tap_ws_bytes_bf16(int N, int H, int W, int C) {
  if (C != 64 && C != 128 && C != 256 && C != 512) return -1;
  const long long npix = (long long)N * H * W;
  const long long a = head_plan(npix, H * W, grid_for(npix * (C / 8), 4096), 64 / (C / 8)).floats;      // fwd_bwd's grid
  const long long b = head_plan(npix, H * W, grid_for(npix * (C / 8), 2048), 64 / (C / 8)).floats;      // fwd's grid
  return std::max(a, b) * 4 + 1024;
}

int fo_lpips_tap_fwd_bf16(const void* f0, const void* f1, const float* lin, float* val, int N, int H, int W, int C, float* ws, void* stream) {
  FO_REQUIRE(ws && (C == 64 || C == 128 || C == 256 || C == 512), FO_E_SHAPE, "lpips_tap_bf16: C must be 64/128/256/512 (got %d), and a workspace", C);
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix * (C / 8), 2048);
  const HeadPlan h = head_plan(npix, H * W, grid, 64 / (C / 8));
#define FO_HEAD_FWD(LPP_)                                                                                                \
  hipLaunchKernelGGL(lpips_head_fwd_bf16_kernel<LPP_>, dim3(grid), dim3(256), 0, (hipStream_t)stream,                    \
                     reinterpret_cast<const bf16x8*>(f0), reinterpret_cast<const bf16x8*>(f1), lin, ws, H * W, npix, h.S)
  if (C == 64) FO_HEAD_FWD(8);
  else if (C == 128) FO_HEAD_FWD(16);
  else if (C == 256) FO_HEAD_FWD(32);
  else FO_HEAD_FWD(64);
#undef FO_HEAD_FWD
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(lpips_val_finish_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, ws, val, H * W, npix, h.per, h.S, 1.f / (float)(H * W));
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_lpips_tap_fwd_bwd_bf16(const void* f0, const void* f1, const float* lin, float* val, const float* gscale, void* gf1, int N, int H, int W,
                              int C, float* ws, void* stream) {
  FO_REQUIRE(ws && (C == 64 || C == 128 || C == 256 || C == 512), FO_E_SHAPE, "lpips_tap_bf16: C must be 64/128/256/512 (got %d), and a workspace", C);
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix * (C / 8), 4096);
  const HeadPlan h = head_plan(npix, H * W, grid, 64 / (C / 8));
  const float ks = 1.f / ((float)(H * W) * (float)N);
#define FO_HEAD_FB(LPP_)                                                                                                 \
  hipLaunchKernelGGL(lpips_head_fwd_bwd_bf16_kernel<LPP_>, dim3(grid), dim3(256), 0, (hipStream_t)stream,                \
                     reinterpret_cast<const bf16x8*>(f0), reinterpret_cast<const bf16x8*>(f1), lin, ws, gscale,           \
                     reinterpret_cast<bf16x8*>(gf1), H * W, npix, h.S, ks)
  if (C == 64) FO_HEAD_FB(8);
  else if (C == 128) FO_HEAD_FB(16);
  else if (C == 256) FO_HEAD_FB(32);
  else FO_HEAD_FB(64);
#undef FO_HEAD_FB
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(lpips_val_finish_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, ws, val, H * W, npix, h.per, h.S, 1.f / (float)(H * W));
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_lpips_tap_fwd_bwd_unpool_bf16(const void* f0, const void* f1, const float* lin, float* val, const float* gscale, const void* gpool,
                                     const void* codes, void* gf1, int N, int H, int W, int C, float* ws, void* stream) {
  FO_REQUIRE(ws && gpool && codes && (C == 64 || C == 128 || C == 256 || C == 512) && H % 2 == 0 && W % 2 == 0, FO_E_SHAPE,
             "lpips_tap_unpool_bf16: C must be 64/128/256/512 (got %d), even H and W, a workspace, the pooled gradient and the pool's codes", C);
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix * (C / 8), 4096);
  const HeadPlan h = head_plan(npix, H * W, grid, 64 / (C / 8));
  const float ks = 1.f / ((float)(H * W) * (float)N);
#define FO_HEAD_FBU(LPP_)                                                                                                \
  hipLaunchKernelGGL((lpips_head_fwd_bwd_bf16_kernel<LPP_, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,        \
                     reinterpret_cast<const bf16x8*>(f0), reinterpret_cast<const bf16x8*>(f1), lin, ws, gscale,           \
                     reinterpret_cast<bf16x8*>(gf1), H * W, npix, h.S, ks, reinterpret_cast<const bf16x8*>(gpool),        \
                     reinterpret_cast<const unsigned short*>(codes), W)
  if (C == 64) FO_HEAD_FBU(8);
  else if (C == 128) FO_HEAD_FBU(16);
  else if (C == 256) FO_HEAD_FBU(32);
  else FO_HEAD_FBU(64);
#undef FO_HEAD_FBU
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(lpips_val_finish_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, ws, val, H * W, npix, h.per, h.S, 1.f / (float)(H * W));
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_lpips_tap_bwd_bf16(const void* f0, const void* f1, const float* lin, const float* gscale, void* gf1, int N, int H, int W,
                          int C, void* stream) {
  const long long npix = (long long)N * H * W;
  const int grid = grid_for(npix * (C / 8), 4096);
  const float ks = 1.f / ((float)(H * W) * (float)N);
#define FO_HEAD_BWD(LPP_)                                                                                                \
  hipLaunchKernelGGL(lpips_head_bwd_bf16_kernel<LPP_>, dim3(grid), dim3(256), 0, (hipStream_t)stream,                    \
                     reinterpret_cast<const bf16x8*>(f0), reinterpret_cast<const bf16x8*>(f1), lin, gscale,               \
                     reinterpret_cast<bf16x8*>(gf1), npix, ks)
  if (C == 64) FO_HEAD_BWD(8);
  else if (C == 128) FO_HEAD_BWD(16);
  else if (C == 256) FO_HEAD_BWD(32);
  else if (C == 512) FO_HEAD_BWD(64);
  else FO_REQUIRE(false, FO_E_SHAPE, "lpips_tap_bf16: C must be 64/128/256/512 (got %d)", C);
#undef FO_HEAD_BWD
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
