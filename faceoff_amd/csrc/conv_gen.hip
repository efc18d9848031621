// General N-d convolution kernels for the MoCoGAN-HD discriminators (BASELINE config 5; reference
// TemporalAlignment/models/mocoganhd_video_disc.py:133-158 Conv3d k4 s2/s1 p2, mocoganhd_content_disc.py the Conv2d twins):
// any kernel size / stride / padding in three dimensions, channels-last [N][D][H][W][C] activations, exact-fp32 MFMA.
//
//   forward      out[n][od][oh][ow][co] = sum_{kd,kh,kw,ci} in[n][od*sD+kd-pD][oh*sH+kh-pH][ow*sW+kw-pW][ci] * w[co][kd,kh,kw][ci]
//   transposed   gin[n][id][ih][iw][ci] = sum_{taps,co : (id+pD-kd) % sD == 0 ...} g[n][(id+pD-kd)/sD][..][..][co] * w[co][taps][ci]
//                (the data gradient, as a gather: no atomics)
//   wgrad        dw[co][ci][tap]        = sum_{n,od,oh,ow} g[..][co] * in[src(od,oh,ow,tap)][ci]
//
// The layers are small (<= 133k output positions, one video per step), so the tile is 64 x 64 (4 waves x one 32x32
// accumulator: many workgroups even for the 4 624-position layers) and the structure is the simple one: per K-step one
// (tap, 32-channel chunk), row validity recomputed per tap (3 unsigned compares per row), addresses = per-row VGPR base +
// per-tap scalar offset, padding via out-of-range buffer offsets, LDS rows of 36 floats (conflict-free b128 fragment reads),
// accumulators stored straight to memory (bias / LeakyReLU / LeakyReLU-backward mask / accumulate in the epilogue).
// The transposed form orders its GEMM rows PHASE-MAJOR (destination coordinate = stride * q + phase): all rows of a tile
// share the phase, so only the taps whose parity matches it are visited (1/8 of a k4 s2 Conv3d's taps) -- the sub-pixel
// decomposition expressed as an index map instead of 8 launches.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = 36;
constexpr unsigned OOB = 0x80000000u;

struct GenArgs {
  fo_convnd_desc d;
  const float* src;
  const float* wp;      // [CdPad64][taps][Cs]
  const float* bias;
  const float* mask;
  float* dst;
  int transposed;
  int Dq, Hq, Wq;       // per-phase row grid (transposed: ceil(Dd/sD) ...; forward: Dd, Hd, Wd)
  int rowsPerPhase;     // N * Dq * Hq * Wq
  int tilesPerPhase, tilesN, taps;
  unsigned srcBytes, wpBytes;
  int margin;
  int ksplit;           // > 1: the K-steps of a tile are cut into this many slices, one workgroup each; the slices leave their 64 x 64
                        // partial tiles in ws and conv_gen_reduce_kernel adds them in slice order (+ the epilogue)
  float* ws;            // [slice][tile][64][64]
};

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__global__ __launch_bounds__(256, 2) void conv_gen_kernel(const GenArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
  __shared__ long long rowdst[BM];      // destination pixel index of each tile row, -1 = no such pixel
  float* As0 = lds;
  float* Bs0 = lds + 2 * BM * LDS_LD;
  const fo_convnd_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;

  const int slice = blockIdx.x % a.ksplit;
  const int tile_n = (blockIdx.x / a.ksplit) % a.tilesN;
  const int tile_m = (blockIdx.x / a.ksplit) / a.tilesN;
  const int phase = tile_m / a.tilesPerPhase;
  const int mt = tile_m - phase * a.tilesPerPhase;
  int phD = 0, phH = 0, phW = 0;
  if (a.transposed) { phW = phase % d.sW; phH = (phase / d.sW) % d.sH; phD = phase / (d.sW * d.sH); }

  // ---- rows: GEMM row m of this phase -> (n, qd, qh, qw); destination coordinate = q (forward) or s*q + phase (transposed)
  auto decode = [&](int m, int& n, int& qd, int& qh, int& qw) {
    qw = m % a.Wq; m /= a.Wq;
    qh = m % a.Hq; m /= a.Hq;
    qd = m % a.Dq; n = m / a.Dq;
  };
  if (tid < BM) {
    const int m = mt * BM + tid;
    long long pix = -1;
    if (m < a.rowsPerPhase) {
      int n, qd, qh, qw;
      decode(m, n, qd, qh, qw);
      const int dd = a.transposed ? qd * d.sD + phD : qd, dh = a.transposed ? qh * d.sH + phH : qh, dw = a.transposed ? qw * d.sW + phW : qw;
      if (dd < d.Dd && dh < d.Hd && dw < d.Wd) pix = (((long long)n * d.Dd + dd) * d.Hd + dh) * d.Wd + dw;
    }
    rowdst[tid] = pix;
  }
  // ---- tap walk.  transposed: only taps with (phase + pad - k) % stride == 0; their source offset e = (phase + pad - k) / s
  const int stD = a.transposed ? d.sD : 1, stH = a.transposed ? d.sH : 1, stW = a.transposed ? d.sW : 1;
  const int k0D = a.transposed ? (phD + d.pD) % d.sD : 0, k0H = a.transposed ? (phH + d.pH) % d.sH : 0, k0W = a.transposed ? (phW + d.pW) % d.sW : 0;
  const int nD = k0D < d.KD ? (d.KD - k0D + stD - 1) / stD : 0, nH = k0H < d.KH ? (d.KH - k0H + stH - 1) / stH : 0,
            nW = k0W < d.KW ? (d.KW - k0W + stW - 1) / stW : 0;
  // the smallest source offset over the visited taps goes into the row base, so that the per-tap scalar part is >= 0
  // (a buffer load's scalar offset is unsigned); the descriptor starts `margin` bytes below src for the rows it makes negative
  int eminD = 0, eminH = 0, eminW = 0;
  if (a.transposed) {
    eminD = nD > 0 ? (phD + d.pD - (k0D + (nD - 1) * stD)) / d.sD : 0;
    eminH = nH > 0 ? (phH + d.pH - (k0H + (nH - 1) * stH)) / d.sH : 0;
    eminW = nW > 0 ? (phW + d.pW - (k0W + (nW - 1) * stW)) / d.sW : 0;
  }
  // loader rows lrow, lrow + 32: base source coordinates (the tap adds a scalar) and the byte offset of (n, bd, bh, bw)
  int bd[2], bh[2], bw[2];
  unsigned rowoff[2];
  bool rv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = mt * BM + lrow + 32 * i;
    rv[i] = m < a.rowsPerPhase;
    int n, qd, qh, qw;
    decode(rv[i] ? m : 0, n, qd, qh, qw);
    if (a.transposed) { bd[i] = qd + eminD; bh[i] = qh + eminH; bw[i] = qw + eminW; }
    else { bd[i] = qd * d.sD - d.pD; bh[i] = qh * d.sH - d.pH; bw[i] = qw * d.sW - d.pW; }
    const long long p = (((long long)n * d.Ds + bd[i]) * d.Hs + bh[i]) * d.Ws + bw[i];
    rowoff[i] = (unsigned)((p * d.ldS + lcol) * 4 + a.margin);
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.src) - a.margin), 0, a.srcBytes + a.margin, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.wpBytes, 0x00020000);
  const int Ktot = a.taps * d.Cs;
  unsigned wrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) wrow[i] = (unsigned)(((size_t)(tile_n * BN + lrow + 32 * i) * Ktot + lcol) * 4);

  // ---- taps that no row of this tile can see (every row would read padding) are left out of the walk.  The rows of a tile
  // are consecutive in (n, qd, qh, qw) order: a coordinate's range over the tile is [first row's, last row's] while all the
  // coordinates above it agree, the whole axis otherwise.  (A k4 p2 s1 Conv3d over 3 frames: 37 % of its depth taps.)
  int w0D = k0D, w0H = k0H, w0W = k0W, wnD = nD, wnH = nH, wnW = nW;      // first tap / number of taps of the walk, per axis
  {
    int n0, d0, h0, q0, n1, d1, h1, q1;
    decode(mt * BM, n0, d0, h0, q0);
    decode(min(mt * BM + BM, a.rowsPerPhase) - 1, n1, d1, h1, q1);
    int dlo = 0, dhi = a.Dq - 1, hlo = 0, hhi = a.Hq - 1, wlo = 0, whi = a.Wq - 1;
    if (n0 == n1) {
      dlo = d0; dhi = d1;
      if (d0 == d1) {
        hlo = h0; hhi = h1;
        if (h0 == h1) { wlo = q0; whi = q1; }
      }
    }
    auto trim = [&](int qlo, int qhi, int K, int s, int p, int S, int ph, int k0, int st, int n, int& first, int& count) {
      int jlo, jhi;
      if (a.transposed) {            // visited tap j = 0 .. n-1 (k = k0 + j s) reads source coordinate q + e0 - j
        const int e0 = (ph + p - k0) / s;
        jlo = max(0, qlo + e0 - (S - 1));
        jhi = min(n - 1, qhi + e0);
      } else {                       // tap k reads source coordinate q s - p + k
        jlo = max(0, p - qhi * s);
        jhi = min(K - 1, S - 1 + p - qlo * s);
      }
      first = k0 + jlo * st;
      count = max(0, jhi - jlo + 1);
    };
    trim(dlo, dhi, d.KD, d.sD, d.pD, d.Ds, phD, k0D, stD, nD, w0D, wnD);
    trim(hlo, hhi, d.KH, d.sH, d.pH, d.Hs, phH, k0H, stH, nH, w0H, wnH);
    trim(wlo, whi, d.KW, d.sW, d.pW, d.Ws, phW, k0W, stW, nW, w0W, wnW);
  }
  const int endD = w0D + wnD * stD, endH = w0H + wnH * stH, endW = w0W + wnW * stW;
  const int chunks = d.Cs / BK;
  const int allsteps = wnD * wnH * wnW * chunks;
  // this workgroup's slice [s0, s1) of the K-steps, and the (tap, chunk) its walk starts at
  const int per_slice = (allsteps + a.ksplit - 1) / a.ksplit;
  const int s0 = min(allsteps, slice * per_slice), s1 = min(allsteps, s0 + per_slice);
  const int nsteps = s1 - s0;
  int ld_kd = w0D, ld_kh = w0H, ld_kw = w0W, ld_chunk = 0;
  if (s0 > 0) {
    int t = s0 / chunks;
    ld_chunk = s0 - t * chunks;
    ld_kw = w0W + (t % wnW) * stW; t /= wnW;
    ld_kh = w0H + (t % wnH) * stH; t /= wnH;
    ld_kd = w0D + t * stD;
  }
  unsigned ld_bad[2] = {0, 0};     // bit 31 set: this row reads padding at the current tap
  int ld_soffA = 0, ld_soffB = 0;
  auto tap_setup = [&]() {
    int ed, eh, ew;
    if (a.transposed) {
      ed = (phD + d.pD - ld_kd) / d.sD - eminD; eh = (phH + d.pH - ld_kh) / d.sH - eminH; ew = (phW + d.pW - ld_kw) / d.sW - eminW;
    }
    else { ed = ld_kd; eh = ld_kh; ew = ld_kw; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = rv[i] & ((unsigned)(bd[i] + ed) < (unsigned)d.Ds) & ((unsigned)(bh[i] + eh) < (unsigned)d.Hs) &
                      ((unsigned)(bw[i] + ew) < (unsigned)d.Ws);
      ld_bad[i] = ok ? 0u : OOB;
    }
    ld_soffA = ((ed * d.Hs + eh) * d.Ws + ew) * d.ldS * 4;
    ld_soffB = (((ld_kd * d.KH + ld_kh) * d.KW + ld_kw) * d.Cs) * 4;
  };
  f32x4 ra[2], rb[2];
  auto load_step = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i] = bufload(rin, ld_bad[i] | rowoff[i], ld_soffA + ld_chunk * (BK * 4));
      rb[i] = bufload(rwp, wrow[i], ld_soffB + ld_chunk * (BK * 4));
    }
    if (++ld_chunk == chunks) {
      ld_chunk = 0;
      ld_kw += stW;
      if (ld_kw >= endW) {
        ld_kw = w0W; ld_kh += stH;
        if (ld_kh >= endH) { ld_kh = w0H; ld_kd += stD; }
      }
      if (ld_kd < endD) tap_setup();
      else { ld_bad[0] = ld_bad[1] = OOB; }
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(As0 + buf * BM * LDS_LD + (lrow + 32 * i) * LDS_LD + lcol) = ra[i];
      *reinterpret_cast<f32x4*>(Bs0 + buf * BN * LDS_LD + (lrow + 32 * i) * LDS_LD + lcol) = rb[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (nsteps > 0) {
    tap_setup();
    load_step();
    store_step(0);
  }
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < nsteps; ++step) {
    const float* As = As0 + cur * BM * LDS_LD + (wm * 32 + l31) * LDS_LD + half * 4;
    const float* Bs = Bs0 + cur * BN * LDS_LD + (wn * 32 + l31) * LDS_LD + half * 4;
    if (step + 1 < nsteps) load_step();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(As + kk * 8);
      const f32x4 fb = *reinterpret_cast<const f32x4*>(Bs + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    }
    if (step + 1 < nsteps) store_step(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: register r of the accumulator = tile row (r&3) + 8(r>>2) + 4*half, column l31 (32 consecutive channels)
  if (a.ksplit > 1) {
    float* t = a.ws + ((size_t)slice * gridDim.x / a.ksplit + blockIdx.x / a.ksplit) * (BM * BN) + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BN] = acc[r];
    return;
  }
  const int co = tile_n * BN + wn * 32 + l31;
  if (co >= d.Cd) return;
  const float bv = (d.flags & FO_BIAS) ? a.bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const long long pix = rowdst[row];
    if (pix < 0) continue;
    float v = acc[r] + bv;
    if (d.flags & FO_OUT_LRELU) v = v > 0.f ? v : v * d.slope;
    if (d.flags & FO_MASK_LRELU) v = a.mask[pix * d.ldMask + co] > 0.f ? v : v * d.slope;
    float* o = a.dst + pix * d.ldD + co;
    if (d.flags & FO_ADD) v += *o;
    *o = v;
  }
}

// K-sliced launches: dst = epilogue(sum over the slices of a tile, in slice order).  One thread per (tile row, 4 columns).
__global__ void conv_gen_reduce_kernel(const GenArgs a, long long ntiles) {
  const fo_convnd_desc& d = a.d;
  const long long total = ntiles * (BM * BN / 4);
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % (BN / 4)), r = (int)((e / (BN / 4)) % BM);
    const long long tile = e / (BM * BN / 4);
    const int tile_n = (int)(tile % a.tilesN), tile_m = (int)(tile / a.tilesN);
    const int phase = tile_m / a.tilesPerPhase, mt = tile_m - phase * a.tilesPerPhase;
    int m = mt * BM + r;
    if (m >= a.rowsPerPhase) continue;
    const int qw = m % a.Wq; m /= a.Wq;
    const int qh = m % a.Hq; m /= a.Hq;
    const int qd = m % a.Dq; const int n = m / a.Dq;
    int dd = qd, dh = qh, dw = qw;
    if (a.transposed) { dw = qw * d.sW + phase % d.sW; dh = qh * d.sH + (phase / d.sW) % d.sH; dd = qd * d.sD + phase / (d.sW * d.sH); }
    if (dd >= d.Dd || dh >= d.Hd || dw >= d.Wd) continue;
    const long long pix = (((long long)n * d.Dd + dd) * d.Hd + dh) * d.Wd + dw;
    const float* p = a.ws + (size_t)tile * (BM * BN) + r * BN + c4 * 4;
    f32x4 sum = *reinterpret_cast<const f32x4*>(p);
    for (int sl = 1; sl < a.ksplit; ++sl) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + (size_t)sl * ntiles * (BM * BN));
      sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = tile_n * BN + c4 * 4 + k;
      if (co >= d.Cd) continue;
      float v = sum[k] + ((d.flags & FO_BIAS) ? a.bias[co] : 0.f);
      if (d.flags & FO_OUT_LRELU) v = v > 0.f ? v : v * d.slope;
      if (d.flags & FO_MASK_LRELU) v = a.mask[pix * d.ldMask + co] > 0.f ? v : v * d.slope;
      float* o = a.dst + pix * d.ldD + co;
      if (d.flags & FO_ADD) v += *o;
      *o = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ filter gradient
//   dw[co][ci][tap] (+)= sum_m g[m][co] * in[src(m, tap)][ci]      one workgroup: (tap, 64 co, 64 ci, one slice of the rows)
struct WgArgs {
  fo_convnd_desc d;
  const float* g;       // [M][ldD]   (M = N*Dd*Hd*Wd)
  const float* src;
  float* dw;            // [Cd][CsReal][taps]
  float* ws;            // splits > 1: [split][tap][tilesCo*64][tilesCi*64] partial sums (combined by wgrad_gen_reduce_kernel)
  int M, taps, tilesCo, tilesCi, splits, CsReal;
  unsigned srcBytes, gBytes;
  int margin;
};

// TM x TN 32 x 32 accumulators per wave (2 x 2 waves): a (64 TM) x (64 TN) block of dw[.][.][tap].  Round 4: 2 x 2 for the layers with >= 128
// channels on both sides -- a quarter of the G / X bytes staged per FLOP and of the per-row address decodes (the 64 x 64 block ran at 60 TFLOP/s).
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_gen_kernel(const WgArgs a) {
  constexpr int RK = 32;                   // rows (GEMM K) per step
  constexpr int BA = 64 * TM, BB = 64 * TN;
  constexpr int PA = BA + 4, PB = BB + 4;  // floats per staged row: [row][channels]
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];       // 2 x RK x (PA + PB) floats (67.6 KB for the 128 x 128 block: dynamic, opted in)
  float (*Gs)[RK * PA] = reinterpret_cast<float (*)[RK * PA]>(wg_lds);
  float (*Xs)[RK * PB] = reinterpret_cast<float (*)[RK * PB]>(wg_lds + 2 * RK * PA);
  const fo_convnd_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int b = blockIdx.x;
  const int split = b % a.splits; b /= a.splits;
  const int tci = b % a.tilesCi; b /= a.tilesCi;
  const int tco = b % a.tilesCo; b /= a.tilesCo;
  const int tap = b;
  const int kw = tap % d.KW, kh = (tap / d.KW) % d.KH, kd = tap / (d.KW * d.KH);

  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, a.gBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.src) - a.margin), 0, a.srcBytes + a.margin, 0x00020000);
  // loaders: 16-byte chunk id = tid + 256 i -> (row id / (16 T), channel (id % (16 T)) * 4) of the 32 x (64 T) staged block
  constexpr int CGA = 16 * TM, CGB = 16 * TN, NLA = 2 * TM, NLB = 2 * TN;
  // rows whose input frame lies in the depth padding for this tap contribute zeros and are left out: the contraction runs over
  // the VALID rows v = (n, od in [odlo, odhi], oh, ow) only, cut into `splits` equal slices
  const int HWd = d.Hd * d.Wd;
  const int odlo = max(0, (d.pD - kd + d.sD - 1) / d.sD);
  const int numhi = d.Ds - 1 + d.pD - kd;
  const int odhi = numhi < 0 ? -1 : min(d.Dd - 1, numhi / d.sD);
  const int seg = max(0, odhi - odlo + 1) * HWd;           // valid rows per sample
  const int V = d.N * seg;
  const int perSplit = ((V + a.splits - 1) / a.splits + 31) / 32 * 32;
  const int m_begin = min(V, split * perSplit), m_end = min(V, m_begin + perSplit);
  f32x4 rgv[NLA], rxv[NLB];
  // valid row v -> output position m (and whether v is inside this slice)
  auto row_m = [&](int v, bool& ok) {
    ok = v < m_end;
    const int vn = ok ? v / seg : 0;
    return ok ? (vn * d.Dd + odlo) * HWd + (v - vn * seg) : 0;
  };
  auto load = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int id = tid + 256 * i;
      bool ok;
      const int m = row_m(m0 + id / CGA, ok);
      const int c = tco * BA + (id % CGA) * 4;
      rgv[i] = bufload(rg, (ok && c < d.Cd) ? (unsigned)(((long long)m * d.ldD + c) * 4) : OOB, 0);     // (Cd may be < 64: the 1-channel head)
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int id = tid + 256 * i;
      bool ok;
      int t = row_m(m0 + id / CGB, ok);
      const int ow = t % d.Wd; t /= d.Wd;
      const int oh = t % d.Hd; t /= d.Hd;
      const int od = t % d.Dd; const int n = t / d.Dd;
      const int idp = od * d.sD - d.pD + kd, ih = oh * d.sH - d.pH + kh, iw = ow * d.sW - d.pW + kw;
      const bool in_ok = ok & ((unsigned)idp < (unsigned)d.Ds) & ((unsigned)ih < (unsigned)d.Hs) & ((unsigned)iw < (unsigned)d.Ws);
      const long long p = (((long long)n * d.Ds + idp) * d.Hs + ih) * d.Ws + iw;
      const int c = tci * BB + (id % CGB) * 4;
      rxv[i] = bufload(rx, (in_ok && c < d.Cs) ? (unsigned)((p * d.ldS + c) * 4 + a.margin) : OOB, 0);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int id = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&Gs[buf][(id / CGA) * PA + (id % CGA) * 4]) = rgv[i];
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int id = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&Xs[buf][(id / CGB) * PB + (id % CGB) * 4]) = rxv[i];
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
  if (m_begin < m_end) { load(m_begin); store(0); }
  __syncthreads();
  int cur = 0;
  for (int m0 = m_begin; m0 < m_end; m0 += RK) {
    if (m0 + RK < m_end) load(m0 + RK);
    const float* gs = &Gs[cur][half * PA + wm * 32 * TM + l31];
    const float* xs = &Xs[cur][half * PB + wn * 32 * TN + l31];
#pragma unroll
    for (int k = 0; k < RK / 2; ++k) {     // MFMA k-slot = lane half = staged row 2k + half
      float gv[TM], xv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) gv[i] = gs[2 * k * PA + 32 * i];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) xv[jn] = xs[2 * k * PB + 32 * jn];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[i], xv[jn], acc[i][jn], 0, 0, 0);
    }
    if (m0 + RK < m_end) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  const int CoP = a.tilesCo * BA, CiP = a.tilesCi * BB;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int ci = tci * BB + wn * 32 * TN + jn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = tco * BA + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (a.splits > 1) a.ws[(((size_t)split * a.taps + tap) * CoP + co) * CiP + ci] = acc[i][jn][r];
        else if (co < d.Cd && ci < a.CsReal) a.dw[((size_t)co * a.CsReal + ci) * a.taps + tap] = acc[i][jn][r];
      }
    }
}

// dw[co][ci][tap] = sum over the row slices, in slice order (bit-reproducible); one thread per (tap, co, ci), ci fastest
__global__ void wgrad_gen_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int taps, int CoP, int CiP, int Cd,
                                        int CsReal) {
  const size_t total = (size_t)taps * Cd * CsReal;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(e % CsReal);
    const int co = (int)((e / CsReal) % Cd);
    const int tap = (int)(e / ((size_t)CsReal * Cd));
    const float* p = ws + ((size_t)tap * CoP + co) * CiP + ci;
    const size_t stride = (size_t)taps * CoP * CiP;
    float sum = 0.f;
    for (int sidx = 0; sidx < splits; ++sidx) sum += p[sidx * stride];
    dw[((size_t)co * CsReal + ci) * taps + tap] = sum;
  }
}

// filters: checkpoint layout w[O][I][taps] -> forward pack [Opad64][taps][Ipad32], transposed pack [Ipad64][taps][Opad32]
__global__ void pack_gen_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int taps, int rowsPad, int colsPad,
                                int transposed) {
  const size_t total = (size_t)rowsPad * taps * colsPad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = e % colsPad;
    const int t = (e / colsPad) % taps;
    const int r = e / ((size_t)colsPad * taps);
    const int o = transposed ? c : r, i = transposed ? r : c;
    out[e] = (o < O && i < I) ? w[((size_t)o * I + i) * taps + t] : 0.f;
  }
}

int check_desc(const fo_convnd_desc* d) {
  FO_REQUIRE(d && d->N > 0 && d->Ds > 0 && d->Hs > 0 && d->Ws > 0 && d->Dd > 0 && d->Hd > 0 && d->Wd > 0, FO_E_SHAPE, "convnd: empty grid");
  FO_REQUIRE(d->KD >= 1 && d->KH >= 1 && d->KW >= 1 && d->sD >= 1 && d->sH >= 1 && d->sW >= 1 && d->pD >= 0 && d->pH >= 0 && d->pW >= 0,
             FO_E_SHAPE, "convnd: bad kernel / stride / padding");
  FO_REQUIRE(d->Cs % 32 == 0 && d->Cs > 0 && d->Cd > 0, FO_E_SHAPE, "convnd: source channels must be a multiple of 32 (pad with zeros)");
  FO_REQUIRE(d->ldS % 4 == 0 && d->ldS >= d->Cs && d->ldD >= d->Cd, FO_E_ALIGN, "convnd: ldS %% 4, ldS >= Cs, ldD >= Cd");
  return FO_OK;
}

}  // namespace

extern "C" {

int fo_pack_convnd(const float* w, float* wp, int O, int I, int taps, int transposed, void* stream) {
  FO_REQUIRE(w && wp && O > 0 && I > 0 && taps > 0, FO_E_SHAPE, "pack_convnd: bad sizes");
  const int rows = transposed ? I : O, cols = transposed ? O : I;
  const int rowsPad = (rows + 63) / 64 * 64, colsPad = (cols + 31) / 32 * 32;
  const size_t total = (size_t)rowsPad * taps * colsPad;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(pack_gen_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wp, O, I, taps, rowsPad, colsPad, transposed);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// tiles and K-slices of a launch: fills the grid members of `a`, returns the number of 64 x 64 tiles
static long long plan_convnd(const fo_convnd_desc* d, int transposed, GenArgs& a) {
  a.taps = d->KD * d->KH * d->KW;
  int phases = 1;
  if (transposed) {
    a.Dq = (d->Dd + d->sD - 1) / d->sD; a.Hq = (d->Hd + d->sH - 1) / d->sH; a.Wq = (d->Wd + d->sW - 1) / d->sW;
    phases = d->sD * d->sH * d->sW;
  } else { a.Dq = d->Dd; a.Hq = d->Hd; a.Wq = d->Wd; }
  const long long rows = (long long)d->N * a.Dq * a.Hq * a.Wq;
  if (rows >= (1ll << 30)) return -1;
  a.rowsPerPhase = (int)rows;
  a.tilesPerPhase = (a.rowsPerPhase + BM - 1) / BM;
  a.tilesN = (d->Cd + BN - 1) / BN;
  const long long tiles = (long long)phases * a.tilesPerPhase * a.tilesN;
  // few tiles but a long contraction (the 256 -> 512 layer: 292 tiles x 512 K-steps): slice K over workgroups
  a.ksplit = 1;
  const int steps = a.taps * (d->Cs / 32) / phases;
  if ((d->flags & FO_KSPLIT) && tiles < 2048 && steps >= 32) {
    long long k = (2048 + tiles - 1) / tiles;
    if (k > steps / 16) k = steps / 16;
    if (k > 1) a.ksplit = (int)k;
  }
  return tiles;
}

// bytes of workspace a launch with FO_KSPLIT in d->flags needs (0: it will not slice)
int64_t fo_convnd_ws_bytes(const fo_convnd_desc* d, int transposed) {
  if (check_desc(d)) return -1;
  GenArgs a;
  const long long tiles = plan_convnd(d, transposed, a);
  return (tiles < 0 || a.ksplit == 1) ? 0 : (int64_t)a.ksplit * tiles * BM * BN * 4;
}

// forward (transposed = 0): src = input, dst = output.  transposed = 1: src = output gradient (on the conv's OUTPUT grid,
// Cs = the conv's Cout), dst = input gradient (on the conv's INPUT grid, Cd = the conv's Cin); K*, s*, p* are the CONV's.
int fo_convnd(const fo_convnd_desc* d, int transposed, const float* src, const float* wp, const float* bias, const float* mask,
              float* dst, float* ws, int64_t ws_bytes, void* stream) {
  if (int rc = check_desc(d)) return rc;
  FO_REQUIRE(src && wp && dst && fo_aligned16(src) && fo_aligned16(wp), FO_E_ALIGN, "convnd: pointers");
  FO_REQUIRE(!(d->flags & FO_BIAS) || bias, FO_E_SHAPE, "convnd: FO_BIAS without bias");
  FO_REQUIRE(!(d->flags & FO_MASK_LRELU) || (mask && d->ldMask >= d->Cd), FO_E_SHAPE, "convnd: FO_MASK_LRELU without mask");
  FO_REQUIRE(!(d->flags & ~(FO_BIAS | FO_OUT_LRELU | FO_MASK_LRELU | FO_ADD | FO_KSPLIT)), FO_E_SHAPE, "convnd: unsupported flag");
  GenArgs a;
  a.d = *d; a.src = src; a.wp = wp; a.bias = bias; a.mask = mask; a.dst = dst; a.transposed = transposed;
  const long long tiles = plan_convnd(d, transposed, a);
  FO_REQUIRE(tiles >= 0, FO_E_SHAPE, "convnd: too many rows");
  a.ws = ws;
  FO_REQUIRE(a.ksplit == 1 || (ws && fo_aligned16(ws) && ws_bytes >= (int64_t)a.ksplit * tiles * BM * BN * 4), FO_E_SHAPE,
             "convnd: FO_KSPLIT needs a workspace of fo_convnd_ws_bytes");
  const unsigned long long srcBytes = (((unsigned long long)d->N * d->Ds * d->Hs * d->Ws - 1) * d->ldS + d->Cs) * 4ull;
  const unsigned long long wpBytes = (unsigned long long)a.tilesN * BN * a.taps * d->Cs * 4ull;
  // the descriptor starts `margin` bytes below src so that (base coordinate < 0) row offsets stay non-negative
  const long long margin = transposed ? ((((long long)(d->KD - 1) * d->Hs + (d->KH - 1)) * d->Ws + (d->KW - 1)) * d->ldS) * 4ll
                                      : ((((long long)d->pD * d->Hs + d->pH) * d->Ws + d->pW) * d->ldS) * 4ll;
  FO_REQUIRE(srcBytes + (unsigned long long)margin < (1ull << 31) && wpBytes < (1ull << 31), FO_E_SHAPE, "convnd: tensor exceeds the 2 GiB window");
  a.srcBytes = (unsigned)srcBytes; a.wpBytes = (unsigned)wpBytes; a.margin = (int)margin;
  const long long grid = tiles * a.ksplit;
  FO_REQUIRE(grid < (1ll << 31), FO_E_SHAPE, "convnd: grid");
  FO_NOTE("conv_gen_kernel");
  hipLaunchKernelGGL(conv_gen_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  FO_CHECK_LAUNCH();
  if (a.ksplit > 1) {
    hipLaunchKernelGGL(conv_gen_reduce_kernel, dim3((unsigned)std::min<long long>((tiles * (BM * BN / 4) + 255) / 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, a, tiles);
    FO_CHECK_LAUNCH();
  }
  return FO_OK;
}

// dw [Cd][CsReal][taps] = filter gradient of the forward conv described by d (src = the conv's input with Cs >= CsReal
// padded channels, g = output gradient [N*Dd*Hd*Wd][ldD]).  With fo_wgradnd_splits(d) > 1 the row slices leave their partial
// sums in ws (fo_wgradnd_ws_bytes) and a second launch adds them in slice order.
// block shape: 128 channels of a side per workgroup where that side has >= 128 (FACEOFF_WGRADND_TILE64=1: the 64 x 64 block everywhere)
static void wgradnd_tile(const fo_convnd_desc* d, int& ta, int& tb) {
  const char* e = getenv("FACEOFF_WGRADND_TILE64");
  const bool small = e && atoi(e);
  ta = (!small && d->Cd >= 128) ? 128 : 64;
  tb = (!small && d->Cs >= 128) ? 128 : 64;
}

int fo_wgradnd_splits(const fo_convnd_desc* d) {
  if (check_desc(d)) return -1;
  int ta, tb;
  wgradnd_tile(d, ta, tb);
  const long long M = (long long)d->N * d->Dd * d->Hd * d->Wd;
  const long long wgs = (long long)d->KD * d->KH * d->KW * ((d->Cd + ta - 1) / ta) * ((d->Cs + tb - 1) / tb);
  long long s = 1;
  while (wgs * s < 1024 && M / (s * 2) >= 256) s *= 2;
  return (int)s;
}

int64_t fo_wgradnd_ws_bytes(const fo_convnd_desc* d) {
  const int s = fo_wgradnd_splits(d);
  if (s <= 1) return 0;
  int ta, tb;
  wgradnd_tile(d, ta, tb);
  return (int64_t)s * d->KD * d->KH * d->KW * ((d->Cd + ta - 1) / ta * ta) * ((d->Cs + tb - 1) / tb * tb) * 4;
}

int fo_wgradnd(const fo_convnd_desc* d, const float* g, const float* src, float* dw, int CsReal, float* ws, int64_t ws_bytes, void* stream) {
  if (int rc = check_desc(d)) return rc;
  FO_REQUIRE(g && src && dw && fo_aligned16(g) && fo_aligned16(src) && d->ldD % 4 == 0, FO_E_ALIGN, "wgradnd: pointers / ldD %% 4");
  FO_REQUIRE(CsReal > 0 && CsReal <= d->Cs, FO_E_SHAPE, "wgradnd: CsReal");
  WgArgs a;
  a.d = *d; a.g = g; a.src = src; a.dw = dw; a.CsReal = CsReal;
  const long long M = (long long)d->N * d->Dd * d->Hd * d->Wd;
  FO_REQUIRE(M < (1ll << 30), FO_E_SHAPE, "wgradnd: too many rows");
  a.M = (int)M;
  a.taps = d->KD * d->KH * d->KW;
  int ta, tb;
  wgradnd_tile(d, ta, tb);
  a.tilesCo = (d->Cd + ta - 1) / ta;
  a.tilesCi = (d->Cs + tb - 1) / tb;
  a.splits = fo_wgradnd_splits(d);
  a.ws = ws;
  FO_REQUIRE(a.splits == 1 || (ws && ws_bytes >= fo_wgradnd_ws_bytes(d)), FO_E_SHAPE, "wgradnd: workspace too small (fo_wgradnd_ws_bytes)");
  const unsigned long long srcBytes = (((unsigned long long)d->N * d->Ds * d->Hs * d->Ws - 1) * d->ldS + d->Cs) * 4ull;
  const unsigned long long gBytes = (((unsigned long long)M - 1) * d->ldD + (unsigned long long)((d->Cd + 3) / 4 * 4)) * 4ull;
  FO_REQUIRE(srcBytes < (1ull << 31) && gBytes < (1ull << 31), FO_E_SHAPE, "wgradnd: tensor exceeds the 2 GiB window");
  a.srcBytes = (unsigned)srcBytes; a.gBytes = (unsigned)gBytes; a.margin = 0;
  const long long grid = (long long)a.taps * a.tilesCo * a.tilesCi * a.splits;
#define FO_WG_LAUNCH(TM_, TN_)                                                                                                   \
  do {                                                                                                                           \
    constexpr int ldsB = 2 * 32 * (64 * TM_ + 4 + 64 * TN_ + 4) * 4;                                                             \
    static fo_lds_once once;                                                                                                     \
    if (ldsB > 64 * 1024 && !fo_lds_optin(once, reinterpret_cast<const void*>(wgrad_gen_kernel<TM_, TN_>), ldsB, "wgradnd")) return FO_E_HIP; \
    FO_NOTE_T("wgrad_gen_kernel", TM_, TN_);                                                                                     \
    hipLaunchKernelGGL((wgrad_gen_kernel<TM_, TN_>), dim3((unsigned)grid), dim3(256), ldsB, (hipStream_t)stream, a);             \
  } while (0)
  if (ta == 128 && tb == 128) FO_WG_LAUNCH(2, 2);
  else if (ta == 128) FO_WG_LAUNCH(2, 1);
  else if (tb == 128) FO_WG_LAUNCH(1, 2);
  else FO_WG_LAUNCH(1, 1);
#undef FO_WG_LAUNCH
  FO_CHECK_LAUNCH();
  if (a.splits > 1) {
    const size_t total = (size_t)a.taps * d->Cd * CsReal;
    hipLaunchKernelGGL(wgrad_gen_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, ws, dw,
                       a.splits, a.taps, a.tilesCo * ta, a.tilesCi * tb, d->Cd, CsReal);
    FO_CHECK_LAUNCH();
  }
  return FO_OK;
}
}
