// Implicit-GEMM convolution for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   out[opix(m)][co] = epilogue( sum_{tap,ci} in[ipix(m,tap)][ci] * wp[co][tap][ci] )
//
// One kernel family serves every conv of the FaceOff VQ-VAE step (reference
// models/vqvae_conv3d_latent.py): Conv2d k4s2 / k3 / k1 (:92,94,109-120,140,208,213), the four
// sub-pixel phases of ConvTranspose2d k4s2 (:150,152,160,215), Conv3d k3 (:181,185) and all
// their data gradients (a dgrad is the same contraction with a re-packed filter).
//
// Design (MI355X-first, no im2col buffer):
//  * channels-last activations: a GEMM-A row is one pixel's contiguous Cin slice, so every global
//    load is a 16-B-per-lane read of a 128-B line;
//  * both operands are K-contiguous (filters packed [co][tap][ci]), staged through LDS with rows
//    padded to 36 floats: the ds_read_b128 fragment reads and ds_write_b128 stores are
//    bank-conflict free (36*r mod 64 is distinct over every 16-lane b128 group);
//  * a lane's float4 holds 4 consecutive k for 4 consecutive MFMAs (the MFMA's two k slots are
//    the two lane halves), so one ds_read_b128 feeds 4 MFMAs per tile;
//  * 128 x BN block tile, 4 waves (one per SIMD) x up to 2x2 32x32 accumulators, LDS double
//    buffered, next K-step prefetched into registers while the current one is on the matrix pipe;
//  * temporal taps that fall entirely into clip padding are skipped per tile (T=5: 2/15 of Conv3d);
//  * blockIdx is remapped so that each XCD's L2 sees a contiguous run of tiles (shared halos).
#include "common.h"

namespace {

struct ConvArgs {
  fo_conv_desc d;
  const float* in;
  const float* wp;
  const float* bias;
  const float* mask;
  const float* add;
  float* out;
  int M;          // N*Hm*Wm
  int HWm;        // Hm*Wm
  int tilesM, tilesN;
  int cinChunks;  // Cin/32 (>=1) ; SMALLC: unused
  int cinShift;   // SMALLC: log2(Cin)
  int Ktot;       // taps*Cin  (row length of wp)
  int frameAligned;  // HWm % 128 == 0  -> a tile never straddles frames
};

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_LD = 36;  // padded row (floats)

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN, bool SMALLC>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(WAVES_M * TM * 32 == BM && WAVES_N * TN * 32 == BN, "tile");
  constexpr int BROWS = BN / 32;  // B-tile row passes per thread
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
  float* As0 = lds;
  float* Bs0 = lds + 2 * BM * LDS_LD;

  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;

  // ---- loader coordinates: thread covers rows lrow + 32*i, 16 bytes at column lcol
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;
  int pn[4], py[4], px[4], pt[4];
  bool pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    pv[i] = m < a.M;
    const int mm = pv[i] ? m : 0;
    const int n = mm / a.HWm;
    const int rem = mm - n * a.HWm;
    const int y = rem / d.Wm;
    const int x = rem - y * d.Wm;
    pn[i] = n;
    pt[i] = n % d.T;
    py[i] = y * d.stride - d.padH;
    px[i] = x * d.stride - d.padW;
  }
  const float* wrow = a.wp + (size_t)(tile_n * BN + lrow) * a.Ktot + lcol;

  // ---- K range, with fully padded temporal taps skipped
  const int khw = d.KH * d.KW;
  int kd_lo = 0, kd_hi = d.KD;
  if (d.KD > 1 && a.frameAligned) {
    const int t0 = ((tile_m * BM) / a.HWm) % d.T;
    kd_lo = max(0, d.padD - t0);
    kd_hi = min(d.KD, d.T - t0 + d.padD);
  }
  int step_begin, step_end;
  if (SMALLC) {
    step_begin = 0;
    step_end = a.Ktot / BK;
  } else {
    step_begin = kd_lo * khw * a.cinChunks;
    step_end = kd_hi * khw * a.cinChunks;
  }

  f32x4 ra[4], rb[BROWS];
  const bool in_relu = d.flags & FO_IN_RELU;

  auto load_regs = [&](int step) {
    int kd, kh, kw, coff;
    if (SMALLC) {
      const int k = step * BK + lcol;
      const int tap = k >> a.cinShift;
      coff = k & (d.Cin - 1);
      kd = 0;
      kh = tap / d.KW;
      kw = tap - kh * d.KW;
    } else {
      const int tap = step / a.cinChunks;
      coff = (step - tap * a.cinChunks) * BK + lcol;
      kd = tap / khw;
      const int r = tap - kd * khw;
      kh = r / d.KW;
      kw = r - kh * d.KW;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int it = pt[i] + kd - d.padD;
      const int iy = py[i] + kh;
      const int ix = px[i] + kw;
      const bool ok = pv[i] && (unsigned)it < (unsigned)d.T && (unsigned)iy < (unsigned)d.Hin &&
                      (unsigned)ix < (unsigned)d.Win;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const size_t pix = ((size_t)(pn[i] + kd - d.padD) * d.Hin + iy) * d.Win + ix;
        v = *reinterpret_cast<const f32x4*>(a.in + pix * d.ldIn + coff);
        if (in_relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < BROWS; ++j)
      rb[j] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(32 * j) * a.Ktot + step * BK);
  };
  auto store_lds = [&](int buf) {
    float* As = As0 + buf * BM * LDS_LD;
    float* Bs = Bs0 + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + (lrow + 32 * i) * LDS_LD + lcol) = ra[i];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * j) * LDS_LD + lcol) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (step_begin < step_end) {
    load_regs(step_begin);
    store_lds(0);
  }
  __syncthreads();
  int cur = 0;
  for (int step = step_begin; step < step_end; ++step) {
    const bool more = step + 1 < step_end;
    if (more) load_regs(step + 1);
    const float* As = As0 + cur * BM * LDS_LD + (wm * TM * 32 + l31) * LDS_LD + half * 4;
    const float* Bs = Bs0 + cur * BN * LDS_LD + (wn * TN * 32 + l31) * LDS_LD + half * 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
    if (more) store_lds(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane holds column (co) l31 of each 32x32 tile, rows (r&3)+8*(r>>2)+4*half
  const int flags = d.flags;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = tile_n * BN + (wn * TN + j) * 32 + l31;
    if (co >= d.Cout) continue;
    const float bv = (flags & FO_BIAS) ? a.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = tile_m * BM + (wm * TM + i) * 32 + row;
        if (m >= a.M) continue;
        size_t opix = m;
        if (d.ostride != 1 || d.Hm != d.Hout || d.Wm != d.Wout) {
          const int n = m / a.HWm;
          const int rem = m - n * a.HWm;
          const int y = rem / d.Wm;
          const int x = rem - y * d.Wm;
          opix = ((size_t)n * d.Hout + (y * d.ostride + d.ophH)) * d.Wout + (x * d.ostride + d.ophW);
        }
        float v = acc[i][j][r] + bv;
        if (flags & FO_MASK) v = (a.mask[opix * d.ldMask + co] > 0.f) ? v : 0.f;
        if (flags & FO_ADD) v += a.add[opix * d.ldAdd + co];
        if (flags & FO_OUT_RELU) v = fmaxf(v, 0.f);
        a.out[opix * d.ldOut + co] = v;
      }
    }
  }
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
int launch(const ConvArgs& a, bool smallc, hipStream_t s) {
  const int grid = a.tilesM * a.tilesN;
  if (smallc)
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WAVES_M, WAVES_N, TM, TN, true>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WAVES_M, WAVES_N, TM, TN, false>), dim3(grid), dim3(256), 0, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

}  // namespace

extern "C" int fo_conv_igemm(const fo_conv_desc* d, const float* in, const float* wp, const float* bias,
                             const float* mask, const float* add, float* out, void* stream) {
  ConvArgs a;
  a.d = *d;
  a.in = in; a.wp = wp; a.bias = bias; a.mask = mask; a.add = add; a.out = out;
  FO_REQUIRE(d->N > 0 && d->T > 0 && d->N % d->T == 0, FO_E_SHAPE, "conv: N=%d not a multiple of T=%d", d->N, d->T);
  FO_REQUIRE(d->Cin % 4 == 0 && d->ldIn % 4 == 0, FO_E_ALIGN, "conv: Cin/ldIn must be multiples of 4");
  FO_REQUIRE(fo_aligned16(in) && fo_aligned16(wp), FO_E_ALIGN, "conv: in/wp must be 16-byte aligned");
  FO_REQUIRE(!(d->flags & FO_BIAS) || bias, FO_E_SHAPE, "conv: FO_BIAS without bias");
  FO_REQUIRE(!(d->flags & FO_MASK) || mask, FO_E_SHAPE, "conv: FO_MASK without mask");
  FO_REQUIRE(!(d->flags & FO_ADD) || add, FO_E_SHAPE, "conv: FO_ADD without add");
  const int taps = d->KD * d->KH * d->KW;
  a.Ktot = taps * d->Cin;
  const bool smallc = d->Cin < 32;
  if (smallc) {
    FO_REQUIRE((d->Cin == 8 || d->Cin == 16) && d->KD == 1 && a.Ktot % 32 == 0, FO_E_SHAPE,
               "conv: small Cin=%d needs Cin in {8,16}, 2-D, taps*Cin %% 32 == 0", d->Cin);
    a.cinShift = d->Cin == 8 ? 3 : 4;
    a.cinChunks = 1;
  } else {
    FO_REQUIRE(d->Cin % 32 == 0, FO_E_SHAPE, "conv: Cin=%d must be a multiple of 32 (or 8/16)", d->Cin);
    a.cinChunks = d->Cin / 32;
    a.cinShift = 0;
  }
  a.HWm = d->Hm * d->Wm;
  const long long M = (long long)d->N * a.HWm;
  FO_REQUIRE(M > 0 && M < (1ll << 31), FO_E_SHAPE, "conv: M out of range");
  a.M = (int)M;
  a.tilesM = (a.M + BM - 1) / BM;
  a.frameAligned = (a.HWm % BM) == 0;
  hipStream_t s = (hipStream_t)stream;
  // BN by output channels (filters are packed with Cout rounded up to the same BN)
  if (d->Cout > 64) {
    a.tilesN = (d->Cout + 127) / 128;
    return launch<128, 2, 2, 2, 2>(a, smallc, s);
  } else if (d->Cout > 32) {
    a.tilesN = 1;
    return launch<64, 2, 2, 2, 1>(a, smallc, s);
  } else {
    a.tilesN = 1;
    return launch<32, 4, 1, 1, 1>(a, smallc, s);
  }
}
