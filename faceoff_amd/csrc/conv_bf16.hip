// Implicit-GEMM 2-D convolution on the gfx950 bf16 MFMA (v_mfma_f32_32x32x16_bf16): bf16 operands, fp32
// accumulate, bf16 result.  This is the LPIPS / VGG-16 branch of the perceptual step at BASELINE config 3
// (reference models/lpips.py:115-152: thirteen 3x3 stride-1 convs + ReLU, forward on two image batches and the
// data gradient through one of them; all LPIPS parameters are frozen, lpips.py:63-64, so there is no wgrad).
//
//   out[m][co] = epilogue( sum_{tap,ci} in[ipix(m,tap)][ci] * wp[co][tap][ci] )          (channels-last)
//
// Same shape of kernel as conv_igemm.hip, re-balanced for a matrix pipe that is 16x faster per byte:
//  * K-step = 64 bf16 = the same 128-B row as the fp32 kernel's 32 floats, so the LDS image (144-B padded rows,
//    conflict-free ds_read_b128 / ds_write_b128) and the 16-B-per-lane bounds-checked buffer loads carry over;
//    one ds_read_b128 is exactly one lane's 8-element MFMA fragment (k = 8*(lane>>5) .. +7);
//  * a K-step is only 16 MFMAs x 32 cycles per wave -- far shorter than an L2 round trip -- so the global loads
//    run TWO K-steps ahead through a two-slot register ring (loads of step n+2 are issued while step n is on the
//    matrix pipe and step n+1 sits in the other slot); LDS is double buffered behind it;
//  * zero padding = out-of-range buffer offset (no branch, no select), tap validity is a per-row bitmask;
//  * the fp32 accumulators are transposed through LDS so that bias / ReLU / ReLU-mask / bf16 rounding / stores
//    are 8 channels = 16 B per lane, whole rows per wave.
// Cin must be a multiple of 64, or 8 (the RGB input padded to 8 channels: one 16-B load = one tap of one pixel,
// K = taps rounded up to a multiple of 8, times 8).
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgsH {
  fo_conv_desc d;
  const void* in;
  const void* wp;
  const float* bias;
  const void* mask;
  const void* add;
  void* out;
  // ReLU masks as bit planes (fo_conv_bf16_ex): [pixel][ldBits = Cout / 8] bytes, bit c % 8 of byte c / 8 = (value > 0).  maskBits replaces the FO_MASK
  // read of the bf16 tensor `mask` (1/16 of its bytes); outBits receives the plane of this launch's bf16 result.
  const unsigned char* maskBits;
  unsigned char* outBits;
  int ldBits;
  int M, HWm, tilesM, tilesN;
  int cinChunks;   // K chunks per tap: Cin/64 (128-row kernels), Cin/32 (256-row kernels); SMALLC: unused
  int Ktot;        // elements per filter row (multiple of 64; 256-row kernels: of 32)
  int ksteps;      // Ktot/64 (256-row kernels: Ktot/32)
  int frameTiles;  // 1: a row tile never straddles a frame (HWm % tile rows == 0): depth taps that fall into clip padding are skipped whole
  unsigned inBytes, wpBytes;
};

// Walk of the contraction index in (depth tap, row tap, column tap, channel chunk) order.  Depth taps outside [kd0, kd1) are never
// visited: for a row tile inside ONE frame at position t of its clip these are the taps that see nothing but clip padding
// (reference Conv3d padding=1, models/vqvae_conv3d_latent.py:181: 2 of the 15 (frame, tap) pairs at T = 5).
struct KWalk {
  int q, tap, kd, kh, kw, chunk;      // q: K-steps issued so far
};
__device__ __forceinline__ void kwalk_range(const ConvArgsH& a, int tile_row0, int& kd0, int& kd1) {
  const fo_conv_desc& d = a.d;
  kd0 = 0; kd1 = d.KD;
  if (d.KD > 1 && a.frameTiles) {
    const int t = (tile_row0 / a.HWm) % d.T;
    kd0 = max(0, d.padD - t);
    kd1 = min(d.KD, d.T + d.padD - t);
  }
}
__device__ __forceinline__ KWalk kwalk_begin(const fo_conv_desc& d, int kd0) {
  KWalk w;
  w.q = 0; w.kd = kd0; w.kh = 0; w.kw = 0; w.chunk = 0; w.tap = kd0 * d.KH * d.KW;
  return w;
}
__device__ __forceinline__ void kwalk_next(KWalk& w, const fo_conv_desc& d, int chunks) {
  ++w.q;
  if (++w.chunk == chunks) {
    w.chunk = 0;
    if (w.tap < 31) ++w.tap;
    if (++w.kw == d.KW) {
      w.kw = 0;
      if (++w.kh == d.KH) { w.kh = 0; ++w.kd; }
    }
  }
}
// bit tp of the result: tap tp of output row (frame n at clip position t, tap-(0,0,0) input coordinates py, px) reads a real pixel
__device__ __forceinline__ unsigned tap_mask(const fo_conv_desc& d, bool pv, int t, int py, int px) {
  unsigned mk = 0;
  int tp = 0;
  for (int kd = 0; kd < d.KD; ++kd) {
    const bool okd = pv & ((unsigned)(t + kd - d.padD) < (unsigned)d.T);
    for (int kh = 0; kh < d.KH; ++kh) {
      const bool okh = okd & ((unsigned)(py + kh) < (unsigned)d.Hin);
      for (int kw = 0; kw < d.KW; ++kw, ++tp) mk |= ((okh & ((unsigned)(px + kw) < (unsigned)d.Win)) ? 1u : 0u) << tp;
    }
  }
  return mk;
}
// relu() of two packed bf16
__device__ __forceinline__ unsigned relu_pk(unsigned w) { return w & ~(((w & 0x80008000u) >> 15) * 0xffffu); }

#ifndef FO_ABLATE_PP      // diagnostic builds of conv_bf16_pp16_kernel (tools/ablate_pp16.sh; results are wrong, only the timing is of interest):
#define FO_ABLATE_PP 0    // bit 0 drop the loop's LDS-DMAs, bit 1 drop its MFMAs (one per phase stays), bit 2 one fragment read per operand
#endif
#ifndef FO_ABLATE_PP_LINES
#define FO_ABLATE_PP_LINES 0
#endif
#ifndef FO_ABLATE_H   // diagnostic builds (tools/ablate_bf16.sh): bit 0 drop the loop's global loads, 1 its LDS stores,
#define FO_ABLATE_H 0 // 2 its fragment reads (results are wrong, only the timing is of interest)
#endif
constexpr int BM = 128;
constexpr int ROWB = 144;            // LDS row: 128 B of data + 16 B pad
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ u32x4 bufload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}

// bit e = (o[e] > 0) of eight packed bf16 (bit patterns 0x0001 .. 0x7fff)
__device__ __forceinline__ unsigned pos_bits8(const bf16x8& o) {
  const u32x4 w = __builtin_bit_cast(u32x4, o);
  unsigned b = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b |= ((((w[q] & 0xffffu) - 1u) < 0x7fffu) ? 1u : 0u) << (2 * q);
    b |= ((((w[q] >> 16) - 1u) < 0x7fffu) ? 1u : 0u) << (2 * q + 1);
  }
  return b;
}
__device__ __forceinline__ unsigned pos_bits4(float v0, float v1, float v2, float v3) {
  return (v0 > 0.f ? 1u : 0u) | (v1 > 0.f ? 2u : 0u) | (v2 > 0.f ? 4u : 0u) | (v3 > 0.f ? 8u : 0u);
}
// 1.0 where the bit is set, 0 elsewhere: what the epilogues' (mask > 0) tests read
__device__ __forceinline__ bf16x8 bits_to_mask8(unsigned b) {
  u32x4 w;
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = (((b >> (2 * q)) & 1u) ? 0x3f80u : 0u) | (((b >> (2 * q + 1)) & 1u) ? 0x3f800000u : 0u);
  return __builtin_bit_cast(bf16x8, w);
}
// the FO_MASK fragment of 8 channels co .. co + 7 (co % 8 == 0) of output pixel opix: from the bf16 tensor or from its bit plane
__device__ __forceinline__ bf16x8 load_mask8(const ConvArgsH& a, size_t opix, int co) {
  if (a.maskBits) return bits_to_mask8(a.maskBits[opix * a.ldBits + (co >> 3)]);
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.mask) + opix * a.d.ldMask + co);
}

// One output row's 8 consecutive channels: v = fp32 accumulators + bias -> ReLU-backward mask -> + add (residual / gradient fan-in)
// -> ReLU -> ONE rounding to bf16 (or kept fp32: FO_OUT_F32, the quantisers' inputs and the decoder output)
__device__ __forceinline__ void emit8(const ConvArgsH& a, int flags, float (&v)[8], const bf16x8& mk, const bf16x8& ad, size_t opix, int co, bool ok) {
  if (flags & FO_MASK) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
  }
  if (flags & FO_ADD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)ad[e];
  }
  if (flags & FO_OUT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (!ok) return;
  if (flags & FO_OUT_F32) {
    float* o = reinterpret_cast<float*>(a.out) + opix * a.d.ldOut + co;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.out) + opix * a.d.ldOut + co) = o;
    if (a.outBits) a.outBits[opix * a.ldBits + (co >> 3)] = (unsigned char)pos_bits8(o);
  }
}

// fp32 C tile in LDS (BM x (BN + 4)) -> epilogue, 8 channels per lane
template <int BN>
__device__ __forceinline__ void store_c_tile(const ConvArgsH& a, const float* Cs, int tile_m, int tile_n, int tid) {
  constexpr int C_LD = BN + 4;
  const fo_conv_desc& d = a.d;
  const int flags = d.flags;
  constexpr int C8 = BN / 8;
  constexpr int RPP = 256 / C8;
  const int c8 = tid % C8;
  const int co = tile_n * BN + c8 * 8;
  if (co >= d.Cout) return;
  // FO_DEPTH2SPACE (a k4 s2 p1 transposed conv as ONE GEMM, reference :147-152,222): GEMM column = phase * Cpp + channel (Cpp = Cout / 4 columns per
  // phase, a multiple of 8), so a lane's 8 columns are 8 channels of ONE output pixel: pixel (2y + ph/2 - ophH, 2x + ph%2 - ophH), d.ophW real channels
  const bool d2s = flags & FO_DEPTH2SPACE;
  const int cpp = d.Cout >> 2;
  const int ph = d2s ? co / cpp : 0;
  const int cc = d2s ? co - ph * cpp : co;      // channel of the OUTPUT tensor (bias, mask, add, store)
  if (d2s && cc >= ((d.ophW + 7) & ~7)) return;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = ((flags & FO_BIAS) && cc + e < (d2s ? d.ophW : d.Cout)) ? a.bias[cc + e] : 0.f;
  const bool identity_pix = (d.ostride == 1) & (d.Hm == d.Hout) & (d.Wm == d.Wout) & !d2s;
  const __bf16* mask = reinterpret_cast<const __bf16*>(a.mask);
  const __bf16* addp = reinterpret_cast<const __bf16*>(a.add);
  const int cst = cc;                         // channel offset of the store
  // rows in batches: all mask / add loads of a batch are in flight before the first is used (see conv_igemm.hip store_tile)
  constexpr int ROWS = BM / RPP, R = ROWS < 8 ? ROWS : 8;
  const int r0 = tid / C8;
#pragma unroll
  for (int b = 0; b < ROWS; b += R) {
    size_t opix[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int m = tile_m * BM + r0 + (b + r) * RPP;
      ok[r] = m < a.M;
      m = min(m, a.M - 1);
      opix[r] = m;
      if (!identity_pix) {
        const int n = m / a.HWm;
        const int rem = m - n * a.HWm;
        const int y = rem / d.Wm;
        const int x = rem - y * d.Wm;
        int oy = y * d.ostride + d.ophH, ox = x * d.ostride + d.ophW;
        if (d2s) {
          oy = 2 * y + (ph >> 1) - d.ophH;
          ox = 2 * x + (ph & 1) - d.ophH;
          ok[r] = ok[r] & ((unsigned)oy < (unsigned)d.Hout) & ((unsigned)ox < (unsigned)d.Wout);
          oy = min(max(oy, 0), d.Hout - 1);
          ox = min(max(ox, 0), d.Wout - 1);
        }
        opix[r] = ((size_t)n * d.Hout + oy) * d.Wout + ox;
      }
    }
    bf16x8 mk[R], ad[R];
    if (flags & FO_MASK) {
#pragma unroll
      for (int r = 0; r < R; ++r) mk[r] = load_mask8(a, opix[r], cc);
    }
    if (flags & FO_ADD) {
#pragma unroll
      for (int r = 0; r < R; ++r) ad[r] = *reinterpret_cast<const bf16x8*>(addp + opix[r] * d.ldAdd + cc);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* crow = Cs + (r0 + (b + r) * RPP) * C_LD + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(crow);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(crow + 4);
      float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
      emit8(a, flags, v, mk[r], ad[r], opix[r], cst, ok[r]);
    }
  }
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN, bool SMALLC, bool INRELU>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const ConvArgsH a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(WAVES_M * TM * 32 == BM && WAVES_N * TN * 32 == BN, "tile");
  constexpr int BROWS = BN / 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * ROWB];
  unsigned char* As0 = lds;
  unsigned char* Bs0 = lds + 2 * BM * ROWB;

  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: LDS-DMA destinations / wave roles must be scalar)
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;
  const int ntaps = d.KD * d.KH * d.KW;
  int kd0, kd1;
  kwalk_range(a, tile_m * BM, kd0, kd1);
  const int nsteps = SMALLC ? a.ksteps : (kd1 - kd0) * d.KH * d.KW * a.cinChunks;

  // ---- loader coordinates: thread covers rows lrow + 32*i, 16 bytes (8 elements) at byte column lcolB
  const int lrow = tid >> 3;
  const int lcolB = (tid & 7) * 16;
  int rowoff[4];
  unsigned tapmask[4];
  int py[4], px[4], pbase[4];
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
    py[i] = y * d.stride - d.padH;
    px[i] = x * d.stride - d.padW;
    pbase[i] = ((n - d.padD) * d.Hin + py[i]) * d.Win + px[i];      // pixel index of tap (0,0,0)
    rowoff[i] = pbase[i] * d.ldIn * 2 + lcolB;
    tapmask[i] = SMALLC ? 0u : tap_mask(d, pv[i], n % d.T, py[i], px[i]);
  }

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  const unsigned wrow = (unsigned)(((size_t)(tile_n * BN + lrow) * a.Ktot) * 2 + lcolB);
  const unsigned wstride32 = (unsigned)((size_t)32 * a.Ktot * 2);

  // incremental (tap, chunk) walk of the next step to load (wave-uniform scalars)
  KWalk kw_ = kwalk_begin(d, kd0);
  // SMALLC: this thread's tap of the step being loaded
  int sc_off = 0, sc_kh = 0, sc_kw = 0;
  bool sc_ok = false;

  u32x4 ra[2][4], rb[2][BROWS];

  auto load_step = [&](int slot) {
    if (SMALLC) {
      const int tap = kw_.q * 8 + (tid & 7);
      sc_kh = tap / d.KW;
      sc_kw = tap - sc_kh * d.KW;
      sc_ok = tap < ntaps;
      sc_off = (sc_kh * d.Win + sc_kw) * d.ldIn * 2;
    }
    const int stepoff = ((((kw_.kd * d.Hin) + kw_.kh) * d.Win + kw_.kw) * d.ldIn + kw_.chunk * 64) * 2;
    const int kpos = SMALLC ? kw_.q : kw_.tap * a.cinChunks + kw_.chunk;      // 64-element position inside a filter row
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (SMALLC) {
        const bool ok = sc_ok & pv[s] & ((unsigned)(py[s] + sc_kh) < (unsigned)d.Hin) & ((unsigned)(px[s] + sc_kw) < (unsigned)d.Win);
        ra[slot][s] = bufload16(rin, ok ? (unsigned)(pbase[s] * d.ldIn * 2 + sc_off) : OOB);
      } else {
        const unsigned pad = (((tapmask[s] >> kw_.tap) & 1u) - 1u) & OOB;     // padding tap -> beyond the descriptor -> zeros
        ra[slot][s] = bufload16(rin, (unsigned)(rowoff[s] + stepoff) | pad);
      }
      if (s < BROWS) rb[slot][s < BROWS ? s : 0] = bufload16(rwp, kw_.q < nsteps ? wrow + s * wstride32 + kpos * 128 : OOB);
    }
    if (SMALLC) ++kw_.q; else kwalk_next(kw_, d, a.cinChunks);
  };
  auto store_step = [&](int slot, int buf, int s_lo = 0, int s_hi = 4) {
    unsigned char* As = As0 + buf * BM * ROWB;
    unsigned char* Bs = Bs0 + buf * BN * ROWB;
#pragma unroll
    for (int s = s_lo; s < s_hi; ++s) {
      if (INRELU) {                   // the leading ReLU of a ResBlock (reference :91), applied as the operand is staged
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[slot][s][e] = relu_pk(ra[slot][s][e]);
      }
      *reinterpret_cast<u32x4*>(As + (lrow + 32 * s) * ROWB + lcolB) = ra[slot][s];
      if (s < BROWS) *reinterpret_cast<u32x4*>(Bs + (lrow + 32 * s) * ROWB + lcolB) = rb[slot][s < BROWS ? s : 0];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K-step `buf` on the matrix pipe; the LDS stores of the NEXT step (ring slot st_slot -> buffer st_buf, loaded a
  // full step ago) ride in its second half, between the MFMAs, instead of in front of the barrier
  auto compute = [&](int buf, int st_slot, int st_buf) {
    const unsigned char* As = As0 + buf * BM * ROWB + (wm * TM * 32 + l31) * ROWB + half * 16;
    const unsigned char* Bs = Bs0 + buf * BN * ROWB + (wn * TN * 32 + l31) * ROWB + half * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 fa[TM], fb[TN];
      const int sr = (FO_ABLATE_H & 4) ? 0 : s;
      if ((FO_ABLATE_H & 4) && s > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, ra[0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = __builtin_bit_cast(bf16x8, ra[1][j]);
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROWB + sr * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROWB + sr * 32);
      }
      if (s >= 2 && !(FO_ABLATE_H & 2)) store_step(st_slot, st_buf, 2 * (s - 2), 2 * (s - 2) + 2);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  // ring slot 0 <- step 0, slot 1 <- step 1; LDS buffer 0 <- step 0
  load_step(0);
  load_step(1);
  store_step(0, 0);
  __syncthreads();
  for (int step = 0; step < nsteps; step += 2) {
    if (!(FO_ABLATE_H & 1)) load_step(0);                 // step + 2
    compute(0, 1, 1);             // step; stores step + 1
    __syncthreads();
    if (step + 1 >= nsteps) break;
    if (!(FO_ABLATE_H & 1)) load_step(1);                 // step + 3
    compute(1, 0, 0);             // step + 1; stores step + 2
    __syncthreads();
  }

  // ---- epilogue through LDS (fp32 tile), 8 channels = 16 B of bf16 per lane
  constexpr int C_LD = BN + 4;
  static_assert(BM * C_LD * 4 <= 2 * (BM + BN) * ROWB, "C tile must fit the staging LDS");
  float* Cs = reinterpret_cast<float*>(lds);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        Cs[row * C_LD + (wn * TN + j) * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();
  store_c_tile<BN>(a, Cs, tile_m, tile_n, tid);
}

// ------------------------------------------------------------------------------------------------ 128-row tiles, LDS-DMA staging
// conv_bf16_kernel's tile (128 x BN, 4 waves, 2-3 workgroups per CU) with the operands DMA'd global -> LDS instead of staged
// through registers: at BN = 64 the register form spends 311 LDS cycles per K-step on its ds_write_b128 (24 KB at ~79 B/clk)
// against 256 cycles of MFMA -- LDS-store-bound; a DMA has no VGPR -> LDS transfer at all.  LDS rows are unpadded 128 B (one
// 64-deep K-step of one GEMM row = eight 16-B chunks), chunk c of row r at position c ^ ((r >> 1) & 7), applied on the DMA
// source address and on the fragment read.  One barrier per K-step (the DMA of step n+1 is issued before the MFMAs of step n
// and drained by the vmcnt(0) of __syncthreads()): the other workgroups on the CU cover the drain.
typedef __attribute__((address_space(3))) unsigned char lds_byte;
// (kept out of the kernel templates: hipcc's host pass drops a kernel template whose value-dependent body holds this builtin)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_byte* dst, unsigned voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voffset, 0, 0, 0);
}
// (per-lane part in a VGPR that is the same for every instruction, wave-uniform part in an SGPR: the range check sees the VGPR part only)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t r, lds_byte* dst, unsigned voffset, unsigned soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_bf16_dma_kernel(const ConvArgsH a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(WAVES_M * TM * 32 == BM && WAVES_N * TN * 32 == BN, "tile");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NPB = BN / 32;                            // B pieces (8 rows x 128 B) per wave and K-step
  constexpr int C_LD = BN + 4;
  constexpr int LDSB = 2 * STAGE > BM * C_LD * 4 ? 2 * STAGE : BM * C_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: LDS-DMA destinations / wave roles must be scalar)
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;
  int kd0, kd1;
  kwalk_range(a, tile_m * BM, kd0, kd1);
  const int nsteps = (kd1 - kd0) * d.KH * d.KW * a.cinChunks;

  // ---- DMA roles: piece i of wave w = tile rows (i*4 + w)*8 .. +7, lane = (row % 8, chunk position)
  const int drow = lane >> 3, dpos = lane & 7;
  int rowoffA[4];
  unsigned tapmaskA[4], woffB[NPB];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + drow;
    const int chunk = dpos ^ ((row >> 1) & 7);
    const int m = tile_m * BM + row;
    const bool pv = m < a.M;
    const int mm = pv ? m : 0;
    const int n = mm / a.HWm;
    const int rem = mm - n * a.HWm;
    const int y = rem / d.Wm;
    const int x = rem - y * d.Wm;
    const int py = y * d.stride - d.padH, px = x * d.stride - d.padW;
    rowoffA[i] = (((n - d.padD) * d.Hin + py) * d.Win + px) * d.ldIn * 2 + chunk * 16;
    tapmaskA[i] = tap_mask(d, pv, n % d.T, py, px);
  }
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int row = (i * 4 + wave) * 8 + drow;
    woffB[i] = (unsigned)(((size_t)(tile_n * BN + row) * a.Ktot) * 2 + (dpos ^ ((row >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;

  KWalk kw_ = kwalk_begin(d, kd0);
  auto dma_step = [&](int stage) {
    const int stepoff = ((((kw_.kd * d.Hin) + kw_.kh) * d.Win + kw_.kw) * d.ldIn + kw_.chunk * 64) * 2;
    const int kpos = kw_.tap * a.cinChunks + kw_.chunk;
    lds_byte* const sa = lds3 + stage * STAGE;
    lds_byte* const sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned pad = (((tapmaskA[i] >> kw_.tap) & 1u) - 1u) & OOB;      // padding tap / row past M -> zeros
      dma16(rin, sa + (i * 4 + wave) * 1024, (unsigned)(rowoffA[i] + stepoff) | pad);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) dma16(rwp, sb + (i * 4 + wave) * 1024, kw_.q < nsteps ? woffB[i] + kpos * 128 : OOB);
    kwalk_next(kw_, d, a.cinChunks);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: row = block row + l31 (blocks are multiples of 32 rows), logical chunk 2s + half
  const int xr = (l31 >> 1) & 7;
  int fpos[4];
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) fpos[sidx] = ((2 * sidx + half) ^ xr) * 16;

  dma_step(0);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    dma_step(cur ^ 1);                                    // step + 1 (past the end: zeros, never read)
    const unsigned char* As = lds + cur * STAGE + (wm * TM * 32 + l31) * 128;
    const unsigned char* Bs = lds + cur * STAGE + BM * 128 + (wn * TN * 32 + l31) * 128;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * 128 + fpos[sidx]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * 128 + fpos[sidx]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                      // (implies vmcnt(0): this wave's DMAs of step + 1 have landed)
  }

  float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        Cs[row * C_LD + (wn * TN + j) * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();
  store_c_tile<BN>(a, Cs, tile_m, tile_n, tid);
}

// ------------------------------------------------------------------------------------------------ RGB layer (8 -> 64 channels, 3x3)
// VGG's first conv has K = 9 taps x 8 padded channels = 72: as a tiled GEMM it is two K-steps per tile, i.e. all prologue and
// epilogue (0.89 ms for 1.34 GB of output).  Here nothing is staged: with K ordered (tap, channel) one MFMA operand fragment of
// v_mfma_f32_32x32x16_bf16 (8 consecutive k per lane) IS one pixel's 16-byte channel vector at one tap, so a lane loads its
// fragment straight from global memory (padding: out-of-range offset -> zeros); the 64 x 72 filter lives in 40 VGPRs per
// lane for the whole kernel; a wave walks blocks of 32 consecutive pixels (persistent grid), 5 loads and 10 MFMAs per block.
// Operands are swapped (C^T = W X^T: rows = output channels, columns = pixels) so that a lane's accumulator holds 4 consecutive
// channels of ONE pixel: bias + ReLU + bf16 pack per lane, a wave-local 4.5 KB LDS patch turns that into 16-byte stores of
// whole 128-byte pixel lines.  0.92 -> 0.44 ms.  (The same trick does NOT carry to the layer's 64 -> 3 data gradient: there a
// pixel is 128 B, adjacent lanes of a fragment load are 128 B apart, and the L1 serves one line per lane: 1.28 vs 1.18 ms tiled.)
template <bool BITS>      // BITS: also the bit plane of the result (a.outBits): one byte per lane and store, through its own descriptor (no branch in the loop)
// (launch bound: the host launches THREE workgroups per CU; without the second argument the kernel compiled to 177-189 VGPRs = two per CU, the third
// queued behind them -- round 5: 159-162 VGPRs, no spills, 0.430 -> 0.391 ms and 0.465 -> 0.414 with the bit plane; four per CU spills 80 registers)
__global__ __launch_bounds__(256, 3) void conv_rgb_bf16_kernel(const ConvArgsH a, int nblocks) {
  constexpr int PITCH = 128 + 16;                         // patch row: one pixel's 64 bf16 + pad
  __shared__ __attribute__((aligned(16))) unsigned char patch_all[4][32 * PITCH];
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: LDS-DMA destinations / wave roles must be scalar)
  const int l31 = lane & 31, half = lane >> 5;
  unsigned char* patch = patch_all[wave];
  if (tid < 64) bias_s[tid] = (d.flags & FO_BIAS) ? a.bias[tid] : 0.f;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  // filter fragments: row (output channel) i*32 + l31, k = s*16 + half*8 .. +8  (taps 2s + half; tap 9 = zeros in the pack)
  bf16x8 wf[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 5; ++s)
      wf[i][s] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.wp) + (size_t)(i * 32 + l31) * a.Ktot + s * 16 + half * 8);
  // this lane's taps
  int tdy[5], tdx[5], toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int t = 2 * s + half;
    tdy[s] = t / 3 - 1;
    tdx[s] = t % 3 - 1;
    toff[s] = t < 9 ? (tdy[s] * d.Win + tdx[s]) * 16 : (int)OOB;
  }
  const bool relu = d.flags & FO_OUT_RELU;
  __bf16* out = reinterpret_cast<__bf16*>(a.out);
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;

  auto load_block = [&](int blk, bf16x8 (&xf)[5]) {
    const int m = blk * 32 + l31;
    const bool pv = m < a.M;
    const int mm = pv ? m : 0;
    const int n = mm / a.HWm;
    const int rem = mm - n * a.HWm;
    const int y = rem / d.Wm;
    const int x = rem - y * d.Wm;
    const int base = mm * 16;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const bool ok = pv & ((unsigned)(y + tdy[s]) < (unsigned)d.Hin) & ((unsigned)(x + tdx[s]) < (unsigned)d.Win) & (toff[s] != (int)OOB);
      xf[s] = __builtin_bit_cast(bf16x8, bufload16(rin, ok ? (unsigned)(base + toff[s]) : OOB));
    }
  };

  // No data-dependent branch in the loop (hipcc otherwise drains every outstanding load, vmcnt(0), in front of each store): blocks
  // past the end load zeros through out-of-range offsets (load_block's pv) and their stores are dropped the same way.
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(((size_t)(a.M - 1) * d.ldOut + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rbits = __builtin_amdgcn_make_buffer_rsrc(BITS ? a.outBits : nullptr, 0, BITS ? (unsigned)((size_t)a.M * 8) : 0u, 0x00020000);
  const float lo = relu ? 0.f : -__builtin_inff();
  bf16x8 xa[5], xb[5];
  load_block(gw, xa);
  const int iters = (nblocks - gw + 2 * nw - 1) / (2 * nw);          // (uniform per wave)
  for (int itn = 0, blk = gw; itn < iters; ++itn, blk += 2 * nw) {
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
      const int cur = blk + phase * nw;
      bf16x8 (&xc)[5] = phase ? xb : xa;
      bf16x8 (&xn)[5] = phase ? xa : xb;
      load_block(cur + nw, xn);                            // next block's fragments fly during this block's MFMAs and stores
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][s], xc[s], acc[i], 0, 0, 0);
      // rows of acc[i] = channels i*32 + (r & 3) + 8 (r >> 2) + 4 half, column = pixel l31
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = i * 32 + 8 * g + 4 * half;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + c0);
          const float v0 = fmaxf(acc[i][4 * g + 0] + b4.x, lo), v1 = fmaxf(acc[i][4 * g + 1] + b4.y, lo);
          const float v2 = fmaxf(acc[i][4 * g + 2] + b4.z, lo), v3 = fmaxf(acc[i][4 * g + 3] + b4.w, lo);
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          bf16x4 o = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
          *reinterpret_cast<bf16x4*>(patch + l31 * PITCH + c0 * 2) = o;
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pix = it * 8 + (lane >> 3), piece = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(patch + pix * PITCH + piece * 16);
        const long long m = (long long)cur * 32 + pix;
        __builtin_amdgcn_raw_buffer_store_b128(v, rout, m < a.M ? (unsigned)((m * d.ldOut + piece * 8) * 2) : OOB, 0, 0);
        if (BITS) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)pos_bits8(__builtin_bit_cast(bf16x8, v)), rbits, m < a.M ? (unsigned)(m * 8 + piece) : OOB, 0, 0);
      }
    }
  }
}

// The RGB layer's data gradient (VGG conv1_1 backwards: 64 -> 3 channels, padded to 8).  The tiled kernel spends a 32-column MFMA
// on 3 columns and re-fetches every gradient pixel nine times (1.1 ms).  Here a workgroup (4 waves, two per CU) walks 64-pixel row segments: the three
// gradient rows a segment needs (with a one-pixel halo) are DMA'd into LDS once (double-buffered: the next segment's rows fly
// during this one's MFMAs), all nine taps read their fragments from there, and the contraction runs on v_mfma_f32_16x16x32_bf16
// with the filter as the ROW operand (16 rows, 3 real; its 18 fragments live in registers) and 16 pixels as columns, so a lane's
// accumulator is 4 consecutive channels of one pixel: lanes 0-31 store 8 bytes each, 256 contiguous bytes per wave.
// LDS rows are pixels of 128 B; chunk c of pixel p sits at position c ^ ((p >> 1) & 7) (DMA source side + fragment read).
__global__ __launch_bounds__(256, 2) void conv_rgb_dgrad_bf16_kernel(const ConvArgsH a, int nseg, int segsPerRow) {
  constexpr int SEG = 64, PPR = 9, PIECES = 28, ROWPX = PPR * 8;   // 8-pixel pieces per stage: 3 rows x 9 (66 pixels + pad), 4 waves x 7
  constexpr int STAGE = PIECES * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: LDS-DMA destinations / wave roles must be scalar)
  const int l15 = lane & 15, quad = lane >> 4;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(((size_t)(a.M - 1) * d.ldOut + 8) * 2), 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;
  // filter fragments: row (output channel) l15, K-step st = (tap, 32-channel half): k = st*32 + quad*8
  bf16x8 wf[18];
#pragma unroll
  for (int st = 0; st < 18; ++st)
    wf[st] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.wp) + (size_t)l15 * a.Ktot + st * 32 + quad * 8);
  // DMA roles: piece id = i*8 + wave (i < 7) = LDS row id / 17, pixels 8*(id % 17) .. +7; lane = (pixel % 8, chunk position)
  int prow[7], ppix[7];
  unsigned pchunk[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int id = i * 4 + wave;
    prow[i] = id < 3 * PPR ? id / PPR : -1000000;         // the piece past the third row: always out of range (zeros into the padding)
    ppix[i] = (id % PPR) * 8 + (lane >> 3);
    pchunk[i] = (unsigned)(((lane & 7) ^ ((ppix[i] >> 1) & 7)) * 16);
  }
  auto dma_segment = [&](int seg, int stage) {
    const int rowid = seg / segsPerRow;                    // (frame, image row), uniform
    const int xs = (seg - rowid * segsPerRow) * SEG;
    const int n = rowid / d.Hin, y = rowid - n * d.Hin;
    const bool sv = seg < nseg;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int iy = y - 1 + prow[i], ix = xs - 1 + ppix[i];
      const bool ok = sv & ((unsigned)iy < (unsigned)d.Hin) & ((unsigned)ix < (unsigned)d.Win);
      dma16(rin, lds3 + stage * STAGE + (i * 4 + wave) * 1024, ok ? (unsigned)(((n * d.Hin + iy) * d.Win + ix) * 128) + pchunk[i] : OOB);
    }
  };
  // fragment addressing: output pixel of this lane = wave*16 + l15; tap (kh, kw) reads LDS row kh, pixel + kw; chunk h*4 + quad
  const int px = wave * 16 + l15;
  int foff[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = px + kw;
      foff[kw][h] = p * 128 + (((h * 4 + quad) ^ ((p >> 1) & 7)) * 16);
    }
  int seg = blockIdx.x;
  dma_segment(seg, 0);
  __syncthreads();
  for (int it = 0; seg < nseg; seg += gridDim.x, ++it) {
    const int cur = it & 1;
    dma_segment(seg + gridDim.x, cur ^ 1);
    const unsigned char* S = lds + cur * STAGE;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(S + kh * (ROWPX * 128) + foff[kw][h]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(kh * 3 + kw) * 2 + h], xf, acc, 0, 0, 0);
        }
    // acc[r] = channel quad*4 + r of pixel px
    const int rowid = seg / segsPerRow;
    const long long m = (long long)rowid * d.Win + (seg - rowid * segsPerRow) * SEG + px;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bf16x4 o = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rout, quad < 2 ? (unsigned)((m * d.ldOut + quad * 4) * 2) : OOB, 0, 0);
    __syncthreads();                                      // (vmcnt(0): the next segment's rows have landed; this stage is free again)
  }
}

// Round 5: the same contraction with every gradient row fetched ONCE.  The kernel above re-fetches the three rows of each 64-pixel segment
// (2.98 GB read per launch for a 1.34 GB operand, profiles/r05_c3_pmc.md).  Here a workgroup walks a contiguous run of output rows DOWN a
// 64-pixel column strip; the rows of the gradient live in a ring of RS = 8 LDS slots (72 pixels x 128 B each), every step DMAs ONE new row --
// AHEAD = 5 rows in front of the row the step needs last, so that ~45 KB per workgroup stay in flight (one row ahead, as the per-segment
// prefetch above, would be 9 KB: latency-bound) -- waits with a counted vmcnt (loads, LDS-DMAs and stores retire in order) and one raw
// s_barrier per step, and reads the three rows of its output row from their slots.  Strips are laid end to end as VIRTUAL rows: strip
// (frame, 64-pixel column) contributes H + 2 of them (a zero row above and below), so a step always advances by one row; the two centre
// rows per strip that are padding store nothing.  Same MFMA order per pixel as the kernel above: bit-identical results.
__global__ __launch_bounds__(256, 2) void conv_rgb_dgrad_ring_bf16_kernel(const ConvArgsH a, int nstrips, int segsPerRow) {
  constexpr int SEG = 64, PPR = 9, ROWB = PPR * 1024, RS = 8, AHEAD = 5;      // a row slot: 9 pieces of 8 pixels x 128 B
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // RS * ROWB
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(((size_t)(a.M - 1) * d.ldOut + 8) * 2), 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;
  bf16x8 wf[18];
#pragma unroll
  for (int st = 0; st < 18; ++st)
    wf[st] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.wp) + (size_t)l15 * a.Ktot + st * 32 + quad * 8);
  // DMA roles: piece id = i * 4 + wave of a row's nine: wave 0 issues three per row, waves 1-3 two (an LDS-DMA costs its wave 100+ clocks of issue
  // whether it fetches or zero-fills, so there are no dummy pieces: the counted waits below differ by wave instead)
  const int npw = wave == 0 ? 3 : 2;
  int ppix[3];
  unsigned pchunk[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int id = i * 4 + wave;
    ppix[i] = id * 8 + (lane >> 3);
    pchunk[i] = (unsigned)(((lane & 7) ^ ((ppix[i] >> 1) & 7)) * 16);
  }
  const int VH = d.Hin + 2;                                                 // virtual rows per strip
  const int G = nstrips * VH;
  auto issue_row = [&](int g) {                                              // virtual row g -> slot g % RS
    const int strip = g / VH, v = g - strip * VH;
    const int n = strip / segsPerRow, xs = (strip - n * segsPerRow) * SEG;
    const int iy = v - 1;
    const bool rowok = (g >= 0) & (g < G) & ((unsigned)iy < (unsigned)d.Hin);
    const int slot = ((g % RS) + RS) % RS;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < npw) {                                                         // (wave-uniform)
        const int ix = xs - 1 + ppix[i];
        const bool ok = rowok & ((unsigned)ix < (unsigned)d.Win);
        dma16(rin, lds3 + slot * ROWB + (i * 4 + wave) * 1024, ok ? (unsigned)(((n * d.Hin + iy) * d.Win + ix) * 128) + pchunk[i] : OOB);
      }
    }
  };
  const int px = wave * 16 + l15;
  int foff[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = px + kw;
      foff[kw][h] = p * 128 + (((h * 4 + quad) ^ ((p >> 1) & 7)) * 16);
    }
  // this workgroup's centres [cA, cB): centre c reads virtual rows c - 1, c, c + 1
  const int per = (G + gridDim.x - 1) / gridDim.x;
  const int cA = blockIdx.x * per, cB = min(G, cA + per);
  if (cA >= cB) return;
#pragma unroll 1
  for (int g = cA - 1; g < cA + AHEAD; ++g) issue_row(g);                    // rows cA - 1 .. cA + 4: AHEAD + 1 rows
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll 1
  for (int c = cA; c < cB; ++c) {
    issue_row(c + AHEAD);
    // behind row c + 1's pieces this wave has issued: rows c + 2 .. c + 5 (4 npw) and, from the fifth step on, the stores of steps c - 4 .. c - 1 (4)
    if (wave == 0) {
      if (c - cA >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      if (c - cA >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                            // every wave's pieces of row c + 1 have landed; everybody is done with row c - 2's slot
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int g = c - 1 + kh;
      const unsigned char* S = lds + (((g % RS) + RS) % RS) * ROWB;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(S + foff[kw][h]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(kh * 3 + kw) * 2 + h], xf, acc, 0, 0, 0);
        }
    }
    const int strip = c / VH, v = c - strip * VH;
    const int n = strip / segsPerRow, xs = (strip - n * segsPerRow) * SEG, y = v - 1;
    const bool rowok = (unsigned)y < (unsigned)d.Hin;
    const long long m = ((long long)n * d.Hin + (rowok ? y : 0)) * d.Win + xs + px;
    const bf16x4 o = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
    // (exactly one store per lane and step, real or out of range: the counted wait above depends on it)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rout, (rowok && quad < 2) ? (unsigned)((m * d.ldOut + quad * 4) * 2) : OOB, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // (the zero-fill DMAs of the rows past the run's end)
}

// ------------------------------------------------------------------------------------------------ big tiles: LDS-DMA ring + ping-pong
// 256 x BN output tile (BN = 256 or 128), 8 waves, one workgroup per CU, for the layers with >= 128 output channels and a
// launch of >= 2-3 rounds of tiles.  What it changes against conv_bf16_kernel above, each measured in one process on one device
// (tools/ab_bf16.py, conv4_2 forward, 160 frames): 256 x 256 register-staged tiles 0.710 ms -> operands DMA'd global -> LDS
// (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write pass) 0.671 -> two wave groups in ping-pong 0.609 ->
// v_mfma_f32_16x16x32_bf16 instead of 32x32x16 0.576 ms (the kernel is power-bound: MFMA-busy 0.55 -> 0.70 took the clock from
// 1.65 to 1.49 GHz; the 16x16x32 shape holds 1.63 GHz at 0.66 busy).
//
// Structure: the K loop runs in 32-deep tiles ("phases"), each phase = { TM + TN ds_read_b128 | barrier | TM x TN MFMA | barrier },
// and the two wave groups (G0 = waves 0-3, G1 = waves 4-7: one wave of each per SIMD) run ONE BARRIER APART -- G1 executes one
// extra s_barrier up front -- so in every inter-barrier segment one group issues MFMAs while the other does its fragment reads
// and its DMA issue.  Operands sit in a ring of four slots (tile p in slot p % 4).  An LDS-DMA writes 64 lanes x 16 B LINEARLY
// from a wave-uniform base (16 rows x 64 B per instruction), so rows cannot be padded; the bank-conflict fix is a swizzle
// applied on the SOURCE side (lane = row l / 4, LDS chunk position l % 4 fetches source chunk (l % 4) ^ SWZ(row)) and again on
// the fragment read, never on the LDS destination.  A lane's A/B fragment is row (lane & 15), 16-B chunk (lane >> 4) of the
// 64-B tile row: one ds_read_b128 per 16-row block and phase; with SWZ(row) = {0, 2, 3, 1}[(row >> 2) & 3] the ds_read_b128 lane
// groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) cover all sixteen 16-B slots of the 256-B bank row once.
//
// Staging protocol, with global segments numbered s = 0, 1, ... (G0: phase p reads in s = 2p, MFMAs in 2p+1; G1: reads in 2p+1,
// MFMAs in 2p+2); NP = DMA instructions per wave and tile; every DMA sits in its wave's READ segment (its issue cost, ~100
// cycles per 1-KB piece, is then in the shadow of the other group's MFMAs):
//     G1, segment 2p+1: issue its rows of tile p+3 (slot of tile p-1: the last reads of that slot, G1's own in segment 2p-1,
//       were retired by the lgkmcnt(0) in front of its MFMAs in segment 2p, before the barrier that opens 2p+1), then
//       s_waitcnt vmcnt(2 NP): only tiles p+2, p+3 may still fly, so its rows of tile p+1 have landed;
//     G0, segment 2p: issue its rows of tile p+2 (slot of tile p-2, last read in segment 2p-3, retired in 2p-2); after its
//       MFMAs, at the end of segment 2p+1, s_waitcnt vmcnt(NP): tile p+2 may fly, its rows of tile p+1 have landed
// -- both before the barrier that opens segment 2p+2, where G0 reads tile p+1 first.  A tile is in flight for four to five
// segments (2000-2500 cycles) and nothing is drained to vmcnt(0) inside the loop; raw s_barrier only (__syncthreads() would
// drain).  Tiles past the end are DMA'd as zeros (out-of-range offsets) so the counts stay constant.  Barrier counts: G0
// 1 + 2 nt + 2, G1 2 + 2 nt + 1.
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
template <int N> __device__ __forceinline__ void wait_vmcnt();              // s_waitcnt takes an immediate
template <> __device__ __forceinline__ void wait_vmcnt<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<5>() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<10>() { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }

template <int BMB, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(512, 1) void conv_bf16_pp16_kernel(const ConvArgsH a) {
  static_assert(WAVES_M * WAVES_N == 8 && (BMB == 256 || BMB == 512) && (BN == 256 || BN == 128), "8 waves");
  constexpr int TM = BMB / WAVES_M / 16, TN = BN / WAVES_N / 16;     // 16 x 16 blocks per wave
  constexpr int WCOLS = TN * 16;
  static_assert(WCOLS == 64 || WCOLS == 128, "the epilogue stores 64- or 128-column wave tiles");
  static_assert(TM % 2 == 0, "the epilogue walks pairs of 16-row blocks");
  constexpr int SLOT = (BMB + BN) * 64;                   // bytes per ring slot
  constexpr int NPB = BN / 128;                           // B pieces per wave and tile
  constexpr int NPA = BMB / 128;                          // A pieces (16 rows x 64 B) per wave and tile
  constexpr int NP = NPA + NPB;
  constexpr int C_LD = WCOLS + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: LDS-DMA destinations / wave roles must be scalar)
  const int l15 = lane & 15, quad = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const bool g1 = __builtin_amdgcn_readfirstlane(wave >> 2) != 0;
  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;
  const int chunks32 = a.cinChunks;                        // (32-deep chunks per tap)
  int kd0, kd1;
  kwalk_range(a, tile_m * BMB, kd0, kd1);
  const int nt = (kd1 - kd0) * d.KH * d.KW * chunks32;    // 32-deep K tiles of this row tile
  const int drow = lane >> 2, dpos = lane & 3;
  int rowoffA[NPA];
  unsigned tapmaskA[NPA], woffB[NPB];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
#if FO_ABLATE_PP_LINES      // diagnostic: a piece fetches 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B -- same bytes, half the lines; wrong results
    const int row = (i * 8 + wave) * 16 + (lane >> 3) * 2;
    const int chunk = lane & 7;
#else
    const int row = (i * 8 + wave) * 16 + drow;
    const int chunk = dpos ^ swz(row);
#endif
    const int m = tile_m * BMB + row;
    const bool pv = m < a.M;
    const int mm = pv ? m : 0;
    const int n = mm / a.HWm;
    const int rem = mm - n * a.HWm;
    const int y = rem / d.Wm;
    const int x = rem - y * d.Wm;
    const int py = y - d.padH, px = x - d.padW;
    rowoffA[i] = (((n - d.padD) * d.Hin + py) * d.Win + px) * d.ldIn * 2 + chunk * 16;
    tapmaskA[i] = tap_mask(d, pv, n % d.T, py, px);
  }
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int row = (i * 8 + wave) * 16 + drow;
    woffB[i] = (unsigned)(((size_t)(tile_n * BN + row) * a.Ktot) * 2 + (dpos ^ swz(row)) * 16);
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;

  KWalk kw_ = kwalk_begin(d, kd0);
  auto dma_tile = [&]() {
    const int stepoff = ((((kw_.kd * d.Hin) + kw_.kh) * d.Win + kw_.kw) * d.ldIn + kw_.chunk * 32) * 2;
    const int kpos = kw_.tap * chunks32 + kw_.chunk;       // 32-element position inside a filter row
    lds_byte* const sa = lds3 + (kw_.q & 3) * SLOT;
    lds_byte* const sb = sa + BMB * 64;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      // padding tap / row past M -> zeros; K tiles past the end: whatever the walk points at (never multiplied in: the loop ends first)
      unsigned pad = (((tapmaskA[i] >> kw_.tap) & 1u) - 1u) & OOB;
      if ((FO_ABLATE_PP & 8) && kw_.kw != 1) pad = OOB;         // diagnostic: the A operand fetched for the centre column tap only (zero-fill DMAs otherwise)
      dma16(rin, sa + (i * 8 + wave) * 1024, (unsigned)(rowoffA[i] + stepoff) | pad);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      dma16(rwp, sb + (i * 8 + wave) * 1024, kw_.q < nt ? woffB[i] + kpos * 64 : OOB);
    kwalk_next(kw_, d, chunks32);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fpos = (quad ^ swz(l15)) * 16;                // (block bases are multiples of 16 rows: the swizzle sees lane & 15 only)
  const int aoff = (wm * TM * 16 + l15) * 64 + fpos, boff = BMB * 64 + (wn * TN * 16 + l15) * 64 + fpos;

  dma_tile();
  dma_tile();
  if (g1) {
    dma_tile();
    wait_vmcnt<2 * NP>();                              // tile 0 of this wave has landed
  } else {
    wait_vmcnt<NP>();
  }
  __builtin_amdgcn_s_barrier();
  if (g1) __builtin_amdgcn_s_barrier();                   // G1 runs one segment behind G0 from here on
  __builtin_amdgcn_sched_barrier(0);

  for (int p = 0; p < nt; ++p) {
    const unsigned char* As = lds + (p & 3) * SLOT + aoff;
    const unsigned char* Bs = lds + (p & 3) * SLOT + boff;
    bf16x8 fa[TM], fb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + ((FO_ABLATE_PP & 4) ? 0 : j) * 16 * 64);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + ((FO_ABLATE_PP & 4) ? 0 : i) * 16 * 64);
    if (!(FO_ABLATE_PP & 1)) dma_tile();                  // G0: tile p+2, G1: tile p+3
    if (g1) wait_vmcnt<2 * NP>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (!(FO_ABLATE_PP & 2) || (i == 0 && j == 0)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (!g1) wait_vmcnt<NP>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!g1) __builtin_amdgcn_s_barrier();                  // G0 waits out G1's last segment
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero-fill DMAs of tiles past the end
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: 32 rows of the wave tile at a time through a wave-local LDS patch, then 8 channels per lane (emit8)
  float* Cs = reinterpret_cast<float*>(lds) + wave * 32 * C_LD;
  const int flags = d.flags;
  constexpr int C8 = WCOLS / 8, RPP = 64 / C8;
  const int c8 = lane % C8, r0 = lane / C8;
  const int co = tile_n * BN + wn * WCOLS + c8 * 8;
  // FO_DEPTH2SPACE, cell form (store_c_tile): GEMM column = phase * Cpp + channel; a lane's 8 columns are 8 channels of ONE output pixel, (2 cy - 1 + ph / 2, 2 cx - 1 + ph % 2)
  const bool d2s = flags & FO_DEPTH2SPACE;
  const int cpp = d.Cout >> 2;
  const int ph = d2s ? co / cpp : 0;
  const int cc = d2s ? co - ph * cpp : co;      // channel of the OUTPUT tensor (bias, mask, add, store)
  const bool live = !d2s || cc < ((d.ophW + 7) & ~7);
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = ((flags & FO_BIAS) && cc + e < (d2s ? d.ophW : d.Cout)) ? a.bias[cc + e] : 0.f;
  const __bf16* mask = reinterpret_cast<const __bf16*>(a.mask);
  const __bf16* addp = reinterpret_cast<const __bf16*>(a.add);
#pragma unroll
  for (int i2 = 0; i2 < TM / 2; ++i2) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(ii * 16 + quad * 4 + r) * C_LD + j * 16 + l15] = acc[i2 * 2 + ii][j][r];
    __builtin_amdgcn_wave_barrier();
    const int mbase = tile_m * BMB + wm * TM * 16 + i2 * 32;
    size_t opix[32 / RPP];
    bool ok[32 / RPP];
#pragma unroll
    for (int pp = 0; pp < 32 / RPP; ++pp) {
      const int m = mbase + pp * RPP + r0;
      ok[pp] = (m < a.M) & live;
      const int mm = m < a.M ? m : 0;
      opix[pp] = (size_t)mm;
      if (d2s) {
        const int n = mm / a.HWm;
        const int rem = mm - n * a.HWm;
        const int y = rem / d.Wm;
        const int x = rem - y * d.Wm;
        int oy = 2 * y + (ph >> 1) - d.ophH, ox = 2 * x + (ph & 1) - d.ophH;
        ok[pp] = ok[pp] & ((unsigned)oy < (unsigned)d.Hout) & ((unsigned)ox < (unsigned)d.Wout);
        oy = min(max(oy, 0), d.Hout - 1);
        ox = min(max(ox, 0), d.Wout - 1);
        opix[pp] = ((size_t)n * d.Hout + oy) * d.Wout + ox;
      }
    }
    bf16x8 mk[32 / RPP], ad[32 / RPP];                    // all mask / add loads of the round in flight before the first use
    if (flags & FO_MASK) {
#pragma unroll
      for (int pp = 0; pp < 32 / RPP; ++pp) mk[pp] = load_mask8(a, opix[pp], live ? cc : 0);
    }
    if (flags & FO_ADD) {
#pragma unroll
      for (int pp = 0; pp < 32 / RPP; ++pp) ad[pp] = *reinterpret_cast<const bf16x8*>(addp + opix[pp] * d.ldAdd + (live ? cc : 0));
    }
#pragma unroll
    for (int pp = 0; pp < 32 / RPP; ++pp) {
      const int row = pp * RPP + r0;
      const float* crow = Cs + row * C_LD + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(crow);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(crow + 4);
      float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
      emit8(a, flags, v, mk[pp], ad[pp], opix[pp], cc, ok[pp]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ big tiles, 3x3: the A operand staged ONCE for the nine taps
// conv_bf16_pp16_kernel DMAs a tile's A rows again for every tap: nine 64-byte fetches of the same pixel per 32 input channels, and what
// bounds that kernel is the LDS-DMA stream, not the matrix pipe (tools/ablate_pp16.sh, 512 x 128 tile on the 64^2 latents: MFMAs alone
// 0.184 ms, DMAs alone 0.196 ms, both 0.252 ms; with the A fetches of the six off-centre taps turned into zero fills, still issued, 0.220).
// Here, for 3x3 (x3) pad-1 stride-1 layers whose tiles are whole image rows (W | BMB, BMB | H W: W is a power of two), a 32-channel chunk of
// the tile's pixels PLUS one image row above and below -- BMB + 2 W rows of 64 B, the "extended tile" -- is DMA'd once per (depth tap, chunk)
// and serves nine phases: tap (kh, kw) reads its fragments at row offset (kh - 1) W + (kw - 1).  kh shifts are multiples of 16 rows
// (block-uniform, the swizzle does not see them); kw shifts move a lane's row by +-1, so the swizzle is 2 ((row >> 2) & 1), conflict-free for
// ds_read_b128 at shifts -1, 0, +1 (exhaustive search over the instruction's lane groups).  In LDS every image row of the extended tile is
// preceded by one 64-byte row of zeros (and the last one followed by one): the pixel left of an image row's first / right of its last is that
// row, with no per-lane select (16-row DMA pieces never straddle an image row, so a piece lands whole at its padded address; a uniform
// shift of a 16-row block by a multiple of 64 B does not change its bank pattern).  The zero rows are written once per kernel.  Rows above /
// below the frame are zero-filled by the DMA (out-of-range offsets), as are depth taps outside the clip (kwalk_range skips those whole).
// LDS-DMA instructions per wave and nine phases: NPE + 9 NPB instead of 9 (NPA + NPB) -- 512 x 128: 15 instead of 45.
//
// Protocol: the B (filter) ring and the two wave groups one barrier apart are conv_bf16_pp16_kernel's.  The extended tiles alternate between
// two slots; group g + 1's pieces are issued one per wave and phase in phases 1 .. NPE of group g (its slot was last read in the last phase
// of group g - 1: retired two segments before the first issue), each behind that phase's B pieces.  vmcnt is in order, so the counted waits
// grow by the A pieces of the phases they let fly: G0 (end of phase j) vmcnt(NPB + a(j)), G1 (after its issue in phase j)
// vmcnt(2 NPB + a(j) + a(j - 1)), a(j) = 1 for 1 <= j <= NPE; the last piece (phase NPE <= 6) has landed by the waits of phase NPE + 2 <= 8.
//
// Tiles: a workgroup walks tiles blockIdx.x, + gridDim.x, ... -- the host launches one workgroup per tile, or (FACEOFF_BF16_PPH_PERSIST=1) one per
// CU.  In the persistent form the NEXT tile's first operands -- extended tile 0 and all four filter-ring slots, free between two tiles -- are
// DMA'd BEFORE this tile's epilogue issues its stores, and waited for with the stores still in flight (vmcnt(#stores): memory operations retire
// in order, every lane issues a known number of stores); phases 0..2 of a tile neither issue nor wait for filter tiles (they are there), so the
// stores drain beside the first three to four phases of the next tile.  Alone on the device that is worth 1-7 % of a launch (conv4_2 0.523 ->
// 0.518 ms, the 64^2 latent layer 0.190 -> 0.177); in the training step it LOSES 0.5 ms of 36.8 (tools/ab_env.sh, same device): a workgroup holds
// every register of its CU, and where each CU used to fall free between two tiles -- letting the side streams' workgroups (LPIPS heads, filter
// gradients, the ground-truth branch) in -- a persistent grid locks them out for the whole launch.  Hence off by default.
// Everything a tile changes is wave-uniform and lives in SGPRs -- a lane's own part of a DMA offset is the same for every piece (lpA / lpB,
// the range check sees the VGPR part only) -- and per-lane fragment addresses are re-made per tile from mbcnt: the next tile's setup runs
// while this tile's 128 accumulators are live, and one spill reload in the K loop is a vmcnt(0) in the middle of the counted waits.
//
// Where a launch goes (in-kernel s_memtime stamps, tools/stamp_pph.py, 2.0-2.1 GHz in-kernel clock): a phase takes ~1 300 clocks of which the two
// groups' 32 MFMAs each are 2 x 512; between two tiles 2-6 thousand clocks of setup + prologue issue and the epilogue -- 10-12 thousand through an fp32
// patch (round 4's first form), 17 thousand with 8-byte stores straight from the accumulators, 6.6-9 thousand as it is now -- against 47 thousand for the
// 36-phase K loop of the 64^2 latent layers and 187 thousand for conv4_2's.
__device__ __forceinline__ int swz2(int row) { return ((row >> 2) & 1) * 2; }
template <int N> __device__ __forceinline__ void wait_vmcnt_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef FO_STAMP_PPH   // diagnostic build only (tools/stamp_pph.sh): segment lengths of the K loop in core clocks, workgroup 0, waves 0 (G0) and 4 (G1)
__device__ unsigned long long fo_pph_stamps[64];
#define PPH_STAMP(var)                                                                  \
  if (stamping) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define PPH_STAMP(var)
#endif

typedef __bf16 pph_bf16x4 __attribute__((ext_vector_type(4)));
// One output pixel's 4 consecutive channels, straight from a lane's accumulator (the filter is the MFMA's ROW operand, so a lane holds channels
// 4 quad .. + 3 of pixel lane & 15; the accumulation STARTED from the bias): ReLU-backward mask -> + add -> ReLU -> ONE rounding to bf16 (or kept fp32).  Exactly one store.
__device__ __forceinline__ void pph_emit4(const ConvArgsH& a, int flags, const f32x4& c, const pph_bf16x4& mk, const pph_bf16x4& ad, size_t opix, int co, bool ok) {
  float v[4] = {c[0], c[1], c[2], c[3]};
  if (flags & FO_MASK) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
  }
  if (flags & FO_ADD) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += (float)ad[e];
  }
  if (flags & FO_OUT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (!ok) return;
  if (flags & FO_OUT_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + opix * a.d.ldOut + co) = f32x4{v[0], v[1], v[2], v[3]};
  else *reinterpret_cast<pph_bf16x4*>(reinterpret_cast<__bf16*>(a.out) + opix * a.d.ldOut + co) = pph_bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

template <int BMB, int BN, int WAVES_M, int WAVES_N, int NPE>
struct PphLds {
  static constexpr int APAD = (BMB / 16 + 3) * 64;        // zero rows: one per image row of the extended tile + 1 (narrowest frame: W = 16)
  static constexpr int ASLOT = NPE * 128 * 64 + APAD;     // an extended tile: NPE pieces of 16 rows per wave
  static constexpr int BSLOT = BN * 64;
  static constexpr int OFF_B = 2 * ASLOT, OFF_D = OFF_B + 4 * BSLOT;      // OFF_D: where pieces past the extended tile's end go
  static constexpr int OFF_BIAS = OFF_D + 1024, MAXC = 1024;              // the layer's bias (zeros without FO_BIAS): the accumulators start from it
  static constexpr int PPITCH = 144;                                      // the epilogue's bf16 patch: 16 pixels x 64 channels per wave, rows of 128 + 16 B
  static constexpr int OFF_P = OFF_BIAS + MAXC * 4;
  static constexpr int BYTES = OFF_P + 8 * 16 * PPITCH;
  static_assert(BYTES <= 160 * 1024, "LDS");
};

template <int BMB, int BN, int WAVES_M, int WAVES_N, int NPE>
__global__ __launch_bounds__(512, 1) void conv_bf16_pph_kernel(const ConvArgsH a) {
  static_assert(WAVES_M * WAVES_N == 8 && (BMB == 256 || BMB == 512) && (BN == 256 || BN == 128) && NPE >= 3 && NPE <= 6, "8 waves");
  using L = PphLds<BMB, BN, WAVES_M, WAVES_N, NPE>;
  constexpr int TM = BMB / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int WCOLS = TN * 16;
  static_assert(WCOLS == 64, "the epilogue stores 64-column wave tiles");
  constexpr int ASLOT = L::ASLOT, BSLOT = L::BSLOT;
  constexpr int NPB = BN / 128;
  constexpr int OFF_B = L::OFF_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const bool g1 = __builtin_amdgcn_readfirstlane(wave >> 2) != 0;
  const int ntiles = a.tilesM * a.tilesN;
  const int chunks32 = a.cinChunks;
  const int W = d.Wm;
  const int wsh = __builtin_ctz((unsigned)W);
  const int flags = d.flags;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;
  {   // the zero rows: before image row s of the extended tile, s = 0 .. BMB / W + 2, at byte s (W + 1) 64 of either slot
    const int nz = (BMB >> wsh) + 3;
    for (int t = tid; t < 2 * nz * 16; t += 512) {
      const int slot = t / (nz * 16), r = t % (nz * 16);
      reinterpret_cast<unsigned*>(lds + slot * ASLOT + (r >> 4) * (W + 1) * 64)[r & 15] = 0u;
    }
  }
  for (int c = tid; c < d.Cout; c += 512) reinterpret_cast<float*>(lds + L::OFF_BIAS)[c] = (flags & FO_BIAS) ? a.bias[c] : 0.f;      // (host: Cout <= MAXC)
  unsigned lpA, lpB;                                      // a lane's part of every A / B piece's offset (piece bases are multiples of 16 rows: the swizzle sees drow only)
  {
    const int lane = tid & 63, drow = lane >> 2, dpos = lane & 3;
    lpA = (unsigned)(drow * d.ldIn * 2 + (dpos ^ swz2(drow)) * 16);
    lpB = (unsigned)(drow * a.Ktot * 2 + (dpos ^ swz2(drow)) * 16);
  }

  // ---- per tile: BMB consecutive pixels of ONE frame starting at an image-row boundary, BN filter rows
  int tile_m, tile_n, kd0, ng, nt;
  int sA[NPE];                                            // byte offset of piece k's first row at depth tap 0, chunk 0 (negative for frame 0 when padD = 1: with goff it is not)
  unsigned validE;                                        // bit k: piece k's rows are pixels of the frame (else: zero fill)
  int sB[NPB];
  // the two DMA streams, each with its own running position
  int bq, bj, bchunk, bkd;                                // next B tile: phase bq = 9 * group + bj, group = (bkd, bchunk)
  int ag, achunk, akd;                                    // next extended tile: group ag = (akd, achunk)
  auto setup = [&](int vb) {
    const int logical = fo_xcd_remap(vb, ntiles);         // (a persistent grid is a multiple of 8 workgroups: vb and blockIdx.x sit on the same XCD)
    tile_n = logical % a.tilesN;
    tile_m = logical / a.tilesN;
    int kd1;
    kwalk_range(a, tile_m * BMB, kd0, kd1);
    ng = (kd1 - kd0) * chunks32;                          // extended tiles (groups of nine phases) of this row tile
    nt = ng * 9;
    const int m0 = tile_m * BMB;
    const int fn = m0 / a.HWm;
    const int y0 = (m0 - fn * a.HWm) >> wsh;
    validE = 0;
#pragma unroll
    for (int k = 0; k < NPE; ++k) {
      const int e = (k * 8 + wave) * 16;                  // extended row of the piece's first row: pixel m0 - W + e
      const int yy = y0 - 1 + (e >> wsh), xx = e & (W - 1);
      const bool ok = e < BMB + 2 * W && (unsigned)yy < (unsigned)d.Hin;
      sA[k] = __builtin_amdgcn_readfirstlane(((((fn - d.padD) * d.Hin + yy) * d.Win + xx) * d.ldIn) * 2);
      validE |= (ok ? 1u : 0u) << k;
    }
    validE = __builtin_amdgcn_readfirstlane(validE);
#pragma unroll
    for (int i = 0; i < NPB; ++i) sB[i] = __builtin_amdgcn_readfirstlane((tile_n * BN + (i * 8 + wave) * 16) * a.Ktot * 2);
    bq = 0; bj = 0; bchunk = 0; bkd = kd0;
    ag = 0; achunk = 0; akd = kd0;
  };
  auto dma_b = [&]() {
    const int kpos = (bkd * 9 + bj) * chunks32 + bchunk;  // 32-element position inside a filter row: tap * chunks + chunk
    lds_byte* const sb = lds3 + OFF_B + (bq & 3) * BSLOT;
    const unsigned vo = lpB | (bq < nt ? 0u : OOB);        // (tiles past the end are zero fills; an OR with an SGPR, not a select between two VGPRs)
#pragma unroll
    for (int i = 0; i < NPB; ++i) dma16s(rwp, sb + (i * 8 + wave) * 1024, vo, (unsigned)(sB[i] + kpos * 64));
    ++bq;
    if (++bj == 9) { bj = 0; if (++bchunk == chunks32) { bchunk = 0; ++bkd; } }
  };
  auto dma_a_piece = [&](int k) {
    const int goff = ((akd * d.Hin * d.Win) * d.ldIn + achunk * 32) * 2;
    const bool ok = ag < ng && ((validE >> k) & 1u);
    const int e = (k * 8 + wave) * 16;
    const int dst = e < BMB + 2 * W ? (ag & 1) * ASLOT + e * 64 + ((e >> wsh) + 1) * 64 : L::OFF_D;      // (every wave issues every piece: the waits count them)
    dma16s(rin, lds3 + dst, lpA | (ok ? 0u : OOB), (unsigned)(sA[k] + goff));
  };
  auto a_next = [&]() { ++ag; if (++achunk == chunks32) { achunk = 0; ++akd; } };
  // a tile's first operands: extended tile 0 whole and the four filter tiles of the ring (every slot is free between two tiles)
  auto prologue = [&]() {
#pragma unroll
    for (int k = 0; k < NPE; ++k) dma_a_piece(k);
    a_next();
    dma_b(); dma_b(); dma_b(); dma_b();
  };
  // zero rows in front of a wave's 16-row block i at tap row kh: (its image row in the extended tile) + 1
  int apad[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) apad[i] = __builtin_amdgcn_readfirstlane((((wm * TM + i) * 16 >> wsh) + 1) * 64);
  const int W65 = (W + 1) * 64;                           // one image row down, in LDS

#ifdef FO_STAMP_PPH
  const bool stamping = blockIdx.x == 0 && (wave & 3) == 0;
  unsigned long long tS = 0, tA = 0, tM = 0, tK0 = 0, rK0 = 0, sumR = 0, sumM = 0, sumW = 0, nph = 0;
  unsigned long long sumF[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tE0 = 0, tE1 = 0, tE2 = 0, tE3 = 0, sumSet = 0, sumEpi = 0, sumWt = 0, ntl = 0;
  if (stamping) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tK0), "=s"(rK0)::"memory");
#endif
  int vb = blockIdx.x;
  setup(vb);
  prologue();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (;;) {
    // fragment addressing, re-made per tile from the lane id (mbcnt: not a live register).  B: row (lane & 15) of a 16-row block, chunk quad.
    // A: extended row W + (wave block) + i * 16 + l15 + (kh - 1) W + (kw - 1), + its zero rows
    int ol = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ol));                          // (opaque: or everything derived from it is hoisted out of the tile loop)
    const int l15 = ol & 15, quad = ol >> 4;
    const int boff = OFF_B + (wn * TN * 16 + l15) * 64 + (quad ^ swz2(l15)) * 16;
    int aoff[3];                                          // per kw: byte offset of this lane's row in block 0 of the wave at kh = 0, without zero rows
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int r = l15 + kw - 1;                         // -1 .. 16 (block bases and kh shifts are multiples of 16: the swizzle sees r only)
      aoff[kw] = (wm * TM * 16 + r) * 64 + (quad ^ swz2(r + 16)) * 16;
    }
    f32x4 acc[TM][TN];
    if (vb != (int)blockIdx.x) {                          // (first tile: the bias is not visible before the barrier below)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + L::OFF_BIAS + (tile_n * BN + wn * WCOLS + j * 16 + quad * 4) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = b4;
      }
    }
    __builtin_amdgcn_s_barrier();                         // every wave's pieces of A(0), B(0..3) have landed (first tile: and the zero rows are written)
    if (vb == (int)blockIdx.x) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + L::OFF_BIAS + (tile_n * BN + wn * WCOLS + j * 16 + quad * 4) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = b4;
      }
    }
    if (g1) __builtin_amdgcn_s_barrier();                 // G1 runs one segment behind G0 from here on
    __builtin_amdgcn_sched_barrier(0);

    for (int g = 0; g < ng; ++g) {
      const unsigned char* Ag = lds + (g & 1) * ASLOT;
      const bool first = g == 0;                          // B(0..3) are there: phases 0..2 neither issue them again nor wait (what they would wait
                                                          // for has landed, and the previous tile's stores may still be in flight in front of the loop's DMAs)
      // (opaque per iteration: otherwise the nine phases' per-lane addresses are hoisted out of the g loop and the kernel spills)
      int ao0 = aoff[0], ao1 = aoff[1], ao2 = aoff[2];
      asm volatile("" : "+v"(ao0), "+v"(ao1), "+v"(ao2));
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int kh = j / 3, kw = j % 3;
        const int p = g * 9 + j;
        PPH_STAMP(tS);
        int bslot = (p & 3) * BSLOT;                      // (opaque: one address add per phase, not four rotating base registers)
        asm volatile("" : "+s"(bslot));
        const unsigned char* Bs = lds + bslot + boff;
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) fb[jn] = *reinterpret_cast<const bf16x8*>(Bs + jn * 16 * 64);
        const int abase = (kw == 0 ? ao0 : kw == 1 ? ao1 : ao2) + kh * W65;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(Ag + abase + i * 16 * 64 + apad[i]);
        // G0 issues B tile p + 2, G1 p + 3 (tiles 0..3 of a row tile come with its prologue)
        if (!(FO_ABLATE_PP & 1) && !(first && j < (g1 ? 1 : 2))) dma_b();
        if (j >= 1 && j <= NPE) {
          if (!(FO_ABLATE_PP & 1)) dma_a_piece(j - 1);    // extended tile g + 1, one piece per phase
          if (j == NPE) a_next();
        }
        if (g1 && !(first && j <= 2)) {
          // in flight at most: this phase's and the previous phase's pieces
          if (j == 0) wait_vmcnt_n<2 * NPB>();
          else if (j == 1) wait_vmcnt_n<2 * NPB + 1>();
          else if (j <= NPE) wait_vmcnt_n<2 * NPB + 2>();
          else if (j == NPE + 1) wait_vmcnt_n<2 * NPB + 1>();
          else wait_vmcnt_n<2 * NPB>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PPH_STAMP(tA);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            if (!(FO_ABLATE_PP & 2) || (i == 0 && jn == 0)) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);      // rows = channels, columns = pixels
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        PPH_STAMP(tM);
        if (!g1 && !(first && j <= 2)) {
          if (j >= 1 && j <= NPE) wait_vmcnt_n<NPB + 1>();
          else wait_vmcnt_n<NPB>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef FO_STAMP_PPH
        if (stamping) {
          unsigned long long tE;
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tE)::"memory");
          sumR += tA - tS; sumM += tM - tA; sumW += tE - tM; ++nph;
          if (first) sumF[j] += tE - tS;
        }
#endif
      }
    }
    if (!g1) __builtin_amdgcn_s_barrier();                // G0 waits out G1's last segment
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill DMAs of tiles past the end
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    PPH_STAMP(tE0);
    const int ctm = tile_m, ctn = tile_n;
    // the ReLU-backward mask of the whole tile (bf16 outputs without FO_ADD: 2 TM fragments of 16 B per lane, the layout the stores below have):
    // issued here, they fly beside the next tile's setup and prologue issue (2-6 thousand clocks) and are older than the prologue's DMAs
    const bool patch_epi = !(flags & (FO_ADD | FO_OUT_F32));
    bf16x8 mkA[TM][2] = {};
    if (patch_epi && (flags & FO_MASK)) {
      int ml = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ml));
      const size_t mp0 = (size_t)ctm * BMB + wm * TM * 16 + (ml >> 3);
      const int mco = ctn * BN + wn * WCOLS + (ml & 7) * 8;
      if (a.maskBits) {                                    // bit plane: one byte per fragment (all loads first, then the expansions)
        unsigned mb8[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) mb8[i][pp] = a.maskBits[(mp0 + i * 16 + pp * 8) * a.ldBits + (mco >> 3)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) mkA[i][pp] = bits_to_mask8(mb8[i][pp]);
      } else {
        const __bf16* mp = reinterpret_cast<const __bf16*>(a.mask) + mp0 * d.ldMask + mco;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) mkA[i][pp] = *reinterpret_cast<const bf16x8*>(mp + (size_t)(i * 16 + pp * 8) * d.ldMask);
      }
    }
    vb += gridDim.x;
    const bool more = vb < ntiles;
    if (more) {
      setup(vb);
      prologue();
    }
    __builtin_amdgcn_sched_barrier(0);
    PPH_STAMP(tE1);

    // ---- epilogue.  A lane's accumulator is 4 consecutive channels of one pixel per 16 x 16 block (the filter is the MFMA's row operand).
    // bf16 outputs without FO_ADD (every LPIPS layer, most of the VQ-VAE's): ReLU, ONE rounding, then a 16-pixel block row goes through the wave's
    // own bf16 patch in LDS (4 ds_write_b64, 2 ds_read_b128 per lane; pitch 144 B: conflict-free writes) so that a store instruction writes whole
    // 128-byte lines, 16 B per lane -- what the store path is priced by is lines per instruction: 8-byte stores straight from the accumulators
    // (16 part-lines per instruction) took 17 thousand clocks per tile, an fp32 patch (four times the LDS traffic, a write -> read -> convert chain
    // per block) 10-12 thousand (tools/stamp_pph.py).  The ReLU-backward mask is applied to the rounded values (a select: exact), its 16-byte
    // fragments for block row i + 1 loaded before block row i is stored.  FO_ADD / fp32 outputs (a few VQ-VAE layers) store from the accumulators.
    if (FO_ABLATE_PP & 16) {                              // diagnostic: no epilogue (one store per lane keeps the accumulators alive)
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (t == 12345.678f) reinterpret_cast<float*>(a.out)[tid] = t;
    } else if (patch_epi) {
      int el = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(el));
      unsigned char* const P = lds + L::OFF_P + wave * 16 * L::PPITCH;
      const int woff = (el & 15) * L::PPITCH + (el >> 4) * 8;               // + j * 32: pixel el & 15, channels j * 16 + 4 (el >> 4) ..
      const int roff = (el >> 3) * L::PPITCH + (el & 7) * 16;               // + pp * 8 rows: pixel pp * 8 + (el >> 3), channels 8 (el & 7) ..
      const int co = ctn * BN + wn * WCOLS + (el & 7) * 8;
      const size_t mb = (size_t)ctm * BMB + wm * TM * 16 + (el >> 3);      // + i * 16 + pp * 8
      __bf16* const outp = reinterpret_cast<__bf16*>(a.out);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x4 v = acc[i][j];
          if (flags & FO_OUT_RELU) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          *reinterpret_cast<pph_bf16x4*>(P + woff + j * 32) = pph_bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          bf16x8 o = *reinterpret_cast<const bf16x8*>(P + roff + pp * 8 * L::PPITCH);
          if (flags & FO_MASK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (float)mkA[i][pp][e] > 0.f ? o[e] : (__bf16)0.f;
          }
          const size_t m = mb + i * 16 + pp * 8;
          // (every row and column of a tile is real -- host: BMB | M, BN | Cout -- so this is exactly one store per lane: the count below depends on it)
          if (!(FO_ABLATE_PP & 32)) *reinterpret_cast<bf16x8*>(outp + ((FO_ABLATE_PP & 64) ? (m & 511) : m) * d.ldOut + co) = o;
          if (a.outBits) a.outBits[m * a.ldBits + (co >> 3)] = (unsigned char)pos_bits8(o);          // (+ one byte store per lane: counted below)
        }
      }
    } else {
      int el = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(el));
      const int cb = ctn * BN + wn * WCOLS + (el >> 4) * 4;                 // + j * 16
      const size_t mb = (size_t)ctm * BMB + wm * TM * 16 + (el & 15);      // + i * 16
      const __bf16* mask = reinterpret_cast<const __bf16*>(a.mask);
      const __bf16* addp = reinterpret_cast<const __bf16*>(a.add);
      pph_bf16x4 mk[2][TN] = {}, ad[2][TN] = {};           // (initialised: undefined values would be carried around the tile loop in registers)
      auto fetch = [&](int i, int buf) {
        if (flags & FO_MASK) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (a.maskBits) {
              const unsigned nb = (a.maskBits[(mb + i * 16) * a.ldBits + ((cb + j * 16) >> 3)] >> (((cb >> 2) & 1) * 4)) & 15u;
              mk[buf][j] = pph_bf16x4{(__bf16)(float)(nb & 1u), (__bf16)(float)((nb >> 1) & 1u), (__bf16)(float)((nb >> 2) & 1u), (__bf16)(float)((nb >> 3) & 1u)};
            } else {
              mk[buf][j] = *reinterpret_cast<const pph_bf16x4*>(mask + (mb + i * 16) * d.ldMask + cb + j * 16);
            }
          }
        }
        if (flags & FO_ADD) {
#pragma unroll
          for (int j = 0; j < TN; ++j) ad[buf][j] = *reinterpret_cast<const pph_bf16x4*>(addp + (mb + i * 16) * d.ldAdd + cb + j * 16);
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i + 1 < TM) fetch(i + 1, (i + 1) & 1);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const size_t m = mb + i * 16;
          pph_emit4(a, flags, acc[i][j], mk[i & 1][j], ad[i & 1][j], (FO_ABLATE_PP & 64) ? (m & 511) : m, cb + j * 16, (FO_ABLATE_PP & 32) ? (flags & 0x4000) != 0 : true);
        }
      }
    }
    PPH_STAMP(tE2);
    if (!more) break;
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's prologue was issued before this tile's stores (2 TM per lane through the patch, TM TN from the accumulators) and its mask / add
    // loads (as many each), and memory operations retire in order: once at most that many are outstanding the prologue has landed (vmcnt holds
    // 6 bits: capped, which only waits for a few stores more)
    if (patch_epi) {
      if (a.outBits) wait_vmcnt_n<4 * TM>();              // (+ the bit plane's byte stores)
      else wait_vmcnt_n<2 * TM>();                        // (the mask fragments were issued before the prologue)
    } else {
      constexpr int NST = TM * TN;
      if ((flags & FO_MASK) && (flags & FO_ADD)) wait_vmcnt_n<(3 * NST < 63 ? 3 * NST : 63)>();
      else if (flags & (FO_MASK | FO_ADD)) wait_vmcnt_n<(2 * NST < 63 ? 2 * NST : 63)>();
      else wait_vmcnt_n<NST>();
    }
#ifdef FO_STAMP_PPH
    PPH_STAMP(tE3);
    if (stamping) { sumSet += tE1 - tE0; sumEpi += tE2 - tE1; sumWt += tE3 - tE2; ++ntl; }
#endif
  }
#ifdef FO_STAMP_PPH
  if (stamping && (threadIdx.x & 63) == 0) {
    unsigned long long tK1, rK1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tK1), "=s"(rK1)::"memory");
    unsigned long long* o = fo_pph_stamps + (wave >> 2) * 32;
    o[0] = sumR; o[1] = sumM; o[2] = sumW; o[3] = nph; o[4] = tK1 - tK0; o[5] = rK1 - rK0;
    o[6] = sumSet; o[7] = sumEpi; o[8] = sumWt; o[9] = ntl;
    for (int q = 0; q < 9; ++q) o[10 + q] = sumF[q];
  }
#endif
}

// Cross-lane moves on the VECTOR ALU (no LDS round trip: __shfl_xor compiles to ds_bpermute_b32, an LDS-pipeline instruction with its latency --
// the pooled epilogue of the halo-tile kernel issued ~48 of them per tile, each feeding the next: in-kernel stamps, round 6, put that epilogue at
// 64 % of a tile).  lane ^ 1 is a DPP quad permutation; the OR over the four 16-lane rows (lanes l, l + 16, l + 32, l + 48) is two of gfx950's row
// swaps: v_permlane16_swap exchanges the odd rows of one operand with the even rows of the other, v_permlane32_swap the upper half with the lower.
__device__ __forceinline__ int lane_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ float lane_xor1(float v) { return __int_as_float(lane_xor1(__float_as_int(v))); }
__device__ __forceinline__ unsigned or_rows(unsigned v) {
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = a[0] | a[1];
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return b[0] | b[1];
}

// ------------------------------------------------------------------------------------------------ 64 input channels, 3x3: halo tiles
// The short-K layers (VGG conv1_2 64 -> 64 at full resolution, conv2_1 64 -> 128, reference models/lpips.py:118-134, and their data
// gradients where the gradient has 64 channels; K = 576) spend a tiled implicit GEMM mostly outside its K loop, and the tile's A operand
// is re-fetched through L1 once per tap.  Here a workgroup of four waves walks output tiles of 4 rows x 32 pixels (persistent grid, TWO
// workgroups per CU: one's epilogue stores and patch wait run beside the other's MFMAs -- with one workgroup per CU the store phase, 0.39 ms
// of a 1.1 ms launch, overlapped nothing):
//   * the tile's INPUT PATCH (6 rows x 34 pixels x 64 channels, zero halo) is DMA'd into LDS once -- as two 32-channel planes of 64-byte
//     pixels, the 256-row kernel's layout and swizzle (conflict-free ds_read_b128), row pitch 48 pixels so that a tap's shift keeps the
//     swizzle a per-lane constant -- and serves all nine taps; the next tile's patch flies during this tile's MFMAs (two LDS stages);
//   * the FILTER never touches LDS: a wave owns 32 output channels of a 64-channel half and keeps their 9 x 64 x 32 fragment set in
//     144 registers for the whole kernel (v_mfma_f32_16x16x32_bf16 with the filter as the ROW operand, 16 pixels as columns);
//   * so a K-slice costs a wave 4 LDS reads for 8 MFMAs and nothing else; a lane's accumulator is 4 consecutive channels of one pixel
//     (bias / ReLU mask / ReLU / bf16 rounding per lane, 8-byte stores).
// Work item = (64-channel half of Cout, tile); a workgroup stays on one half.
struct HaloArgs {
  const __bf16* in;
  const __bf16* wp;          // [Cout][9][64]
  const float* bias;
  const __bf16* mask;
  __bf16* out;
  __bf16* pooled;           // optional: the 2x2 max-pool of `out` ([N][H/2][W/2][ldPooled]) written from the same accumulators
  unsigned char* pidx;      // optional, with pooled: its arg-max codes, 2 bits per channel ([N][H/2][W/2][Cout/4] bytes: fo_maxpool2_fwd_idx_bf16)
  const unsigned char* maskBits;   // optional: the FO_MASK tensor as a bit plane [pixel][Cout/8] (ConvArgsH)
  unsigned char* outBits;          // optional: the plane of `out`
  unsigned char* pooledBits;       // optional, with pooled: the plane of the pooled output [N][H/2][W/2][Cout/8]
  int N, H, W, Cout, halves, tilesX, tilesY, ntiles;
  int ldIn, ldOut, ldMask, ldPooled, flags;
  unsigned inBytes;
};

#ifdef FO_STAMP_H64   // diagnostic build only (tools/stamp_h64.sh): where a tile of the halo-tile kernel goes, in core clocks (workgroup 0, wave 0)
__device__ unsigned long long fo_h64_stamps[16];
#define H64_STAMP(var)                                                                  \
  if (stamping) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define H64_STAMP(var)
#endif

// MASKT / MASKB: FO_MASK from the bf16 tensor / from its bit plane; POOL: the pooled output (its codes and plane where given); OBITS: the result's plane.
// (compile-time: with all of it behind run-time tests the kernel spills -- 144 registers hold the filter)
// LINES (round 5): a 16-pixel block's result goes through a wave-private 1.25 KB patch in LDS (2 ds_write_b64, 1 ds_read_b128 per lane) so that a
// store is 16 B per lane and a pixel's 64 bytes (the wave's 32 channels) leave in ONE piece -- 4 store instructions of 16 half-lines per tile and
// wave instead of 8 of 16 quarter-lines (the store path is priced by line pieces per instruction: the extended-tile kernel's epilogue, above).
// Patch swizzle: chunk c of patch pixel p sits at position c ^ H64_SWZ(p), on the DMA source side and in the fragment reads.  Round 5: the extended-tile
// kernel's 2 ((p >> 2) & 1), conflict-free for ds_read_b128 at the three tap shifts kw = 0, 1, 2 (exhaustive search over the instruction's lane groups),
// instead of {0, 2, 3, 1}[(p >> 2) & 3], which is conflict-free at kw = 0 only (SQ_LDS_BANK_CONFLICT 6-7.7 % of the wave cycles, profiles/r04_c3_pmc.md).
#ifndef H64_SWZ
#define H64_SWZ(p) swz2(p)
#endif
#ifndef FO_H64_PRIO
#define FO_H64_PRIO 0
#endif
#ifndef FO_H64_EPI_DEAD
#define FO_H64_EPI_DEAD 1
#endif
#ifndef FO_H64_STAGGER
#define FO_H64_STAGGER 0
#endif
template <bool MASKT, bool MASKB, bool POOL, bool OBITS, bool LINES>
__global__ __launch_bounds__(256, 2) void conv_halo64_bf16_kernel(const HaloArgs a) {
  constexpr int PITCH = 48, ROWS = 6, PLANE = ROWS * PITCH * 64, STAGE = 2 * PLANE;     // bytes
  constexpr int EPITCH = 80, EPATCH = 16 * EPITCH;         // the epilogue patch: 16 pixels x (64 + 16) B per wave, behind the two stages
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (uniform: the DMA destinations and scalar offsets are derived from it)
  const int l15 = lane & 15, quad = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;                 // wave = rows 2 wm, 2 wm + 1 of the tile x output channels 32 wn .. + 31 of the half
  const int half = blockIdx.x % a.halves;
  const int wgInHalf = blockIdx.x / a.halves, wgsPerHalf = gridDim.x / a.halves;
  // (the descriptor starts ONE PIXEL before the tensor, so that the patch's left halo column is a non-negative offset)
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in) - a.ldIn, 0, a.inBytes + a.ldIn * 2, 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;

  // ---- the filter fragments of this wave (wave = tile rows 2 wm, 2 wm + 1 x output channels 32 wn .. + 31 of the half):
  // row (output channel) l15 of block j, k = tap * 64 + slice * 32 + quad * 8
  bf16x8 wf[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[t][sl][j] = *reinterpret_cast<const bf16x8*>(a.wp + (size_t)(half * 64 + wn * 32 + j * 16 + l15) * 576 + t * 64 + sl * 32 + quad * 8);

  // ---- DMA roles: instruction id = (plane, patch row, 16-pixel group) = 36 per tile, dealt over the 4 waves; lane = (pixel l / 4, chunk
  // position l % 4).  A lane's share of the source address is ONE constant register (pixel and swizzled chunk), the rest is scalar.
  const int dpix = lane >> 2, dpos = lane & 3;
  const unsigned dlane = (unsigned)(dpix * a.ldIn * 2 + ((dpos ^ H64_SWZ(dpix)) * 16));
  // A workgroup walks a contiguous run of tiles DOWN the columns of 32 pixels (tile = (n, tx, ty), ty fastest): tile s + 1's patch rows 0, 1 are
  // tile s's rows 4, 5 -- copied LDS -> LDS while tile s computes (12 KB, three 16-byte moves per thread) instead of fetched again.  Only the
  // four new rows are DMA'd: 24 instructions per tile instead of 36, and 1.07 x the input read from memory instead of 1.59 x (every patch row
  // fetched per tile; a column walk WITHOUT the copy measured 1.44 x: an XCD's L2 turns over in 1.5 tile times).  The first tile of a run and of
  // a column fetch all six rows.
  auto tile_of = [&](int tile, int& n, int& ty, int& tx) {
    ty = tile % a.tilesY;
    const int r1 = tile / a.tilesY;
    tx = r1 % a.tilesX;
    n = r1 / a.tilesX;
  };
  // DMA roles (round 6): a wave fetches whole patch ROWS -- row 2 + wave of a tile that continues its column (rows 0, 1 are copied), rows wave and
  // wave + 4 of a tile that starts one -- so a piece's (plane, 16-pixel group) is a compile-time constant: per row ONE scalar base offset and three lane
  // predicates, per piece one scalar add.  (Pieces dealt round-robin over the waves made every piece's row / group / plane a run-time scalar: 76 SGPRs
  // spilled into VGPR lanes and ~20 scalar instructions + 4 v_readlane per piece -- in-kernel stamps put the issue phase at a quarter of a tile.)
  const unsigned gB = (unsigned)(16 * a.ldIn * 2);        // bytes between the 16-pixel groups of a row
  auto dma_row = [&](int stage, int r, int n, int iy, int x0, bool live) {
    const bool rowok = live & ((unsigned)iy < (unsigned)a.H);
    const unsigned srow = rowok ? (unsigned)((((size_t)n * a.H + iy) * a.W + x0) * a.ldIn * 2) : 0u;
    lds_byte* const drow = lds3 + stage * STAGE + r * (PITCH * 64);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int ix = x0 - 1 + g * 16 + dpix;               // image column of this lane's pixel
      // (the third 16-pixel group holds the patch's last two columns: its other 14 lanes fetch nothing)
      const bool ok = rowok & ((unsigned)ix < (unsigned)a.W) & (g * 16 + dpix < 34);
      const unsigned vo = ok ? dlane : OOB;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) dma16s(rin, drow + pl * PLANE + g * (16 * 64), vo, srow + g * gB + pl * 64);
    }
  };
  auto dma_tile = [&](int tile, int stage, bool reuse, int n, int ty, int tx) {      // (n, ty, tx) = tile_of(tile)
    if (FO_ABLATE_H & 8) return;
    const bool live = tile < a.ntiles;
    const int y0 = ty * 4 - 1, x0 = tx * 32;               // (x0: patch column 0 is image column x0 - 1 = descriptor pixel x0)
    if (reuse) {
      dma_row(stage, 2 + wave, n, y0 + 2 + wave, x0, live);
    } else {
      dma_row(stage, wave, n, y0 + wave, x0, live);
      if (wave < 2) dma_row(stage, 4 + wave, n, y0 + 4 + wave, x0, live);
    }
  };

  // ---- fragment addressing: pixel column l15 of a 16-pixel block at tap shift kw: patch pixel x16 + l15 + kw, chunk quad ^ swz
  int cq[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) cq[kw] = (l15 + kw) * 64 + ((quad ^ H64_SWZ(l15 + kw)) * 16);


  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = (a.flags & FO_BIAS) ? a.bias[half * 64 + wn * 32 + j * 16 + quad * 4 + r] : 0.f;

  const int per = (a.ntiles + wgsPerHalf - 1) / wgsPerHalf;
  const int tile_end = min(a.ntiles, (wgInHalf + 1) * per);
  int tile = wgInHalf * per;
  if (tile >= tile_end) return;
#if FO_H64_STAGGER
  // (diagnostic variants) the CU's two workgroups run the same program from the same start: hold one of them back by about half a tile so that
  // one's epilogue / issue phases fall beside the other's MFMA loop instead of beside its epilogue
  if (FO_H64_STAGGER == 1 ? (blockIdx.x & 1) : FO_H64_STAGGER == 2 ? (blockIdx.x >= gridDim.x / 2) : ((blockIdx.x >> 3) & 1)) {
    __builtin_amdgcn_s_sleep(100);
  }
#endif
  // this tile's coordinates, stepped from tile to tile (tile_of costs two integer divisions, and the loop needed it three times per tile)
  int n, ty, tx;
  tile_of(tile, n, ty, tx);
  dma_tile(tile, 0, false, n, ty, tx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef FO_STAMP_H64
  const bool stamping = blockIdx.x == 0 && wave == 0;
  unsigned long long h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, sIssue = 0, sMfma = 0, sEpi = 0, sWait = 0, nT = 0, hK0 = 0, hR0 = 0;
  if (stamping) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(hK0), "=s"(hR0)::"memory");
#endif
  for (int it = 0; tile < tile_end; ++tile, ++it) {
    const int st = it & 1;
    const bool more = tile + 1 < tile_end;
    int n1 = n, ty1 = ty + 1, tx1 = tx;                     // the next tile: down the column, then the next column, then the next image
    if (ty1 == a.tilesY) { ty1 = 0; if (++tx1 == a.tilesX) { tx1 = 0; ++n1; } }
    const bool reuse = more && ty1 != 0;                    // the next tile is the one below this one
    H64_STAMP(h0);
    // the next tile's rows 0, 1 = this patch's rows 4, 5, both planes (whole 48-pixel rows: same layout, same swizzle).  The reads are issued FIRST and
    // land under the DMA issue's scalar work; the writes follow it (read -> wait -> write per plane ahead of everything else was two exposed LDS round trips)
    constexpr int NQ = 2 * PITCH * 64 / 16;                 // 384 sixteen-byte pieces per plane
    u32x4 t3[2][2];
    if (reuse) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (tid + q * 256 < NQ) t3[pl][q] = *reinterpret_cast<const u32x4*>(lds + st * STAGE + pl * PLANE + 4 * PITCH * 64 + (tid + q * 256) * 16);
    }
    if (more) dma_tile(tile + 1, st ^ 1, reuse, n1, ty1, tx1);          // next tile's patch: lands during this tile's MFMAs
    if (reuse) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (tid + q * 256 < NQ) *reinterpret_cast<u32x4*>(lds + (st ^ 1) * STAGE + pl * PLANE + (tid + q * 256) * 16) = t3[pl][q];
    }
    // the tile's ReLU mask from its bit plane: one dword per pixel block holds the nibbles of both channel blocks; loaded HERE, the four loads land
    // under the MFMAs (as 8-byte fragments of the bf16 tensor, fetched block by block in the epilogue, the mask made conv1_2's data gradient
    // 1.26 ms against 0.90 for the forward)
    unsigned mw[4] = {0u, 0u, 0u, 0u};
    if (MASKB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t pix = ((size_t)n * a.H + ty * 4 + 2 * wm + (i >> 1)) * a.W + tx * 32 + (i & 1) * 16 + l15;
        mw[i] = *reinterpret_cast<const unsigned*>(a.maskBits + pix * (a.Cout / 8) + half * 8 + wn * 4);
      }
    }
    H64_STAMP(h1);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const base = lds + st * STAGE;
    // 36 half-steps h = (tap, K-slice, pair of pixel blocks): two fragment reads, four MFMAs.  The reads run AHEAD half-steps in front of their
    // MFMAs (a ring of AHEAD + 1 fragment pairs): in-kernel stamps (tools/stamp_h64.py) put the loop at 5 900-6 000 clocks per tile against 2 304 of
    // MFMA issue -- each half-step waited out its own LDS round trip whenever the SIMD's other wave (the CU's other workgroup) was not in its
    // MFMA loop too.
#ifndef FO_H64_AHEAD
#define FO_H64_AHEAD 2
#endif
    constexpr int AHEAD = FO_H64_AHEAD, NH = (FO_ABLATE_H & 16) ? 0 : 36;
    bf16x8 xf[AHEAD + 1][2];
    auto frag = [&](int h, bf16x8 (&x)[2]) {
      const int tap = h >> 2, sl = (h >> 1) & 1, pr = h & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int e = 0; e < 2; ++e)        // block i = 2 pr + e: tile row 2 wm + pr, pixels 16 e .. + 15
        x[e] = *reinterpret_cast<const bf16x8*>(base + sl * PLANE + ((2 * wm + pr + kh) * PITCH + e * 16) * 64 + cq[kw]);
    };
#pragma unroll
    for (int h = 0; h < AHEAD && h < NH; ++h) frag(h, xf[h]);
#if FO_H64_PRIO
    __builtin_amdgcn_s_setprio(1);                         // (diagnostic variant: the wave in its MFMA loop outranks its SIMD partner's epilogue)
#endif
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int tap = h >> 2, sl = (h >> 1) & 1, pr = h & 1;
      if (h + AHEAD < NH) frag(h + AHEAD, xf[(h + AHEAD) % (AHEAD + 1)]);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[2 * pr + e][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap][sl][j], xf[h % (AHEAD + 1)][e], acc[2 * pr + e][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if FO_H64_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    H64_STAMP(h2);
    // ---- epilogue: acc[i][j][r] = channel 16 j + 4 quad + r of pixel (row 2 wm + (i >> 1), column 16 (i & 1) + l15)
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#if FO_H64_EPI_DEAD
    // (round 6) the four pixel blocks' line patches live in THIS tile's input stage, which is dead once every wave has left the MFMA loop (its rows 4, 5
    // were copied at the top; the next DMA goes to the other stage): all eight ds_write_b64 first, then the four ds_read_b128 + stores -- ONE LDS round
    // trip per tile instead of four (write -> read -> store per block through a single 1.25 KB patch: the epilogue was 37 % of a tile in the stamps)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned char* const epatch = lds + st * STAGE + wave * (4 * EPATCH);
#else
    unsigned char* const epatch = lds + 2 * STAGE + wave * EPATCH;
#endif
    // (max-pool 2x2 riding along: a wave's two tile rows are one pooled row; the vertical partner of a pixel is the same lane's other
    // accumulator, the horizontal one the neighbouring lane.  max commutes with the monotonic bias + ReLU + rounding, so the pooled tensor is
    // bit for bit the pool of the stored one.)
    float pm[2][2][4];
    unsigned rowbit[2][2] = {{0u, 0u}, {0u, 0u}};            // bit r: the lower pixel of the column is strictly larger (ties: the upper one, first in scan order)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t pix = ((size_t)n * a.H + ty * 4 + 2 * wm + (i >> 1)) * a.W + tx * 32 + (i & 1) * 16 + l15;
      unsigned obw = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = half * 64 + wn * 32 + j * 16 + quad * 4;
        float v[4] = {acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};
        if (MASKT || MASKB) {
          if (MASKB) {                                // the lane's 4 channels are one nibble: byte j * 2 + quad / 2 of the block's dword
            const unsigned nb = mw[i] >> (j * 16 + (quad >> 1) * 8 + (quad & 1) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((nb >> r) & 1u) ? v[r] : 0.f;
          } else {
            const bf16x4 mk = *reinterpret_cast<const bf16x4*>(a.mask + pix * a.ldMask + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (float)mk[r] > 0.f ? v[r] : 0.f;
          }
        }
        if (a.flags & FO_OUT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        if (LINES) *reinterpret_cast<bf16x4*>(epatch + (FO_H64_EPI_DEAD ? i * EPATCH : 0) + l15 * EPITCH + (j * 16 + quad * 4) * 2) = o;
        else if (!(FO_ABLATE_H & 32)) *reinterpret_cast<bf16x4*>(a.out + pix * a.ldOut + co) = o;
        if (OBITS) obw |= pos_bits4((float)o[0], (float)o[1], (float)o[2], (float)o[3]) << (j * 16 + quad * 4);      // the wave's 32 channels of this pixel: one dword
        if (POOL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (i >> 1) rowbit[i & 1][j] |= ((float)o[r] > pm[i & 1][j][r] ? 1u : 0u) << r;
            pm[i & 1][j][r] = (i >> 1) ? fmaxf(pm[i & 1][j][r], (float)o[r]) : (float)o[r];
          }
        }
      }
      if (LINES && !FO_H64_EPI_DEAD) {                     // lane = (pixel lane / 4 of the block, 16-byte piece lane % 4 of its 64 bytes)
        __builtin_amdgcn_wave_barrier();
        const bf16x8 ln = *reinterpret_cast<const bf16x8*>(epatch + (lane >> 2) * EPITCH + (lane & 3) * 16);
        const size_t pix2 = pix - l15 + (lane >> 2);
        if (!(FO_ABLATE_H & 32)) *reinterpret_cast<bf16x8*>(a.out + pix2 * a.ldOut + half * 64 + wn * 32 + (lane & 3) * 8) = ln;
        __builtin_amdgcn_wave_barrier();                  // (the next block's writes stay behind this read: LDS operations of a wave execute in order)
      }
      if (OBITS) {                                         // the four quads' nibbles meet in one dword (two shuffles): one 4-byte store per pixel
        obw = or_rows(obw);
        if (quad == 0) *reinterpret_cast<unsigned*>(a.outBits + pix * (a.Cout / 8) + half * 8 + wn * 4) = obw;
      }
      __builtin_amdgcn_sched_barrier(0);                  // (one pixel block's mask loads and addresses at a time: the filter holds the registers)
    }
    if (LINES && FO_H64_EPI_DEAD) {                        // lane = (pixel lane / 4 of a block, 16-byte piece lane % 4 of its 64 bytes)
      __builtin_amdgcn_wave_barrier();
      bf16x8 ln[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ln[i] = *reinterpret_cast<const bf16x8*>(epatch + i * EPATCH + (lane >> 2) * EPITCH + (lane & 3) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t pix2 = ((size_t)n * a.H + ty * 4 + 2 * wm + (i >> 1)) * a.W + tx * 32 + (i & 1) * 16 + (lane >> 2);
        if (!(FO_ABLATE_H & 32)) *reinterpret_cast<bf16x8*>(a.out + pix2 * a.ldOut + half * 64 + wn * 32 + (lane & 3) * 8) = ln[i];
      }
    }
    if (POOL) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const size_t ppix = ((size_t)n * (a.H / 2) + ty * 2 + wm) * (a.W / 2) + tx * 16 + e * 8 + (l15 >> 1);
        unsigned cw[2] = {0u, 0u}, pw = 0;                 // this lane's share of the pooled pixel's codes (8 bytes per wave) and sign plane (4 bytes)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float m[4];
          float pmR[4];                                    // the horizontal partner's column maximum
#pragma unroll
          for (int r = 0; r < 4; ++r) { pmR[r] = lane_xor1(pm[e][j][r]); m[r] = fmaxf(pm[e][j][r], pmR[r]); }
          if (a.pidx) {                                    // even lane = left column: the first maximum in scan order (0,0) (0,1) (1,0) (1,1)
            const unsigned rbR = (unsigned)lane_xor1((int)rowbit[e][j]);
            unsigned code = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float vL = pm[e][j][r], vR = pmR[r];
              const unsigned iL = ((rowbit[e][j] >> r) & 1u) * 2u, iR = ((rbR >> r) & 1u) * 2u + 1u;
              code |= (vL > vR ? iL : vR > vL ? iR : min(iL, iR)) << (2 * r);
            }
            cw[j] = code << (8 * quad);                    // byte j * 4 + quad of the wave's eight
          }
          if (a.pooledBits) pw |= pos_bits4((float)(__bf16)m[0], (float)(__bf16)m[1], (float)(__bf16)m[2], (float)(__bf16)m[3]) << (j * 16 + quad * 4);
          if (!(l15 & 1)) {
            const bf16x4 mo = {(__bf16)m[0], (__bf16)m[1], (__bf16)m[2], (__bf16)m[3]};
            // LINES: the wave's 16 pooled pixels (8 per e) through its patch, stored below as ONE instruction of 16-byte pieces instead of four of 8 bytes
            if (LINES) *reinterpret_cast<bf16x4*>(lds + 2 * STAGE + wave * EPATCH + (e * 8 + (l15 >> 1)) * EPITCH + (j * 16 + quad * 4) * 2) = mo;
            else *reinterpret_cast<bf16x4*>(a.pooled + ppix * a.ldPooled + half * 64 + wn * 32 + j * 16 + quad * 4) = mo;
          }
        }
        // the four quads' bytes / nibbles meet by two shuffles each: ONE 8-byte and ONE 4-byte store per pooled pixel instead of eight 1-byte stores
        // scattered over the wave (those made the launch 1.26 ms against 0.90 without them)
        if (a.pidx) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            cw[j] = or_rows(cw[j]);
          }
          if (!(l15 & 1) && quad == 0) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(a.pidx + ppix * (a.Cout / 4) + half * 16 + wn * 8) = u32x2{cw[0], cw[1]};
          }
        }
        if (a.pooledBits) {
          pw = or_rows(pw);
          if (!(l15 & 1) && quad == 0) *reinterpret_cast<unsigned*>(a.pooledBits + ppix * (a.Cout / 8) + half * 8 + wn * 4) = pw;
        }
      }
      if (LINES) {                                         // lane = (pooled pixel lane / 4 of the wave's 16, 16-byte piece lane % 4 of its 64 bytes)
        __builtin_amdgcn_wave_barrier();
        const int pp = lane >> 2;
        const bf16x8 ln = *reinterpret_cast<const bf16x8*>(lds + 2 * STAGE + wave * EPATCH + pp * EPITCH + (lane & 3) * 16);
        const size_t ppix2 = ((size_t)n * (a.H / 2) + ty * 2 + wm) * (a.W / 2) + tx * 16 + pp;       // (e = pp / 8, pooled column e * 8 + pp % 8 = pp)
        *reinterpret_cast<bf16x8*>(a.pooled + ppix2 * a.ldPooled + half * 64 + wn * 32 + (lane & 3) * 8) = ln;
        __builtin_amdgcn_wave_barrier();
      }
    }
    // the next patch has landed: vmcnt retires in order and the 8 youngest operations are this tile's stores, which may keep flying
    // (nothing reads them; the LDS stage they came from is not involved)
    H64_STAMP(h3);
    // (8 result stores per lane; + 4 bit-plane dwords; + 4 pooled stores, + 2 code stores, + 2 pooled-plane dwords.  A count that is too SMALL only
    // waits for a few of the stores as well; lgkmcnt: the row copy's LDS writes)
    {
      const int nst = (LINES ? 4 : 8) + (OBITS ? 4 : 0) + (POOL ? (LINES ? 1 : 4) : 0) + (POOL && a.pidx ? 2 : 0) + (POOL && a.pooledBits ? 2 : 0);
      if (nst >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      else if (nst >= 14) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
      else if (nst >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      else if (nst >= 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
      else if (nst >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    n = n1; ty = ty1; tx = tx1;
#ifdef FO_STAMP_H64
    H64_STAMP(h4);
    if (stamping) { sIssue += h1 - h0; sMfma += h2 - h1; sEpi += h3 - h2; sWait += h4 - h3; ++nT; }
#endif
  }
#ifdef FO_STAMP_H64
  if (stamping && (threadIdx.x & 63) == 0) {
    unsigned long long k1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(k1), "=s"(r1)::"memory");
    fo_h64_stamps[0] = sIssue; fo_h64_stamps[1] = sMfma; fo_h64_stamps[2] = sEpi; fo_h64_stamps[3] = sWait; fo_h64_stamps[4] = nT;
    fo_h64_stamps[5] = k1 - hK0; fo_h64_stamps[6] = r1 - hR0;
  }
#endif
}

// ------------------------------------------------------------------------------------------------ VGG conv1_1 + conv1_2 in one launch
// relu1_1 (64 channels at full resolution: 1.34 GB in bf16 at 160 x 256^2) is written by conv1_1 only to be read back -- 1.6 x, through the halo
// overlap -- by conv1_2; the ground-truth branch of LPIPS (lpips.py:80-93) never needs it again, the reconstruction branch only as the ReLU
// mask of its backward.  Here the halo-tile kernel above MAKES its input patch instead of fetching it: per tile the 8 x 36-pixel patch of the
// scaled image (16 bytes per pixel: 4.6 KB) is DMA'd, conv1_1 (K = 9 taps x 8 channels = 72, padded to 96) runs on it for the 6 x 34 pixels of
// the relu1_1 patch -- v_mfma_f32_16x16x32_bf16 with the filter (13 KB, in LDS) as the row operand, a pixel's tap as one 16-byte fragment read --
// and bias + ReLU + zero halo + bf16 rounding go straight into the LDS image the nine taps of conv1_2 read (same layout and swizzle).  +37 %
// MFMA work in a launch that was bound by its 2.1 GB of patch reads; conv1_1's own launch (0.40 ms) disappears.  out1 (relu1_1) is optional.
struct Vgg1Args {
  const __bf16* x8;          // [N][H][W][8]
  const __bf16* wp1;         // [64][128], k = tap * 8 + channel (zero from k = 72 up)
  const float* b1;
  const __bf16* wp2;         // [64][9][64]
  const float* b2;
  __bf16* out1;              // relu1_1 [N][H][W][64] (OUT1) or unused
  __bf16* out2;              // relu1_2 [N][H][W][64]
  __bf16* pooled;            // max-pool 2x2 of relu1_2 [N][H/2][W/2][64] (POOL) or unused
  int N, H, W, tilesX, tilesY, ntiles;
  unsigned x8Bytes, out1Bytes;
};

template <bool OUT1, bool POOL>
__global__ __launch_bounds__(256, 2) void vgg_conv1_fused_bf16_kernel(const Vgg1Args a) {
  constexpr int PITCH = 48, ROWS = 6, PLANE = ROWS * PITCH * 64;       // the relu1_1 patch: two 32-channel planes of 64-byte pixels
  constexpr int OFF_RGB = 2 * PLANE, RGBB = 8 * 1024;                  // two image-patch buffers of 512 pixels x 16 B (288 used)
  constexpr int OFF_W1 = OFF_RGB + 2 * RGBB, W1P = 208;                // conv1_1 filter: 64 rows of 96 k, row pitch 208 B (conflict-free 16-B reads)
  constexpr int OFF_B1 = OFF_W1 + 64 * W1P;                            // conv1_1 bias, 64 floats
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x8), 0, a.x8Bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(a.out1, 0, OUT1 ? a.out1Bytes : 0u, 0x00020000);
  lds_byte* const lds3 = (lds_byte*)lds;

  // conv1_2's filter fragments, resident (as conv_halo64_bf16_kernel)
  bf16x8 wf[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[t][sl][j] = *reinterpret_cast<const bf16x8*>(a.wp2 + (size_t)(wn * 32 + j * 16 + l15) * 576 + t * 64 + sl * 32 + quad * 8);
  // conv1_1's filter and bias -> LDS
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int id = tid + 256 * i, row = id / 12, c16 = id - row * 12;
    *reinterpret_cast<u32x4*>(lds + OFF_W1 + row * W1P + c16 * 16) = *reinterpret_cast<const u32x4*>(a.wp1 + (size_t)row * 128 + c16 * 8);
  }
  if (tid < 64) reinterpret_cast<float*>(lds + OFF_B1)[tid] = a.b1[tid];

  auto dma_rgb = [&](int tile, int buf) {
    const bool live = tile < a.ntiles;
    const int tx = tile % a.tilesX, r1 = tile / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pc = wave + 4 * k;                         // piece of 64 pixels; patch pixel q = row * 36 + column, rows y0 - 2 .., columns x0 - 2 ..
      const int q = pc * 64 + lane, r = q / 36, c = q - r * 36;
      const int iy = ty * 4 - 2 + r, ix = tx * 32 - 2 + c;
      const bool ok = live & (q < 288) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
      dma16(rx, lds3 + OFF_RGB + buf * RGBB + pc * 1024, ok ? (unsigned)((((size_t)n * a.H + iy) * a.W + ix) * 16) : OOB);
    }
  };

  // per-lane constants of the producer: the tap this lane's k-chunk belongs to in each of the three 32-deep steps (taps >= 9: the filter is zero there)
  int xoff[3];
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2) {
    const int t = min(4 * s2 + quad, 8);
    xoff[s2] = ((t / 3) * 36 + (t % 3)) * 16;
  }
  int cq[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) cq[kw] = (l15 + kw) * 64 + ((quad ^ swz(l15 + kw)) * 16);
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = a.b2[wn * 32 + j * 16 + quad * 4 + r];

  int tile = blockIdx.x;
  dma_rgb(tile, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int it = 0; tile < a.ntiles; tile += gridDim.x, ++it) {
    const int buf = it & 1;
    const int tx = tile % a.tilesX, r1 = tile / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    dma_rgb(tile + gridDim.x, buf ^ 1);                    // the next tile's image patch flies during this tile's work
    // ---- producer: relu1_1 on the 6 x 34 patch (18 blocks of 16 pixels: patch row blk / 3, columns 16 (blk % 3) ..; dealt over the 4 waves)
    const unsigned char* const rgb = lds + OFF_RGB + buf * RGBB;
    // conv1_1's filter fragments live in registers for the length of this phase only (conv1_2's accumulators and fragments are dead here), one
    // 32-channel half of the output at a time: read per block they made the phase a chain of dependent LDS reads -- 15 per 12 MFMAs -- and the
    // launch no faster than the two it replaces; all twelve at once spilled
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {                       // output channels 32 hc .. + 31 = plane hc of the patch
      bf16x8 w1r[2][3];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
          w1r[j][s2] = *reinterpret_cast<const bf16x8*>(lds + OFF_W1 + ((hc * 2 + j) * 16 + l15) * W1P + s2 * 64 + quad * 16);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int blk = wave + 4 * k;
        const int r = blk / 3, c = (blk - r * 3) * 16 + l15;
        const bool live = blk < 18;                        // (wave-uniform)
        f32x4 p1[2];
        if (live) {
          p1[0] = p1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
          const unsigned char* const xb = rgb + (r * 36 + c) * 16;
          bf16x8 xf1[3];
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) xf1[s2] = *reinterpret_cast<const bf16x8*>(xb + xoff[s2]);
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
            for (int j = 0; j < 2; ++j) p1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[j][s2], xf1[s2], p1[j], 0, 0, 0);
        }
        const int iy = ty * 4 - 1 + r, ix = tx * 32 - 1 + c;
        const bool inimg = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W) & (c < 34);
        const bool interior = live & (r >= 1) & (r <= 4) & (c >= 1) & (c <= 32);
        const unsigned o1 = interior ? (unsigned)((((size_t)n * a.H + iy) * a.W + ix) * 128 + hc * 64 + quad * 8) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x4 o = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
          if (live) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + (hc * 32 + j * 16 + quad * 4) * 4);
            const float v0 = inimg ? fmaxf(p1[j][0] + b4.x, 0.f) : 0.f, v1 = inimg ? fmaxf(p1[j][1] + b4.y, 0.f) : 0.f;
            const float v2 = inimg ? fmaxf(p1[j][2] + b4.z, 0.f) : 0.f, v3 = inimg ? fmaxf(p1[j][3] + b4.w, 0.f) : 0.f;
            o = bf16x4{(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
            // channel 32 hc + 16 j + 4 quad .. + 3 of patch pixel (r, c): plane hc, logical 16-byte chunk 2 j + (quad >> 1), half quad & 1
            *reinterpret_cast<bf16x4*>(lds + hc * PLANE + (r * PITCH + c) * 64 + (((2 * j + (quad >> 1)) ^ swz(c & 15)) * 16) + (quad & 1) * 8) = o;
          }
          if (OUT1) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ro1, o1 == OOB ? OOB : o1 + j * 32, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // the patch is complete
    // ---- conv1_2 on the patch: conv_halo64_bf16_kernel's loop
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 36; ++h) {
      const int tap = h >> 2, sl = (h >> 1) & 1, pr = h & 1, kh = tap / 3, kw = tap - kh * 3;
      bf16x8 xf[2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
        xf[e] = *reinterpret_cast<const bf16x8*>(lds + sl * PLANE + ((2 * wm + pr + kh) * PITCH + e * 16) * 64 + cq[kw]);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[2 * pr + e][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap][sl][j], xf[e], acc[2 * pr + e][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: bias + ReLU + bf16, 8-byte stores; the 2x2 max-pool rides along (see conv_halo64_bf16_kernel)
    float pm[2][2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t pix = ((size_t)n * a.H + ty * 4 + 2 * wm + (i >> 1)) * a.W + tx * 32 + (i & 1) * 16 + l15;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = wn * 32 + j * 16 + quad * 4;
        const bf16x4 o = {(__bf16)fmaxf(acc[i][j][0] + bv[j][0], 0.f), (__bf16)fmaxf(acc[i][j][1] + bv[j][1], 0.f),
                          (__bf16)fmaxf(acc[i][j][2] + bv[j][2], 0.f), (__bf16)fmaxf(acc[i][j][3] + bv[j][3], 0.f)};
        *reinterpret_cast<bf16x4*>(a.out2 + pix * 64 + co) = o;
        if (POOL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pm[i & 1][j][r] = (i >> 1) ? fmaxf(pm[i & 1][j][r], (float)o[r]) : (float)o[r];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (POOL) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const size_t ppix = ((size_t)n * (a.H / 2) + ty * 2 + wm) * (a.W / 2) + tx * 16 + e * 8 + (l15 >> 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float m[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) m[r] = fmaxf(pm[e][j][r], lane_xor1(pm[e][j][r]));
          if (!(l15 & 1))
            *reinterpret_cast<bf16x4*>(a.pooled + ppix * 64 + wn * 32 + j * 16 + quad * 4) = bf16x4{(__bf16)m[0], (__bf16)m[1], (__bf16)m[2], (__bf16)m[3]};
        }
      }
    }
    // the next image patch has landed: vmcnt retires in order and everything younger than its two DMAs is this tile's stores, which may fly on
    wait_vmcnt_n<(OUT1 ? 20 : 0) + 8 + (POOL ? 4 : 0)>();
    __builtin_amdgcn_s_barrier();                          // ... and every wave is done with the relu1_1 patch
  }
}

static int launch_halo64(const ConvArgsH& c, hipStream_t s, void* pooled = nullptr, int ldPooled = 0, void* pidx = nullptr, void* pooledBits = nullptr) {
  const fo_conv_desc& d = c.d;
  HaloArgs a;
  a.pooled = reinterpret_cast<__bf16*>(pooled); a.ldPooled = ldPooled; a.pidx = pooled ? reinterpret_cast<unsigned char*>(pidx) : nullptr;
  a.maskBits = c.maskBits; a.outBits = c.outBits; a.pooledBits = pooled ? reinterpret_cast<unsigned char*>(pooledBits) : nullptr;
  a.in = reinterpret_cast<const __bf16*>(c.in); a.wp = reinterpret_cast<const __bf16*>(c.wp); a.bias = c.bias;
  a.mask = reinterpret_cast<const __bf16*>(c.mask); a.out = reinterpret_cast<__bf16*>(c.out);
  a.N = d.N; a.H = d.Hin; a.W = d.Win; a.Cout = d.Cout; a.halves = d.Cout / 64;
  a.tilesX = d.Win / 32; a.tilesY = d.Hin / 4; a.ntiles = d.N * a.tilesX * a.tilesY;
  a.ldIn = d.ldIn; a.ldOut = d.ldOut; a.ldMask = d.ldMask; a.flags = d.flags; a.inBytes = c.inBytes;
  const bool lines = !getenv("FACEOFF_H64_NO_LINES");       // (read per call: tools/ab_bf16.py A/Bs the two epilogues in one process)
  const int ldsBytes = 2 * 2 * 6 * 48 * 64 + (lines ? 4 * 16 * 80 : 0);            // two stages of two planes (+ the four waves' epilogue patches)
  const bool maskt = (d.flags & FO_MASK) && !a.maskBits, maskb = (d.flags & FO_MASK) && a.maskBits;
  FO_REQUIRE(!(a.pooled && ((d.flags & FO_MASK) || a.outBits)), FO_E_SHAPE, "conv_bf16 (halo64): pooled output with a mask or out_bits is not built");
  const int cus = fo_cu_count();
  int grid = std::min(2 * cus / a.halves * a.halves, a.ntiles * a.halves);          // two workgroups per CU
  grid = std::max(a.halves, grid / a.halves * a.halves);
#define FO_HALO64(MT, MB, PL, OB)                                                                                                            \
  do {                                                                                                                                       \
    static fo_lds_once once;                                                                                                                 \
    static fo_lds_once once0;                                                                                                                \
    void (*kern)(const HaloArgs) = conv_halo64_bf16_kernel<MT, MB, PL, OB, true>;                                                             \
    void (*kern0)(const HaloArgs) = conv_halo64_bf16_kernel<MT, MB, PL, OB, false>;                                                           \
    if (!fo_lds_optin(lines ? once : once0, reinterpret_cast<const void*>(lines ? kern : kern0), ldsBytes, "conv_bf16 (halo64)")) return FO_E_HIP; \
    if (lines) FO_NOTE_T("conv_halo64_bf16_kernel", MT, MB, PL, OB, true); else FO_NOTE_T("conv_halo64_bf16_kernel", MT, MB, PL, OB, false);  \
    hipLaunchKernelGGL(lines ? kern : kern0, dim3(grid), dim3(256), ldsBytes, s, a);                                                          \
  } while (0)
  if (a.pooled) FO_HALO64(false, false, true, false);
  else if (maskt) { if (a.outBits) FO_HALO64(true, false, false, true); else FO_HALO64(true, false, false, false); }
  else if (maskb) FO_HALO64(false, true, false, false);
  else if (a.outBits) FO_HALO64(false, false, false, true);
  else FO_HALO64(false, false, false, false);
#undef FO_HALO64
  FO_CHECK_LAUNCH();
  return FO_OK;
}

template <int BMB, int BN, int WAVES_M, int WAVES_N>
int launch_pp16(const ConvArgsH& a, hipStream_t s) {
  constexpr int ldsBytes = 4 * (BMB + BN) * 64;
  static fo_lds_once once;
  void (*kern)(const ConvArgsH) = conv_bf16_pp16_kernel<BMB, BN, WAVES_M, WAVES_N>;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(kern), ldsBytes, "conv_bf16 (pp16)")) return FO_E_HIP;
  FO_NOTE_T("conv_bf16_pp16_kernel", BMB, BN, WAVES_M, WAVES_N);
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN), dim3(512), ldsBytes, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

template <int BMB, int BN, int WAVES_M, int WAVES_N, int NPE>
int launch_pph(const ConvArgsH& a, hipStream_t s) {
  constexpr int ldsBytes = PphLds<BMB, BN, WAVES_M, WAVES_N, NPE>::BYTES;
  static fo_lds_once once;
  void (*kern)(const ConvArgsH) = conv_bf16_pph_kernel<BMB, BN, WAVES_M, WAVES_N, NPE>;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(kern), ldsBytes, "conv_bf16 (pph)")) return FO_E_HIP;
  // one workgroup per tile; FACEOFF_BF16_PPH_PERSIST=1: one per CU walking tiles blockIdx.x, + grid, ... (a multiple of 8 workgroups: the XCD remap
  // stays consistent) -- faster alone, slower beside the side streams (see the kernel)
  const char* pe = getenv("FACEOFF_BF16_PPH_PERSIST");
  const int ntiles = a.tilesM * a.tilesN;
  const int cus = fo_cu_count() / 8 * 8;
  const int grid = (pe && atoi(pe)) && cus >= 8 ? std::min(ntiles, cus) : ntiles;
  FO_NOTE_T("conv_bf16_pph_kernel", BMB, BN, WAVES_M, WAVES_N, NPE);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsBytes, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// the extended-tile kernel takes 3x3 (x3) pad-1 layers whose BMB-row tiles are whole image rows of one frame; NPE = 16-row pieces per wave
template <int BMB>
static int pph_pieces(const fo_conv_desc* d, const ConvArgsH& a) {
  const char* off = getenv("FACEOFF_BF16_NO_PPH");                   // diagnostics / A-B: conv_bf16_pp16_kernel everywhere
  if (off && atoi(off)) return 0;
  if (d->KH != 3 || d->KW != 3 || d->padH != 1 || d->padW != 1 || (d->KD != 1 && (d->KD != 3 || d->padD != 1)) || (d->KD == 1 && d->padD != 0)) return 0;
  const int W = d->Wm;
  if (W % 16 != 0 || BMB % W != 0 || a.HWm % BMB != 0 || a.M % BMB != 0 || d->Cout > 1024) return 0;      // (1024: the bias patch in LDS)
  return (BMB + 2 * W + 127) / 128;
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
int launch_dma(const ConvArgsH& a, hipStream_t s) {
  FO_NOTE_T("conv_bf16_dma_kernel", BN, WAVES_M, WAVES_N, TM, TN);
  hipLaunchKernelGGL((conv_bf16_dma_kernel<BN, WAVES_M, WAVES_N, TM, TN>), dim3(a.tilesM * a.tilesN), dim3(256), 0, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
int launch(const ConvArgsH& a, bool smallc, hipStream_t s) {
  const int grid = a.tilesM * a.tilesN;
  if (smallc) {
    FO_NOTE_T("conv_bf16_kernel", BN, WAVES_M, WAVES_N, TM, TN, true, false);
    hipLaunchKernelGGL((conv_bf16_kernel<BN, WAVES_M, WAVES_N, TM, TN, true, false>), dim3(grid), dim3(256), 0, s, a);
  } else if (a.d.flags & FO_IN_RELU) {
    FO_NOTE_T("conv_bf16_kernel", BN, WAVES_M, WAVES_N, TM, TN, false, true);
    hipLaunchKernelGGL((conv_bf16_kernel<BN, WAVES_M, WAVES_N, TM, TN, false, true>), dim3(grid), dim3(256), 0, s, a);
  } else {
    FO_NOTE_T("conv_bf16_kernel", BN, WAVES_M, WAVES_N, TM, TN, false, false);
    hipLaunchKernelGGL((conv_bf16_kernel<BN, WAVES_M, WAVES_N, TM, TN, false, false>), dim3(grid), dim3(256), 0, s, a);
  }
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// wp[o][t][i] = bf16(w[o][i][t])  (o < O, i < I, t < taps; zero elsewhere), t < tapsPad
__global__ void pack_conv_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int O, int I, int taps, int Opad,
                                      int Ipad, int tapsPad) {
  const size_t total = (size_t)Opad * tapsPad * Ipad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = e % Ipad;
    const int t = (e / Ipad) % tapsPad;
    const int o = e / ((size_t)Ipad * tapsPad);
    wp[e] = (__bf16)((o < O && i < I && t < taps) ? w[((size_t)o * I + i) * taps + t] : 0.f);
  }
}

// wp[i][t][o] = bf16(w[o][i][taps-1-t])  (stride-1 dgrad: flipped taps, swapped channels)
__global__ void pack_conv_dgrad_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int O, int I, int taps,
                                            int Opad, int Ipad) {
  const size_t total = (size_t)Ipad * taps * Opad;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int o = e % Opad;
    const int t = (e / Opad) % taps;
    const int i = e / ((size_t)Opad * taps);
    wp[e] = (__bf16)((o < O && i < I) ? w[((size_t)o * I + i) * taps + (taps - 1 - t)] : 0.f);
  }
}

inline int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 8192); }

}  // namespace

#ifdef FO_STAMP_H64
extern "C" int fo_debug_read_h64_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fo_h64_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
#ifdef FO_STAMP_PPH
extern "C" int fo_debug_read_pph_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fo_pph_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

int fo_pack_conv_bf16(const float* w, void* wp, int O, int I, int taps, int Opad, int Ipad, int tapsPad, void* stream) {
  FO_REQUIRE(Opad >= O && Ipad >= I && taps > 0 && tapsPad >= taps, FO_E_SHAPE, "pack_conv_bf16: bad padding");
  hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3(grid_for((size_t)Opad * tapsPad * Ipad)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<__bf16*>(wp), O, I, taps, Opad, Ipad, tapsPad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_pack_conv_dgrad_bf16(const float* w, void* wp, int O, int I, int taps, int Opad, int Ipad, void* stream) {
  FO_REQUIRE(Opad >= O && Ipad >= I && taps > 0, FO_E_SHAPE, "pack_conv_dgrad_bf16: bad padding");
  hipLaunchKernelGGL(pack_conv_dgrad_bf16_kernel, dim3(grid_for((size_t)Opad * taps * Ipad)), dim3(256), 0, (hipStream_t)stream,
                     w, reinterpret_cast<__bf16*>(wp), O, I, taps, Opad, Ipad);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

static int conv_bf16_impl(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, const void* add,
                          void* out, void* stream, void* pooled = nullptr, int ldPooled = 0, void* pidx = nullptr, const void* maskBits = nullptr,
                          void* outBits = nullptr, void* pooledBits = nullptr) {
  ConvArgsH a;
  a.d = *d;
  a.maskBits = reinterpret_cast<const unsigned char*>(maskBits);
  a.outBits = reinterpret_cast<unsigned char*>(outBits);
  a.ldBits = d->Cout / 8;
  {   // diagnostics (timing only, WRONG gradients): no ReLU-mask reads at all -- what a bit plane in their place could save at most
    static const char* nomask = getenv("FACEOFF_DIAG_BF16_NO_MASK");
    if (nomask && atoi(nomask)) { a.d.flags &= ~FO_MASK; a.maskBits = nullptr; maskBits = nullptr; }
  }
  d = &a.d;
  a.in = in; a.wp = wp; a.bias = bias; a.mask = mask; a.add = add; a.out = out;
  const int flags = d->flags;
  const bool f32out = flags & FO_OUT_F32, d2s = flags & FO_DEPTH2SPACE;
  FO_REQUIRE(d->N > 0 && d->T > 0 && d->N % d->T == 0 && d->KD >= 1 && (d->KD == 1 || d->stride == 1), FO_E_SHAPE,
             "conv_bf16: N=%d must be whole clips of T=%d; depth taps need stride 1", d->N, d->T);
  FO_REQUIRE(!(flags & ~(FO_IN_RELU | FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU | FO_DEPTH2SPACE | FO_OUT_F32)), FO_E_SHAPE, "conv_bf16: unknown flag");
  FO_REQUIRE(d->ldIn % 8 == 0 && fo_aligned16(in) && fo_aligned16(wp) && fo_aligned16(out) && d->ldOut % (f32out ? 4 : 8) == 0, FO_E_ALIGN,
             "conv_bf16: 16-byte alignment (bf16 ld %% 8 == 0, fp32 ld %% 4 == 0)");
  if (d2s) {
    FO_REQUIRE(d->Cout % 32 == 0 && d->ldOut >= (d->ophW + 7) / 8 * 8 && (d->ophH == 0 || d->ophH == 1) && d->Hout == 2 * (d->Hm - d->ophH) &&
                   d->Wout == 2 * (d->Wm - d->ophH) && d->ophW >= 1 && d->ophW <= d->Cout / 4 && (d->Cout == 32 || d->ophH == 1),
               FO_E_SHAPE, "conv_bf16: FO_DEPTH2SPACE needs Cout = 4 x (a multiple of 8) columns, 1 <= ophW <= Cout / 4 real channels, ldOut >= ophW rounded up to 8, "
               "a 2x output grid (more than 8 columns per phase: the cell form, ophH = 1)");
  } else {
    FO_REQUIRE(d->ldOut >= (d->Cout + 7) / 8 * 8, FO_E_ALIGN, "conv_bf16: ldOut must hold Cout rounded up to 8");
  }
  FO_REQUIRE(!(flags & FO_MASK) || maskBits || (mask && fo_aligned16(mask) && d->ldMask % 8 == 0), FO_E_ALIGN, "conv_bf16: mask alignment");
  FO_REQUIRE((!maskBits && !outBits && !pooledBits) || (d->Cout % 8 == 0 && !d2s && !f32out), FO_E_SHAPE,
             "conv_bf16: bit planes need Cout %% 8 == 0, a bf16 result and no FO_DEPTH2SPACE");
  FO_REQUIRE(!maskBits || (flags & FO_MASK), FO_E_SHAPE, "conv_bf16: mask_bits without FO_MASK");
  FO_REQUIRE(!outBits || !(flags & FO_ADD), FO_E_SHAPE, "conv_bf16: out_bits with FO_ADD is not built");
  FO_REQUIRE(!pooledBits || pooled, FO_E_SHAPE, "conv_bf16: pooled_bits without pooled");
  FO_REQUIRE(!(flags & FO_ADD) || (add && fo_aligned16(add) && d->ldAdd % 8 == 0), FO_E_ALIGN, "conv_bf16: add alignment");
  FO_REQUIRE(!(flags & FO_BIAS) || bias, FO_E_SHAPE, "conv_bf16: FO_BIAS without bias");
  const int taps = d->KD * d->KH * d->KW;
  FO_REQUIRE(taps >= 1 && taps <= 31, FO_E_SHAPE, "conv_bf16: at most 31 taps (got %d)", taps);
  const bool smallc = d->Cin < 32;
  if (smallc) {
    FO_REQUIRE(d->Cin == 8 && d->ldIn == 8 && d->KD == 1 && !(flags & FO_IN_RELU), FO_E_SHAPE,
               "conv_bf16: small Cin must be 8 with 16-byte pixels, 2-D, no input ReLU (got %d)", d->Cin);
    a.Ktot = (taps + 7) / 8 * 64;
  } else {
    FO_REQUIRE(d->Cin % 32 == 0, FO_E_SHAPE, "conv_bf16: Cin=%d must be a multiple of 32 (or 8)", d->Cin);
    a.Ktot = taps * d->Cin;
  }
  a.HWm = d->Hm * d->Wm;
  const long long M = (long long)d->N * a.HWm;
  FO_REQUIRE(M > 0 && M < (1ll << 31), FO_E_SHAPE, "conv_bf16: M out of range");
  a.M = (int)M;
  a.tilesM = (a.M + BM - 1) / BM;
  const unsigned long long inBytes = (((unsigned long long)d->N * d->Hin * d->Win - 1) * d->ldIn + d->Cin) * 2ull;
  const int opad = d->Cout > 64 ? (d->Cout + 127) / 128 * 128 : (d->Cout > 32 ? 64 : 32);
  const unsigned long long wpBytes = (unsigned long long)opad * a.Ktot * 2ull;
  // (the offset arithmetic is 32-bit and starts up to one padding frame + row + column below the tensor)
  const unsigned long long margin = ((((unsigned long long)d->padD * d->Hin + d->padH) * d->Win + d->padW) * d->ldIn + 64) * 2ull;
  FO_REQUIRE(inBytes + margin < (1ull << 31) && wpBytes < (1ull << 31), FO_E_SHAPE, "conv_bf16: tensor exceeds the 2 GiB buffer-descriptor window");
  a.inBytes = (unsigned)inBytes;
  a.wpBytes = (unsigned)wpBytes;
  hipStream_t s = (hipStream_t)stream;
  const bool plain_epi = !(flags & (FO_ADD | FO_OUT_F32 | FO_DEPTH2SPACE | FO_IN_RELU));
  const char* norgb = getenv("FACEOFF_BF16_NO_RGB");                 // diagnostics: the tiled kernel for the RGB layer too
  if (smallc && plain_epi && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->ostride == 1 && d->padH == 1 && d->padW == 1 &&
      d->Hin == d->Hm && d->Win == d->Wm && d->Hm == d->Hout && d->Wm == d->Wout && d->ldOut == 64 && !(flags & FO_MASK) &&
      a.Ktot == 128 && inBytes < (1ull << 31) && !(norgb && atoi(norgb))) {
    const int nblocks = (a.M + 31) / 32;
    const int grid = std::min((nblocks + 3) / 4, fo_cu_count() * 3);
    if (a.outBits) {
      FO_NOTE("conv_rgb_bf16_kernel<true>");
      hipLaunchKernelGGL(conv_rgb_bf16_kernel<true>, dim3(grid), dim3(256), 0, s, a, nblocks);
    } else {
      FO_NOTE("conv_rgb_bf16_kernel<false>");
      hipLaunchKernelGGL(conv_rgb_bf16_kernel<false>, dim3(grid), dim3(256), 0, s, a, nblocks);
    }
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  if (!smallc && d->KD == 1 && d->Cin == 64 && d->ldIn == 64 && d->Cout <= 8 && d->ldOut == 8 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->ostride == 1 &&
      d->padH == 1 && d->padW == 1 && d->Hin == d->Hm && d->Win == d->Wm && d->Hm == d->Hout && d->Wm == d->Wout && flags == 0 &&
      d->Win % 64 == 0 && a.M >= 64 * 1024 && !(norgb && atoi(norgb)) && !outBits) {
    const bool ring = !getenv("FACEOFF_RGB_DGRAD_NO_RING");          // (read per call: tests compare the two kernels in one process)
    if (ring) {                                                              // round 5: every gradient row fetched once (ring of eight rows in LDS)
      constexpr int ringBytes = 8 * 9 * 1024;
      static fo_lds_once once_r;
      if (!fo_lds_optin(once_r, reinterpret_cast<const void*>(conv_rgb_dgrad_ring_bf16_kernel), ringBytes, "conv_bf16 (rgb dgrad, ring)")) return FO_E_HIP;
      const int spr = d->Win / 64, nstrips = d->N * spr;
      FO_NOTE("conv_rgb_dgrad_ring_bf16_kernel");
      hipLaunchKernelGGL(conv_rgb_dgrad_ring_bf16_kernel, dim3(2 * fo_cu_count()), dim3(256), ringBytes, s, a, nstrips, spr);
      FO_CHECK_LAUNCH();
      return FO_OK;
    }
    constexpr int ldsBytes = 2 * 28 * 1024;
    static fo_lds_once once;
    if (!fo_lds_optin(once, reinterpret_cast<const void*>(conv_rgb_dgrad_bf16_kernel), ldsBytes, "conv_bf16 (rgb dgrad)")) return FO_E_HIP;
    const int segsPerRow = d->Win / 64, nseg = d->N * d->Hin * segsPerRow;
    FO_NOTE("conv_rgb_dgrad_bf16_kernel");
    hipLaunchKernelGGL(conv_rgb_dgrad_bf16_kernel, dim3(std::min(nseg, 2 * fo_cu_count())), dim3(256), ldsBytes, s, a, nseg, segsPerRow);
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  // a ResBlock's 3x3 128 -> 32 (ReLU in, bias, ReLU out): its own halo-tile kernel (resblock_bf16.hip)
  const bool planes = maskBits || outBits || pooledBits;
  if (!planes && !mask && !add && !pooled && d->Cin == 128 && d->Cout == 32 && fo_conv3x3_c128to32_halo_bf16_try(d, in, wp, bias, out, s)) {
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  if (!planes && !bias && !pooled && d->Cin == 32 && d->Cout == 128 && fo_conv3x3_c32to128_halo_bf16_try(d, in, wp, mask, add, out, s)) {   // ... and its data gradient
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  // 64 input channels, 3x3, same size, frames of whole 4 x 32 tiles: the halo-tile kernel (filter in registers, input patch staged once)
  {
    const char* nohalo = getenv("FACEOFF_BF16_NO_HALO");             // diagnostics / A-B
    const char* fhalo = getenv("FACEOFF_BF16_FORCE_HALO");           // tests: at any size
    const long long tiles = (long long)d->N * (d->Hin / 4) * (d->Win / 32) * (d->Cout / 64);
    if (!smallc && d->KD == 1 && d->Cin == 64 && d->Cout % 64 == 0 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->ostride == 1 && d->padH == 1 && d->padW == 1 &&
        d->Hin == d->Hm && d->Win == d->Wm && d->Hm == d->Hout && d->Wm == d->Wout && d->Hin % 4 == 0 && d->Win % 32 == 0 &&
        !(flags & ~(FO_BIAS | FO_MASK | FO_OUT_RELU)) && d->ldOut % 4 == 0 && (!(flags & FO_MASK) || d->ldMask % 4 == 0) &&
        (tiles >= 8LL * fo_cu_count() || (fhalo && atoi(fhalo))) && !(nohalo && atoi(nohalo)) &&
        !(maskBits && outBits))                            // (that instantiation spills; nothing asks for it: a data gradient writes no plane)
      return launch_halo64(a, s, pooled, ldPooled, pidx, pooledBits);
  }
  FO_REQUIRE(!pooled && !pooledBits, FO_E_SHAPE, "conv_bf16: the pooled second output exists for the 64-input-channel halo-tile kernel only (3x3, whole 4 x 32 tiles, >= 8 tiles per CU)");
  // big tiles (one workgroup per CU) where the launch still fills the chip for a few rounds: stride-1 same-size convs.
  // Measured at the C3 shapes and at a fifth of them (tools/ab_bf16.py): 256-column tiles +22...30 % over conv_bf16_kernel
  // on conv3_x / conv4_x, 128-column tiles +5...10 % on conv2_x and +12...15 % on 2-round launches, the 512 x 128 tile (32 MFMAs per
  // phase and wave instead of 16: half the barriers per FLOP) another +14...17 % where there are >= 5 whole rounds of them (round 3: the VQ-VAE's 64^2 latents are exactly 5 rounds, config 3 -0.6...-1.6 ms; the first threshold, 8, was only ever measured on the VGG shapes) (a 512 x 64 tile for the 64-column layers measured -8 %); 64-column layers
  // (K = 576: 18 phases) stay on conv_bf16_kernel, whose second workgroup hides the prologue and epilogue
  // the cell form of a k4 s2 transposed layer (FO_DEPTH2SPACE, 4 Cpp >= 256 GEMM columns, K = 4 Cin: 16 phases of 32): the per-tap big-tile kernel instead of the
  // 128-row one -- twice the MFMAs per barrier and the A tile staged once per 256 (128) columns (FACEOFF_BF16_CELLS_SMALL_TILES=1: the 128-row kernel)
  if (d2s && d->ophH == 1 && !smallc && d->KD == 1 && d->KH == 2 && d->KW == 2 && d->stride == 1 && !(flags & FO_IN_RELU) && d->Cout % 128 == 0 && !maskBits && !outBits &&
      !getenv("FACEOFF_BF16_CELLS_SMALL_TILES")) {
    a.cinChunks = d->Cin / 32;
    a.ksteps = a.Ktot / 32;
    const long long tilesM256 = (a.M + 255) / 256;
    const char* forceb = getenv("FACEOFF_BF16_BIG_TILES");           // tests: at any size
    const long long cusb = (forceb && atoi(forceb)) ? 0 : fo_cu_count();
    const char* n256 = getenv("FACEOFF_BF16_CELLS_NO_256");          // tests: the 256 x 128 tile where the 256 x 256 one would run
    if (d->Cout % 256 == 0 && tilesM256 * (d->Cout / 256) >= 3 * cusb && !(n256 && atoi(n256))) {
      a.tilesM = (int)tilesM256; a.tilesN = d->Cout / 256;
      a.frameTiles = a.HWm % 256 == 0;
      return launch_pp16<256, 256, 2, 4>(a, s);
    }
    if (tilesM256 * (d->Cout / 128) >= 2 * cusb) {
      a.tilesM = (int)tilesM256; a.tilesN = d->Cout / 128;
      a.frameTiles = a.HWm % 256 == 0;
      return launch_pp16<256, 128, 4, 2>(a, s);
    }
  }
  const bool same = d->stride == 1 && d->ostride == 1 && d->Hm == d->Hout && d->Wm == d->Wout && d->Hin == d->Hm && d->Win == d->Wm && !d2s;
  const char* nobig = getenv("FACEOFF_BF16_SMALL_TILES");            // diagnostics / tests: never
  const char* force = getenv("FACEOFF_BF16_BIG_TILES");              // tests: at any size
  const bool odd32 = !smallc && d->Cin % 64 != 0;                    // 32-channel inputs (the ResBlocks' hidden tensor): only the 256-row kernels walk K in 32-deep tiles
  if (!smallc && same && !(flags & FO_IN_RELU) && (!(nobig && atoi(nobig)) || odd32) && d->Cout % 128 == 0) {
    a.cinChunks = d->Cin / 32;
    a.ksteps = a.Ktot / 32;
    const long long tilesM256 = (a.M + 255) / 256;
    const long long cus = ((force && atoi(force)) || odd32) ? 0 : fo_cu_count();
    if (d->Cout % 256 == 0 && tilesM256 * (d->Cout / 256) >= 3 * cus) {
      a.tilesM = (int)tilesM256; a.tilesN = d->Cout / 256;
      a.frameTiles = a.HWm % 256 == 0;
      const int npe = pph_pieces<256>(d, a);
      if (npe == 3) return launch_pph<256, 256, 2, 4, 3>(a, s);
      if (npe == 4) return launch_pph<256, 256, 2, 4, 4>(a, s);
      return launch_pp16<256, 256, 2, 4>(a, s);
    }
    const char* t512 = getenv("FACEOFF_BF16_TILE512");               // 0: never the 512-row tile (diagnostics)
    const long long tilesM512 = (a.M + 511) / 512;
    if (!(t512 && !atoi(t512)) && tilesM512 * (d->Cout / 128) >= 5 * cus && (cus > 0 || (t512 && atoi(t512)))) {
      a.tilesM = (int)tilesM512; a.tilesN = d->Cout / 128;
      a.frameTiles = a.HWm % 512 == 0;
      const int npe = pph_pieces<512>(d, a);
      if (npe == 5) return launch_pph<512, 128, 4, 2, 5>(a, s);     // (4 x 2 waves: 8 x 4 blocks per wave as the 256 x 256 tile; 8 x 1 spills with nine phases unrolled)
      if (npe == 6) return launch_pph<512, 128, 4, 2, 6>(a, s);
      return launch_pp16<512, 128, 8, 1>(a, s);
    }
    if (tilesM256 * (d->Cout / 128) >= 2 * cus) {
      a.tilesM = (int)tilesM256; a.tilesN = d->Cout / 128;
      a.frameTiles = a.HWm % 256 == 0;
      const int npe = pph_pieces<256>(d, a);
      if (npe == 3) return launch_pph<256, 128, 4, 2, 3>(a, s);
      if (npe == 4) return launch_pph<256, 128, 4, 2, 4>(a, s);
      return launch_pp16<256, 128, 4, 2>(a, s);
    }
  }
  FO_REQUIRE(smallc || d->Cin % 64 == 0, FO_E_SHAPE,
             "conv_bf16: Cin=%d (a multiple of 32 but not of 64) is only taken by the same-size stride-1 kernels with Cout %% 128 == 0", d->Cin);
  a.cinChunks = smallc ? 1 : d->Cin / 64;
  a.ksteps = a.Ktot / 64;
  a.frameTiles = a.HWm % BM == 0;
  const char* nodma = getenv("FACEOFF_BF16_NO_DMA");                // diagnostics: the register-staged kernel everywhere
  const bool dma = !smallc && !(flags & FO_IN_RELU) && !(nodma && atoi(nodma));
  if (d->Cout > 64) {
    a.tilesN = (d->Cout + 127) / 128;
    return dma ? launch_dma<128, 2, 2, 2, 2>(a, s) : launch<128, 2, 2, 2, 2>(a, smallc, s);
  } else if (d->Cout > 32) {
    a.tilesN = 1;
    return dma ? launch_dma<64, 2, 2, 2, 1>(a, s) : launch<64, 2, 2, 2, 1>(a, smallc, s);
  } else {
    a.tilesN = 1;
    return launch<32, 4, 1, 1, 1>(a, smallc, s);
  }
}

// VGG conv1_1 + conv1_2 (+ the 2x2 max-pool) in one launch: vgg_conv1_fused_bf16_kernel.  x8 [N][H][W][8] bf16 (ScalingLayer output), wp1 = fo_pack_conv_bf16
// of the 64 x 8 x 3 x 3 filter with taps padded to 16 (rows of 128), wp2 = fo_pack_conv_bf16 of the 64 x 64 x 3 x 3 filter; out1 (relu1_1) and pooled may be NULL.
// Returns FO_E_SHAPE for frames that are not whole 4 x 32 tiles or too few tiles to fill the chip twice (the caller then runs the two layers one by one).
int fo_vgg_conv1_fused_bf16(const void* x8, const void* wp1, const float* b1, const void* wp2, const float* b2, void* out1, void* out2, void* pooled,
                            int N, int H, int W, void* stream) {
  FO_REQUIRE(x8 && wp1 && b1 && wp2 && b2 && out2 && fo_aligned16(x8) && fo_aligned16(wp1) && fo_aligned16(wp2) && fo_aligned16(out2), FO_E_ALIGN,
             "vgg_conv1_fused: null / unaligned pointer");
  FO_REQUIRE(N > 0 && H % 4 == 0 && W % 32 == 0 && (!pooled || (H % 2 == 0)), FO_E_SHAPE, "vgg_conv1_fused: frames must be whole 4 x 32 tiles (H %% 4, W %% 32)");
  const unsigned long long px = (unsigned long long)N * H * W;
  FO_REQUIRE(px * 128 < (1ull << 31), FO_E_SHAPE, "vgg_conv1_fused: relu1_1 / relu1_2 exceed the 2 GiB buffer window (run in frame chunks)");
  Vgg1Args a;
  a.x8 = reinterpret_cast<const __bf16*>(x8); a.wp1 = reinterpret_cast<const __bf16*>(wp1); a.b1 = b1;
  a.wp2 = reinterpret_cast<const __bf16*>(wp2); a.b2 = b2;
  a.out1 = reinterpret_cast<__bf16*>(out1); a.out2 = reinterpret_cast<__bf16*>(out2); a.pooled = reinterpret_cast<__bf16*>(pooled);
  a.N = N; a.H = H; a.W = W; a.tilesX = W / 32; a.tilesY = H / 4; a.ntiles = N * a.tilesX * a.tilesY;
  a.x8Bytes = (unsigned)(px * 16); a.out1Bytes = (unsigned)(px * 128);
  const int cus = fo_cu_count();
  FO_REQUIRE(a.ntiles >= 4 * cus || getenv("FACEOFF_BF16_FORCE_HALO"), FO_E_SHAPE, "vgg_conv1_fused: fewer than 4 tiles per CU");
  constexpr int ldsBytes = 2 * 6 * 48 * 64 + 2 * 8192 + 64 * 208 + 256;
  const int grid = std::min(2 * cus, a.ntiles);
  hipStream_t s = (hipStream_t)stream;
#define FO_V1(O1_, PL_)                                                                                                              \
  do {                                                                                                                               \
    static fo_lds_once once;                                                                                                         \
    if (!fo_lds_optin(once, reinterpret_cast<const void*>(vgg_conv1_fused_bf16_kernel<O1_, PL_>), ldsBytes, "vgg_conv1_fused")) return FO_E_HIP; \
    FO_NOTE_T("vgg_conv1_fused_bf16_kernel", O1_, PL_);                                                                              \
    hipLaunchKernelGGL((vgg_conv1_fused_bf16_kernel<O1_, PL_>), dim3(grid), dim3(256), ldsBytes, s, a);                               \
  } while (0)
  if (out1 && pooled) FO_V1(true, true);
  else if (out1) FO_V1(true, false);
  else if (pooled) FO_V1(false, true);
  else FO_V1(false, false);
#undef FO_V1
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_conv_igemm_bf16(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, void* out,
                       void* stream) {
  FO_REQUIRE(d->KD == 1 && d->padD == 0, FO_E_SHAPE, "conv_igemm_bf16: 2-D only (fo_conv_bf16 takes depth taps)");
  FO_REQUIRE(!(d->flags & ~(FO_BIAS | FO_MASK | FO_OUT_RELU)), FO_E_SHAPE, "conv_igemm_bf16: flags other than BIAS|MASK|OUT_RELU");
  fo_conv_desc e = *d;
  e.T = 1;
  return conv_bf16_impl(&e, in, wp, bias, mask, nullptr, out, stream);
}

int fo_conv_bf16(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, const void* add, void* out,
                 void* stream) {
  return conv_bf16_impl(d, in, wp, bias, mask, add, out, stream);
}

int fo_conv_igemm_bf16_pool(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, void* out, void* pooled, int ldPooled,
                            void* stream) {
  FO_REQUIRE(d->KD == 1 && d->padD == 0 && !(d->flags & ~(FO_BIAS | FO_OUT_RELU)), FO_E_SHAPE, "conv_igemm_bf16_pool: 2-D, flags BIAS|OUT_RELU only");
  FO_REQUIRE(pooled && fo_aligned16(pooled) && ldPooled % 4 == 0 && ldPooled >= d->Cout && d->Hout % 2 == 0 && d->Wout % 2 == 0, FO_E_SHAPE,
             "conv_igemm_bf16_pool: pooled output [N][H/2][W/2][ldPooled >= Cout]");
  fo_conv_desc e = *d;
  e.T = 1;
  return conv_bf16_impl(&e, in, wp, bias, nullptr, nullptr, out, stream, pooled, ldPooled);
}

int fo_conv_bf16_ex(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, const void* add, void* out,
                    const fo_conv_extra* ex, void* stream) {
  FO_REQUIRE(ex, FO_E_SHAPE, "conv_bf16_ex: ex == NULL (use fo_conv_bf16)");
  if (ex->pooled) {
    FO_REQUIRE(d->KD == 1 && d->padD == 0 && !(d->flags & ~(FO_BIAS | FO_OUT_RELU)) && !add, FO_E_SHAPE, "conv_bf16_ex: pooled: 2-D, flags BIAS|OUT_RELU only");
    FO_REQUIRE(fo_aligned16(ex->pooled) && ex->ldPooled % 4 == 0 && ex->ldPooled >= d->Cout && d->Hout % 2 == 0 && d->Wout % 2 == 0, FO_E_SHAPE,
               "conv_bf16_ex: pooled output [N][H/2][W/2][ldPooled >= Cout]");
  }
  return conv_bf16_impl(d, in, wp, bias, mask, add, out, stream, ex->pooled, ex->ldPooled, ex->pool_idx, ex->mask_bits, ex->out_bits, ex->pooled_bits);
}

int fo_conv_igemm_bf16_pool_idx(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, void* out, void* pooled, int ldPooled,
                                void* idx, void* stream) {
  FO_REQUIRE(d->KD == 1 && d->padD == 0 && !(d->flags & ~(FO_BIAS | FO_OUT_RELU)), FO_E_SHAPE, "conv_igemm_bf16_pool: 2-D, flags BIAS|OUT_RELU only");
  FO_REQUIRE(pooled && fo_aligned16(pooled) && ldPooled % 4 == 0 && ldPooled >= d->Cout && d->Hout % 2 == 0 && d->Wout % 2 == 0, FO_E_SHAPE,
             "conv_igemm_bf16_pool: pooled output [N][H/2][W/2][ldPooled >= Cout]");
  fo_conv_desc e = *d;
  e.T = 1;
  return conv_bf16_impl(&e, in, wp, bias, nullptr, nullptr, out, stream, pooled, ldPooled, idx);
}
}
