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
  void* out;
  int M, HWm, tilesM, tilesN;
  int cinChunks;   // Cin/64 ; SMALLC: unused
  int Ktot;        // elements per filter row (multiple of 64)
  int ksteps;      // Ktot/64
  unsigned inBytes, wpBytes;
};

#ifndef FO_ABLATE_H   // diagnostic builds (tools/ablate_bf16.sh): bit 0 drop the loop's global loads, 1 its LDS stores,
#define FO_ABLATE_H 0 // 2 its fragment reads (results are wrong, only the timing is of interest)
#endif
constexpr int BM = 128;
constexpr int ROWB = 144;            // LDS row: 128 B of data + 16 B pad
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ u32x4 bufload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN, bool SMALLC>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const ConvArgsH a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(WAVES_M * TM * 32 == BM && WAVES_N * TN * 32 == BN, "tile");
  constexpr int BROWS = BN / 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * ROWB];
  unsigned char* As0 = lds;
  unsigned char* Bs0 = lds + 2 * BM * ROWB;

  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;
  const int ntaps = d.KH * d.KW;

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
    pbase[i] = (n * d.Hin + py[i]) * d.Win + px[i];      // pixel index of tap (0,0)
    rowoff[i] = pbase[i] * d.ldIn * 2 + lcolB;
    unsigned mk = 0;
    if (!SMALLC) {
      for (int tp = 0; tp < ntaps; ++tp) {
        const int kh = tp / d.KW, kw = tp - kh * d.KW;
        const bool ok = pv[i] & ((unsigned)(py[i] + kh) < (unsigned)d.Hin) & ((unsigned)(px[i] + kw) < (unsigned)d.Win);
        mk |= (ok ? 1u : 0u) << tp;
      }
    }
    tapmask[i] = mk;
  }

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  const unsigned wrow = (unsigned)(((size_t)(tile_n * BN + lrow) * a.Ktot) * 2 + lcolB);
  const unsigned wstride32 = (unsigned)((size_t)32 * a.Ktot * 2);

  // incremental (tap, chunk) walk of the next step to load (wave-uniform scalars)
  int ld_step = 0, ld_tap = 0, ld_kh = 0, ld_kw = 0, ld_chunk = 0;
  // SMALLC: this thread's tap of the step being loaded
  int sc_off = 0, sc_kh = 0, sc_kw = 0;
  bool sc_ok = false;

  u32x4 ra[2][4], rb[2][BROWS];

  auto load_step = [&](int slot) {
    if (SMALLC) {
      const int tap = ld_step * 8 + (tid & 7);
      sc_kh = tap / d.KW;
      sc_kw = tap - sc_kh * d.KW;
      sc_ok = tap < ntaps;
      sc_off = (sc_kh * d.Win + sc_kw) * d.ldIn * 2;
    }
    const int stepoff = ((ld_kh * d.Win + ld_kw) * d.ldIn + ld_chunk * 64) * 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (SMALLC) {
        const bool ok = sc_ok & pv[s] & ((unsigned)(py[s] + sc_kh) < (unsigned)d.Hin) & ((unsigned)(px[s] + sc_kw) < (unsigned)d.Win);
        ra[slot][s] = bufload16(rin, ok ? (unsigned)(pbase[s] * d.ldIn * 2 + sc_off) : OOB);
      } else {
        const unsigned pad = (((tapmask[s] >> ld_tap) & 1u) - 1u) & OOB;     // padding tap -> beyond the descriptor -> zeros
        ra[slot][s] = bufload16(rin, (unsigned)(rowoff[s] + stepoff) | pad);
      }
      if (s < BROWS) rb[slot][s < BROWS ? s : 0] = bufload16(rwp, ld_step < a.ksteps ? wrow + s * wstride32 + ld_step * 128 : OOB);
    }
    ++ld_step;
    if (!SMALLC && ++ld_chunk == a.cinChunks) {
      ld_chunk = 0;
      ++ld_tap;
      if (++ld_kw == d.KW) { ld_kw = 0; ++ld_kh; }
    }
  };
  auto store_step = [&](int slot, int buf, int s_lo = 0, int s_hi = 4) {
    unsigned char* As = As0 + buf * BM * ROWB;
    unsigned char* Bs = Bs0 + buf * BN * ROWB;
#pragma unroll
    for (int s = s_lo; s < s_hi; ++s) {
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
  const int nsteps = a.ksteps;
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
  const int flags = d.flags;
  constexpr int C8 = BN / 8;
  constexpr int RPP = 256 / C8;
  const int c8 = tid % C8;
  const int co = tile_n * BN + c8 * 8;
  if (co >= d.Cout) return;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = ((flags & FO_BIAS) && co + e < d.Cout) ? a.bias[co + e] : 0.f;
  const bool identity_pix = (d.ostride == 1) & (d.Hm == d.Hout) & (d.Wm == d.Wout);
  __bf16* out = reinterpret_cast<__bf16*>(a.out);
  const __bf16* mask = reinterpret_cast<const __bf16*>(a.mask);
  // rows in batches: all mask loads of a batch are in flight before the first is used (see conv_igemm.hip store_tile)
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
        opix[r] = ((size_t)n * d.Hout + (y * d.ostride + d.ophH)) * d.Wout + (x * d.ostride + d.ophW);
      }
    }
    bf16x8 mk[R];
    if (flags & FO_MASK) {
#pragma unroll
      for (int r = 0; r < R; ++r) mk[r] = *reinterpret_cast<const bf16x8*>(mask + opix[r] * d.ldMask + co);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* crow = Cs + (r0 + (b + r) * RPP) * C_LD + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(crow);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(crow + 4);
      float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
      if (flags & FO_MASK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)mk[r][e] > 0.f ? v[e] : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)((flags & FO_OUT_RELU) ? fmaxf(v[e], 0.f) : v[e]);
      if (ok[r]) *reinterpret_cast<bf16x8*>(out + opix[r] * d.ldOut + co) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ big tiles
// 256 x BN tile, ONE workgroup per CU with 8 waves (two per SIMD), each wave (TM x TN) 32x32 accumulators (128 x 64 for the
// 256 x 256 tile; a four-wave 128 x 128-per-wave form measured 15 % slower: one wave per SIMD leaves nobody to cover its
// fragment reads and barrier waits).  Why: at the bf16 rate a 128 x 128 tile with 2 x 2 accumulators per wave asks for ~39 TB/s from L2 (32 KB
// per 2.1 MFLOP K-step) and ~190 B/clk from LDS (every fragment feeds only 2 MFMAs) -- both beyond the hardware.  256 x 256
// halves the L2 bytes per FLOP, 4 x 2 accumulators per wave cut the LDS reads per MFMA by a third, and a K-step becomes
// 64 MFMAs per wave x 2 waves per SIMD = ~4000 cycles -- longer than an HBM miss -- so a one-deep prefetch (loads of step
// n+1 issued before step n's MFMAs, stored to the other LDS buffer behind them) covers the memory latency, while the two
// waves of a SIMD cover each other's fragment reads and barrier waits.  Epilogue: each wave transposes its accumulators 32
// rows at a time through its own LDS patch (wave-local) so that bias / ReLU / mask / rounding / stores are 16 B per lane.
// Stride-1, same-size convolutions with Cin % 64 == 0 (every LPIPS layer but the RGB one).
template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 1) void conv_bf16_big_kernel(const ConvArgsH a) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;              // threads
  constexpr int BMB = WAVES_M * TM * 32;
  static_assert(WAVES_N * TN * 32 == BN && BMB == 256, "tile");
  constexpr int RP = NT / 8;                              // rows per loader pass (8 threads x 16 B per 128-B row)
  constexpr int AR = BMB / RP, BR = BN / RP;
  static_assert(AR >= 1 && BR >= 1, "loader passes");
  constexpr int C_LD = TN * 32 + 4;                       // floats per row of a wave's epilogue patch
  static_assert(WAVES_M * WAVES_N * 32 * C_LD * 4 <= 2 * (BMB + BN) * ROWB, "epilogue patches must fit the staging LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* As0 = lds;
  unsigned char* Bs0 = lds + 2 * BMB * ROWB;

  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;
  const int ntaps = d.KH * d.KW;

  const int lrow = tid >> 3;
  const int lcolB = (tid & 7) * 16;
  int rowoff[AR];
  unsigned tapmask[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = tile_m * BMB + lrow + RP * i;
    const bool pv = m < a.M;
    const int mm = pv ? m : 0;
    const int n = mm / a.HWm;
    const int rem = mm - n * a.HWm;
    const int y = rem / d.Wm;
    const int x = rem - y * d.Wm;
    const int py = y - d.padH, px = x - d.padW;
    rowoff[i] = ((n * d.Hin + py) * d.Win + px) * d.ldIn * 2 + lcolB;
    unsigned mk = 0;
    for (int tp = 0; tp < ntaps; ++tp) {
      const int kh = tp / d.KW, kw = tp - kh * d.KW;
      const bool ok = pv & ((unsigned)(py + kh) < (unsigned)d.Hin) & ((unsigned)(px + kw) < (unsigned)d.Win);
      mk |= (ok ? 1u : 0u) << tp;
    }
    tapmask[i] = mk;
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  const unsigned wrow = (unsigned)(((size_t)(tile_n * BN + lrow) * a.Ktot) * 2 + lcolB);
  const unsigned wstrideRP = (unsigned)((size_t)RP * a.Ktot * 2);

  int ld_step = 0, ld_tap = 0, ld_kh = 0, ld_kw = 0, ld_chunk = 0;
  u32x4 ra[AR], rb[BR];
  auto load_step = [&]() {
    const int stepoff = ((ld_kh * d.Win + ld_kw) * d.ldIn + ld_chunk * 64) * 2;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const unsigned pad = (((tapmask[i] >> ld_tap) & 1u) - 1u) & OOB;
      ra[i] = bufload16(rin, (unsigned)(rowoff[i] + stepoff) | pad);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) rb[i] = bufload16(rwp, ld_step < a.ksteps ? wrow + i * wstrideRP + ld_step * 128 : OOB);
    ++ld_step;
    if (++ld_chunk == a.cinChunks) {
      ld_chunk = 0;
      ++ld_tap;
      if (++ld_kw == d.KW) { ld_kw = 0; ++ld_kh; }
    }
  };
  auto store_rows = [&](int buf, int lo, int hi) {       // passes lo..hi-1 of the combined (A then B) row-pass list
    unsigned char* As = As0 + buf * BMB * ROWB;
    unsigned char* Bs = Bs0 + buf * BN * ROWB;
#pragma unroll
    for (int i = 0; i < AR + BR; ++i) {
      if (i < lo || i >= hi) continue;
      if (i < AR) *reinterpret_cast<u32x4*>(As + (lrow + RP * i) * ROWB + lcolB) = ra[i];
      else *reinterpret_cast<u32x4*>(Bs + (lrow + RP * (i - AR)) * ROWB + lcolB) = rb[i - AR];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_step();
  store_rows(0, 0, AR + BR);
  __syncthreads();
  const int nsteps = a.ksteps;
  int cur = 0;
  for (int step = 0; step < nsteps; ++step) {
    const unsigned char* As = As0 + cur * BMB * ROWB + (wm * TM * 32 + l31) * ROWB + half * 16;
    const unsigned char* Bs = Bs0 + cur * BN * ROWB + (wn * TN * 32 + l31) * ROWB + half * 16;
    load_step();                                          // step + 1 (past the end: zeros, never used)
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROWB);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROWB);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s < 3) {                                        // fragments of the next 16-deep slice land under this slice's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROWB + (s + 1) * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROWB + (s + 1) * 32);
      }
      // the LDS stores of step + 1 ride behind the last two slices' MFMAs
      if (s == 2) store_rows(cur ^ 1, 0, (AR + BR) / 2);
      if (s == 3) store_rows(cur ^ 1, (AR + BR) / 2, AR + BR);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: 32 rows of this wave's tile at a time through its own LDS patch
  float* Cs = reinterpret_cast<float*>(lds) + wave * 32 * C_LD;
  const int flags = d.flags;
  constexpr int C8 = TN * 4;                              // 8-channel groups per patch row
  constexpr int RPP = 64 / C8;                            // patch rows per pass of the wave
  const int c8 = lane % C8, r0 = lane / C8;
  const int co = tile_n * BN + wn * TN * 32 + c8 * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = ((flags & FO_BIAS) && co + e < d.Cout) ? a.bias[co + e] : 0.f;
  __bf16* out = reinterpret_cast<__bf16*>(a.out);
  const __bf16* mask = reinterpret_cast<const __bf16*>(a.mask);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * half) * C_LD + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    const int mbase = tile_m * BMB + (wm * TM + i) * 32;
#pragma unroll
    for (int p = 0; p < 32 / RPP; ++p) {
      const int row = p * RPP + r0;
      const int m = mbase + row;
      const float* crow = Cs + row * C_LD + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(crow);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(crow + 4);
      if (m >= a.M || co >= d.Cout) continue;
      float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
      if (flags & FO_MASK) {
        const bf16x8 mk = *reinterpret_cast<const bf16x8*>(mask + (size_t)m * d.ldMask + co);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)((flags & FO_OUT_RELU) ? fmaxf(v[e], 0.f) : v[e]);
      *reinterpret_cast<bf16x8*>(out + (size_t)m * d.ldOut + co) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ big tiles, LDS-DMA staging
// The 256 x 256 tile again, with the operands going global -> LDS directly (buffer_load_dwordx4 ... lds): no staging VGPRs,
// no ds_write pass.  An LDS-DMA writes 64 lanes x 16 B LINEARLY from a wave-uniform base, so rows cannot be padded; the
// bank-conflict fix moves to a swizzle instead: LDS rows are 128 B (one 64-deep K-step of one GEMM row = eight 16-B chunks)
// and chunk c of row r lives at chunk position c ^ ((r >> 1) & 7).  The swizzle is applied on the SOURCE side (lane l of a
// DMA instruction = row l / 8, LDS chunk position l % 8 fetches source chunk (l % 8) ^ ((row >> 1) & 7)) and again on the
// fragment read (same involution), never on the LDS destination.  With it the 16 lanes of a ds_read_b128 group (rows r..r+15
// at one logical chunk) cover all 16 sixteen-byte bank groups.  Two LDS stages: the DMAs of step n+1 are issued at the top
// of step n (the stage they fill was last read in step n-1, behind that step's barrier) and retired by the vmcnt(0) that
// __syncthreads() implies at the bottom -- a K-step is ~4000 cycles for the two waves of a SIMD, longer than an HBM miss.
template <int TM, int TN>
__global__ __launch_bounds__(512, 1) void conv_bf16_dma_kernel(const ConvArgsH a) {
  constexpr int BMB = 256, BNB = 256, WAVES_N = 4;
  static_assert(TM == 4 && TN == 2, "2 x 4 waves of 128 x 64");
  constexpr int STAGE = (BMB + BNB) * 128;                // bytes per stage
  constexpr int C_LD = TN * 32 + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % a.tilesN;
  const int tile_m = logical / a.tilesN;
  const int ntaps = d.KH * d.KW;

  // ---- DMA roles: instruction i (0..3) of wave w stages tile rows (i*8 + w)*8 .. +7, lane = (row % 8, chunk position)
  const int drow = lane >> 3, dpos = lane & 7;
  int rowoffA[4];
  unsigned tapmaskA[4], woffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 8 + wave) * 8 + drow;            // row of the tile (A: pixel, B: output channel)
    const int chunk = dpos ^ ((row >> 1) & 7);            // source chunk this lane fetches
    const int m = tile_m * BMB + row;
    const bool pv = m < a.M;
    const int mm = pv ? m : 0;
    const int n = mm / a.HWm;
    const int rem = mm - n * a.HWm;
    const int y = rem / d.Wm;
    const int x = rem - y * d.Wm;
    const int py = y - d.padH, px = x - d.padW;
    rowoffA[i] = ((n * d.Hin + py) * d.Win + px) * d.ldIn * 2 + chunk * 16;
    unsigned mk = 0;
    for (int tp = 0; tp < ntaps; ++tp) {
      const int kh = tp / d.KW, kw = tp - kh * d.KW;
      const bool ok = pv & ((unsigned)(py + kh) < (unsigned)d.Hin) & ((unsigned)(px + kw) < (unsigned)d.Win);
      mk |= (ok ? 1u : 0u) << tp;
    }
    tapmaskA[i] = mk;
    woffB[i] = (unsigned)(((size_t)(tile_n * BNB + row) * a.Ktot) * 2 + chunk * 16);
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, a.wpBytes, 0x00020000);
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  lds_byte* const lds3 = (lds_byte*)lds;

  int ld_step = 0, ld_tap = 0, ld_kh = 0, ld_kw = 0, ld_chunk = 0;
  auto dma_step = [&](int stage) {
    const int stepoff = ((ld_kh * d.Win + ld_kw) * d.ldIn + ld_chunk * 64) * 2;
    lds_byte* const sa = lds3 + stage * STAGE;
    lds_byte* const sb = sa + BMB * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned pad = (((tapmaskA[i] >> ld_tap) & 1u) - 1u) & OOB;       // padding tap / row past M -> zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(sa + (i * 8 + wave) * 1024), 16,
                                               (unsigned)(rowoffA[i] + stepoff) | pad, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rwp, (__attribute__((address_space(3))) void*)(sb + (i * 8 + wave) * 1024), 16,
                                               ld_step < a.ksteps ? woffB[i] + ld_step * 128 : OOB, 0, 0, 0);
    ++ld_step;
    if (++ld_chunk == a.cinChunks) {
      ld_chunk = 0;
      ++ld_tap;
      if (++ld_kw == d.KW) { ld_kw = 0; ++ld_kh; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: row = block row + l31 (blocks are multiples of 32: (row >> 1) & 7 = (l31 >> 1) & 7), logical chunk 2s + half
  const int xr = (l31 >> 1) & 7;
  int swz[4];
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) swz[sidx] = ((2 * sidx + half) ^ xr) * 16;

  dma_step(0);
  __syncthreads();
  const int nsteps = a.ksteps;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    dma_step(cur ^ 1);                                    // step + 1 (past the end: zeros, never read)
    const unsigned char* As = lds + cur * STAGE + (wm * TM * 32 + l31) * 128;
    const unsigned char* Bs = lds + cur * STAGE + BMB * 128 + (wn * TN * 32 + l31) * 128;
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * 128 + swz[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * 128 + swz[0]);
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      __builtin_amdgcn_sched_barrier(0);
      if (sidx < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[(sidx + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * 128 + swz[sidx + 1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[(sidx + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * 128 + swz[sidx + 1]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sidx & 1][i], fb[sidx & 1][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                      // (implies vmcnt(0): this wave's DMAs of step + 1 have landed)
  }

  // ---- epilogue: as conv_bf16_big_kernel
  float* Cs = reinterpret_cast<float*>(lds) + wave * 32 * C_LD;
  const int flags = d.flags;
  constexpr int C8 = TN * 4, RPP = 64 / C8;
  const int c8 = lane % C8, r0 = lane / C8;
  const int co = tile_n * BNB + wn * TN * 32 + c8 * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = ((flags & FO_BIAS) && co + e < d.Cout) ? a.bias[co + e] : 0.f;
  __bf16* out = reinterpret_cast<__bf16*>(a.out);
  const __bf16* mask = reinterpret_cast<const __bf16*>(a.mask);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * half) * C_LD + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    const int mbase = tile_m * BMB + (wm * TM + i) * 32;
#pragma unroll
    for (int p = 0; p < 32 / RPP; ++p) {
      const int row = p * RPP + r0;
      const int m = mbase + row;
      const float* crow = Cs + row * C_LD + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(crow);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(crow + 4);
      if (m >= a.M || co >= d.Cout) continue;
      float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
      if (flags & FO_MASK) {
        const bf16x8 mk = *reinterpret_cast<const bf16x8*>(mask + (size_t)m * d.ldMask + co);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)((flags & FO_OUT_RELU) ? fmaxf(v[e], 0.f) : v[e]);
      *reinterpret_cast<bf16x8*>(out + (size_t)m * d.ldOut + co) = o;
    }
  }
}

int launch_dma(const ConvArgsH& a, hipStream_t s) {
  constexpr int ldsBytes = 2 * (256 + 256) * 128;
  static bool attr_set = false;
  auto kern = conv_bf16_dma_kernel<4, 2>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes) != hipSuccess) {
      fo_set_error("conv_bf16: cannot reserve %d bytes of LDS", ldsBytes);
      return FO_E_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN), dim3(512), ldsBytes, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
int launch_big(const ConvArgsH& a, hipStream_t s) {
  constexpr int ldsBytes = 2 * (256 + BN) * ROWB;
  static bool attr_set = false;                            // > 64 KB of dynamic LDS needs the opt-in once per process
  auto kern = conv_bf16_big_kernel<BN, WAVES_M, WAVES_N, TM, TN>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes) != hipSuccess) {
      fo_set_error("conv_bf16: cannot reserve %d bytes of LDS", ldsBytes);
      return FO_E_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tilesM * a.tilesN), dim3(64 * WAVES_M * WAVES_N), ldsBytes, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
int launch(const ConvArgsH& a, bool smallc, hipStream_t s) {
  const int grid = a.tilesM * a.tilesN;
  if (smallc)
    hipLaunchKernelGGL((conv_bf16_kernel<BN, WAVES_M, WAVES_N, TM, TN, true>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_bf16_kernel<BN, WAVES_M, WAVES_N, TM, TN, false>), dim3(grid), dim3(256), 0, s, a);
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

int fo_conv_igemm_bf16(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, const void* mask, void* out,
                       void* stream) {
  ConvArgsH a;
  a.d = *d;
  a.in = in; a.wp = wp; a.bias = bias; a.mask = mask; a.out = out;
  FO_REQUIRE(d->N > 0 && d->KD == 1 && d->padD == 0, FO_E_SHAPE, "conv_bf16: 2-D only (KD == 1)");
  FO_REQUIRE(!(d->flags & ~(FO_BIAS | FO_MASK | FO_OUT_RELU)), FO_E_SHAPE, "conv_bf16: flags other than BIAS|MASK|OUT_RELU");
  FO_REQUIRE(d->ldIn % 8 == 0 && d->ldOut % 8 == 0 && fo_aligned16(in) && fo_aligned16(wp) && fo_aligned16(out), FO_E_ALIGN,
             "conv_bf16: 16-byte alignment (ld %% 8 == 0)");
  FO_REQUIRE(d->ldOut >= (d->Cout + 7) / 8 * 8, FO_E_ALIGN, "conv_bf16: ldOut must hold Cout rounded up to 8");
  FO_REQUIRE(!(d->flags & FO_MASK) || (mask && fo_aligned16(mask) && d->ldMask % 8 == 0), FO_E_ALIGN, "conv_bf16: mask alignment");
  FO_REQUIRE(!(d->flags & FO_BIAS) || bias, FO_E_SHAPE, "conv_bf16: FO_BIAS without bias");
  const int taps = d->KH * d->KW;
  FO_REQUIRE(taps >= 1 && taps <= 31, FO_E_SHAPE, "conv_bf16: at most 31 taps (got %d)", taps);
  const bool smallc = d->Cin < 64;
  if (smallc) {
    FO_REQUIRE(d->Cin == 8 && d->ldIn == 8, FO_E_SHAPE, "conv_bf16: small Cin must be 8 with 16-byte pixels (got %d)", d->Cin);
    a.Ktot = (taps + 7) / 8 * 64;
    a.cinChunks = 1;
  } else {
    FO_REQUIRE(d->Cin % 64 == 0, FO_E_SHAPE, "conv_bf16: Cin=%d must be a multiple of 64 (or 8)", d->Cin);
    a.Ktot = taps * d->Cin;
    a.cinChunks = d->Cin / 64;
  }
  a.ksteps = a.Ktot / 64;
  a.HWm = d->Hm * d->Wm;
  const long long M = (long long)d->N * a.HWm;
  FO_REQUIRE(M > 0 && M < (1ll << 31), FO_E_SHAPE, "conv_bf16: M out of range");
  a.M = (int)M;
  a.tilesM = (a.M + BM - 1) / BM;
  const unsigned long long inBytes = (((unsigned long long)d->N * d->Hin * d->Win - 1) * d->ldIn + d->Cin) * 2ull;
  const int opad = d->Cout > 64 ? (d->Cout + 127) / 128 * 128 : (d->Cout > 32 ? 64 : 32);
  const unsigned long long wpBytes = (unsigned long long)opad * a.Ktot * 2ull;
  FO_REQUIRE(inBytes < (1ull << 31) && wpBytes < (1ull << 31), FO_E_SHAPE, "conv_bf16: tensor exceeds the 2 GiB buffer-descriptor window");
  a.inBytes = (unsigned)inBytes;
  a.wpBytes = (unsigned)wpBytes;
  hipStream_t s = (hipStream_t)stream;
  // big tiles (one workgroup per CU) where the launch still fills the chip for a few rounds: stride-1 same-size convs
  const bool same = d->stride == 1 && d->ostride == 1 && d->Hm == d->Hout && d->Wm == d->Wout && d->Hin == d->Hm && d->Win == d->Wm;
  const char* nobig = getenv("FACEOFF_BF16_SMALL_TILES");
  if (!smallc && same && !(nobig && atoi(nobig)) && d->Cout >= 64 && d->Cout % 64 == 0) {
    const int tilesM256 = (a.M + 255) / 256;
    const char* force = getenv("FACEOFF_BF16_BIG_TILES");          // tests: big tiles at any size
    const int cus = (force && atoi(force)) ? 0 : fo_cu_count();
    // measured at the C3 shapes (tools/bench_bf16.py): 256-column tiles win everywhere they apply (conv3_x / conv4_x forward
    // 820-860 -> 1000-1050 TFLOP/s, masked data gradients 775-845 -> 860-990); the 128- and 64-column variants lose to
    // the two-workgroups-per-CU kernel on their short-K layers (K = 576 / 1152: the unoverlapped prologue and epilogue of a
    // lone workgroup) and are only taken when forced (tests)
    const bool forced = force && atoi(force);
    if (d->Cout % 256 == 0 && (long long)tilesM256 * (d->Cout / 256) >= 3ll * cus) {
      a.tilesM = tilesM256; a.tilesN = d->Cout / 256;
      const char* nodma = getenv("FACEOFF_BF16_NO_DMA");              // diagnostics: register-staged form of the same tile
      if (!(nodma && atoi(nodma))) return launch_dma(a, s);
      return launch_big<256, 2, 4, 4, 2>(a, s);
    }
    if (forced && d->Cout % 128 == 0) {
      a.tilesM = tilesM256; a.tilesN = d->Cout / 128;
      return launch_big<128, 4, 2, 2, 2>(a, s);
    }
    if (forced && d->Cout == 64) {
      a.tilesM = tilesM256; a.tilesN = 1;
      return launch_big<64, 8, 1, 1, 2>(a, s);
    }
  }
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
}
