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
//    the two lane halves), so one ds_read_b128 feeds 4 MFMAs per tile; fragments are double
//    buffered in registers so each LDS read has a full 16-MFMA group to land under;
//  * 128 x BN block tile, 4 waves (one per SIMD) x up to 2x2 32x32 accumulators, LDS double
//    buffered, next K-step prefetched into registers while the current one is on the matrix pipe;
//  * loads are bounds-checked buffer loads: padding taps use an out-of-range offset and read
//    zeros -- no branch, no select after the load, so the loads stay in flight under the MFMAs;
//    per-row tap validity is a bitmask built once per tile, the K-step -> (tap, chunk) walk is
//    incremental scalar arithmetic (no divisions in the loop);
//  * temporal taps that fall entirely into clip padding are skipped per tile (T=5: 2/15 of Conv3d);
//  * blockIdx is remapped so that each XCD's L2 sees a contiguous run of tiles (shared halos).
#include <stdlib.h>
#include <type_traits>
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
  int wShift;        // log2(Wm) if Wm is a power of two, else -1
  int hwShift;       // log2(Hm*Wm) likewise
  int kwShift;       // log2(KW) likewise
  int scRow;         // SMALLC: a K-step is one filter row (KW * Cin == 32)
  int bankFrames;    // > 0: the filter bank is chosen by frame index, bank = frame / bankFrames (Winograd plane stacks)
  unsigned bankBytes;  // bytes between consecutive filter banks
  int margin;        // bytes the input descriptor starts below `in`, so that per-row base offsets are never negative
  unsigned inBytes;  // addressable extent behind `in` (buffer descriptor bound)
  unsigned wpBytes;
  // fused ResBlock tail (FUSE instantiation): out2 = [relu]( h * wp2^T + bias2 + add ), h = this conv's 32-channel result
  const float* wp2;   // [128][32] packed 1x1 filter
  const float* bias2;
  float* out2;
  int ldOut2;
  int relu2;
};

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_LD = 36;  // padded row (floats)
constexpr unsigned OOB = 0x80000000u;
#ifdef FO_STAMP   // diagnostic build only (tools/): per-step s_memtime of workgroup 0 / wave 0
__device__ unsigned long long fo_stamps[4096];
#define FO_STAMP_AT(i)                                                                            \
  if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 4096) {                                        \
    unsigned long long t_;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    fo_stamps[(i)] = t_;                                                                          \
  }
#else
#define FO_STAMP_AT(i)
#endif
#ifndef FO_ABLATE   // diagnostic builds (tools/ablate_igemm.sh): bit 0 drop the loop's global loads, 1 its LDS stores,
#define FO_ABLATE 0 // 2 its fragment reads, 3 its barrier -- results are wrong, only the timing is of interest
#endif

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// m -> (frame, y, x) of the GEMM-M grid; shifts when the grid is a power of two (every C2 shape), divisions otherwise
__device__ __forceinline__ void decode_pixel(const ConvArgs& a, int m, int& n, int& y, int& x) {
  int rem;
  if (a.hwShift >= 0) { n = m >> a.hwShift; rem = m & (a.HWm - 1); }
  else { n = m / a.HWm; rem = m - n * a.HWm; }
  if (a.wShift >= 0) { y = rem >> a.wShift; x = rem & (a.d.Wm - 1); }
  else { y = rem / a.d.Wm; x = rem - y * a.d.Wm; }
}

// Rows of the LDS-transposed C tile -> global memory: bias, ReLU mask, residual, ReLU, 16 B per lane.  A thread's rows
// go in batches of up to 8: the mask / residual loads of a whole batch are issued before the first one is used (a plain
// row loop costs one L2 round trip per operand per row, which is what the short-K launches -- 1x1 convs, the dgrads of
// the 32-channel ResBlock halves -- would then spend most of their time on).  Loads of rows past M are clamped to the
// last row instead of predicated (a predicated load makes hipcc branch and drain vmcnt per row); only stores are guarded.
template <int BN>
__device__ __forceinline__ void store_tile(const ConvArgs& a, const float* Cs, int tile_m, int tile_n, int tid) {
  const fo_conv_desc& d = a.d;
  constexpr int C_LD = BN + 4, C4 = BN / 4, RPP = 256 / C4, ROWS = BM / RPP, R = ROWS < 8 ? ROWS : 8;
  const int flags = d.flags;
  const int c4 = tid % C4, r0 = tid / C4;
  const int co = tile_n * BN + c4 * 4;
  if (co >= d.Cout) return;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (flags & FO_BIAS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (co + e < d.Cout) ? a.bias[co + e] : 0.f;
  }
  const bool identity_pix = (d.ostride == 1) & (d.Hm == d.Hout) & (d.Wm == d.Wout);
  // identity pixel map (every launch but the sub-pixel phases of a transposed conv): addresses are a uniform 64-bit
  // tile base + a 32-bit per-thread offset, so a row costs one VALU add instead of 64-bit multiplies per operand
  auto rows = [&](auto ident_c) {
    constexpr bool IDENT = decltype(ident_c)::value;
    const size_t tbase = IDENT ? (size_t)tile_m * BM : 0;
    const float* maskb = a.mask + tbase * d.ldMask;
    const float* addb = a.add + tbase * d.ldAdd;
    float* outb = a.out + tbase * d.ldOut;
    const int mrem = a.M - 1 - tile_m * BM;          // last valid row of this tile
#pragma unroll
    for (int b = 0; b < ROWS; b += R) {
      size_t opix[R];
      bool ok[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = r0 + (b + r) * RPP;
        ok[r] = row <= mrem;
        if (IDENT) {
          opix[r] = (unsigned)min(row, mrem);
        } else {
          const int m = min(tile_m * BM + row, a.M - 1);
          int n, y, x;
          decode_pixel(a, m, n, y, x);
          opix[r] = ((size_t)n * d.Hout + (y * d.ostride + d.ophH)) * d.Wout + (x * d.ostride + d.ophW);
        }
      }
      auto off = [&](int r, int ld) -> size_t { return IDENT ? (size_t)(unsigned)((int)opix[r] * ld + co) : opix[r] * ld + co; };
      f32x4 mk[R], ad[R];
      if (flags & FO_MASK) {
#pragma unroll
        for (int r = 0; r < R; ++r) mk[r] = *reinterpret_cast<const f32x4*>(maskb + off(r, d.ldMask));
      }
      if (flags & FO_ADD) {
#pragma unroll
        for (int r = 0; r < R; ++r) ad[r] = *reinterpret_cast<const f32x4*>(addb + off(r, d.ldAdd));
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + (b + r) * RPP) * C_LD + c4 * 4) + bv;
        if (flags & FO_MASK) {
          v.x = mk[r].x > 0.f ? v.x : 0.f; v.y = mk[r].y > 0.f ? v.y : 0.f; v.z = mk[r].z > 0.f ? v.z : 0.f; v.w = mk[r].w > 0.f ? v.w : 0.f;
        }
        if (flags & FO_ADD) v += ad[r];
        if (flags & FO_OUT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (ok[r]) *reinterpret_cast<f32x4*>(outb + off(r, d.ldOut)) = v;
      }
    }
  };
  if (identity_pix) rows(std::true_type{});
  else rows(std::false_type{});
}

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN, bool SMALLC, bool INRELU, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  static_assert(!FUSE || (BN == 32 && WAVES_M == 4 && TM == 1 && TN == 1), "the fused ResBlock tail is built on the 32-column tile");
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

  const int khw = d.KH * d.KW;
  const int ntaps = d.KD * khw;

  // ---- loader coordinates: thread covers rows lrow + 32*i, 16 bytes at column lcol
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;
  int rowoff[4];        // byte offset of (pixel at tap (padD,0,0)... see below) + lcol, may be negative
  unsigned tapmask[4];  // bit t = tap t reads inside the image for this row
  int py[4], px[4], pbase[4];
  bool pv[4];
  // The tile set-up is VALU work too (it runs beside the other resident workgroup's MFMAs, on the same ALUs): the frame
  // of a frame-aligned tile is a scalar, power-of-two widths decode with shifts, and tap validity is separable
  // (depth x row x column) instead of a loop over all taps.
  int n_tile = 0, t_tile = 0;
  if (a.frameAligned) { n_tile = (tile_m * BM) / a.HWm; t_tile = n_tile % d.T; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    pv[i] = m < a.M;
    const int mm = pv[i] ? m : 0;
    int n, rem, t, y, x;
    if (a.frameAligned) { n = n_tile; t = t_tile; rem = mm - n_tile * a.HWm; }
    else if (a.hwShift >= 0) { n = mm >> a.hwShift; rem = mm & (a.HWm - 1); t = n % d.T; }
    else { n = mm / a.HWm; rem = mm - n * a.HWm; t = n % d.T; }
    if (a.wShift >= 0) { y = rem >> a.wShift; x = rem & (d.Wm - 1); }
    else { y = rem / d.Wm; x = rem - y * d.Wm; }
    py[i] = y * d.stride - d.padH;
    px[i] = x * d.stride - d.padW;
    pbase[i] = ((n - d.padD) * d.Hin + py[i]) * d.Win + px[i];   // pixel index of tap (0,0,0)
    rowoff[i] = (pbase[i] * d.ldIn + lcol) * 4 + a.margin;   // >= 0: the descriptor starts `margin` bytes below `in`
    unsigned mk = 0;
    if (!SMALLC) {
      unsigned mw = 0, mrow = 0;
      for (int kw = 0; kw < d.KW; ++kw) mw |= ((unsigned)(px[i] + kw) < (unsigned)d.Win ? 1u : 0u) << kw;
      for (int kh = 0; kh < d.KH; ++kh) mrow |= ((unsigned)(py[i] + kh) < (unsigned)d.Hin ? mw : 0u) << (kh * d.KW);
      for (int kd = 0; kd < d.KD; ++kd) mk |= ((unsigned)(t + kd - d.padD) < (unsigned)d.T ? mrow : 0u) << (kd * khw);
      mk = pv[i] ? mk : 0u;
    }
    if (SMALLC && a.scRow) {
      // small-channel layers whose K-step is exactly one filter row (KW * Cin == 32, e.g. k4 over 8 channels): this
      // thread's kw and channel offset never change, so column validity is folded into the row's VGPR offset once
      // per tile and the per-step part (kh) is a scalar offset + one bit of a row mask -- as on the regular path
      const int kw_t = lcol >> a.cinShift, coff_t = lcol & (d.Cin - 1);
      const bool xok = pv[i] & ((unsigned)(px[i] + kw_t) < (unsigned)d.Win);
      rowoff[i] = xok ? ((pbase[i] + kw_t) * d.ldIn + coff_t) * 4 + a.margin : (int)OOB;
      for (int kh = 0; kh < d.KH; ++kh) mk |= ((unsigned)(py[i] + kh) < (unsigned)d.Hin ? 1u : 0u) << kh;
    }
    tapmask[i] = ~mk;   // bit t SET = tap t is padding (or past the last tap) for this row
  }

  // ---- K range, with fully padded temporal taps skipped
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

  // The fp32 MFMA runs on the SIMD's fp32 ALUs (its rate IS the vector FMA rate), so every VALU instruction in the
  // K-loop is taken from the matrix pipe: measured 0.85 -> 0.92 MFMA-busy with the loop's VALU work removed.  Hence:
  // the per-step part of every address is a SCALAR offset (soffset of the buffer load), the per-row part a VGPR that
  // only changes with the tap, padding is two VALU per row and step, and the input ReLU is a template parameter.
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.in) - a.margin), 0, a.inBytes + a.margin, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, a.wpBytes, 0x00020000);
  unsigned wrow[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) wrow[i] = (unsigned)(((size_t)(tile_n * BN + lrow + 32 * i) * a.Ktot + lcol) * 4);
  if (a.bankFrames > 0) {   // a tile never straddles a bank (rows per bank % 128 == 0): the bank of its first row
    const unsigned boff = (unsigned)(((tile_m * BM) / a.HWm) / a.bankFrames) * a.bankBytes;
#pragma unroll
    for (int i = 0; i < BROWS; ++i) wrow[i] += boff;
  }

  const int step_last = step_end - 1;   // filter loads past the last step re-read it (their data is never used)
  // incremental (tap, chunk) walk of the NEXT step to load (wave-uniform scalars)
  int ld_step = step_begin;
  int ld_tap = kd_lo * khw, ld_kh = 0, ld_kw = 0, ld_kd = kd_lo, ld_chunk = 0;

  f32x4 ra[4], rb[BROWS];

  // The next K-step's loads are issued in four parts (one A row + one B row each) so that each part's
  // address arithmetic sits between two 4-MFMA bursts and runs while the matrix pipe is busy.  Past
  // the last step the walk yields out-of-range offsets (zeros): no branch on "is there a next step".
  auto load_part = [&](int s) {
    if (SMALLC && a.scRow) {
      ra[s] = bufload(rin, ((tapmask[s] >> min(ld_step, 31)) << 31) | (unsigned)rowoff[s], ld_step * d.Win * d.ldIn * 4);
    } else if (SMALLC) {
      const int k = ld_step * BK + lcol;
      const int tap = k >> a.cinShift;
      const int coff = k & (d.Cin - 1);
      int kh, kw;
      if (a.kwShift >= 0) { kh = tap >> a.kwShift; kw = tap & (d.KW - 1); }
      else { kh = tap / d.KW; kw = tap - kh * d.KW; }
      const bool ok = pv[s] & (tap < ntaps) & ((unsigned)(py[s] + kh) < (unsigned)d.Hin) & ((unsigned)(px[s] + kw) < (unsigned)d.Win);
      const unsigned off = (unsigned)(((pbase[s] + kh * d.Win + kw) * d.ldIn + coff) * 4 + a.margin);
      ra[s] = bufload(rin, ok ? off : OOB, 0);
    } else {
      const int stepoff = ((((ld_kd * d.Hin) + ld_kh) * d.Win + ld_kw) * d.ldIn + ld_chunk * BK) * 4;   // scalar
      // branch-free: a padding tap gets bit 31 of the VGPR offset set = beyond the descriptor = zeros
      ra[s] = bufload(rin, ((tapmask[s] >> ld_tap) << 31) | (unsigned)rowoff[s], stepoff);
    }
    if (s < BROWS) rb[s < BROWS ? s : 0] = bufload(rwp, wrow[s < BROWS ? s : 0], min(ld_step, step_last) * (BK * 4));
    if (s == 3) {  // advance the (tap, chunk) walk
      ++ld_step;
      if (!SMALLC && ++ld_chunk == a.cinChunks) {
        ld_chunk = 0;
        ++ld_tap;
        if (++ld_kw == d.KW) {
          ld_kw = 0;
          if (++ld_kh == d.KH) { ld_kh = 0; ++ld_kd; }
        }
      }
    }
  };
  auto store_part = [&](int s, int buf) {
    float* As = As0 + buf * BM * LDS_LD;
    float* Bs = Bs0 + buf * BN * LDS_LD;
    f32x4 v = ra[s];
    if (INRELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(As + (lrow + 32 * s) * LDS_LD + lcol) = v;
    if (s < BROWS) *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * s) * LDS_LD + lcol) = rb[s < BROWS ? s : 0];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < 4; ++s) load_part(s);
#pragma unroll
  for (int s = 0; s < 4; ++s) store_part(s, 0);
  __syncthreads();
  int cur = 0;
  for (int step = step_begin; step < step_end; ++step) {
    FO_STAMP_AT(8 * (step - step_begin));
    const float* As = As0 + cur * BM * LDS_LD + (wm * TM * 32 + l31) * LDS_LD + half * 4;
    const float* Bs = Bs0 + cur * BN * LDS_LD + (wn * TN * 32 + l31) * LDS_LD + half * 4;
    f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // Scheduling fences pin the software pipeline hipcc would otherwise undo:
      //  * fragments of k-group kk+1 are requested before group kk's 16 MFMAs (1024 pipe cycles to land);
      //  * group 0 carries the next step's address arithmetic + buffer loads, 1/4 per 4-MFMA burst;
      //  * group 3 carries the LDS stores of those loads into the other buffer (landed ~2000 cycles ago).
      // Only the barrier and the first fragment read of a step are left uncovered.
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0 && kk < 3 && !(FO_ABLATE & 4)) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD + (kk + 1) * 8);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD + (kk + 1) * 8);
        }
        if (kk == 0 && !(FO_ABLATE & 1)) load_part(s);
        if (kk == 3 && !(FO_ABLATE & 2)) store_part(s, cur ^ 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(FO_ABLATE & 4) ? 0 : (kk & 1)][i][s], fb[(FO_ABLATE & 4) ? 0 : (kk & 1)][j][s], acc[i][j], 0, 0, 0);
        // Interleave 1 MFMA : a few issue slots of the side work, so the side work issues in the shadow
        // of an MFMA (64 pipe cycles each) instead of in front of the burst.
        // masks: 0x8 MFMA, 0x2 VALU, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
#pragma unroll
        for (int q = 0; q < TM * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          if (s == 0 && kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN + TM * TN - 1) / (TM * TN), 0);
          if (kk == 0) {
            __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            if (q == TM * TN - 1) __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
          }
          if (kk == 3) {
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            if (q >= TM * TN - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
      }
      FO_STAMP_AT(8 * (step - step_begin) + 1 + kk);
    }
    __builtin_amdgcn_sched_barrier(0);
    FO_STAMP_AT(8 * (step - step_begin) + 5);
    if (!(FO_ABLATE & 8)) __syncthreads();
    cur ^= 1;
  }

  if constexpr (FUSE) {
    // ---- ResBlock tail (reference models/vqvae_conv3d_latent.py:90-99): this tile's h = relu(conv3x3(relu(x)) + b1) never
    // leaves the CU before the 1x1 conv consumes it: it goes to LDS in the A-operand layout (and from there to memory once,
    // 16 B per lane: the backward needs it), the 128 x 32 filter of the 1x1 conv joins it, each wave contracts its 32 rows
    // against all 128 output columns (64 MFMAs), and the residual `out += input` (+ the encoder's / decoder's ReLU) is the
    // epilogue, straight from the accumulators (one dword per lane = whole 128-byte lines).
    float* Hs = lds;                       // [128][36]
    float* W3s = lds + BM * LDS_LD;        // [128][36]
    __syncthreads();
    {
      const float b1 = a.bias[l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) Hs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDS_LD + l31] = fmaxf(acc[0][0][r] + b1, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(W3s + (lrow + 32 * i) * LDS_LD + lcol) = *reinterpret_cast<const f32x4*>(a.wp2 + (lrow + 32 * i) * 32 + lcol);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // h -> memory (hidden activation, kept for the backward)
      const int row = lrow + 32 * i;
      const long long m = (long long)tile_m * BM + row;
      if (m < a.M) *reinterpret_cast<f32x4*>(a.out + m * d.ldOut + lcol) = *reinterpret_cast<const f32x4*>(Hs + row * LDS_LD + lcol);
    }
    f32x16 acc2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    const float* Ha = Hs + (wm * 32 + l31) * LDS_LD + half * 4;
    const float* Wb = W3s + l31 * LDS_LD + half * 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(Ha + kk * 8);
      f32x4 fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Wb + j * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[j][s], acc2[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = j * 32 + l31;
      const float b3 = a.bias2[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = (long long)tile_m * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= a.M) continue;
        float v = acc2[j][r] + b3 + a.add[m * d.ldAdd + col];
        if (a.relu2) v = fmaxf(v, 0.f);
        a.out2[m * a.ldOut2 + col] = v;
      }
    }
    return;
  }

  // ---- epilogue.  An MFMA accumulator holds one output column per lane, so storing it directly means
  // 4-byte stores in 128-B pieces (and as many scalar loads for the mask / residual operands).  The tile
  // is transposed through LDS instead (the staging buffers are free now): every thread then owns float4
  // runs of an output row, and bias / ReLU-mask / residual / stores are all 16 B per lane, whole rows
  // per wave.  This is what the short-K launches (1x1 convs, dgrads of the 32-channel layers) live on.
  constexpr int C_LD = BN + 4;
  static_assert(BM * C_LD <= 2 * (BM + BN) * LDS_LD, "C tile must fit the staging LDS");
  float* Cs = lds;
  __syncthreads();   // every wave is done reading the last K-step's fragments (and the spare stage stores)
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
  constexpr int C4 = BN / 4;                 // float4 groups per tile row
  constexpr int RPP = 256 / C4;              // rows per pass
  const int c4 = tid % C4;
  const int co = tile_n * BN + c4 * 4;
  if ((flags & FO_DEPTH2SPACE) && co < d.Cout) {
    // fused 4-phase transposed conv: GEMM column co = phase*8 + channel; pixel m = (n, y, x) of the INPUT grid
    // writes output pixel (2y + py, 2x + px), channels c..c+3 of an 8-float pixel (d.ophW real channels carry a bias)
    const int ph = co >> 3, c = co & 7, py = ph >> 1, px = ph & 1;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (flags & FO_BIAS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (c + e < d.ophW) ? a.bias[c + e] : 0.f;
    }
    for (int row = tid / C4; row < BM; row += RPP) {
      const int m = tile_m * BM + row;
      if (m >= a.M) break;
      int n, y, x;
      decode_pixel(a, m, n, y, x);
      // ophH = 1: cell form (k2 full correlation over an (Hin+1) x (Win+1) grid): the cell's pixels sit one row / column up-left
      const int oy = 2 * y + py - d.ophH, ox = 2 * x + px - d.ophH;
      if ((unsigned)oy >= (unsigned)d.Hout || (unsigned)ox >= (unsigned)d.Wout) continue;
      const size_t opix = ((size_t)n * d.Hout + oy) * d.Wout + ox;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * C_LD + c4 * 4) + bv;
      if (flags & FO_OUT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<f32x4*>(a.out + opix * d.ldOut + c) = v;
    }
    return;
  }
  store_tile<BN>(a, Cs, tile_m, tile_n, tid);
}

#include "conv_igemm3.inc"

template <int BN, int WAVES_M, int WAVES_N, int TM, int TN>
int launch(const ConvArgs& a, bool smallc, hipStream_t s) {
  const int grid = a.tilesM * a.tilesN;
  const bool inrelu = (a.d.flags & FO_IN_RELU) != 0;
  if (smallc) { if (inrelu) FO_NOTE_T("conv_igemm_kernel", BN, WAVES_M, WAVES_N, TM, TN, true, true, false); else FO_NOTE_T("conv_igemm_kernel", BN, WAVES_M, WAVES_N, TM, TN, true, false, false); }
  else { if (inrelu) FO_NOTE_T("conv_igemm_kernel", BN, WAVES_M, WAVES_N, TM, TN, false, true, false); else FO_NOTE_T("conv_igemm_kernel", BN, WAVES_M, WAVES_N, TM, TN, false, false, false); }
  if (smallc && inrelu)
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WAVES_M, WAVES_N, TM, TN, true, true>), dim3(grid), dim3(256), 0, s, a);
  else if (smallc)
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WAVES_M, WAVES_N, TM, TN, true, false>), dim3(grid), dim3(256), 0, s, a);
  else if (inrelu)
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WAVES_M, WAVES_N, TM, TN, false, true>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BN, WAVES_M, WAVES_N, TM, TN, false, false>), dim3(grid), dim3(256), 0, s, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

}  // namespace

#ifdef FO_STAMP
extern "C" int fo_debug_read_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fo_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

// Which instantiation fo_conv_igemm launches for d: 128 / 64 / 32 = conv_igemm_kernel<BN>, 3 = conv_igemm3_kernel.
static int pick_variant(const fo_conv_desc* d) {
  if (d->Cout <= 32) return 32;
  if (d->Cout <= 64) return 64;
  // 3 taps per staged A tile, one workgroup per CU: ahead of the generic kernel (two workgroups per CU) on the long-K
  // Conv3d launches (162-165 vs 160 TFLOP/s nominal at 64^2, and 1280-tile launches fill 256 slots in whole rounds),
  // behind it on short-K 2-D tiles (exposed prologue / epilogue): used for every eligible Conv3d launch.
  bool use3 = d->KD > 1;
  const char* force3 = getenv("FACEOFF_IGEMM3");   // diagnostics: 0 never, 1 every eligible launch
  if (force3) use3 = atoi(force3) != 0;
  return (d->Cin >= 32 && igemm3_eligible(d) && use3) ? 3 : 128;
}

extern "C" int fo_conv_igemm_variant(const fo_conv_desc* d) { return d ? pick_variant(d) : FO_E_SHAPE; }

static int fill_args(const fo_conv_desc* d, const float* in, const float* wp, const float* bias, const float* mask,
                     const float* add, float* out, int bank_frames, ConvArgs& a) {
  a.d = *d;
  a.wp2 = nullptr; a.bias2 = nullptr; a.out2 = nullptr; a.ldOut2 = 0; a.relu2 = 0;
  a.in = in; a.wp = wp; a.bias = bias; a.mask = mask; a.add = add; a.out = out;
  FO_REQUIRE(d->N > 0 && d->T > 0 && d->N % d->T == 0, FO_E_SHAPE, "conv: N=%d not a multiple of T=%d", d->N, d->T);
  FO_REQUIRE(d->Cin % 4 == 0 && d->ldIn % 4 == 0, FO_E_ALIGN, "conv: Cin/ldIn must be multiples of 4");
  FO_REQUIRE(fo_aligned16(in) && fo_aligned16(wp), FO_E_ALIGN, "conv: in/wp must be 16-byte aligned");
  if (d->flags & FO_DEPTH2SPACE) {
    FO_REQUIRE(d->Cout == 32 && d->ldOut >= 8 && d->ldOut % 4 == 0 && fo_aligned16(out) && (d->ophH == 0 || d->ophH == 1) &&
                   d->Hout == 2 * (d->Hm - d->ophH) && d->Wout == 2 * (d->Wm - d->ophH) && !(d->flags & (FO_MASK | FO_ADD)) &&
                   d->ophW >= 1 && d->ophW <= 8,
               FO_E_SHAPE, "conv: FO_DEPTH2SPACE needs Cout == 32 (4 phases x 8), ldOut >= 8, a 2x output grid, no mask/add");
  } else
    FO_REQUIRE(fo_aligned16(out) && d->ldOut % 4 == 0 && d->ldOut >= (d->Cout + 3) / 4 * 4, FO_E_ALIGN,
               "conv: out must be 16-byte aligned with ldOut %% 4 == 0 and room for Cout rounded up to 4");
  FO_REQUIRE(!(d->flags & FO_MASK) || (fo_aligned16(mask) && d->ldMask % 4 == 0), FO_E_ALIGN, "conv: mask alignment");
  FO_REQUIRE(!(d->flags & FO_ADD) || (fo_aligned16(add) && d->ldAdd % 4 == 0), FO_E_ALIGN, "conv: add alignment");
  FO_REQUIRE(!(d->flags & FO_BIAS) || bias, FO_E_SHAPE, "conv: FO_BIAS without bias");
  FO_REQUIRE(!(d->flags & FO_MASK) || mask, FO_E_SHAPE, "conv: FO_MASK without mask");
  FO_REQUIRE(!(d->flags & FO_ADD) || add, FO_E_SHAPE, "conv: FO_ADD without add");
  const int taps = d->KD * d->KH * d->KW;
  FO_REQUIRE(taps >= 1 && taps <= 31, FO_E_SHAPE, "conv: at most 31 taps (got %d)", taps);
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
  a.scRow = smallc && d->KW * d->Cin == BK && d->KD == 1;
  a.wShift = a.hwShift = a.kwShift = -1;
  for (int sft = 0; sft < 30; ++sft) {
    if ((1 << sft) == d->Wm) a.wShift = sft;
    if ((1 << sft) == a.HWm) a.hwShift = sft;
    if ((1 << sft) == d->KW) a.kwShift = sft;
  }
  const unsigned long long inBytes = (((unsigned long long)d->N * d->Hin * d->Win - 1) * d->ldIn + d->Cin) * 4ull;
  const int opad = d->Cout > 64 ? (d->Cout + 127) / 128 * 128 : (d->Cout > 32 ? 64 : 32);
  const unsigned long long wpBytes = (unsigned long long)opad * a.Ktot * 4ull;
  FO_REQUIRE(inBytes < (1ull << 31) && wpBytes < (1ull << 31), FO_E_SHAPE, "conv: tensor exceeds the 2 GiB buffer-descriptor window");
  const long long margin = (((long long)d->padD * d->Hin + d->padH) * d->Win + d->padW) * d->ldIn * 4ll;
  FO_REQUIRE(margin >= 0 && inBytes + (unsigned long long)margin < (1ull << 31), FO_E_SHAPE, "conv: tensor exceeds the 2 GiB buffer-descriptor window");
  a.margin = (int)margin;
  a.inBytes = (unsigned)inBytes;
  a.wpBytes = (unsigned)wpBytes;
  a.bankFrames = bank_frames;
  a.bankBytes = (unsigned)wpBytes;
  if (bank_frames > 0) {
    FO_REQUIRE(((long long)bank_frames * a.HWm) % BM == 0 && !smallc && d->N % bank_frames == 0, FO_E_SHAPE,
               "conv: a filter bank's rows (bank_frames * Hm * Wm) must be a multiple of 128");
    const unsigned long long allBanks = (unsigned long long)(d->N / bank_frames) * wpBytes;
    FO_REQUIRE(allBanks < (1ull << 31), FO_E_SHAPE, "conv: filter banks exceed the 2 GiB window");
    a.wpBytes = (unsigned)allBanks;
  }
  return FO_OK;
}

static int conv_igemm_impl(const fo_conv_desc* d, const float* in, const float* wp, const float* bias, const float* mask,
                           const float* add, float* out, void* stream, int bank_frames) {
  ConvArgs a;
  if (int rc = fill_args(d, in, wp, bias, mask, add, out, bank_frames, a)) return rc;
  const bool smallc = d->Cin < 32;
  hipStream_t s = (hipStream_t)stream;
  if (smallc && bank_frames == 0 && fo_conv_img_try(d, in, wp, bias, mask, add, out, s) == 0) return FO_OK;
  if (bank_frames == 0 && !bias && d->Cin == 32 && d->Cout == 128 && fo_conv3x3_c32_halo_try(d, in, wp, mask, add, out, s)) {
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  // BN by output channels (filters are packed with Cout rounded up to the same BN)
  if (d->Cout > 64) {
    a.tilesN = (d->Cout + 127) / 128;
    if (pick_variant(d) == 3 && bank_frames == 0) {
      if (d->flags & FO_IN_RELU) FO_NOTE_T("conv_igemm3_kernel", 2, 2, true); else FO_NOTE_T("conv_igemm3_kernel", 2, 2, false);
      if (d->flags & FO_IN_RELU) hipLaunchKernelGGL((conv_igemm3_kernel<2, 2, true>), dim3(a.tilesM * a.tilesN), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((conv_igemm3_kernel<2, 2, false>), dim3(a.tilesM * a.tilesN), dim3(256), 0, s, a);
      FO_CHECK_LAUNCH();
      return FO_OK;
    }
    return launch<128, 2, 2, 2, 2>(a, smallc, s);
  } else if (d->Cout > 32) {
    a.tilesN = 1;
    return launch<64, 2, 2, 2, 1>(a, smallc, s);
  } else {
    a.tilesN = 1;
    return launch<32, 4, 1, 1, 1>(a, smallc, s);
  }
}

extern "C" int fo_conv_igemm(const fo_conv_desc* d, const float* in, const float* wp, const float* bias,
                             const float* mask, const float* add, float* out, void* stream) {
  return conv_igemm_impl(d, in, wp, bias, mask, add, out, stream, 0);
}

// ResBlock forward in ONE launch (reference models/vqvae_conv3d_latent.py:86-101): h = relu(conv3x3(relu(x)) + b1) is written
// to `hbuf` (the backward needs it) and out = [relu](conv1x1(h) + b3 + x) to `out`.  d describes the 3x3 conv (Cin = C,
// Cout = 32, k 1x3x3 p1 s1 on a same-size grid; ldOut = pixel stride of hbuf, ldAdd = pixel stride of x); wp1 / wp3 are the
// fo_pack_conv packs of the 3x3 (32 rows) and 1x1 (128 rows x 32) filters.  C must be 128.
extern "C" int fo_resblock_fwd(const fo_conv_desc* d, const float* x, const float* wp1, const float* b1, const float* wp3,
                               const float* b3, float* hbuf, float* out, int ldOut2, int out_relu, void* stream) {
  FO_REQUIRE(d && x && wp1 && b1 && wp3 && b3 && hbuf && out, FO_E_SHAPE, "resblock_fwd: null pointer");
  FO_REQUIRE(d->Cout == 32 && d->Cin == 128 && d->KD == 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->ostride == 1 && d->padH == 1 &&
                 d->padW == 1 && d->Hm == d->Hin && d->Wm == d->Win && d->Hout == d->Hin && d->Wout == d->Win && d->T >= 1,
             FO_E_SHAPE, "resblock_fwd: a 3x3 pad-1 stride-1 conv 128 -> 32 on a same-size grid is required");
  FO_REQUIRE(d->ldAdd >= 128 && d->ldAdd % 4 == 0 && ldOut2 >= 128 && ldOut2 % 4 == 0 && d->ldOut >= 32 && d->ldOut % 4 == 0 &&
                 fo_aligned16(hbuf) && fo_aligned16(out) && fo_aligned16(wp3), FO_E_ALIGN, "resblock_fwd: strides / alignment");
  if (fo_resblock_halo_try(d, x, wp1, b1, wp3, b3, hbuf, out, ldOut2, out_relu, (hipStream_t)stream)) {   // widths that are multiples of 32
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  fo_conv_desc dd = *d;
  dd.flags = FO_IN_RELU | FO_BIAS | FO_OUT_RELU;
  ConvArgs a;
  if (int rc = fill_args(&dd, x, wp1, b1, nullptr, x, hbuf, 0, a)) return rc;
  a.wp2 = wp3; a.bias2 = b3; a.out2 = out; a.ldOut2 = ldOut2; a.relu2 = out_relu;
  a.tilesN = 1;
  FO_NOTE_T("conv_igemm_kernel", 32, 4, 1, 1, 1, false, true, true);
  hipLaunchKernelGGL((conv_igemm_kernel<32, 4, 1, 1, 1, false, true, true>), dim3(a.tilesM * a.tilesN), dim3(256), 0, (hipStream_t)stream, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// The same contraction with the filter chosen per frame: frames [b*bank_frames, (b+1)*bank_frames) use the b-th filter
// bank (banks are consecutive packed filters of Opad*taps*Cin floats).  The 16 GEMMs of a Winograd-transformed Conv3d are
// one such launch over the stack of transformed planes (winograd.hip).  bank_frames * Hm * Wm must be a multiple of 128.
extern "C" int fo_conv_igemm_banked(const fo_conv_desc* d, const float* in, const float* wp, float* out, int bank_frames,
                                    void* stream) {
  FO_REQUIRE(bank_frames > 0 && !(d->flags & (FO_BIAS | FO_MASK | FO_ADD | FO_DEPTH2SPACE)), FO_E_SHAPE, "conv_banked: plain GEMM only");
  return conv_igemm_impl(d, in, wp, nullptr, nullptr, nullptr, out, stream, bank_frames);
}
