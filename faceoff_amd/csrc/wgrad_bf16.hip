// Filter gradients of the bf16-operand VQ-VAE step on the gfx950 bf16 MFMA (v_mfma_f32_16x16x32_bf16): bf16 activations and bf16
// activation gradients in, fp32 accumulation, fp32 filter gradient out in the checkpoint layout.  Replaces the cuDNN wgrad kernels
// torch.autocast(bfloat16) would run under loss.backward() (reference train_faceoff_perceptual.py:100) for every conv of
// models/vqvae_conv3d_latent.py:92-190.
//
//   dW[a][b][tap] = sum_m P[m][a] * Q[qpix(m, tap)][b]        m over the N * Hm * Wm positions of P (the tensor on the conv's OUTPUT
//                                                             grid), qpix = (frame + kd - padD, y*stride + kh - padH, x*stride + kw - padW)
//
// A TN GEMM: the contraction index is the ROW of both channels-last operands, while the MFMA wants eight consecutive k per lane.
// As in wino_wgrad_split.hip the K-step's rows are stored in LDS as they arrive, [k][channel], and gfx950's transposing LDS read
// (ds_read_b64_tr_b16: lane i of a 16-lane block receives element i & 3 of the 8-byte chunk lane 4e + (i >> 2) addressed, e = 0..3)
// hands each lane its fragment -- no register transposes, no 2-byte stores.
//
// Two forms of one kernel template:
//  * ROW RUNS (Wm % 32 == 0: every C2 shape).  A K-step is a run of 32 consecutive output pixels of one image row; the pixels of Q
//    that ALL the kw taps of one (kd, kh) need for it are consecutive too (32 stride + KW - stride of them), so they are staged
//    ONCE and tap kw reads its fragments at a row offset: a workgroup owns the NKW = KW blocks dW[.][.][kd][kh][0..KW) (3 x 128 x 128
//    for a 3x3 / 3x3x3 conv, 4 x 128 x 64 for the k4 s2 stems) over a slab of the runs.  Runs whose (frame, row) of Q is padding
//    for this (kd, kh) are skipped whole.
//  * GATHER (any other geometry): a K-step is 32 consecutive positions m, each row's source pixel is decoded on its own, one tap
//    per workgroup.
// The 8-channel image layers (enc_b.blocks.0, dec.blocks.6: Cb = 8, KW = 4) run the row-run form with the FOUR kw taps as the
// b index: the 4 pixels x 8 channels a tap row needs are 64 contiguous bytes of the input row (b' = kw * 8 + c, "SMALLC").
// Slabs are summed in a fixed order by wgrad_bf16_reduce_kernel (bitwise reproducible, no atomics), which also clips to the real
// channel counts and writes [a][b][taps].
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"

#ifndef FO_ABLATE_W   // diagnostic builds (hipcc -DFO_ABLATE_W=<bits>, a library per variant handed in through FACEOFF_HIP_LIB; profiles/r06_experiments.md): bit 0 no
#define FO_ABLATE_W 0 // global loads / out-of-range DMAs only, 1 no LDS stores (row-run form), 2 no fragment reads / MFMAs, 3 no epilogue stores (wgrad9).  Results are
                      // wrong, only the timing is of interest
#endif

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WArgs {
  fo_conv_desc d;            // Cout = channels of P (a), Cin = channels of Q (b); ldOut = ldP, ldIn = ldQ; Hm/Wm = grid of P
  const __bf16* P;
  const __bf16* Q;
  unsigned Pbytes, Qbytes;   // extents of P and Q (< 2^31: the loads are buffer loads with 32-bit byte offsets; a masked element asks for offset WOOB and gets zeros)
  float* ws;                 // [slabs][taps][Apad][Bpad]
  int tilesA, tilesB, tapRows, taps;
  int Apad, Bpad;
  int units;                 // gather / image-layer forms: 32-position steps (row runs) in all
  int runsPerRow;            // Wm / 32 (row-run form)
  int inrelu;
  int biasTapRow;            // the tap row whose workgroups also sum the columns of P (it visits every position), -1: no bias gradient
  float* wsBias;             // [slabs of the bias tap row][Apad]
  // Slabs: tap row tr is cut into X * mOf[tr] slabs of EQUAL WORK (X = 8: slab s runs on the XCD that holds the slabs of the other tap
  // rows covering the same stretch of positions -- blocks b and b + 8 share an XCD's L2; X = 1 for small problems).  A tap row's work
  // is the positions whose (frame, row) of Q is not padding for it: (T - 1) of T frames for the outer depth taps, (H - 1) of H rows
  // for the outer kh, so the outer tap rows get fewer slabs.
  int X, wgPerX;             // wgPerX = sum of mOf
  short mOf[32];
};

// relu() of two packed bf16: as 16-bit integers the negative patterns are exactly those with the sign bit, so it is one v_pk_max_i16 against 0
// (the same bits as `w & ~(((w & 0x80008000u) >> 15) * 0xffffu)`, four instructions)
__device__ __forceinline__ unsigned relu_pk(unsigned w) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), s16x2{0, 0}));
}

constexpr unsigned WOOB = 0x80000000u;
__device__ __forceinline__ u32x4 wbufload16(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }

__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
}

// LDS row pitch for C channels whose rows are read at a row stride S (1, or 2 for the k4 s2 stems' Q): the four rows a 16-lane
// block reads (k, k+1, k+2, k+3 times S) must sit in four different 64-byte bank groups: pitch * S = 64 (mod 256).
constexpr int row_pitch(int C, int S) { return S == 2 ? (C <= 64 ? 160 : 288) : (C == 32 ? 64 : 320); }
constexpr int wgrad_lds_bytes_rt(int TA, int TB, int NKW, int KR, bool fast, bool smallc) {
  const int S = (NKW == 4 && !smallc) ? 2 : 1;                       // row stride of Q in LDS (k4 s2)
  const int NQ = smallc ? 128 : (fast ? 32 * S + NKW - S : 32);      // rows of Q staged per run
  return 2 * KR * (32 * row_pitch(TA, 1) + NQ * row_pitch(TB, S));   // two stages
}

// TA x TB block, WA x (8 / WA) waves, NKW taps per workgroup (1, 3 or 4), KR runs per barrier pair (thin blocks: more MFMAs per step).
// FAST: row-run form.  SMALLC: the image layers (Cb = 8, k4 s2): b' = kw * 8 + c, and the NKW = 4 taps of the workgroup are the four kh.
// THIN blocks (two runs per barrier and at most 3 x 32 x 128 accumulator elements per workgroup): the kernel is bound by load -> LDS -> barrier
// latency, not by the matrix pipe (MFMA-busy 0.11 on the ResBlocks' 3x3 128 -> 32 filter gradient), so TWO workgroups share a CU -- the launch
// bound keeps them at <= 128 VGPRs (they compiled to 85-130), and the planner hands out two rounds of slabs where their LDS fits twice (round 5).
constexpr bool wgrad_thin(int TA, int TB, int NKW, int KR) { return KR == 2 && NKW * TA * TB <= 3 * 32 * 128; }
// ... and of those the forms the planner really hands two workgroups per CU (make_plan: NKW >= 3 -- the 1x1 forms measured nothing from it and keep one):
// only they carry the 128-VGPR launch bound (a 1x1 thin variant above 128 registers would have spilled for no occupancy gain)
constexpr bool wgrad_two_per_cu(int TA, int TB, int NKW, int KR) { return NKW >= 3 && wgrad_thin(TA, TB, NKW, KR); }
// LDS bytes of an instantiation: ONE formula for the kernel's launch (launch_w) and the planner (make_plan)
constexpr int row_pitch(int C, int S);
constexpr int wgrad_lds_bytes_rt(int TA, int TB, int NKW, int KR, bool fast, bool smallc);

template <int TA, int TB, int WA, int NKW, int KR, bool FAST, bool SMALLC>
__global__ __launch_bounds__(512, (wgrad_two_per_cu(TA, TB, NKW, KR) ? 4 : 2)) void wgrad_bf16_kernel(const WArgs a) {
  constexpr int WB = 8 / WA;
  constexpr int MA = TA / WA / 16, MB = TB / WB / 16;                // 16 x 16 tiles per wave
  static_assert(MA >= 1 && MB >= 1 && TA % (WA * 16) == 0 && TB % (WB * 16) == 0, "wave tiling");
  static_assert(!SMALLC || (FAST && NKW == 4 && TB == 32), "image-layer form");
  constexpr int S = (NKW == 4 && !SMALLC) ? 2 : 1;                   // row stride of Q in LDS (k4 s2)
  constexpr int NQ = SMALLC ? 128 : (FAST ? 32 * S + NKW - S : 32);  // rows of Q staged per run
  constexpr int PA = row_pitch(TA, 1), PB = row_pitch(TB, S);
  constexpr int SWB = S == 2 ? 4 : 3;                                // granule-swap parity bit of a Q row: (row >> SWB) & 1
  constexpr int CA = TA / 8, CB = TB / 8;                            // 16-byte chunks per row
  constexpr int NPA = (32 * CA + 511) / 512, NPB = (NQ * CB + 511) / 512;
  constexpr int SA = 32 * PA, SB = NQ * PB;                          // bytes per run
  constexpr int STAGE = KR * (SA + SB);                              // two stages: step n+1 is stored while step n is on the matrix pipe
  static_assert(2 * STAGE <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 * STAGE bytes (wgrad_lds_bytes)
  unsigned char* const As = lds;
  unsigned char* const Bs = lds + KR * SA;
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave / WB, wb = wave % WB;
  const int l15 = lane & 15, kg = lane >> 4;

  // ---- this workgroup: tile (a, b), then X-way interleaved (tap row, slab) pairs: see WArgs
  int w = blockIdx.x;
  const int perTile = a.X * a.wgPerX;
  const int tile = w / perTile; w -= tile * perTile;
  const int tb = tile % a.tilesB, ta = tile / a.tilesB;
  const int xcd = w % a.X;
  int li = w / a.X, tr = 0;
  while (li >= a.mOf[tr]) { li -= a.mOf[tr]; ++tr; }
  const int slab = xcd * a.mOf[tr] + li, nslab = a.X * a.mOf[tr];
  // tap row -> (kd, kh) [row-run form: all kw] or one tap (kd, kh, kw) [gather form]; image layers: all of kh, kw
  int kd, kh, kw0;
  if (SMALLC) { kd = 0; kh = 0; kw0 = 0; }
  else if (FAST) { kd = tr / d.KH; kh = tr - kd * d.KH; kw0 = 0; }
  else { kd = tr / (d.KH * d.KW); const int r = tr - kd * d.KH * d.KW; kh = r / d.KW; kw0 = r - kh * d.KW; }
  // the positions this tap row really visits (row-run form): frames tlo..thi of a clip, rows ylo..yhi -- "virtual units" v = (clip, t', y', run)
  int tlo = 0, Tv = d.T, ylo = 0, Hv = d.Hm, vunits = a.units;
  if (FAST && !SMALLC) {
    tlo = max(0, d.padD - kd);
    Tv = min(d.T - 1, d.T - 1 + d.padD - kd) - tlo + 1;
    ylo = max(0, (d.padH - kh + d.stride - 1) / d.stride);
    Hv = min(d.Hm - 1, (d.Hin - 1 + d.padH - kh) / d.stride) - ylo + 1;
    vunits = (Tv > 0 && Hv > 0) ? (d.N / d.T) * Tv * Hv * a.runsPerRow : 0;
  }
  const int u0 = (int)((long long)vunits * slab / nslab), u1 = (int)((long long)vunits * (slab + 1) / nslab);
  const int HWm = d.Hm * d.Wm;
  const int M = d.N * HWm;
  const bool bias_wg = tb == 0 && tr == a.biasTapRow;                // this workgroup also sums the columns of P (the bias gradient)

  // ---- loader roles: chunk id = tid + 512 i -> (row, 16-byte chunk)
  int prow[NPA], pcol[NPA], qrow[NPB], qcol[NPB];
#pragma unroll
  for (int i = 0; i < NPA; ++i) { const int id = tid + 512 * i; prow[i] = id / CA; pcol[i] = id - prow[i] * CA; }
#pragma unroll
  for (int i = 0; i < NPB; ++i) { const int id = tid + 512 * i; qrow[i] = id / CB; qcol[i] = id - qrow[i] * CB; }
  const int a0 = ta * TA, b0 = tb * TB;
  constexpr int D = KR == 1 ? 3 : 2;                                 // K-steps of global loads in flight (register ring)
  u32x4 rp[D][KR][NPA], rq[D][KR][NPB];

  // Row-run form: the units are requested strictly in increasing order, so the position of the next one is a cursor advanced by
  // increments (no divisions in the loop), and each thread's share of an address is a constant offset from a wave-uniform base.
  int c_run = 0, c_y = 0, c_t = 0, c_clip = 0;                       // cursor = virtual unit (clip, t', y', run) of the next load
  if (FAST) {
    int v = u0;
    c_run = v % a.runsPerRow; v /= a.runsPerRow;
    c_y = v % max(Hv, 1); v /= max(Hv, 1);
    c_t = v % max(Tv, 1); c_clip = v / max(Tv, 1);
  }
  int offP[NPA], offQ[NPB];
  bool okP[NPA], okQ[NPB];
#pragma unroll
  for (int i = 0; i < NPA; ++i) { offP[i] = (prow[i] * d.ldOut + a0 + pcol[i] * 8) * 2; okP[i] = prow[i] < 32 && a0 + pcol[i] * 8 < d.Cout; }     // (bytes)
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    if (SMALLC) { offQ[i] = ((qrow[i] >> 5) * d.Win + (qrow[i] & 31) * d.stride + qcol[i]) * d.ldIn * 2; okQ[i] = qrow[i] < NQ; }
    else { offQ[i] = (qrow[i] * d.ldIn + b0 + qcol[i] * 8) * 2; okQ[i] = qrow[i] < NQ && b0 + qcol[i] * 8 < d.Cin; }
  }
  // Buffer loads, no branches: a masked element (padding column, channel tail, a unit past the slab's end) asks for offset WOOB and the hardware's
  // range check returns zeros.  The loads of a step are then straight-line code and the compiler COUNTS them (s_waitcnt vmcnt(N) before the LDS
  // stores of the step requested D - 1 steps earlier); with a branch around every load it could not, and every step drained the whole ring
  // (vmcnt(0): one step of latency hidden instead of D - 1 -- round 6).
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.P), 0, a.Pbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.Q), 0, a.Qbytes, 0x00020000);
  auto load = [&](int u, u32x4 (&xp)[NPA], u32x4 (&xq)[NPB]) {       // u >= u1: zeros (a K-step's unused run)
    const bool live = u < u1 && !(FO_ABLATE_W & 1);
    if (FAST) {
      const int n = c_clip * d.T + tlo + c_t, y = ylo + c_y, x0 = c_run * 32;
      if (++c_run == a.runsPerRow) { c_run = 0; if (++c_y == Hv) { c_y = 0; if (++c_t == Tv) { c_t = 0; ++c_clip; } } }
      const int pb = (((n * d.Hm + y) * d.Wm + x0) * d.ldOut) * 2;                    // byte offsets (wave-uniform part); a dead unit's may be anything
      const int qx0 = x0 * d.stride - d.padW, qy0 = y * d.stride + kh - d.padH;     // (image layers: kh = 0, the thread's own kh is in offQ)
      const int qb = ((((n + kd - d.padD) * d.Hin + qy0) * d.Win + qx0) * d.ldIn) * 2;
#pragma unroll
      for (int i = 0; i < NPA; ++i) xp[i] = wbufload16(rP, (live && okP[i]) ? (unsigned)(pb + offP[i]) : WOOB);
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        bool ok = live && okQ[i];
        if (SMALLC) ok = ok && (unsigned)(qy0 + (qrow[i] >> 5)) < (unsigned)d.Hin && (unsigned)(qx0 + (qrow[i] & 31) * d.stride + qcol[i]) < (unsigned)d.Win;
        else ok = ok && (unsigned)(qx0 + qrow[i]) < (unsigned)d.Win;
        xq[i] = wbufload16(rQ, ok ? (unsigned)(qb + offQ[i]) : WOOB);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int m = u * 32 + prow[i];
        const bool ok = live && prow[i] < 32 && m < M && a0 + pcol[i] * 8 < d.Cout;
        xp[i] = wbufload16(rP, ok ? (unsigned)((m * d.ldOut + a0 + pcol[i] * 8) * 2) : WOOB);
      }
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int m = u * 32 + qrow[i];
        const int mm = (live && m < M) ? m : 0;
        const int n = mm / HWm, rem = mm - n * HWm, y = rem / d.Wm, x = rem - y * d.Wm;
        const int t = n % d.T;
        const int qy = y * d.stride + kh - d.padH, qx = x * d.stride + kw0 - d.padW;
        const bool ok = live && qrow[i] < 32 && m < M && ((unsigned)(t + kd - d.padD) < (unsigned)d.T) && ((unsigned)qy < (unsigned)d.Hin) &&
                        ((unsigned)qx < (unsigned)d.Win) && b0 + qcol[i] * 8 < d.Cin;
        xq[i] = wbufload16(rQ, ok ? (unsigned)(((((n + kd - d.padD) * d.Hin + qy) * d.Win + qx) * d.ldIn + b0 + qcol[i] * 8) * 2) : WOOB);
      }
    }
  };
  auto store = [&](int st, int rr, const u32x4 (&xp)[NPA], const u32x4 (&xq)[NPB]) {
    if (FO_ABLATE_W & 2) return;
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      if (prow[i] < 32) *reinterpret_cast<u32x4*>(As + st * STAGE + rr * SA + prow[i] * PA + ((pcol[i] * 16) ^ (((prow[i] >> 3) & 1) << 5))) = xp[i];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      if (qrow[i] < NQ) {
        u32x4 v = xq[i];
        if (a.inrelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu_pk(v[e]);
        }
        *reinterpret_cast<u32x4*>(Bs + st * STAGE + rr * SB + qrow[i] * PB + ((qcol[i] * 16) ^ (((qrow[i] >> SWB) & 1) << 5))) = v;
      }
  };

  f32x4 acc[NKW][MA][MB], accb[MA];
#pragma unroll
  for (int k = 0; k < NKW; ++k)
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) acc[k][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MA; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

  // ---- fragment addressing.  A: rows k = 8 kg + (l15 >> 2) (+4), channels 16 i + 4 (l15 & 3); a 16-channel tile is 32 bytes, and the
  // granule swap of a row (XOR 32 of the byte column, applied to the WHOLE column) exchanges neighbouring tiles
  const int krow = kg * 8 + (l15 >> 2);
  const int parA = kg & 1;                                           // ((krow) >> 3) & 1 == ((krow + 4) >> 3) & 1
  const unsigned char* const Af = As + krow * PA + (l15 & 3) * 8;
  const int colA = (wa * (TA / WA)) * 2, colB = (wb * (TB / WB)) * 2;
  // B: row of LDS = k * S + kw  (image layers: kh * 32 + k)
  int rowB[NKW][2], parB[NKW][2];
#pragma unroll
  for (int k = 0; k < NKW; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = SMALLC ? k * 32 + krow + 4 * h : (krow + 4 * h) * S + (FAST ? k : 0);
      rowB[k][h] = r * PB + (l15 & 3) * 8;
      parB[k][h] = (r >> SWB) & 1;
    }

  // one run on the matrix pipe: ALL of its fragments are read first (20 transposing reads in flight at once: their latency is paid once
  // per run, not once per group of MFMAs -- the two waves of a SIMD leave the barrier together, so neither covers the other's waits),
  // then the MFMAs issue back to back
  auto compute = [&](int st, int rr, auto&& mid) {     // mid(): issued after the first third of the MFMAs (the next step's LDS stores and requests)
    if (FO_ABLATE_W & 4) { mid(); return; }
    bf16x8 fa[MA], fb[NKW][MB];
#pragma unroll
    for (int i = 0; i < MA; ++i) {
      const bf16x4 lo = tr_read(Af + st * STAGE + rr * SA + ((colA + i * 32) ^ (parA << 5)));
      const bf16x4 hi = tr_read(Af + st * STAGE + rr * SA + 4 * PA + ((colA + i * 32) ^ (parA << 5)));
      fa[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int k = 0; k < NKW; ++k)
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const bf16x4 lo = tr_read(Bs + st * STAGE + rr * SB + rowB[k][0] + ((colB + j * 32) ^ (parB[k][0] << 5)));
        const bf16x4 hi = tr_read(Bs + st * STAGE + rr * SB + rowB[k][1] + ((colB + j * 32) ^ (parB[k][1] << 5)));
        fb[k][j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    __builtin_amdgcn_sched_barrier(0);
    if (bias_wg) {       // column sums of P: an MFMA against a fragment of ones, the MA tiles dealt over the WB waves that hold the same rows
#pragma unroll
      for (int i = 0; i < MA; ++i)
        if (i % WB == wb % (MA < WB ? MA : WB) && wb < (MA < WB ? MA : WB)) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int i = 0; i < MA; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[k][j], acc[k][i][j], 0, 0, 0);
      if (k == 0) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // K-steps of KR runs each (a last step's unused runs are zeros).  LDS holds two stages, registers a ring of D steps: iteration n puts
  // stage n & 1 on the matrix pipe, then stores step n+1 (requested D compute phases ago: an HBM miss under load takes several) into the
  // other stage and requests step n+1+D into the slot that freed.  ONE barrier per step.
  if (u0 < u1) {
#pragma unroll
    for (int rr = 0; rr < KR; ++rr) load(u0 + rr, rp[0][rr], rq[0][rr]);
#pragma unroll
    for (int rr = 0; rr < KR; ++rr) store(0, rr, rp[0][rr], rq[0][rr]);
#pragma unroll
    for (int k = 1; k <= D; ++k) {                                    // steps 1 .. D -> slots 1, 2, .., 0
#pragma unroll
      for (int rr = 0; rr < KR; ++rr) load(u0 + k * KR + rr, rp[k % D][rr], rq[k % D][rr]);
    }
    __syncthreads();
    int u = u0;
    while (u < u1) {
#pragma unroll
      for (int sl = 0; sl < 2 * D; ++sl) {     // step index n = sl (mod 2 D): LDS stage sl & 1 and register slot (sl + 1) % D are compile-time
        if (u >= u1) break;
        constexpr int dummy = 0; (void)dummy;
        const int st = sl & 1;
        // the next step's LDS stores and the request for the step after are issued INSIDE this step's MFMAs (after the first tap's): as a phase
        // of their own behind the MFMAs -- all eight waves are in the same phase -- they left the matrix pipe idle (tools/ablate_wgrad.sh, 64^2
        // Conv3d: 0.74 ms with, 0.41 ms without the stores)
        auto stage_next = [&]() {              // (unconditional: units past the slab's end load zeros from WOOB and nobody reads their stage)
#pragma unroll
          for (int rr = 0; rr < KR; ++rr) store(st ^ 1, rr, rp[(sl + 1) % D][rr], rq[(sl + 1) % D][rr]);
#pragma unroll
          for (int rr = 0; rr < KR; ++rr) load(u + (D + 1) * KR + rr, rp[(sl + 1) % D][rr], rq[(sl + 1) % D][rr]);
        };
        if (KR == 1) {                           // (two thin runs per step: measured slower with the stores inside, +8 % on the ResBlock 3x3)
          compute(st, 0, stage_next);
        } else {
#pragma unroll
          for (int rr = 0; rr < KR; ++rr)
            if (u + rr < u1) compute(st, rr, [] {});
          stage_next();
        }
        __syncthreads();
        u += KR;
      }
    }
  }

  // ---- partial blocks -> ws[slab][tap][a][b]: accumulator register r of lane l = (a = 4 (l >> 4) + r, b = l & 15)
#pragma unroll
  for (int k = 0; k < NKW; ++k) {
    const int tap = SMALLC ? k : (FAST ? tr * NKW + k : tr);
    float* o = a.ws + (((long long)slab * (SMALLC ? 4 : a.taps) + tap) * a.Apad + a0 + wa * (TA / WA)) * a.Bpad + b0 + wb * (TB / WB);   // (slab < X * mOf[tap row])
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(long long)(i * 16 + kg * 4 + r) * a.Bpad + j * 16 + l15] = acc[k][i][j][r];
  }
  if (bias_wg && l15 == 0) {
    float* ob = a.wsBias + (long long)slab * a.Apad + a0 + wa * (TA / WA);
#pragma unroll
    for (int i = 0; i < MA; ++i)
      if (i % WB == wb % (MA < WB ? MA : WB) && wb < (MA < WB ? MA : WB)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[i * 16 + kg * 4 + r] = accb[i][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------ all nine taps of a 3x3 (x KD) filter per workgroup
// The row-run form above gives a workgroup the three kw taps of ONE (kd, kh): the three kh of a plane are three workgroups that each stream P
// and Q from memory (rocprofv3 --pmc, 128 x 128 at 64^2: 770 MB fetched per launch for 336 MB of operands, L2 hit 0.23), and the launch is as long
// as its loads and its MFMAs added up (ablations: loads + LDS stores alone 0.146 ms, fragment reads + MFMAs alone 0.151, together 0.26).  Here a
// workgroup owns a 128 (channels of P) x 64 (channels of Q) block of ALL NINE taps of one depth plane -- 9 x 128 x 64 fp32 accumulators, 144
// registers per lane -- so P is streamed once per 64 channels of Q and the three rows of Q a run needs (y - 1, y, y + 1) come out of L2 two times
// in three (the same workgroup asked for them one and two rows earlier).  36 MFMAs per wave and K-step instead of 24.
//   * Staging is LDS-DMA (buffer_load .. lds): no staging registers, no ds_write, nothing between a tile's request and its use but a counted
//     s_waitcnt vmcnt.  The LDS image is therefore lane-linear (1 KB per wave instruction); the conflict-free placement the transposing read
//     needs -- the four rows k .. k+3 of a 16-lane block in four different 64-byte bank groups, rows k + 8 .. in the other 32-byte half -- is
//     an XOR of the 16-byte chunk index by a function of the row, applied to the DMA's SOURCE address and to the fragment address:
//     P (256-byte rows): chunk ^ (4 (r & 3) ^ 2 ((r >> 3) & 1));  Q (128-byte rows): chunk ^ (2 ((r >> 1) & 1) + 4 ((r >> 3) & 1)).
//     Padding (rows of Q above / below the image, the pixels left and right of a row, channel tails, units past the slab) is an out-of-range
//     offset: the DMA writes zeros.
//   * One barrier per K-step, every wave the same program, software-pipelined over the three rows of taps: while the 12 MFMAs of row kh issue,
//     the six fragment reads of row kh + 1 are in flight; behind the barrier (after row 1) the next K-step's fragments of P and of row 0 are
//     requested, under the MFMAs of row 2.  Two fragment sets of P and two of Q's rows alternate (the loop body is two K-steps), so nothing
//     waits for a read it has just issued, and the two waves of a SIMD interleave freely.  (A first version ran two wave groups one barrier
//     apart as conv_bf16_pp16_kernel does: with ONE wave per SIMD issuing MFMAs at a time the matrix pipe stayed below 0.6 busy, and the
//     ~120 scalar / address instructions a wave spends per tile request were as long as the other group's MFMAs.)
//     The transposing reads are inline assembly (see w9_tr) with hand-counted s_waitcnt lgkmcnt(N): LDS returns in order.
//   * Ring of W9_NSLOT tiles requested W9_NSLOT - 1 K-steps ahead; a wave waits for ITS pieces of the next tile (counted vmcnt) in front of
//     the barrier; raw s_barrier, no vmcnt(0) in the loop.  The position of the next tile is a byte offset advanced by a constant (P and Q have
//     the same pixel grid; frames a plane skips are a jump at the clip boundary).
// Units, slabs, the X-way interleave of the planes' slabs and the scratch layout are the row-run form's (tap row = depth plane, nine taps each):
// wgrad_bf16_reduce_kernel sums the slabs.
typedef __attribute__((address_space(3))) unsigned char w9_lds_byte;
__device__ __forceinline__ void w9_dma16(__amdgpu_buffer_rsrc_t r, w9_lds_byte* dst, unsigned voffset) {      // (outside the kernel: see conv_bf16.hip dma16)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voffset, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void w9_wait_vmcnt();
template <> __device__ __forceinline__ void w9_wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void w9_wait_vmcnt<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void w9_wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void w9_wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void w9_wait_vmcnt<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void w9_wait_vmcnt<9>() { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
template <> __device__ __forceinline__ void w9_wait_vmcnt<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
// The transposing read as inline assembly: behind the builtin (no memory operand the waitcnt pass could tell from the DMAs' LDS writes) hipcc puts
// s_waitcnt vmcnt(0) in front of the first fragment read of every K-step, i.e. drains the ring.  The compiler does not know that the result
// arrives later: every use sits behind a w9_wait_* that names the registers as in/out operands (nothing can be scheduled, copied or folded
// across it) and carries the count.
template <int OFF> __device__ __forceinline__ void w9_tr(bf16x4& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
// a kernel argument the loop needs, made resident in an SGPR HERE: left to itself hipcc waits for the argument's scalar load at its first use -- inside
// the loop, with s_waitcnt lgkmcnt(0), which also drains the fragment reads in flight
__device__ __forceinline__ int w9_pin(int x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ unsigned w9_pin(unsigned x) { asm volatile("" : "+s"(x)); return x; }
struct W9Frag { bf16x4 lo, hi; };                 // rows k .. k+3 and k+4 .. k+7 of one 16-channel fragment
#define W9_TIE(f) "+v"((f).lo), "+v"((f).hi)
// wait until at most N of the LDS reads issued so far are outstanding: the fragments named have landed
template <int N> __device__ __forceinline__ void w9_wait_a_b(W9Frag (&fa)[4], W9Frag (&fb)[3]) {
  asm volatile("s_waitcnt lgkmcnt(%14)" : W9_TIE(fa[0]), W9_TIE(fa[1]), W9_TIE(fa[2]), W9_TIE(fa[3]), W9_TIE(fb[0]), W9_TIE(fb[1]), W9_TIE(fb[2]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void w9_wait_a_b(W9Frag (&fa)[2], W9Frag (&fb)[3]) {
  asm volatile("s_waitcnt lgkmcnt(%10)" : W9_TIE(fa[0]), W9_TIE(fa[1]), W9_TIE(fb[0]), W9_TIE(fb[1]), W9_TIE(fb[2]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void w9_wait_b(W9Frag (&fb)[3]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : W9_TIE(fb[0]), W9_TIE(fb[1]), W9_TIE(fb[2]) : "n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 w9_join(const W9Frag& f) { return bf16x8{f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]}; }
#ifndef W9_NSLOT
#define W9_NSLOT 4                                // ring slots; tiles are requested W9_NSLOT - 1 K-steps ahead
#endif
constexpr int W9_AHEAD = W9_NSLOT - 1;
// Block shapes: 128 x 64 (the 128 -> 128 / 64 -> 128 layers: 8 waves as 2 x 4, a wave holds 64 x 16 x nine taps) and 32 x 128 (the ResBlocks'
// 3x3 128 -> 32: 1 x 8 waves, 32 x 16 x nine taps).  Geometry of a ring slot, in 1-KB DMA pieces (64 lanes x 16 bytes, lane-linear):
template <int TA, int TB> struct W9Geo {
  static constexpr int RA = TA * 2, RB = TB * 2;                     // bytes per row of the P tile / of a row tile of Q
  static constexpr int A_BYTES = 32 * RA, BROW = 40 * RB;            // 32 positions of P; 34 pixels of Q (40 staged)
  static constexpr int SLOT = A_BYTES + 3 * BROW;
  static constexpr int PA = A_BYTES / 1024, PQ = BROW / 1024;        // pieces of P, pieces per row tile of Q
  static constexpr int NPIECE = PA + 3 * PQ;
  static constexpr int NP = (NPIECE + 7) / 8;                        // DMAs per wave and tile (pieces wave, wave + 8, ..; past NPIECE: zeros to the spare KB)
  static constexpr int LDS = W9_NSLOT * SLOT + 1024;
  static constexpr int WA = TA >= 128 ? 2 : 1, WB = 8 / WA;          // waves along P's / Q's channels
  static constexpr int MA = TA / WA / 16;                            // 16-channel tiles of P per wave (Q: one)
  static_assert(TB / WB == 16 && (TA == 128 || TA == 32) && (TB == 64 || TB == 128), "wave tiling");
};
// chunk swizzle of a row of RBYTES bytes (see the header): the 16-byte chunk at position c of row r holds chunk c ^ swz(r)
template <int RBYTES> __device__ __forceinline__ int w9_swz(int r) {
  if constexpr (RBYTES == 256) return (4 * (r & 3)) ^ (2 * ((r >> 3) & 1));
  else if constexpr (RBYTES == 128) return 2 * ((r >> 1) & 1) + 4 * ((r >> 3) & 1);
  else return 2 * ((r >> 3) & 1);                                    // 64-byte rows: four rows fill the 64 banks, rows + 8 take the other 32-byte half
}

template <int TA, int TB, bool RELU>                  // RELU: Q is the INPUT of a ReLU -> conv pair (the fragments of Q are clamped as they are used)
__global__ __launch_bounds__(512, 2) void wgrad9_bf16_kernel(const WArgs a) {
  using G = W9Geo<TA, TB>;
  constexpr int NP = G::NP, MA = G::MA, SLOT = G::SLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave / G::WB, wb = wave % G::WB;                    // wave tile: channels 16 MA wa .. of P x channels 16 wb .. of Q, nine taps
  const int l15 = lane & 15, kg = lane >> 4;

  // ---- this workgroup: tile (a, b), depth plane, slab (as the row-run form)
  int w = blockIdx.x;
  const int perTile = a.X * a.wgPerX;
  const int tile = w / perTile; w -= tile * perTile;
  const int tb = tile % a.tilesB, ta = tile / a.tilesB;
  const int xcd = w % a.X;
  int li = w / a.X, kd = 0;
  while (li >= a.mOf[kd]) { li -= a.mOf[kd]; ++kd; }
  const int slab = xcd * a.mOf[kd] + li, nslab = a.X * a.mOf[kd];
  const int tlo = max(0, d.padD - kd);
  const int Tv = min(d.T - 1, d.T - 1 + d.padD - kd) - tlo + 1;
  const int vunits = Tv > 0 ? (d.N / d.T) * Tv * d.Hm * a.runsPerRow : 0;
  const int u0 = (int)((long long)vunits * slab / nslab), u1 = (int)((long long)vunits * (slab + 1) / nslab);
  const int nt = u1 - u0;
  const int a0 = ta * TA, b0 = tb * TB;
  const bool bias_wg = tb == 0 && kd == a.biasTapRow;

  // ---- DMA roles: pieces wave, wave + 8, .. of the slot's NPIECE (P's first, then the three row tiles of Q).  A lane of a piece = (row, LDS chunk
  // position); the chunk it FETCHES is position ^ swizzle(row).  Offsets of masked lanes are 0 and their kill word is WOOB: `(base + offset) | kill`
  // is branch-free and lands at or above 2^31, out of range.  left / right: WOOB on the lane that holds the pixel left of / right of the run.
  unsigned off[NP], kill[NP], left[NP], right[NP];
  int khOf[NP], dst[NP];                                             // row tile of a Q piece (-1: a piece of P, 3: the spare), byte offset inside the slot
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int pce = wave + 8 * j;
    left[j] = right[j] = 0u;
    if (pce < G::PA) {
      constexpr int CPR = G::RA / 16, RPP = 64 / CPR;                // chunks per row, rows per piece
      const int row = RPP * pce + lane / CPR, pc = lane % CPR;
      const int c = pc ^ w9_swz<G::RA>(row);
      const bool ok = a0 + 8 * c < d.Cout;
      off[j] = ok ? (unsigned)((row * d.ldOut + a0 + 8 * c) * 2) : 0u;
      kill[j] = ok ? 0u : WOOB;
      khOf[j] = -1;
      dst[j] = pce * 1024;
    } else {
      constexpr int CPR = G::RB / 16, RPP = 64 / CPR;
      const int pq = pce - G::PA, kh = pq / G::PQ, part = pq % G::PQ;
      const int row = RPP * part + lane / CPR, pc = lane % CPR;
      const int c = pc ^ w9_swz<G::RB>(row);
      const bool ok = pce < G::NPIECE && row < 34 && b0 + 8 * c < d.Cin;
      off[j] = ok ? (unsigned)((row * d.ldIn + b0 + 8 * c) * 2) : 0u;
      kill[j] = ok ? 0u : WOOB;
      left[j] = row == 0 ? WOOB : 0u;
      right[j] = row == 33 ? WOOB : 0u;
      khOf[j] = kh;
      dst[j] = pce < G::NPIECE ? G::A_BYTES + kh * G::BROW + part * 1024 : -1;
    }
  }
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.P), 0, a.Pbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.Q), 0, a.Qbytes, 0x00020000);
  w9_lds_byte* const lds3 = (w9_lds_byte*)lds;

  // cursor = virtual unit (clip, t', y, run) of the next tile to request, as byte offsets of its first position in P and of the pixel above-left of it
  // in Q (modular: the lanes that would read in front of the tensor are killed); tiles past the slab's end are requested as zeros (constant DMA counts)
  int c_run, c_y, c_t, q = 0, qslot = 0;
  unsigned pOff, qOff;
  {
    int v = u0;
    c_run = v % a.runsPerRow; v /= a.runsPerRow;
    c_y = v % d.Hm; v /= d.Hm;
    c_t = v % max(Tv, 1);
    const int clip = v / max(Tv, 1), n = clip * d.T + tlo + c_t;
    pOff = (unsigned)((((n * d.Hm + c_y) * d.Wm + c_run * 32) * d.ldOut) * 2);
    qOff = (unsigned)(((((n + kd - d.padD) * d.Hin + c_y - 1) * d.Win + c_run * 32 - 1) * d.ldIn) * 2);
  }
  const unsigned stepP = (unsigned)(32 * d.ldOut * 2), stepQ = (unsigned)(32 * d.ldIn * 2);
  const unsigned jumpP = w9_pin((unsigned)((d.T - Tv) * d.Hm * d.Wm * d.ldOut * 2)), jumpQ = w9_pin((unsigned)((d.T - Tv) * d.Hin * d.Win * d.ldIn * 2));
  const int runs = w9_pin(a.runsPerRow), Hm = w9_pin(d.Hm), TvP = w9_pin(Tv), ntP = nt;       // (the operands of the cursor's rarely taken branches)
  const int lastRun = runs - 1, lastY = Hm - 1;
  const unsigned rowBytesQ = (unsigned)(d.Win * d.ldIn * 2);
  w9_lds_byte* slotq = lds3;                                         // the slot of tile q
  auto dma_piece = [&](auto J) {                                      // one of the NP DMAs of a tile request
    constexpr int j = decltype(J)::value;
    const bool live = q < ntP && !(FO_ABLATE_W & 1);
    if (khOf[j] < 0) {
      w9_dma16(rP, slotq + dst[j], ((live ? pOff : WOOB) + off[j]) | kill[j]);
    } else {                                                          // rows of Q: kh = 0 needs y > 0, kh = 2 needs y < H - 1 (kh = 3: the spare piece, killed anyway)
      const bool rowok = live && (khOf[j] == 0 ? c_y > 0 : (khOf[j] == 2 ? c_y < lastY : true));
      const unsigned edge = kill[j] | (left[j] & (c_run == 0 ? WOOB : 0u)) | (right[j] & (c_run == lastRun ? WOOB : 0u));
      w9_dma16(rQ, dst[j] >= 0 ? slotq + dst[j] : lds3 + W9_NSLOT * SLOT, ((rowok ? qOff + (unsigned)khOf[j] * rowBytesQ : WOOB) + off[j]) | edge);
    }
  };
  auto dma_next = [&]() {
    pOff += stepP; qOff += stepQ;
    if (++c_run == runs) { c_run = 0; if (++c_y == Hm) { c_y = 0; if (++c_t == TvP) { c_t = 0; pOff += jumpP; qOff += jumpQ; } } }
    ++q;
    qslot = qslot + 1 == W9_NSLOT ? 0 : qslot + 1;
    slotq = lds3 + qslot * SLOT;
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  auto dma_tile = [&]() {
    dma_piece(I0{}); dma_piece(I1{}); dma_piece(I2{});
    if constexpr (NP == 4) dma_piece(I3{});
    dma_next();
  };

  f32x4 acc[9][MA], accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < MA; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

  // ---- fragment addressing (see the header: same lane -> (row, 4 channels) map as the row-run form, swizzled instead of padded): LDS byte
  // addresses inside slot 0; rows k and k + 4 of P share the swizzle (same r & 3 and (r >> 3) & 1), so the upper half of a fragment is + 4 rows
  const unsigned ldsbase = (unsigned)(size_t)lds3;
  const int krow = kg * 8 + (l15 >> 2);
  unsigned fA[MA], fB[3][2];
#pragma unroll
  for (int i = 0; i < MA; ++i) fA[i] = ldsbase + krow * G::RA + (((wa * MA + i) * 32) ^ (16 * w9_swz<G::RA>(krow))) + (l15 & 3) * 8;
#pragma unroll
  for (int k = 0; k < 3; ++k)                                         // tap kw, half h: row r = krow + 4 h + kw of the 34
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = krow + 4 * h + k;
      fB[k][h] = ldsbase + r * G::RB + ((32 * wb) ^ (16 * w9_swz<G::RB>(r))) + (l15 & 3) * 8;
    }
  unsigned bm[MA];                                                    // bias: tile wb of the wave's MA (waves wb < MA)
#pragma unroll
  for (int i = 0; i < MA; ++i) bm[i] = wb == i ? ~0u : 0u;

  W9Frag fa[2][MA], fb[3][3];                                         // two sets of P's fragments (even / odd K-steps), one per row of Q's taps
  unsigned bq[3][2], aq[MA];                                          // fB, fA + a tile's slot
  auto read_a = [&](W9Frag (&f)[MA], const unsigned (&ad)[MA]) {      // 2 MA reads
#pragma unroll
    for (int i = 0; i < MA; ++i) { w9_tr<0>(f[i].lo, ad[i]); w9_tr<4 * G::RA>(f[i].hi, ad[i]); }
  };
  auto read_b = [&](W9Frag (&f)[3], auto KH) {                        // 6 reads: the three kw fragments of row kh of the tile bq points into
    constexpr int kh = decltype(KH)::value;
#pragma unroll
    for (int k = 0; k < 3; ++k) { w9_tr<G::A_BYTES + kh * G::BROW>(f[k].lo, bq[k][0]); w9_tr<G::A_BYTES + kh * G::BROW>(f[k].hi, bq[k][1]); }
  };
  auto set_aq = [&](unsigned sl) {
#pragma unroll
    for (int i = 0; i < MA; ++i) aq[i] = fA[i] + sl;
  };
  auto set_bq = [&](unsigned sl) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { bq[k][0] = fB[k][0] + sl; bq[k][1] = fB[k][1] + sl; }
  };
  auto join_b = [&](const W9Frag& f) {
    bf16x8 v = w9_join(f);
    if (RELU) {
      u32x4 u = __builtin_bit_cast(u32x4, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = relu_pk(u[e]);
      v = __builtin_bit_cast(bf16x8, u);
    }
    return v;
  };
  // the 3 MA MFMAs of one row of taps, back to back (the side work of a row -- reads, addresses, the request -- is issued in front of it: dropped
  // between the MFMAs one piece at a time it measured 5 % slower, behind the row 5 % slower still)
  auto mfma_row = [&](const W9Frag (&A)[MA], const W9Frag (&B)[3], auto KH) {
    constexpr int kh = decltype(KH)::value;
    if (FO_ABLATE_W & 4) return;
    bf16x8 va[MA], vb[3];
#pragma unroll
    for (int i = 0; i < MA; ++i) va[i] = w9_join(A[i]);
#pragma unroll
    for (int k = 0; k < 3; ++k) vb[k] = join_b(B[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < MA; ++i) acc[kh * 3 + k][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[i], vb[k], acc[kh * 3 + k][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (nt > 0) {
#pragma unroll
    for (int i = 0; i < W9_AHEAD; ++i) dma_tile();
    w9_wait_vmcnt<(W9_AHEAD - 1) * NP>();                             // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    unsigned sl = 0;                                                  // byte offset of the current tile's slot
    set_bq(0); set_aq(0);
    read_a(fa[0], aq);
    read_b(fb[0], I0{});
    read_b(fb[1], I1{});
    w9_wait_a_b<6>(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    // One K-step, PAR = its parity.  Row kh of Q lives in fb[kh], P in fa[PAR]; every fragment is requested about two rows before the row that uses it:
    //   on entry   P and row 0 have landed, row 1 is in flight
    //   row 0      in front of it: the six reads of row 2 (fb[2]: the previous step's row 2 has issued), the next slot's addresses.  Behind it this wave's
    //              pieces of the next tile are awaited and the barrier taken (everybody's pieces have landed; everybody has finished with the tile
    //              before this one, whose slot the next request overwrites)
    //   row 1      in front of it (once row 1 itself has landed): the next tile's row 0 -> fb[0] and its fragments of P -> fa[PAR ^ 1]
    //   row 2      in front of it (once row 2 has landed): the next tile's row 1 -> fb[1]; the request of tile p + W9_AHEAD
    auto kstep = [&](auto PAR) {
      constexpr int par = decltype(PAR)::value;
      read_b(fb[2], I2{});
      sl = sl + SLOT == W9_NSLOT * SLOT ? 0u : sl + SLOT;
      set_bq(sl); set_aq(sl);
      __builtin_amdgcn_sched_barrier(0);
      if (bias_wg && !(FO_ABLATE_W & 4)) {                            // column sums of P (mask arithmetic: a select between the register sets -- ?: or
        u32x4 r = {0u, 0u, 0u, 0u};                                   // branches -- hipcc turns into an indexed scratch array)
#pragma unroll
        for (int i = 0; i < MA; ++i) {
          const u32x4 x = __builtin_bit_cast(u32x4, w9_join(fa[par][i]));
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] |= x[e] & bm[i];
        }
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r), ones, accb, 0, 0, 0);
      }
      mfma_row(fa[par], fb[0], I0{});
      w9_wait_vmcnt<(W9_AHEAD - 2) * NP>();                           // this wave's pieces of the next tile (tiles p + 2 .. may fly)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      w9_wait_b<6>(fb[1]);                                             // row 1 (requested two rows ago; row 2's six may fly) -- lgkmcnt is a 4-bit count:
      __builtin_amdgcn_sched_barrier(0);                              // never more than 15 reads behind the one awaited
      read_b(fb[0], I0{});                                            // (bq: the next tile's slot)
      read_a(fa[par ^ 1], aq);
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(fa[par], fb[1], I1{});
      w9_wait_b<6 + 2 * MA>(fb[2]);                                    // row 2 (the next tile's row 0 and P may fly)
      __builtin_amdgcn_sched_barrier(0);
      read_b(fb[1], I1{});
      dma_tile();                                                     // tile p + W9_AHEAD, into the slot of tile p - 1
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(fa[par], fb[2], I2{});
      w9_wait_a_b<6>(fa[par ^ 1], fb[0]);                             // the next step's P and row 0 (its row 1 may fly)
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int p = 0; p < ntP; p += 2) {
      kstep(I0{});
      if (p + 1 >= ntP) break;
      kstep(I1{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the zero-fill DMAs of the tiles past the end, the reads requested for a step that does not exist
  }

  // ---- partial blocks -> ws[slab][tap][a][b]: accumulator register r of lane l = (a = 4 (l >> 4) + r, b = l & 15)
  if ((FO_ABLATE_W & 8) && acc[0][0][0] != 12345.f) return;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* o = a.ws + (((long long)slab * a.taps + kd * 9 + t) * a.Apad + a0 + wa * MA * 16) * a.Bpad + b0 + wb * 16;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(long long)(i * 16 + kg * 4 + r) * a.Bpad + l15] = acc[t][i][r];
  }
  if (bias_wg && l15 == 0 && wb < MA) {
    float* ob = a.wsBias + (long long)slab * a.Apad + a0 + wa * MA * 16 + wb * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) ob[kg * 4 + r] = accb[r];
  }
}

// dw[a][b][tap] = sum over the slabs (fixed order) of ws[slab][tap][a][b], a < Areal, b < Breal; work items in ws order (coalesced
// reads of the slabs, one scattered 4-byte store each); a block = 64 consecutive outputs x 4 groups of lanes that each take every
// fourth slab, two loads in flight per lane (thin layers have up to 256 slabs of a few KB: a serial walk would be latency-bound).
// smallc: ws[slab][kh][a][kw * 8 + c] -> dw[a][c][kh * 4 + kw].
// The tail of the index space sums the bias-gradient slabs: db[a] = sum_slab wsBias[slab][a].
struct RArgs {
  int taps, Apad, Bpad, Areal, Breal, smallc, X, tapsPerRow, biasTapRow;
  short mOf[32];             // slabs of tap row tr = X * mOf[tr]; tap row of tap t = t / tapsPerRow
};
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, const RArgs q,
                                                               const float* __restrict__ wsBias, float* __restrict__ db) {
  const int taps = q.taps, Apad = q.Apad, Bpad = q.Bpad, Areal = q.Areal, Breal = q.Breal, smallc = q.smallc;
  __shared__ float red[4][64];
  const long long total = (long long)Areal * Breal * taps;
  const long long slabStride = (long long)(smallc ? 4 : taps) * Apad * Bpad;
  const long long all = total + (db ? Areal : 0);
  const int li = threadIdx.x & 63, part = threadIdx.x >> 6;
  for (long long i0 = (long long)blockIdx.x * 64; i0 < all; i0 += (long long)gridDim.x * 64) {
    const long long i = i0 + li;
    const float* src = ws;
    long long stride = slabStride, dst = 0;
    bool ok = i < all, isb = false;
    int slabs = 0;
    if (ok && i >= total) {
      isb = true;
      src = wsBias + (i - total);
      stride = Apad;
      slabs = q.X * q.mOf[q.biasTapRow];
    } else if (ok) {
      const int b = (int)(i % Breal);
      const int aa = (int)((i / Breal) % Areal);
      const int tap = (int)(i / ((long long)Breal * Areal));
      src = ws + (smallc ? (((long long)(tap >> 2) * Apad + aa) * Bpad + (tap & 3) * 8 + b) : (((long long)tap * Apad + aa) * Bpad + b));
      dst = ((long long)aa * Breal + b) * taps + tap;
      slabs = q.X * q.mOf[smallc ? 0 : tap / q.tapsPerRow];
    }
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
      int k = part;
      for (; k + 4 < slabs; k += 8) { s0 += src[k * stride]; s1 += src[(k + 4) * stride]; }
      if (k < slabs) s0 += src[k * stride];
    }
    red[part][li] = s0 + s1;
    __syncthreads();
    if (part == 0 && ok) {
      const float v = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
      if (isb) db[i - total] = v; else dw[dst] = v;
    }
    __syncthreads();
  }
}

// column sums of a bf16 [rows][ld] tensor (bias gradients): one workgroup per 64 channels x slab of rows, then a fixed-order reduce
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ g, long long rows, int C, int ld, float* __restrict__ ws, int slabs) {
  __shared__ float red[4][64];
  const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;               // 8 lanes x 8 channels = 64 channels, 32 rows per pass
  const int cblk = blockIdx.x % ((C + 63) / 64), slab = blockIdx.x / ((C + 63) / 64);
  const int c = cblk * 64 + c8 * 8;
  const long long ra = rows * slab / slabs, rb = rows * (slab + 1) / slabs;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    long long r = ra + r0;
    for (; r + 96 < rb; r += 128) {                     // four rows' loads in flight per thread (one was 16 KB per CU: 2 TB/s); the adds keep their order
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(g + (r + 32 * u) * ld + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
    }
    for (; r < rb; r += 32) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(g + r * ld + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  }
  // reduce the 32 row-lanes of each channel group: within a wave (8 rows) by shuffles, across the 4 waves through LDS
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = s[e];
    v += lane_xor<8>(v); v += lane_xor<16>(v); v += lane_xor<32>(v);
    if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][c8 * 8 + e] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    ws[(long long)slab * ((C + 63) / 64 * 64) + cblk * 64 + threadIdx.x] = v;
  }
}
// db[c] = sum over the slabs of ws[slab][c]: one block per 64 channels, sixteen groups of lanes each take every sixteenth slab, four loads in
// flight (four groups x two loads walked 1024 slabs in 128 dependent steps: 32 us per bias gradient), fixed order
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ ws, float* __restrict__ db, int slabs, int Cpad, int Creal) {
  __shared__ float red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = part;
  for (; k + 48 < slabs; k += 64) {
    s0 += ws[(long long)k * Cpad + c]; s1 += ws[(long long)(k + 16) * Cpad + c];
    s2 += ws[(long long)(k + 32) * Cpad + c]; s3 += ws[(long long)(k + 48) * Cpad + c];
  }
  for (; k < slabs; k += 16) s0 += ws[(long long)k * Cpad + c];
  red[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && c < Creal) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    db[c] = t;
  }
}

struct WPlan {
  int TA, TB, WA, NKW, KR;
  bool fast, smallc, all9;   // all9: wgrad9_bf16_kernel (a workgroup = 128 x 64 x the nine taps of one depth plane; NKW = 9, tap rows = planes)
  int tilesA, tilesB, tapRows, taps, units, Apad, Bpad, biasTapRow;
  int X, wgPerX, maxSlabs, perCU;
  short mOf[32];
};

int make_plan(const fo_conv_desc* d, WPlan* p) {
  const int A = d->Cout, B = d->Cin;
  p->taps = d->KD * d->KH * d->KW;
  FO_REQUIRE(A % 8 == 0 && B % 8 == 0 && d->ldIn % 8 == 0 && d->ldOut % 8 == 0, FO_E_ALIGN, "wgrad_bf16: channel counts / strides must be multiples of 8");
  FO_REQUIRE(d->N > 0 && d->T > 0 && d->N % d->T == 0, FO_E_SHAPE, "wgrad_bf16: N must be whole clips");
  FO_REQUIRE((long long)d->N * d->Hm * d->Wm < (1ll << 31) && (long long)d->N * d->Hin * d->Win < (1ll << 31), FO_E_SHAPE, "wgrad_bf16: too many positions");
  FO_REQUIRE(p->taps <= 32, FO_E_SHAPE, "wgrad_bf16: at most 32 taps");
  // the image layers' 8-channel Q takes its own form where the geometry allows (else the gather form with 24 of 32 columns masked)
  p->smallc = B == 8 && d->KW == 4 && d->KH == 4 && d->KD == 1 && d->stride == 2 && d->ldIn == 8 && d->Wm % 32 == 0 && A % 64 == 0;
  if (p->smallc) {
    p->TA = 64; p->TB = 32; p->NKW = 4; p->fast = true;
    p->tapRows = 1;
    p->biasTapRow = 0;
  } else {
    p->fast = B >= 32 && d->Wm % 32 == 0 && ((d->KW == 4 && d->stride == 2) || ((d->KW == 3 || d->KW == 1) && d->stride == 1));
    p->NKW = p->fast ? d->KW : 1;
    p->TA = A >= 128 ? 128 : (A > 32 ? 64 : 32);
    p->TB = B >= 128 ? 128 : (B > 32 ? 64 : 32);
    if (p->NKW == 4 && p->TA == 128 && p->TB == 128) p->TB = 64;      // (accumulator budget: 4 x 128 x 64)
    if (p->TA == 32 && p->TB == 32) p->TB = 64;                       // (eight waves need eight 16 x 16 tiles)
    p->tapRows = p->fast ? d->KD * d->KH : p->taps;
    // the centre tap's row reads a real pixel of Q for EVERY position of P: its workgroups see every row of P
    p->biasTapRow = p->fast ? d->padD * d->KH + d->padH : (d->padD * d->KH + d->padH) * d->KW + d->padW;
    // 3x3 (x KD) pad-1 stride-1 filters between >= 128 and >= 64 channels: all nine taps of a plane per workgroup (FACEOFF_WGRAD_ROWS=1: the row-run form)
    const bool wide = A >= 128 && B >= 64, thin = A <= 32 && B >= 128 && getenv("FACEOFF_WGRAD_THIN_ROWS") == nullptr;   // 128 x 64 blocks; 32 x 128 (the ResBlocks' 128 -> 32)
    p->all9 = p->fast && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->padH == 1 && d->padW == 1 && d->Hm == d->Hin && d->Wm == d->Win &&
              (wide || thin) && d->padD < d->KD && getenv("FACEOFF_WGRAD_ROWS") == nullptr;
    if (p->all9) { p->TA = wide ? 128 : 32; p->TB = wide ? 64 : 128; p->NKW = 9; p->tapRows = d->KD; p->biasTapRow = d->padD; }
  }
  if (p->smallc) p->all9 = false;
  // waves: WA x (8 / WA) with at least one 16 x 16 tile per wave in either direction
  p->WA = p->all9 ? (p->TA == 128 ? 2 : 1) : (p->TA == 128 && p->TB == 128) ? 2 : (p->TA == 128 ? 4 : (p->TA == 64 ? (p->TB == 32 ? 4 : 2) : (p->TB == 128 ? 1 : 2)));
  const int perRun = p->NKW * (p->TA / p->WA / 16) * (p->TB / (8 / p->WA) / 16);      // MFMAs per wave and run
  p->KR = perRun <= 8 ? 2 : 1;                                       // thin blocks: two runs per barrier
  p->tilesA = (A + p->TA - 1) / p->TA;
  p->tilesB = p->smallc ? 1 : (B + p->TB - 1) / p->TB;
  p->Apad = p->tilesA * p->TA;
  p->Bpad = p->tilesB * p->TB;
  p->units = p->fast ? d->N * d->Hm * (d->Wm / 32) : (int)(((long long)d->N * d->Hm * d->Wm + 31) / 32);
  // ---- slabs per tap row, proportional to the positions the tap row really visits (see WArgs)
  long long work[32], total = 0;
  for (int tr = 0; tr < p->tapRows; ++tr) {
    work[tr] = p->units;
    if (p->all9) {
      const int tlo = std::max(0, d->padD - tr), thi = std::min(d->T - 1, d->T - 1 + d->padD - tr);
      work[tr] = thi >= tlo ? (long long)(d->N / d->T) * (thi - tlo + 1) * d->Hm * (d->Wm / 32) : 0;
    } else if (p->fast && !p->smallc) {
      const int kd = tr / d->KH, kh = tr % d->KH;
      const int tlo = std::max(0, d->padD - kd), thi = std::min(d->T - 1, d->T - 1 + d->padD - kd);
      const int ylo = std::max(0, (d->padH - kh + d->stride - 1) / d->stride), yhi = std::min(d->Hm - 1, (d->Hin - 1 + d->padH - kh) / d->stride);
      work[tr] = (thi >= tlo && yhi >= ylo) ? (long long)(d->N / d->T) * (thi - tlo + 1) * (yhi - ylo + 1) * (d->Wm / 32) : 0;
    }
    total += work[tr];
  }
  const int tiles = p->tilesA * p->tilesB;
  // workgroups per CU: two for the thin blocks whose LDS fits twice (wgrad_thin; FACEOFF_WGRAD_ONE_PER_CU=1: round 4's one)
  {
    const int lds = p->all9 ? (p->TA == 128 ? W9Geo<128, 64>::LDS : W9Geo<32, 128>::LDS) : wgrad_lds_bytes_rt(p->TA, p->TB, p->NKW, p->KR, p->fast, p->smallc);
    const bool one = getenv("FACEOFF_WGRAD_ONE_PER_CU") != nullptr;        // (read per call, like the other A/B switches)
    // (measured, tools/bench_wgrad_bf16.py, same device: the ResBlocks' 3x3 128 -> 32 0.152 -> 0.114 ms, the image layers 0.182 -> 0.141; the 1x1
    // forms get nothing from it -- 0.050 -> 0.054 with twice the slabs to reduce -- and keep one)
    p->perCU = (!one && !p->all9 && wgrad_two_per_cu(p->TA, p->TB, p->NKW, p->KR) && 2 * lds <= 160 * 1024) ? 2 : 1;
  }
  const int slots = p->perCU * fo_cu_count();
  const int budget = std::max(p->tapRows, slots / tiles);             // workgroups per tile: one round of the chip's resident slots
  p->X = budget >= 8 * p->tapRows ? 8 : 1;
  int sum = 0;
  for (int tr = 0; tr < p->tapRows; ++tr) {
    long long m = total > 0 ? ((long long)budget * work[tr] + (long long)p->X * total / 2) / ((long long)p->X * total) : 1;   // round(budget / X * share)
    const long long cap = work[tr] / ((long long)4 * p->KR * p->X);   // at least 4 K-steps per slab
    m = std::max<long long>(1, std::min<long long>(m, std::max<long long>(1, cap)));
    p->mOf[tr] = (short)std::min<long long>(m, 255);
    sum += p->mOf[tr];
  }
  // never a second, nearly empty round of workgroups: trim the tap rows with the least work per slab
  while ((long long)sum * p->X * tiles > slots && sum > p->tapRows) {
    int best = -1;
    for (int tr = 0; tr < p->tapRows; ++tr)
      if (p->mOf[tr] > 1 && (best < 0 || work[tr] * p->mOf[best] < work[best] * p->mOf[tr])) best = tr;
    if (best < 0) break;
    --p->mOf[best]; --sum;
  }
  p->wgPerX = sum;
  p->maxSlabs = 0;
  for (int tr = 0; tr < p->tapRows; ++tr) p->maxSlabs = std::max(p->maxSlabs, p->X * p->mOf[tr]);
  for (int tr = p->tapRows; tr < 32; ++tr) p->mOf[tr] = 0;
  return FO_OK;
}

template <int TA, int TB, int NKW, int KR, bool FAST, bool SMALLC>
constexpr int wgrad_lds_bytes() { return wgrad_lds_bytes_rt(TA, TB, NKW, KR, FAST, SMALLC); }

template <int TA, int TB, int WA, int NKW, int KR, bool FAST, bool SMALLC>
int launch_w(const WArgs& a, int grid, hipStream_t s) {       // 1 = launched, -1 = the LDS opt-in failed (fo_last_error says why)
  constexpr int ldsBytes = wgrad_lds_bytes<TA, TB, NKW, KR, FAST, SMALLC>();
  void (*kern)(const WArgs) = wgrad_bf16_kernel<TA, TB, WA, NKW, KR, FAST, SMALLC>;
  static fo_lds_once once;
  if (ldsBytes > 48 * 1024 && !fo_lds_optin(once, reinterpret_cast<const void*>(kern), ldsBytes, "wgrad_bf16")) return -1;
  FO_NOTE_T("wgrad_bf16_kernel", TA, TB, WA, NKW, KR, FAST, SMALLC);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsBytes, s, a);
  return 1;
}

template <int TA, int TB, bool RELU>
int launch_w9_t(const WArgs& a, int grid, hipStream_t s) {
  static fo_lds_once once;
  void (*kern)(const WArgs) = wgrad9_bf16_kernel<TA, TB, RELU>;
  constexpr int ldsBytes = W9Geo<TA, TB>::LDS;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(kern), ldsBytes, "wgrad9_bf16")) return -1;
  FO_NOTE_T("wgrad9_bf16_kernel", TA, TB, RELU);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsBytes, s, a);
  return 1;
}
int launch_w9(const WPlan& p, const WArgs& a, int grid, hipStream_t s) {
  if (p.TA == 128) return a.inrelu ? launch_w9_t<128, 64, true>(a, grid, s) : launch_w9_t<128, 64, false>(a, grid, s);
  return a.inrelu ? launch_w9_t<32, 128, true>(a, grid, s) : launch_w9_t<32, 128, false>(a, grid, s);
}

// (block shape, taps per workgroup) -> instantiation; KR and WA follow from them (make_plan)
template <int NKW, bool FAST>
int dispatch(const WPlan& p, const WArgs& a, int grid, hipStream_t s) {       // as launch_w; 0 = no instantiation for this block shape
  const int key = p.TA * 1000 + p.TB;
  constexpr int K2 = 2;
  if (key == 128128) return launch_w<128, 128, 2, NKW, (NKW * 4 * 2 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 128064) return launch_w<128, 64, 4, NKW, (NKW * 2 * 2 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 128032) return launch_w<128, 32, 4, NKW, (NKW * 2 * 1 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 64128) return launch_w<64, 128, 2, NKW, (NKW * 2 * 2 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 64064) return launch_w<64, 64, 2, NKW, (NKW * 2 * 1 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 64032) return launch_w<64, 32, 4, NKW, (NKW * 1 * 1 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 32128) return launch_w<32, 128, 1, NKW, (NKW * 2 * 1 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  else if (key == 32064) return launch_w<32, 64, 2, NKW, (NKW * 1 * 1 <= 8 ? K2 : 1), FAST, false>(a, grid, s);
  return 0;
}

}  // namespace

// scratch: the slabs of partial filter gradients, then the slabs of partial bias gradients
static int64_t ws_floats_main(const WPlan& p) { return (int64_t)p.maxSlabs * (p.smallc ? 4 : p.taps) * p.Apad * p.Bpad; }

extern "C" int64_t fo_wgrad_bf16_ws_bytes(const fo_conv_desc* d) {
  WPlan p;
  if (make_plan(d, &p) != FO_OK) return -1;
  return (ws_floats_main(p) + (int64_t)p.maxSlabs * p.Apad) * 4;
}

extern "C" int fo_conv_wgrad_bf16(const fo_conv_desc* d, const void* P, const void* Q, float* dw, int Areal, int Breal, float* dbias, float* ws,
                                  int64_t ws_bytes, void* stream) {
  WPlan p;
  const int rc = make_plan(d, &p);
  if (rc != FO_OK) return rc;
  FO_REQUIRE(P && Q && dw && ws && fo_aligned16(P) && fo_aligned16(Q) && fo_aligned16(ws), FO_E_ALIGN, "wgrad_bf16: 16-byte alignment");
  FO_REQUIRE(Areal >= 1 && Areal <= d->Cout && Breal >= 1 && Breal <= d->Cin, FO_E_SHAPE, "wgrad_bf16: real channel counts out of range");
  const int64_t need = fo_wgrad_bf16_ws_bytes(d);
  FO_REQUIRE(ws_bytes >= need, FO_E_WORKSPACE, "wgrad_bf16: workspace of %lld bytes, %lld needed", (long long)ws_bytes, (long long)need);
  WArgs a;
  a.d = *d;
  a.P = reinterpret_cast<const __bf16*>(P); a.Q = reinterpret_cast<const __bf16*>(Q); a.ws = ws;
  const int64_t pbytes = (int64_t)d->N * d->Hm * d->Wm * d->ldOut * 2, qbytes = (int64_t)d->N * d->Hin * d->Win * d->ldIn * 2;
  FO_REQUIRE(pbytes < (1LL << 31) && qbytes < (1LL << 31), FO_E_SHAPE, "wgrad_bf16: operands of %lld / %lld bytes (32-bit buffer offsets: < 2 GiB each)",
             (long long)pbytes, (long long)qbytes);
  a.Pbytes = (unsigned)pbytes; a.Qbytes = (unsigned)qbytes;
  a.tilesA = p.tilesA; a.tilesB = p.tilesB; a.tapRows = p.tapRows; a.taps = p.taps;
  a.X = p.X; a.wgPerX = p.wgPerX;
  for (int i = 0; i < 32; ++i) a.mOf[i] = p.mOf[i];
  a.Apad = p.Apad; a.Bpad = p.Bpad; a.units = p.units; a.runsPerRow = p.fast ? d->Wm / 32 : 1;
  a.inrelu = (d->flags & FO_IN_RELU) ? 1 : 0;
  a.biasTapRow = dbias ? p.biasTapRow : -1;
  a.wsBias = ws + ws_floats_main(p);
  const int grid = p.tilesA * p.tilesB * p.X * p.wgPerX;
  hipStream_t s = (hipStream_t)stream;
  int ok;
  if (p.all9) ok = launch_w9(p, a, grid, s);
  else if (p.smallc) ok = launch_w<64, 32, 4, 4, 2, true, true>(a, grid, s);
  else if (p.fast && p.NKW == 3) ok = dispatch<3, true>(p, a, grid, s);
  else if (p.fast && p.NKW == 4) ok = dispatch<4, true>(p, a, grid, s);
  else if (p.fast) ok = dispatch<1, true>(p, a, grid, s);
  else ok = dispatch<1, false>(p, a, grid, s);
  if (ok < 0) return FO_E_HIP;
  FO_REQUIRE(ok, FO_E_SHAPE, "wgrad_bf16: no kernel for a %d x %d block with %d taps per workgroup", p.TA, p.TB, p.NKW);
  FO_CHECK_LAUNCH();
  const long long total = (long long)Areal * Breal * p.taps + (dbias ? Areal : 0);
  const int rblocks = (int)std::min<long long>((total + 63) / 64, 16LL * fo_cu_count());
  RArgs q;
  q.taps = p.taps; q.Apad = p.Apad; q.Bpad = p.Bpad; q.Areal = Areal; q.Breal = Breal; q.smallc = p.smallc ? 1 : 0; q.X = p.X;
  q.tapsPerRow = p.fast ? p.NKW : 1; q.biasTapRow = p.biasTapRow;
  for (int i = 0; i < 32; ++i) q.mOf[i] = p.mOf[i];
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(rblocks), dim3(256), 0, s, ws, dw, q, a.wsBias, dbias);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

// db[c] = sum over the rows of g[row][c] (c < Creal), g bf16 [rows][ld]; ws: at least fo_bias_grad_bf16_ws_bytes(C) bytes
extern "C" int64_t fo_bias_grad_bf16_ws_bytes(int C) { return (int64_t)1024 * ((C + 63) / 64 * 64) * 4; }
extern "C" int fo_bias_grad_bf16(const void* g, float* db, int64_t rows, int C, int Creal, int ld, float* ws, void* stream) {
  FO_REQUIRE(g && db && ws && rows > 0 && C % 8 == 0 && ld % 8 == 0 && Creal <= C && fo_aligned16(g), FO_E_SHAPE, "bias_grad_bf16: bad arguments");
  const int cblks = (C + 63) / 64;
  // enough workgroups to stream at full rate (4 per CU), each with at least 64 rows
  const int slabs = (int)std::max<int64_t>(1, std::min<int64_t>(1024, std::min<int64_t>(rows / 64, 4 * (int64_t)fo_cu_count() / cblks)));
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cblks * slabs), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(g), (long long)rows, C, ld, ws,
                     slabs);
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cblks), dim3(1024), 0, (hipStream_t)stream, ws, db, slabs, cblks * 64, Creal);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
