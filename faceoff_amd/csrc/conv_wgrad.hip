// Filter-gradient (wgrad) kernels for gfx950 on the exact-fp32 MFMA.
//
//   dW[a][b][tap] = sum_m P[m][a] * Q[qpix(m,tap)][b]          qpix: coord = m*stride + k - pad
//
// Conv2d/Conv3d: P = grad_out (a = Cout), Q = layer input (b = Cin)      -> OIHW / OIDHW
// ConvTranspose2d k4s2: P = layer input (a = Cin), Q = grad_out (b = Cout) -> [Ci][Co][kh][kw]
// (autograd of the reference's convs under loss.backward(), train_faceoff_perceptual.py:100).
//
// The contraction runs over pixels, which is the row index of both channels-last operands, so
// the LDS images are the natural [pixel][channel] rows and the MFMA fragments are conflict-free
// ds_read_b32 row reads (lane -> consecutive channel).  K (pixels) is split across workgroups:
// each writes an fp32 partial slab [tap][a][b]; a second kernel sums the slabs in a fixed order
// (bitwise reproducible, no atomics) and scatters into the checkpoint layout.  Temporal taps that
// only see clip padding are skipped per frame.  The conv bias gradient (column sums of P) rides
// along in the centre-tap workgroups.
//
// A K-step is 32 consecutive pixels.  When the image width is a multiple of 32 (every layer of the
// 256x256 model) a step is a run inside one image row, so for a fixed tap the shifted Q rows are a
// linear run as well: the step's base offsets and a 32-bit row-validity mask are wave-uniform
// SCALAR arithmetic (FASTROW), and a thread's share is one add per load.  Other shapes take the
// generic per-row walk.  Either way the next step's loads go out inside MFMA group 0, the cursor
// advances inside group 1, the LDS stores happen inside group 3 (1 MFMA : a few side instructions).
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

struct WgradArgs {
  fo_conv_desc d;
  const float* P;
  const float* Q;
  float* ws;      // [nchunks][taps][Apad][Bpad]
  float* wsBias;  // [nchunks][Apad] or null
  int M, HWm;
  int chunk;      // pixels per chunk (multiple of 32)
  int nchunks;
  int taps;       // tap groups iterated by blocks (SMALLC: KH)
  int tilesA, tilesB;
  int Apad, Bpad;
  int biasTap;    // tap whose blocks also produce the bias partial (-1: none)
  int stepFrameAligned;  // HWm % 32 == 0
  unsigned pBytes, qBytes;  // addressable extents behind P and Q (buffer descriptor bounds)
  int qMargin;              // bytes Q's descriptor starts below Q (see the kernel)
  int banks;                // > 1: the pixel axis is `banks` independent planes of N/banks frames; chunks never straddle a
  int cpp;                  //      plane (cpp chunks per plane) and the slabs of a plane reduce to their own dW (Winograd)
  int fastWalk;             // Wm >= 32: a 32-pixel step wraps at most one image row
  int strideShift;          // log2(stride) (stride is 1 or 2)
  int kdLoop;               // Conv3d: a workgroup owns (chunk, kh, kw) and runs the KD depth taps one after another
  int tapsSlab;             // taps per chunk in the slab layout (= KD*KH*KW; `taps` counts workgroups per chunk)
};

constexpr int WK = 32;  // pixels per K-step
constexpr unsigned OOB = 0x80000000u;
#ifdef FO_STAMP   // diagnostic build only (tools/): s_memtime of workgroup 0 / wave 0
__device__ unsigned long long fo_wstamps[4096];
#define FO_WSTAMP_AT(i)                                                                           \
  if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 4096) {                                        \
    unsigned long long t_;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    fo_wstamps[(i)] = t_;                                                                         \
  }
#else
#define FO_WSTAMP_AT(i)
#endif

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned off, int soff = 0) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0));
}

template <int TA, int TB, int WAVES_A, int WAVES_B, int TMA, int TNB, bool SMALLC, bool FASTROW, bool INRELU>
__global__ __launch_bounds__(64 * WAVES_A * WAVES_B, 2) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int NT = 64 * WAVES_A * WAVES_B;
  static_assert(WAVES_A * TMA * 32 == TA && WAVES_B * TNB * 32 == TB, "tile");
  static_assert(!(SMALLC && FASTROW), "FASTROW is for the regular channel layout");
  static_assert(FASTROW || !INRELU, "the input ReLU is a template parameter on the FASTROW path, a runtime floor elsewhere");
  constexpr int PA = (WK * TA / 4) / NT;  // float4 loads per thread per step for P
  constexpr int PB = (WK * TB / 4) / NT;
  constexpr int RPA = NT / (TA / 4);      // rows covered per pass
  constexpr int RPB = NT / (TB / 4);
  static_assert(PA >= 1 && PB >= 1 && PA <= 4 && PB <= 4, "loads per step must fit the 4 parts");
  __shared__ __attribute__((aligned(16))) float lds[2 * WK * (TA + TB)];
  float* Ps0 = lds;
  float* Qs0 = lds + 2 * WK * TA;

  const fo_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wa = wave / WAVES_B, wb = wave % WAVES_B;

  // logical block -> (chunk, tap, tileA, tileB); taps of one chunk are neighbours on one XCD
  int logical = fo_xcd_remap(blockIdx.x, gridDim.x);
  const int tileB = logical % a.tilesB; logical /= a.tilesB;
  const int tileA = logical % a.tilesA; logical /= a.tilesA;
  const int tap = logical % a.taps;
  const int chunk = logical / a.taps;

  int kd, kh, kw;
  const int khw = d.KH * d.KW;
  if (SMALLC) { kd = 0; kh = tap; kw = 0; }
  else if (a.kdLoop) { kd = 0; kh = tap / d.KW; kw = tap - kh * d.KW; }
  else { kd = tap / khw; const int r = tap - kd * khw; kh = r / d.KW; kw = r - kh * d.KW; }

  int m_begin = chunk * a.chunk;
  int m_end = min(a.M, m_begin + a.chunk);
  if (a.kdLoop) {
    // Chunk boundaries by equal WORK: frame t is visited by w_t = #{kd : 0 <= t + kd - padD < T} of the depth passes
    // (the others see clip padding and are skipped), so equal pixel counts would give workgroups of unequal length --
    // and these workgroups are half the launch long, so the longest one sets the launch time.
    auto boundary = [&](int k) -> int {
      if (k >= a.nchunks) return a.M;
      int wclip = 0;
      for (int t = 0; t < d.T; ++t) wclip += min(d.KD, d.T - t + d.padD) - max(0, d.padD - t);
      const long long Wclip = (long long)wclip * a.HWm;
      if (a.banks > 1) {   // per plane: cpp equal-work chunks of its N/banks frames
        const int bank = k / a.cpp, kk = k - bank * a.cpp;
        const int planeM = a.M / a.banks;
        const long long target = Wclip * (d.N / a.banks / d.T) * kk / a.cpp;
        const int clip = (int)(target / Wclip);
        long long rem = target - (long long)clip * Wclip;
        int t = 0, px = 0;
        for (; t < d.T; ++t) {
          const int wt = min(d.KD, d.T - t + d.padD) - max(0, d.padD - t);
          if (rem < (long long)wt * a.HWm) { px = (int)(rem / wt); break; }
          rem -= (long long)wt * a.HWm;
        }
        return bank * planeM + min(planeM, ((clip * d.T + t) * a.HWm + px) & ~31);
      }
      const long long target = Wclip * (d.N / d.T) * k / a.nchunks;
      const int clip = (int)(target / Wclip);
      long long rem = target - (long long)clip * Wclip;
      int t = 0, px = 0;
      for (; t < d.T; ++t) {
        const int wt = min(d.KD, d.T - t + d.padD) - max(0, d.padD - t);
        if (rem < (long long)wt * a.HWm) { px = (int)(rem / wt); break; }
        rem -= (long long)wt * a.HWm;
      }
      const int m = (clip * d.T + t) * a.HWm + px;
      return min(a.M, m & ~31);
    };
    m_begin = boundary(chunk);
    m_end = boundary(chunk + 1);
  }

  const int pcolA = (tid % (TA / 4)) * 4, prowA = tid / (TA / 4);
  const int pcolB = (tid % (TB / 4)) * 4, prowB = tid / (TB / 4);
  // bounds-checked buffer loads: padded taps / chunk tails read zeros without a branch or a select,
  // so the loads stay in flight under the MFMAs (see conv_igemm.hip)
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, a.pBytes, 0x00020000);
  // Q's descriptor starts qMargin bytes below Q so that the scalar per-step offset (first pixel of a tap's row run,
  // which may lie in the left / top padding) is never negative; padding pixels themselves are never fetched.
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.Q) - a.qMargin), 0, a.qBytes + a.qMargin, 0x00020000);
  const float relu_floor = (d.flags & FO_IN_RELU) ? 0.f : -INFINITY;
  bool do_bias = a.wsBias && tap == a.biasTap && tileB == 0;

  f32x4 rp[PA], rq[PB];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  // ------------------------------------------------------------------ cursor over the K-steps to load
  // wave-uniform: first pixel of the step, its frame / row / column, and whether it exists
  int c_pos = m_begin, c_n, c_t, c_y, c_x;
  auto cursor_reset = [&]() {
    c_pos = m_begin;
    c_n = c_pos / a.HWm;
    const int rem = c_pos - c_n * a.HWm;
    c_y = rem / d.Wm;
    c_x = rem - c_y * d.Wm;
    c_t = c_n % d.T;
  };
  cursor_reset();
  const bool can_skip = d.KD > 1 && a.stepFrameAligned;
  auto frame_ok = [&]() { return !can_skip || (unsigned)(c_t + kd - d.padD) < (unsigned)d.T; };
  auto skip_bad_frames = [&]() {   // jump to the start of the next frame while the tap only sees clip padding
    while (c_pos < m_end && !frame_ok()) {
      c_pos += a.HWm - (c_y * d.Wm + c_x);
      c_x = 0; c_y = 0; ++c_n;
      c_t = (c_t + 1 == d.T) ? 0 : c_t + 1;
    }
  };
  // generic walk: per-thread (frame, y, x) of each Q row
  int qn[PB], qy[PB], qx[PB], qt[PB];
  auto decode_rows = [&]() {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int m = c_pos + prowB + RPB * i;
      qn[i] = m / a.HWm;
      const int rem = m - qn[i] * a.HWm;
      qy[i] = rem / d.Wm;
      qx[i] = rem - qy[i] * d.Wm;
      qt[i] = qn[i] % d.T;
    }
  };
  auto cursor_advance = [&]() {
    const int before = c_pos;
    c_pos += WK;
    c_x += WK;
    if (FASTROW) {
      if (c_x == d.Wm) {
        c_x = 0;
        if (++c_y == d.Hm) { c_y = 0; ++c_n; c_t = (c_t + 1 == d.T) ? 0 : c_t + 1; }
      }
    } else {
      while (c_x >= d.Wm) {
        c_x -= d.Wm;
        if (++c_y == d.Hm) { c_y = 0; ++c_n; c_t = (c_t + 1 == d.T) ? 0 : c_t + 1; }
      }
    }
    skip_bad_frames();
    if (!FASTROW) {
      if (a.fastWalk && c_pos == before + WK) {   // one wrap check per row instead of two divisions
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          qx[i] += WK;
          const bool c = qx[i] >= d.Wm;
          qx[i] -= c ? d.Wm : 0;
          qy[i] += c ? 1 : 0;
          const bool c2 = qy[i] >= d.Hm;
          qy[i] = c2 ? 0 : qy[i];
          qn[i] += c2 ? 1 : 0;
          qt[i] = c2 ? ((qt[i] + 1 == d.T) ? 0 : qt[i] + 1) : qt[i];
        }
      } else {
        decode_rows();
      }
    }
  };

  // per-thread constants of the FASTROW address arithmetic
  unsigned pconst[PA], qconst[PB], qrow[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) pconst[i] = (unsigned)(((prowA + RPA * i) * d.ldOut + tileA * TA + pcolA) * 4);
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    qconst[i] = (unsigned)((((prowB + RPB * i) << a.strideShift) * d.ldIn + tileB * TB + pcolB) * 4);
    qrow[i] = prowB + RPB * i;
  }

  // One quarter of a step's loads: P row q and Q row q of this thread (sits between two MFMA bursts).
  // Past the last step every offset is out of range: zeros, no branch.
  auto load_part = [&](int q) {
    const unsigned endpad = c_pos < m_end ? 0u : OOB;
    if (FASTROW) {
      // Every per-step quantity is a wave-uniform scalar and rides in the loads' scalar offset; a thread's share is a
      // constant VGPR (P) or that constant with the padding bit of its pixel (Q): the fp32 MFMA runs on the SIMD's
      // fp32 ALUs, so VALU work in this loop comes straight out of the matrix rate.  Past the chunk's last step the
      // P loads re-read a valid row (their data is never used), the Q loads are all padding.
      if (q < PA) rp[q < PA ? q : 0] = bufload(rP, pconst[q < PA ? q : 0], min(c_pos, a.M - WK) * d.ldOut * 4);
      if (q < PB) {
        const int i = q < PB ? q : 0;
        // scalar: input row / first input column of the step, row-validity mask over its 32 pixels
        const int iy = (c_y << a.strideShift) - d.padH + kh;
        const int xs = (c_x << a.strideShift) - d.padW + kw;
        const int lo = max(0, (-xs + d.stride - 1) >> a.strideShift);
        const int hi = min(WK, (d.Win - xs + d.stride - 1) >> a.strideShift);
        unsigned mask = 0;
        if ((unsigned)iy < (unsigned)d.Hin && hi > lo && c_pos < m_end)
          mask = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        const int qbase = (((c_n + kd - d.padD) * d.Hin + iy) * d.Win + xs) * d.ldIn * 4 + a.qMargin;
        rq[i] = bufload(rQ, ((~mask >> qrow[i]) << 31) | qconst[i], c_pos < m_end ? qbase : 0);
      }
    } else {
      if (q < PA) {
        const int m = c_pos + prowA + RPA * q;
        const unsigned pad = (m < m_end ? 0u : OOB) | endpad;
        rp[q < PA ? q : 0] = bufload(rP, (((unsigned)m * (unsigned)d.ldOut + (unsigned)(tileA * TA + pcolA)) * 4u) | pad);
      }
      if (q < PB) {
        const int i = q < PB ? q : 0;
        const int n = qn[i], y = qy[i], x = qx[i];
        int kwt = kw, coff = tileB * TB + pcolB;
        if (SMALLC) { kwt = pcolB >> 3; coff = pcolB & 7; }
        const int it = qt[i] + kd - d.padD;
        const int iy = y * d.stride - d.padH + kh;
        const int ix = x * d.stride - d.padW + kwt;
        const bool ok = (c_pos + prowB + RPB * i < m_end) & ((unsigned)it < (unsigned)d.T) & ((unsigned)iy < (unsigned)d.Hin) &
                        ((unsigned)ix < (unsigned)d.Win);
        const unsigned pix = (unsigned)(((n + kd - d.padD) * d.Hin + iy) * d.Win + ix);
        rq[i] = bufload(rQ, ((pix * (unsigned)d.ldIn + (unsigned)coff) * 4u + (unsigned)a.qMargin) | (ok ? 0u : OOB) | endpad);
      }
    }
  };
  auto store_part = [&](int q, int buf, auto bias_c, bool real) {   // real: the rows belong to the chunk (not the overshoot prefetch)
    constexpr bool BIAS = decltype(bias_c)::value;
    float* Ps = Ps0 + buf * WK * TA;
    float* Qs = Qs0 + buf * WK * TB;
    if (q < PA) {
      const int i = q < PA ? q : 0;
      *reinterpret_cast<f32x4*>(Ps + (prowA + RPA * i) * TA + pcolA) = rp[i];
      if (BIAS && real) bsum += rp[i];
    }
    if (q < PB) {
      const int i = q < PB ? q : 0;
      f32x4 v = rq[i];
      if (FASTROW) {
        if (INRELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      } else {
        v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
      }
      *reinterpret_cast<f32x4*>(Qs + (prowB + RPB * i) * TB + pcolB) = v;
    }
  };

  // Conv3d (kdLoop): a workgroup keeps its (chunk, kh, kw) and runs the depth taps one after another, one slab each.
  // Every workgroup of the launch then does the same work (chunks are cut by work, above), and the KH*KW workgroups
  // of a chunk walk the same P / Q rows at the same time (L2 reuse) -- per-tap workgroups would differ in length by
  // T/(T-1), and cutting their chunks differently per tap loses the shared walk.
  const int npass = a.kdLoop ? d.KD : 1;
  const int tap2d = tap;
  f32x16 acc[TMA][TNB];
  for (int pass = 0; pass < npass; ++pass) {
  int tapFull = tap2d;
  if (a.kdLoop) {
    kd = pass;
    tapFull = pass * khw + tap2d;
    do_bias = a.wsBias && tapFull == a.biasTap && tileB == 0;
    cursor_reset();
  }
#pragma unroll
  for (int i = 0; i < TMA; ++i)
#pragma unroll
    for (int j = 0; j < TNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // the K loop, compiled twice: the workgroups of one tap per chunk also sum P's columns (the conv bias gradient)
  auto run_pass = [&](auto bias_c) {
  skip_bad_frames();
  if (!FASTROW) decode_rows();
  bool have = c_pos < m_end;          // a step is staged in LDS buffer `cur`
#pragma unroll
  for (int q = 0; q < 4; ++q) load_part(q);
#pragma unroll
  for (int q = 0; q < 4; ++q) store_part(q, 0, bias_c, have);
  if (have) cursor_advance();
  __syncthreads();
  int cur = 0;
  int stamp_i = 0;
  while (have) {
    FO_WSTAMP_AT(8 * stamp_i);
    const bool have_next = c_pos < m_end;
    // A wave's 32-channel tiles are interleaved with the other waves' (tile i of wave wa = channels (i*WAVES_A + wa)*32),
    // so every fragment of a step sits a multiple of 256 B from one per-lane base: ds_read2st64_b32 reaches them all
    // with immediate offsets (no VALU add per pair -- VALU time comes out of the fp32 MFMA rate).
    const float* Ps = Ps0 + cur * WK * TA + half * TA + wa * 32 + l31;
    const float* Qs = Qs0 + cur * WK * TB + half * TB + wb * 32 + l31;
    // 4 groups of 4 k-pairs.  Fragments of group g+1 are requested inside group g, the next step's
    // loads go out inside group 0, the cursor moves on inside group 1, the LDS stores of the loaded
    // data happen inside group 3 -- each piece between two MFMAs (sched_group_barrier pins
    // 1 MFMA : a few side instructions; hipcc would otherwise cluster the side work in front).
    float fa[2][4][TMA], fb[2][4][TNB];
    FO_WSTAMP_AT(8 * stamp_i + 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < TMA; ++i) fa[0][q][i] = Ps[q * 2 * TA + i * WAVES_A * 32];
#pragma unroll
      for (int j = 0; j < TNB; ++j) fb[0][q][j] = Qs[q * 2 * TB + j * WAVES_B * 32];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (g < 3) {   // fragments of k-pair q of the next group
#pragma unroll
          for (int i = 0; i < TMA; ++i) fa[(g + 1) & 1][q][i] = Ps[((g + 1) * 4 + q) * 2 * TA + i * WAVES_A * 32];
#pragma unroll
          for (int j = 0; j < TNB; ++j) fb[(g + 1) & 1][q][j] = Qs[((g + 1) * 4 + q) * 2 * TB + j * WAVES_B * 32];
        }
        if (g == 0) load_part(q);
        if (g == 1 && q == 0 && have_next) cursor_advance();
        if (g == 3) store_part(q, cur ^ 1, bias_c, have_next);
#pragma unroll
        for (int i = 0; i < TMA; ++i)
#pragma unroll
          for (int j = 0; j < TNB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][q][i], fb[g & 1][q][j], acc[i][j], 0, 0, 0);
        // masks: 0x8 MFMA, 0x2 VALU, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
#pragma unroll
        for (int u = 0; u < TMA * TNB; ++u) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, (TMA + TNB + TMA * TNB - 1) / (TMA * TNB), 0);
          if (g == 0) {
            __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
            if (u == TMA * TNB - 1) __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
          }
          if (g == 3) {
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            if (u >= TMA * TNB - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
      }
      FO_WSTAMP_AT(8 * stamp_i + 2 + g);
    }
    __builtin_amdgcn_sched_barrier(0);
    FO_WSTAMP_AT(8 * stamp_i + 6);
    __syncthreads();
    cur ^= 1;
    have = have_next;
    ++stamp_i;
  }
  };   // run_pass
  if (do_bias) run_pass(std::true_type{});
  else run_pass(std::false_type{});

  // ---- partial slab [chunk][tap][Apad][Bpad]
  float* slab = a.ws + ((size_t)chunk * a.tapsSlab + tapFull) * a.Apad * a.Bpad;
#pragma unroll
  for (int i = 0; i < TMA; ++i)
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
      const int col = tileB * TB + (j * WAVES_B + wb) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tileA * TA + (i * WAVES_A + wa) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        slab[(size_t)row * a.Bpad + col] = acc[i][j][r];
      }
    }

  if (do_bias) {  // column sums of P over this chunk: reduce the RPA threads sharing a column group
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    red[tid] = bsum;
    __syncthreads();
    if (tid < TA / 4) {
      f32x4 t = red[tid];
      for (int r = 1; r < RPA; ++r) t += red[tid + r * (TA / 4)];
      *reinterpret_cast<f32x4*>(a.wsBias + (size_t)chunk * a.Apad + tileA * TA + tid * 4) = t;
    }
    __syncthreads();
  }
  }  // pass
}

// Sum the slabs in chunk order and scatter to the checkpoint layout dW[a][b][tap].
// A workgroup owns 64 consecutive slab elements; its CL "chunk lanes" (blockDim = 64*CL) each sum every
// CL-th slab with four independent accumulators (loads in flight instead of a latency chain), then
// the lanes are combined through LDS in a fixed order: bitwise reproducible, no atomics.
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nchunks, int taps,
                                    int Apad, int Bpad, int Areal, int Breal, int smallc, int KW) {
  __shared__ float red[1024];
  const size_t slabElems = (size_t)taps * Apad * Bpad;
  // blockIdx.y = bank: its nchunks slabs follow each other in ws, its dW follows the previous bank's
  ws += (size_t)blockIdx.y * nchunks * slabElems;
  dw += (size_t)blockIdx.y * Areal * Breal * taps * (smallc ? KW : 1);
  const int CL = blockDim.x >> 6, cl = threadIdx.x >> 6, li = threadIdx.x & 63;
  const size_t e = (size_t)blockIdx.x * 64 + li;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < slabElems) {
    const float* p = ws + e;
    int c = cl;
    for (; c + 3 * CL < nchunks; c += 4 * CL) {
      s0 += p[(size_t)c * slabElems];
      s1 += p[(size_t)(c + CL) * slabElems];
      s2 += p[(size_t)(c + 2 * CL) * slabElems];
      s3 += p[(size_t)(c + 3 * CL) * slabElems];
    }
    for (; c < nchunks; c += CL) s0 += p[(size_t)c * slabElems];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cl == 0 && e < slabElems) {
    float s = red[li];
    for (int k = 1; k < CL; ++k) s += red[k * 64 + li];
    const int b = e % Bpad;
    const int aidx = (e / Bpad) % Apad;
    const int tap = e / ((size_t)Bpad * Apad);
    int breal = b, tapOut = tap, tapsOut = taps;
    if (smallc) { breal = b & 7; tapOut = tap * KW + (b >> 3); tapsOut = taps * KW; }
    if (aidx < Areal && breal < Breal) dw[((size_t)aidx * Breal + breal) * tapsOut + tapOut] = s;
  }
}

__global__ void bias_reduce_kernel(const float* __restrict__ ws, float* __restrict__ db, int nchunks, int Apad, int Areal) {
  __shared__ float red[1024];
  const int CL = blockDim.x >> 6, cl = threadIdx.x >> 6, li = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + li;
  float s0 = 0.f, s1 = 0.f;
  if (c < Areal) {
    int k = cl;
    for (; k + CL < nchunks; k += 2 * CL) { s0 += ws[(size_t)k * Apad + c]; s1 += ws[(size_t)(k + CL) * Apad + c]; }
    for (; k < nchunks; k += CL) s0 += ws[(size_t)k * Apad + c];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (cl == 0 && c < Areal) {
    float s = red[li];
    for (int k = 1; k < CL; ++k) s += red[k * 64 + li];
    db[c] = s;
  }
}

struct Plan {
  int TA, TB, tilesA, tilesB, Apad, Bpad, taps, chunk, nchunks, kdLoop, tapsSlab, cpp;
  bool smallc;
};

int make_plan(const fo_conv_desc* d, Plan* p, int banks = 1) {
  const int A = d->Cout, B = d->Cin;  // channels of P (a) and Q (b)
  p->smallc = B < 32;
  if (p->smallc) {
    FO_REQUIRE(B == 8 && d->KW == 4 && d->KD == 1, FO_E_SHAPE, "wgrad: small Q channels need Cb==8, KW==4 (got %d,%d)", B, d->KW);
    FO_REQUIRE(A == 64, FO_E_SHAPE, "wgrad: small-channel path needs Ca==64 (got %d)", A);
    p->TA = 64; p->TB = 32; p->taps = d->KH; p->tilesA = 1; p->tilesB = 1;
  } else {
    FO_REQUIRE(A % 32 == 0 && B % 32 == 0, FO_E_SHAPE, "wgrad: channels must be multiples of 32 (a=%d b=%d)", A, B);
    p->TA = A % 128 == 0 ? 128 : (A % 64 == 0 ? 64 : 32);
    p->TB = B % 128 == 0 ? 128 : (B % 64 == 0 ? 64 : 32);
    p->taps = d->KD * d->KH * d->KW;
    p->tilesA = A / p->TA; p->tilesB = B / p->TB;
  }
  p->tapsSlab = p->taps;
  p->kdLoop = 0;
  if (!p->smallc && ((d->KD > 1 && d->T > 1 && !getenv("FACEOFF_NO_KDLOOP")) || banks > 1) && (d->Hm * d->Wm) % 32 == 0) {
    p->kdLoop = 1;                 // workgroups enumerate (chunk, kh, kw) and loop over kd; chunks cut by work
    p->taps = d->KH * d->KW;
  }
  p->Apad = p->TA * p->tilesA; p->Bpad = p->TB * p->tilesB;
  const long long M = (long long)d->N * d->Hm * d->Wm;
  FO_REQUIRE(M > 0 && M < (1ll << 31), FO_E_SHAPE, "wgrad: M out of range");
  // chunking: the grid must fill the chip's resident workgroup slots a whole number of times -- these
  // workgroups are long (hundreds of K-steps), so 2052 of them on 512 slots would run 5 rounds where
  // 4.008 are needed.  Size the grid to just under two full rounds.
  const long long perChunkBlocks = (long long)p->taps * p->tilesA * p->tilesB;
  const long long ldsBytes = 2ll * WK * (p->TA + p->TB) * 4;
  const long long perCU = std::max(1ll, std::min(4ll, (160ll * 1024) / ldsBytes));
  const long long slots = (long long)fo_cu_count() * perCU;
  long long want = std::max(1ll, (2 * slots) / perChunkBlocks);
  want = std::min(want, std::max(1ll, M / 512));   // >= 16 K-steps per workgroup
  long long chunk = (M + want - 1) / want;
  chunk = (chunk + 31) / 32 * 32;
  p->chunk = (int)chunk;
  p->nchunks = (int)((M + chunk - 1) / chunk);
  p->cpp = p->nchunks;
  if (banks > 1) {   // chunks per plane: the same total number of workgroups, never straddling a plane
    FO_REQUIRE(p->kdLoop && d->N % (banks * d->T) == 0, FO_E_SHAPE, "wgrad: banks need Hm*Wm %% 32 == 0 and whole clips per plane");
    p->cpp = std::max(1, p->nchunks / banks);
    p->nchunks = p->cpp * banks;
  }
  return FO_OK;
}

}  // namespace

#ifdef FO_STAMP
extern "C" int fo_debug_read_wstamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fo_wstamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

static int64_t wgrad_ws_bytes(const fo_conv_desc* d, int banks) {
  Plan p;
  if (make_plan(d, &p, banks) != FO_OK) return -1;
  const int64_t tiled = ((int64_t)p.nchunks * p.tapsSlab * p.Apad * p.Bpad + (int64_t)p.nchunks * p.Apad) * 4 + 256;
  return banks == 1 ? std::max(std::max(tiled, fo_wgrad_img_ws_bytes(d)), fo_resblock_wgrad1_halo_ws_bytes(d)) : tiled;
}

extern "C" int64_t fo_wgrad_ws_bytes(const fo_conv_desc* d) { return wgrad_ws_bytes(d, 1); }
extern "C" int64_t fo_wgrad_banked_ws_bytes(const fo_conv_desc* d, int banks) { return wgrad_ws_bytes(d, banks); }

#define WG_LAUNCH1(TA_, TB_, WA_, WB_, TM_, TN_, SC_, FR_, IR_)                                                       \
  do {                                                                                                                \
    FO_NOTE_T("conv_wgrad_kernel", TA_, TB_, WA_, WB_, TM_, TN_, SC_, FR_, IR_);                                       \
    hipLaunchKernelGGL((conv_wgrad_kernel<TA_, TB_, WA_, WB_, TM_, TN_, SC_, FR_, IR_>), dim3(grid), dim3(64 * WA_ * WB_), \
                       0, s, a);                                                                                      \
  } while (0)
#define WG_LAUNCH(TA_, TB_, WA_, WB_, TM_, TN_)                     \
  do {                                                               \
    if (fastrow && inrelu) WG_LAUNCH1(TA_, TB_, WA_, WB_, TM_, TN_, false, true, true); \
    else if (fastrow) WG_LAUNCH1(TA_, TB_, WA_, WB_, TM_, TN_, false, true, false); \
    else WG_LAUNCH1(TA_, TB_, WA_, WB_, TM_, TN_, false, false, false);     \
  } while (0)

static int conv_wgrad_impl(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal,
                           float* dbias, float* ws, int64_t ws_bytes, void* stream, int banks) {
  Plan p;
  int rc = make_plan(d, &p, banks);
  if (rc != FO_OK) return rc;
  FO_REQUIRE(ws_bytes >= wgrad_ws_bytes(d, banks), FO_E_WORKSPACE, "wgrad: workspace too small");
  FO_REQUIRE(banks == 1 || !dbias, FO_E_SHAPE, "wgrad: no bias sum with banks");
  FO_REQUIRE(fo_aligned16(P) && fo_aligned16(Q) && fo_aligned16(ws) && d->ldIn % 4 == 0 && d->ldOut % 4 == 0, FO_E_ALIGN,
             "wgrad: operands must be 16-byte aligned with ld %% 4 == 0");
  if (banks == 1 && fo_wgrad_img_try(d, P, Q, dw, Areal, Breal, dbias, ws, ws_bytes, (hipStream_t)stream) == 0) return FO_OK;
  if (banks == 1 && fo_resblock_wgrad1_halo_try(d, P, Q, dw, Areal, Breal, dbias, ws, ws_bytes, (hipStream_t)stream)) {
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  WgradArgs a;
  a.d = *d; a.P = P; a.Q = Q; a.ws = ws;
  a.HWm = d->Hm * d->Wm;
  a.M = d->N * a.HWm;
  a.chunk = p.chunk; a.nchunks = p.nchunks; a.taps = p.taps;
  a.kdLoop = p.kdLoop; a.tapsSlab = p.tapsSlab;
  a.banks = banks; a.cpp = p.cpp;
  a.tilesA = p.tilesA; a.tilesB = p.tilesB; a.Apad = p.Apad; a.Bpad = p.Bpad;
  const size_t slab = (size_t)p.nchunks * p.tapsSlab * p.Apad * p.Bpad;
  a.wsBias = dbias ? ws + ((slab + 63) / 64) * 64 : nullptr;
  a.biasTap = p.smallc ? 0 : d->padD * d->KH * d->KW;
  a.stepFrameAligned = (a.HWm % WK) == 0 && (p.chunk % WK) == 0;
  const unsigned long long pBytes = (((unsigned long long)a.M - 1) * d->ldOut + d->Cout) * 4ull;
  const unsigned long long qBytes = (((unsigned long long)d->N * d->Hin * d->Win - 1) * d->ldIn + d->Cin) * 4ull;
  FO_REQUIRE(pBytes < (1ull << 31) && qBytes < (1ull << 31), FO_E_SHAPE, "wgrad: tensor exceeds the 2 GiB buffer-descriptor window");
  a.pBytes = (unsigned)pBytes;
  a.fastWalk = d->Wm >= WK;
  FO_REQUIRE(d->stride == 1 || d->stride == 2, FO_E_SHAPE, "wgrad: stride must be 1 or 2");
  a.strideShift = d->stride == 2 ? 1 : 0;
  // FASTROW: a 32-pixel K-step is a run inside one image row, chunks start on step boundaries
  const bool fastrow = !p.smallc && d->Wm % WK == 0 && p.chunk % WK == 0;
  a.qBytes = (unsigned)qBytes;
  const long long qMargin = (((long long)d->padD * d->Hin + d->padH) * d->Win + d->padW) * d->ldIn * 4ll;
  FO_REQUIRE(qBytes + (unsigned long long)qMargin < (1ull << 31), FO_E_SHAPE, "wgrad: tensor exceeds the 2 GiB buffer-descriptor window");
  a.qMargin = (int)qMargin;
  hipStream_t s = (hipStream_t)stream;
  const int grid = p.nchunks * p.taps * p.tilesA * p.tilesB;
  const bool inrelu = (d->flags & FO_IN_RELU) != 0;
  if (p.smallc) WG_LAUNCH1(64, 32, 2, 1, 1, 1, true, false, false);
  else if (p.TA == 128 && p.TB == 128) WG_LAUNCH(128, 128, 2, 2, 2, 2);
  else if (p.TA == 128 && p.TB == 64) WG_LAUNCH(128, 64, 2, 2, 2, 1);
  else if (p.TA == 64 && p.TB == 128) WG_LAUNCH(64, 128, 2, 2, 1, 2);
  else if (p.TA == 128 && p.TB == 32) WG_LAUNCH(128, 32, 4, 1, 1, 1);
  else if (p.TA == 32 && p.TB == 128) WG_LAUNCH(32, 128, 1, 4, 1, 1);
  else if (p.TA == 64 && p.TB == 64) WG_LAUNCH(64, 64, 2, 2, 1, 1);
  else if (p.TA == 64 && p.TB == 32) WG_LAUNCH(64, 32, 2, 1, 1, 1);
  else if (p.TA == 32 && p.TB == 64) WG_LAUNCH(32, 64, 1, 2, 1, 1);
  else if (p.TA == 32 && p.TB == 32) WG_LAUNCH(32, 32, 1, 1, 1, 1);
  else FO_REQUIRE(false, FO_E_SHAPE, "wgrad: unsupported tile %dx%d", p.TA, p.TB);
  FO_CHECK_LAUNCH();
  const size_t slabElems = (size_t)p.tapsSlab * p.Apad * p.Bpad;
  const int CL = p.cpp >= 256 ? 16 : 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((slabElems + 63) / 64), banks), dim3(64 * CL), 0, s, ws, dw, p.cpp,
                     p.tapsSlab, p.Apad, p.Bpad, Areal, Breal, p.smallc ? 1 : 0, d->KW);
  FO_CHECK_LAUNCH();
  if (dbias) {
    hipLaunchKernelGGL(bias_reduce_kernel, dim3((Areal + 63) / 64), dim3(1024), 0, s, a.wsBias, dbias, p.nchunks, p.Apad, Areal);
    FO_CHECK_LAUNCH();
  }
  return FO_OK;
}

extern "C" int fo_conv_wgrad(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal,
                             float* dbias, float* ws, int64_t ws_bytes, void* stream) {
  return conv_wgrad_impl(d, P, Q, dw, Areal, Breal, dbias, ws, ws_bytes, stream, 1);
}

// `banks` independent filter gradients in one launch: the frame axis of P and Q is `banks` planes of N/banks frames
// (whole clips each); dw receives banks consecutive [Areal][Breal][taps] tensors.  Conv3d geometry only (KD > 1): this is
// the weight-gradient GEMM of the Winograd-transformed Conv3d, dU[xi] = sum_m dM[xi][m] (x) V[xi][m + kd - 1] (winograd.hip).
extern "C" int fo_conv_wgrad_banked(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal,
                                    float* ws, int64_t ws_bytes, int banks, void* stream) {
  FO_REQUIRE(banks >= 1, FO_E_SHAPE, "wgrad_banked: banks >= 1");
  return conv_wgrad_impl(d, P, Q, dw, Areal, Breal, nullptr, ws, ws_bytes, stream, banks);
}
