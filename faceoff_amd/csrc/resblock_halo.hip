// ResBlock.forward (reference models/vqvae_conv3d_latent.py:86-101: ReLU -> Conv 3x3 128 -> 32 -> ReLU -> Conv 1x1 32 -> 128 -> `out += input`)
// in fp32 on the fp32 MFMA, as a HALO-TILE kernel.  It replaces the tiled form (conv_igemm_kernel<32, ..., FUSE>) for frames whose width is a
// multiple of 32 (every C2 shape).  What the counters said about the tiled form (profiles/r03_pmc.md: MFMA-busy 0.58, 30 % of the wave cycles
// waiting, 1.2 GB read per 64^2 launch): a 128 x 32 tile gives a wave only 16 MFMAs per K-step behind a barrier, every K-step stages 20 KB
// through LDS -- the input rows nine times over (once per tap) and the 36.9 KB-per-channel-chunk filter again for every tile.  Here:
//   * a workgroup (4 waves, two workgroups per CU) walks output tiles of 2 rows x 32 pixels; the contraction index is SPLIT OVER THE WAVES:
//     wave w contracts input channels 32 w .. 32 w + 31 for all 64 pixels, so the filter slice it needs (9 taps x 32 channels x 32 outputs
//     = 36.9 KB) fits its registers -- 144 VGPRs, loaded once per kernel -- and the filter never touches LDS;
//   * so the tile's INPUT PATCH (4 rows x 34 pixels, zero halo) is four WAVE-PRIVATE 32-channel slices: a wave DMAs its own slice
//     (`buffer_load_dwordx4 ... lds`, no staging registers; 128-byte pixels, 16-byte granules XOR-swizzled on the source address so that the
//     fragment `ds_read_b128` is conflict-free) once for all nine taps and waits for it with its own `vmcnt` -- no workgroup barrier between
//     staging and the MFMAs; the block's leading ReLU is applied to the fragments (8 v_max per 8 MFMAs); fragments are read one step ahead;
//   * the four partial sums meet in LDS (each wave writes over the head of its own slice), + bias, ReLU = the hidden tile, which goes to
//     memory once (the backward needs it) and, from LDS, into the 1x1 contraction: wave w produces output channels 32 w .. + 31 (its 32 x 32
//     filter block in 16 registers); the next tile's slice is in flight from here on; residual, optional ReLU and the store are the epilogue,
//     one dword per lane = whole 128-byte lines.
// A wave issues 288 + 32 MFMAs per tile and meets two barriers.  Results differ from the tiled form by summation order only (the K-split adds
// four partial sums instead of running one chain): tests hold both to the same bound against torch-CPU.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <stdio.h>
#include "common.h"

#ifndef FO_RB_ST   // cache policy of the output / hidden stores.  -DFO_RB_ST=2 (nt): the outputs stop evicting the input frame from the XCD's L2 --
#define FO_RB_ST 0  // HBM reads 499 -> 349 MB per 64^2 launch (1.49 -> 1.04 x the input) -- but the launch is 9 % SLOWER (0.549 -> 0.600 ms): not used
#endif
#ifndef FO_ABLATE_RB   // diagnostic builds: bit 0 no patch DMA, 1 no 3x3 MFMAs, 2 no epilogue residual loads / stores
#define FO_ABLATE_RB 0
#endif

#ifndef FO_RB_STAMP
#define FO_RB_STAMP 0
#endif
#if FO_RB_STAMP
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

struct RBArgs {
  const float* x;       // [N][H][W][ldX], 128 channels
  const float* wp1;     // [32][9][128]   (fo_pack_conv of the 3x3 filter)
  const float* b1;      // [32]
  const float* wp3;     // [128][32]      (fo_pack_conv of the 1x1 filter)
  const float* b3;      // [128]
  float* h;             // [N][H][W][ldH], 32 channels
  float* out;           // [N][H][W][ldO], 128 channels
  int N, H, W, ldX, ldH, ldO, relu2;
  int tilesX, tilesY, ntiles, perXcd;
  unsigned xBytes, oBytes, hBytes;
  unsigned long long* stamps;   // diagnostic builds (FO_RB_STAMP): [workgroup][wave][phase] cycle sums
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;   // a voffset past any descriptor: the DMA writes zeros
constexpr int PW = 34;                  // patch pixels per row (32 + halo)
constexpr int PR = 4;                   // patch rows (2 + halo)
constexpr int SLICEB = PR * PW * 128;   // bytes of a wave's 32-channel slice (128-byte pixels)
constexpr int HLD = 36;                 // hidden tile row pitch (floats)
constexpr int PLD = 36;                 // partial sums: floats per pixel (32 + 4: the 16-byte writes of 8 consecutive pixels spread over the banks)
constexpr int LDS_BYTES = 4 * SLICEB + 64 * HLD * 4 + 128;  // 78 976: slices, hidden tile, b1

// granule swizzle of a patch pixel: column c keeps its 16-byte granule g at position g ^ ((c >> 1) & 7).  A ds_read_b128 lane group is 16
// lanes = 16 consecutive columns (mod 16: MI355X_MICROARCH.md, LDS table) x one granule; their slots (c & 1) * 8 + (g ^ ((c >> 1) & 7)) are the
// 16 distinct slots of the 256-byte bank row for any row base (the row pitch is even) and any tap shift.
__device__ __forceinline__ int gswz(int c) { return (c >> 1) & 7; }

__global__ __launch_bounds__(256, 2) void resblock_halo_fwd_kernel(const RBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (uniform: DMA destinations and scalar offsets are derived from it)
  const int l31 = lane & 31, half = lane >> 5;
  lds_byte* const lds3 = (lds_byte*)ldsb;
  // (the descriptor starts ONE PIXEL before the tensor, so that the patch's left halo column is a non-negative offset)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) - a.ldX, 0, a.xBytes + a.ldX * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.oBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, a.hBytes, 0x00020000);

  // ---- work distribution: XCD x (blockIdx % 8) owns a contiguous range of tiles and its workgroups walk it side by side, so that the halo
  // rows two vertically adjacent tiles share are fetched into ONE L2
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int tEnd = min(a.ntiles, (xcd + 1) * a.perXcd);
  int tile = xcd * a.perXcd + slot;

  // ---- DMA roles: a wave fills its own slice: 4 patch rows x 5 pieces of 8 pixels (the fifth: 2 pixels); lane = (pixel l / 8, granule
  // position l % 8).  A lane's share of the source address is a constant (pixel, swizzled granule: columns 8 g + lp swizzle by lp >> 1, and
  // by 4 more for odd g), the rest is scalar.
  const int lp = lane >> 3, pos = lane & 7;
  const unsigned dlane0 = (unsigned)(lp * a.ldX * 4 + ((pos ^ (lp >> 1)) * 16));
  const unsigned dlane1 = (unsigned)(lp * a.ldX * 4 + ((pos ^ (lp >> 1) ^ 4) * 16));
  auto dma_tile = [&](int t) {
    if (FO_ABLATE_RB & 1) return;
    if (t >= tEnd) return;
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    const int y0 = ty * 2 - 1;
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const int iy = y0 + r;
      const bool rowok = (unsigned)iy < (unsigned)a.H;
      const unsigned rowoff = rowok ? (unsigned)((((size_t)n * a.H + iy) * a.W + tx * 32) * a.ldX * 4) + wave * 128 : 0u;
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        const int ix = tx * 32 - 1 + g * 8 + lp;           // image column of this lane's pixel
        const bool ok = rowok & ((unsigned)ix < (unsigned)a.W);
        const unsigned vo = ok ? ((g & 1) ? dlane1 : dlane0) : OOB;
        lds_byte* const dst = lds3 + wave * SLICEB + (r * PW + g * 8) * 128;
        if (g < 4 || lane < 16)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, vo, rowoff + g * 8 * a.ldX * 4, 0, 0);
      }
    }
  };
  dma_tile(tile);

  // ---- this wave's filter slices, resident for the whole kernel.
  // The FILTER is the MFMAs' row operand and the pixels are the columns, so that a lane's accumulator quad is four consecutive channels of
  // ONE pixel: partial sums, residual and output move 16 bytes per lane (as the column operand's result -- one dword per lane and
  // instruction -- the 32 + 32 + 32 memory instructions per lane and tile cost a wave ~150 cycles each: 30 % of the tile).
  // 3x3: output channel l31, k = tap * 128 + 32 wave + 8 kk + 4 half .. + 3 (the lane's float4 feeds four consecutive MFMAs)
  f32x4 wf[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      wf[t][kk] = *reinterpret_cast<const f32x4*>(a.wp1 + (size_t)l31 * 1152 + t * 128 + wave * 32 + kk * 8 + half * 4);
  // 1x1: output channel 32 wave + l31, k = 8 kk + 4 half .. + 3
  f32x4 w3f[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w3f[kk] = *reinterpret_cast<const f32x4*>(a.wp3 + (size_t)(wave * 32 + l31) * 32 + kk * 8 + half * 4);
  const float b3a = half == 0 ? a.b3[wave * 32 + l31] : 0.f;             // b3 enters the 1x1 as one more contraction step against a row of ones

  // ---- fragment addressing: pixel column l31 + kw of a patch row, granule 2 kk + half at its swizzled position
  int cq[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) cq[kw][kk] = wave * SLICEB + (l31 + kw) * 128 + (((2 * kk + half) ^ gswz(l31 + kw)) * 16);

  float* const Hs = reinterpret_cast<float*>(ldsb + 4 * SLICEB);         // hidden tile [64][HLD]
  float* const B1s = Hs + 64 * HLD;
  if (tid < 32) B1s[tid] = a.b1[tid];                                     // (visible to all behind the first tile's first barrier)
  // lane shares of the residual / output / hidden addresses (the rest is scalar); rx starts one pixel early
  const unsigned xlane = (unsigned)(((l31 + 1) * a.ldX + wave * 32 + 4 * half) * 4);
  const unsigned olane = (unsigned)((l31 * a.ldO + wave * 32 + 4 * half) * 4);
  const unsigned hlane = (unsigned)(((((tid >> 2) >> 5) * a.W + ((tid >> 2) & 31)) * a.ldH + (tid & 3) * 8) * 4);
  // What the FO_RB_STAMP build measured (s_memtime per phase, 2.0 GHz under this load): a wave spends 45 000 cycles per tile, 27 000 of them in
  // the 3x3 loop (18 400 of MFMA issue, the rest sharing the pipe with the CU's other workgroup), so that a SIMD's MFMA pipe is 0.9 busy in
  // the steady state -- the launch's 0.70 of the 2.4-GHz peak is that x the clock x ramp and tail.  Measured null, each on one box: starting
  // every second workgroup half a tile late, or at a higher s_setprio (whichever pairing of workgroups was assumed); issuing the next patch's
  // rows 0, 1 from inside the 3x3 loop (the time moves into the loop: the total is conserved); drawing tiles from a per-XCD counter instead of
  // the fixed stride (per-wave times 0.88..1.13 of the mean became 0.95..1.07, the launch 1 % shorter); walking DOWN the image with the patch as
  // a ring of four row slots, so that a tile brings in only its two new rows (half the DMA pieces and L2 traffic: the same 0.47 ms -- and the
  // variant was not run-to-run reproducible at full size, a hazard that was not tracked down; dropped).
#if FO_RB_STAMP
  unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long t0m = tlast, t0r = __builtin_amdgcn_s_memrealtime();
#endif
  for (; tile < tEnd; tile += slots) {
    const int tx = tile % a.tilesX, r1 = tile / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    // this wave's slice has landed (nobody else reads it: no barrier).  Vector memory operations retire in order and the 8 youngest are the
    // previous tile's output stores (the first tile: the filter loads), which may keep flying
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    STAMP(0);

    // ---- 3x3 over this wave's channel slice: acc[mb] = 32 outputs x pixels of tile row mb (partial over 32 of the 128 input channels);
    // 36 steps (tap, kk) of two fragment reads + 8 MFMAs, the reads one step ahead
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    constexpr int NSTEP = (FO_ABLATE_RB & 2) ? 0 : 36;
    f32x4 fa[2][2];
    auto frag = [&](int st, int mb) {
      const int tap = st >> 2, kk = st & 3, kh = tap / 3, kw = tap - kh * 3;
      return *reinterpret_cast<const f32x4*>(ldsb + cq[kw][kk] + (mb + kh) * PW * 128);
    };
    if (NSTEP) { fa[0][0] = frag(0, 0); fa[0][1] = frag(0, 1); }
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int cur = st & 1;
      if (st + 1 < NSTEP) { fa[cur ^ 1][0] = frag(st + 1, 0); fa[cur ^ 1][1] = frag(st + 1, 1); }
      __builtin_amdgcn_sched_barrier(0);                   // (the next step's reads are issued HERE, before this step's MFMAs)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)                       // the block's leading nn.ReLU (:91); one v_max per element (fmaxf adds a canonicalize)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("v_max_f32 %0, 0, %0" : "+v"(fa[cur][mb][s]));
      __builtin_amdgcn_sched_barrier(0);                   // (all eight before the first MFMA: every element is written >= 4 instructions before it is read)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[st >> 2][st & 3][s], fa[cur][mb][s], acc[mb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                   // (reads stay ONE step ahead: the registers belong to the filter)
    }

    STAMP(1);
    // ---- the four partial sums -> LDS, each wave over the head of its own slice (which only it reads): partial[pixel 0..63][PLD]; the lane
    // holds pixel 32 mb + l31, channels 8 j + 4 half .. + 3 in accumulator quad j
    float* const Pw = reinterpret_cast<float*>(ldsb + wave * SLICEB) + l31 * PLD + 4 * half;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(Pw + mb * 32 * PLD + 8 * j) = f32x4{acc[mb][4 * j], acc[mb][4 * j + 1], acc[mb][4 * j + 2], acc[mb][4 * j + 3]};
    // the residual (`out += input`, :99: the RAW input, not its ReLU) is what the 1x1's accumulators START from: the lane's 8 loads go
    // straight into them, in flight across the two barriers below, and are older than the next patch's DMAs (loads retire in order: the 1x1
    // never waits for the patch).  Accumulator quad j of block mb = pixel pix0 + mb W + l31, channels 32 wave + 8 j + 4 half .. + 3.
    const int pix0 = (n * a.H + ty * 2) * a.W + tx * 32;
    f32x16 acc2[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = (FO_ABLATE_RB & 4) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                           : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xlane + 32 * j, (pix0 + mb * a.W) * a.ldX * 4, 0));
        acc2[mb][4 * j] = v.x; acc2[mb][4 * j + 1] = v.y; acc2[mb][4 * j + 2] = v.z; acc2[mb][4 * j + 3] = v.w;
      }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    // hidden tile = relu(sum of the partials + b1): thread -> (pixel tid / 4, channels 8 (tid % 4) .. + 7); to memory (16 B per lane) and to
    // LDS in the 1x1's A-operand layout [pixel][36]
    {
      const int hp = tid >> 2, hc = (tid & 3) * 8;
      f32x4 s0 = *reinterpret_cast<const f32x4*>(B1s + hc), s1 = *reinterpret_cast<const f32x4*>(B1s + hc + 4);   // (from LDS: a global load here would wait behind the residual loads)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* const P = reinterpret_cast<const float*>(ldsb + w * SLICEB) + hp * PLD + hc;
        s0 += *reinterpret_cast<const f32x4*>(P);
        s1 += *reinterpret_cast<const f32x4*>(P + 4);
      }
      s0.x = fmaxf(s0.x, 0.f); s0.y = fmaxf(s0.y, 0.f); s0.z = fmaxf(s0.z, 0.f); s0.w = fmaxf(s0.w, 0.f);
      s1.x = fmaxf(s1.x, 0.f); s1.y = fmaxf(s1.y, 0.f); s1.z = fmaxf(s1.z, 0.f); s1.w = fmaxf(s1.w, 0.f);
      *reinterpret_cast<f32x4*>(Hs + hp * HLD + hc) = s0;
      *reinterpret_cast<f32x4*>(Hs + hp * HLD + hc + 4) = s1;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s0), rh, hlane, pix0 * a.ldH * 4, FO_RB_ST);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s1), rh, hlane + 16, pix0 * a.ldH * 4, FO_RB_ST);
    }
    STAMP(4);
    __syncthreads();                                       // the partials are consumed: the slices are free
    STAMP(5);
    dma_tile(tile + slots);                                // the next patch flies during the 1x1 and the epilogue (issued earlier -- the pieces
                                                           // beside the partial sums right behind the MFMAs -- it measured no faster)

    STAMP(6);
    // ---- 1x1: this wave's 32 output channels for the 64 pixels, K = 32 (+ 1: the bias against ones), on top of the residual
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(b3a, 1.0f, acc2[mb], 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 fb[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) fb[mb] = *reinterpret_cast<const f32x4*>(Hs + (mb * 32 + l31) * HLD + half * 4 + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w3f[kk][s], fb[mb][s], acc2[mb], 0, 0, 0);
    }
    STAMP(7);
    // ---- epilogue: [the encoder's / decoder's trailing ReLU,] 16 bytes per lane
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = {acc2[mb][4 * j], acc2[mb][4 * j + 1], acc2[mb][4 * j + 2], acc2[mb][4 * j + 3]};
        if (a.relu2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (!(FO_ABLATE_RB & 4) || v.x == 12345.f)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, olane + 32 * j, (pix0 + mb * a.W) * a.ldO * 4, FO_RB_ST);
      }
    STAMP(8);
    // (no barrier here: the next tile's first shared write is its partial sums -- over a wave's own slice -- and the hidden tile is only
    // rewritten behind the next tile's first barrier, which every wave reaches after its 1x1 reads)
  }
#if FO_RB_STAMP
  if (lane == 0 && a.stamps)
    for (int i = 0; i < 9; ++i) a.stamps[(blockIdx.x * 4 + wave) * 9 + i] = ph[i];
  if (blockIdx.x == 0 && tid == 0 && a.stamps) {
    a.stamps[gridDim.x * 36] = __builtin_amdgcn_s_memtime() - t0m;
    a.stamps[gridDim.x * 36 + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
#endif
}

}  // namespace

// 1 = launched, 0 = geometry not applicable (the caller falls back to the tiled kernel)
int fo_resblock_halo_try(const fo_conv_desc* d, const float* x, const float* wp1, const float* b1, const float* wp3, const float* b3, float* hbuf,
                         float* out, int ldOut2, int out_relu, hipStream_t stream) {
  static const bool off = getenv("FACEOFF_NO_RESBLOCK_HALO") != nullptr;
  static const bool force = getenv("FACEOFF_FORCE_RESBLOCK_HALO") != nullptr;          // tests: at any size
  if (off || d->Win % 32 != 0 || d->Hin % 2 != 0 || d->Cin != 128 || d->Cout != 32) return 0;
  RBArgs a;
  a.x = x; a.wp1 = wp1; a.b1 = b1; a.wp3 = wp3; a.b3 = b3; a.h = hbuf; a.out = out;
  a.N = d->N; a.H = d->Hin; a.W = d->Win; a.ldX = d->ldIn; a.ldH = d->ldOut; a.ldO = ldOut2; a.relu2 = out_relu;
  a.tilesX = d->Win / 32; a.tilesY = d->Hin / 2; a.ntiles = d->N * a.tilesX * a.tilesY;
  const size_t xBytes = ((size_t)d->N * d->Hin * d->Win - 1) * d->ldIn * 4 + 512;
  if (xBytes >= 0x7fffffffull) return 0;                   // (the patch DMA addresses the input through one 32-bit descriptor)
  a.xBytes = (unsigned)xBytes;
  const size_t npix = (size_t)d->N * d->Hin * d->Win;
  const size_t oBytes = (npix - 1) * ldOut2 * 4 + 512, hBytes = (npix - 1) * d->ldOut * 4 + 128;
  if (oBytes >= 0x7fffffffull || hBytes >= 0x7fffffffull) return 0;
  a.oBytes = (unsigned)oBytes; a.hBytes = (unsigned)hBytes;
  a.perXcd = (a.ntiles + 7) / 8;
  const int cus = fo_cu_count();
  if (a.ntiles < 4 * cus && !force) return 0;              // small launches: the tiled kernel's many small workgroups fill the chip better
  constexpr int ldsBytes = LDS_BYTES;
  static fo_lds_once once;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(resblock_halo_fwd_kernel), ldsBytes, "resblock_halo")) return 0;           // -> the tiled kernel
  a.stamps = nullptr;
#if FO_RB_STAMP
  const int grid_ = std::max(8, std::min((a.ntiles + 7) / 8 * 8, 2 * cus) / 8 * 8);
  static unsigned long long* dstamps = nullptr;
  if (!dstamps) (void)hipMalloc(&dstamps, (size_t)grid_ * 4 * 9 * 8 + 16);
  a.stamps = dstamps;
#endif
  FO_NOTE("resblock_halo_fwd_kernel");
  hipLaunchKernelGGL(resblock_halo_fwd_kernel, dim3(std::max(8, std::min((a.ntiles + 7) / 8 * 8, 2 * cus) / 8 * 8)), dim3(256), ldsBytes, stream, a);
#if FO_RB_STAMP
  {
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned long long> hs((size_t)grid_ * 4 * 9 + 2);
    (void)hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls == 5) {
      const char* names[9] = {"patch wait", "3x3 MFMAs", "partials + residual issue", "barrier 1", "sum + hidden store", "barrier 2", "DMA issue",
                              "1x1 (residual wait)", "stores"};
      double tot[9] = {0}; double all = 0;
      fprintf(stderr, "[rb stamp] s_memtime / s_memrealtime(100 MHz) over workgroup 0: %llu / %llu -> s_memtime runs at %.0f MHz\n", hs[hs.size() - 2], hs[hs.size() - 1],
              100.0 * (double)hs[hs.size() - 2] / (double)hs[hs.size() - 1]);
      for (size_t i = 0; i + 2 < hs.size(); ++i) { tot[i % 9] += (double)hs[i]; all += (double)hs[i]; }
      const double tiles_per_wave = (double)a.ntiles / grid_;
      for (int i = 0; i < 9; ++i) fprintf(stderr, "[rb stamp] %-40s %8.0f cycles / tile  (%.1f %%)\n", names[i], tot[i] / (grid_ * 4) / tiles_per_wave, 100 * tot[i] / all);
      fprintf(stderr, "[rb stamp] total %8.0f cycles / tile / wave\n", all / (grid_ * 4) / tiles_per_wave);
      double mn = 1e30, mx = 0;
      for (int w = 0; w < grid_ * 4; ++w) { double t = 0; for (int i = 0; i < 9; ++i) t += (double)hs[(size_t)w * 9 + i]; mn = std::min(mn, t); mx = std::max(mx, t); }
      fprintf(stderr, "[rb stamp] per-wave loop time: min %.0f mean %.0f max %.0f cycles\n", mn, all / (grid_ * 4), mx);
    }
  }
#endif
  return 1;
}
