// Backward through a ResBlock's SECOND convolution (reference models/vqvae_conv3d_latent.py:86-101: ... ReLU -> Conv 1x1 32 -> 128, `out += input`)
// in ONE pass over the block's output gradient g [M][128] and its hidden activation h [M][32] (post-ReLU, kept by the forward):
//     g_h [M][32]   = (g W3) * (h > 0)            the data gradient, through the ReLU in front of the 1x1
//     dW3 [128][32] = g^T h                       the filter gradient
//     db3 [128]     = column sums of g            the bias gradient
// As three launches (1x1 data gradient, 1x1 filter gradient, column sums) each of them streamed g again -- 335 MB per 64^2 block -- and all three
// sat at the HBM roof: 0.099 + 0.134 + 0.07 ms.  Here g is read once (419 MB in, 84 MB out per 64^2 block).
//   * persistent four-wave workgroups (two per CU) walk tiles of 64 pixels, double-buffered: a tile of g is four 32-channel slices (128-byte
//     pixels, the halo kernel's granule swizzle), each DMA'd by one wave, with the tile of h beside it; the next tile's DMAs go out at the top
//     of a tile, so ONE barrier per tile does both jobs (this tile's data is visible; everyone is done with the buffer about to be refilled);
//   * data gradient: wave w owns pixels 16 w .. + 15 and contracts over all 128 channels (v_mfma_f32_16x16x4_f32, W3 as the row operand -- 64
//     registers -- so that a lane's accumulator is four hidden channels of one pixel): no partial sums, * (h > 0), 16-byte stores;
//   * filter gradient: wave w owns output channels 32 w .. + 31: its slice transposed (channels = rows, pixels = the contraction index, one dword
//     per lane) against the h tile, accumulated in 16 registers over ALL the workgroup's tiles; the bias gradient is the running sum of those same
//     row fragments;
//   * one slab of partial dW3 / db3 per workgroup, summed in a fixed order by a second small kernel (the walk is a fixed stride: bit-reproducible).
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace {

struct RB3Args {
  const float* g;       // [M][ldG], 128 channels
  const float* h;       // [M][ldH], 32 channels
  const float* wp3;     // [128][32]  (fo_pack_conv of the 1x1 filter = its [co][ci] matrix)
  float* gh;            // [M][ldGh], 32 channels
  float* ws;            // [grid][128 * 32 + 128] partial dW3, db3
  int M, ldG, ldH, ldGh, ntiles;
  unsigned gBytes, hBytes, ghBytes;
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
constexpr int TP = 64;                    // pixels per tile
constexpr int SLICEB = TP * 128;          // bytes of a 32-channel slice of g
constexpr int STAGEB = 4 * SLICEB + TP * 128;             // a stage: four slices + the h tile = 40 960
constexpr int LDS_BYTES = 2 * STAGEB;

__device__ __forceinline__ int gswz(int p) { return (p >> 1) & 7; }     // granule g of pixel p sits at position g ^ gswz(p) (resblock_halo.hip)

__global__ __launch_bounds__(256, 2) void resblock_bwd_conv3_kernel(const RB3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, q = lane >> 4;
  lds_byte* const lds3 = (lds_byte*)ldsb;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, a.gBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h), 0, a.hBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rgh = __builtin_amdgcn_make_buffer_rsrc(a.gh, 0, a.ghBytes, 0x00020000);

  // ---- DMA roles.  g: wave w fills slice w, 8 pieces of 8 pixels; lane = (pixel l / 8, granule position l % 8), source granule
  // position ^ gswz(pixel): pixels 8 k + lp swizzle by (4 k + (lp >> 1)) & 7 = (lp >> 1) ^ (4 (k & 1)).  h: wave w fills pixels 16 w .. + 15.
  const int lp = lane >> 3, pos = lane & 7;
  const unsigned glane0 = (unsigned)(lp * a.ldG * 4 + ((pos ^ (lp >> 1)) * 16));
  const unsigned glane1 = (unsigned)(lp * a.ldG * 4 + ((pos ^ (lp >> 1) ^ 4) * 16));
  const unsigned hlaneD = (unsigned)(lp * a.ldH * 4 + pos * 16);
  auto dma_tile = [&](int t, int stage) {
    if (t >= a.ntiles) return;
    const int m0 = t * TP;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = m0 + k * 8 + lp < a.M;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(lds3 + stage * STAGEB + wave * SLICEB + k * 1024), 16,
                                               ok ? ((k & 1) ? glane1 : glane0) : OOB, (unsigned)((size_t)(m0 + k * 8) * a.ldG * 4) + wave * 128, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p0 = wave * 16 + k * 8;
      const bool ok = m0 + p0 + lp < a.M;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (__attribute__((address_space(3))) void*)(lds3 + stage * STAGEB + 4 * SLICEB + p0 * 128), 16,
                                               ok ? hlaneD : OOB, (unsigned)((size_t)(m0 + p0) * a.ldH * 4), 0, 0);
    }
  };
  int tile = blockIdx.x;
  dma_tile(tile, 0);

  // ---- W3 for the data gradient: row operand of v_mfma_f32_16x16x4_f32, row = hidden channel 16 nb + l15; a lane reads four consecutive
  // output channels of its pixel at once (ds_read_b128) and feeds them to four MFMAs, so MFMA (j, s) contracts over the channels
  // 16 j + 4 q + s, q = 0..3: the lane's filter value for it is W3[16 j + 4 q + s][16 nb + l15]
  float wd[8][4][2];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) wd[j][s][nb] = a.wp3[(size_t)(16 * j + 4 * q + s) * 32 + 16 * nb + l15];

  // fragment addresses inside a stage.  Data gradient: pixel 16 wave + l15, channels 16 j + 4 q .. + 3 = slice j / 2, granule 4 (j & 1) + q
  const int px = wave * 16 + l15;
  int dq[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) dq[jj] = px * 128 + (((4 * jj + q) ^ gswz(px)) * 16);
  // filter gradient: row = channel l31 of slice `wave` at pixel 2 s + half: granule (l31 >> 2) ^ (s & 7), dword l31 & 3
  const int aq = wave * SLICEB + half * 128 + (l31 & 3) * 4, ag = l31 >> 2;
  const int bq = 4 * SLICEB + half * 128 + l31 * 4;
  const int hq = 4 * SLICEB + px * 128 + q * 16;           // the mask: h[pixel][16 nb + 4 q .. + 3]
  const unsigned ghlane = (unsigned)((px * a.ldGh + 4 * q) * 4);

  f32x16 accw;                                             // dW3[32 wave + row][l31], over all tiles
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  float dbp = 0.f;                                         // db3[32 wave + l31], this lane's pixel parity

  for (int it = 0; tile < a.ntiles; tile += gridDim.x, ++it) {
    const unsigned char* const st = ldsb + (it & 1) * STAGEB;
    // this wave's pieces of the tile have landed (vector-memory operations retire in order; the 2 youngest are the previous tile's stores);
    // the barrier makes everyone's pieces visible AND tells that everyone is done with the other stage, which the next tile's DMAs refill
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();
    dma_tile(tile + gridDim.x, (it & 1) ^ 1);

    // ---- data gradient of this wave's 16 pixels: acc[nb] = hidden channels 16 nb + 4 q .. + 3 of pixel 16 wave + l15
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(st + (j >> 1) * SLICEB + dq[j & 1]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[j][s][nb], gv[s], acc[nb], 0, 0, 0);
    }
    // ---- filter gradient: rows = this wave's 32 output channels, contraction over the tile's 64 pixels (two per MFMA), columns = hidden
#pragma unroll
    for (int s = 0; s < TP / 2; ++s) {
      const float av = *reinterpret_cast<const float*>(st + aq + s * 256 + ((ag ^ (s & 7)) * 16));
      const float bv = *reinterpret_cast<const float*>(st + bq + s * 256);
      dbp += av;
      accw = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw, 0, 0, 0);
    }
    // ---- g_h = data gradient * (h > 0), 16 bytes per lane and hidden-channel block
    {
      const bool ok = tile * TP + px < a.M;
      const unsigned so = (unsigned)((size_t)tile * TP * a.ldGh * 4);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(st + hq + nb * 64);
        f32x4 v = acc[nb];
        v.x = hv.x > 0.f ? v.x : 0.f; v.y = hv.y > 0.f ? v.y : 0.f; v.z = hv.z > 0.f ? v.z : 0.f; v.w = hv.w > 0.f ? v.w : 0.f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rgh, ok ? ghlane + nb * 64 : OOB, so, 0);
      }
    }
  }

  // ---- this workgroup's slab: dW3[co][ci] partial, then db3[co] partial
  float* const slab = a.ws + (size_t)blockIdx.x * (128 * 32 + 128);
#pragma unroll
  for (int r = 0; r < 16; ++r) slab[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = accw[r];
  dbp += lane_xor<32>(dbp);
  if (half == 0) slab[128 * 32 + wave * 32 + l31] = dbp;
}

// dW3 / db3 = the slabs summed in a fixed order: a workgroup = 64 consecutive elements x 8 slab groups (group k: slabs k, k + 8, ... in four
// running sums), the groups combined through LDS in group order
__global__ __launch_bounds__(512) void resblock_bwd_conv3_reduce_kernel(const float* __restrict__ ws, int nslabs, float* __restrict__ dw3,
                                                                        float* __restrict__ db3) {
  __shared__ float red[512];
  constexpr int NE = 128 * 32 + 128;
  const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + li;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < NE) {
    const float* p = ws + e;
    int b = grp;
    for (; b + 24 < nslabs; b += 32) {
      s0 += p[(size_t)b * NE]; s1 += p[(size_t)(b + 8) * NE]; s2 += p[(size_t)(b + 16) * NE]; s3 += p[(size_t)(b + 24) * NE];
    }
    for (; b < nslabs; b += 8) s0 += p[(size_t)b * NE];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < NE) {
    float s = red[li];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k * 64 + li];
    if (e < 128 * 32) dw3[e] = s;
    else if (db3) db3[e - 128 * 32] = s;
  }
}

int grid_for(int ntiles) { return std::max(1, std::min(ntiles, 2 * fo_cu_count())); }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------------
// 3x3 pad-1 stride-1 convolution 32 -> 128 with the data-gradient epilogue -- the backward of a ResBlock's FIRST convolution
// (reference models/vqvae_conv3d_latent.py:92-93: g_x = conv3x3^T(g_h) * (x > 0) + g_out) -- as a halo-tile kernel in the manner of
// resblock_halo.hip, simpler than the forward because the 32-channel side is the INPUT:
//   * the tile's patch (4 rows x 34 pixels x 32 channels = 17 KB, zero halo) is one slice shared by the four waves, double-buffered, the next
//     tile's DMAs issued at the top of a tile: one barrier per tile;
//   * the OUTPUT channels are split over the waves (wave w: 32 w .. + 31): each keeps its 9 x 32 x 32 filter block in 144 registers, contracts
//     all of K = 288 itself -- no partial sums -- with the filter as the MFMA's row operand, so that a lane's accumulator quad is four
//     channels of one pixel: mask, residual gradient and result move 16 bytes per lane.
namespace {

struct D3Args {
  const float* gh;      // [N][H][W][ldGh], 32 channels
  const float* wpd;     // [128][9][32]  (the data-gradient pack of the 3x3 filter: fo_pack_conv_dgrad)
  const float* mask;    // [N][H][W][ldM], 128 channels, or null
  const float* add;     // [N][H][W][ldA], 128 channels, or null
  float* out;           // [N][H][W][ldO], 128 channels
  int N, H, W, ldGh, ldM, ldA, ldO;
  int tilesX, tilesY, ntiles, perXcd;
  unsigned ghBytes, mBytes, aBytes, oBytes;
};

constexpr int PW3 = 34, PR3 = 4, PATCHB = PR3 * PW3 * 128;

template <bool MASK, bool ADD>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_halo_kernel(const D3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];       // two patches
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  lds_byte* const lds3 = (lds_byte*)ldsb;
  // (the descriptor starts ONE PIXEL before the tensor, so that the patch's left halo column is a non-negative offset)
  const __amdgpu_buffer_rsrc_t rgh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gh) - a.ldGh, 0, a.ghBytes + a.ldGh * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask), 0, a.mBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.add), 0, a.aBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.oBytes, 0x00020000);

  // XCD x (blockIdx % 8) owns a contiguous range of tiles, walked side by side by its workgroups (shared halo rows meet in one L2)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int tEnd = min(a.ntiles, (xcd + 1) * a.perXcd);
  int tile = xcd * a.perXcd + slot;

  // DMA roles: 4 patch rows x 5 pieces of 8 pixels (the fifth: 2 pixels) = 20 pieces, five per wave; lane = (pixel l / 8, granule l % 8)
  const int lp = lane >> 3, pos = lane & 7;
  const unsigned dlane0 = (unsigned)(lp * a.ldGh * 4 + ((pos ^ (lp >> 1)) * 16));
  const unsigned dlane1 = (unsigned)(lp * a.ldGh * 4 + ((pos ^ (lp >> 1) ^ 4) * 16));
  auto dma_tile = [&](int t, int stage) {
    if (t >= tEnd) return;
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int id = wave * 5 + k, r = id / 5, g = id - r * 5;           // (= row `wave`, piece k)
      const int iy = ty * 2 - 1 + r;
      const bool rowok = (unsigned)iy < (unsigned)a.H;
      const unsigned rowoff = rowok ? (unsigned)((((size_t)n * a.H + iy) * a.W + tx * 32) * a.ldGh * 4) : 0u;
      const bool ok = rowok & !((g == 0) & (tx == 0) & (lp == 0)) & !((g == 4) & (tx == a.tilesX - 1) & (lp == 1));
      const unsigned vo = ok ? ((g & 1) ? dlane1 : dlane0) : OOB;
      lds_byte* const dst = lds3 + stage * PATCHB + (r * PW3 + g * 8) * 128;
      if (g < 4 || lane < 16)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rgh, (__attribute__((address_space(3))) void*)dst, 16, vo, rowoff + g * 8 * a.ldGh * 4, 0, 0);
    }
  };
  dma_tile(tile, 0);

  // filter block: row = output channel 32 wave + l31, k = tap * 32 + 8 kk + 4 half .. + 3
  f32x4 wf[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wf[t][kk] = *reinterpret_cast<const f32x4*>(a.wpd + (size_t)(wave * 32 + l31) * 288 + t * 32 + kk * 8 + half * 4);
  // fragment address of (tap column kw, channel group kk) = cq[kw] ^ (kk << 5): the granule index (2 kk + half) ^ swizzle differs from
  // kk = 0's in bits 1..2 only (12 address registers would not fit beside the filter's 144)
  int cq[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) cq[kw] = (l31 + kw) * 128 + ((half ^ gswz(l31 + kw)) * 16);
  const unsigned mlane = (unsigned)((l31 * a.ldM + wave * 32 + 4 * half) * 4);
  const unsigned alane = (unsigned)((l31 * a.ldA + wave * 32 + 4 * half) * 4);
  const unsigned olane = (unsigned)((l31 * a.ldO + wave * 32 + 4 * half) * 4);

  for (int it = 0; tile < tEnd; tile += slots, ++it) {
    const int tx = tile % a.tilesX, r1 = tile / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    const int stoff = (it & 1) * PATCHB;
    // this wave's pieces have landed (they are older than the previous tile's epilogue loads, which its stores waited for; the 4 youngest
    // vector-memory operations are that tile's last stores); the barrier makes the whole patch visible and tells that everyone is done with
    // the other stage, which the next tile's DMAs refill
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    dma_tile(tile + slots, (it & 1) ^ 1);

    // one tile row (32 pixels) at a time: its mask / residual loads go out first and land under its 144 MFMAs; accumulator quad j =
    // channels 32 wave + 8 j + 4 half .. + 3 of pixel (row mb, column l31)
    const int pix0 = (n * a.H + ty * 2) * a.W + tx * 32;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x4 mk[4], ad[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MASK) mk[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mlane + 32 * j, (pix0 + mb * a.W) * a.ldM * 4, 0));
        if (ADD) ad[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, alane + 32 * j, (pix0 + mb * a.W) * a.ldA * 4, 0));
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      f32x4 fa[2];
      auto frag = [&](int s9) {
        const int tap = s9 >> 2, kk = s9 & 3, kh = tap / 3, kw = tap - kh * 3;
        return *reinterpret_cast<const f32x4*>(ldsb + ((cq[kw] + stoff) ^ (kk << 5)) + (mb + kh) * PW3 * 128);
      };
      fa[0] = frag(0);
#pragma unroll
      for (int s9 = 0; s9 < 36; ++s9) {
        const int cur = s9 & 1;
        if (s9 + 1 < 36) fa[cur ^ 1] = frag(s9 + 1);
        __builtin_amdgcn_sched_barrier(0);                 // (the next step's read is issued here, before this step's MFMAs)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s9 >> 2][s9 & 3][s], fa[cur][s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = {acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
        if (MASK) {
          const f32x4 m = mk[j];
          v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (ADD) v += ad[j];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, olane + 32 * j, (pix0 + mb * a.W) * a.ldO * 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

}  // namespace

// 1 = launched, 0 = not this kernel's geometry (the caller goes on to the tiled implicit GEMM)
int fo_conv3x3_c32_halo_try(const fo_conv_desc* d, const float* in, const float* wp, const float* mask, const float* add, float* out, hipStream_t stream) {
  static const bool off = getenv("FACEOFF_NO_RESBLOCK_HALO") != nullptr;
  static const bool force = getenv("FACEOFF_FORCE_RESBLOCK_HALO") != nullptr;          // tests: at any size
  if (off || d->Cin != 32 || d->Cout != 128 || d->KD != 1 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->ostride != 1 || d->padH != 1 ||
      d->padW != 1 || d->Hm != d->Hin || d->Wm != d->Win || d->Hout != d->Hin || d->Wout != d->Win || d->Win % 32 != 0 || d->Hin % 2 != 0)
    return 0;
  if ((d->flags & ~(FO_MASK | FO_ADD)) != 0 || !!(d->flags & FO_MASK) != (mask != nullptr) || !!(d->flags & FO_ADD) != (add != nullptr)) return 0;
  if (d->ldIn % 4 != 0 || d->ldOut % 4 != 0 || (mask && d->ldMask % 4 != 0) || (add && d->ldAdd % 4 != 0)) return 0;
  D3Args a;
  a.gh = in; a.wpd = wp; a.mask = mask; a.add = add; a.out = out;
  a.N = d->N; a.H = d->Hin; a.W = d->Win; a.ldGh = d->ldIn; a.ldM = mask ? d->ldMask : 0; a.ldA = add ? d->ldAdd : 0; a.ldO = d->ldOut;
  a.tilesX = d->Win / 32; a.tilesY = d->Hin / 2; a.ntiles = d->N * a.tilesX * a.tilesY; a.perXcd = (a.ntiles + 7) / 8;
  const int cus = fo_cu_count();
  if (a.ntiles < 4 * cus && !force) return 0;
  const size_t npix = (size_t)d->N * d->Hin * d->Win;
  const size_t ghB = (npix - 1) * d->ldIn * 4 + 128, mB = mask ? (npix - 1) * d->ldMask * 4 + 512 : 16, aB = add ? (npix - 1) * d->ldAdd * 4 + 512 : 16,
               oB = (npix - 1) * d->ldOut * 4 + 512;
  if (ghB >= 0x7fffffffull || mB >= 0x7fffffffull || aB >= 0x7fffffffull || oB >= 0x7fffffffull) return 0;
  a.ghBytes = (unsigned)ghB; a.mBytes = (unsigned)mB; a.aBytes = (unsigned)aB; a.oBytes = (unsigned)oB;
  const dim3 grid(std::max(8, std::min((a.ntiles + 7) / 8 * 8, 2 * cus) / 8 * 8));
  if (mask) { if (add) FO_NOTE_T("conv3x3_c32_halo_kernel", true, true); else FO_NOTE_T("conv3x3_c32_halo_kernel", true, false); }
  else { if (add) FO_NOTE_T("conv3x3_c32_halo_kernel", false, true); else FO_NOTE_T("conv3x3_c32_halo_kernel", false, false); }
  if (mask && add) hipLaunchKernelGGL((conv3x3_c32_halo_kernel<true, true>), grid, dim3(256), 2 * PATCHB, stream, a);
  else if (mask) hipLaunchKernelGGL((conv3x3_c32_halo_kernel<true, false>), grid, dim3(256), 2 * PATCHB, stream, a);
  else if (add) hipLaunchKernelGGL((conv3x3_c32_halo_kernel<false, true>), grid, dim3(256), 2 * PATCHB, stream, a);
  else hipLaunchKernelGGL((conv3x3_c32_halo_kernel<false, false>), grid, dim3(256), 2 * PATCHB, stream, a);
  return 1;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Filter (and bias) gradient of a ResBlock's FIRST convolution (ReLU -> Conv2d(128, 32, 3, padding=1), reference :91-92):
//     dW1[co][ci][kh][kw] = sum_pixels relu(x)[pixel + (kh - 1, kw - 1)][ci] * g_h[pixel][co],     db1[co] = sum_pixels g_h[pixel][co]
// as a halo-tile kernel: the same wave-private 32-channel slices of the input patch as the forward (resblock_halo.hip), staged once for all nine
// taps; wave w owns input channels 32 w .. + 31 and keeps its 9 x 32 x 32 block of dW1 in 144 accumulator registers over ALL its tiles
// (rows = input channels, columns = output channels, the contraction runs over pixels, two per v_mfma_f32_32x32x2_f32); both operands are read
// one dword per lane from [pixel][channel] images, which for this access pattern (32 consecutive channels of one pixel per half-wave) is
// conflict-free without a swizzle; one slab per workgroup, summed in a fixed order by a second kernel which also transposes to the checkpoint
// layout.  (The tiled conv_wgrad_kernel<32, 128> re-stages the input rows once per tap and turns its accumulators over every chunk.)
namespace {

struct W1Args {
  const float* x;       // [N][H][W][ldX], 128 channels (the block's input; ReLU applied on the fly)
  const float* gh;      // [N][H][W][ldGh], 32 channels
  float* ws;            // [grid][9 * 128 * 32 + 32]
  int N, H, W, ldX, ldGh;
  int tilesX, tilesY, ntiles, perXcd;
  unsigned xBytes, ghBytes;
};

constexpr int W1_SLICEB = 4 * 34 * 128;                    // a wave's 32-channel slice of the 4 x 34-pixel patch
constexpr int W1_LDS = 4 * W1_SLICEB + 64 * 128;           // + the g_h tile: 77 824
constexpr int W1_SLAB = 9 * 128 * 32 + 32;

__global__ __launch_bounds__(256, 2) void resblock_wgrad1_halo_kernel(const W1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  lds_byte* const lds3 = (lds_byte*)ldsb;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) - a.ldX, 0, a.xBytes + a.ldX * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gh), 0, a.ghBytes, 0x00020000);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int tEnd = min(a.ntiles, (xcd + 1) * a.perXcd);
  int tile = xcd * a.perXcd + slot;

  const int lp = lane >> 3, pos = lane & 7;
  const unsigned xlaneD = (unsigned)(lp * a.ldX * 4 + pos * 16), glaneD = (unsigned)(lp * a.ldGh * 4 + pos * 16);
  auto dma_tile = [&](int t) {
    if (t >= tEnd) return;
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iy = ty * 2 - 1 + r;
      const bool rowok = (unsigned)iy < (unsigned)a.H;
      const unsigned rowoff = rowok ? (unsigned)((((size_t)n * a.H + iy) * a.W + tx * 32) * a.ldX * 4) + wave * 128 : 0u;
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        const bool ok = rowok & !((g == 0) & (tx == 0) & (lp == 0)) & !((g == 4) & (tx == a.tilesX - 1) & (lp == 1));
        lds_byte* const dst = lds3 + wave * W1_SLICEB + (r * 34 + g * 8) * 128;
        if (g < 4 || lane < 16)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? xlaneD : OOB, rowoff + g * 8 * a.ldX * 4, 0, 0);
      }
    }
    // g_h: wave w fills tile pixels 16 w .. + 15 (tile row w / 2, columns 16 (w % 2) .. + 15)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p0 = wave * 16 + k * 8;
      const unsigned so = (unsigned)((((size_t)n * a.H + ty * 2 + (p0 >> 5)) * a.W + tx * 32 + (p0 & 31)) * a.ldGh * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(lds3 + 4 * W1_SLICEB + p0 * 128), 16, glaneD, so, 0, 0);
    }
  };
  dma_tile(tile);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbp = 0.f;
  // a lane's share of both fragment addresses: pixel parity `half`, channel l31
  const int aq = wave * W1_SLICEB + half * 128 + l31 * 4, bq = 4 * W1_SLICEB + half * 128 + l31 * 4;

  for (; tile < tEnd; tile += slots) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // everyone's quarter of the g_h tile is visible
    // 32 steps (tile row mb = i / 16, pixels 2 s, 2 s + 1 with s = i % 16): ten fragment reads (one per tap + g_h), issued one step ahead,
    // nine MFMAs into nine independent accumulators
    float av[2][9], bv[2];
    auto load_step = [&](int i, int buf) {
      const int mb = i >> 4, s = i & 15;
      bv[buf] = *reinterpret_cast<const float*>(ldsb + bq + (mb * 32 + 2 * s) * 128);
#pragma unroll
      for (int t = 0; t < 9; ++t) av[buf][t] = *reinterpret_cast<const float*>(ldsb + aq + ((mb + t / 3) * 34 + 2 * s + t % 3) * 128);
    };
    load_step(0, 0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int cur = i & 1;
      if (i + 1 < 32) load_step(i + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) asm volatile("v_max_f32 %0, 0, %0" : "+v"(av[cur][t]));      // the block's leading ReLU (:91)
      dbp += bv[cur];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][t], bv[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                       // the g_h tile is consumed (the slice is this wave's own)
    dma_tile(tile + slots);
  }

  // slab: [tap][ci][co] (lanes along co), then db1 from wave 0 (every wave summed the same g_h tile)
  float* const slab = a.ws + (size_t)blockIdx.x * W1_SLAB;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(t * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[t][r];
  dbp += lane_xor<32>(dbp);
  if (wave == 0 && half == 0) slab[9 * 128 * 32 + l31] = dbp;
}

// dW1[co][ci][tap] / db1 = the slabs summed in slab order (64 consecutive slab elements x 8 slab groups per workgroup, as the 1x1's reduce)
__global__ __launch_bounds__(512) void resblock_wgrad1_reduce_kernel(const float* __restrict__ ws, int nslabs, float* __restrict__ dw1, float* __restrict__ db1) {
  __shared__ float red[512];
  const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + li;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < W1_SLAB) {
    const float* p = ws + e;
    int b = grp;
    for (; b + 24 < nslabs; b += 32) {
      s0 += p[(size_t)b * W1_SLAB]; s1 += p[(size_t)(b + 8) * W1_SLAB]; s2 += p[(size_t)(b + 16) * W1_SLAB]; s3 += p[(size_t)(b + 24) * W1_SLAB];
    }
    for (; b < nslabs; b += 8) s0 += p[(size_t)b * W1_SLAB];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < W1_SLAB) {
    float s = red[li];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k * 64 + li];
    if (e < 9 * 128 * 32) {
      const int co = e & 31, ci = (e >> 5) & 127, t = e >> 12;
      dw1[(co * 128 + ci) * 9 + t] = s;
    } else if (db1) {
      db1[e - 9 * 128 * 32] = s;
    }
  }
}

}  // namespace

int64_t fo_resblock_wgrad1_halo_ws_bytes(const fo_conv_desc* d) {
  if (d->Cin != 128 || d->Cout != 32 || d->KD != 1 || d->KH != 3 || d->KW != 3) return 0;
  return (int64_t)2 * fo_cu_count() * W1_SLAB * 4;
}

// 1 = launched, 0 = not this kernel's geometry.  P = g_h [N][H][W][ldOut] (32), Q = x [N][H][W][ldIn] (128), FO_IN_RELU set.
int fo_resblock_wgrad1_halo_try(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal, float* dbias, float* ws,
                                int64_t ws_bytes, hipStream_t stream) {
  static const bool off = getenv("FACEOFF_NO_RESBLOCK_HALO") != nullptr;
  static const bool force = getenv("FACEOFF_FORCE_RESBLOCK_HALO") != nullptr;
  if (off || Areal != 32 || Breal != 128 || d->Cin != 128 || d->Cout != 32 || d->KD != 1 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->padH != 1 ||
      d->padW != 1 || d->Hm != d->Hin || d->Wm != d->Win || d->Win % 32 != 0 || d->Hin % 2 != 0 || !(d->flags & FO_IN_RELU) || d->ldIn % 4 != 0 ||
      d->ldOut % 4 != 0)
    return 0;
  W1Args a;
  a.x = Q; a.gh = P; a.ws = ws;
  a.N = d->N; a.H = d->Hin; a.W = d->Win; a.ldX = d->ldIn; a.ldGh = d->ldOut;
  a.tilesX = d->Win / 32; a.tilesY = d->Hin / 2; a.ntiles = d->N * a.tilesX * a.tilesY; a.perXcd = (a.ntiles + 7) / 8;
  const int cus = fo_cu_count();
  if (a.ntiles < 4 * cus && !force) return 0;
  const size_t npix = (size_t)d->N * d->Hin * d->Win;
  const size_t xB = (npix - 1) * d->ldIn * 4 + 512, gB = (npix - 1) * d->ldOut * 4 + 128;
  if (xB >= 0x7fffffffull || gB >= 0x7fffffffull) return 0;
  a.xBytes = (unsigned)xB; a.ghBytes = (unsigned)gB;
  const int grid = std::max(8, std::min((a.ntiles + 7) / 8 * 8, 2 * cus) / 8 * 8);
  if (ws_bytes < (int64_t)grid * W1_SLAB * 4) return 0;
  static fo_lds_once once;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(resblock_wgrad1_halo_kernel), W1_LDS, "resblock_wgrad1")) return 0;        // -> the tiled kernel
  FO_NOTE("resblock_wgrad1_halo_kernel");
  hipLaunchKernelGGL(resblock_wgrad1_halo_kernel, dim3(grid), dim3(256), W1_LDS, stream, a);
  hipLaunchKernelGGL(resblock_wgrad1_reduce_kernel, dim3((W1_SLAB + 63) / 64), dim3(512), 0, stream, ws, grid, dw, dbias);
  return 1;
}

extern "C" int64_t fo_resblock_bwd_conv3_ws_bytes(int64_t M) {
  if (M <= 0) return -1;
  const int ntiles = (int)((M + TP - 1) / TP);
  return (int64_t)grid_for(ntiles) * (128 * 32 + 128) * 4;
}

extern "C" int fo_resblock_bwd_conv3(int64_t M, const float* g, int ldG, const float* h, int ldH, const float* wp3, float* gh, int ldGh,
                                     float* dw3, float* db3, float* ws, int64_t ws_bytes, void* stream) {
  FO_REQUIRE(M > 0 && ldG >= 128 && ldH >= 32 && ldGh >= 32 && ldG % 4 == 0 && ldH % 4 == 0 && ldGh % 4 == 0, FO_E_SHAPE,
             "resblock_bwd_conv3: 128-channel gradient, 32-channel hidden tensors, 16-byte aligned pixels");
  const size_t gB = ((size_t)M - 1) * ldG * 4 + 512, hB = ((size_t)M - 1) * ldH * 4 + 128, ghB = ((size_t)M - 1) * ldGh * 4 + 128;
  FO_REQUIRE(gB < 0x7fffffffull && hB < 0x7fffffffull && ghB < 0x7fffffffull, FO_E_SHAPE, "resblock_bwd_conv3: tensors past the 2 GiB buffer window");
  FO_REQUIRE(ws_bytes >= fo_resblock_bwd_conv3_ws_bytes(M), FO_E_SHAPE, "resblock_bwd_conv3: workspace too small");
  RB3Args a;
  a.g = g; a.h = h; a.wp3 = wp3; a.gh = gh; a.ws = ws;
  a.M = (int)M; a.ldG = ldG; a.ldH = ldH; a.ldGh = ldGh; a.ntiles = (int)((M + TP - 1) / TP);
  a.gBytes = (unsigned)gB; a.hBytes = (unsigned)hB; a.ghBytes = (unsigned)ghB;
  const int grid = grid_for(a.ntiles);
  hipStream_t s = (hipStream_t)stream;
  static fo_lds_once once;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(resblock_bwd_conv3_kernel), LDS_BYTES, "resblock_bwd_conv3")) return FO_E_HIP;
  FO_NOTE("resblock_bwd_conv3_kernel");
  hipLaunchKernelGGL(resblock_bwd_conv3_kernel, dim3(grid), dim3(256), LDS_BYTES, s, a);
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(resblock_bwd_conv3_reduce_kernel, dim3((128 * 32 + 128 + 63) / 64), dim3(512), 0, s, ws, grid, dw3, db3);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
