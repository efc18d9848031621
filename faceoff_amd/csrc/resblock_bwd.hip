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
  dbp += __shfl_xor(dbp, 32);
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
  static bool attr_set = false;
  if (!attr_set) {
    FO_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_bwd_conv3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess,
               FO_E_HIP, "resblock_bwd_conv3: cannot reserve %d bytes of LDS", LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(resblock_bwd_conv3_kernel, dim3(grid), dim3(256), LDS_BYTES, s, a);
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(resblock_bwd_conv3_reduce_kernel, dim3((128 * 32 + 128 + 63) / 64), dim3(512), 0, s, ws, grid, dw3, db3);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
