// The bf16-operand engine's ResBlock convolutions as halo-tile kernels (config 3; reference models/vqvae_conv3d_latent.py:86-101).
//
// conv3x3_c128to32_halo_bf16_kernel: ReLU -> Conv2d(128, 32, 3, padding=1) -> ReLU (:91-93), bf16 in, bf16 out, fp32 accumulate.  On the bf16
// matrix pipe this layer is 48 GFLOP = 0.02 ms of MFMA per 64^2 block; the tiled kernel (conv_bf16_kernel<32>, which re-stages the input rows for
// every tap) took 0.140 ms against 210 MB = 0.042 ms of HBM traffic.  Here, as in resblock_halo.hip (the fp32 forward):
//   * four-wave workgroups, two per CU, walk tiles of 2 rows x 32 pixels; the contraction is split over the waves (wave w: input channels
//     32 w .. + 31; its 9 x 32 x 32 filter slice = 72 registers, resident for the launch);
//   * the tile's patch (4 x 34 pixels) is four WAVE-PRIVATE 32-channel slices of 64-byte pixels, DMA'd by their owners into the stage the
//     next tile will use while this tile is computed (two stages); the leading ReLU is applied ONCE, in place, when a slice has landed (every
//     patch element feeds up to nine taps);
//   * the four fp32 partial sums meet in LDS (each wave over its own, consumed, slice), + bias, ReLU, rounded to bf16 once: 16-byte stores.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ unsigned relu_pk(unsigned w) { return w & ~(((w & 0x80008000u) >> 15) * 0xffffu); }    // relu() of two packed bf16

struct C1Args {
  const __bf16* x;      // [N][H][W][ldX], 128 channels
  const __bf16* wp;     // [32][9][128]  (fo_pack_conv, rounded to bf16)
  const float* bias;    // [32]
  __bf16* h;            // [N][H][W][ldH], 32 channels
  int N, H, W, ldX, ldH;
  int tilesX, tilesY, ntiles, perXcd;
  unsigned xBytes, hBytes;
};

constexpr int PP = 36;                     // patch row pitch in pixels (34 used; a multiple of 4 keeps the granule swizzle a function of the column)
constexpr int SLB = 4 * PP * 64;           // a wave's slice: 4 rows x 36 pixels x 64 B = 9 216
constexpr int STB = 4 * SLB;               // a stage
constexpr int C1_LDS = 2 * STB;            // 73 728

// granule g (16 B = 8 channels) of the pixel in patch column c sits at position g ^ ((c >> 2) & 3): the 16 lanes of a ds_read_b128 group (16
// consecutive columns, one granule) then cover the 16 slots of a 256-byte bank row -- (c & 3) * 4 + (g ^ ((c >> 2) & 3)) is a bijection of c mod 16
__device__ __forceinline__ int gsw(int c) { return (c >> 2) & 3; }

__global__ __launch_bounds__(256, 2) void conv3x3_c128to32_halo_bf16_kernel(const C1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  lds_byte* const lds3 = (lds_byte*)ldsb;
  // (the descriptor starts ONE PIXEL before the tensor, so that the patch's left halo column is a non-negative offset)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x) - a.ldX, 0, a.xBytes + a.ldX * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, a.hBytes, 0x00020000);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int tEnd = min(a.ntiles, (xcd + 1) * a.perXcd);
  int tile = xcd * a.perXcd + slot;

  // DMA: a wave fills its own slice: 4 rows x 3 pieces of 16 pixels (the third: 2 pixels); lane = (pixel l / 4, granule position l % 4)
  const int lp = lane >> 2, pos = lane & 3;
  const unsigned dlane = (unsigned)(lp * a.ldX * 2 + ((pos ^ gsw(lp)) * 16));           // (columns 16 g + lp: the swizzle sees lp only)
  auto dma_tile = [&](int t, int stage) {
    if (t >= tEnd) return;
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iy = ty * 2 - 1 + r;
      const bool rowok = (unsigned)iy < (unsigned)a.H;
      const unsigned rowoff = rowok ? (unsigned)((((size_t)n * a.H + iy) * a.W + tx * 32) * a.ldX * 2) + wave * 64 : 0u;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const bool ok = rowok & !((g == 0) & (tx == 0) & (lp == 0)) & !((g == 2) & (tx == a.tilesX - 1) & (lp == 1));
        lds_byte* const dst = lds3 + stage * STB + wave * SLB + (r * PP + g * 16) * 64;
        if (g < 2 || lane < 8)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? dlane : OOB, rowoff + g * 16 * a.ldX * 2, 0, 0);
      }
    }
  };
  dma_tile(tile, 0);

  // filter slice: row operand of v_mfma_f32_32x32x16_bf16: row = output channel l31, k = input channels 32 wave + 16 ks + 8 half .. + 7 of tap t
  bf16x8 wf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const bf16x8*>(a.wp + (size_t)(l31 * 9 + t) * 128 + wave * 32 + ks * 16 + half * 8);
  // fragment address of (tap column kw, k-step ks): pixel column l31 + kw, granule 2 ks + half: cq[kw] ^ (ks << 5)
  int cq[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) cq[kw] = wave * SLB + (l31 + kw) * 64 + ((half ^ gsw(l31 + kw)) * 16);
  const int hp = tid >> 2, hc = (tid & 3) * 8;            // reduction role: pixel hp of the tile, hidden channels hc .. hc + 7
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + hc), b1 = *reinterpret_cast<const f32x4*>(a.bias + hc + 4);
  const unsigned hlane = (unsigned)((((hp >> 5) * a.W + (hp & 31)) * a.ldH + hc) * 2);

  for (int it = 0; tile < tEnd; tile += slots, ++it) {
    const int tx = tile % a.tilesX, r1 = tile / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    const int st = (it & 1) * STB;
    // this wave's slice has landed (the youngest vector-memory operation is the previous tile's store); the other stage is free: its partial
    // sums were consumed before the previous tile's second barrier
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    dma_tile(tile + slots, (it & 1) ^ 1);
    // the block's leading ReLU (:91), once, in place: 9 216 B per wave = 9 x 16 B per lane
    {
      unsigned char* const sl = ldsb + st + wave * SLB + lane * 16;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        u32x4 v = *reinterpret_cast<u32x4*>(sl + i * 1024);
        v.x = relu_pk(v.x); v.y = relu_pk(v.y); v.z = relu_pk(v.z); v.w = relu_pk(v.w);
        *reinterpret_cast<u32x4*>(sl + i * 1024) = v;
      }
    }
    // ---- 3x3 over this wave's channel slice: acc[mb] = 32 outputs x pixels of tile row mb; 18 steps (tap, ks) of two fragment reads + 2 MFMAs
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int s18 = 0; s18 < 18; ++s18) {
      const int tap = s18 >> 1, ks = s18 & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ldsb + st + ((cq[kw] ^ (ks << 5)) + (mb + kh) * PP * 64));
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap][ks], fb, acc[mb], 0, 0, 0);
      }
    }
    // ---- the four partial sums -> LDS, each wave over its own (consumed) slice: [pixel 0..63][128 B], granule g of pixel p at g ^ (p & 7);
    // the lane holds pixel 32 mb + l31, channels 8 j + 4 half .. + 3 in accumulator quad j
    unsigned char* const Pw = ldsb + st + wave * SLB + l31 * 128;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(Pw + mb * 32 * 128 + (((2 * j + half) ^ (l31 & 7)) * 16)) = f32x4{acc[mb][4 * j], acc[mb][4 * j + 1], acc[mb][4 * j + 2], acc[mb][4 * j + 3]};
    __syncthreads();
    {
      f32x4 s0 = b0, s1 = b1;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned char* const P = ldsb + st + w * SLB + hp * 128;
        s0 += *reinterpret_cast<const f32x4*>(P + (((hc >> 2) ^ (hp & 7)) * 16));
        s1 += *reinterpret_cast<const f32x4*>(P + ((((hc >> 2) + 1) ^ (hp & 7)) * 16));
      }
      const bf16x8 o = {(__bf16)fmaxf(s0.x, 0.f), (__bf16)fmaxf(s0.y, 0.f), (__bf16)fmaxf(s0.z, 0.f), (__bf16)fmaxf(s0.w, 0.f),
                        (__bf16)fmaxf(s1.x, 0.f), (__bf16)fmaxf(s1.y, 0.f), (__bf16)fmaxf(s1.z, 0.f), (__bf16)fmaxf(s1.w, 0.f)};
      const unsigned so = (unsigned)((((size_t)n * a.H + ty * 2) * a.W + tx * 32) * a.ldH * 2);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rh, hlane, so, 0);
    }
    __syncthreads();                                       // the partial sums are consumed: this stage may be refilled (by the next tile's top)
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// conv3x3_c32to128_halo_bf16_kernel: the data gradient of that convolution, g_x = conv3x3^T(g_h) * (x > 0) + g_out (:91-92 backwards), bf16 in /
// out.  The 32-channel side is the input: ONE shared 9 KB patch per tile (two stages, one barrier per tile), OUTPUT channels split over the
// waves (72 filter registers each, no partial sums).  48 GFLOP = 0.02 ms of MFMA against 546 MB of traffic: the kernel is a stream, and what
// matters is that the mask / residual loads of a tile are issued a whole tile before they are needed (two register sets).
struct D1Args {
  const __bf16* gh;     // [N][H][W][ldGh], 32 channels
  const __bf16* wpd;    // [128][9][32]  (fo_pack_conv_dgrad, rounded to bf16)
  const __bf16* mask;   // [N][H][W][ldM], 128 channels
  const __bf16* add;    // [N][H][W][ldA], 128 channels
  __bf16* out;          // [N][H][W][ldO], 128 channels
  int N, H, W, ldGh, ldM, ldA, ldO;
  int tilesX, tilesY, ntiles, perXcd;
  unsigned ghBytes, mBytes, aBytes, oBytes;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void conv3x3_c32to128_halo_bf16_kernel(const D1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];       // two patches of SLB bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  lds_byte* const lds3 = (lds_byte*)ldsb;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.gh) - a.ldGh, 0, a.ghBytes + a.ldGh * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.mask), 0, a.mBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.add), 0, a.aBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.oBytes, 0x00020000);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int tEnd = min(a.ntiles, (xcd + 1) * a.perXcd);
  int tile = xcd * a.perXcd + slot;

  // DMA: 4 patch rows x 3 pieces of 16 pixels (the third: 2 pixels): wave w fills row w
  const int lp = lane >> 2, pos = lane & 3;
  const unsigned dlane = (unsigned)(lp * a.ldGh * 2 + ((pos ^ gsw(lp)) * 16));
  auto dma_tile = [&](int t, int stage) {
    if (t >= tEnd) return;
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    const int iy = ty * 2 - 1 + wave;
    const bool rowok = (unsigned)iy < (unsigned)a.H;
    const unsigned rowoff = rowok ? (unsigned)((((size_t)n * a.H + iy) * a.W + tx * 32) * a.ldGh * 2) : 0u;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const bool ok = rowok & !((g == 0) & (tx == 0) & (lp == 0)) & !((g == 2) & (tx == a.tilesX - 1) & (lp == 1));
      lds_byte* const dst = lds3 + stage * SLB + (wave * PP + g * 16) * 64;
      if (g < 2 || lane < 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)dst, 16, ok ? dlane : OOB, rowoff + g * 16 * a.ldGh * 2, 0, 0);
    }
  };
  dma_tile(tile, 0);

  // filter block: row = output channel 32 wave + l31, k = tap * 32 + 16 ks + 8 half .. + 7
  bf16x8 wf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const bf16x8*>(a.wpd + (size_t)((wave * 32 + l31) * 9 + t) * 32 + ks * 16 + half * 8);
  int cq[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) cq[kw] = (l31 + kw) * 64 + ((half ^ gsw(l31 + kw)) * 16);
  const unsigned mlane = (unsigned)((l31 * a.ldM + wave * 32 + 4 * half) * 2);
  const unsigned alane = (unsigned)((l31 * a.ldA + wave * 32 + 4 * half) * 2);
  const unsigned olane = (unsigned)((l31 * a.ldO + wave * 32 + 4 * half) * 2);

  // The tile's mask and residual (accumulator quad j of row mb = channels 32 wave + 8 j + 4 half .. + 3 of pixel (row mb, column l31)) are
  // loaded ONE TILE AHEAD into the other of two register sets (the loop is unrolled by two so that the sets need no copies): at 3.7 TB/s of
  // traffic a load issued at the top of its own tile came back after the tile's 36 MFMAs were long done.
  u32x2 mk[2][2][4], ad[2][2][4];
  auto issue_loads = [&](int t, int set) {
    if (t >= tEnd) return;
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    const int pix0 = (n * a.H + ty * 2) * a.W + tx * 32;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mk[set][mb][j] = __builtin_amdgcn_raw_buffer_load_b64(rm, mlane + 16 * j, (pix0 + mb * a.W) * a.ldM * 2, 0);
        ad[set][mb][j] = __builtin_amdgcn_raw_buffer_load_b64(ra, alane + 16 * j, (pix0 + mb * a.W) * a.ldA * 2, 0);
      }
  };
  auto step = [&](int t, int set) {                       // set = stage = parity of the tile in this workgroup's walk (a compile-time constant at both call sites)
    const int tx = t % a.tilesX, r1 = t / a.tilesX;
    const int ty = r1 % a.tilesY, n = r1 / a.tilesY;
    const int st = set * SLB;
    // this wave's row of the patch has landed: the vector-memory operations younger than its DMAs are this tile's 16 mask / residual loads
    // and the previous tile's 8 stores; the barrier makes the patch whole and frees the other stage
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __syncthreads();
    dma_tile(t + slots, set ^ 1);
    issue_loads(t + slots, set ^ 1);
    const int pix0 = (n * a.H + ty * 2) * a.W + tx * 32;
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int s18 = 0; s18 < 18; ++s18) {
      const int tap = s18 >> 1, ks = s18 & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ldsb + st + ((cq[kw] ^ (ks << 5)) + (mb + kh) * PP * 64));
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap][ks], fb, acc[mb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 m = __builtin_bit_cast(bf16x4, mk[set][mb][j]), ar = __builtin_bit_cast(bf16x4, ad[set][mb][j]);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float c = (float)m[e] > 0.f ? acc[mb][4 * j + e] : 0.f;
          o[e] = (__bf16)(c + (float)ar[e]);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ro, olane + 16 * j, (pix0 + mb * a.W) * a.ldO * 2, 0);
      }
  };
  issue_loads(tile, 0);
  while (tile < tEnd) {
    step(tile, 0);
    tile += slots;
    if (tile >= tEnd) break;
    step(tile, 1);
    tile += slots;
  }
}

}  // namespace

// 1 = launched, 0 = not this kernel's geometry.  ReLU -> 3x3 pad-1 conv 128 -> 32 -> + bias -> ReLU on bf16 tensors.
int fo_conv3x3_c128to32_halo_bf16_try(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, void* out, hipStream_t stream) {
  static const bool off = getenv("FACEOFF_NO_RESBLOCK_HALO") != nullptr;
  static const bool force = getenv("FACEOFF_FORCE_RESBLOCK_HALO") != nullptr;
  if (off || d->Cin != 128 || d->Cout != 32 || d->KD != 1 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->ostride != 1 || d->padH != 1 ||
      d->padW != 1 || d->Hm != d->Hin || d->Wm != d->Win || d->Hout != d->Hin || d->Wout != d->Win || d->Win % 32 != 0 || d->Hin % 2 != 0 ||
      d->flags != (FO_IN_RELU | FO_BIAS | FO_OUT_RELU) || !bias || d->ldIn % 8 != 0 || d->ldOut % 8 != 0)
    return 0;
  C1Args a;
  a.x = reinterpret_cast<const __bf16*>(in); a.wp = reinterpret_cast<const __bf16*>(wp); a.bias = bias; a.h = reinterpret_cast<__bf16*>(out);
  a.N = d->N; a.H = d->Hin; a.W = d->Win; a.ldX = d->ldIn; a.ldH = d->ldOut;
  a.tilesX = d->Win / 32; a.tilesY = d->Hin / 2; a.ntiles = d->N * a.tilesX * a.tilesY; a.perXcd = (a.ntiles + 7) / 8;
  const int cus = fo_cu_count();
  if (a.ntiles < 4 * cus && !force) return 0;
  const size_t npix = (size_t)d->N * d->Hin * d->Win;
  const size_t xB = (npix - 1) * d->ldIn * 2 + 256, hB = (npix - 1) * d->ldOut * 2 + 64;
  if (xB >= 0x7fffffffull || hB >= 0x7fffffffull) return 0;
  a.xBytes = (unsigned)xB; a.hBytes = (unsigned)hB;
  static fo_lds_once once;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(conv3x3_c128to32_halo_bf16_kernel), C1_LDS, "resblock_bf16")) return 0;   // -> the tiled kernel
  FO_NOTE("conv3x3_c128to32_halo_bf16_kernel");
  hipLaunchKernelGGL(conv3x3_c128to32_halo_bf16_kernel, dim3(std::max(8, std::min((a.ntiles + 7) / 8 * 8, 2 * cus) / 8 * 8)), dim3(256), C1_LDS, stream, a);
  return 1;
}

// 1 = launched, 0 = not this kernel's geometry.  3x3 pad-1 conv 32 -> 128 with FO_MASK | FO_ADD on bf16 tensors (the ResBlock's data gradient).
int fo_conv3x3_c32to128_halo_bf16_try(const fo_conv_desc* d, const void* in, const void* wp, const void* mask, const void* add, void* out,
                                      hipStream_t stream) {
  static const bool off = getenv("FACEOFF_NO_RESBLOCK_HALO") != nullptr;
  static const bool force = getenv("FACEOFF_FORCE_RESBLOCK_HALO") != nullptr;
  if (off || d->Cin != 32 || d->Cout != 128 || d->KD != 1 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->ostride != 1 || d->padH != 1 ||
      d->padW != 1 || d->Hm != d->Hin || d->Wm != d->Win || d->Hout != d->Hin || d->Wout != d->Win || d->Win % 32 != 0 || d->Hin % 2 != 0 ||
      d->flags != (FO_MASK | FO_ADD) || !mask || !add || d->ldIn % 8 != 0 || d->ldOut % 4 != 0 || d->ldMask % 4 != 0 || d->ldAdd % 4 != 0)
    return 0;
  D1Args a;
  a.gh = reinterpret_cast<const __bf16*>(in); a.wpd = reinterpret_cast<const __bf16*>(wp); a.mask = reinterpret_cast<const __bf16*>(mask);
  a.add = reinterpret_cast<const __bf16*>(add); a.out = reinterpret_cast<__bf16*>(out);
  a.N = d->N; a.H = d->Hin; a.W = d->Win; a.ldGh = d->ldIn; a.ldM = d->ldMask; a.ldA = d->ldAdd; a.ldO = d->ldOut;
  a.tilesX = d->Win / 32; a.tilesY = d->Hin / 2; a.ntiles = d->N * a.tilesX * a.tilesY; a.perXcd = (a.ntiles + 7) / 8;
  const int cus = fo_cu_count();
  if (a.ntiles < 4 * cus && !force) return 0;
  const size_t npix = (size_t)d->N * d->Hin * d->Win;
  const size_t gB = (npix - 1) * d->ldIn * 2 + 64, mB = (npix - 1) * d->ldMask * 2 + 256, aB = (npix - 1) * d->ldAdd * 2 + 256, oB = (npix - 1) * d->ldOut * 2 + 256;
  if (gB >= 0x7fffffffull || mB >= 0x7fffffffull || aB >= 0x7fffffffull || oB >= 0x7fffffffull) return 0;
  a.ghBytes = (unsigned)gB; a.mBytes = (unsigned)mB; a.aBytes = (unsigned)aB; a.oBytes = (unsigned)oB;
  FO_NOTE("conv3x3_c32to128_halo_bf16_kernel");
  hipLaunchKernelGGL(conv3x3_c32to128_halo_bf16_kernel, dim3(std::max(8, std::min((a.ntiles + 7) / 8 * 8, 2 * cus) / 8 * 8)), dim3(256), 2 * SLB, stream, a);
  return 1;
}
