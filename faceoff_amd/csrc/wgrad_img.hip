// Filter gradient of the 8-channel image layers: Conv2d k4 s2 p1 8 (6) -> 64 (enc_b.blocks.0, reference
// models/vqvae_conv3d_latent.py:108) and ConvTranspose2d 64 -> 6 (dec.blocks.6, :160), both
//     dW[a][b][kh][kw] = sum over pixels m of P[m][a] * Q[2 oy - 1 + kh][2 ox - 1 + kw][b]        a < 64, b < 8,
// P = the 64-channel tensor on the H/2 x W/2 grid, Q = the 8-channel tensor on the H x W grid.
//
// As a GEMM this is [64] x [128 = 16 taps x 8] with the PIXELS as the contraction index: 43 GFLOP behind 1 GB of reads.  The tiled
// kernel (conv_wgrad_kernel<64, 32>) stages both operands through LDS and re-reads P once per 32-column tile: 0.55 ms.  Here a
// v_mfma_f32_32x32x2_f32 contracts TWO pixels per instruction (lane half h supplies pixel 2u + h), and both operand fragments are
// plain coalesced dword loads -- A: P[m][rb*32 + lane] (128 contiguous bytes per pixel), B: for filter row kh the 32 lanes are
// (kw, b) = 4 neighbouring input pixels x 8 channels = 128 contiguous bytes of Q.  Six loads feed eight MFMAs; nothing goes
// through LDS until the end, where a workgroup's four 64 x 128 partials are added into one slab wave after wave; a second
// kernel adds the slabs in a fixed order (deterministic) and writes dW in the checkpoint layout (+ the bias gradient = column
// sums of P, accumulated from the A fragments on the way).
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace {

struct WImgArgs {
  const float* P;
  const float* Q;
  float* ws;
  int N, Hm, Wm, ldP, M;          // P grid; Q grid is 2Hm x 2Wm, 8 floats per pixel
  int perWave;                    // pixels per wave (even)
  unsigned pBytes, qBytes;
};

constexpr unsigned OOB = 0x80000000u;
constexpr int SLAB = 64 * 128 + 64;   // floats per workgroup slab: D[64][128] + column sums of P
constexpr int U = 4;                  // pixel pairs per loop iteration

__device__ __forceinline__ float bufload1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

__global__ __launch_bounds__(256, 2) void wgrad_img_kernel(const WImgArgs a) {
  __shared__ float slab[SLAB];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int kw = l31 >> 3, b = l31 & 7;
  for (int i = tid; i < SLAB; i += 256) slab[i] = 0.f;
  __syncthreads();
  // Addresses: a lane-constant VGPR offset + a wave-uniform SCALAR offset (the buffer load's soffset) that walks the pixels, so the
  // loop's address arithmetic is scalar instructions -- the fp32 MFMA runs on the vector ALUs, and with only 8 MFMAs per pixel pair
  // every vector instruction per pair costs ~1 % of the kernel.  What remains per pair: 6 selects (padding -> out-of-range offset)
  // and the 2 bias adds.  The Q descriptor starts `margin` bytes below Q so that the scalar offset is never negative.
  const int Hq = 2 * a.Hm, Wq = 2 * a.Wm;
  const int margin = (Wq + 1) * 32;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, a.pBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.Q)) - margin, 0,
                                                                     a.qBytes + margin, 0x00020000);
  const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  const int m_begin = gw * a.perWave;                      // (even; M < 2^31 / 256)
  const int m_end = min(m_begin + a.perWave, a.M);
  const int steps = m_end > m_begin ? (m_end - m_begin + 1) / 2 : 0;
  // wave-uniform walk state: pixel pair (m_u, m_u + 1) = (n_u, oy_u, ox_u), (n_u, oy_u, ox_u + 1)
  int m_u = m_begin;
  int n_u = m_begin / (a.Hm * a.Wm);
  int oy_u = (m_begin - n_u * a.Hm * a.Wm) / a.Wm;
  int ox_u = m_begin - n_u * a.Hm * a.Wm - oy_u * a.Wm;
  unsigned soffP = (unsigned)m_u * a.ldP * 4;
  unsigned soffQ = (unsigned)(((n_u * Hq + 2 * oy_u - 1) * Wq + 2 * ox_u - 1) * 32 + margin);
  // lane constants
  unsigned vP[2], vQ[4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) vP[rb] = (unsigned)((half * a.ldP + rb * 32 + l31) * 4);
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) vQ[kh] = (unsigned)((half * 16 + kw * 8 + b) * 4 + kh * Wq * 32);
  const bool laneL = (half == 0) & (kw == 0), laneR = (half == 1) & (kw == 3);   // the taps that leave the image at the row ends

  f32x16 acc[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
  float bsum[2] = {0.f, 0.f};

  int done = 0;                                            // pairs loaded so far
  auto load_set = [&](float (&fa)[U][2], float (&fb)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool pv = done < steps;                        // (uniform)
      const bool badx = !pv | ((ox_u == 0) & laneL) | ((ox_u == a.Wm - 2) & laneR);
      fa[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, pv ? vP[0] : OOB, soffP, 0));
      fa[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, pv ? vP[1] : OOB, soffP, 0));
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const bool bad = badx | ((kh == 0) & (oy_u == 0)) | ((kh == 3) & (oy_u == a.Hm - 1));
        fb[u][kh] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, bad ? OOB : vQ[kh], soffQ, 0));
      }
      // scalar advance by one pair (selects, no branch); frames are contiguous, so the frame wrap needs no offset fix
      ++done;
      soffP += 2u * a.ldP * 4u;
      soffQ += 128u;
      ox_u += 2;
      const bool wrap = ox_u >= a.Wm;
      ox_u = wrap ? 0 : ox_u;
      soffQ += wrap ? (unsigned)(Wq * 32) : 0u;
      oy_u += wrap ? 1 : 0;
      oy_u = oy_u >= a.Hm ? 0 : oy_u;
    }
  };
  auto contract = [&](float (&fa)[U][2], float (&fb)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bsum[0] += fa[u][0];
      bsum[1] += fa[u][1];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][rb], fb[u][cb], acc[rb][cb], 0, 0, 0);
    }
  };
  // two register sets: the loads of the next U pairs fly during the MFMAs of the current ones (sets past the end load zeros)
  float fa0[U][2], fb0[U][4], fa1[U][2], fb1[U][4];
  load_set(fa0, fb0);
  for (int s0 = 0; s0 < steps; s0 += 2 * U) {
    load_set(fa1, fb1);
    contract(fa0, fb0);
    load_set(fa0, fb0);
    contract(fa1, fb1);
  }
  // acc[rb][cb][r]: row a = rb*32 + (r & 3) + 8 (r >> 2) + 4 half, column n = cb*32 + l31.  The four waves add their partials to
  // the slab one after the other (every element has one owner lane per wave): a fixed order, so the sum is reproducible.
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[row * 128 + cb * 32 + l31] += acc[rb][cb][r];
          }
      // column sums of P: the two lane halves hold the same channel
      const float b0 = bsum[0] + lane_xor<32>(bsum[0]), b1 = bsum[1] + lane_xor<32>(bsum[1]);
      if (half == 0) {
        slab[64 * 128 + l31] += b0;
        slab[64 * 128 + 32 + l31] += b1;
      }
    }
    __syncthreads();
  }
  float* dst = a.ws + (size_t)blockIdx.x * SLAB;
  for (int i = tid; i < SLAB; i += 256) dst[i] = slab[i];
}

// dW[a][b][kh][kw] = sum over slabs of D[a][kh*32 + kw*8 + b]   (b < Breal);   dbias[a] = sum of the column sums.
// 64 elements per workgroup; four groups of lanes walk every fourth slab with eight loads in flight each, then the four partial
// sums are added in a fixed order (the result does not depend on timing).
__global__ __launch_bounds__(256) void wgrad_img_reduce_kernel(const float* __restrict__ ws, int nslabs, float* __restrict__ dw, int Areal,
                                                               int Breal, float* __restrict__ dbias) {
  __shared__ float part[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < SLAB) {
    for (int k0 = kg; k0 < nslabs; k0 += 32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 4 * j;
        acc[j] += k < nslabs ? ws[(size_t)k * SLAB + e] : 0.f;
      }
    }
  }
  part[kg][threadIdx.x & 63] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (kg != 0 || e >= SLAB) return;
  const float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
  if (e < 64 * 128) {
    const int arow = e >> 7, ncol = e & 127, kh = ncol >> 5, kw = (ncol >> 3) & 3, b = ncol & 7;
    if (arow < Areal && b < Breal) dw[((size_t)arow * Breal + b) * 16 + kh * 4 + kw] = s;
  } else if (dbias && e - 64 * 128 < Areal) {
    dbias[e - 64 * 128] = s;
  }
}

bool applicable(const fo_conv_desc* d, int Areal, int Breal) {
  static const bool off = [] { const char* e = getenv("FACEOFF_NO_IMG_KERNEL"); return e && atoi(e); }();
  if (off) return false;
  const long long M = (long long)d->N * d->Hm * d->Wm;
  return d->Cin == 8 && d->ldIn == 8 && d->Cout == 64 && Areal <= 64 && Breal <= 8 && d->KD == 1 && d->KH == 4 && d->KW == 4 && d->stride == 2 &&
         d->padH == 1 && d->padW == 1 && d->padD == 0 && 2 * d->Hm == d->Hin && 2 * d->Wm == d->Win && d->Wm % 2 == 0 && d->ldOut >= 64 &&
         !(d->flags & FO_IN_RELU) && M >= 256 * 1024 && M * d->ldOut * 4 < (1ll << 31) - (1 << 20) && M * 4 * 32 < (1ll << 31) - (1 << 20);
}

int nslabs() { return 2 * fo_cu_count(); }

}  // namespace

int64_t fo_wgrad_img_ws_bytes(const fo_conv_desc* d) { return applicable(d, 64, 8) ? (int64_t)nslabs() * SLAB * 4 + 256 : 0; }

// 0 = launched, 1 = geometry not applicable (the tiled kernel runs)
int fo_wgrad_img_try(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal, float* dbias, float* ws,
                     int64_t ws_bytes, hipStream_t stream) {
  if (!applicable(d, Areal, Breal) || ws_bytes < (int64_t)nslabs() * SLAB * 4 || !ws) return 1;
  WImgArgs a;
  a.P = P; a.Q = Q; a.ws = ws;
  a.N = d->N; a.Hm = d->Hm; a.Wm = d->Wm; a.ldP = d->ldOut;
  const long long M = (long long)d->N * d->Hm * d->Wm;
  a.M = (int)M;
  const int waves = nslabs() * 4;
  a.perWave = (int)(((M + waves - 1) / waves + 1) / 2 * 2);
  a.pBytes = (unsigned)(((unsigned long long)(M - 1) * d->ldOut + 64) * 4ull);
  a.qBytes = (unsigned)((unsigned long long)M * 4 * 32ull);
  FO_NOTE("wgrad_img_kernel");
  hipLaunchKernelGGL(wgrad_img_kernel, dim3(nslabs()), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(wgrad_img_reduce_kernel, dim3((SLAB + 63) / 64), dim3(256), 0, stream, ws, nslabs(), dw, Areal, Breal, dbias);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
