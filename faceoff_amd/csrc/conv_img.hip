// The 8-channel image layer: Conv2d k4 s2 p1, 8 (6 real) -> 64 channels -- enc_b.blocks.0 forward (reference
// models/vqvae_conv3d_latent.py:108) and the data gradient of dec.blocks.6 (:160), both on 256 x 256 frames.
//
// K = 16 taps x 8 channels = 128: as a tiled GEMM that is four K-steps per tile, i.e. mostly prologue and epilogue (0.54 ms for
// 43 GFLOP, against 0.27 ms of matrix time).  Here nothing is staged through LDS: with K ordered (tap, channel) four consecutive
// k of one tap are 16 bytes of one input pixel, so a lane loads its operand fragments straight from global memory -- one
// 16-byte load per tap = four v_mfma_f32_32x32x2_f32 (the pairing of k between the two lane halves is free as long as both
// operands use the same one: MFMA i of tap t contracts channels i and 4 + i).  The 64 x 128 filter stays
// in LDS as ready-made fragments; a wave walks blocks of 32 consecutive output pixels (persistent grid, two waves per SIMD), re-issuing
// the load of tap t for the NEXT block right behind the MFMAs that consumed it (8 192 matrix cycles per block cover any miss).
// Operands are swapped (rows = output channels, columns = pixels), so a lane's accumulators are 4 consecutive channels of one
// pixel; a wave-local LDS patch turns them into 256-byte pixel rows for the epilogue (bias, ReLU mask, residual, ReLU).
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace {

struct ImgArgs {
  const float* in;
  const float* wp;
  const float* bias;
  const float* mask;
  const float* add;
  float* out;
  int N, Hin, Win, Hout, Wout, M;
  int ldOut, ldMask, ldAdd, flags, nblocks;
  unsigned inBytes, outBytes, maskBytes, addBytes;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
constexpr int PITCH = 64 + 4;   // floats per patch row

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// (no data-dependent control flow anywhere in the loop: with branches hipcc loses count of the outstanding loads and drains them
//  all -- vmcnt(0) -- in front of every store and every 16th MFMA group, which at one wave per SIMD halves the kernel's speed;
//  padding, the pixel tail and "no next block" are out-of-range buffer offsets instead: loads return zeros, stores are dropped)
template <bool MASK, bool ADD>
__global__ __launch_bounds__(256, 2) void conv_img_kernel(const ImgArgs a) {
  __shared__ __attribute__((aligned(16))) float patch_all[4][32 * PITCH];
  __shared__ f32x4 wfrag[32][64];                          // filter fragments [tap * 2 + block][lane]: 32 KB, read conflict-free
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  float* patch = patch_all[wave];
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.inBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.outBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASK ? a.mask : a.in), 0, MASK ? a.maskBytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ADD ? a.add : a.in), 0, ADD ? a.addBytes : 0, 0x00020000);

  // filter fragments: row (output channel) b*32 + l31, tap t, channels 4*half .. 4*half+3 -- the same for every wave, kept in LDS
  // (128 VGPRs per lane otherwise: one wave per SIMD, and nothing to cover a wave's epilogue)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int f = wave * 8 + q, t = f >> 1, b = f & 1;
    wfrag[f][lane] = *reinterpret_cast<const f32x4*>(a.wp + (size_t)(b * 32 + l31) * 128 + t * 8 + 4 * half);
  }
  __syncthreads();

  // The grid is sized so that one sweep of all waves (gridDim * 4 blocks of 32 pixels) is a whole number of frames, F: a wave's
  // block keeps its (oy, ox) for the whole kernel and only its frame advances, by F per iteration -- the per-iteration address
  // work is 16 adds + 16 selects instead of three integer divisions and 16 bounds checks (VALU instructions run on the ALUs the
  // fp32 MFMA uses: at one wave per SIMD every one of them is matrix time).
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int HWo = a.Hout * a.Wout;
  const int F = (nw * 32) / HWo;                           // frames per sweep (host guarantees divisibility)
  const int m0 = gw * 32 + l31;
  int n = m0 / HWo;                                        // uniform over the wave: HWo % 32 == 0
  const int rem = m0 - n * HWo;
  const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
  const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
  int rel[16];                                             // byte offset of tap t inside the frame, or -1 (padding)
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int kh = t >> 2, kw = t & 3;
    const bool ok = ((unsigned)(iy0 + kh) < (unsigned)a.Hin) & ((unsigned)(ix0 + kw) < (unsigned)a.Win);
    rel[t] = ok ? ((iy0 + kh) * a.Win + ix0 + kw) * 32 + half * 16 : -1;
  }
  const int frameIn = a.Hin * a.Win * 32;
  auto tap_offset = [&](int frame, int t) -> unsigned {
    return (frame < a.N && rel[t] >= 0) ? (unsigned)(frame * frameIn + rel[t]) : OOB;
  };
  // epilogue roles: 16 lanes x 16 B = one pixel's 64 channels, 4 pixels per pass
  const int c4 = lane & 15;
  int prel[8];                                             // float index of (pixel, c4) inside the frame's output
#pragma unroll
  for (int it = 0; it < 8; ++it) prel[it] = (rem - l31) + it * 4 + (lane >> 4);
  const f32x4 bv = (a.flags & FO_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float lo = (a.flags & FO_OUT_RELU) ? 0.f : -__builtin_inff();      // ReLU as a max with 0 or -inf: no branch

  f32x4 xf[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) xf[t] = bufload(rin, tap_offset(n, t));
  for (; n < a.N; n += F) {
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const f32x4 w0 = wfrag[2 * t][lane], w1 = wfrag[2 * t + 1][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[i], xf[t][i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[i], xf[t][i], acc[1], 0, 0, 0);
      }
      xf[t] = bufload(rin, tap_offset(n + F, t));          // the next block's tap t, behind the MFMAs that read this one
    }
    // acc[b][r]: channel b*32 + (r & 3) + 8 (r >> 2) + 4 half of pixel l31  ->  patch[pixel][channel]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
        *reinterpret_cast<f32x4*>(patch + l31 * PITCH + b * 32 + 8 * g + 4 * half) = v;
      }
    __builtin_amdgcn_wave_barrier();
    const long long pbase = (long long)n * HWo;            // first output pixel of this frame
    f32x4 mk[8], ad[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if (MASK) mk[it] = bufload(rmask, (unsigned)(((pbase + prel[it]) * a.ldMask + c4 * 4) * 4));
      if (ADD) ad[it] = bufload(radd, (unsigned)(((pbase + prel[it]) * a.ldAdd + c4 * 4) * 4));
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int pix = it * 4 + (lane >> 4);
      f32x4 v = *reinterpret_cast<const f32x4*>(patch + pix * PITCH + c4 * 4);
      v += bv;
      if (MASK) { v.x = mk[it].x > 0.f ? v.x : 0.f; v.y = mk[it].y > 0.f ? v.y : 0.f; v.z = mk[it].z > 0.f ? v.z : 0.f; v.w = mk[it].w > 0.f ? v.w : 0.f; }
      if (ADD) v += ad[it];
      v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (unsigned)(((pbase + prel[it]) * a.ldOut + c4 * 4) * 4), 0, 0);
    }
  }
}

}  // namespace

// Called by fo_conv_igemm for this geometry (returns 1 = not applicable: the tiled kernel runs).
int fo_conv_img_try(const fo_conv_desc* d, const float* in, const float* wp, const float* bias, const float* mask, const float* add,
                    float* out, hipStream_t stream) {
  static const bool off = [] { const char* e = getenv("FACEOFF_NO_IMG_KERNEL"); return e && atoi(e); }();
  if (off) return 1;
  if (!(d->Cin == 8 && d->ldIn == 8 && d->Cout == 64 && d->KD == 1 && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->ostride == 1 &&
        d->padH == 1 && d->padW == 1 && d->padD == 0 && d->Hm == d->Hout && d->Wm == d->Wout && 2 * d->Hm == d->Hin && 2 * d->Wm == d->Win &&
        !(d->flags & ~(FO_BIAS | FO_MASK | FO_ADD | FO_OUT_RELU)) && d->ldOut % 4 == 0))
    return 1;
  const long long M = (long long)d->N * d->Hm * d->Wm;
  const unsigned long long inBytes = (unsigned long long)d->N * d->Hin * d->Win * 32ull;
  const unsigned long long outBytes = ((unsigned long long)(M - 1) * d->ldOut + 64) * 4ull;
  const unsigned long long maskBytes = (d->flags & FO_MASK) ? ((unsigned long long)(M - 1) * d->ldMask + 64) * 4ull : 0;
  const unsigned long long addBytes = (d->flags & FO_ADD) ? ((unsigned long long)(M - 1) * d->ldAdd + 64) * 4ull : 0;
  if (M >= (1ll << 31) - 64 || inBytes >= (1ull << 31) || outBytes >= (1ull << 31) || maskBytes >= (1ull << 31) || addBytes >= (1ull << 31) ||
      M < 32 * 1024)
    return 1;                                             // small launches and > 2 GiB tensors: the tiled kernel
  if ((d->flags & FO_MASK) && (!mask || d->ldMask % 4)) return 1;
  if ((d->flags & FO_ADD) && (!add || d->ldAdd % 4)) return 1;
  ImgArgs a;
  a.in = in; a.wp = wp; a.bias = bias; a.mask = mask; a.add = add; a.out = out;
  a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.Hout = d->Hm; a.Wout = d->Wm; a.M = (int)M;
  a.ldOut = d->ldOut; a.ldMask = d->ldMask; a.ldAdd = d->ldAdd; a.flags = d->flags;
  a.nblocks = (int)((M + 31) / 32);
  a.inBytes = (unsigned)inBytes; a.outBytes = (unsigned)outBytes; a.maskBytes = (unsigned)maskBytes; a.addBytes = (unsigned)addBytes;
  const int HWo = d->Hm * d->Wm;
  if (HWo % 32) return 1;
  int grid = 0;                                           // gridDim * 128 pixels = a whole number of frames, two workgroups per CU
  for (int g = 2 * fo_cu_count(); g <= 8 * fo_cu_count(); g += 2 * fo_cu_count())
    if (((long long)g * 128) % HWo == 0) { grid = g; break; }
  if (!grid || (long long)grid * 128 > M) return 1;
  const bool hm = d->flags & FO_MASK, ha = d->flags & FO_ADD;
  if (hm) { if (ha) FO_NOTE_T("conv_img_kernel", true, true); else FO_NOTE_T("conv_img_kernel", true, false); }
  else { if (ha) FO_NOTE_T("conv_img_kernel", false, true); else FO_NOTE_T("conv_img_kernel", false, false); }
  if (hm && ha) hipLaunchKernelGGL((conv_img_kernel<true, true>), dim3(grid), dim3(256), 0, stream, a);
  else if (hm) hipLaunchKernelGGL((conv_img_kernel<true, false>), dim3(grid), dim3(256), 0, stream, a);
  else if (ha) hipLaunchKernelGGL((conv_img_kernel<false, true>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_img_kernel<false, false>), dim3(grid), dim3(256), 0, stream, a);
  if (hipGetLastError() != hipSuccess) return 1;
  return 0;
}
