// Shared helpers for the FaceOff gfx950 kernels (internal; the public ABI is include/faceoff_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "faceoff_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void fo_set_error(const char* fmt, ...);

// lane_xor<K>(v): the value of lane (l ^ K) -- what __shfl_xor(v, K) returns -- on the VECTOR ALU.  hipcc compiles __shfl_xor to ds_bpermute_b32, an
// LDS-pipeline instruction with its round trip; a butterfly reduction is a chain of them.  gfx950 has the moves on the VALU: K = 1, 2 are DPP quad
// permutations, K = 4 two bank-masked DPP row shifts (banks of four lanes: even banks take lane + 4, odd banks lane - 4), K = 8 a rotation of the
// 16-lane row by eight, K = 16 / 32 the row swaps v_permlane16_swap / v_permlane32_swap (swap(v, v): one result holds the lower partner's rows, the
// other the upper's; each lane picks the one that is not its own).  Same partner, same bits as the shuffle (round 6: the pooled epilogue of the
// halo-tile kernel, the LPIPS heads' group sums, the discriminator heads' wave sums).
template <int K> __device__ __forceinline__ int lane_xor_i(int v) {
  static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "lane_xor: a power of two below 64");
#ifdef FO_LANE_XOR_SHFL                                                                         // A/B build: the ds_bpermute form (tools/r06 A/B)
  return __shfl_xor(v, K);
#endif
  if constexpr (K == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
  else if constexpr (K == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
  else if constexpr (K == 4) {
    const int up = __builtin_amdgcn_update_dpp(0, v, 0x104 /* row_shl:4: lane l takes l + 4 */, 0xF, 0x5, false);
    return __builtin_amdgcn_update_dpp(up, v, 0x114 /* row_shr:4: lane l takes l - 4 */, 0xF, 0xA, false);
  } else if constexpr (K == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128 /* row_ror:8 */, 0xF, 0xF, true);
  else if constexpr (K == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);     // r[0]: rows 0 0 2 2, r[1]: rows 1 1 3 3
    return (__lane_id() & 16) ? (int)r[0] : (int)r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);     // r[0]: lower half twice, r[1]: upper half twice
    return (__lane_id() & 32) ? (int)r[0] : (int)r[1];
  }
}
template <int K> __device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor_i<K>(__float_as_int(v))); }
template <int K> __device__ __forceinline__ int lane_xor(int v) { return lane_xor_i<K>(v); }
// v + the values of the other lanes of its aligned group of W lanes, by the xor butterfly W/2, W/4, .., 1: the association order of
// `for (o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o)`, hence the same bits
template <int W> __device__ __forceinline__ float group_sum_valu(float v) {
  if constexpr (W >= 64) v += lane_xor<32>(v);
  if constexpr (W >= 32) v += lane_xor<16>(v);
  if constexpr (W >= 16) v += lane_xor<8>(v);
  if constexpr (W >= 8) v += lane_xor<4>(v);
  if constexpr (W >= 4) v += lane_xor<2>(v);
  if constexpr (W >= 2) v += lane_xor<1>(v);
  return v;
}


// Kernel notes (fo_kernel_notes / fo_last_kernel, include/faceoff_hip.h): while enabled, every launcher records the symbol of the MAIN kernel it
// launches -- name and template arguments exactly as rocprofv3 prints them -- so that bench.py's per-kernel entries carry the names found in
// profiles/*_kernel_stats.md.  FO_NOTE("wino_gemm_kernel"); FO_NOTE_T("conv_bf16_pp16_kernel", BMB, BN, WAVES_M, WAVES_N) -> "...<256, 256, 2, 4>".
extern std::atomic<int> faceoff_notes_on;
void fo_note_kernel(const char* base, const char* pretty_targs);
template <auto... V> struct fo_vals {};
template <class T> const char* fo_tname() { return __PRETTY_FUNCTION__; }      // "... [T = fo_vals<256, 256, 2, 4>]": clang prints what the demangler prints
#define FO_NOTE(base) do { if (faceoff_notes_on.load(std::memory_order_relaxed)) fo_note_kernel(base, nullptr); } while (0)
#define FO_NOTE_T(base, ...) do { if (faceoff_notes_on.load(std::memory_order_relaxed)) fo_note_kernel(base, fo_tname<fo_vals<__VA_ARGS__>>()); } while (0)
int fo_cu_count();  // compute units of the current device (cached; 256 on MI355X)
// wgrad_img.hip: the filter gradient of the same layers (pixels as the contraction index, operands straight from global memory)
int64_t fo_wgrad_img_ws_bytes(const fo_conv_desc* d);
int fo_wgrad_img_try(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal, float* dbias, float* ws,
                     int64_t ws_bytes, hipStream_t stream);
// conv_img.hip: the 8 -> 64 channel k4 s2 p1 image layer without LDS staging; 0 = launched, 1 = geometry not applicable
int fo_conv_img_try(const fo_conv_desc* d, const float* in, const float* wp, const float* bias, const float* mask, const float* add,
                    float* out, hipStream_t stream);

// resblock_halo.hip: ResBlock forward as a halo-tile kernel; 1 = launched, 0 = geometry not applicable
int fo_resblock_halo_try(const fo_conv_desc* d, const float* x, const float* wp1, const float* b1, const float* wp3, const float* b3, float* hbuf,
                         float* out, int ldOut2, int out_relu, hipStream_t stream);

// resblock_bwd.hip: 3x3 pad-1 stride-1 conv 32 -> 128 with mask / add epilogue (a ResBlock's first conv, backwards) as a halo-tile kernel;
// 1 = launched, 0 = geometry not applicable
int fo_conv3x3_c32_halo_try(const fo_conv_desc* d, const float* in, const float* wp, const float* mask, const float* add, float* out, hipStream_t stream);

// resblock_bwd.hip: filter + bias gradient of a ResBlock's first conv (3x3 128 -> 32, ReLU'd input) as a halo-tile kernel; 1 = launched
int64_t fo_resblock_wgrad1_halo_ws_bytes(const fo_conv_desc* d);
int fo_resblock_wgrad1_halo_try(const fo_conv_desc* d, const float* P, const float* Q, float* dw, int Areal, int Breal, float* dbias, float* ws,
                                int64_t ws_bytes, hipStream_t stream);

// resblock_bf16.hip: ReLU -> 3x3 conv 128 -> 32 -> bias -> ReLU on bf16 tensors as a halo-tile kernel; 1 = launched
int fo_conv3x3_c128to32_halo_bf16_try(const fo_conv_desc* d, const void* in, const void* wp, const float* bias, void* out, hipStream_t stream);

int fo_conv3x3_c32to128_halo_bf16_try(const fo_conv_desc* d, const void* in, const void* wp, const void* mask, const void* add, void* out,
                                      hipStream_t stream);

// elementwise.hip: out[c] = sum of the nblk partial rows ws[b][C], c < Creal (the second stage of every column sum).
// Internal (C++ linkage): exports.map keeps everything but the C names of include/faceoff_hip.h out of the dynamic symbol table.
int fo_colsum_finish(const float* ws, float* out, int nblk, int C, int Creal, void* stream);
// elementwise.hip: out[0] = part[0] + ... + part[n-1] in a fixed order (one wave): how every loss scalar leaves its kernel's per-workgroup partials
int fo_ordered_sum(const float* part, int n, float* out, void* stream);

#define FO_CHECK_LAUNCH()                                                     \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      fo_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return FO_E_HIP;                                                        \
    }                                                                         \
  } while (0)

#define FO_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) {                  \
      fo_set_error(__VA_ARGS__);    \
      return (code);                \
    }                               \
  } while (0)

// Opt a kernel in to more than the default 64 KB of dynamic LDS.  hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute
// (fo_comm_init / hipSetDevice allow several devices in one process), so the "already done" state is a bit per device, and atomic because
// two host threads may launch on distinct streams.  Returns false (with fo_last_error set) if the runtime refuses.
struct fo_lds_once { std::atomic<unsigned long long> devs{0}; };
static inline bool fo_lds_optin(fo_lds_once& once, const void* kern, int bytes, const char* who) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (dev < 64 && (once.devs.load(std::memory_order_acquire) & bit)) return true;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    fo_set_error("%s: cannot reserve %d bytes of LDS on device %d: %s", who, bytes, dev, hipGetErrorString(e));
    return false;
  }
  if (dev < 64) once.devs.fetch_or(bit, std::memory_order_release);
  return true;
}

static inline bool fo_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Blocks b and b+8 share an XCD (observed round-robin placement; speed only).  Give each XCD a
// contiguous range of logical tiles so neighbouring tiles (shared halos / shared filter panels)
// hit the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int fo_xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
