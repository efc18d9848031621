// The Winograd-domain GEMMs of wino_gemm.hip with the fp32 products formed on the bf16 matrix pipe.
//
// gfx950 multiplies bf16 at 16x the rate of fp32 (v_mfma_f32_16x16x32_bf16: 16 K FLOP in 16 cycles; v_mfma_f32_32x32x2_f32:
// 4 K FLOP in 64), and the fp32 instruction also occupies the SIMD's vector ALUs.  An fp32 number is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 significand bits, same exponent range):  x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1) (the subtractions are exact).  So
//     a * b = sum_{i,j} a_i b_j,        |a_i b_j| <= 2^(-8(i+j)) |a b|
// and keeping the six terms with i + j <= 2 leaves a relative error of 2 * 2^-24 + 2^-32 per product -- the size of ONE fp32
// rounding -- with the sum accumulated in fp32 by the MFMA exactly as before.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA
// they replace.  This is the same arithmetic contract as an fp32 FMA chain to within the last bit or two (tests/
// test_split_gpu.py measures both kernels against an fp64 product), not a reduced-precision mode: nothing is rounded to bf16
// that is not carried by another piece.
//
// Same interface, tile walk and K segments as wino_gemm_kernel (that file has the derivation): persistent workgroups, 128 x 128
// tiles, 32-deep K-steps, depth taps as K segments with per-row validity.  What differs:
//   * a K-step's operands are loaded as fp32 (the HBM traffic is unchanged), split in registers (11 VALU instructions per pair
//     of values: 3 packed converts, 4 re-expansions, 4 subtractions) and written to LDS as three bf16 planes per operand:
//     [piece][row][32 k] = 64-byte rows, chunk c (8 k) of row r at position c ^ swz(r) -- conflict-free for the ds_read_b128
//     fragment reads of the 16x16x32 MFMA and for the ds_write_b64 of the split;
//   * LDS is single-buffered (48 KB, two workgroups per CU): the global loads run TWO K-steps ahead in registers (a K-step is
//     ~1500 matrix cycles, less than an HBM round trip under load), the split + LDS writes sit between two barriers, and the CU's other workgroup has the matrix pipe meanwhile (a
//     bf16 MFMA does not use the vector ALUs, so one wave's split overlaps the other's MFMAs);
//   * the filter fragment is the MFMA's row operand, so an accumulator register quartet is 4 consecutive output channels of
//     one row: the epilogue is one 16-byte store per 16x16 block and lane, 64-byte runs per row.
#include "common.h"
#include <stdlib.h>

namespace {

struct WGArgs {
  const float* V;
  const float* U;
  float* M;
  int tiles, tilesN, tilesPerPlane;
  unsigned planeMagic;
  int P, pShift;
  unsigned pMagic;
  int T;
  unsigned tMagic;
  int KD, padD;
  int cinChunks, Ktot;
  int ldV, ldM;
  unsigned bankBytes;
  unsigned vBytes, uBytes, mBytes;
  unsigned margin;
};

constexpr int BM = 128, BN = 128, BK = 32;
constexpr unsigned OOB = 0x80000000u;
constexpr int PIECE = 128 * 64;            // bytes of one bf16 plane of one operand tile
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned udiv(unsigned x, int d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
__device__ __forceinline__ unsigned pk(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// four fp32 -> three pieces of four bf16 each (exact: x == p0 + p1 + p2)
__device__ __forceinline__ void split4(const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float a = x[2 * h], b = x[2 * h + 1];
    const unsigned q0 = pk(a, b);
    const float a1 = a - __uint_as_float(q0 << 16), b1 = b - __uint_as_float(q0 & 0xffff0000u);
    const unsigned q1 = pk(a1, b1);
    const float a2 = a1 - __uint_as_float(q1 << 16), b2 = b1 - __uint_as_float(q1 & 0xffff0000u);
    p0[h] = q0; p1[h] = q1; p2[h] = pk(a2, b2);
  }
}

template <int DIAG>
__global__ __launch_bounds__(256, 2) void wino_gemm_split_kernel(const WGArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * PIECE];
  __shared__ int fifo[4][4];                 // tiles the loader has entered and the matrix side has not: row0, tile_n, K-steps
  unsigned char* As0 = lds;
  unsigned char* Bs0 = lds + 3 * PIECE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.V) - a.margin), 0, a.vBytes + a.margin, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U), 0, a.uBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, a.mBytes, 0x00020000);

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int lo = (int)(((long long)a.tiles * xcd) >> 3), hi = (int)(((long long)a.tiles * (xcd + 1)) >> 3);
  int ld_tile = lo + slot;
  if (ld_tile >= hi) return;

  // ---- loader state (one K-step ahead of the matrix pipe), as in wino_gemm_kernel
  unsigned ld_rowoff[4], ld_mask[4], ld_wrow[4];
  int ld_kd = 0, ld_kd_hi = 0, ld_chunk = 0;
  int fifo_w = 0, fifo_r = 0;
  bool ld_live = true;

  auto setup = [&](int tile) {
    const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
    const int row0 = tile_m * BM;
    const int plane = (int)udiv((unsigned)tile_m, a.tilesPerPlane, a.planeMagic);
    const unsigned boff = (unsigned)plane * a.bankBytes;
    unsigned any = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + lrow + 32 * i;
      const unsigned F = a.pShift >= 0 ? (unsigned)r >> a.pShift : udiv((unsigned)r, a.P, a.pMagic);
      const int t = (int)(F - udiv(F, a.T, a.tMagic) * (unsigned)a.T);
      unsigned bad = 0;
      for (int kd = 0; kd < a.KD; ++kd) bad |= ((unsigned)(t + kd - a.padD) < (unsigned)a.T ? 0u : 1u) << kd;
      ld_mask[i] = bad;
      ld_rowoff[i] = (unsigned)(r * a.ldV + lcol) * 4u;
      ld_wrow[i] = boff + (unsigned)((tile_n * BN + lrow + 32 * i) * a.Ktot + lcol) * 4u;
    }
    {
      const unsigned F0 = a.pShift >= 0 ? (unsigned)row0 >> a.pShift : udiv((unsigned)row0, a.P, a.pMagic);
      const unsigned F1 = a.pShift >= 0 ? (unsigned)(row0 + BM - 1) >> a.pShift : udiv((unsigned)(row0 + BM - 1), a.P, a.pMagic);
      if (F1 - F0 > 1) any = (1u << a.KD) - 1;
      else {
        const int t0 = (int)(F0 - udiv(F0, a.T, a.tMagic) * (unsigned)a.T), t1 = (int)(F1 - udiv(F1, a.T, a.tMagic) * (unsigned)a.T);
        for (int kd = 0; kd < a.KD; ++kd)
          any |= ((((unsigned)(t0 + kd - a.padD) < (unsigned)a.T) | ((unsigned)(t1 + kd - a.padD) < (unsigned)a.T)) ? 1u : 0u) << kd;
      }
    }
    int klo = 0, khi = a.KD;
    while (klo < a.KD - 1 && !((any >> klo) & 1)) ++klo;
    while (khi > klo + 1 && !((any >> (khi - 1)) & 1)) --khi;
    ld_kd = klo; ld_kd_hi = khi; ld_chunk = 0;
    if (tid == 0) { fifo[fifo_w & 3][0] = row0; fifo[fifo_w & 3][1] = tile_n; fifo[fifo_w & 3][2] = (khi - klo) * a.cinChunks; }
    ++fifo_w;
  };

  f32x4 ra[2][4], rb[2][4];                  // two K-steps of operands in flight
  auto load_step = [&](int set) {
    const int soffA = (ld_kd * a.P * a.ldV + ld_chunk * BK) * 4;
    const int soffB = (ld_kd * a.cinChunks + ld_chunk) * (BK * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra[set][s] = bufload(rin, DIAG == 5 ? OOB : (((ld_mask[s] >> ld_kd) << 31) | ld_rowoff[s]), soffA);
      rb[set][s] = bufload(rwp, ld_wrow[s], soffB);
    }
    if (++ld_chunk == a.cinChunks) {
      ld_chunk = 0;
      if (++ld_kd == ld_kd_hi) {
        ld_tile += per;
        if (ld_live && ld_tile < hi) setup(ld_tile);
        else {
          ld_live = false;
          ld_kd = 0; ld_kd_hi = 1 << 30;
#pragma unroll
          for (int i = 0; i < 4; ++i) { ld_rowoff[i] = OOB; ld_wrow[i] = OOB; ld_mask[i] = 0; }
        }
      }
    }
  };
  // this thread's 8 bytes (4 k) of rows lrow + 32 s in each plane
  unsigned wr_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = lrow + 32 * s;
    wr_off[s] = (unsigned)(row * 64 + ((((tid & 7) >> 1) ^ swz(row)) << 4) + (tid & 1) * 8);
  }
  auto split_store = [&](int set) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x2 p0, p1, p2;
      if (DIAG == 1) { p0 = u32x2{__float_as_uint(ra[set][s][0]), __float_as_uint(ra[set][s][1])}; p1 = u32x2{__float_as_uint(ra[set][s][2]), __float_as_uint(ra[set][s][3])}; p2 = p0; }
      else split4(ra[set][s], p0, p1, p2);
      if (DIAG == 3) { if (p0[0] == 0x12345678u) *reinterpret_cast<u32x2*>(As0 + wr_off[s]) = p0 + p1 + p2; } else {
      *reinterpret_cast<u32x2*>(As0 + wr_off[s]) = p0;
      *reinterpret_cast<u32x2*>(As0 + PIECE + wr_off[s]) = p1;
      *reinterpret_cast<u32x2*>(As0 + 2 * PIECE + wr_off[s]) = p2; }
      if (DIAG == 1) { p0 = u32x2{__float_as_uint(rb[set][s][0]), __float_as_uint(rb[set][s][1])}; p1 = u32x2{__float_as_uint(rb[set][s][2]), __float_as_uint(rb[set][s][3])}; p2 = p0; }
      else split4(rb[set][s], p0, p1, p2);
      if (DIAG == 3) { if (p0[0] == 0x12345678u) *reinterpret_cast<u32x2*>(Bs0 + wr_off[s]) = p0 + p1 + p2; } else {
      *reinterpret_cast<u32x2*>(Bs0 + wr_off[s]) = p0;
      *reinterpret_cast<u32x2*>(Bs0 + PIECE + wr_off[s]) = p1;
      *reinterpret_cast<u32x2*>(Bs0 + 2 * PIECE + wr_off[s]) = p2; }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row (block base + l15), chunk quad
  const unsigned fr = (unsigned)(l15 * 64 + ((quad ^ swz(l15)) << 4));     // block bases are multiples of 16 rows
  const unsigned char* Af = As0 + (wm * 64) * 64 + fr;
  const unsigned char* Bf = Bs0 + (wn * 64) * 64 + fr;

  setup(ld_tile);
  load_step(0);
  load_step(1);
  __syncthreads();                             // (the first tile's FIFO entry)
  int cur_row0 = fifo[0][0], cur_tn = fifo[0][1], cur_left = fifo[0][2];
  fifo_r = 1;
  split_store(0);
  __syncthreads();
  auto body = [&](int set) -> bool {           // one K-step; `set` = the register set K-step n+2 is loaded into (= the one K-step n used)
    load_step(set);                            // K-step n+2: in flight during the MFMAs of n and n+1
    bf16x8 fa[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const bf16x8*>(Af + p * PIECE + i * (16 * 64));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(Bf + p * PIECE + j * (16 * 64));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (DIAG == 2) { acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0] + fb[1] + fb[2], fa[i][0] + fa[i][1] + fa[i][2], acc[i][j], 0, 0, 0); continue; }
        // smallest terms first; the filter is the row operand (accumulator = 4 consecutive output channels of row l15)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][0], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                           // every wave has read this K-step's planes
    if (--cur_left == 0) {
      const unsigned voff = (unsigned)((cur_row0 + wm * 64 + l15) * a.ldM + cur_tn * BN + wn * 64 + quad * 4) * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (DIAG != 4 || acc[i][j][0] == 1.2345f)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, acc[i][j]), rout,
                                                 voff + j * 64, i * 16 * a.ldM * 4, 0);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (fifo_r == fifo_w) return false;      // the loader entered no further tile: that was this workgroup's last
      cur_row0 = fifo[fifo_r & 3][0]; cur_tn = fifo[fifo_r & 3][1]; cur_left = fifo[fifo_r & 3][2];
      ++fifo_r;
    }
    split_store(set ^ 1);                      // K-step n+1
    __syncthreads();
    return true;
  };
  while (true) {
    if (!body(0)) break;
    if (!body(1)) break;
  }
}

static unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

}  // namespace

extern "C" int fo_wino_gemm_split(const float* V, const float* U, float* M, int planes, int N, int T, int P, int Cin, int Cout, int KD,
                                  void* stream) {
  FO_REQUIRE(V && U && M && planes > 0 && N > 0 && T > 0 && N % T == 0 && P > 0, FO_E_SHAPE, "wino_gemm_split: bad sizes");
  FO_REQUIRE(KD == 1 || KD == 3, FO_E_SHAPE, "wino_gemm_split: KD must be 1 or 3 (got %d)", KD);
  FO_REQUIRE(Cin % 32 == 0 && Cin >= 64 && Cout % 128 == 0, FO_E_SHAPE, "wino_gemm_split: Cin %% 32 == 0, Cin >= 64, Cout %% 128 == 0");
  FO_REQUIRE(((long long)N * P) % 128 == 0, FO_E_SHAPE, "wino_gemm_split: a plane (N * P = %lld rows) must be whole 128-row tiles",
             (long long)N * P);
  FO_REQUIRE(fo_aligned16(V) && fo_aligned16(U) && fo_aligned16(M), FO_E_ALIGN, "wino_gemm_split: 16-byte alignment");
  const long long rows = (long long)planes * N * P;
  const unsigned long long vBytes = (unsigned long long)rows * Cin * 4ull, mBytes = (unsigned long long)rows * Cout * 4ull;
  const unsigned long long bankBytes = (unsigned long long)Cout * KD * Cin * 4ull, uBytes = bankBytes * planes;
  const unsigned long long margin = (unsigned long long)(KD / 2) * P * Cin * 4ull;
  FO_REQUIRE(vBytes + 2 * margin < (1ull << 31) && mBytes < (1ull << 31) && uBytes < (1ull << 31), FO_E_SHAPE,
             "wino_gemm_split: plane stack exceeds the 2 GiB buffer-descriptor window");
  WGArgs a;
  a.V = V; a.U = U; a.M = M;
  a.tilesN = Cout / 128;
  a.tilesPerPlane = (int)(((long long)N * P) / 128);
  a.tiles = (int)(rows / 128) * a.tilesN;
  a.planeMagic = magic_of((unsigned)a.tilesPerPlane);
  a.P = P; a.pShift = -1;
  for (int s = 0; s < 30; ++s) if ((1 << s) == P) a.pShift = s;
  a.pMagic = magic_of((unsigned)P);
  a.T = T; a.tMagic = magic_of((unsigned)T);
  a.KD = KD; a.padD = KD / 2;
  a.cinChunks = Cin / 32;
  a.Ktot = KD * Cin;
  a.ldV = Cin; a.ldM = Cout;
  a.bankBytes = (unsigned)bankBytes;
  a.vBytes = (unsigned)vBytes; a.uBytes = (unsigned)uBytes; a.mBytes = (unsigned)mBytes;
  a.margin = (unsigned)margin;
  int grid = 2 * fo_cu_count();
  grid = (grid + 7) / 8 * 8;
  const int maxUseful = ((a.tiles + 7) / 8) * 8;
  if (grid > maxUseful) grid = maxUseful;
  static const int diag = getenv("FACEOFF_SPLIT_DIAG") ? atoi(getenv("FACEOFF_SPLIT_DIAG")) : 0;
  if (diag == 1) hipLaunchKernelGGL(wino_gemm_split_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (diag == 2) hipLaunchKernelGGL(wino_gemm_split_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (diag == 4) hipLaunchKernelGGL(wino_gemm_split_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (diag == 5) hipLaunchKernelGGL(wino_gemm_split_kernel<5>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (diag == 3) hipLaunchKernelGGL(wino_gemm_split_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(wino_gemm_split_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
