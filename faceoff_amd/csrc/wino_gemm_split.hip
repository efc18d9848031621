// The Winograd-domain GEMMs of wino_gemm.hip with the fp32 products formed on the bf16 matrix pipe.
//
// gfx950 multiplies bf16 at 16x the rate of fp32 (v_mfma_f32_16x16x32_bf16: 16 K FLOP in 16 cycles; v_mfma_f32_32x32x2_f32:
// 4 K FLOP in 64).  An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand bits, same exponent
// range):  x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (the subtractions are exact).  So
//     a * b = sum_{i,j} a_i b_j,        |a_i b_j| <= 2^(-8(i+j)) |a b|
// and keeping the six terms with i + j <= 2 leaves a relative error of 2 * 2^-24 + 2^-32 per product -- the size of ONE fp32
// rounding -- with the sum accumulated in fp32 by the MFMA exactly as before.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA
// they replace.  This is the arithmetic contract of an fp32 FMA chain to within the last bit (tests/test_split_gpu.py
// measures both kernels against an fp64 product: this one is the closer of the two), not a reduced-precision mode: nothing
// is rounded to bf16 that is not carried by another piece.
//
// Same interface, tile walk and K segments as wino_gemm_kernel (that file has the derivation): persistent workgroups, 128 x 128
// tiles, 32-deep K-steps, depth taps as K segments with per-row validity.  What differs:
//   * the filter banks U are split once per call by a small kernel into three bf16 planes (caller's scratch) and reach LDS by
//     LDS-DMA, double-buffered: no registers, no vector-ALU work in the GEMM;
//   * the planes V stay fp32 in HBM (traffic unchanged; the filter-gradient path reads the same tensor): a K-step's rows are
//     loaded two steps ahead into registers, split there (9 VALU instructions per pair of values: 3 packed converts, 4
//     re-expansions, 2 packed subtractions) and written to LDS as three bf16 planes.  tools/ubench/mfma_valu_overlap.hip: on gfx950 a
//     vector-ALU instruction and a bf16 MFMA of the same SIMD never overlap, not even from different waves, so the split is
//     paid in matrix time (352 cycles against 1536 of MFMA per wave and K-step) -- half of what splitting both operands cost;
//   * LDS rows are 64 bytes ([piece][row][32 k]), chunk c (8 k) of row r at position c ^ swz(r): conflict-free for the
//     ds_read_b128 fragment reads of the 16x16x32 MFMA, the ds_write_b64 of the split and the DMA (swizzle on the source side);
//   * V's LDS planes are single-buffered (72 KB per workgroup in all, two workgroups per CU): the split + LDS writes sit
//     between two barriers while the CU's other workgroup has the matrix pipe;
//   * the filter fragment is the MFMA's row operand, so an accumulator register quartet is 4 consecutive output channels of
//     one row: the epilogue is one 16-byte store per 16x16 block and lane, 64-byte runs per row.
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace {

struct WGArgs {
  const float* V;
  const void* U3;       // [planes][3][Cout][Ktot] bf16
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
  unsigned bankBytes;   // bytes of one plane's three bf16 banks
  unsigned pieceBytes;  // bytes of one bf16 bank
  unsigned vBytes, uBytes, mBytes;
  unsigned margin;
};

constexpr int BM = 128, BN = 128, BK = 32;
constexpr unsigned OOB = 0x80000000u;
constexpr int PIECE = 128 * 64;            // bytes of one bf16 plane of one operand tile
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_byte* dst, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ unsigned udiv(unsigned x, int d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
__device__ __forceinline__ unsigned pk(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// two fp32 -> three packed bf16 pairs (exact: x == p0 + p1 + p2 for each of the two); the subtractions as packed fp32 ops
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned& q0, unsigned& q1, unsigned& q2) {
  f32x2 x = {a, b};
  q0 = pk(x[0], x[1]);
  x = x - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
  q1 = pk(x[0], x[1]);
  x = x - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
  q2 = pk(x[0], x[1]);
}

// U [planes][n] fp32 -> U3 [planes][3][n] bf16 (n % 4 == 0)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ U, unsigned* __restrict__ U3, int n4, int planes) {
  const int total = n4 * planes;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int plane = i / n4, e = i - plane * n4;
    const f32x4 x = reinterpret_cast<const f32x4*>(U)[i];
    unsigned q[6];
    split2(x[0], x[1], q[0], q[2], q[4]);
    split2(x[2], x[3], q[1], q[3], q[5]);
    const u32x2 p0 = {q[0], q[1]}, p1 = {q[2], q[3]}, p2 = {q[4], q[5]};
    u32x2* o = reinterpret_cast<u32x2*>(U3) + (size_t)plane * 3 * n4 + e;
    o[0] = p0; o[n4] = p1; o[2 * n4] = p2;
  }
}

__global__ __launch_bounds__(256, 2) void wino_gemm_split_kernel(const WGArgs a) {
  // V planes, then two stages of U planes.  (Dynamic on purpose: with a static array hipcc orders every LDS read after the
  // LDS-DMAs in flight -- an s_waitcnt vmcnt(0) in front of each K-step's fragment reads, the whole load latency exposed.)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
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
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.U3), 0, a.uBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, a.mBytes, 0x00020000);

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int lo = (int)(((long long)a.tiles * xcd) >> 3), hi = (int)(((long long)a.tiles * (xcd + 1)) >> 3);
  int ld_tile = lo + slot;
  if (ld_tile >= hi) return;

  // ---- loader state (one K-step ahead of the matrix pipe), as in wino_gemm_kernel.  The U planes: 24 wave-DMAs of 16 rows
  // x 64 B per K-step, six per wave -- DMA d = wave * 6 + q is piece d / 8, rows (d % 8) * 16 .. + 15; lane = (row % 16, position)
  unsigned ld_rowoff[4], ld_mask[4], ld_woff[6];
  int ld_kd = 0, ld_seq = 0, ld_left = 0, ld_chunk = 0;   // this depth tap, the taps after it (2 bits each), how many in all
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
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int d = wave * 6 + q, piece = d >> 3, row = (d & 7) * 16 + (lane >> 2);
      ld_woff[q] = boff + (unsigned)piece * a.pieceBytes + (unsigned)((tile_n * BN + row) * a.Ktot) * 2u + (unsigned)(((lane & 3) ^ swz(row)) << 4);
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
    // The taps in the order kd = (2 - frame + i) mod 3: workgroups on neighbouring frames start together, and in this order
    // the three of them that read a frame's rows (as its tap 0, 1, 2) do so in the same third of their tiles -- one fetch
    // from HBM serves all three out of L2 instead of three fetches a third of a tile (~4 MB of traffic per XCD) apart.
    {
      const unsigned F0 = a.pShift >= 0 ? (unsigned)row0 >> a.pShift : udiv((unsigned)row0, a.P, a.pMagic);
      const int c = (int)((5u - (F0 - __umulhi(F0, 0x55555556u) * 3u)) % 3u);
      int seq = 0, n = 0;
      for (int i = 0; i < a.KD; ++i) {
        int kd = c + i;
        if (kd >= a.KD) kd -= a.KD;
        if (a.KD == 1) kd = 0;
        if (kd >= klo && kd < khi) { seq |= kd << (2 * n); ++n; }
      }
      ld_kd = seq & 3; ld_seq = seq >> 2; ld_left = n; ld_chunk = 0;
    }
    if (tid == 0) { fifo[fifo_w & 3][0] = row0; fifo[fifo_w & 3][1] = tile_n; fifo[fifo_w & 3][2] = (khi - klo) * a.cinChunks; }
    ++fifo_w;
  };

  // V rows run two K-steps ahead in registers (split during the MFMAs of the step before they are needed), the U planes one
  // step ahead by DMA: the loader walks at V's pace and hands each step's U offsets to the next call
  f32x4 ra[2][4];
  unsigned pb_woff[6];
  int pb_soff = 0;
#pragma unroll
  for (int q = 0; q < 6; ++q) pb_woff[q] = OOB;
  auto load_step = [&](int set, int stage, bool dma) {
    if (dma) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int d = wave * 6 + q;            // (wave-uniform: the LDS address stays scalar)
        dma16(rwp, (lds_byte*)(Bs0) + stage * (3 * PIECE) + (d >> 3) * PIECE + (d & 7) * (16 * 64), pb_woff[q], pb_soff);
      }
    }
    const int soffA = (ld_kd * a.P * a.ldV + ld_chunk * BK) * 4;
    pb_soff = (ld_kd * a.cinChunks + ld_chunk) * (BK * 2);
#pragma unroll
    for (int q = 0; q < 6; ++q) pb_woff[q] = ld_woff[q];
#pragma unroll
    for (int s = 0; s < 4; ++s) ra[set][s] = bufload(rin, ((ld_mask[s] >> ld_kd) << 31) | ld_rowoff[s], soffA);
    if (++ld_chunk == a.cinChunks) {
      ld_chunk = 0;
      if (--ld_left != 0) { ld_kd = ld_seq & 3; ld_seq >>= 2; }
      else {
        ld_tile += per;
        if (ld_live && ld_tile < hi) setup(ld_tile);
        else {
          ld_live = false;
          ld_kd = 0; ld_seq = 0; ld_left = 1 << 30;
#pragma unroll
          for (int i = 0; i < 4; ++i) { ld_rowoff[i] = OOB; ld_mask[i] = 0; }
#pragma unroll
          for (int q = 0; q < 6; ++q) ld_woff[q] = OOB;
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
  unsigned pc[4][6];                           // the split of one K-step's rows, waiting for the LDS planes to be free
  auto split_rows = [&](int set, int s) {
    split2(ra[set][s][0], ra[set][s][1], pc[s][0], pc[s][2], pc[s][4]);
    split2(ra[set][s][2], ra[set][s][3], pc[s][1], pc[s][3], pc[s][5]);
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      *reinterpret_cast<u32x2*>(As0 + wr_off[s]) = u32x2{pc[s][0], pc[s][1]};
      *reinterpret_cast<u32x2*>(As0 + PIECE + wr_off[s]) = u32x2{pc[s][2], pc[s][3]};
      *reinterpret_cast<u32x2*>(As0 + 2 * PIECE + wr_off[s]) = u32x2{pc[s][4], pc[s][5]};
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
  load_step(0, 0, false);                      // V(0); U(0)'s offsets noted
  load_step(1, 0, true);                       // U(0) -> stage 0; V(1)
  __syncthreads();                             // the first tile's FIFO entry; everything above has landed
  int cur_row0 = fifo[0][0], cur_tn = fifo[0][1], cur_left = fifo[0][2];
  fifo_r = 1;
#pragma unroll
  for (int s = 0; s < 4; ++s) split_rows(0, s);
  store_rows();
  __syncthreads();
  // one K-step n: `set` = the register set V(n+2) is loaded into (V(n+1) sits in the other), `stage` = U(n)'s LDS stage
  auto body = [&](int set, int stage) -> bool {
    load_step(set, stage ^ 1, true);           // U(n+1) by DMA, then V(n+2)
    const unsigned char* Bq = Bf + stage * (3 * PIECE);
    bf16x8 fa[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const bf16x8*>(Af + p * PIECE + i * (16 * 64));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(Bq + p * PIECE + j * (16 * 64));
      // the six terms, smallest first, four independent accumulators between two uses of one; the filter is the row
      // operand (accumulator = 4 consecutive output channels of row l15)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][2], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][0], acc[i][j], 0, 0, 0);
      split_rows(set ^ 1, j);                  // a quarter of V(n+1)'s split (vector ALU: matrix time either way)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // every wave has read this K-step's planes
    if (--cur_left == 0) {
      const unsigned voff = (unsigned)((cur_row0 + wm * 64 + l15) * a.ldM + cur_tn * BN + wn * 64 + quad * 4) * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rout, voff + j * 64, i * 16 * a.ldM * 4, 0);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (fifo_r == fifo_w) return false;      // the loader entered no further tile: that was this workgroup's last
      cur_row0 = fifo[fifo_r & 3][0]; cur_tn = fifo[fifo_r & 3][1]; cur_left = fifo[fifo_r & 3][2];
      ++fifo_r;
      store_rows();                            // V(n+1)'s planes
      // U(n+1)'s DMAs were issued before V(n+2)'s four loads and the tile's 16 stores: all older than those have landed
      asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    } else {
      store_rows();
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    return true;
  };
  while (true) {
    if (!body(0, 0)) break;
    if (!body(1, 1)) break;
  }
}

// ---------------------------------------------------------------------------------------------------- 256 x 128 tiles
// One workgroup of eight waves per CU on a 256 x 128 tile (planes that are whole 256-row tiles: every C2 shape).  Against the
// kernel above: a K-step's U planes serve twice the rows (the L2 -> LDS traffic of the filter banks, the larger of the two
// operand streams at 6 B per element, halves); V's LDS planes are double-buffered too (144 KB in all), so the split's LDS
// writes go into the other stage during the MFMAs and a K-step has ONE barrier.
__global__ __launch_bounds__(512, 1) void wino_gemm_split256_kernel(const WGArgs a) {
  // two stages of V planes (256 rows), then two stages of U planes (128 rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __shared__ int fifo[4][4];                 // tiles the loader has entered and the matrix side has not: row0, tile_n, K-steps
  unsigned char* As0 = lds;
  constexpr int ASTAGE = 6 * PIECE, BSTAGE = 3 * PIECE, APIECE = 2 * PIECE;
  unsigned char* Bs0 = lds + 2 * ASTAGE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;             // 4 x 2 waves of 64 x 64
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;      // rows lrow + 64 s

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.V) - a.margin), 0, a.vBytes + a.margin, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.U3), 0, a.uBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, a.mBytes, 0x00020000);

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int lo = (int)(((long long)a.tiles * xcd) >> 3), hi = (int)(((long long)a.tiles * (xcd + 1)) >> 3);
  int ld_tile = lo + slot;
  if (ld_tile >= hi) return;

  // ---- loader state (one K-step ahead of the matrix pipe), as in wino_gemm_kernel.  The U planes: 24 wave-DMAs of 16 rows
  // x 64 B per K-step, six per wave -- DMA d = wave * 6 + q is piece d / 8, rows (d % 8) * 16 .. + 15; lane = (row % 16, position)
  unsigned ld_rowoff[4], ld_mask[4], ld_woff[3];
  int ld_kd = 0, ld_seq = 0, ld_left = 0, ld_chunk = 0;   // this depth tap, the taps after it (2 bits each), how many in all
  int fifo_w = 0, fifo_r = 0;
  bool ld_live = true;

  auto setup = [&](int tile) {
    const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
    const int row0 = tile_m * 256;
    const int plane = (int)udiv((unsigned)tile_m, a.tilesPerPlane, a.planeMagic);
    const unsigned boff = (unsigned)plane * a.bankBytes;
    unsigned any = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + lrow + 64 * i;
      const unsigned F = a.pShift >= 0 ? (unsigned)r >> a.pShift : udiv((unsigned)r, a.P, a.pMagic);
      const int t = (int)(F - udiv(F, a.T, a.tMagic) * (unsigned)a.T);
      unsigned bad = 0;
      for (int kd = 0; kd < a.KD; ++kd) bad |= ((unsigned)(t + kd - a.padD) < (unsigned)a.T ? 0u : 1u) << kd;
      ld_mask[i] = bad;
      ld_rowoff[i] = (unsigned)(r * a.ldV + lcol) * 4u;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int d = wave * 3 + q, piece = d >> 3, row = (d & 7) * 16 + (lane >> 2);
      ld_woff[q] = boff + (unsigned)piece * a.pieceBytes + (unsigned)((tile_n * BN + row) * a.Ktot) * 2u + (unsigned)(((lane & 3) ^ swz(row)) << 4);
    }
    {
      const unsigned F0 = a.pShift >= 0 ? (unsigned)row0 >> a.pShift : udiv((unsigned)row0, a.P, a.pMagic);
      const unsigned F1 = a.pShift >= 0 ? (unsigned)(row0 + 255) >> a.pShift : udiv((unsigned)(row0 + 255), a.P, a.pMagic);
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
    // The taps in the order kd = (2 - frame + i) mod 3: workgroups on neighbouring frames start together, and in this order
    // the three of them that read a frame's rows (as its tap 0, 1, 2) do so in the same third of their tiles -- one fetch
    // from HBM serves all three out of L2 instead of three fetches a third of a tile (~4 MB of traffic per XCD) apart.
    {
      const unsigned F0 = a.pShift >= 0 ? (unsigned)row0 >> a.pShift : udiv((unsigned)row0, a.P, a.pMagic);
      const int c = (int)((5u - (F0 - __umulhi(F0, 0x55555556u) * 3u)) % 3u);
      int seq = 0, n = 0;
      for (int i = 0; i < a.KD; ++i) {
        int kd = c + i;
        if (kd >= a.KD) kd -= a.KD;
        if (a.KD == 1) kd = 0;
        if (kd >= klo && kd < khi) { seq |= kd << (2 * n); ++n; }
      }
      ld_kd = seq & 3; ld_seq = seq >> 2; ld_left = n; ld_chunk = 0;
    }
    if (tid == 0) { fifo[fifo_w & 3][0] = row0; fifo[fifo_w & 3][1] = tile_n; fifo[fifo_w & 3][2] = (khi - klo) * a.cinChunks; }
    ++fifo_w;
  };

  // V rows run two K-steps ahead in registers (split during the MFMAs of the step before they are needed), the U planes one
  // step ahead by DMA: the loader walks at V's pace and hands each step's U offsets to the next call
  f32x4 ra[2][4];
  unsigned pb_woff[3];
  int pb_soff = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q) pb_woff[q] = OOB;
  auto load_step = [&](int set, int stage, bool dma) {
    if (dma) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int d = wave * 3 + q;            // (wave-uniform: the LDS address stays scalar)
        dma16(rwp, (lds_byte*)(Bs0) + stage * BSTAGE + (d >> 3) * PIECE + (d & 7) * (16 * 64), pb_woff[q], pb_soff);
      }
    }
    const int soffA = (ld_kd * a.P * a.ldV + ld_chunk * BK) * 4;
    pb_soff = (ld_kd * a.cinChunks + ld_chunk) * (BK * 2);
#pragma unroll
    for (int q = 0; q < 3; ++q) pb_woff[q] = ld_woff[q];
#pragma unroll
    for (int s = 0; s < 4; ++s) ra[set][s] = bufload(rin, ((ld_mask[s] >> ld_kd) << 31) | ld_rowoff[s], soffA);
    if (++ld_chunk == a.cinChunks) {
      ld_chunk = 0;
      if (--ld_left != 0) { ld_kd = ld_seq & 3; ld_seq >>= 2; }
      else {
        ld_tile += per;
        if (ld_live && ld_tile < hi) setup(ld_tile);
        else {
          ld_live = false;
          ld_kd = 0; ld_seq = 0; ld_left = 1 << 30;
#pragma unroll
          for (int i = 0; i < 4; ++i) { ld_rowoff[i] = OOB; ld_mask[i] = 0; }
#pragma unroll
          for (int q = 0; q < 3; ++q) ld_woff[q] = OOB;
        }
      }
    }
  };
  // this thread's 8 bytes (4 k) of rows lrow + 32 s in each plane
  unsigned wr_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = lrow + 64 * s;
    wr_off[s] = (unsigned)(row * 64 + ((((tid & 7) >> 1) ^ swz(row)) << 4) + (tid & 1) * 8);
  }
  unsigned pc[4][6];                           // the split of one K-step's rows, waiting for the LDS planes to be free
  auto split_rows = [&](int set, int s) {
    split2(ra[set][s][0], ra[set][s][1], pc[s][0], pc[s][2], pc[s][4]);
    split2(ra[set][s][2], ra[set][s][3], pc[s][1], pc[s][3], pc[s][5]);
  };
  auto store_rows = [&](int s, int astage) {
    unsigned char* A = As0 + astage * ASTAGE + wr_off[s];
    *reinterpret_cast<u32x2*>(A) = u32x2{pc[s][0], pc[s][1]};
    *reinterpret_cast<u32x2*>(A + APIECE) = u32x2{pc[s][2], pc[s][3]};
    *reinterpret_cast<u32x2*>(A + 2 * APIECE) = u32x2{pc[s][4], pc[s][5]};
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
  load_step(0, 0, false);                      // V(0); U(0)'s offsets noted
  load_step(1, 0, true);                       // U(0) -> stage 0; V(1)
  __syncthreads();                             // the first tile's FIFO entry; everything above has landed
  int cur_row0 = fifo[0][0], cur_tn = fifo[0][1], cur_left = fifo[0][2];
  fifo_r = 1;
#pragma unroll
  for (int s = 0; s < 4; ++s) { split_rows(0, s); store_rows(s, 0); }
  __syncthreads();
  // one K-step n: `set` = the register set V(n+2) is loaded into (V(n+1) sits in the other), `stage` = the LDS stage of V(n), U(n)
  auto body = [&](int set, int stage) -> bool {
    load_step(set, stage ^ 1, true);           // U(n+1) by DMA, then V(n+2)
    const unsigned char* Aq = Af + stage * ASTAGE;
    const unsigned char* Bq = Bf + stage * BSTAGE;
    bf16x8 fa[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const bf16x8*>(Aq + p * APIECE + i * (16 * 64));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(Bq + p * PIECE + j * (16 * 64));
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][2], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[i][0], acc[i][j], 0, 0, 0);
      split_rows(set ^ 1, j);                  // a quarter of V(n+1): split, and straight into the other stage
      store_rows(j, stage ^ 1);
    }
    const bool tile_end = --cur_left == 0;
    if (tile_end) {
      const unsigned voff = (unsigned)((cur_row0 + wm * 64 + l15) * a.ldM + cur_tn * BN + wn * 64 + quad * 4) * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rout, voff + j * 64, i * 16 * a.ldM * 4, 0);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (fifo_r == fifo_w) return false;      // the loader entered no further tile: that was this workgroup's last
      cur_row0 = fifo[fifo_r & 3][0]; cur_tn = fifo[fifo_r & 3][1]; cur_left = fifo[fifo_r & 3][2];
      ++fifo_r;
      // U(n+1)'s DMAs were issued before V(n+2)'s four loads and the tile's 16 stores: all older than those have landed
      asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();              // K-step n read by every wave; V(n+1), U(n+1) in place
    return true;
  };
  while (true) {
    if (!body(0, 0)) break;
    if (!body(1, 1)) break;
  }
}

static unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

}  // namespace

extern "C" int64_t fo_wino_gemm_split_ws_bytes(int planes, int Cin, int Cout, int KD) {
  return (int64_t)planes * 3 * Cout * KD * Cin * 2;
}

extern "C" int fo_wino_gemm_split(const float* V, const float* U, void* U3, float* M, int planes, int N, int T, int P, int Cin, int Cout,
                                  int KD, void* stream) {
  FO_REQUIRE(V && U && U3 && M && planes > 0 && N > 0 && T > 0 && N % T == 0 && P > 0, FO_E_SHAPE, "wino_gemm_split: bad sizes");
  FO_REQUIRE(KD == 1 || KD == 3, FO_E_SHAPE, "wino_gemm_split: KD must be 1 or 3 (got %d)", KD);
  FO_REQUIRE(Cin % 32 == 0 && Cin >= 64 && Cout % 128 == 0, FO_E_SHAPE, "wino_gemm_split: Cin %% 32 == 0, Cin >= 64, Cout %% 128 == 0");
  FO_REQUIRE(((long long)N * P) % 128 == 0, FO_E_SHAPE, "wino_gemm_split: a plane (N * P = %lld rows) must be whole 128-row tiles",
             (long long)N * P);
  FO_REQUIRE(fo_aligned16(V) && fo_aligned16(U) && fo_aligned16(U3) && fo_aligned16(M), FO_E_ALIGN, "wino_gemm_split: 16-byte alignment");
  const long long rows = (long long)planes * N * P;
  const unsigned long long vBytes = (unsigned long long)rows * Cin * 4ull, mBytes = (unsigned long long)rows * Cout * 4ull;
  const unsigned long long pieceBytes = (unsigned long long)Cout * KD * Cin * 2ull, bankBytes = 3 * pieceBytes, uBytes = bankBytes * planes;
  const unsigned long long margin = (unsigned long long)(KD / 2) * P * Cin * 4ull;
  FO_REQUIRE(vBytes + 2 * margin < (1ull << 31) && mBytes < (1ull << 31) && uBytes < (1ull << 31), FO_E_SHAPE,
             "wino_gemm_split: plane stack exceeds the 2 GiB buffer-descriptor window");
  {
    const int n4 = Cout * KD * Cin / 4;
    const int blocks = (int)std::min<long long>(((long long)n4 * planes + 255) / 256, 4LL * fo_cu_count());
    hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, reinterpret_cast<unsigned*>(U3), n4, planes);
    FO_CHECK_LAUNCH();
  }
  WGArgs a;
  a.V = V; a.U3 = U3; a.M = M;
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
  a.bankBytes = (unsigned)bankBytes; a.pieceBytes = (unsigned)pieceBytes;
  a.vBytes = (unsigned)vBytes; a.uBytes = (unsigned)uBytes; a.mBytes = (unsigned)mBytes;
  a.margin = (unsigned)margin;
  static const bool no256 = getenv("FACEOFF_SPLIT_NO256") != nullptr;
  if (((long long)N * P) % 256 == 0 && !no256) {           // 256-row tiles, one workgroup of eight waves per CU
    a.tilesPerPlane = (int)(((long long)N * P) / 256);
    a.tiles = (int)(rows / 256) * a.tilesN;
    a.planeMagic = magic_of((unsigned)a.tilesPerPlane);
    int grid = (fo_cu_count() + 7) / 8 * 8;
    const int maxUseful = ((a.tiles + 7) / 8) * 8;
    if (grid > maxUseful) grid = maxUseful;
    constexpr int ldsBytes = 18 * PIECE;
    static fo_lds_once once256;
    if (!fo_lds_optin(once256, reinterpret_cast<const void*>(wino_gemm_split256_kernel), ldsBytes, "wino_gemm_split256")) return FO_E_HIP;
    FO_NOTE("wino_gemm_split256_kernel");
    hipLaunchKernelGGL(wino_gemm_split256_kernel, dim3(grid), dim3(512), ldsBytes, (hipStream_t)stream, a);
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  int grid = 2 * fo_cu_count();
  grid = (grid + 7) / 8 * 8;
  const int maxUseful = ((a.tiles + 7) / 8) * 8;
  if (grid > maxUseful) grid = maxUseful;
  constexpr int ldsBytes = 9 * PIECE;
  static fo_lds_once once;
  if (!fo_lds_optin(once, reinterpret_cast<const void*>(wino_gemm_split_kernel), ldsBytes, "wino_gemm_split")) return FO_E_HIP;
  FO_NOTE("wino_gemm_split_kernel");
  hipLaunchKernelGGL(wino_gemm_split_kernel, dim3(grid), dim3(256), ldsBytes, (hipStream_t)stream, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
