// The GEMMs of the Winograd-domain convolution (winograd.hip), as ONE persistent launch over the stack of transformed
// planes:
//
//   M[xi][r][co] = sum_{kd, ci}  V[xi][r + (kd - KD/2) * P][ci] * U[xi][co][kd][ci]          r = (frame, tile) row,
//                                                                                             P = rows per frame
// i.e. a plain NT GEMM per plane whose K axis has KD segments that read the same rows of the neighbouring frames
// (a segment is skipped where the neighbour lies outside the clip), with the filter bank picked by the plane.
// Replaces fo_conv_igemm_banked for the C2 shapes (reference models/vqvae_conv3d_latent.py:181,185 forward and data
// gradient; :113,140 with KD = 1).
//
// Why a kernel of its own: K is only KD * Cin = 384 (or 128) -- 12 (4) K-steps -- so in the general implicit-GEMM kernel
// the per-tile set-up (pixel decode, tap masks) and the LDS-transposed epilogue were a quarter of a tile's time
// (MFMA-busy 0.71).  Here
//   * workgroups are PERSISTENT: each walks a contiguous run of tiles of its XCD's share, and the K-loop runs straight
//     across tile boundaries -- the first K-step of the next tile is in flight (global loads, then LDS stores) while
//     the last K-step of the current tile is on the matrix pipe, so there is no prologue per tile;
//   * the tile set-up is a dozen scalar instructions + 4 rows of (frame -> clip position -> 3 validity bits);
//   * the epilogue has no LDS round trip: accumulators go straight to memory, one dword per lane, every store
//     instruction covering whole 128-byte lines (an accumulator register is 32 consecutive output channels of one row);
//     M needs no bias / mask / residual -- those live in the output transform;
//   * the K-step body is the one of conv_igemm.hip (K-contiguous operands in 36-float LDS rows, conflict-free b128
//     fragment reads double-buffered in registers, loads and LDS stores interleaved 1:few with the MFMAs, scalar
//     per-step offsets: the fp32 MFMA runs on the SIMD's fp32 ALUs, every VALU instruction in the loop costs matrix rate).
#include "common.h"

namespace {

struct WGArgs {
  const float* V;
  const float* U;
  float* M;
  int tiles;            // tilesM * tilesN
  int tilesN;           // Cout / 128
  int tilesPerPlane;    // (N * P) / 128
  unsigned planeMagic;  // ceil(2^32 / tilesPerPlane)
  int P, pShift;        // rows per frame; log2(P) or -1
  unsigned pMagic;      // ceil(2^32 / P)
  int T;
  unsigned tMagic;      // ceil(2^32 / T)
  int KD, padD;
  int cinChunks;        // Cin / 32
  int Ktot;             // KD * Cin
  int ldV, ldM;         // = Cin, Cout (dense planes)
  unsigned bankBytes;   // bytes per filter bank
  unsigned vBytes, uBytes, mBytes;
  unsigned margin;      // padD * P * ldV * 4: the V descriptor starts this far below V
};

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// x / d for x * d < 2^32 with magic = ceil(2^32 / d) (which does not fit 32 bits for d == 1)
__device__ __forceinline__ unsigned udiv(unsigned x, int d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }

__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const WGArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
  float* As0 = lds;
  float* Bs0 = lds + 2 * BM * LDS_LD;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.V) - a.margin), 0, a.vBytes + a.margin, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U), 0, a.uBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, a.mBytes, 0x00020000);

  // ---- this workgroup's tiles: XCD x owns the contiguous eighth [lo, hi) of the tile list (neighbouring tiles read the
  // same rows through their depth taps and the same filter bank: one L2), its workgroups take them round-robin
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int lo = (int)(((long long)a.tiles * xcd) >> 3), hi = (int)(((long long)a.tiles * (xcd + 1)) >> 3);
  int ld_tile = lo + slot;
  if (ld_tile >= hi) return;

  // ---- loader state (one K-step ahead of the matrix pipe)
  unsigned ld_rowoff[4], ld_mask[4], ld_wrow[4];
  int ld_kd = 0, ld_kd_hi = 0, ld_chunk = 0;
  int nx_row0 = 0, nx_tn = 0, nx_steps = 0;     // the tile the loader is in, for the compute side to adopt
  bool ld_live = true;

  auto setup = [&](int tile) {
    const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
    const int row0 = tile_m * BM;
    const int plane = (int)udiv((unsigned)tile_m, a.tilesPerPlane, a.planeMagic);
    const unsigned boff = (unsigned)plane * a.bankBytes;
    unsigned any = 0;      // bit kd set: some row of the tile has a real frame behind depth tap kd
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
    {  // tile-uniform K range: the frames of the first and the last row (a tile spans at most two frames when P >= 64)
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
    nx_row0 = row0; nx_tn = tile_n; nx_steps = (khi - klo) * a.cinChunks;
  };

  f32x4 ra[4], rb[4];
  auto load_part = [&](int s) {
    const int soffA = (ld_kd * a.P * a.ldV + ld_chunk * BK) * 4;            // scalar
    const int soffB = (ld_kd * a.cinChunks + ld_chunk) * (BK * 4);          // scalar
    ra[s] = bufload(rin, ((ld_mask[s] >> ld_kd) << 31) | ld_rowoff[s], soffA);
    rb[s] = bufload(rwp, ld_wrow[s], soffB);
    if (s == 3) {        // advance the walk: chunk -> depth tap -> next tile of this workgroup
      if (++ld_chunk == a.cinChunks) {
        ld_chunk = 0;
        if (++ld_kd == ld_kd_hi) {
          ld_tile += per;
          if (ld_live && ld_tile < hi) setup(ld_tile);
          else {         // past the last tile: the remaining loads read zeros
            ld_live = false;
            ld_kd = 0; ld_kd_hi = 1 << 30;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ld_rowoff[i] = OOB; ld_wrow[i] = OOB; ld_mask[i] = 0; }
          }
        }
      }
    }
  };
  auto store_part = [&](int s, int buf) {
    *reinterpret_cast<f32x4*>(As0 + buf * BM * LDS_LD + (lrow + 32 * s) * LDS_LD + lcol) = ra[s];
    *reinterpret_cast<f32x4*>(Bs0 + buf * BN * LDS_LD + (lrow + 32 * s) * LDS_LD + lcol) = rb[s];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  setup(ld_tile);
  int cur_row0 = nx_row0, cur_tn = nx_tn, cur_left = nx_steps;
#pragma unroll
  for (int s = 0; s < 4; ++s) load_part(s);
#pragma unroll
  for (int s = 0; s < 4; ++s) store_part(s, 0);
  __syncthreads();
  int cur = 0;
  while (true) {
    const float* As = As0 + cur * BM * LDS_LD + (wm * 64 + l31) * LDS_LD + half * 4;
    const float* Bs = Bs0 + cur * BN * LDS_LD + (wn * 64 + l31) * LDS_LD + half * 4;
    f32x4 fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0 && kk < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD + (kk + 1) * 8);
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD + (kk + 1) * 8);
        }
        if (kk == 0) load_part(s);
        if (kk == 3) store_part(s, cur ^ 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][s], fb[kk & 1][j][s], acc[i][j], 0, 0, 0);
        // masks: 0x8 MFMA, 0x2 VALU, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          if (s == 0 && kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (kk == 0) {
            __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            if (q == 3) __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
          }
          if (kk == 3) {
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    cur ^= 1;
    if (--cur_left == 0) {
      // ---- tile done: accumulators straight to M (register r of a 32x32 accumulator = row (r&3) + 8(r>>2) + 4*half,
      // the 32 lanes of a half = 32 consecutive output channels = one 128-byte line), then clear them
      const unsigned voff = (unsigned)((cur_row0 + wm * 64 + 4 * half) * a.ldM + cur_tn * BN + wn * 64 + l31) * 4u;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int soff = (i * 32 + (r & 3) + 8 * (r >> 2)) * a.ldM * 4;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float v = acc[i][j][r];     // (a bit_cast applied to the vector element itself reads element 0)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, voff + j * 128, soff, 0);
          }
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (!ld_live && nx_row0 == cur_row0 && nx_tn == cur_tn) break;   // that was the last tile of this workgroup
      cur_row0 = nx_row0; cur_tn = nx_tn; cur_left = nx_steps;
    }
  }
}

static unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

}  // namespace

extern "C" int fo_wino_gemm(const float* V, const float* U, float* M, int planes, int N, int T, int P, int Cin, int Cout, int KD,
                            void* stream) {
  FO_REQUIRE(V && U && M && planes > 0 && N > 0 && T > 0 && N % T == 0 && P > 0, FO_E_SHAPE, "wino_gemm: bad sizes");
  FO_REQUIRE(KD == 1 || KD == 3, FO_E_SHAPE, "wino_gemm: KD must be 1 or 3 (got %d)", KD);
  FO_REQUIRE(Cin % 32 == 0 && Cin >= 64 && Cout % 128 == 0, FO_E_SHAPE, "wino_gemm: Cin %% 32 == 0, Cin >= 64, Cout %% 128 == 0");
  FO_REQUIRE(((long long)N * P) % 128 == 0, FO_E_SHAPE, "wino_gemm: a plane (N * P = %lld rows) must be whole 128-row tiles", (long long)N * P);
  FO_REQUIRE(fo_aligned16(V) && fo_aligned16(U) && fo_aligned16(M), FO_E_ALIGN, "wino_gemm: 16-byte alignment");
  const long long rows = (long long)planes * N * P;
  const unsigned long long vBytes = (unsigned long long)rows * Cin * 4ull, mBytes = (unsigned long long)rows * Cout * 4ull;
  const unsigned long long bankBytes = (unsigned long long)Cout * KD * Cin * 4ull, uBytes = bankBytes * planes;
  const unsigned long long margin = (unsigned long long)(KD / 2) * P * Cin * 4ull;
  FO_REQUIRE(vBytes + 2 * margin < (1ull << 31) && mBytes < (1ull << 31) && uBytes < (1ull << 31), FO_E_SHAPE,
             "wino_gemm: plane stack exceeds the 2 GiB buffer-descriptor window");
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
  // two workgroups per CU (73 KB of LDS each), a multiple of 8 so that every XCD gets the same number
  int grid = 2 * fo_cu_count();
  grid = (grid + 7) / 8 * 8;
  const int maxUseful = ((a.tiles + 7) / 8) * 8;
  if (grid > maxUseful) grid = maxUseful;
  FO_NOTE("wino_gemm_kernel");
  hipLaunchKernelGGL(wino_gemm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
