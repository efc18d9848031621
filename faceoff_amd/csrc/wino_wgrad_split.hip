// The Winograd-domain FILTER-GRADIENT GEMMs with the fp32 products formed on the bf16 matrix pipe (see wino_gemm_split.hip for
// the arithmetic: exact three-way bf16 split of both operands, six partial products, fp32 accumulation):
//
//   dU[xi][co][ci][kd] = sum_r  dM[xi][r][co] * V[xi][r + (kd - KD/2) * P][ci]          r over the N * P rows of a plane,
//                                                                                        a term skipped where the shifted frame
// lies outside its clip of T frames.  Same result as fo_conv_wgrad_banked on the (KD,1,1) geometry of ops._wgrad_winograd_dU
// (reference models/vqvae_conv3d_latent.py:181,185 and :110,113,140,157,160 under loss.backward()).
//
// This is a TN GEMM: the contraction index is the ROW of both channels-last operands, while a 16x16x32 bf16 MFMA wants eight
// consecutive k per lane.  gfx950's transposing LDS read supplies exactly that: ds_read_b64_tr_b16 hands lane i of a block of 16
// lanes element (i & 3) of the 8-byte chunk lane 4e + (i >> 2) addressed, for e = 0..3 (tools/ubench/ds_read_tr_probe.hip) -- so
// with the LDS tile stored as it arrives, [k][channel], lane j pointing at row k0 + (j >> 2), channels 4 (j & 3) .. + 3 gives lane i
// the four values X[k0 .. k0+3][m0 + i]; two such reads are one MFMA fragment.  No transposition in registers, no 2-byte stores.
//
//   * a workgroup (4 waves, 2 x 2 of 64 x 64) owns one 128 (co) x 128 (ci) block of one plane and depth tap over a SLAB of the
//     rows (split-K; the slabs are summed, in a fixed order, by wgrad_split_reduce_kernel, which also writes the [co][ci][kd]
//     layout fo_wino_wgrad_out reads);
//   * a K-step is 32 rows: both operands' rows are loaded as fp32 one step ahead into registers, split there (the vector ALU and
//     the bf16 MFMA never overlap on gfx950, so this costs matrix time: ~700 cycles against 1 536 of MFMA per wave and step) and
//     written as three bf16 planes each with 320-byte rows (256 + 64 of padding: conflict-free for the ds_write_b64 and for the
//     transposing reads); single-buffered, 60 KB, two workgroups per CU;
//   * frames whose shifted neighbour is outside the clip are skipped whole (a K-step never straddles a frame: P % 32 == 0).
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

struct WSArgs {
  const float* dM;
  const float* V;
  float* ws;            // [slabs][planes][KD][Cout][Cin]
  int planes, KD, tilesA, tilesB, slabs;
  int N, T, P;          // frames per plane, clip length, rows per frame
  int stepsPerFrame;    // P / 32
  int Cin, Cout;
  long long planeRows;  // N * P
};

constexpr int ROWB = 320;                  // bytes per LDS row: 128 bf16 + 64 B of padding
constexpr int PLANE = 32 * ROWB;           // one bf16 piece of one operand's K-step
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned& q0, unsigned& q1, unsigned& q2) {
  f32x2 x = {a, b};
  q0 = pk(x[0], x[1]);
  x = x - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
  q1 = pk(x[0], x[1]);
  x = x - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
  q2 = pk(x[0], x[1]);
}
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* p) {       // rows k0..k0+3 at p, rows k0+4..k0+7 four rows further
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * ROWB));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256, 2) void wino_wgrad_split_kernel(const WSArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * PLANE];     // dM pieces 0..2, V pieces 0..2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- this workgroup's block
  int w = blockIdx.x;
  const int slab = w % a.slabs; w /= a.slabs;
  const int tb = w % a.tilesB; w /= a.tilesB;
  const int ta = w % a.tilesA; w /= a.tilesA;
  const int kd = w % a.KD;
  const int plane = w / a.KD;
  const int shift = kd - a.KD / 2;                       // frames
  const long long stepsTotal = a.planeRows / 32;
  const int step0 = (int)(stepsTotal * slab / a.slabs), step1 = (int)(stepsTotal * (slab + 1) / a.slabs);

  // ---- loader: thread = (row tid >> 3 of the K-step, 4 channels (tid & 7) * 4 + 32 s)
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const float* pA = a.dM + ((long long)plane * a.planeRows + lrow) * a.Cout + ta * 128 + lcol;
  const float* pB = a.V + ((long long)plane * a.planeRows + lrow + (long long)shift * a.P) * a.Cin + tb * 128 + lcol;
  f32x4 ra[4], rb[4];
  int ld_step = step0;
  auto next_valid = [&](int s) {                         // first step >= s whose frame has the shifted neighbour inside its clip
    while (s < step1) {
      const int f = s / a.stepsPerFrame;
      const int t = f % a.T;
      if ((unsigned)(t + shift) < (unsigned)a.T) break;
      s = (f + 1) * a.stepsPerFrame;
    }
    return s;
  };
  auto load = [&]() {
    const long long r = (long long)ld_step * 32;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra[s] = *reinterpret_cast<const f32x4*>(pA + r * a.Cout + 32 * s);
      rb[s] = *reinterpret_cast<const f32x4*>(pB + r * a.Cin + 32 * s);
    }
  };
  // (32-byte granules of rows 8..15 and 24..31 swapped in pairs: the four 16-lane blocks of a transposing read -- rows 8 kg .. --
  // then sit on complementary halves of the banks, two by two)
  const unsigned wr = (unsigned)(lrow * ROWB + ((lcol * 2) ^ (((lrow >> 3) & 1) << 5)));
  auto split_store = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      unsigned q[6];
      split2(ra[s][0], ra[s][1], q[0], q[2], q[4]);
      split2(ra[s][2], ra[s][3], q[1], q[3], q[5]);
      unsigned char* d = lds + wr + s * 64;
      *reinterpret_cast<u32x2*>(d) = u32x2{q[0], q[1]};
      *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{q[2], q[3]};
      *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{q[4], q[5]};
      split2(rb[s][0], rb[s][1], q[0], q[2], q[4]);
      split2(rb[s][2], rb[s][3], q[1], q[3], q[5]);
      d += 3 * PLANE;
      *reinterpret_cast<u32x2*>(d) = u32x2{q[0], q[1]};
      *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{q[2], q[3]};
      *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{q[4], q[5]};
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment base: lane j of a 16-lane block -> row (8 kgroup + (j >> 2)), channels 4 (j & 3)
  const int l15 = lane & 15, kg = lane >> 4;
  const unsigned fr = (unsigned)((kg * 8 + (l15 >> 2)) * ROWB + (l15 & 3) * 8);
  const int sw = (kg & 1) << 5;                          // the granule swap of this lane's rows
  const unsigned char* Af = lds + fr + (wm * 64) * 2;
  const unsigned char* Bf = lds + 3 * PLANE + fr + (wn * 64) * 2;

  ld_step = next_valid(ld_step);
  if (ld_step < step1) {
    load();
    while (true) {
      split_store();
      __syncthreads();
      ld_step = next_valid(ld_step + 1);
      const bool more = ld_step < step1;
      if (more) load();                                  // K-step n+1: in flight during the MFMAs
      bf16x8 fa[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[i][p] = frag_tr(Af + p * PLANE + ((i * 32) ^ sw));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 fb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[p] = frag_tr(Bf + p * PLANE + ((j * 32) ^ sw));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fb[0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[2], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[0], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      if (!more) break;
    }
  }
  // ---- partial block -> ws[slab][plane][kd][co][ci]: accumulator register r of lane l = (co = 4 (l >> 4) + r, ci = l & 15)
  float* o = a.ws + ((((long long)slab * a.planes + plane) * a.KD + kd) * a.Cout + ta * 128 + wm * 64) * a.Cin + tb * 128 + wn * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(long long)(i * 16 + kg * 4 + r) * a.Cin + j * 16 + l15] = acc[i][j][r];
}

// ---------------------------------------------------------------------------------------------------- KD = 3: one pass for the three taps
// The kernel above run once per depth tap reads every dM row three times and every V row three times (3.4 GB per 64^2 launch at
// 3.1 TB/s: it is HBM-bound).  Here a workgroup (8 waves: 2 x 4 of 64 x 32) owns the 128 x 128 blocks of ALL THREE taps (96
// accumulator registers per lane) and walks a clip frame by frame at a fixed 32-row position: step t loads ONE dM tile (frame t)
// and ONE V tile (frame t + 1), keeps the V tiles of frames t-1, t, t+1 in a ring of three LDS slots, and does the (up to) three
// products dM[t]^T V[t + kd - 1] -- 10 tile loads per clip position for 13 products instead of 26, three times the MFMAs per split
// value.  The V tiles form one linear stream over (clip position, frame): tile m lives in slot m % 3, step n reads tiles n-1, n, n+1
// and prefetches tile n+2 (stored after the step's barrier, into the slot tile n-1 leaves).  120 KB of LDS, one workgroup per CU.
struct WS3Args {
  const float* dM;
  const float* V;
  float* ws;            // [slabs][planes][3][Cout][Cin]
  int planes, tilesA, tilesB, slabs;
  int T, P, chunks;     // clip length, rows per frame, P / 32
  int units;            // clips * chunks: (clip, 32-row position) pairs per plane
  int Cin, Cout;
  long long planeRows;
};

__global__ __launch_bounds__(512, 1) void wino_wgrad_split3_kernel(const WS3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[12 * PLANE];    // dM pieces 0..2, then three slots of V pieces 0..2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave >> 2, wb = wave & 3;
  int w = blockIdx.x;
  const int slab = w % a.slabs; w /= a.slabs;
  const int tb = w % a.tilesB; w /= a.tilesB;
  const int ta = w % a.tilesA;
  const int plane = w / a.tilesA;
  const int u0 = (int)((long long)a.units * slab / a.slabs), u1 = (int)((long long)a.units * (slab + 1) / a.slabs);
  const int nsteps = (u1 - u0) * a.T;

  // ---- loader: thread = (row tid >> 4, 4 channels (tid & 15) * 4 + 64 s); linear tile m -> unit u0 + m / T, frame m % T
  const int lrow = tid >> 4, lcol = (tid & 15) * 4;
  const float* pA = a.dM + ((long long)plane * a.planeRows + lrow) * a.Cout + ta * 128 + lcol;
  const float* pB = a.V + ((long long)plane * a.planeRows + lrow) * a.Cin + tb * 128 + lcol;
  // two cursors over the linear tile stream (unit, frame) -> first row of the tile; advanced one tile at a time (no divisions)
  struct Cursor { int t, chunk, clip; long long row; };
  auto cursor_at = [&](int m) {
    Cursor c;
    const int u = u0 + m / a.T;
    c.t = m - (m / a.T) * a.T; c.clip = u / a.chunks; c.chunk = u - c.clip * a.chunks;
    c.row = (long long)(c.clip * a.T + c.t) * a.P + c.chunk * 32;
    return c;
  };
  auto advance = [&](Cursor& c) {
    c.row += a.P;
    if (++c.t == a.T) {
      c.t = 0;
      if (++c.chunk == a.chunks) { c.chunk = 0; ++c.clip; }
      c.row = (long long)c.clip * a.T * a.P + c.chunk * 32;
    }
  };
  f32x4 ra[2], rb[2];
  auto load_a = [&](const Cursor& c) {
#pragma unroll
    for (int s = 0; s < 2; ++s) ra[s] = *reinterpret_cast<const f32x4*>(pA + c.row * a.Cout + 64 * s);
  };
  auto load_b = [&](const Cursor& c) {
#pragma unroll
    for (int s = 0; s < 2; ++s) rb[s] = *reinterpret_cast<const f32x4*>(pB + c.row * a.Cin + 64 * s);
  };
  const unsigned wr = (unsigned)(lrow * ROWB + ((lcol * 2) ^ (((lrow >> 3) & 1) << 5)));
  auto store_tile = [&](const f32x4* r, unsigned char* base) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned q[6];
      split2(r[s][0], r[s][1], q[0], q[2], q[4]);
      split2(r[s][2], r[s][3], q[1], q[3], q[5]);
      unsigned char* d = base + wr + s * 128;
      *reinterpret_cast<u32x2*>(d) = u32x2{q[0], q[1]};
      *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{q[2], q[3]};
      *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{q[4], q[5]};
    }
  };
  unsigned char* const Bs = lds + 3 * PLANE;

  f32x4 acc[3][4][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[k][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, kg = lane >> 4;
  const unsigned fr = (unsigned)((kg * 8 + (l15 >> 2)) * ROWB + (l15 & 3) * 8);
  const int sw = (kg & 1) << 5;
  const unsigned char* Af = lds + fr + (wa * 64) * 2;
  const unsigned char* Bf = Bs + fr + (wb * 32) * 2;

  if (nsteps > 0) {
    // prologue: dM tile 0, V tiles 0 and 1
    Cursor ca = cursor_at(0), cb = cursor_at(0);
    load_a(ca); load_b(cb);
    store_tile(ra, lds); store_tile(rb, Bs);
    advance(ca); advance(cb);                              // ca: dM tile 1, cb: V tile 1
    if (nsteps > 1) { load_b(cb); store_tile(rb, Bs + 3 * PLANE); }
    advance(cb);                                           // cb: V tile 2
    __syncthreads();
    int t = 0, slot = 0;                                   // frame of step n within its clip; n % 3
    for (int n = 0; n < nsteps; ++n) {
      if (n + 1 < nsteps) load_a(ca);                      // dM tile n+1, V tile n+2: in flight during the MFMAs
      if (n + 2 < nsteps) load_b(cb);
      advance(ca); advance(cb);
      bf16x8 fa[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[i][p] = frag_tr(Af + p * PLANE + ((i * 32) ^ sw));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if ((unsigned)(t + k - 1) >= (unsigned)a.T) continue;              // the neighbour frame lies outside the clip
        int sl = slot + k - 1;                                             // slot of V tile n + k - 1
        sl = sl < 0 ? 2 : (sl > 2 ? 0 : sl);
        const unsigned char* Bq = Bf + sl * (3 * PLANE);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 fb[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) fb[p] = frag_tr(Bq + p * PLANE + ((j * 32) ^ sw));
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fb[0], acc[k][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[2], acc[k][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[1], acc[k][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[0], acc[k][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[1], acc[k][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[0], acc[k][i][j], 0, 0, 0);
        }
      }
      __syncthreads();                                     // every wave has read dM tile n and V tile n-1
      if (n + 1 < nsteps) store_tile(ra, lds);             // dM tile n+1
      if (n + 2 < nsteps) store_tile(rb, Bs + (slot == 0 ? 2 : slot - 1) * (3 * PLANE));     // V tile n+2 -> slot (n+2) % 3 = (n-1) % 3
      __syncthreads();
      if (++t == a.T) t = 0;
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  // ---- partial blocks -> ws[slab][plane][kd][co][ci]
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float* o = a.ws + ((((long long)slab * a.planes + plane) * 3 + k) * a.Cout + ta * 128 + wa * 64) * a.Cin + tb * 128 + wb * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(long long)(i * 16 + kg * 4 + r) * a.Cin + j * 16 + l15] = acc[k][i][j][r];
  }
}

// dU[plane][co][ci][kd] = sum over the slabs (fixed order) of ws[slab][plane][kd][co][ci]
__global__ __launch_bounds__(256) void wgrad_split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dU, int slabs, int planes, int KD,
                                                                 long long cc /* Cout * Cin */) {
  const long long total = (long long)planes * KD * cc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long e = i % cc;
    const int kd = (int)((i / cc) % KD);
    const long long plane = i / (cc * KD);
    float s = 0.f;
    for (int b = 0; b < slabs; ++b) s += ws[(long long)b * total + i];
    dU[(plane * cc + e) * KD + kd] = s;
  }
}

int pick_slabs(int blocks, long long steps) {
  int s = 2 * fo_cu_count() / blocks;                      // (one round of workgroups, two per CU)
  const long long cap = std::max<long long>(1, steps / 16);      // at least 16 K-steps per slab
  return (int)std::max<long long>(1, std::min<long long>(s, cap));
}

}  // namespace

// the one-pass KD = 3 kernel: one workgroup per CU; a slab is a run of (clip, 32-row position) units
static int pick_slabs3(int blocks, int units) {
  const int s = fo_cu_count() / blocks;                    // one round of workgroups: a second, nearly empty round would double the time
  return std::max(1, std::min(s, units));
}
static bool use_pass3(int N, int T, int P, int KD) {
  static const bool off = getenv("FACEOFF_WGRAD_SPLIT_PER_TAP") != nullptr;
  return KD == 3 && P % 32 == 0 && N % T == 0 && !off;
}

extern "C" int64_t fo_wino_wgrad_split_ws_bytes(int planes, int N, int P, int Cin, int Cout, int KD) {
  if (planes <= 0 || Cin % 128 || Cout % 128 || ((long long)N * P) % 32) return -1;
  const int blocks = planes * KD * (Cout / 128) * (Cin / 128);
  // (either kernel's slab count: the larger)
  const int s1 = pick_slabs(blocks, (long long)N * P / 32);
  const int s3 = KD == 3 && P % 32 == 0 ? pick_slabs3(blocks / KD, (int)((long long)N * P / 32)) : 0;
  return (int64_t)std::max(s1, s3) * planes * KD * Cout * Cin * 4;
}

extern "C" int fo_wino_wgrad_split(const float* dM, const float* V, float* dU, float* ws, int64_t ws_bytes, int planes, int N, int T, int P,
                                   int Cin, int Cout, int KD, void* stream) {
  FO_REQUIRE(dM && V && dU && ws && planes > 0 && N > 0 && T > 0 && N % T == 0 && P > 0, FO_E_SHAPE, "wino_wgrad_split: bad sizes");
  FO_REQUIRE(KD == 1 || KD == 3, FO_E_SHAPE, "wino_wgrad_split: KD must be 1 or 3 (got %d)", KD);
  FO_REQUIRE(Cin % 128 == 0 && Cout % 128 == 0, FO_E_SHAPE, "wino_wgrad_split: Cin and Cout must be multiples of 128");
  FO_REQUIRE(((long long)N * P) % 32 == 0 && (KD == 1 || P % 32 == 0), FO_E_SHAPE,
             "wino_wgrad_split: a plane must be whole 32-row K-steps (and a frame too when KD = 3)");
  FO_REQUIRE(fo_aligned16(dM) && fo_aligned16(V) && fo_aligned16(dU) && fo_aligned16(ws), FO_E_ALIGN, "wino_wgrad_split: 16-byte alignment");
  const int64_t need = fo_wino_wgrad_split_ws_bytes(planes, N, P, Cin, Cout, KD);
  FO_REQUIRE(ws_bytes >= need, FO_E_SHAPE, "wino_wgrad_split: workspace of %lld bytes, %lld needed", (long long)ws_bytes, (long long)need);
  if (use_pass3(N, T, P, KD)) {
    WS3Args b;
    b.dM = dM; b.V = V; b.ws = ws;
    b.planes = planes; b.tilesA = Cout / 128; b.tilesB = Cin / 128;
    b.T = T; b.P = P; b.chunks = P / 32; b.units = (N / T) * b.chunks;
    b.Cin = Cin; b.Cout = Cout; b.planeRows = (long long)N * P;
    const int blocks3 = planes * b.tilesA * b.tilesB;
    b.slabs = pick_slabs3(blocks3, b.units);
    FO_NOTE("wino_wgrad_split3_kernel");
    hipLaunchKernelGGL(wino_wgrad_split3_kernel, dim3(blocks3 * b.slabs), dim3(512), 0, (hipStream_t)stream, b);
    FO_CHECK_LAUNCH();
    const long long total3 = (long long)planes * KD * Cout * Cin;
    const int rb3 = (int)std::min<long long>((total3 + 255) / 256, 8LL * fo_cu_count());
    hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3(rb3), dim3(256), 0, (hipStream_t)stream, ws, dU, b.slabs, planes, KD, (long long)Cout * Cin);
    FO_CHECK_LAUNCH();
    return FO_OK;
  }
  WSArgs a;
  a.dM = dM; a.V = V; a.ws = ws;
  a.planes = planes; a.KD = KD; a.tilesA = Cout / 128; a.tilesB = Cin / 128;
  a.N = N; a.T = T; a.P = P;
  a.planeRows = (long long)N * P;
  // KD = 1: no frame is ever skipped; treat the plane as one frame so that P need not be a multiple of 32
  a.stepsPerFrame = KD == 1 ? (int)(a.planeRows / 32) : P / 32;
  if (KD == 1) { a.T = 1; }
  a.Cin = Cin; a.Cout = Cout;
  const int blocks = planes * KD * a.tilesA * a.tilesB;
  a.slabs = pick_slabs(blocks, a.planeRows / 32);
  FO_NOTE("wino_wgrad_split_kernel");
  hipLaunchKernelGGL(wino_wgrad_split_kernel, dim3(blocks * a.slabs), dim3(256), 0, (hipStream_t)stream, a);
  FO_CHECK_LAUNCH();
  const long long total = (long long)planes * KD * Cout * Cin;
  const int rblocks = (int)std::min<long long>((total + 255) / 256, 8LL * fo_cu_count());
  hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, ws, dU, a.slabs, planes, KD, (long long)Cout * Cin);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
