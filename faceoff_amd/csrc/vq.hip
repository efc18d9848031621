// Vector quantiser kernels (reference Quantize.forward, models/vqvae_conv3d_latent.py:47-83).
//
// fo_vq_assign fuses what the reference does with two sgemms, two materialised [Nvec,512]
// matrices (dist :49-53, one_hot :55) and ~10 elementwise kernels:
//   distance (expanded form, exact-fp32 MFMA)  ->  running first-index arg-min  ->  codebook
//   gather + straight-through value (:57,78)   ->  commitment sum (:77);  fo_vq_stats produces the
//   EMA statistics (:60-61) from LDS-privatised tables -- `dist` and the one-hot never exist in memory.
//
// Numerics contract (restated bit-for-bit by oracle/vq_oracle.c):
//   dot(x,e)  = fp32 fma chain over k = 0..63 in order, starting from 0 (what the MFMA computes)
//   ||v||^2   = chain over even k  +  chain over odd k
//   dist      = (||x||^2 - 2*dot) + ||e||^2,   index = first minimum (torch.max tie-break on -dist)
#include <algorithm>
#include "common.h"

namespace {

constexpr int VQ_D = 64, VQ_K = 512, VQ_LD = 65;

__global__ void vq_prepare_kernel(const float* __restrict__ embed, float* __restrict__ embedT, float* __restrict__ enorm) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= VQ_K) return;
  float ev = 0.f, od = 0.f;
  for (int k = 0; k < VQ_D; k += 2) {
    const float a = embed[(size_t)k * VQ_K + c], b = embed[(size_t)(k + 1) * VQ_K + c];
    embedT[(size_t)c * VQ_D + k] = a;
    embedT[(size_t)c * VQ_D + k + 1] = b;
    ev = fmaf(a, a, ev);
    od = fmaf(b, b, od);
  }
  enorm[c] = ev + od;
}

// Persistent: one workgroup (8 waves, two per SIMD) per CU keeps the whole codebook (512 x 64, rows padded to 65 floats:
// conflict-free both for the MFMA A-fragment column reads and the row gathers) in LDS and its
// waves walk 32-vector tiles independently.  MFMA roles: A = codes (rows), B = vectors
// (columns), so a lane owns ONE vector and sees 16 codes per 32-code tile in its accumulator:
// the arg-min is a per-lane scan plus one cross-half exchange.
__global__ __launch_bounds__(512, 1) void vq_assign_kernel(const float* __restrict__ x, int ldx, long long nvec,
                                                           const float* __restrict__ embedT,
                                                           const float* __restrict__ enorm, long long* __restrict__ ind,
                                                           float* __restrict__ qout, int ldq, float* __restrict__ sq_part,
                                                           __bf16* __restrict__ qb16, int ldqb, const long long* __restrict__ forced) {
  __shared__ float E[VQ_K * VQ_LD + VQ_K];
  __shared__ float wave_sq[8];
  float* En = E + VQ_K * VQ_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  for (int idx = tid; idx < VQ_K * VQ_D; idx += 512) E[(idx >> 6) * VQ_LD + (idx & 63)] = embedT[idx];
  for (int idx = tid; idx < VQ_K; idx += 512) En[idx] = enorm[idx];
  __syncthreads();

  const long long ntiles = (nvec + 31) / 32;
  float sq = 0.f;
  for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long long)gridDim.x * 8) {
    const long long v = tile * 32 + l31;
    const bool valid = v < nvec;
    float xr[32];
    const float* xp = x + (size_t)(valid ? v : 0) * ldx + half;
#pragma unroll
    for (int s = 0; s < 32; ++s) xr[s] = valid ? xp[2 * s] : 0.f;
    float xg[32];                                          // the same 32 rows, lane = dimension (for the gather at the end)
#pragma unroll
    for (int u = 0; u < 32; ++u) xg[u] = x[(size_t)min<long long>(tile * 32 + u, nvec - 1) * ldx + lane];
    float xh = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) xh = fmaf(xr[s], xr[s], xh);
    const float xx = xh + lane_xor<32>(xh);

    float best_d = INFINITY;
    int best_i = 0;
    if (forced) {       // teacher-forced codes (wave-uniform branch): gather, straight-through and commitment sum on the GIVEN indices
      best_i = valid ? (int)forced[v] & (VQ_K - 1) : 0;      // (memory-safe whatever the caller hands in; the binding validates under FACEOFF_DEBUG)
    } else
    for (int ct = 0; ct < VQ_K / 32; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* Erow = E + (ct * 32 + l31) * VQ_LD + half;
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Erow[2 * s], xr[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int code = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float d = (xx - 2.f * acc[r]) + En[code];
        if (d < best_d) { best_d = d; best_i = code; }
      }
    }
    if (!forced) {
      const float od = lane_xor<32>(best_d);
      const int oi = lane_xor<32>(best_i);
      if (od < best_d || (od == best_d && oi < best_i)) { best_d = od; best_i = oi; }
    }
    if (half == 0 && valid) ind[v] = best_i;

    // gather + straight-through: the wave walks its 32 vectors, 64 lanes = 64 dims.  The x rows (this layout: lane = dimension)
    // were requested before the distance loop (xg), so nothing waits on memory here
    const int nhere = (int)min<long long>(32, nvec - tile * 32);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (u < nhere) {
        const int idx = __builtin_amdgcn_readlane(best_i, u);     // (wave-uniform: v_readlane, not the ds_bpermute __shfl compiles to; the codebook row's LDS address is then scalar + lane)
        const size_t vj = (size_t)(tile * 32 + u);
        const float diff = E[idx * VQ_LD + lane] - xg[u];
        const float ste = xg[u] + diff;        // input + (quantize - input).detach()   (:78)
        qout[vj * ldq + lane] = ste;
        if (qb16) qb16[vj * ldqb + lane] = (__bf16)ste;     // the bf16-operand engine's copy for the next conv (rounded once)
        sq = fmaf(diff, diff, sq);
      }
    }
  }
  // commitment sum without atomics: wave reduce, the eight waves' sums added in wave order, one partial per workgroup; fo_ordered_sum
  // adds the partials in workgroup order -- the printed latent loss is the same bits run after run
  sq = group_sum_valu<64>(sq);
  if (lane == 0) wave_sq[wave] = sq;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += wave_sq[w];
    sq_part[blockIdx.x] = t;
  }
}


// EMA statistics (:60-61): counts[c] = #vectors assigned to c, esum[c][:] = sum of those vectors.
// The reference builds a [Nvec,512] one-hot and runs a second sgemm; here each workgroup keeps a private [512][64] table in LDS,
// writes it as a slab, and a second kernel sums the slabs in a fixed order.  No atomics anywhere and a summation order that depends
// on the data only -- the codebook update is bit-reproducible run to run:
//   * the 16 waves stage 64 vectors at a time in LDS (wave w loads rows 4w .. 4w+3; the next two blocks' rows are in flight meanwhile);
//   * wave w owns the codes c with c % 16 == w: one ballot over the block's 64 indices finds its vectors, a second one per distinct
//     code gathers that code's vectors, which are summed in vector order in registers (64 dims on the 64 lanes) and added to the
//     table row once.  A degenerate codebook (every vector on one code) costs 64 LDS reads per block on one wave, not 64 serialised
//     read-modify-writes.
__global__ __launch_bounds__(1024, 1) void vq_stats_kernel(const float* __restrict__ x, int ldx, long long nvec,
                                                           const long long* __restrict__ ind, float* __restrict__ ws) {
  __shared__ float tab[VQ_K * VQ_D + VQ_K];
  __shared__ float stage[64 * VQ_D];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < VQ_K * VQ_D + VQ_K; i += 1024) tab[i] = 0.f;
  const long long per = (nvec + gridDim.x - 1) / gridDim.x;
  const long long v0 = blockIdx.x * per, v1 = min(nvec, v0 + per);
  // rows of the next TWO blocks wait in registers (ra: next block, rb: the one after) while the staged block is consumed
  float ra[4], rb[4];
  auto load_rows = [&](long long base, float* r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long vv = base + wave * 4 + u;
      r[u] = vv < v1 ? x[(size_t)vv * ldx + lane] : 0.f;
    }
  };
  auto block = [&](long long base, float* r, int id) {     // r: this block's rows; refilled with the rows of block base + 128
    __syncthreads();                 // the previous block has been consumed (first pass: the table is zero)
#pragma unroll
    for (int u = 0; u < 4; ++u) stage[(wave * 4 + u) * VQ_D + lane] = r[u];
    __syncthreads();
    if (base + 128 < v1) load_rows(base + 128, r);
    unsigned long long m = __ballot(id >= 0 && (id & 15) == wave);
    while (m) {
      const int c = __builtin_amdgcn_readlane(id, (int)__ffsll((long long)m) - 1);
      unsigned long long mc = __ballot(id == c);
      m &= ~mc;
      const float cnt = (float)__popcll(mc);
      float acc = 0.f;
      while (mc) {
        int b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b[u] = mc ? (int)__ffsll((long long)mc) - 1 : -1;
          if (mc) mc &= mc - 1;
        }
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = b[u] >= 0 ? stage[b[u] * VQ_D + lane] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (b[u] >= 0) acc += t[u];
      }
      tab[c * VQ_D + lane] += acc;
      if (lane == 0) tab[VQ_K * VQ_D + c] += cnt;
    }
  };
  auto index_of = [&](long long base) { return (base + lane < v1) ? (int)ind[base + lane] : -1; };
  if (v0 < v1) load_rows(v0, ra);
  if (v0 + 64 < v1) load_rows(v0 + 64, rb);
  int ida = index_of(v0), idb = index_of(v0 + 64);
  for (long long base = v0; base < v1; base += 128) {
    const int id0 = ida, id1 = idb;
    ida = index_of(base + 128);
    idb = index_of(base + 192);
    block(base, ra, id0);
    if (base + 64 < v1) block(base + 64, rb, id1);
  }
  __syncthreads();
  float* slab = ws + (size_t)blockIdx.x * (VQ_K * VQ_D + VQ_K);
  for (int i = tid; i < VQ_K * VQ_D + VQ_K; i += 1024) slab[i] = tab[i];
}

// counts / esum = the slabs summed in a fixed order: a workgroup = 64 consecutive elements x 8 slab groups (group k: slabs k, k + 8, ... in
// four running sums -- loads in flight instead of one latency chain: 60 -> 10 us), the groups combined through LDS in group order
__global__ __launch_bounds__(512) void vq_stats_reduce_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ counts,
                                                              float* __restrict__ esum) {
  __shared__ float red[512];
  constexpr int NE = VQ_K * VQ_D + VQ_K;
  const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + li;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < NE) {
    const float* p = ws + e;
    int b = grp;
    for (; b + 24 < nblk; b += 32) {
      s0 += p[(size_t)b * NE]; s1 += p[(size_t)(b + 8) * NE]; s2 += p[(size_t)(b + 16) * NE]; s3 += p[(size_t)(b + 24) * NE];
    }
    for (; b < nblk; b += 8) s0 += p[(size_t)b * NE];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < NE) {
    float s = red[li];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k * 64 + li];
    if (e < VQ_K * VQ_D) esum[e] = s;
    else counts[e - VQ_K * VQ_D] = s;
  }
}

// one workgroup per CU (the 131 KB table allows no more); the slab count is part of the workspace contract (fo_vq_stats_ws_bytes)
inline int stats_blocks(int64_t nvec) { return (int)std::max<int64_t>(1, std::min<int64_t>(fo_cu_count(), nvec / 512)); }

// EMA update (:66-75).  One workgroup, one thread per code.
__global__ __launch_bounds__(VQ_K) void vq_ema_kernel(float* embed, float* cluster_size, float* embed_avg,
                                                     const float* __restrict__ counts, const float* __restrict__ esum,
                                                     float decay, float alpha, float eps) {
  __shared__ float red[VQ_K];
  const int c = threadIdx.x;
  const float cs = cluster_size[c] * decay + counts[c] * alpha;
  cluster_size[c] = cs;
  red[c] = cs;
  __syncthreads();
  for (int o = VQ_K / 2; o > 0; o >>= 1) {
    if (c < o) red[c] += red[c + o];
    __syncthreads();
  }
  const float n = red[0];
  const float csn = (cs + eps) / (n + VQ_K * eps) * n;
  for (int d = 0; d < VQ_D; ++d) {
    const float ea = embed_avg[(size_t)d * VQ_K + c] * decay + esum[(size_t)c * VQ_D + d] * alpha;
    embed_avg[(size_t)d * VQ_K + c] = ea;
    embed[(size_t)d * VQ_K + c] = ea / csn;
  }
}

__global__ void vq_bwd_kernel(const float* __restrict__ gq, int ldg, const float* __restrict__ x, int ldx,
                              const float* __restrict__ q, int ldq, const float* __restrict__ gdiff, float scale,
                              float* __restrict__ gx, int ldgx, long long nvec) {
  const float gs = gdiff[0] * scale;
  const long long total = nvec * (VQ_D / 4);
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long v = e / (VQ_D / 4);
    const int c = (int)(e % (VQ_D / 4)) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gq + v * ldg + c);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + v * ldx + c);
    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + v * ldq + c);
    *reinterpret_cast<f32x4*>(gx + v * ldgx + c) = g + gs * (xv - qv);
  }
}

__global__ void vq_gather_kernel(const long long* __restrict__ ind, const float* __restrict__ embedT,
                                 float* __restrict__ q, int ldq, long long nvec) {
  const long long total = nvec * (VQ_D / 4);
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long v = e / (VQ_D / 4);
    const int c = (int)(e % (VQ_D / 4)) * 4;
    const long long i = ind[v];
    *reinterpret_cast<f32x4*>(q + v * ldq + c) = *reinterpret_cast<const f32x4*>(embedT + i * VQ_D + c);
  }
}

}  // namespace

extern "C" {

int fo_vq_prepare(const float* embed, float* embedT, float* enorm, void* stream) {
  hipLaunchKernelGGL(vq_prepare_kernel, dim3(VQ_K / 64), dim3(64), 0, (hipStream_t)stream, embed, embedT, enorm);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int64_t fo_vq_assign_ws_bytes() { return 4096; }      // one float per workgroup (one workgroup per CU)

int fo_vq_assign2(const float* x, int ldx, int64_t nvec, const float* embedT, const float* enorm, int64_t* ind,
                  float* q_ste, int ldq, float* sq_sum, void* q_bf16, int ldqb, const int64_t* forced_ind, float* ws, void* stream) {
  FO_REQUIRE(nvec > 0 && ldx >= VQ_D && ldq >= VQ_D && (!q_bf16 || ldqb >= VQ_D) && ws && sq_sum, FO_E_SHAPE, "vq_assign: bad shape / null workspace");
  const int cus = fo_cu_count();
  const int64_t ntiles = (nvec + 31) / 32;
  const int grid = (int)std::min<int64_t>(std::min(cus, 1024), (ntiles + 7) / 8);
  hipLaunchKernelGGL(vq_assign_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, x, ldx, (long long)nvec, embedT, enorm,
                     (long long*)ind, q_ste, ldq, ws, reinterpret_cast<__bf16*>(q_bf16), ldqb, (const long long*)forced_ind);
  FO_CHECK_LAUNCH();
  return fo_ordered_sum(ws, grid, sq_sum, stream);
}

int fo_vq_assign(const float* x, int ldx, int64_t nvec, const float* embedT, const float* enorm, int64_t* ind,
                 float* q_ste, int ldq, float* sq_sum, float* ws, void* stream) {
  return fo_vq_assign2(x, ldx, nvec, embedT, enorm, ind, q_ste, ldq, sq_sum, nullptr, 0, nullptr, ws, stream);
}

int64_t fo_vq_stats_ws_bytes(int64_t nvec) { return (int64_t)stats_blocks(nvec) * (VQ_K * VQ_D + VQ_K) * 4; }

int fo_vq_stats(const float* x, int ldx, int64_t nvec, const int64_t* ind, float* counts, float* esum, float* ws,
                void* stream) {
  FO_REQUIRE(nvec > 0 && ldx >= VQ_D, FO_E_SHAPE, "vq_stats: bad shape");
  const int nblk = stats_blocks(nvec);
  hipLaunchKernelGGL(vq_stats_kernel, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, x, ldx, (long long)nvec,
                     (const long long*)ind, ws);
  FO_CHECK_LAUNCH();
  hipLaunchKernelGGL(vq_stats_reduce_kernel, dim3((VQ_K * VQ_D + VQ_K + 63) / 64), dim3(512), 0, (hipStream_t)stream, ws,
                     nblk, counts, esum);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_vq_ema(float* embed, float* cluster_size, float* embed_avg, const float* counts, const float* esum, float decay,
              float alpha, float eps, void* stream) {
  hipLaunchKernelGGL(vq_ema_kernel, dim3(1), dim3(VQ_K), 0, (hipStream_t)stream, embed, cluster_size, embed_avg, counts, esum,
                     decay, alpha, eps);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_vq_bwd(const float* gq, int ldg, const float* x, int ldx, const float* q, int ldq, const float* gdiff, float scale,
              float* gx, int ldgx, int64_t nvec, void* stream) {
  const int grid = (int)std::min<int64_t>((nvec * 16 + 255) / 256, 4096);
  hipLaunchKernelGGL(vq_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gq, ldg, x, ldx, q, ldq, gdiff, scale, gx,
                     ldgx, (long long)nvec);
  FO_CHECK_LAUNCH();
  return FO_OK;
}

int fo_vq_gather(const int64_t* ind, const float* embedT, float* q, int ldq, int64_t nvec, void* stream) {
  const int grid = (int)std::min<int64_t>((nvec * 16 + 255) / 256, 4096);
  hipLaunchKernelGGL(vq_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long long*)ind, embedT, q, ldq,
                     (long long)nvec);
  FO_CHECK_LAUNCH();
  return FO_OK;
}
}
