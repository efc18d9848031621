/*
 * vq_oracle.c -- TEST INFRASTRUCTURE.  Plain-C restatement of the nearest-code search of
 * Quantize.forward (reference models/vqvae_conv3d_latent.py:47-57,77) with the summation order
 * pinned, so that code indices can be compared BIT-EXACTLY with the HIP kernel (fo_vq_assign).
 *
 * The reference computes, in fp32 (torch CPU / MKL sgemm, order unspecified):
 *     dist = x.pow(2).sum(1) - 2 * x @ embed + embed.pow(2).sum(0)          (:49-53)
 *     ind  = argmax(-dist)  (first maximal index)                           (:54)
 * This file pins the order: dot = fma chain over k = 0..63 from 0; ||v||^2 = (chain over even k)
 * + (chain over odd k); dist = (xx - 2*dot) + ee.  Against the reference's own indices the
 * result can differ only where the top-2 margin is within fp32 rounding of the distance
 * (tests gate on the margin stored in tests/golden/quantize_kat.npz).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o oracle/_build/libvq_oracle.so oracle/vq_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>

#define D 64

static float sqnorm(const float* v, int stride) {
  float ev = 0.f, od = 0.f;
  for (int k = 0; k < D; k += 2) {
    ev = fmaf(v[(long)k * stride], v[(long)k * stride], ev);
    od = fmaf(v[(long)(k + 1) * stride], v[(long)(k + 1) * stride], od);
  }
  return ev + od;
}

/* x[nvec][64] row-major, embed[64][n_embed] (the reference buffer layout).
 * Outputs: ind[nvec] (int64), q_ste[nvec][64] = x + (q - x), *sq_sum = sum (q-x)^2 (double),
 * counts[n_embed], esum[64][n_embed] (reference layout, :60-61), best_dist[nvec] (may be NULL). */
void vq_oracle_assign(const float* x, long nvec, const float* embed, int n_embed, int64_t* ind, float* q_ste,
                      double* sq_sum, float* counts, float* esum, float* best_dist) {
  float ee[4096];
  for (int c = 0; c < n_embed; ++c) ee[c] = sqnorm(embed + c, n_embed);
  for (int c = 0; c < n_embed; ++c) counts[c] = 0.f;
  for (long i = 0; i < (long)D * n_embed; ++i) esum[i] = 0.f;
  double sq = 0.0;
  for (long v = 0; v < nvec; ++v) {
    const float* xv = x + v * D;
    const float xx = sqnorm(xv, 1);
    float best = INFINITY;
    int bi = 0;
    for (int c = 0; c < n_embed; ++c) {
      float dot = 0.f;
      for (int k = 0; k < D; ++k) dot = fmaf(embed[(long)k * n_embed + c], xv[k], dot);
      const float two = 2.f * dot;
      const float t = xx - two;
      const float d = t + ee[c];
      if (d < best) { best = d; bi = c; }
    }
    ind[v] = bi;
    if (best_dist) best_dist[v] = best;
    counts[bi] += 1.f;
    for (int k = 0; k < D; ++k) {
      const float q = embed[(long)k * n_embed + bi];
      const float diff = q - xv[k];
      q_ste[v * D + k] = xv[k] + diff;
      sq += (double)diff * (double)diff;
      esum[(long)k * n_embed + bi] += xv[k];
    }
  }
  *sq_sum = sq;
}
