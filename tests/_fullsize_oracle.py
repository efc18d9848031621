"""The CPU oracle at the sizes the metric is quoted on: 32 clips x 5 frames of 256x256 in one
`oracle.train_step` would hold ~12 GB of autograd state for the VQ-VAE alone and several times that with the LPIPS branch, so the
step is evaluated CLIP CHUNK BY CLIP CHUNK and added up -- which is the same function, by linearity:

  * clips never interact in the forward (Conv3d mixes frames within a clip only, models/vqvae_conv3d_latent.py:247-251), so
    `dec`, the code indices and the quantisers' inputs of a chunk are those of the whole batch;
  * every loss term is a MEAN over frames / latent vectors (train_faceoff_perceptual.py:37-40, loss.py:33, vqvae_conv3d_latent.py:77),
    so loss = sum_chunks (frames_chunk / frames) * loss_chunk and the gradient of the whole step is the same weighted sum of the chunks'
    gradients (the only difference to one big backward is fp32 summation order);
  * the EMA statistics are SUMS over all vectors (:60-64): the buffers are made at the end by the oracle's own `quantize_forward` on the
    concatenated quantiser inputs with the (chunk-wise) codes forced.

`tests/test_fullsize_oracle_cpu.py` holds this file to `oracle.train_step` on the whole batch at a size where both run.
Test infrastructure only (imports oracle/)."""
import os

import numpy as np
import torch

from oracle import faceoff_oracle as O


def cgroup_cpus():
    """CPUs this process may use (the GPU boxes give 16 of 256 by cgroup quota; oversubscribing them is 100x slower)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(float(quota) / float(period)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


# oracle ReLU site (oracle.faceoff_oracle.ForcedReLU) -> the tensor of VQVAEEngine's forward state S whose sign is that ReLU's branch:
# post-ReLU outputs, and for a ResBlock's leading ReLU the block's raw input (engine.stage_encode / stage_conv3d / stage_quantize / stage_decode)
RELU_SITES = {
    "enc_b.blocks.0": "a0", "enc_b.blocks.2": "a1",
    "enc_b.blocks.5.in": "a2", "enc_b.blocks.5.hid": "h_eb5", "enc_b.blocks.6.in": "a3", "enc_b.blocks.6.hid": "h_eb6", "enc_b.blocks.6.out": "eb",
    "enc_t.blocks.0": "t0", "enc_t.blocks.3.in": "t1", "enc_t.blocks.3.hid": "h_et3", "enc_t.blocks.4.in": "t2", "enc_t.blocks.4.hid": "h_et4",
    "enc_t.blocks.4.out": "et",
    "conv3d_encoded_b.conv3d.0": "c1", "conv3d_encoded_b.conv3d.1": "c2", "conv3d_encoded_t.conv3d.0": "d1", "conv3d_encoded_t.conv3d.1": "d2",
    "dec_t.blocks.1.in": "u0", "dec_t.blocks.1.hid": "h_dt1", "dec_t.blocks.2.in": "u1", "dec_t.blocks.2.hid": "h_dt2", "dec_t.blocks.2.out": "u2",
    "dec.blocks.1.in": "v0", "dec.blocks.1.hid": "h_d1", "dec.blocks.2.in": "v1", "dec.blocks.2.hid": "h_d2", "dec.blocks.2.out": "v2",
    "dec.blocks.4": "w1",
}


def engine_relu_masks(S):
    """The ReLU branches the engine took in the forward that left S (fp32 engine): site -> bool [N,C,H,W] on the CPU."""
    out = {}
    for site, key in RELU_SITES.items():
        t = S[key]
        out[site] = (t > 0).permute(0, 3, 1, 2).contiguous().cpu()
    return out


def forced_relu(masks, f0, f1, T):
    """oracle.ForcedReLU for frames f0..f1 of the batch (clips of T frames); the Conv3d sites see [B,C,T,H,W]"""
    sl = {}
    for site, m in masks.items():
        m = m[f0:f1]
        if site.startswith("conv3d_"):
            n, c, h, w = m.shape
            m = m.reshape(n // T, T, c, h, w).permute(0, 2, 1, 3, 4)
        sl[site] = m
    return O.ForcedReLU(sl)


def oracle_step_chunked(img, gt, sd, lpips_state=None, bf16sim=False, lpips_bf16sim=False, force_ids=None, clips_per_chunk=4,
                        keep_dec=True, threads=None, relu_masks=None):
    """img [B,T,6,H,W], gt [B,T,3,H,W] (CPU float tensors), sd = reference-keyed numpy state dict.
    Returns dict(recon, latent, perceptual: python floats; grads {name: tensor}; id_t, id_b; qt_in, qb_in (fp32, detached);
    dec [N,6,H,W] or None; buffers {name: tensor} = the six EMA buffers after the step; relu_diffs: with relu_masks (engine_relu_masks of the
    engine's forward state: the oracle takes the ENGINE's ReLU branches) the list of (site, units, largest |x| / tensor scale) where those
    differ from the oracle's own x > 0)."""
    prev = torch.get_num_threads()
    torch.set_num_threads(threads or min(cgroup_cpus(), 32))
    try:
        B, T = img.shape[:2]
        N = B * T
        p = O.to_torch_state(sd)
        lpt = None if lpips_state is None else {k: torch.as_tensor(v) for k, v in lpips_state.items()}
        acc = {k: torch.zeros_like(v) for k, v in p.items() if v.requires_grad}
        tot = dict(recon=0.0, latent=0.0, perceptual=0.0)
        parts = dict(id_t=[], id_b=[], qt_in=[], qb_in=[], dec=[])
        relu_diffs = []
        for c0 in range(0, B, clips_per_chunk):
            c1 = min(B, c0 + clips_per_chunk)
            wgt = (c1 - c0) / B
            fi = None
            if force_ids is not None:
                fi = tuple(f.reshape(B, T, *f.shape[-2:])[c0:c1].reshape(-1, *f.shape[-2:]) for f in force_ids)
            for v in p.values():
                v.grad = None
            fr = None if relu_masks is None else forced_relu(relu_masks, c0 * T, c1 * T, T)
            r = O.run_step(img[c0:c1], gt[c0:c1], p, lpt, training=True, lpips_bf16sim=lpips_bf16sim, bf16sim=bf16sim, force_ids=fi, relu=fr)
            if fr is not None:
                relu_diffs += fr.diffs
            (r["loss"] * wgt).backward()
            for k in acc:
                acc[k] += p[k].grad
            for k in tot:
                tot[k] += wgt * float(r[k].detach())
            fw = r["fw"]
            parts["id_t"].append(fw["id_t"])
            parts["id_b"].append(fw["id_b"])
            parts["qt_in"].append(fw["qt_in"].detach())
            parts["qb_in"].append(fw["qb_in"].detach())
            if keep_dec:
                parts["dec"].append(fw["dec"].detach())
            del r, fw
        out = {k: torch.cat(v) for k, v in parts.items() if v}
        out.setdefault("dec", None)
        out.update(tot)
        out["grads"] = acc
        out["relu_diffs"] = relu_diffs
        # EMA buffers from the statistics of ALL vectors: the oracle's own Quantize restatement on the concatenated inputs, codes forced
        # to the ones the chunks chose (identical to what it would choose: same distances) -- in slices of vectors, because the EMA update
        # is linear in the two statistics and a [655 360, 512] one-hot + distance matrix pair is 2.7 GB
        bufs = {}
        with torch.no_grad():
            for lvl in "tb":
                names = [f"quantize_{lvl}.{s}" for s in ("embed", "cluster_size", "embed_avg")]
                embed, cs, ea = (p[n] for n in names)
                x = out[f"q{lvl}_in"].reshape(-1, embed.shape[0])
                ids = out["id_" + lvl].reshape(-1)
                onehot_sum = torch.zeros_like(cs, dtype=torch.float64)
                embed_sum = torch.zeros_like(ea, dtype=torch.float64)
                for s in range(0, x.shape[0], 1 << 16):
                    sums = {}

                    def grab(t, _s=sums):
                        _s[t.dim()] = t
                        return t
                    O.quantize_forward(x[s:s + (1 << 16)], embed, torch.zeros_like(cs), torch.zeros_like(ea), True, all_reduce=grab,
                                       force_ind=ids[s:s + (1 << 16)])
                    onehot_sum += sums[1].double()
                    embed_sum += sums[2].double()
                # ... and the update itself (:66-75) by the oracle on those global sums: all_reduce hands them in
                it = iter((onehot_sum.float(), embed_sum.float()))
                _, _, _, new = O.quantize_forward(x[:1], embed, cs, ea, True, all_reduce=lambda t: next(it), force_ind=ids[:1])
                for n, s in zip(names, ("embed", "cluster_size", "embed_avg")):
                    bufs[n] = new[s]
        out["buffers"] = bufs
        return out
    finally:
        torch.set_num_threads(prev)


def rel_to_scale(got, want):
    """max |got - want| over the tensor's scale (max(rms, max|want|) -- the e2e tests' measure)"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(float(np.sqrt((want ** 2).mean())), float(np.abs(want).max())) + 1e-30
    return float(np.abs(got - want).max() / scale)
