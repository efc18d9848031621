"""Size-independent properties at BASELINE config 2's FULL sizes (160 frames of 256x256, T=5), where no CPU oracle
finishes in seconds: the three kernels of every conv layer must be mutually adjoint,

    <conv_W(x), g>  ==  <x, dgrad_W(g)>  ==  <W, wgrad(x, g)>            (bilinear form, fp64 dot products on the GPU)

which exercises the full-size code paths the small parity cases cannot: 655 360-pixel GEMM-M (int32 pixel and 64-bit byte
offsets, the 2 GiB buffer-descriptor windows with their margins), clip-padding tap skipping, the one-workgroup-per-CU
Conv3d kernel, split-K chunks cut by equal work, slab reduction over > 100 chunks.  Plus: bias gradient == column sums,
a whole C2 training step is finite, bit-reproducible in its gradients and actually moves every parameter tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, T = 160, 5
RTOL = 2e-4          # fp32 accumulation over up to 6.5e5 pixels x 3456 taps*channels per output element


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.rand(shape, device="cuda", generator=g) * 2 - 1) * scale


def _check(vals, what):
    ref = vals[0]
    for name, v in zip(("fwd", "dgrad", "wgrad"), vals):
        assert abs(v - ref) <= RTOL * max(abs(ref), 1e-6) + 1e-3 * RTOL, f"{what}: <{name}> = {v!r} vs <fwd> = {ref!r}"


@pytest.mark.parametrize("name,H,ci,co,k3d", [("conv3d_b 128->128 @64^2", 64, 128, 128, True), ("conv3d_t 128->128 @32^2", 32, 128, 128, True),
                                              ("conv2d k3 128->128 @64^2", 64, 128, 128, False), ("resblock k3 128->32 @64^2", 64, 128, 32, False)])
def test_conv_triple_is_adjoint_at_c2_size(name, H, ci, co, k3d):
    from faceoff_amd import ops
    kd = 3 if k3d else 1
    k, pad, Tt = (kd, 3, 3), (1 if k3d else 0, 1, 1), (T if k3d else 1)
    x = _rand((N, H, H, ci), 1)
    g = _rand((N, H, H, co), 2)
    w = _rand((co, ci, 3, 3, 3) if k3d else (co, ci, 3, 3), 3, scale=0.05)
    y = torch.empty((N, H, H, co), device="cuda")
    ops.conv_igemm(x, ops.pack_conv(w), None, y, T=Tt, k=k, pad=pad, cin=ci, cout=co)
    gx = torch.empty_like(x)
    ops.conv_igemm(g, ops.pack_conv_dgrad(w.reshape(co, ci, -1)), None, gx, T=Tt, k=k, pad=pad, cin=co, cout=ci)
    dw, db = torch.empty_like(w), torch.empty(co, device="cuda")
    ops.conv_wgrad(g, x, dw, db, T=Tt, k=k, pad=pad, a_real=co, b_real=ci)
    _check((_dot(y, g), _dot(x, gx), _dot(w, dw)), name)
    colsum = g.double().sum(dim=(0, 1, 2))
    assert (db.double() - colsum).abs().max().item() <= RTOL * colsum.abs().max().item() + 1e-2


def test_strided_and_transposed_convs_are_adjoint_at_c2_size():
    """enc_b.blocks.2 (Conv2d 64->128 k4 s2 p1 at 128^2 -> 64^2): its dgrad is the 4-phase transposed conv."""
    from faceoff_amd import ops
    ci, co = 64, 128
    x = _rand((N, 128, 128, ci), 4)
    g = _rand((N, 64, 64, co), 5)
    w = _rand((co, ci, 4, 4), 6, scale=0.05)
    y = torch.empty((N, 64, 64, co), device="cuda")
    ops.conv_igemm(x, ops.pack_conv(w), None, y, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=ci, cout=co)
    gx = torch.empty_like(x)
    ops.convT_phases(g, ops.pack_convT(w), None, gx, cin=co, cout=ci)
    dw = torch.empty_like(w)
    ops.conv_wgrad(g, x, dw, None, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=co, b_real=ci)
    _check((_dot(y, g), _dot(x, gx), _dot(w, dw)), "k4s2 64->128")


def test_c2_training_step_is_finite_reproducible_and_updates_everything():
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.synth import make_state_dict
    from faceoff_amd.trainer import FaceOffTrainer
    img, gt = _rand((N, 6, 256, 256), 7), _rand((N, 3, 256, 256), 8)
    runs = []
    for _ in range(2):
        eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), "cuda:0")
        before = eng.flat_params.clone()
        tr = FaceOffTrainer(eng)
        recon, latent, _ = tr.step(img, gt, T=T)
        torch.cuda.synchronize()
        assert torch.isfinite(recon).all() and torch.isfinite(latent).all()
        assert torch.isfinite(eng.flat_grads).all() and torch.isfinite(eng.flat_params).all()
        for key, (off, n) in eng.offsets.items():            # every one of the 70 tensors got a gradient and moved
            assert eng.flat_grads[off:off + n].abs().max().item() > 0, key
            assert not torch.equal(eng.flat_params[off:off + n], before[off:off + n]), key
        runs.append((eng.flat_grads.clone(), eng.flat_params.clone(), eng.buffers["quantize_b.embed"].clone(), recon.item(), latent.item()))
    # split-K slabs are reduced in a fixed order and the EMA code statistics are summed in an order that depends on the data only:
    # gradients, updated parameters and the updated codebook are the same bits every time.
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert torch.equal(runs[0][2], runs[1][2])
    # ... and so are the printed losses: the MSE and commitment sums leave their kernels as per-workgroup partials added in a fixed order
    # (float atomics before round 4)
    assert runs[0][3:] == runs[1][3:], (runs[0][3:], runs[1][3:])


# ---------------------------------------------------------------------------------------------------------------------
# The path bench.py times by default runs the Conv3d layers and the two 3x3 128->128 Conv2d layers as Winograd
# F(4x4,3x3): 36 banked planes x 160 frames through fo_conv_igemm_banked / fo_conv_wgrad_banked.  Value checks of that
# path AT THE TIMED SIZE (N = 160, T = 5) against the direct kernels on the same tensors (the direct kernels are oracle-
# and adjointness-checked above and in test_ops_gpu.py).  Bound: 2e-4 of the tensor's scale (F(4x4) fp32 error).
WINO_TOL = 2e-4


def _relmax(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("name,H,kd", [("conv3d_b @64^2", 64, 3), ("conv3d_t @32^2", 32, 3), ("conv2d 3x3 128->128 @64^2", 64, 1)])
def test_winograd_ops_equal_direct_kernels_at_c2_size(name, H, kd):
    from faceoff_amd import ops
    from faceoff_amd.ops import FO_OUT_RELU
    ci = co = 128
    Tt = T if kd == 3 else 1
    k, pad = (kd, 3, 3), (kd // 2, 1, 1)
    m = ops.wino_tile(H, H, N)
    assert m == 4, "C2 latents run F(4x4,3x3)"
    x = _rand((N, H, H, ci), 11)
    g = _rand((N, H, H, co), 12)
    msk = _rand((N, H, H, ci), 13)                      # a saved activation: its sign is the ReLU-backward mask
    res = _rand((N, H, H, ci), 14)                      # gradient fan-in operand
    w = _rand((co, ci, 3, 3, 3) if kd == 3 else (co, ci, 3, 3), 15, scale=0.05)
    b = _rand((co,), 16)
    obs = {}
    # forward (+ bias + ReLU), keeping the transformed input for the filter gradient like the training forward does
    y_d, y_w = torch.empty((N, H, H, co), device="cuda"), torch.empty((N, H, H, co), device="cuda")
    ops.conv_igemm(x, ops.pack_conv(w), b, y_d, T=Tt, k=k, pad=pad, cin=ci, cout=co, flags=FO_OUT_RELU)
    keep = ops.wino_wgrad_ok(H, H, N, Tt, m, kd)
    assert keep, "the timed path keeps V for the filter gradient at this size"
    V = ops.conv3d_winograd(x, ops.wino_filter(w, m=m), b, y_w, T=Tt, cin=ci, cout=co, flags=FO_OUT_RELU, keep_v=True, m=m, kd=kd)
    obs["fwd"] = _relmax(y_w, y_d)
    # data gradient with ReLU mask and fan-in add
    gx_d, gx_w = torch.empty_like(x), torch.empty_like(x)
    ops.conv_igemm(g, ops.pack_conv_dgrad(w.reshape(co, ci, -1)), None, gx_d, T=Tt, k=k, pad=pad, cin=co, cout=ci, mask=msk, add=res)
    ops.conv3d_winograd(g, ops.wino_filter(w, dgrad=True, m=m), None, gx_w, T=Tt, cin=co, cout=ci, mask=msk, add=res, m=m, kd=kd)
    obs["dgrad"] = _relmax(gx_w, gx_d)
    # filter + bias gradient, from the kept V (the timed path) and from a fresh input transform
    dw_d, db_d = torch.empty_like(w), torch.empty(co, device="cuda")
    ops.conv_wgrad(g, x, dw_d, db_d, T=Tt, k=k, pad=pad, a_real=co, b_real=ci)
    for tag, v in (("wgrad(kept V)", V), ("wgrad(fresh V)", None)):
        dw_w, db_w = torch.empty_like(w), torch.empty(co, device="cuda")
        ops.conv3d_wgrad_winograd(g, x, dw_w, db_w, T=Tt, a_real=co, b_real=ci, V=v, m=m, kd=kd)
        obs[tag] = _relmax(dw_w, dw_d)
        assert _relmax(db_w, db_d) <= 1e-5
    # the adjoint triple THROUGH the Winograd ops (no bias / mask / add): <conv(x), g> = <x, dgrad(g)> = <W, wgrad(x, g)>
    y0, gx0 = torch.empty_like(y_w), torch.empty_like(x)
    ops.conv3d_winograd(x, ops.wino_filter(w, m=m), None, y0, T=Tt, cin=ci, cout=co, m=m, kd=kd)
    ops.conv3d_winograd(g, ops.wino_filter(w, dgrad=True, m=m), None, gx0, T=Tt, cin=co, cout=ci, m=m, kd=kd)
    _check((_dot(y0, g), _dot(x, gx0), _dot(w, dw_w)), name + " (winograd)")
    print(f"[winograd vs direct, N={N} {name}] max rel err {obs}")
    for kname, e in obs.items():
        assert e <= WINO_TOL, (kname, e)


def test_c2_step_winograd_engine_equals_direct_engine():
    """The whole timed C2 step (160 frames of 256x256, T=5) on the default Winograd engine against the same engine on
    the direct kernels (FACEOFF_NO_WINOGRAD): code indices equal (a mismatch only where the top-2 margin, in fp64, is a
    near-tie below 1e-4), losses to 1e-5, decoder output to 1e-4, every one of the 70 gradient tensors to 1e-3 of its
    scale (the north-star parity bound)."""
    import os
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.synth import make_state_dict
    img, gt = _rand((N, 6, 256, 256), 7), _rand((N, 3, 256, 256), 8)
    res = []
    for direct in (False, True):
        if direct:
            os.environ["FACEOFF_NO_WINOGRAD"] = "1"
        try:
            eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), "cuda:0")
        finally:
            os.environ.pop("FACEOFF_NO_WINOGRAD", None)
        assert eng.winograd == (not direct)
        recon, diff, S = eng.loss_and_backward(img, gt, T=T)
        torch.cuda.synchronize()
        res.append(dict(dec=S["dec"].clone(), id_t=S["id_t"].clone(), id_b=S["id_b"].clone(), qt_in=S["qt_in"].clone(),
                        qb_in=S["qb_in"].clone(), recon=recon.item(), diff=diff.item(), grads=eng.flat_grads.clone(),
                        offsets=dict(eng.offsets), embed={l: eng.buffers[f"quantize_{l}.embed"].clone() for l in "tb"}))
        del eng, S
        torch.cuda.empty_cache()
    w, d = res
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    flips = 0
    for lvl in "tb":
        bad = (w["id_" + lvl] != d["id_" + lvl]).reshape(-1)
        if lvl == "b" and flips:
            # a flipped TOP code (a near-tie, gated above) is decoded into the bottom quantizer's input: around it that input differs
            # by O(1) between the two engines and the bottom codes with it.  Those positions are not compared (at most a 24 x 24
            # neighbourhood per flipped top code: dec_t's three 3x3 stages and its stride-2 stem); every other mismatch is gated.
            moved = (w["qb_in"] - d["qb_in"]).abs().reshape(-1, 64).max(1).values > 1e-3
            assert int(moved.sum()) <= 24 * 24 * flips, (int(moved.sum()), flips)
            bad = bad & ~moved
        nbad = int(bad.sum())
        if nbad:
            x = d[f"q{lvl}_in"].reshape(-1, 64)[bad].double()
            e = torch.from_numpy(sd[f"quantize_{lvl}.embed"]).cuda().double()
            dist = x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True)
            top2 = torch.topk(-dist, 2, dim=1).values
            margin = (top2[:, 0] - top2[:, 1]).max().item()
            assert margin < 1e-4, f"id_{lvl}: {nbad} mismatches, one with a top-2 margin of {margin:.3e} (outside the near-tie gate)"
            assert nbad <= 1e-5 * bad.numel() + 2
        flips += nbad
    np.testing.assert_allclose([w["recon"], w["diff"]], [d["recon"], d["diff"]], rtol=1e-5)
    dec_err = _relmax(w["dec"], d["dec"])
    worst = (0.0, "")
    tol = 1e-3 if flips == 0 else 2e-2
    for key, (off, n) in d["offsets"].items():
        a, b = w["grads"][off:off + n], d["grads"][off:off + n]
        err = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
        worst = max(worst, (err, key))
        assert err <= tol, (key, err)
    print(f"[C2 step winograd vs direct engine, N={N}] index flips {flips}; dec rel err {dec_err:.2e}; worst gradient {worst}")
    if flips == 0:
        assert dec_err <= 1e-4


def test_resblock_halo_kernels_equal_the_tiled_kernels_at_c2_size_and_are_reproducible():
    """The ResBlock family's halo-tile kernels (csrc/resblock_halo.hip, csrc/resblock_bwd.hip) at the size they are built for -- 160 frames of
    64 x 64 x 128, 10 240 tiles on 512 persistent workgroups, where tile walks, stage hand-overs and slab reductions run thousands of times
    (a hazard between a DMA and a late reader shows here and not in a three-tile test) -- against the tiled kernels they replace, same inputs:
    forward (hidden + output), the 3x3's masked data gradient with residual, its filter + bias gradient, the 1x1's one-pass backward.
    Equal up to summation order; each halo result bit-identical between two runs."""
    import json, os, subprocess, sys
    code = r"""
import sys, json, torch
sys.path.insert(0, %r)
from faceoff_amd import ops
Nn, H, W = 160, 64, 64
g = torch.Generator(device="cuda").manual_seed(7)
def rnd(shape, s=1.0): return (torch.rand(shape, device="cuda", generator=g) * 2 - 1) * s
x, go = rnd((Nn, H, W, 128)), rnd((Nn, H, W, 128))
w1, b1 = rnd((32, 128, 3, 3), 0.05), rnd((32,), 0.1)
w3, b3 = rnd((128, 32, 1, 1), 0.1), rnd((128,), 0.1)
wp1, wp3, wpd1 = ops.pack_conv(w1), ops.pack_conv(w3), ops.pack_conv_dgrad(w1.reshape(32, 128, -1))
h_in, gh_in = torch.relu(rnd((Nn, H, W, 32))), rnd((Nn, H, W, 32))
def run():
    hb = torch.empty((Nn, H, W, 32), device="cuda"); out = torch.empty((Nn, H, W, 128), device="cuda")
    ops.resblock_fwd(x, wp1, b1, wp3, b3, hb, out, True)
    # (the backward kernels get inputs that do not depend on the mode: a hidden value within rounding of zero would be a ReLU mask on
    # either side of the tie, and the comparison is of kernels, not of tie-breaking)
    gh = torch.empty_like(hb); dw3 = torch.empty((128, 32), device="cuda"); db3 = torch.empty((128,), device="cuda")
    ops.resblock_bwd_conv3(go, h_in, wp3, gh, dw3, db3)
    gx = torch.empty_like(x)
    ops.conv_igemm(gh_in, wpd1, None, gx, k=(1, 3, 3), stride=1, pad=(0, 1, 1), cin=32, cout=128, mask=x, add=go)
    dw1 = torch.empty((32, 128, 3, 3), device="cuda"); db1 = torch.empty((32,), device="cuda")
    ops.conv_wgrad(gh_in, x, dw1, db1, k=(1, 3, 3), stride=1, pad=(0, 1, 1), a_real=32, b_real=128, in_relu=True)
    torch.cuda.synchronize()
    return dict(h=hb, out=out, gh=gh, dw3=dw3, db3=db3, gx=gx, dw1=dw1, db1=db1)
r1, r2 = run(), run()
rep = {k: bool(torch.equal(r1[k], r2[k])) for k in r1}
torch.save({k: v.cpu() for k, v in r1.items()}, sys.argv[1])
print(json.dumps(rep))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    import tempfile
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("halo", "tiled"):
            env = dict(os.environ)
            env.pop("FACEOFF_FORCE_RESBLOCK_HALO", None)
            if mode == "tiled":
                env["FACEOFF_NO_RESBLOCK_HALO"] = "1"
            path = os.path.join(td, mode + ".pt")
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            rep = json.loads(r.stdout.strip().splitlines()[-1])
            assert all(rep.values()), (mode, rep)                       # run-to-run bit-identical (fixed tile walks, fixed slab orders)
            outs[mode] = torch.load(path)
    for k, tol in (("h", 2e-6), ("out", 2e-6), ("gh", 2e-6), ("gx", 4e-6), ("dw3", 2e-5), ("db3", 2e-5), ("dw1", 2e-5), ("db1", 2e-5)):
        a, b = outs["halo"][k].double(), outs["tiled"][k].double()
        assert (a - b).abs().max().item() <= tol * b.abs().max().item(), (k, (a - b).abs().max().item(), b.abs().max().item())

