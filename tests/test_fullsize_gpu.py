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
        runs.append((eng.flat_grads.clone(), eng.flat_params.clone(), eng.buffers["quantize_b.embed"].clone()))
    # split-K slabs are reduced in a fixed order: gradients and updated parameters are the same bits every time.  The
    # EMA code statistics are summed with LDS float atomics (order varies inside a workgroup): equal to rounding.
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    e0, e1 = runs[0][2], runs[1][2]
    assert (e0 - e1).abs().max().item() <= 1e-5 * e0.abs().max().item()
