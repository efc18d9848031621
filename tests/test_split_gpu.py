"""The split-bf16 Winograd-domain GEMM (csrc/wino_gemm_split.hip) against the fp32-MFMA one and an fp64 product:
the six-term three-piece product is fp32 arithmetic to within a rounding or two, not a reduced-precision mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(sym, V, U, planes, N, T, P, cin, cout, kd):
    from faceoff_amd import _lib
    M = torch.full((planes, N * P, cout), float("nan"), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    if sym == "fo_wino_gemm_split":
        nb = _lib.load().fo_wino_gemm_split_ws_bytes(planes, cin, cout, kd)
        assert nb == planes * 3 * cout * kd * cin * 2
        ws = torch.full((nb // 2,), float("nan"), device="cuda", dtype=torch.bfloat16)
        _lib.call(sym, V.data_ptr(), U.data_ptr(), ws.data_ptr(), M.data_ptr(), planes, N, T, P, cin, cout, kd, s)
    else:
        _lib.call(sym, V.data_ptr(), U.data_ptr(), M.data_ptr(), planes, N, T, P, cin, cout, kd, s)
    torch.cuda.synchronize()
    return M


def _ref64(V, U, planes, N, T, P, cin, cout, kd):
    V64 = V.double().view(planes, N, P, cin)
    U64 = U.double().view(planes, cout, kd, cin)
    M = torch.zeros((planes, N, P, cout), dtype=torch.float64, device=V.device)
    for k in range(kd):
        s = k - kd // 2
        for n in range(N):
            t = n % T
            if 0 <= t + s < T:
                M[:, n] += torch.einsum("xpc,xoc->xpo", V64[:, n + s], U64[:, :, k])
    return M.view(planes, N * P, cout)


@pytest.mark.parametrize("planes,N,T,P,cin,cout,kd", [(36, 10, 5, 64, 128, 128, 3), (25, 1, 1, 1152, 256, 128, 1), (36, 6, 3, 64, 64, 256, 3),
                                                      (4, 8, 1, 16, 96, 128, 1),
                                                      (6, 10, 5, 128, 128, 128, 3), (4, 6, 3, 256, 64, 256, 3), (3, 4, 2, 384, 96, 128, 3),
                                                      (300, 5, 5, 128, 128, 128, 3)])
def test_split_gemm_is_fp32_accurate(planes, N, T, P, cin, cout, kd):
    g = torch.Generator().manual_seed(planes + cin)
    # wide dynamic range inside a row (Winograd-domain planes differ by orders of magnitude)
    V = (torch.randn((planes, N * P, cin), generator=g) * torch.exp(2 * torch.randn((planes, N * P, 1), generator=g))).cuda()
    U = (torch.randn((planes, cout, kd * cin), generator=g) * torch.exp(torch.randn((planes, cout, 1), generator=g))).cuda()
    ref = _ref64(V, U, planes, N, T, P, cin, cout, kd)
    native = _gemm("fo_wino_gemm", V, U, planes, N, T, P, cin, cout, kd).double()
    split = _gemm("fo_wino_gemm_split", V, U, planes, N, T, P, cin, cout, kd).double()
    assert torch.isfinite(split).all()
    # error relative to the magnitude an fp32 dot product is judged against: sum |v||u|
    mag = _ref64(V.abs(), U.abs(), planes, N, T, P, cin, cout, kd) + 1e-300
    e_native = ((native - ref).abs() / mag).max().item()
    e_split = ((split - ref).abs() / mag).max().item()
    r_native = ((native - ref).abs() / mag).pow(2).mean().sqrt().item()
    r_split = ((split - ref).abs() / mag).pow(2).mean().sqrt().item()
    print(f"[split gemm {planes}x{N * P}x{kd * cin}->{cout}] max err / sum|v||u|: fp32 MFMA {e_native:.2e}, bf16x6 {e_split:.2e};"
          f" rms {r_native:.2e} vs {r_split:.2e}")
    assert e_split <= 1.5 * e_native + 1e-9 and e_split <= 2.0 ** -20          # a few fp32 roundings (2^-24 each)
    assert r_split <= 1.5 * r_native + 1e-9


def test_split_gemm_exact_on_bf16_representable_inputs():
    """inputs with 8 significant bits: the first piece carries everything, products are exact in both kernels, and the two
    accumulate in fp32 -- results agree to accumulation-order rounding"""
    g = torch.Generator().manual_seed(3)
    planes, N, T, P, cin, cout, kd = 4, 4, 1, 32, 64, 128, 1
    V = torch.randn((planes, N * P, cin), generator=g).bfloat16().float().cuda()
    U = torch.randn((planes, cout, cin), generator=g).bfloat16().float().cuda()
    a = _gemm("fo_wino_gemm", V, U, planes, N, T, P, cin, cout, kd)
    b = _gemm("fo_wino_gemm_split", V, U, planes, N, T, P, cin, cout, kd)
    assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()


# ---------------------------------------------------------------- the training step with the split GEMMs (FACEOFF_BF16X6=1)
@pytest.fixture
def split_gemms(monkeypatch):
    from faceoff_amd import _lib, ops
    monkeypatch.setattr(ops, "BF16X6", True)
    calls = {"n": 0}
    real = _lib.call

    def spy(name, *a):
        if name == "fo_wino_gemm_split":
            calls["n"] += 1
        if name == "fo_wino_wgrad_split":
            calls["wgrad"] = calls.get("wgrad", 0) + 1
        assert name != "fo_wino_gemm", "the fp32-MFMA GEMM ran although BF16X6 is set"
        return real(name, *a)
    monkeypatch.setattr(_lib, "call", spy)
    return calls


def test_c2_clip_with_split_gemms(golden_dir, monkeypatch):
    """The reference's fixture at the C2 shape (256x256, one clip of 5) through the engine twice: GEMMs on the fp32 MFMA, and
    on the bf16 pipe.  Forward: every saved activation within 2e-5, code indices equal, the golden's losses within 1e-3, all 70
    gradient L2 norms within 1e-3 of the golden's.  Gradient entries: within 1e-3 of the fp32-MFMA engine's unless the 1e-6
    forward difference flips a ReLU mask (counted; the bound widens as for VQ near-ties -- tests/test_w42_gpu.py has the same
    gate; against the golden the fp32-MFMA engine itself sits at 0.98e-3 on enc_b.blocks.0.weight for that reason)."""
    import os
    from faceoff_amd import _lib, ops
    from test_e2e_gpu import _engine_step, _stats
    g = np.load(os.path.join(golden_dir, "c2_oneclip.npz"))
    e0, r0, d0, S0, *_ = _engine_step(g)
    calls = {"n": 0}
    real = _lib.call

    def spy(name, *a):
        calls["n"] += name == "fo_wino_gemm_split"
        assert name != "fo_wino_gemm"
        return real(name, *a)
    monkeypatch.setattr(_lib, "call", spy)
    monkeypatch.setattr(ops, "BF16X6", True)
    e1, r1, d1, S1, *_ = _engine_step(g)
    assert calls["n"] >= 20
    flips = 0
    for k, a in S0.items():
        b = S1.get(k)
        if torch.is_tensor(a) and torch.is_tensor(b) and a.is_floating_point() and a.shape == b.shape:
            assert (a - b).abs().max().item() <= 2e-5 * (a.abs().max().item() + 1e-30), k
            flips += int(((a > 0) != (b > 0)).sum().item())
    assert torch.equal(S0["id_t"], S1["id_t"]) and torch.equal(S0["id_b"], S1["id_b"])
    np.testing.assert_allclose([r1.item(), d1.item()], [float(g["recon"]), float(g["latent"])], rtol=1e-3)
    names = [str(n) for n in g["param_names"]]
    l2 = np.sqrt(np.stack([_stats(e1.grads[n]) for n in names])[:, 1])
    want_l2 = np.sqrt(g["grad_stats"][:, 1])
    assert np.max(np.abs(l2 - want_l2) / want_l2) <= 1e-3
    tol = 1e-3 if flips == 0 else 5e-2
    worst = max(((e1.grads[k] - v).abs().max().item() / (v.abs().max().item() + 1e-30), k) for k, v in e0.grads.items())
    print(f"[c2 one clip: GEMMs on the bf16 pipe vs fp32 MFMA] ReLU-mask flips {flips}, worst gradient rel diff {worst}, "
          f"gradient L2 norms vs golden {np.max(np.abs(l2 - want_l2) / want_l2):.2e}")
    assert worst[0] <= tol, (worst, flips)


@pytest.mark.parametrize("name,H,kd", [("conv3d_b @64^2", 64, 3), ("conv3d_t @32^2", 32, 3), ("conv2d 3x3 128->128 @64^2", 64, 1)])
def test_winograd_ops_equal_direct_kernels_at_c2_size_with_split_gemms(name, H, kd, split_gemms):
    import test_fullsize_gpu as F
    F.test_winograd_ops_equal_direct_kernels_at_c2_size(name, H, kd)
    assert split_gemms["n"] >= 2


def test_c2_step_with_split_gemms_equals_direct_engine_and_is_reproducible(split_gemms):
    import test_fullsize_gpu as F
    F.test_c2_step_winograd_engine_equals_direct_engine()
    F.test_c2_training_step_is_finite_reproducible_and_updates_everything()
    assert split_gemms["n"] > 20 and split_gemms.get("wgrad", 0) >= 11        # 6 Conv3d + 2 Conv2d 3x3 + 3 stems per backward


@pytest.mark.parametrize("cin,cout,hw", [(64, 128, (64, 64)), (32, 128, (48, 80))])
def test_stem_winograd_vs_torch_with_split_gemms(cin, cout, hw, split_gemms, monkeypatch):
    import test_w42_gpu as W
    W.test_conv_k4s2_winograd_vs_torch(cin, cout, hw, monkeypatch)
    assert split_gemms["n"] > 0


# ---------------------------------------------------------------- the filter-gradient GEMMs (csrc/wino_wgrad_split.hip)
def _wgrad_ref64(dM, V, planes, N, T, P, cin, cout, kd):
    A = dM.double().view(planes, N, P, cout)
    B = V.double().view(planes, N, P, cin)
    dU = torch.zeros((planes, cout, cin, kd), dtype=torch.float64, device=dM.device)
    for k in range(kd):
        s = k - kd // 2
        for n in range(N):
            if 0 <= n % T + s < T:
                dU[..., k] += torch.einsum("xpo,xpc->xoc", A[:, n], B[:, n + s])
    return dU


@pytest.mark.parametrize("planes,N,T,P,cin,cout,kd", [(36, 10, 5, 64, 128, 128, 3), (25, 1, 1, 1152, 256, 128, 1), (5, 6, 3, 32, 128, 256, 3),
                                                      (4, 7, 1, 96, 128, 128, 1), (2, 160, 5, 256, 128, 128, 3)])      # the last: C2's K
def test_split_wgrad_gemm_vs_fp64_and_the_fp32_kernel(planes, N, T, P, cin, cout, kd):
    from faceoff_amd import _lib
    from faceoff_amd.ops import _desc
    import ctypes as C
    g = torch.Generator().manual_seed(planes * 7 + cin)
    dM = (torch.randn((planes, N * P, cout), generator=g) * torch.exp(2 * torch.randn((planes, N * P, 1), generator=g))).cuda()
    V = (torch.randn((planes, N * P, cin), generator=g) * torch.exp(torch.randn((planes, N * P, 1), generator=g))).cuda()
    s = torch.cuda.current_stream().cuda_stream
    nb = _lib.load().fo_wino_wgrad_split_ws_bytes(planes, N, P, cin, cout, kd)
    assert nb > 0
    ws = torch.full((nb // 4,), float("nan"), device="cuda")
    dU = torch.full((planes, cout, cin, kd), float("nan"), device="cuda")
    _lib.call("fo_wino_wgrad_split", dM.data_ptr(), V.data_ptr(), dU.data_ptr(), ws.data_ptr(), C.c_int64(nb), planes, N, T, P, cin, cout, kd, s)
    # the fp32-MFMA kernel on the same operands (the call of ops._wgrad_winograd_dU)
    d = _desc(N=planes * N, T=T if kd > 1 else 1, Hin=1, Win=P, Hm=1, Wm=P, Hout=1, Wout=P, Cin=cin, Cout=cout, KD=kd, KH=1, KW=1, stride=1,
              padD=kd // 2, padH=0, padW=0, ostride=1, ophH=0, ophW=0, ldIn=cin, ldOut=cout, ldMask=0, ldAdd=0, flags=0)
    nb2 = _lib.load().fo_wgrad_banked_ws_bytes(C.byref(d), planes)
    ws2 = torch.empty(nb2 // 4 + 16, device="cuda")
    dU2 = torch.full((planes, cout, cin, kd), float("nan"), device="cuda")
    _lib.call("fo_conv_wgrad_banked", C.byref(d), dM.data_ptr(), V.data_ptr(), dU2.data_ptr(), cout, cin, ws2.data_ptr(), C.c_int64(nb2), planes, s)
    torch.cuda.synchronize()
    ref = _wgrad_ref64(dM, V, planes, N, T, P, cin, cout, kd)
    mag = _wgrad_ref64(dM.abs(), V.abs(), planes, N, T, P, cin, cout, kd) + 1e-300
    e_split = ((dU.double() - ref).abs() / mag).max().item()
    e_native = ((dU2.double() - ref).abs() / mag).max().item()
    print(f"[split wgrad gemm {planes}x{N * P} rows {cout}x{kd * cin}] max err / sum|a||b|: fp32 MFMA {e_native:.2e}, bf16x6 {e_split:.2e}")
    assert torch.isfinite(dU).all()
    assert e_split <= 2 * e_native + 1e-9          # (K = N * P rows: the fp32 accumulation error grows with it in both)
