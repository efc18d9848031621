"""The split-bf16 Winograd-domain GEMM (csrc/wino_gemm_split.hip) against the fp32-MFMA one and an fp64 product:
the six-term three-piece product is fp32 arithmetic to within a rounding or two, not a reduced-precision mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(sym, V, U, planes, N, T, P, cin, cout, kd):
    from faceoff_amd import _lib
    M = torch.full((planes, N * P, cout), float("nan"), device="cuda")
    _lib.call(sym, V.data_ptr(), U.data_ptr(), M.data_ptr(), planes, N, T, P, cin, cout, kd, torch.cuda.current_stream().cuda_stream)
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
                                                      (4, 8, 1, 16, 96, 128, 1)])
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
    assert e_split <= 2.0 ** -21          # a few fp32 roundings (2^-24 each)
    assert r_split <= 4 * r_native + 1e-9


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
