import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd import ops, _lib
from faceoff_amd.ops import _ptr, ld_of, _stream
N, H, ci, co, T, kd = 160, 64, 128, 128, 5, 3
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((N, H, H, ci), device="cuda", generator=g) * 2 - 1
w = (torch.rand((co, ci, 3, 3, 3), device="cuda", generator=g) * 2 - 1) * 0.05
Ht = Wt = H // 4
pv = N * Ht * Wt * ci
V0 = torch.empty(36 * pv, device="cuda"); V1 = torch.full((36 * pv,), float("nan"), device="cuda")
_lib.call("fo_wino_input", _ptr(x), ld_of(x), _ptr(V0), N, H, H, ci, 4, _stream())
for h in (0, 1):
    _lib.call("fo_wino_input_rows", _ptr(x), ld_of(x), _ptr(V1), N, H, H, ci, h, _stream())
torch.cuda.synchronize()
print("V equal:", torch.equal(V0, V1), "max diff", (V0 - V1).abs().max().item())
for p in range(36):
    if not torch.equal(V0[p * pv:(p + 1) * pv], V1[p * pv:(p + 1) * pv]):
        print(" plane", p, (V0[p * pv:(p + 1) * pv] - V1[p * pv:(p + 1) * pv]).abs().max().item())
U = ops.wino_filter(w, m=4)
pm = N * Ht * Wt * co
M0 = torch.empty(36 * pm, device="cuda"); M1 = torch.full((36 * pm,), float("nan"), device="cuda")
ops.wino_gemm(V0, U, M0, 36, N, T, Ht * Wt, ci, co, kd)
bank = co * kd * ci
ops.wino_gemm(V0, U, M1, 18, N, T, Ht * Wt, ci, co, kd)
ops.wino_gemm(V0[18 * pv:], U[18 * bank:], M1[18 * pm:], 18, N, T, Ht * Wt, ci, co, kd)
torch.cuda.synchronize()
print("M equal:", torch.equal(M0, M1))
for p in range(36):
    a, b = M0[p * pm:(p + 1) * pm], M1[p * pm:(p + 1) * pm]
    if not torch.equal(a, b):
        print(" M plane", p, "nan" if torch.isnan(b).any().item() else (a - b).abs().max().item())
