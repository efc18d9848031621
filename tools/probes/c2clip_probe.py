"""Probe: the 256x256 one-clip golden against the engine in its three conv3d modes; where does the first-layer gradient error come from?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_e2e_gpu import _engine_step, _sub
g = np.load("tests/golden/c2_oneclip.npz")
names = [str(n) for n in g["param_names"]]
res = {}
MODES = [("default", {}), ("direct", {"FACEOFF_NO_WINOGRAD": "1"})]
for lay in ("enc_b.blocks.2", "dec.blocks.4", "dec_t.blocks.4"):
    for ps in ("fwd", "dgrad", "wgrad"):
        MODES.append((f"skip {lay}:{ps}", {"FACEOFF_W42_SKIP": f"{lay}:{ps}"}))
MODES.append(("skip enc_b.2 all", {"FACEOFF_W42_SKIP": "enc_b.blocks.2:fwd,enc_b.blocks.2:dgrad,enc_b.blocks.2:wgrad"}))
for mode, env in MODES:
    os.environ.update(env)
    eng, recon, diff, S, img, gt = _engine_step(g)
    for k in env: os.environ.pop(k)
    off = 0; errs = []
    for i, n in enumerate(names):
        got = _sub(eng.grads[n]); want = g["grad_sub"][off:off + len(got)]; off += len(got)
        scale = max(np.sqrt(g["grad_stats"][i, 1] / eng.grads[n].numel()), np.abs(want).max()) + 1e-30
        errs.append((float(np.abs(got - want).max() / scale), n))
    errs.sort(reverse=True)
    print(mode, "worst 4:", [(f"{e:.2e}", n) for e, n in errs[:4]], "median %.2e" % errs[len(errs)//2][0])
    res[mode] = {k: v.clone() for k, v in S.items() if torch.is_tensor(v) and v.is_floating_point() and v.dim() == 4}
    res[mode + "_g"] = {n: eng.grads[n].clone() for n in names}
