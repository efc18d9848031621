import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_e2e_gpu as T
g = np.load("tests/golden/c2_oneclip.npz")
eng, recon, diff, S, img, gt = T._engine_step(g)
names = [str(n) for n in g["param_names"]]
gs = np.stack([T._stats(eng.grads[n]) for n in names])
a, b = np.sqrt(gs[:, 1]), np.sqrt(g["grad_stats"][:, 1])
for n, x, y in zip(names, a, b):
    if abs(x - y) > 3e-3 * abs(y): print(n, x, y)
