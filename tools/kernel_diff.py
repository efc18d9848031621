#!/usr/bin/env python3
"""Per-kernel HIP-event totals of config 3 as timed under two environments, side by side (same process order on one device):
    python tools/kernel_diff.py "VAR=1 VAR2=0"        (the first run is the plain environment)"""
import json
import os
import subprocess
import sys

ARGS = "--perceptual --vqvae-dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-c3 --no-x6-leg --no-direct-leg --no-c5 --no-h2d-leg".split()


def run(extra):
    env = dict(os.environ)
    env.update(dict(kv.split("=", 1) for kv in extra.split()))
    r = subprocess.run([sys.executable, "bench.py"] + ARGS, env=env, capture_output=True, text=True)
    if not r.stdout.strip():
        sys.exit("bench.py printed nothing:\n" + r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return d["ms_per_step"], d.get("kernels", {})


a_ms, a = run("")
b_ms, b = run(sys.argv[1])
print(f"step: {a_ms:.3f} ms plain, {b_ms:.3f} ms with {sys.argv[1]}")
rows = []
for k in sorted(set(a) | set(b)):
    ta = a.get(k, {}).get("ms_per_step", 0.0)
    tb = b.get(k, {}).get("ms_per_step", 0.0)
    rows.append((tb - ta, k, ta, tb))
for dlt, k, ta, tb in sorted(rows)[:12] + sorted(rows)[-5:]:
    print(f"  {dlt:+.3f} ms  {ta:.3f} -> {tb:.3f}  {k}")
print(f"  sum of per-kernel changes {sum(r[0] for r in rows):+.3f} ms (serial: side streams joined)")
