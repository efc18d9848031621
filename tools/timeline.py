#!/usr/bin/env python3
"""Who is alone on the GPU?  From a rocprofv3 --kernel-trace CSV of an OVERLAPPED step: wall time covered by kernels, split by what
is running: only matrix-bound kernels, only HBM-bound kernels (transforms, VQ, reductions, ...), both, nothing.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 3 --warmup 2 --no-c3 --no-c5 \\
        --no-direct-leg --no-cpu-baseline --no-kernel-events ;  python tools/timeline.py gpurun_out/tl"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
MATRIX = ("wino_gemm", "wino_wgrad_split", "conv_igemm", "conv_wgrad_kernel", "vq_assign", "conv_gen_kernel", "wgrad_gen_kernel", "conv_bf16", "conv_img_kernel",
          "wgrad_img_kernel", "conv_rgb", "resblock_halo", "resblock_wgrad1", "conv3x3_c32", "conv3x3_c128", "conv_halo64", "wgrad_bf16_kernel", "wgrad9_bf16_kernel")
# the last full step: find adam kernels
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1            # window = the last `back` optimiser launches (GAN: 3 per G+D pair)
lo, hi = rows[adam[-1 - back]][1], rows[adam[-1]][1]
ev = []
for s, e, n in rows:
    if e <= lo or s >= hi:
        continue
    kind = 0 if any(m in n for m in MATRIX) else 1
    ev.append((max(s, lo), 1, kind))
    ev.append((min(e, hi), -1, kind))
ev.sort()
cnt = [0, 0]
acc = {"matrix only": 0, "hbm only": 0, "both": 0, "idle": 0}
t = lo
for ts, d, kind in ev:
    dt = ts - t
    key = "both" if cnt[0] and cnt[1] else "matrix only" if cnt[0] else "hbm only" if cnt[1] else "idle"
    acc[key] += dt
    t = ts
    cnt[kind] += d
tot = hi - lo
print(f"step {tot / 1e6:.2f} ms: " + ", ".join(f"{k} {v / 1e6:.2f} ms" for k, v in acc.items()))
import re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"[<(].*$", "", n).strip()[:28]


# HBM-only time attributed to the kernels running then (split evenly), and a coarse map of where in the step it happens
alone = {}
where = [0.0] * 20
active = {}
ev2 = []
for i, (s, e, n) in enumerate(rows):
    if e <= lo or s >= hi:
        continue
    ev2.append((max(s, lo), 1, i))
    ev2.append((min(e, hi), -1, i))
ev2.sort()
t = lo
for ts, d, i in ev2:
    dt = ts - t
    if dt > 0 and active and not any(any(m in rows[j][2] for m in MATRIX) for j in active):
        for j in active:
            alone[short(rows[j][2])] = alone.get(short(rows[j][2]), 0) + dt / len(active)
        where[min(19, int((t - lo) * 20 / tot))] += dt
    t = ts
    if d > 0:
        active[i] = 1
    else:
        active.pop(i, None)
print("HBM-only time by kernel (ms):", ", ".join(f"{k} {v / 1e6:.2f}" for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:16]))
print("HBM-only ms per 5 % slice of the step:", " ".join(f"{v / 1e6:.2f}" for v in where))
