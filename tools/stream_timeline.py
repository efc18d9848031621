#!/usr/bin/env python3
"""Per-stream view of one OVERLAPPED step from a rocprofv3 --kernel-trace CSV: for the last full step (between the last two Adam launches) the busy time of
every HIP stream / queue, when each goes quiet, the time nothing runs, and the kernels of the busiest stream in order with the gaps in front of them.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --perceptual --vqvae-dtype bf16 --steps 3 --warmup 2 \\
        --no-cpu-baseline --no-kernel-events --no-c5 --no-h2d-leg ;  python tools/stream_timeline.py gpurun_out/tl [top [optimiser launches in the window]]"""
import csv, glob, re, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], f"{r.get('Stream_Id')} (queue {r.get('Queue_Id')})"))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1                 # window = the last `back` optimiser launches (GAN: 3 per generator + discriminator pair)
lo, hi = rows[adam[-1 - back]][1], rows[adam[-1]][1]
step = [(max(s, lo), min(e, hi), n, q) for s, e, n, q in rows if e > lo and s < hi]
print(f"step {(hi - lo) / 1e6:.3f} ms, {len(step)} launches")
byq = collections.defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s, e, n))
ev = sorted([(s, 1) for s, e, n, q in step] + [(e, -1) for s, e, n, q in step])
cnt, last, hist = 0, lo, collections.Counter()
for t, d in ev:
    hist[min(cnt, 3)] += t - last
    last, cnt = t, cnt + d
hist[0] += hi - last
print("wall time with k kernels in flight:", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, n in kv[1])):
    busy = sum(e - s for s, e, n in ks)
    print(f"stream {q}: {len(ks):4d} launches, busy {busy / 1e6:7.3f} ms, first start +{(ks[0][0] - lo) / 1e6:.3f}, last end +{(ks[-1][1] - lo) / 1e6:.3f}")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if top:
    q = max(byq, key=lambda q: sum(e - s for s, e, n in byq[q]))
    prev = lo
    gaps = []
    for s, e, n in byq[q]:
        gaps.append((s - prev, n, s - lo, e - s)); prev = e
    print(f"largest gaps on stream {q} (gap us, at ms, kernel, its us):")
    for g, n, at, dur in sorted(gaps, reverse=True)[:top]:
        print(f"  {g / 1e3:8.1f}  +{at / 1e6:7.3f}  {re.sub(r'^void |.anonymous namespace.::', '', n)[:70]:70s} {dur / 1e3:8.1f}")
