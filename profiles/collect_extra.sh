#!/bin/bash
# rocprofv3 kernel-trace summaries of the config-3 and config-5 legs:  bash profiles/collect_extra.sh r02
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_${TAG}_extra
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o c3 -- python3 bench.py --steps 2 --warmup 1 --perceptual --vqvae-dtype bf16 --no-cpu-baseline --no-kernel-events --no-c5 --no-h2d-leg --serial-streams > "$OUT/c3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o c5 -- python3 tools/bench_gan.py 6 > "$OUT/c5.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, "profiles")
from summarize import demangle
def short(n):
    n = demangle(n); n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*\)$", "", n).strip()
for leg, title in (("c3", "bench.py --perceptual --vqvae-dtype bf16 --serial-streams --steps 2 --warmup 1 (C3, bf16 MFMA operands throughout: 3 steps traced)"), ("c5", "tools/bench_gan.py 6 (C5: 2 warm-up + 7-8 GAN iterations traced)")):
    rows = []
    for f in glob.glob(os.path.join(out, leg, "**", "*kernel_stats.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows: continue
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    os.makedirs("gpurun_out/profiles_" + tag, exist_ok=True)
    with open(f"gpurun_out/profiles_{tag}/{tag}_{leg}_kernel_stats.md", "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --stats: {title}, tag {tag}\n\n| kernel | calls | total ms | avg us | % of GPU time |\n|---|---|---|---|---|\n")
        for r in rows[:30]:
            fh.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.2f} |\n")
        fh.write(f"\nTotal GPU kernel time: {tot/1e6:.2f} ms over the traced run.\n")
    print("wrote", leg)
PY
