#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (profiles/collect.sh) into small, committable files.

  python3 profiles/summarize.py <raw_dir> <out_dir> <tag>

Outputs: <tag>_kernel_stats.md (per-kernel time table from --kernel-trace --stats),
<tag>_pmc.md (per-kernel counters: MFMA busy share, effective clock, HBM bytes with the gfx950
FETCH_SIZE x2 correction for 16-B/lane streams, L2 hit rate) and pmc_traffic.json (bench.py kernel
label -> HBM bytes per launch, read back by bench.py's roofline.traffic).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

CUS, SIMDS = 256, 4


_DEMANGLED = {}


def demangle(name):
    """rocprofv3 leaves some symbols mangled (a kernel whose parameter list names a vector type): llvm-cxxfilt, where the ROCm image has it."""
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLED:
        out = name
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                import subprocess
                out = subprocess.run([tool, name], capture_output=True, text=True, timeout=10).stdout.strip() or name
                break
            except Exception:
                continue
        if out.startswith("_Z"):                           # (llvm-cxxfilt does not know the __bf16 vector mangling): the identifier at least
            m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", out) or re.match(r"_Z(\d+)", out)
            if m:
                n0 = m.end()
                out = out[n0:n0 + int(m.group(1))]
        _DEMANGLED[name] = out
    return _DEMANGLED[name]


def short(name):
    name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.strip()


def label(name):
    """bench.py's per-kernel key for a rocprof kernel name: since round 4 that IS the short symbol (ops.KernelProfiler keys launches by what
    the library reports through fo_last_kernel, which is what rocprofv3 prints).  Small reduce / layout kernels are left out of the table."""
    n = short(name)
    if re.search(r"(conv_|wgrad\d*_|wino_gemm|wino_wgrad|resblock_|vq_assign|disc_head)", n) and "reduce" not in n and "pack" not in n:
        return n
    return None


def find(raw, sub, pattern):
    return sorted(glob.glob(os.path.join(raw, sub, "**", pattern), recursive=True))


def read_counters(raw, sub):
    """kernel -> counter -> list of per-dispatch values ; kernel -> list of durations (ns)"""
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(dict)
    for f in find(raw, sub, "*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r.get("Kernel_Name", ""))
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                try:
                    durs[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
    return vals, durs


def main():
    raw, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    config = sys.argv[4] if len(sys.argv) > 4 else "c2"       # c3: the traced command is bench.py --perceptual --vqvae-dtype bf16 (profiles/collect.sh)
    what = {"c2": "bench.py --steps 2 --warmup 1 --serial-streams",
            "c3": "bench.py --perceptual --vqvae-dtype bf16 --serial-streams --steps 2 --warmup 1 (config 3 as timed: bf16 MFMA operands for the VQ-VAE and LPIPS)",
            "c5": "tools/bench_gan.py 6 --serial (config 5: GAN iterations on one 30-frame clip, generator and discriminator alternating, side streams folded)"}[config]
    os.makedirs(out, exist_ok=True)
    # ---- kernel stats
    stats = []
    for f in find(raw, "trace", "*kernel_stats.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                stats.append(r)
    trace_avg, trace_share = {}, {}
    if stats:
        tot = sum(float(r["TotalDurationNs"]) for r in stats) or 1.0
        stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
        with open(os.path.join(out, f"{tag}_kernel_stats.md"), "w") as fh:
            fh.write(f"# rocprofv3 --kernel-trace --stats: {what} ({'3 steps' if config != 'c5' else '2 warm-up + 7-8 iterations'} traced), tag {tag}\n\n")
            fh.write("| kernel | calls | total ms | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|---|\n")
            for r in stats[:40]:
                n = short(r["Name"])
                trace_avg[n] = float(r["AverageNs"])
                trace_share[n] = float(r["TotalDurationNs"]) / tot
                fh.write(f"| `{n}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                         f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.2f} |\n")
            fh.write(f"\nTotal GPU kernel time: {tot/1e6:.2f} ms over the traced run.\n")
    # per-dispatch durations of the trace run, first dispatch of each kernel dropped (code-object load / cold caches: a 3-call kernel's
    # average is otherwise its first call) -- the figure the consistency check below compares
    trace_warm = {}
    per = defaultdict(list)
    for f in find(raw, "trace", "*kernel_trace.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    for k, v in per.items():
        v = [d for _, d in sorted(v)]
        v = v[1:] if len(v) > 1 else v
        trace_warm[k] = sum(v) / len(v)
    # ---- counters
    sq, sq_d = read_counters(raw, "pmc_sq")
    fe, fe_d = read_counters(raw, "pmc_fetch")
    wr, _ = read_counters(raw, "pmc_write")
    l2, _ = read_counters(raw, "pmc_l2")
    kernels = sorted(set(sq) | set(fe) | set(wr) | set(l2), key=lambda k: -sum(sq_d.get(k, {}).values()))
    traffic = {}
    disagree = []
    if kernels:
        with open(os.path.join(out, f"{tag}_pmc.md"), "w") as fh:
            fh.write(f"# rocprofv3 --pmc passes (separate runs) of {what}, per-dispatch averages, tag {tag}\n\n")
            fh.write("MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 4 SIMD * 256 CU); effective clock = "
                     "GRBM_GUI_ACTIVE / 8 / duration.  HBM read = 2 x FETCH_SIZE KB (gfx950 counts 128-B requests of wide "
                     "streams as 64 B; MI355X_MICROARCH.md, HBM), HBM write = WRITE_SIZE KB.\n\n")
            fh.write("Consistency (round 5): `trace avg ms` is the same kernel's average in the --kernel-trace --stats run of the same command (both columns without the kernel's first dispatch); a row whose two "
                     "averages differ by more than 10 % (kernels with >= 1 % of the GPU time) is flagged and makes this script exit non-zero -- the counters "
                     "of such a row do not describe the launches that cost the time (round 4: conv5_x data gradients beside the LPIPS heads).\n\n")
            fh.write("| kernel | dispatches | avg ms (profiled) | trace avg ms | MFMA busy share | eff. clock GHz | wave cycles waiting (WAIT_ANY/WAVE_CYCLES) | "
                     "LDS bank-conflict cycles / wave cycles | HBM read MB/launch | HBM write MB/launch | L2 hit rate |\n|" + "---|" * 11 + "\n")
            for k in kernels[:30]:
                def avg(src, c):
                    v = src.get(k, {}).get(c)
                    return sum(v) / len(v) if v else None
                dl = list(sq_d.get(k, {}).values()) or list(fe_d.get(k, {}).values())
                dur = sum(dl) / len(dl) if dl else None
                mf, gui = avg(sq, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(sq, "GRBM_GUI_ACTIVE")
                wc, wa, ldsb = avg(sq, "SQ_WAVE_CYCLES"), avg(sq, "SQ_WAIT_ANY"), avg(sq, "SQ_LDS_BANK_CONFLICT")
                f_, w_ = avg(fe, "FETCH_SIZE"), avg(wr, "WRITE_SIZE")
                hit, miss = avg(l2, "TCC_HIT_sum"), avg(l2, "TCC_MISS_sum")
                share = mf / (gui / 8 * SIMDS * CUS) if mf and gui else None
                clk = gui / 8 / dur if gui and dur else None
                rd = 2 * f_ * 1024 if f_ is not None else None
                wrb = w_ * 1024 if w_ is not None else None
                lab = label(k)
                if lab and rd is not None:
                    traffic[lab] = int(rd + (wrb or 0))
                fmt = lambda v, s="{:.3f}": s.format(v) if v is not None else "-"
                tavg = trace_warm.get(k, trace_avg.get(k))
                if dl and len(dl) > 1:              # the same rule on this side: without the kernel's first dispatch
                    ids = sorted((sq_d.get(k) or fe_d.get(k)).items(), key=lambda kv: int(kv[0]))
                    dur = sum(d for _, d in ids[1:]) / (len(ids) - 1)
                flag = ""
                if tavg and dur and trace_share.get(k, 0) >= 0.01 and abs(dur - tavg) > 0.10 * tavg:
                    flag = " **DISAGREES**"
                    disagree.append((k, dur / 1e6, tavg / 1e6))
                fh.write(f"| `{k}` | {len(dl)} | {fmt(dur/1e6 if dur else None)} | {fmt(tavg/1e6 if tavg else None)}{flag} | {fmt(share)} | {fmt(clk)} | "
                         f"{fmt(wa/wc if wa and wc else None)} | {fmt(ldsb/wc if ldsb is not None and wc else None, '{:.4f}')} | "
                         f"{fmt(rd/1e6 if rd is not None else None, '{:.1f}')} | {fmt(wrb/1e6 if wrb is not None else None, '{:.1f}')} | "
                         f"{fmt(hit/(hit+miss) if hit is not None and miss is not None and hit+miss > 0 else None)} |\n")
        traffic["_source"] = f"profiles/{tag}_pmc.md"
        # tie the numbers to the kernel sources they were measured on (the GPU box has no .git): bench.py compares this with the tree it runs
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from faceoff_amd._lib import kernel_source_sha16
        traffic["_kernel_source_sha16"] = kernel_source_sha16()
        with open(os.path.join(out, "pmc_traffic.json" if config == "c2" else f"pmc_traffic_{config}.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    print("summaries written to", out)
    if disagree:
        for k, a, b in disagree:
            print(f"INCONSISTENT: {k}: {a:.3f} ms under --pmc vs {b:.3f} ms in the kernel trace (> 10 %)")
        sys.exit(3)


if __name__ == "__main__":
    main()
