#!/usr/bin/env python3
"""Build-time guard for the kernels whose correctness rests on HAND-COUNTED s_waitcnt vmcnt(N): conv_rgb_dgrad_ring_bf16_kernel counts
exactly its own LDS-DMAs and ONE buffer store per lane and step; conv_halo64_bf16_kernel and the extended-tile kernels count their epilogue stores.  A
scratch spill, or a store the compiler split in two, sits between the counted operations and lets a wave read an LDS slot before its DMA has landed --
nothing at run time would say so except a flaky bit-equality test.  This script compiles csrc/conv_bf16.hip to assembly (device only, ~30 s) and asserts

  * no scratch (private segment 0, no scratch_* instruction) and no VGPR spill in every conv_rgb_dgrad_ring / conv_halo64 / conv_bf16_pph kernel;
  * the ring kernel's VMEM instruction mix: its LDS-DMAs, its stores and its vmcnt waits, as recorded when the counts were last verified by hand;
  * wgrad9_bf16_kernel (csrc/wgrad_bf16.hip, inline-assembly LDS reads and waits): no scratch, and its K-loop holds exactly the hand-written waits.

    python tools/check_isa.py        (`make -C faceoff_amd/csrc isa-check`; tests/test_host_cpu.py runs it)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "faceoff_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernels(asm):
    lines = asm.split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s", l + " ")
        if m and i + 1 < len(lines):
            j = i
            while j < len(lines) and "s_endpgm" not in lines[j]:
                j += 1
            out[m.group(1)] = lines[i:j + 1]
    return out


def mix(body):
    c = collections.Counter()
    for l in body:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("buffer_load") and " lds" in l:
            c["dma"] += 1
        elif op.startswith(("buffer_store", "global_store")):
            c["store"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
    return dict(c)


def main():
    with tempfile.TemporaryDirectory() as td:
        asm_path = os.path.join(td, "conv_bf16.s")
        cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-result", "-S", "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage", "-o", asm_path, os.path.join(CSRC, "conv_bf16.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            return 2
        asm = open(asm_path).read()
    bad = []
    # resource remarks: Function Name ... ScratchSize ... VGPRs Spill
    for blk in r.stderr.split("Function Name: ")[1:]:
        name = blk.split(" ")[0]
        if not any(k in name for k in ("conv_rgb_dgrad_ring", "conv_halo64_bf16", "conv_bf16_pph")):
            continue
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        vspill = int(re.search(r"VGPRs Spill: (\d+)", blk).group(1))
        # the 256 x 256 / 512 x 128 extended-tile kernels keep 5-9 VGPRs in scratch OUTSIDE their counted loops since round 4 (prologue address tables): tolerated up to 16
        limit = 16 if "conv_bf16_pph" in name else 0
        if vspill > limit or (scratch > 4 * limit):
            bad.append(f"{name}: scratch {scratch} B/lane, {vspill} VGPRs spilled (limit {limit})")
    ks = kernels(asm)
    ring = [k for k in ks if "conv_rgb_dgrad_ring_bf16_kernel" in k]
    if len(ring) != 1:
        bad.append(f"expected one conv_rgb_dgrad_ring_bf16_kernel symbol, found {ring}")
    else:
        got = mix(ks[ring[0]])
        want = {"dma": 6, "store": 1}          # static instructions (the per-row DMA loop and the one store of a step are not unrolled)
        if got.get("scratch"):
            bad.append(f"ring kernel uses scratch: {got}")
        if {k: got.get(k, 0) for k in want} != want:
            bad.append(f"ring kernel VMEM mix changed: {got}, verified form {want} -- re-derive its s_waitcnt vmcnt(N) counts by hand (csrc/conv_bf16.hip) and update tools/check_isa.py")
    bad += check_wgrad9()
    for b in bad:
        print("check_isa:", b)
    if not bad:
        print("check_isa: ok (no scratch in the counted-vmcnt kernels; ring kernel VMEM mix as verified; wgrad9 loop waits as counted by hand)")
    return 1 if bad else 0


def check_wgrad9():
    """wgrad9_bf16_kernel (csrc/wgrad_bf16.hip) issues its transposing LDS reads and its waits as inline assembly: the compiler does not know that a fragment
    register is not valid until the wait.  Guarded here: no scratch at all (a spill right behind a read would store a register that has not landed), and the
    K-loop (two K-steps per iteration) holds exactly the waits written by hand -- per K-step vmcnt((AHEAD - 2) NP) in front of the barrier and lgkmcnt 6,
    6 + 2 MA, 6 -- and nothing the compiler added (it drained the DMA ring with vmcnt(0) behind the ds_read_tr builtin; a kernel argument first used inside the
    loop brought an lgkmcnt(0) with it)."""
    bad = []
    with tempfile.TemporaryDirectory() as td:
        asm_path = os.path.join(td, "wgrad_bf16.s")
        cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-result", "-S", "--cuda-device-only",
               "-o", asm_path, os.path.join(CSRC, "wgrad_bf16.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return [r.stderr[-3000:]]
        lines = open(asm_path).read().split("\n")
    names = [l.split(":")[0] for l in lines if re.match(r"^_ZN\S*wgrad9_bf16_kernel\S*:", l)]
    if len(names) != 4:
        bad.append(f"expected four wgrad9_bf16_kernel instantiations, found {len(names)}")
    for name in names:
        st = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[st:en]
        if any(re.match(r"\s+scratch_", l) for l in body) or not any("ScratchSize: 0" in l for l in lines[en:en + 120]):
            bad.append(f"{name}: uses scratch")
        heads = [i for i, l in enumerate(body) if "Loop Header" in l]
        waits, dmas, reads = [], 0, 0
        for l in body[heads[-1]:]:
            t = l.strip()
            if t.startswith("global_store"):
                break
            if t.startswith("s_waitcnt"):
                waits.append(t.replace("s_waitcnt ", ""))
            dmas += t.startswith("buffer_load") and " lds" in t
            reads += t.startswith("ds_read_b64_tr_b16")
        thin = "ILi32ELi128E" in name
        ma, np_ = (2, 4) if thin else (4, 3)
        step = [f"vmcnt({np_})", "lgkmcnt(6)", f"lgkmcnt({6 + 2 * ma})", "lgkmcnt(6)"]
        want = step + step + ["vmcnt(0) lgkmcnt(0)"]
        if waits[:len(want)] != want:
            bad.append(f"{name}: K-loop waits {waits[:len(want)]}, written by hand {want}")
        # (static instructions: a wave's first piece is of P or of Q by its wave number -- two DMA instructions, one executed)
        if dmas != 2 * (np_ + 1) or reads != 2 * (18 + 2 * ma):
            bad.append(f"{name}: K-loop holds {dmas} LDS-DMAs and {reads} transposing reads, expected {2 * (np_ + 1)} and {2 * (18 + 2 * ma)}")
    return bad


if __name__ == "__main__":
    raise SystemExit(main())
