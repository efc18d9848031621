#!/usr/bin/env python3
"""Build-time guard for the kernels whose correctness rests on HAND-COUNTED s_waitcnt vmcnt(N): conv_rgb_dgrad_ring_bf16_kernel counts
exactly its own LDS-DMAs and ONE buffer store per lane and step; conv_halo64_bf16_kernel and the extended-tile kernels count their epilogue stores.  A
scratch spill, or a store the compiler split in two, sits between the counted operations and lets a wave read an LDS slot before its DMA has landed --
nothing at run time would say so except a flaky bit-equality test.  This script compiles csrc/conv_bf16.hip to assembly (device only, ~30 s) and asserts

  * no scratch (private segment 0, no scratch_* instruction) and no VGPR spill in every conv_rgb_dgrad_ring / conv_halo64 / conv_bf16_pph kernel;
  * the ring kernel's VMEM instruction mix: its LDS-DMAs, its stores and its vmcnt waits, as recorded when the counts were last verified by hand.

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
    for b in bad:
        print("check_isa:", b)
    if not bad:
        print("check_isa: ok (no scratch in the counted-vmcnt kernels; ring kernel VMEM mix as verified)")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
