#!/usr/bin/env python3
"""Host model of conv_gen_kernel's tap walk (csrc/conv_gen.hip: 64-row tiles of consecutive (n, d, h, w) positions, per-axis tap trimming per tile)
for the discriminators' forward layers: nominal (every tap) vs executed (the taps a tile walks) vs algorithmic (the (position, tap) pairs whose
input exists) MAC counts -- how much of the executed MFMA work multiplies padding, per layer.   python tools/probes/conv_gen_waste.py [tile_rows]"""
import sys

K, PAD = 4, 2
CH = (64, 128, 256, 512, 1)
STR = (2, 2, 2, 1, 1)


def out(n, s):
    return (n + 2 * PAD - K) // s + 1


def axis_valid(nin, nout, s):
    return [[0 <= o * s + t - PAD < nin for t in range(K)] for o in range(nout)]


def layer(N, sd, s, BM, wfirst=False):
    dd = tuple(out(n, s) if n > 1 else 1 for n in sd)
    ks = tuple(K if n > 1 else 1 for n in sd)
    va = [axis_valid(nin, nout, s) if nin > 1 else [[True]] for nin, nout in zip(sd, dd)]
    rows = [(n, d, h, w) for n in range(N) for d in range(dd[0]) for h in range(dd[1]) for w in range(dd[2])]
    if wfirst:       # alternative row order: w slowest inside a sample -- every row of a tile shares its W (and mostly H) tap validity
        rows = [(n, d, h, w) for n in range(N) for w in range(dd[2]) for h in range(dd[1]) for d in range(dd[0])]
    nominal = len(rows) * ks[0] * ks[1] * ks[2]
    alg = sum(sum(va[0][d]) * sum(va[1][h]) * sum(va[2][w]) for n, d, h, w in rows)
    execd = 0
    for t0 in range(0, len(rows), BM):
        tile = rows[t0:t0 + BM]
        cnt = 1
        for ax in range(3):
            taps = set()
            for r in tile:
                taps |= {t for t, ok in enumerate(va[ax][r[1 + ax]]) if ok}
            cnt *= (max(taps) - min(taps) + 1) if taps else 0
        execd += BM * cnt          # a partial last tile still issues whole MFMAs
    return dd, nominal, execd, alg


def main():
    BM = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for name, sd0 in (("video scale 0", (15, 256, 256)), ("video scale 1", (8, 128, 128)), ("image scale 0", (1, 256, 256)), ("image scale 1", (1, 128, 128))):
        sd = tuple(out(n, 2) if n > 1 else 1 for n in sd0)          # layer 0 (its own space-to-depth form)
        cin = CH[0]
        for j in range(1, 4):
            for wf in (False, True):
                dd, nom, ex, alg = layer(2, sd, STR[j], BM, wf)
                macs = cin * CH[j]
                print(f"{name} layer{j} {cin:3d}->{CH[j]:3d} s{STR[j]} in {sd} out {dd} {'w-first' if wf else 'natural'}: nominal {nom * macs * 2 / 1e9:7.2f} GFLOP, "
                      f"executed {ex * macs * 2 / 1e9:7.2f}, algorithmic {alg * macs * 2 / 1e9:7.2f}  -> executed/algorithmic {ex / alg:.3f}")
            sd, cin = dd, CH[j]


if __name__ == "__main__":
    main()
