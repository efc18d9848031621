#!/usr/bin/env python3
"""One command for the day a node with >= 2 MI355X is available (nothing here can run on the one-GPU development box,
where RCCL refuses two ranks on one device):

    python tools/ddp_selfcheck.py [--gpus N] [--steps K] [--warmup W]

1. the `-m gpu` tests that skip on one device -- two real RCCL ranks against one process on the concatenated batch AND against the
   CPU oracle (tests/test_ddp_gpu.py), `bench.py --gpus 2` launching its own ranks, two fo_comm ranks -- and config 5's data-parallel tests
   (tests/test_gan_gpu.py: two ranks against the per-rank oracle, the fo_comm / RCCL transports);
2. `bench.py --gpus N` for N = 2 .. the number of visible devices (powers of two), and from each line: `comm.ranks_in_group` (max-reduced
   over the process group: what the group REALLY had), `comm.path` / `fo_comm_issued_per_step` (the C-ABI communicator carried every
   bucket + both quantisers' statistics), `comm.exposed_ms` (all-reduce time left behind backward), slowest / fastest rank, and the
   weak-scaling ratio against this script's own N = 1 run; asserted: ranks_in_group == N, fo_comm_issued_per_step == buckets + 2, slowest and fastest
   rank within 3 %, config 5's collectives per iteration.

Every rank is a fresh interpreter started before this process makes any HIP call (`torch.cuda.device_count()` does not initialise the
runtime; bench.py's launch_ranks / faceoff_amd.distributed.launch spawn children, nothing re-execs with a live GPU context).
Reference: distributed/launch.py:22-92, train_faceoff_perceptual.py:164-169."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="largest rank count to bench (default: every visible device)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip-tests", action="store_true")
    args = ap.parse_args()
    import torch
    have = torch.cuda.device_count()                      # (no HIP initialisation)
    if have < 2:
        print(f"ddp_selfcheck: {have} GPU(s) visible -- needs >= 2 (RCCL refuses two ranks on one device); nothing run")
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    rc = 0
    if not args.skip_tests:
        sel = ["tests/test_ddp_gpu.py::test_two_rccl_ranks_on_two_gpus_equal_one_process_on_the_concatenated_batch",
               "tests/test_ddp_gpu.py::test_bench_launches_its_own_two_ranks",
               "tests/test_ddp_gpu.py::test_two_ranks_equal_one_process_on_the_concatenated_batch",
               "tests/test_ddp_gpu.py::test_c_abi_communicator_two_ranks_on_two_gpus",
               # config 5's data-parallel half: two ranks (gloo on one device: the semantics) against the per-rank oracle, and the fo_comm / RCCL transports
               "tests/test_gan_gpu.py::test_two_rank_gan_iterations_vs_oracle",
               "tests/test_gan_gpu.py::test_gan_collectives_in_a_one_rank_world"]
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-s", "-m", "gpu", "-rs", *sel], cwd=ROOT, env=env)
        print(f"ddp_selfcheck: multi-GPU tests rc={r.returncode}")
        rc = rc or r.returncode
    top = min(args.gpus or have, have)
    ns = [1] + [n for n in (2, 4, 8, 16) if n <= top]
    base = None
    for n in ns:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--no-cpu-baseline", "--no-x6-leg", "--no-direct-leg", "--no-h2d-leg"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        if r.returncode != 0 or len(lines) != 1:
            print(f"ddp_selfcheck: bench.py --gpus {n} failed rc={r.returncode}\n{r.stderr[-2000:]}")
            rc = rc or (r.returncode or 1)
            continue
        d = json.loads(lines[0])
        c = d.get("comm") or {}
        if n == 1:
            base = d["value"]
        line = {"n_gpus": d["n_gpus"], "frames_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                "vs_n1": None if not base else round(d["value"] / base, 3),
                "ranks_in_group": c.get("ranks_in_group"), "path": c.get("path"), "fo_comm_issued_per_step": c.get("fo_comm_issued_per_step"),
                "buckets": c.get("buckets"), "allreduce_bytes_per_step": c.get("allreduce_bytes_per_step"), "exposed_ms": c.get("exposed_ms"),
                "ms_per_step_min_rank": c.get("ms_per_step_min_rank"), "ms_per_step_max_rank": c.get("ms_per_step_max_rank"),
                "c3_frames_per_s": (d.get("c3") or {}).get("value"), "c5_frames_per_s": (d.get("c5") or {}).get("value")}
        print("ddp_selfcheck:", json.dumps(line))
        if n > 1:
            ok = c.get("ranks_in_group") == n and c.get("path") == "fo_comm" and c.get("fo_comm_issued_per_step") == (c.get("buckets") or 0) + 2
            if not ok:
                print(f"ddp_selfcheck: bench.py --gpus {n}: the comm block does not show {n} RCCL ranks through fo_comm_* "
                      f"(ranks_in_group {c.get('ranks_in_group')}, path {c.get('path')}, issued per step {c.get('fo_comm_issued_per_step')} vs buckets + 2 = {(c.get('buckets') or 0) + 2})")
                rc = rc or 1
            lo, hi = c.get("ms_per_step_min_rank"), c.get("ms_per_step_max_rank")
            if lo and hi and hi > 1.03 * lo:               # a straggler: every rank waits for it in the all-reduce
                print(f"ddp_selfcheck: bench.py --gpus {n}: slowest rank {hi} ms vs fastest {lo} ms per step (> 3 % apart)")
                rc = rc or 1
            c5 = d.get("c5") or {}
            if c5 and (c5.get("comm") or {}).get("collectives_per_iteration") not in (None, 3.5):
                print(f"ddp_selfcheck: bench.py --gpus {n}: config 5 issued {c5['comm']['collectives_per_iteration']} arena all-reduces + statistics broadcasts "
                      "per iteration, expected (1 + 2 + 2 + 2) / 2 = 3.5")
                rc = rc or 1
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
