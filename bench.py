#!/usr/bin/env python3
"""FaceOff MI355X bench: train frames/s of the VQ-VAE-2 + Conv3d-latent step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or -- when WORLD_SIZE is not set -- this process starts the N ranks itself
(fresh interpreters, before anything here touches the GPU; the reference spawns its ranks the same way:
train_faceoff_perceptual.py:253 -> distributed/launch.py:22-49) and exits with their status.  Fewer than N visible
GPUs is an error, never a silent one-GPU run.

Workload (BASELINE.json configs[1], "C2"): 256x256, T=5, bs=32 clips per GPU (160 frames/GPU/step),
recon + VQ loss, fp32, synthetic inputs resident in HBM, random-init weights.  A step is one full
training iteration: forward, fused losses, backward (all 70 gradients), bucketed RCCL gradient
all-reduce overlapped with backward (N>1), one multi-tensor Adam launch.  Weak scaling: per-GPU work
is fixed.  Rank 0 prints ONE JSON line.  Besides the headline it carries

  roofline              dominant kernel of the timed (Winograd) step, HIP events per launch
  roofline.direct_conv  the same step on the direct implicit-GEMM kernels (no Winograd): step time + its dominant kernel
  c3                    BASELINE configs[2]: the same step + LPIPS/VGG-16 perceptual loss in bf16 (seeded VGG weights)
  c5                    BASELINE configs[4] on one GPU: the GAN iteration, with its own roofline / kernels block
  cpu_baseline          the CPU oracle on the host cores, bounded sample, at 32 threads and at every host core (N = 1 only)
  comm                  (N > 1, or FACEOFF_BENCH_FORCE_DDP=1) the exchange: path (fo_comm = the C-ABI communicator, default), bytes, buckets, exposed time

Per-kernel entries (`roofline.kernel`, `kernels`, `c3.kernels`, `c5.kernels`) are keyed by the kernel SYMBOL as rocprofv3 prints it
(reported by the library: fo_kernel_notes / fo_last_kernel), so a reader can look each one up in profiles/*_kernel_stats.md.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense (same table)
FLOP_PER_FRAME = 64.1e9                # 63.77 GFLOP conv + 0.34 GFLOP VQ distance per frame
VQ_FLOP_PER_FRAME = 0.34e9             # 2 * (32*32 + 64*64) vectors * 64 * 512 at 256x256
LPIPS_FLOP_PER_FRAME = 120.3e9         # 20.04 GMAC x (2 forward + 1 dgrad)
HBM_BW_ACHIEVABLE = 6.29e12            # B/s, measured float4 copy (same guide, chip-level parameters / HBM): the bandwidth a floor is priced at
# the clock the chip HOLDS under each matrix-kernel class (GRBM_GUI_ACTIVE / wall in the committed --pmc passes: profiles/r05_pmc.md 2.19 GHz under the
# fp32 Winograd GEMMs, profiles/r05_c3_pmc.md 1.94-1.96 GHz under the bf16 extended-tile kernels; the nameplate peaks assume 2.4 GHz -- guide, DVFS)
HELD_CLOCK_GHZ = {"f32": 2.19, "bf16": 1.95}


def _cpu_oracle_rate(T, H, W, cores, steps, warm=2):
    """frames/s of the CPU oracle's training step on one clip with `cores` torch-CPU threads"""
    from oracle import faceoff_oracle as O
    from faceoff_amd.synth import make_state_dict, make_batch
    torch.set_num_threads(cores)
    img, gt = make_batch(1, 1, T, H, W)
    img, gt = torch.from_numpy(img), torch.from_numpy(gt)
    p = O.to_torch_state(make_state_dict(0, codebook_scale=0.3, gain=2.0))
    st = {}
    for _ in range(warm):
        O.train_step(img, gt, p, adam_state=st)      # warm-up (oneDNN primitive caches, thread pool)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(img, gt, p, adam_state=st)
    dt = (time.perf_counter() - t0) / steps
    return {"frames_per_s": round(T / dt, 3), "steps": steps, "seconds": round(dt * steps, 1)}


def _cgroup_cpus():
    """CPUs the container may use according to its cgroup quota (None if unlimited / unreadable)"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        return None


def cpu_baseline(T, H, W, steps=20, all_cores_budget_s=40):
    """The CPU oracle (torch-CPU restatement of the reference step, kind "port") timed on this box's host cores on a bounded sample: one
    clip of T frames at HxW, forward + backward + Adam.  BASELINE.md section 3 says "k = all host cores of the box": what the box GIVES
    this process is its cgroup CPU quota (16 CPUs on the pool's boxes, whatever os.cpu_count() says -- 256), so the step is timed at
    the quota's thread count and at 32 threads (rounds 1-3), `value` = the faster.  os.cpu_count() threads are only tried -- in a child
    process, under a wall-clock budget -- where no quota restricts the process: under a 16-CPU quota 256 threads measured 0.043 frames/s
    (116 s per step, gpurun_out/r04d/bench.json), 200x slower than 32."""
    host = os.cpu_count() or 1
    quota = _cgroup_cpus()
    counts = {min(host, 32)}
    if quota is not None and quota >= 1:
        counts.add(max(1, min(host, int(round(quota)))))
    runs = {c: _cpu_oracle_rate(T, H, W, c, steps) for c in sorted(counts)}
    if host not in runs:
        if quota is not None and quota < host:
            runs[host] = {"frames_per_s": None, "note": f"not run: the cgroup quota gives this process {quota:g} CPUs; {host} threads on them measured 0.043 frames/s "
                                                        "in round 4 (gpurun_out/r04d/bench.json: 116 s per step)"}
        else:
            code = (f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; "
                    f"print('CPU_RATE ' + json.dumps(bench._cpu_oracle_rate({T}, {H}, {W}, {host}, 3, warm=1)))")
            try:
                env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")          # a CPU-only child
                res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=all_cores_budget_s, env=env)
                line = [l for l in res.stdout.splitlines() if l.startswith("CPU_RATE ")]
                runs[host] = json.loads(line[0][9:]) if line else {"frames_per_s": None, "note": "child failed: " + res.stderr[-200:]}
            except subprocess.TimeoutExpired:
                runs[host] = {"frames_per_s": None, "note": f"1 warm-up + 3 steps did not finish within {all_cores_budget_s} s"}
    done = {c: v for c, v in runs.items() if v.get("frames_per_s")}
    best = max(done, key=lambda c: done[c]["frames_per_s"])
    return {"value": runs[best]["frames_per_s"], "unit": "frames/s", "cores": best, "host_cores": host, "cgroup_cpu_quota": quota, "kind": "port",
            "by_threads": {str(c): v for c, v in sorted(runs.items())},
            "sample": f"1 clip x {T} frames {H}x{W}, fwd+bwd+Adam, torch-CPU oracle, {steps} timed steps after 2 warm-ups at each of " +
                      ", ".join(f"{c} threads (~{v['seconds']:.0f} s)" for c, v in sorted(done.items())) + "; value = the faster.  frames/s of ONE clip: the "
                      f"{32 * T}-frame step of the GPU workload is not run on the CPU -- its rate is an extrapolation from this sample (the step is linear in clips)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=32, help="clips per GPU (BASELINE: 32)")
    ap.add_argument("--frames", type=int, default=5, help="frames per clip T (BASELINE: 5)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--perceptual", action="store_true",
                    help="make BASELINE config 3 (C2 + LPIPS/VGG-16 term, --lpips-dtype) the timed workload; "
                         "the headline metric is config 2, which reports config 3 under the key `c3` anyway")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the short config-3 leg")
    ap.add_argument("--no-x6-leg", action="store_true", help="skip the leg with the Winograd-domain GEMMs on the bf16 matrix pipe")
    ap.add_argument("--no-direct-leg", action="store_true", help="skip the short direct-convolution leg")
    ap.add_argument("--no-c5", action="store_true", help="skip the short config-5 (GAN iteration) leg")
    ap.add_argument("--c3-sustained", type=int, default=100,
                    help="steps of the sustained config-3 measurement (c3.ms_per_step_sustained: one timed region of this many steps right after "
                         "the c3 leg's own --steps; 0 = skip)")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the short host-fed leg")
    ap.add_argument("--lpips-dtype", choices=("bf16", "fp32"), default="bf16",
                    help="arithmetic of the LPIPS / VGG-16 branch (BASELINE config 3 = bf16)")
    ap.add_argument("--vqvae-dtype", choices=("fp32", "bf16"), default="fp32",
                    help="arithmetic of the VQ-VAE convolutions in the TIMED workload (default fp32 = BASELINE config 2; with --perceptual, "
                         "bf16 makes the timed workload config 3 as the `c3` leg runs it: what profiles/collect_extra.sh traces)")
    ap.add_argument("--direct-conv", action="store_true",
                    help="run the Conv3d / 3x3 128->128 layers on the direct implicit-GEMM kernels instead of Winograd "
                         "(the kernel-quality reference: same results to fp32 rounding, 1.5x the step time)")
    ap.add_argument("--comm", choices=("abi", "torch"), default=None,
                    help="multi-rank gradient / VQ-statistics exchange: `abi` = the C-ABI communicator (fo_comm_* over RCCL, csrc/comm.cpp; "
                         "torch.distributed only does the rendezvous and the bench's own barriers), `torch` = torch.distributed all-reduces. "
                         "Default: abi whenever the process group is RCCL, torch over gloo (tests sharing one GPU)")
    ap.add_argument("--host-cpus", type=int, default=0,
                    help="rehearsal of the 8-rank host budget on one GPU (VERDICT r05 item 5): confine this process to the first K CPUs it may use "
                         "(os.sched_setaffinity, before any GPU call) and run torch with one intra-op thread -- on the pool a rank of an 8-GPU job "
                         "gets ~2 of the 16 CPUs the cgroup quota allows.  The CPU baseline is skipped (it would time 2 CPUs)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP-event timing")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the step without side-stream overlap (what profiles/collect.sh traces: kernels run alone)")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """Start n ranks of this script (one per GPU) and return their worst exit status.  Runs BEFORE this process has
    made any HIP call: torch.cuda.device_count() does not initialise the runtime, and the children are fresh
    interpreters, so nothing re-execs or forks with a live GPU context."""
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write(f"bench.py --gpus {n}: only {have} GPU(s) visible -- refusing to fall back to fewer\n")
        return 2
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        # rank 0 owns stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                for q in alive:               # one rank failed: the others would hang in a collective
                    q.terminate()
    return rc


def main():
    args = parse_args()
    if args.host_cpus > 0:             # before anything touches the GPU (and before the ranks are spawned: children inherit the mask)
        os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:args.host_cpus]))
        torch.set_num_threads(1)
        args.no_cpu_baseline = True
    if args.direct_conv:
        os.environ["FACEOFF_NO_WINOGRAD"] = "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # FACEOFF_BENCH_SHARE_GPU=1 + FACEOFF_BENCH_BACKEND=gloo (tests on a one-GPU box): every rank on device 0, collectives over
    # gloo -- the multi-rank control flow of this script (barriers, max over ranks, the legs, rank 0's JSON line) without RCCL
    share_gpu = bool(os.environ.get("FACEOFF_BENCH_SHARE_GPU"))
    backend = os.environ.get("FACEOFF_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: LOCAL_RANK={local_rank} but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FACEOFF_BENCH_FORCE_DDP=1 (diagnostic): take the multi-GPU code path -- RCCL process group, bucketed gradient
    # all-reduce on the side stream, VQ-statistics all-reduce, barrier, max-over-ranks -- with however many ranks there
    # are, including one (the GPU box used for development has a single device)
    ddp = world > 1 or bool(os.environ.get("FACEOFF_BENCH_FORCE_DDP"))
    # The contract is ONE JSON line on stdout.  RCCL prints its version banner to stdout, so everything any library
    # writes to fd 1 from here on goes to stderr; the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    from faceoff_amd import ops
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.synth import make_state_dict
    from faceoff_amd.trainer import FaceOffTrainer

    # the gradient / VQ-statistics exchange ("Collectives: fo_comm_{init,allreduce_async,wait,destroy} over RCCL, ncclUniqueId
    # exchanged through the existing TCP dist_url"): by default the C-ABI communicator; torch.distributed hands rank 0's id to the others
    comm_path = args.comm or ("abi" if backend == "nccl" else "torch")
    abi_comm = None
    if ddp and comm_path == "abi":
        if backend != "nccl":
            raise SystemExit("bench.py --comm abi needs one GPU per rank (RCCL): FACEOFF_BENCH_BACKEND=gloo shares a device between ranks")
        from faceoff_amd.distributed.comm import AbiComm, exchange_via_torch_store
        abi_comm = AbiComm.create(rank if world > 1 else 0, world, dev, exchange_via_torch_store())

    B, T, H = args.clips, args.frames, args.size
    frames = B * T
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    img = torch.rand((frames, 6, H, H), device=dev, generator=gen) * 2 - 1       # U(-1,1): dataset.py:240-247
    gt = torch.rand((frames, 3, H, H), device=dev, generator=gen) * 2 - 1

    def sync():
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not ddp:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return t.item()

    def min_over_ranks(x):
        return -max_over_ranks(-x)

    def make_trainer(winograd=True, perceptual=False, dtype="fp32"):
        eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev, dtype=dtype)
        if not winograd:
            eng.winograd = False
        vqlpips = None
        if perceptual:
            from faceoff_amd.loss import VQLPIPS
            from faceoff_amd.synth import make_vgg_lpips_state
            vqlpips = VQLPIPS(make_vgg_lpips_state(7), dtype=args.lpips_dtype).to(dev)
        tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=vqlpips, force_collectives=ddp and world == 1, comm=abi_comm)
        if args.serial_streams:
            eng.set_stream_overlap(False)
        return eng, tr

    def timed(tr, steps, warmup):
        """W untimed + K timed steps bracketed by barrier + synchronize on both sides; max over ranks.
        Returns (seconds for the K steps, HIP-event ms per step on this rank, last losses)."""
        for _ in range(warmup):
            tr.step(img, gt, T=T)
        if tr.reducer is not None:
            tr.reducer.time_exposed = True
        sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            losses = tr.step(img, gt, T=T)
        ev1.record()                      # every side stream is joined into this one at the end of a step
        sync()
        dt = time.perf_counter() - t0
        if tr.reducer is not None:
            tr.reducer.time_exposed = False
        timed.last_dt_this_rank = dt
        return max_over_ranks(dt), ev0.elapsed_time(ev1) / steps, losses

    def per_kernel(eng, tr, steps):
        """The same steps again with the side streams folded into the main one, so that every launch has the GPU to
        itself (kernels sharing the chip stretch each other's event-to-event time); HIP events bracket each launch on
        the stream it is launched on.  Returns (KernelProfiler summary, serial ms per step)."""
        eng.set_stream_overlap(False)
        tr.step(img, gt, T=T)
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        sync()
        t1 = time.perf_counter()
        for _ in range(steps):
            tr.step(img, gt, T=T)
        sync()
        ms_serial = (time.perf_counter() - t1) / steps * 1e3
        ops.PROFILER = None
        prof.close()
        eng.set_stream_overlap(not args.serial_streams)
        return prof.summary(), ms_serial

    def is_bf16_kernel(name):          # (kernel symbols as rocprofv3 prints them: the library reports them, ops.KernelProfiler)
        return "bf16" in name

    def step_floor(run_step, steps, ms_step, fold):
        """The step's ATTAINABLE FLOOR given each launch's own bound: `steps` more steps with the side streams folded and an
        ops.StepLedger on every C-ABI call; per launch max(algorithmic bytes / 6.29 TB/s, algorithmic FLOP / (dense peak of its type x held clock /
        2.4 GHz)), summed.  `step_floor_ms` lets no two launches overlap (each at its own bound, back to back); `step_floor_perfect_overlap_ms` =
        max(sum of the HBM times, sum of the matrix times) is what perfect overlap of memory-bound beside matrix-bound launches could reach at best.
        Both are floors of THIS design's launches (Winograd-domain GEMMs counted as the GEMMs they are, transforms as the bytes they move)."""
        fold(True)
        run_step()
        led = ops.StepLedger().open()
        try:
            sync()
            for _ in range(steps):
                run_step()
            sync()
        finally:
            led.close()
            fold(False)
        f = led.floor(steps, HBM_BW_ACHIEVABLE, lambda k: BF16_MFMA_PEAK_TFLOPS if is_bf16_kernel(k) else FP32_MFMA_PEAK_TFLOPS,
                      lambda k: HELD_CLOCK_GHZ["bf16" if is_bf16_kernel(k) else "f32"])
        f["step_frac_of_floor"] = round(f["step_floor_ms"] / ms_step, 4)
        f["priced_at"] = {"hbm_TBps": HBM_BW_ACHIEVABLE / 1e12, "fp32_mfma_tflops": FP32_MFMA_PEAK_TFLOPS, "bf16_mfma_tflops": BF16_MFMA_PEAK_TFLOPS,
                          "held_clock_ghz": HELD_CLOCK_GHZ, "nameplate_clock_ghz": 2.4}
        f["what"] = ("sum over every C-ABI launch of one step of max(bytes / HBM, FLOP / (peak x held clock / 2.4)); bytes = the distinct tensors handed to the "
                     "launch, each once, scratch excluded (errs low); step_frac_of_floor = step_floor_ms / the timed ms per step (1.0 = every launch at its "
                     "own bound, no overlap); largest_gaps: kernels by measured (launch alone on the GPU) minus floor; floor_above_measured_ms: launches whose floor "
                     "exceeds their own measured time, summed -- the byte model counting too much (0 = it never does)")
        return f

    def dominant(summ, steps, ms_serial):
        dom = max(summ, key=lambda k: summ[k]["total_ms"])
        d = summ[dom]
        peak = BF16_MFMA_PEAK_TFLOPS if is_bf16_kernel(dom) else FP32_MFMA_PEAK_TFLOPS
        return {"bound": "mfma", "kernel": dom, "kernel_is": "the symbol rocprofv3 --kernel-trace prints for these launches (profiles/*_kernel_stats.md)",
                "achieved": round(d["tflops"], 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(d["tflops"] / peak, 4), "traffic": None,
                "achieved_padded_taps_counted": round(d["tflops_nominal"], 2),
                "launches_per_step": d["launches"] / steps, "avg_launch_ms": round(d["avg_ms"], 4),
                "algorithmic_gflop_per_launch": round(d["flops_per_launch"] / 1e9, 3),
                "share_of_step_time": round(d["total_ms"] / (ms_serial * steps), 4)}

    # every leg below honours --steps / --warmup (they ran 5-step bursts after 1-2 warm-ups of a fresh engine, whose
    # Winograd filter banks and workspaces are made lazily -- the host-fed leg came out "faster than resident").  At least 3 warm-ups
    # after every fresh engine.
    k_leg, w_leg = max(2, args.steps), max(3, args.warmup)

    # ------------------------------------------------------------------ the timed workload
    eng, trainer = make_trainer(winograd=not args.direct_conv, perceptual=args.perceptual, dtype=args.vqvae_dtype)
    issued0 = abi_comm.issued if abi_comm is not None else 0
    dt, ms_events, (recon, latent, _) = timed(trainer, args.steps, args.warmup)
    comm = None
    if ddp:
        # Self-proving multi-GPU line: how many ranks the process group REALLY had (max-reduced over it), what moved per step,
        # how much of the gradient all-reduce was left exposed behind backward, and the slowest / fastest rank's step time.
        red = trainer.reducer
        ranks = int(max_over_ranks(float(torch.distributed.get_world_size())))
        exposed = red.exposed_ms() if red is not None else None
        my_ms = timed.last_dt_this_rank / args.steps * 1e3
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            rccl = None
        desc = red.describe() if red is not None else {"buckets": 0, "bucket_bytes": [], "grad_bytes_per_step": 0}
        vq_bytes = 2 * (512 + 512 * 64) * 4                       # one fused [512] + [64,512] message per quantiser (:63-64)
        comm = {"path": "fo_comm" if abi_comm is not None else "torch.distributed",
                "path_what": ("gradient buckets and VQ statistics through the C-ABI communicator (fo_comm_allreduce_async / fo_comm_wait, csrc/comm.cpp: RCCL on the "
                              "communicator's own stream); torch.distributed did the rendezvous (rank 0's ncclUniqueId) and this script's barriers"
                              if abi_comm is not None else "torch.distributed all-reduces on the reducer's side stream"),
                "fo_comm_issued_per_step": None if abi_comm is None else round((abi_comm.issued - issued0) / float(args.steps + args.warmup), 2),
                "ranks_in_group": ranks, "backend": torch.distributed.get_backend(), "rccl_version": rccl,
                "collectives_forced_in_one_rank_group": bool(world == 1),
                "allreduce_bytes_per_step": desc["grad_bytes_per_step"] + vq_bytes,
                "grad_allreduce_bytes_per_step": desc["grad_bytes_per_step"], "vq_stats_allreduce_bytes_per_step": vq_bytes,
                "buckets": desc["buckets"], "bucket_bytes": desc["bucket_bytes"],
                "exposed_ms": None if exposed is None else round(max_over_ranks(exposed), 4),
                "exposed_ms_what": "HIP events on the compute stream around its wait for the all-reduce side stream in reducer.finish(), mean over the timed steps, max over ranks",
                "ms_per_step_min_rank": round(min_over_ranks(my_ms), 3), "ms_per_step_max_rank": round(max_over_ranks(my_ms), 3)}
    summ = ms_serial = floor_main = None
    if not args.no_kernel_events:
        summ, ms_serial = per_kernel(eng, trainer, args.steps)
        floor_main = step_floor(lambda: trainer.step(img, gt, T=T), min(args.steps, 3), dt / args.steps * 1e3,
                                lambda on, e=eng: e.set_stream_overlap(not on and not args.serial_streams))
    winograd_on = eng.winograd
    wino_max_tile = eng.winograd_max_tile
    del trainer, eng
    torch.cuda.empty_cache()

    ms = dt / args.steps * 1e3
    fps = world * frames * args.steps / dt
    out = {
        "metric": "train frames/sec (256x256, T=5, bs=32/GPU)", "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world if not share_gpu else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "ms_per_step_hip_events": round(ms_events, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic U(-1,1) inputs resident in HBM; random-init weights (kaiming-uniform x2 gain, codebook N(0,0.3^2))",
        "config": {"workload": (f"C2: VQ-VAE-2 + Conv3d latent, {H}x{H}, T={T}, {B} clips/GPU, recon+VQ loss, fwd+bwd+Adam"
                                if not args.perceptual else
                                f"C3: C2 (fp32) + LPIPS/VGG-16 perceptual loss in {args.lpips_dtype} (seeded VGG weights), {H}x{H}, T={T}, {B} clips/GPU"),
                   "global_clips": B * world, "frames_per_step": frames * world, "parallelism": f"dp{world}" + (" (ranks sharing one GPU over gloo: a control-flow test, not a measurement)" if share_gpu else ""),
                   "conv3d_algorithm": ("winograd (fwd, dgrad, wgrad), F(4x4,3x3) where a plane is whole GEMM tiles else F(2x2,3x3)"
                                        if winograd_on else "direct")},
        "loss": {"recon": round(recon.item(), 6), "latent": round(latent.item(), 6)},
    }
    if args.host_cpus > 0:
        out["host"] = {"confined_to_cpus": sorted(os.sched_getaffinity(0)), "torch_threads": torch.get_num_threads(),
                       "what": "--host-cpus: the whole process (launch thread, torch's helper threads, RCCL proxies if any) shares these CPUs"}
    if comm is not None:
        out["comm"] = comm
    if args.perceptual:
        out["dtype"] = "f32 (VQ-VAE) + %s (LPIPS)" % ("bf16" if args.lpips_dtype == "bf16" else "f32")
    if args.vqvae_dtype == "bf16":
        out["dtype"] = "bf16 operands / fp32 accumulate & master weights (VQ-VAE convolutions%s); VQ, losses, Adam fp32" % (
            " and LPIPS" if args.perceptual and args.lpips_dtype == "bf16" else "")
        out["config"]["workload"] = out["config"]["workload"].replace("C2 (fp32)", "C2").replace("C2:", "C2 on bf16 MFMA operands:")
    if ops.BF16X6:       # FACEOFF_BF16X6=1 for the whole run: say so in the line (the default run reports this path as the `bf16x6` leg)
        out["dtype"] += " [Winograd-domain GEMMs: each fp32 product as 6 bf16 MFMA partial products of an exact 3-way split, fp32 accumulate]"
        out["config"]["winograd_gemm_arithmetic"] = "bf16x6 (FACEOFF_BF16X6=1)"
    # Speed against an IDEAL direct-convolution implementation (SURVEY 8(d)'s FLOP count at the dense MFMA peak of the
    # type each part runs in).  NOT a roofline fraction: Winograd executes 2.25-4x fewer multiplies on the Conv3d and
    # 3x3 128->128 layers, so this ratio exceeds 1 when the step is faster than any direct convolution could be.
    ideal_s = FLOP_PER_FRAME / (FP32_MFMA_PEAK_TFLOPS * 1e12)
    if args.perceptual:
        ideal_s += LPIPS_FLOP_PER_FRAME / ((BF16_MFMA_PEAK_TFLOPS if args.lpips_dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS) * 1e12)
    out["speed_vs_ideal_direct_conv"] = round(ideal_s * fps / world, 4)
    if summ is not None:
        out["roofline"] = dominant(summ, args.steps, ms_serial)
        out["roofline"]["measured"] = ("HIP events per launch over a second K-step region with the side streams joined "
                                       "(kernels run alone; ms_per_step_serial is that region's step time incl. event overhead)")
        out["ms_per_step_serial"] = round(ms_serial, 3)
        out["step_floor"] = floor_main
        out["kernels"] = {k: {"launches_per_step": v["launches"] / args.steps, "avg_ms": round(v["avg_ms"], 4),
                              "tflops": round(v["tflops"], 2), "tflops_padded_taps_counted": round(v["tflops_nominal"], 2),
                              "ms_per_step": round(v["total_ms"] / args.steps, 3)}
                          for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        # whole-step matrix-pipe fraction from what the launches really execute: sum of every profiled launch's FLOP
        # (Winograd-domain GEMMs counted as the GEMMs they are, clip-padding taps excluded) + the VQ distance GEMM
        fp32_flop = sum(v["flops_per_launch"] * v["launches"] for k, v in summ.items() if not is_bf16_kernel(k)) / args.steps
        fp32_flop += VQ_FLOP_PER_FRAME * frames
        out["executed_matrix_tflop_per_step"] = round(fp32_flop / 1e12, 4)
        if not args.perceptual:
            out["step_frac_executed_flop"] = round(fp32_flop / (ms * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # written by profiles/collect.sh from rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                tr_json = json.load(open(pmc))
                out["roofline"]["traffic"] = tr_json.get(out["roofline"]["kernel"])
                from faceoff_amd._lib import kernel_source_sha16
                stamp = tr_json.get("_kernel_source_sha16")
                out["roofline"]["traffic_source"] = ("HBM bytes per launch from the committed rocprofv3 --pmc passes "
                                                     f"({tr_json.get('_source', 'profiles/pmc_traffic.json')}), not re-measured in this run; "
                                                     f"measured on kernel sources {stamp or 'unstamped (round 2)'}" +
                                                     (f", this run's sources {kernel_source_sha16()}" if stamp != kernel_source_sha16() else " = this run's sources"))
            except Exception:
                pass

    # ------------------------------------------------------------------ the same step on direct convolutions
    if winograd_on and not args.no_direct_leg and not args.no_kernel_events and not args.perceptual and args.vqvae_dtype == "fp32":
        eng_d, tr_d = make_trainer(winograd=False)
        k_d = k_leg
        dt_d, _, _ = timed(tr_d, k_d, w_leg)
        summ_d, ms_serial_d = per_kernel(eng_d, tr_d, k_d)
        dd = dominant(summ_d, k_d, ms_serial_d)
        dd.pop("traffic")
        dd.update(ms_per_step=round(dt_d / k_d * 1e3, 3), steps=k_d,
                  frames_per_s=round(world * frames * k_d / dt_d, 2),
                  note="Conv3d and 3x3 128->128 layers on conv_igemm3 / conv_igemm instead of Winograd, same process")
        # fraction of the fp32 MFMA peak over the whole step from what the launches EXECUTE: clip-padding taps excluded (2 of 15 of the Conv3d
        # work at T = 5), the stride-2 stems -- still on F(4x4,2x2) in this leg -- counted as the GEMMs they run; + the VQ distance GEMMs
        exec_d = sum(v["flops_per_launch"] * v["launches"] for v in summ_d.values()) / k_d + VQ_FLOP_PER_FRAME * frames
        dd["executed_matrix_tflop_per_step"] = round(exec_d / 1e12, 4)
        dd["step_frac_executed_flop"] = round(exec_d / (dt_d / k_d) / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
        dd["step_frac_nominal_direct_conv_flop"] = round(FLOP_PER_FRAME / (FP32_MFMA_PEAK_TFLOPS * 1e12) * frames * k_d / dt_d, 4)
        dd["step_frac_nominal_what"] = ("SURVEY 8(d)'s 64.1 GFLOP/frame (every tap of every layer as a direct convolution, clip-padding taps included) over this "
                                        "step time and the peak: NOT executed work -- it overstates by the padded taps and the stems' Winograd savings")
        out["roofline"]["direct_conv"] = dd
        del eng_d, tr_d
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ the same step, fp32 products on the bf16 matrix pipe
    # (opt-in FACEOFF_BF16X6=1; `value` above is the fp32-MFMA path unless that variable is set for the whole run)
    if winograd_on and not args.no_x6_leg and not args.perceptual and not ops.BF16X6 and args.vqvae_dtype == "fp32":
        ops.BF16X6 = True
        eng_x, tr_x = make_trainer(winograd=True)
        k_x = k_leg
        dt_x, _, (r_x, l_x, _) = timed(tr_x, k_x, w_leg)
        kern_x = None
        if not args.no_kernel_events:
            summ_x, _ = per_kernel(eng_x, tr_x, k_x)
            kern_x = {}
            for name in sorted(summ_x):
                if name.startswith(("wino_gemm_split", "wino_wgrad_split")):
                    v = summ_x[name]
                    # 6 bf16 MFMA FLOP are executed per algorithmic fp32 FLOP: the matrix-pipe roofline of these launches is the bf16 peak
                    kern_x[name] = {"launches_per_step": v["launches"] / k_x, "avg_ms": round(v["avg_ms"], 4),
                                    "fp32_equivalent_tflops": round(v["tflops"], 2), "executed_bf16_tflops": round(6 * v["tflops"], 1),
                                    "peak_bf16_tflops": BF16_MFMA_PEAK_TFLOPS, "frac_of_bf16_peak": round(6 * v["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4),
                                    "ms_per_step": round(v["total_ms"] / k_x, 3)}
        ops.BF16X6 = False
        out["bf16x6"] = {
            "what": ("the Winograd-domain GEMMs -- forward, data gradient (csrc/wino_gemm_split.hip) and filter gradient "
                     "(csrc/wino_wgrad_split.hip) -- form each fp32 product as six bf16 MFMA partial products of an exact three-way "
                     "bf16 split of both operands, accumulated in fp32: relative error 2^-23 per product (tests/test_split_gpu.py: "
                     "closer to an fp64 product than the fp32 MFMA kernels). Everything else unchanged."),
            "value": round(world * frames * k_x / dt_x, 2), "unit": "frames/s", "ms_per_step": round(dt_x / k_x * 1e3, 3), "steps": k_x, "warmup": w_leg,
            "loss": {"recon": round(r_x.item(), 6), "latent": round(l_x.item(), 6)}}
        if kern_x:
            out["bf16x6"]["kernels"] = kern_x
        del eng_x, tr_x
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ BASELINE config 3: + LPIPS, bf16 MFMA operands throughout
    # ("C3: + LPIPS, bf16 MFMA inputs / fp32 accumulate & master weights": the VQ-VAE's own convolutions on the bf16 matrix
    # pipe as well -- engine dtype="bf16": fp32 master weights, fp32 accumulation, fp32 VQ; parity against the oracle with the same
    # rounding points, tests/test_bf16_engine_gpu.py -- next to the same step with the VQ-VAE left in fp32, `c3.fp32_vqvae`)
    if not args.no_c3 and not args.perceptual and not args.direct_conv:
        ideal_c = lambda vq_peak: FLOP_PER_FRAME / (vq_peak * 1e12) + LPIPS_FLOP_PER_FRAME / (
            (BF16_MFMA_PEAK_TFLOPS if args.lpips_dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS) * 1e12)
        k_c = k_leg
        eng_c, tr_c = make_trainer(perceptual=True, dtype="bf16")
        dt_c, _, (r_c, l_c, p_c) = timed(tr_c, k_c, w_leg)
        fps_c = world * frames * k_c / dt_c
        out["c3"] = {"workload": f"C3: C2 + LPIPS/VGG-16 perceptual loss, bf16 MFMA operands / fp32 accumulate & master weights for the VQ-VAE and the LPIPS "
                                 f"branch alike (fp32 VQ, losses, Adam; seeded VGG weights), {H}x{H}, T={T}, {B} clips/GPU",
                     "value": round(fps_c, 2), "unit": "frames/s", "ms_per_step": round(dt_c / k_c * 1e3, 3), "steps": k_c, "warmup": w_leg,
                     "dtype": "bf16 (VQ-VAE convolutions and LPIPS: bf16 operands, fp32 accumulate; VQ, losses, master weights, Adam fp32)",
                     "speed_vs_ideal_direct_conv": round(ideal_c(BF16_MFMA_PEAK_TFLOPS) * fps_c / world, 4),
                     "loss": {"recon": round(r_c.item(), 6), "latent": round(l_c.item(), 6), "perceptual": round(p_c.item(), 6)}}
        if args.c3_sustained > 0:
            # a --steps burst right after the warm-ups runs at the clock the chip still holds; over 100+ steps it settles ~2 % lower
            # (tools/soak_c3.py: 35.6 ms sustained where the 5-step burst of round 4 printed 34.9): both are on the line
            dt_s, _, _ = timed(tr_c, args.c3_sustained, 0)
            out["c3"]["ms_per_step_sustained"] = round(dt_s / args.c3_sustained * 1e3, 3)
            out["c3"]["value_sustained"] = round(world * frames * args.c3_sustained / dt_s, 2)
            out["c3"]["sustained_steps"] = args.c3_sustained
        if not args.no_kernel_events:
            summ_c, ms_serial_c = per_kernel(eng_c, tr_c, k_c)
            rc = dominant(summ_c, k_c, ms_serial_c)
            rc["measured"] = "HIP events per launch, side streams joined (as `roofline`); algorithmic FLOP of the launches (clip-padding taps excluded) against the dense bf16 MFMA peak"
            rc["ms_per_step_serial"] = round(ms_serial_c, 3)
            pmc_c = os.path.join(ROOT, "profiles", "pmc_traffic_c3.json")        # CONFIG=c3 bash profiles/collect.sh <tag>: the same counters on this leg's command
            if os.path.exists(pmc_c):
                try:
                    tj = json.load(open(pmc_c))
                    from faceoff_amd._lib import kernel_source_sha16
                    rc["traffic"] = tj.get(rc["kernel"])
                    rc["traffic_source"] = (f"HBM bytes per launch from the committed rocprofv3 --pmc passes ({tj.get('_source')}), measured on kernel sources "
                                            f"{tj.get('_kernel_source_sha16')}" + (" = this run's sources" if tj.get("_kernel_source_sha16") == kernel_source_sha16()
                                                                                    else f", this run's sources {kernel_source_sha16()}"))
                except Exception:
                    pass
            out["c3"]["roofline"] = rc
            out["c3"]["kernels"] = {k: {"launches_per_step": v["launches"] / k_c, "avg_ms": round(v["avg_ms"], 4), "tflops": round(v["tflops"], 1),
                                        "frac_of_bf16_peak": round(v["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4), "ms_per_step": round(v["total_ms"] / k_c, 3)}
                                    for k, v in sorted(summ_c.items(), key=lambda kv: -kv[1]["total_ms"])[:12]}
            bf16_flop = sum(v["flops_per_launch"] * v["launches"] for k, v in summ_c.items()) / k_c
            out["c3"]["executed_matrix_tflop_per_step"] = round(bf16_flop / 1e12, 3)
            out["c3"]["step_frac_executed_flop_of_bf16_peak"] = round(bf16_flop / (dt_c / k_c) / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4)
            out["c3"]["step_floor"] = step_floor(lambda: tr_c.step(img, gt, T=T), min(k_c, 3), out["c3"].get("ms_per_step_sustained", dt_c / k_c * 1e3),
                                                 lambda on, e=eng_c: e.set_stream_overlap(not on and not args.serial_streams))
        del eng_c, tr_c
        torch.cuda.empty_cache()
        # the bf16 VQ-VAE step alone (recon + VQ loss, no LPIPS): what the bf16 matrix pipe does to config 2's workload
        eng_b, tr_b = make_trainer(dtype="bf16")
        dt_b, _, (r_b, l_b, _) = timed(tr_b, k_c, w_leg)
        out["c3"]["vqvae_only_bf16"] = {"value": round(world * frames * k_c / dt_b, 2), "unit": "frames/s", "ms_per_step": round(dt_b / k_c * 1e3, 3),
                                        "loss": {"recon": round(r_b.item(), 6), "latent": round(l_b.item(), 6)},
                                        "note": "config 2's step with bf16 MFMA operands (no LPIPS); `value` of this line stays the fp32 step"}
        del eng_b, tr_b
        torch.cuda.empty_cache()
        # round 2's form of config 3: fp32 VQ-VAE + bf16 LPIPS
        eng_f, tr_f = make_trainer(winograd=True, perceptual=True)
        dt_f, _, (r_f, l_f, p_f) = timed(tr_f, k_c, w_leg)
        out["c3"]["fp32_vqvae"] = {"value": round(world * frames * k_c / dt_f, 2), "unit": "frames/s", "ms_per_step": round(dt_f / k_c * 1e3, 3),
                                   "dtype": "f32 (VQ-VAE) + %s (LPIPS)" % args.lpips_dtype,
                                   "loss": {"recon": round(r_f.item(), 6), "latent": round(l_f.item(), 6), "perceptual": round(p_f.item(), 6)}}
        del eng_f, tr_f
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ the headline step fed from host memory (SURVEY 8 f4 "feeding")
    # utils.py:29-38 moves a clip to the GPU inside the step; here the loader's 5-tuples (pinned host memory, as DataLoader(pin_memory=True)
    # yields them) are copied by faceoff_amd.feeder.HostFedBatches on a copy stream beside the previous step: 377 MB per step over PCIe
    if not args.no_h2d_leg and not args.perceptual and not args.direct_conv:
        eng_h, tr_h = make_trainer(winograd=True)
        k_h = k_leg
        gcpu = torch.Generator().manual_seed(4321 + rank)
        host = [tuple((torch.rand((B, T, 3, H, H), generator=gcpu) * 2 - 1).pin_memory() if i != 4 else torch.empty(0) for i in range(5)) for _ in range(2)]
        loader = [host[i % 2] for i in range(w_leg + k_h)]
        sync()
        it = tr_h.run_host_fed(loader)
        for _ in range(w_leg):
            next(it)
        sync()
        t0 = time.perf_counter()
        for _ in range(k_h):
            r_h, l_h, _, _ = next(it)
        sync()
        dt_h = max_over_ranks(time.perf_counter() - t0)
        it.close()
        # the same trainer, same process state, with its inputs resident: the ratio of the two is the cost of feeding
        dt_r, _, _ = timed(tr_h, k_h, w_leg)
        out["h2d_fed"] = {"what": "the headline step with every batch copied from pinned host memory (source, background, source_images: "
                                  f"{3 * frames * 3 * H * H * 4 / 1e6:.0f} MB per step) by a double-buffered copy stream beside the previous step; `value` keeps inputs resident",
                          "value": round(world * frames * k_h / dt_h, 2), "unit": "frames/s", "ms_per_step": round(dt_h / k_h * 1e3, 3), "steps": k_h, "warmup": w_leg,
                          "resident_ms_per_step_same_trainer": round(dt_r / k_h * 1e3, 3), "slowdown_vs_resident": round(dt_h / dt_r - 1.0, 4),
                          "loss": {"recon": round(r_h.item(), 6), "latent": round(l_h.item(), 6)}}
        del eng_h, tr_h, host, loader
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ BASELINE config 5: the two-optimiser GAN iteration
    if not args.no_c5 and not args.perceptual and not args.direct_conv:
        from faceoff_amd.disc import DiscEngine
        from faceoff_amd.gan_trainer import GANTrainer
        from faceoff_amd.synth import make_disc_state
        import random as _random
        clip, win = 30, 16            # one clip of up to 30 frames per iteration (TemporalAlignmentDataset('train', 30)), 16-frame window
        eng_g = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
        gan = GANTrainer(eng_g, DiscEngine(make_disc_state(1, 3), dev, dims=3, n_frames=win - 1), DiscEngine(make_disc_state(2, 2), dev, dims=2),
                         window=win, rng=_random.Random(7 + rank), comm=abi_comm, force_collectives=ddp and world == 1)
        cimg, cgt = img[:clip].contiguous(), gt[:clip].contiguous()
        iters = 2 * ((k_leg + 1) // 2)          # generator and discriminator iterations alternate: an even count, --steps of them
        for _ in range(2 * ((w_leg + 1) // 2)):
            gan.step(cimg, cgt)
        sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            o5 = gan.step(cimg, cgt)
        sync()
        dt5 = max_over_ranks(time.perf_counter() - t0)
        out["c5"] = {"workload": f"C5: GAN iteration of disc_trainers/train_vqvae_mocoganhd_disc.py on one {clip}-frame clip per GPU, {H}x{H}: "
                                 f"VQ-VAE generator + MoCoGAN-HD video (15 frame pairs) and image discriminators, RaLSGAN, alternating G / D updates",
                     "value": round(world * clip * iters / dt5, 2), "unit": "frames/s", "ms_per_iteration": round(dt5 / iters * 1e3, 3),
                     "iterations": iters, "warmup": 2 * ((w_leg + 1) // 2), "dtype": "f32", "loss": {k: round(v.item(), 6) for k, v in o5.items()}}
        if gan.collectives:         # data parallel: per iteration (1 + 2) [generator: G arena, two running-statistics broadcasts] or (2 + 2) [discriminators]
            out["c5"]["comm"] = {"collectives_per_iteration": round(gan.collectives_issued / float(iters + 2 * ((w_leg + 1) // 2)), 3),
                                 "what": "gradient-arena all-reduces + running-statistics broadcasts (the quantisers' statistics all-reduces count in `comm`)",
                                 "path": "fo_comm" if abi_comm is not None else "torch.distributed"}
        if not args.no_kernel_events:
            # per-kernel timing as for `roofline`: side streams folded, HIP events per launch, two generator + two discriminator iterations
            eng_g.set_stream_overlap(False)
            gan.overlap_d2 = False
            gan.d3.overlap_scales = gan.d2.overlap_scales = False
            gan.step(cimg, cgt); gan.step(cimg, cgt)
            prof5 = ops.KernelProfiler()
            ops.PROFILER = prof5
            sync()
            t1 = time.perf_counter()
            k5 = 4
            for _ in range(k5):
                gan.step(cimg, cgt)
            sync()
            ms_serial5 = (time.perf_counter() - t1) / k5 * 1e3
            ops.PROFILER = None
            prof5.close()
            summ5 = prof5.summary()
            r5 = dominant(summ5, k5, ms_serial5)
            r5["launches_per_iteration"] = r5.pop("launches_per_step")
            r5["share_of_iteration_time"] = r5.pop("share_of_step_time")
            r5["measured"] = ("HIP events per launch over 2 generator + 2 discriminator iterations with the side streams joined; algorithmic FLOP of the launches "
                              "(taps that only reach padding excluded) against the fp32 MFMA peak")
            r5["ms_per_iteration_serial"] = round(ms_serial5, 3)
            pmc5 = os.path.join(ROOT, "profiles", "pmc_traffic_c5.json")        # CONFIG=c5 bash profiles/collect.sh <tag>: the counters on tools/bench_gan.py --serial
            if os.path.exists(pmc5):
                try:
                    tj = json.load(open(pmc5))
                    from faceoff_amd._lib import kernel_source_sha16
                    r5["traffic"] = tj.get(r5["kernel"])
                    r5["traffic_source"] = (f"HBM bytes per launch from the committed rocprofv3 --pmc passes ({tj.get('_source')}), measured on kernel sources "
                                            f"{tj.get('_kernel_source_sha16')}" + (" = this run's sources" if tj.get("_kernel_source_sha16") == kernel_source_sha16()
                                                                                    else f", this run's sources {kernel_source_sha16()}"))
                except Exception:
                    pass
            out["c5"]["roofline"] = r5
            out["c5"]["kernels"] = {k: {"launches_per_iteration": v["launches"] / k5, "avg_ms": round(v["avg_ms"], 4), "tflops": round(v["tflops"], 2),
                                        "tflops_padded_taps_counted": round(v["tflops_nominal"], 2), "ms_per_iteration": round(v["total_ms"] / k5, 3)}
                                    for k, v in sorted(summ5.items(), key=lambda kv: -kv[1]["total_ms"])[:14]}
            exec5 = sum(v["flops_per_launch"] * v["launches"] for v in summ5.values()) / k5 + VQ_FLOP_PER_FRAME * clip
            out["c5"]["executed_matrix_tflop_per_iteration"] = round(exec5 / 1e12, 4)
            out["c5"]["iteration_frac_executed_flop"] = round(exec5 / (dt5 / iters) / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
            f5 = step_floor(lambda: gan.step(cimg, cgt), 4, dt5 / iters * 1e3, lambda on: None)          # (streams are still folded from the block above)
            out["c5"]["iteration_floor"] = {("iteration" + k[4:] if k.startswith("step_") else k): v for k, v in f5.items()}
        del gan, eng_g
        torch.cuda.empty_cache()

    if rank != 0:
        if abi_comm is not None:
            abi_comm.destroy()
        torch.distributed.destroy_process_group()
        return
    if world == 1 and not args.no_cpu_baseline and not args.perceptual:
        out["cpu_baseline"] = cpu_baseline(T, H, H)
    if abi_comm is not None:
        abi_comm.destroy()
    if ddp:
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
