#!/usr/bin/env python3
"""FaceOff MI355X bench: train frames/s of the VQ-VAE-2 + Conv3d-latent step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], "C2"): 256x256, T=5, bs=32 clips per GPU (160 frames/GPU/step),
recon + VQ loss, fp32, synthetic inputs resident in HBM, random-init weights.  A step is one full
training iteration: forward, fused losses, backward (all 70 gradients), bucketed RCCL gradient
all-reduce overlapped with backward (N>1), one multi-tensor Adam launch.  Weak scaling: per-GPU work
is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense (same table)
FLOP_PER_FRAME = 64.1e9                # 63.77 GFLOP conv + 0.34 GFLOP VQ distance per frame


def cpu_baseline(T, H, W, steps=20):
    """The CPU oracle (torch-CPU restatement of the reference step, kind "port") timed on this box's
    host cores on a bounded sample: one clip of T frames at HxW, forward + backward + Adam."""
    from oracle import faceoff_oracle as O
    from faceoff_amd.synth import make_state_dict, make_batch
    # torch-CPU (oneDNN) stops scaling well past a few dozen threads on these layer sizes: use up to 32
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    p = O.to_torch_state(make_state_dict(0, codebook_scale=0.3, gain=2.0))
    img, gt = make_batch(1, 1, T, H, W)
    img, gt = torch.from_numpy(img), torch.from_numpy(gt)
    st = {}
    for _ in range(2):
        O.train_step(img, gt, p, adam_state=st)      # warm-up (oneDNN primitive caches, thread pool)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(img, gt, p, adam_state=st)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(T / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 clip x {T} frames {H}x{W}, fwd+bwd+Adam, {steps} timed steps (~{dt * steps:.0f} s) after 2 warm-ups, torch-CPU oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=32, help="clips per GPU (BASELINE: 32)")
    ap.add_argument("--frames", type=int, default=5, help="frames per clip T (BASELINE: 5)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--perceptual", action="store_true",
                    help="BASELINE config 3: add the LPIPS/VGG-16 term (train_faceoff_perceptual.py path, seeded VGG "
                         "weights, --lpips-dtype); the headline metric is config 2 (without it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lpips-dtype", choices=("bf16", "fp32"), default="bf16",
                    help="--perceptual: arithmetic of the LPIPS / VGG-16 branch (BASELINE config 3 = bf16)")
    ap.add_argument("--direct-conv", action="store_true",
                    help="run the Conv3d / 3x3 128->128 layers on the direct implicit-GEMM kernels instead of Winograd "
                         "(the kernel-quality reference: same results to fp32 rounding, 1.5x the step time)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP-event timing")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the step without side-stream overlap (what profiles/collect.sh traces: kernels run alone)")
    args = ap.parse_args()

    if args.direct_conv:
        os.environ["FACEOFF_NO_WINOGRAD"] = "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
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
        torch.distributed.init_process_group("nccl", device_id=dev)

    from faceoff_amd import ops
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.synth import make_state_dict
    from faceoff_amd.trainer import FaceOffTrainer

    B, T, H = args.clips, args.frames, args.size
    eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
    vqlpips = None
    if args.perceptual:
        from faceoff_amd.loss import VQLPIPS
        from faceoff_amd.synth import make_vgg_lpips_state
        vqlpips = VQLPIPS(make_vgg_lpips_state(7), dtype=args.lpips_dtype).to(dev)
    trainer = FaceOffTrainer(eng, lr=3e-4, vqlpips=vqlpips, force_collectives=ddp and world == 1)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    img = torch.rand((B * T, 6, H, H), device=dev, generator=gen) * 2 - 1       # U(-1,1): dataset.py:240-247
    gt = torch.rand((B * T, 3, H, H), device=dev, generator=gen) * 2 - 1

    def sync():
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.serial_streams:
        eng.set_stream_overlap(False)
    for _ in range(args.warmup):
        trainer.step(img, gt, T=T)
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        recon, latent, _ = trainer.step(img, gt, T=T)
    ev1.record()                      # every side stream is joined into this one at the end of a step
    sync()
    dt = time.perf_counter() - t0
    ms_events = ev0.elapsed_time(ev1) / args.steps
    # Per-kernel durations: the same K steps again with the side streams folded into the main one, so that every
    # launch has the GPU to itself (kernels sharing the chip stretch each other's event-to-event time; the step time
    # above is the overlapped one).  HIP events bracket each launch on the stream it is launched on.
    prof = None
    ms_serial = None
    if not args.no_kernel_events:
        eng.set_stream_overlap(False)
        trainer.step(img, gt, T=T)
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(img, gt, T=T)
        sync()
        ms_serial = (time.perf_counter() - t1) / args.steps * 1e3
        ops.PROFILER = None
    if ddp:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = tmax.item()
    if rank != 0:
        torch.distributed.destroy_process_group()
        return

    ms = dt / args.steps * 1e3
    fps = world * B * T * args.steps / dt
    out = {
        "metric": "train frames/sec (256x256, T=5, bs=32/GPU)", "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "ms_per_step_hip_events": round(ms_events, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic U(-1,1) inputs resident in HBM; random-init weights (kaiming-uniform x2 gain, codebook N(0,0.3^2))",
        "config": {"workload": (f"C2: VQ-VAE-2 + Conv3d latent, {H}x{H}, T={T}, {B} clips/GPU, recon+VQ loss, fwd+bwd+Adam"
                                if not args.perceptual else
                                f"C3: C2 (fp32) + LPIPS/VGG-16 perceptual loss in {args.lpips_dtype} (seeded VGG weights), {H}x{H}, T={T}, {B} clips/GPU"),
                   "global_clips": B * world, "frames_per_step": B * T * world, "parallelism": f"dp{world}"},
        "loss": {"recon": round(recon.item(), 6), "latent": round(latent.item(), 6)},
    }
    # whole-step roofline: time the step's algorithmic FLOP would take at the dense MFMA peak of the type each part runs in
    ideal_s = FLOP_PER_FRAME / (FP32_MFMA_PEAK_TFLOPS * 1e12)
    if args.perceptual:
        ideal_s += 120.3e9 / ((BF16_MFMA_PEAK_TFLOPS if args.lpips_dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS) * 1e12)
        out["dtype"] = "f32 (VQ-VAE) + %s (LPIPS)" % ("bf16" if args.lpips_dtype == "bf16" else "f32")
    out["step_frac_of_mfma_roofline"] = round(ideal_s * fps / world, 4)      # SURVEY 8(d) FLOP count: padded taps included
    # the same with only the FLOP the matrix pipe actually executes: (a) Conv3d taps that fall into clip padding are
    # structural zeros, skipped by the kernels (2 of 15 (frame, depth tap) pairs at T=5; Conv3d is 63.7 % of the conv
    # FLOP, fwd / dgrad / wgrad alike); (b) Conv3d forward, data gradient and filter gradient run as Winograd F(2x2,3x3):
    # 16 multiplies per 2x2 outputs instead of 36.  With (b) the direct-convolution FLOP count is no longer a bound.
    from faceoff_amd.ops import temporal_share
    conv3d_flop = 3 * 2 * 6.795e9                                           # per frame: 6.795 GMAC fwd x 3 passes
    executed = FLOP_PER_FRAME - conv3d_flop * (1.0 - temporal_share(T))
    out["step_frac_direct_flop_without_padding_taps"] = round(executed * fps / world / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
    if eng.winograd:
        # F(4x4,3x3) = 36 multiplies per 16 outputs instead of 144, F(2x2,3x3) = 16 per 4 instead of 36 (per-frame GMAC of
        # the bottom / top Conv3d chains: SURVEY 8a, row a4)
        from faceoff_amd.ops import wino_tile
        frac = {4: 36.0 / 144.0, 2: 16.0 / 36.0, 0: 1.0}
        f_b = frac[min(wino_tile(H // 4, H // 4, B * T), eng.winograd_max_tile)]
        f_t = frac[min(wino_tile(H // 8, H // 8, B * T), eng.winograd_max_tile)]
        executed -= 3 * 2 * temporal_share(T) * (5.436e9 * (1.0 - f_b) + 1.359e9 * (1.0 - f_t))
    out["step_frac_executed_flop"] = round(executed * fps / world / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
    out["config"]["conv3d_algorithm"] = ("winograd (fwd, dgrad, wgrad), F(4x4,3x3) where a plane is whole GEMM tiles else F(2x2,3x3)"
                                         if eng.winograd else "direct")
    if prof is not None:
        summ = prof.summary()
        dom = max(summ, key=lambda k: summ[k]["total_ms"])
        d = summ[dom]
        peak = BF16_MFMA_PEAK_TFLOPS if dom.startswith("conv_bf16") else FP32_MFMA_PEAK_TFLOPS
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(d["tflops"], 2), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(d["tflops"] / peak, 4), "traffic": None,
                           "achieved_padded_taps_counted": round(d["tflops_nominal"], 2),
                           "launches_per_step": d["launches"] / args.steps, "avg_launch_ms": round(d["avg_ms"], 4),
                           "algorithmic_gflop_per_launch": round(d["flops_per_launch"] / 1e9, 3),
                           "share_of_step_time": round(d["total_ms"] / (ms_serial * args.steps), 4),
                           "measured": "HIP events per launch over a second K-step region with the side streams joined "
                                       "(kernels run alone; ms_per_step_serial is that region's step time incl. event overhead)"}
        if eng.winograd:
            out["roofline"]["note"] = ("launches of this kernel are mostly Winograd-domain GEMMs (K = 3*128 or 128: short); "
                                       "bench.py --direct-conv runs the same step on direct convolutions: 1.5x slower, "
                                       "dominant kernel conv_igemm3 at 0.91 of the fp32 MFMA peak (profiles/r01_bench_direct_conv3d.json)")
        out["ms_per_step_serial"] = round(ms_serial, 3)
        out["kernels"] = {k: {"launches_per_step": v["launches"] / args.steps, "avg_ms": round(v["avg_ms"], 4),
                              "tflops": round(v["tflops"], 2), "tflops_padded_taps_counted": round(v["tflops_nominal"], 2), "ms_per_step": round(v["total_ms"] / args.steps, 3)}
                          for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # written by profiles/collect.sh from rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                out["roofline"]["traffic"] = json.load(open(pmc)).get(dom)
            except Exception:
                pass
    if world == 1 and not args.no_cpu_baseline and not args.perceptual:
        out["cpu_baseline"] = cpu_baseline(T, H, H)
    if ddp:
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
