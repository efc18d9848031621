"""bench.py contract (the driver parses ONE JSON line): keys, types, and that the roofline leg is self-consistent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_valid_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--c3-sustained", "6"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 160 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]        # frames/s = frames per step / step time
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic FLOP per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_ms"]) < 0.02 * r["achieved"]
    # per-kernel entries are keyed by the kernel SYMBOL the library reports (what rocprofv3 prints: profiles/*_kernel_stats.md), not by a label
    assert r["kernel"] == "wino_gemm_kernel", r["kernel"]
    assert "resblock_halo_fwd_kernel" in d["kernels"] and any(k.startswith("conv_wgrad_kernel<") for k in d["kernels"]), list(d["kernels"])
    assert not any(k.startswith(("conv_igemm_bn", "fo_")) for k in d["kernels"]), list(d["kernels"])       # no made-up bucket names left
    dc = r["direct_conv"]
    assert dc["kernel"].startswith("conv_igemm3_kernel<") and 0 < dc["step_frac_executed_flop"] < dc["step_frac_nominal_direct_conv_flop"] <= 1.0
    # the opt-in leg with fp32 products on the bf16 matrix pipe: its own step time (its losses are those of its own, fresh
    # trainer after its own number of steps; tests/test_split_gpu.py compares the two paths on equal inputs)
    # config 3 (bf16 MFMA operands for the VQ-VAE and the LPIPS branch): its own value, roofline block against the bf16 peak, and the two
    # comparison steps (bf16 VQ-VAE without LPIPS; round 2's fp32 VQ-VAE + bf16 LPIPS)
    c3 = d["c3"]
    assert c3["value"] > 0 and abs(c3["value"] - 160 / (c3["ms_per_step"] * 1e-3)) < 1e-2 * c3["value"] and c3["dtype"].startswith("bf16")
    # every leg honours --steps (and warms a fresh engine at least three times); config 3 also carries a sustained figure
    assert c3["steps"] == 2 and c3["warmup"] == 3 and d["h2d_fed"]["steps"] == 2 and d["bf16x6"]["steps"] == 2 and d["roofline"]["direct_conv"]["steps"] == 2
    assert c3["sustained_steps"] == 6 and 0.7 * c3["ms_per_step"] < c3["ms_per_step_sustained"] < 1.3 * c3["ms_per_step"]
    assert d["c5"]["iterations"] == 2 and d["c5"]["warmup"] == 4
    r3 = c3["roofline"]
    assert r3["peak"] == 2500.0 and r3["bound"] == "mfma" and 0 < r3["frac"] <= 1.0 and abs(r3["frac"] - r3["achieved"] / r3["peak"]) < 1e-3
    assert c3["vqvae_only_bf16"]["value"] > c3["value"] and c3["fp32_vqvae"]["value"] > 0
    assert 0 < c3["loss"]["perceptual"] and 0 < c3["loss"]["recon"] < 1
    assert r3["kernel"].startswith(("conv_bf16_", "wgrad_bf16_kernel<", "wgrad9_bf16_kernel<", "conv_halo64_bf16")) and not any(k.startswith(("conv_bf16_bn", "conv_bf16_big")) for k in c3["kernels"])
    # config 5: its own roofline / kernels block
    c5 = d["c5"]
    r5 = c5["roofline"]
    assert r5["peak"] == 157.3 and 0 < r5["frac"] <= 1.0 and abs(r5["frac"] - r5["achieved"] / r5["peak"]) < 1e-3 and r5["kernel"] in c5["kernels"]
    assert "conv_gen_kernel" in c5["kernels"] and any(k.startswith("wgrad_gen_kernel<") for k in c5["kernels"]) and 0 < c5["iteration_frac_executed_flop"] < 1
    x = d["bf16x6"]
    assert any(k.startswith("wino_gemm_split") for k in x["kernels"]), list(x["kernels"])
    # the step-level attainable floor (round 6): every launch of a step at its own bound, summed -- below the timed step, above the perfect-overlap
    # figure, the byte model never counting a launch above its own measured time, the gaps named by kernel symbol
    for leg, f, key in ((d, d["step_floor"], "step"), (c3, c3["step_floor"], "step"), (c5, c5["iteration_floor"], "iteration")):
        assert 0 < f[f"{key}_floor_perfect_overlap_ms"] <= f[f"{key}_floor_ms"] and 0 < f[f"{key}_frac_of_floor"] <= 1.0, (key, f)
        assert abs(f[f"{key}_floor_perfect_overlap_ms"] - max(f["sum_hbm_ms"], f["sum_mfma_ms"])) < 1e-2
        assert f["floor_above_measured_ms"] <= 0.02 * f[f"{key}_floor_ms"], f["floor_above_measured_ms"]
        assert f["launches_per_step"] > 100 and len(f["largest_gaps"]) >= 4 and all(g["measured_ms"] >= g["floor_ms"] * 0.98 for g in f["largest_gaps"])
    assert d["step_floor"]["largest_gaps"][0]["kernel"] in d["kernels"] or d["step_floor"]["largest_gaps"][0]["kernel"].startswith(("fo_", "wino_", "conv_"))
    assert x["value"] > 0 and abs(x["value"] - 160 / (x["ms_per_step"] * 1e-3)) < 1e-2 * x["value"]
    assert 0 < x["loss"]["recon"] < 1 and 0 < x["loss"]["latent"] < 1


def test_bench_multi_gpu_code_path_with_one_rank():
    """The branch the driver's 2/4/8-GPU runs take (RCCL group, bucketed all-reduce, barrier, max over ranks), forced on
    with a single rank: still exactly one JSON line on stdout (RCCL writes its banner to stdout) and a sane value."""
    env = dict(os.environ, FACEOFF_BENCH_FORCE_DDP="1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-kernel-events", "--c3-sustained", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    # the self-proving `comm` block: ranks the group really had, bytes moved, buckets, exposed all-reduce time, per-rank spread
    c = d["comm"]
    assert c["ranks_in_group"] == 1 and c["backend"] == "nccl" and c["rccl_version"] and c["collectives_forced_in_one_rank_group"] is True
    # default exchange for an RCCL group: the C-ABI communicator -- every bucket + the two quantisers' statistics per step went through fo_comm_*
    assert c["path"] == "fo_comm" and c["fo_comm_issued_per_step"] == c["buckets"] + 2, (c["path"], c["fo_comm_issued_per_step"], c["buckets"])
    assert c["grad_allreduce_bytes_per_step"] >= 4 * 4049990 and c["grad_allreduce_bytes_per_step"] % 16 == 0 and c["vq_stats_allreduce_bytes_per_step"] == 2 * (512 + 512 * 64) * 4
    assert c["allreduce_bytes_per_step"] == c["grad_allreduce_bytes_per_step"] + c["vq_stats_allreduce_bytes_per_step"]
    assert c["buckets"] == len(c["bucket_bytes"]) >= 3 and sum(c["bucket_bytes"]) == c["grad_allreduce_bytes_per_step"]
    assert c["exposed_ms"] is not None and 0 <= c["exposed_ms"] < d["ms_per_step"]
    assert c["ms_per_step_min_rank"] <= c["ms_per_step_max_rank"] and abs(c["ms_per_step_max_rank"] - d["ms_per_step"]) < 1e-2 * d["ms_per_step"]
    # config 5's data-parallel branches ride along (round 6): per iteration (1 + 2) [generator arena + two running-statistics broadcasts] or (2 + 2)
    c5 = d["c5"]["comm"]
    assert c5["path"] == "fo_comm" and c5["collectives_per_iteration"] == 3.5, c5


def test_bench_multi_gpu_code_path_over_torch_distributed():
    """--comm torch: the same forced one-rank run with the all-reduces issued through torch.distributed (round 3's path)."""
    env = dict(os.environ, FACEOFF_BENCH_FORCE_DDP="1", MASTER_PORT="29578")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events",
                          "--no-c3", "--no-c5", "--no-x6-leg", "--no-h2d-leg", "--comm", "torch", "--host-cpus", "2"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert len(d["host"]["confined_to_cpus"]) == 2 and d["host"]["torch_threads"] == 1 and "cpu_baseline" not in d      # (--host-cpus: the 8-rank host budget rehearsal)
    c = d["comm"]
    assert c["path"] == "torch.distributed" and c["fo_comm_issued_per_step"] is None and c["buckets"] >= 3


def test_bench_two_rank_control_flow_on_one_gpu():
    """The multi-rank flow of bench.py -- process group, barriers, max over ranks, every leg (direct-conv, c3, c5) run by
    every rank, rank 0 alone printing the JSON line, clean exits -- with two ranks sharing device 0 over gloo (the box has one
    GPU; RCCL refuses two ranks on one device).  Small clip count: this checks the flow, not the speed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FACEOFF_BENCH_SHARE_GPU="1", FACEOFF_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips", "8",
                                       "--no-cpu-baseline", "--c3-sustained", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines0 = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines0) == 1 and not outs[1][0].strip(), (lines0, outs[1][0][-300:])
    d = json.loads(lines0[0])
    assert d["config"]["global_clips"] == 16 and d["config"]["frames_per_step"] == 80 and "c3" in d and "c5" in d and "direct_conv" in d["roofline"]
    assert d["comm"]["ranks_in_group"] == 2 and d["comm"]["backend"] == "gloo" and d["comm"]["collectives_forced_in_one_rank_group"] is False
    assert d["comm"]["exposed_ms"] is not None and d["comm"]["ms_per_step_min_rank"] <= d["comm"]["ms_per_step_max_rank"]
    assert abs(d["value"] - 80 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
