"""Data-parallel semantics on the real GPU path: two ranks (gloo over CUDA tensors, both on cuda:0 because
the GPU box has one device) run FaceOffTrainer with the bucketed side-stream all-reduce and the fused VQ
statistics all-reduce; the result must equal ONE process stepping on the concatenated batch (DDP mean of
per-rank means over equal shards = global mean; VQ statistics are summed; reference
train_faceoff_perceptual.py:164-169, vqvae_conv3d_latent.py:63-64)."""
import os
import tempfile

import numpy as np
import pytest
import torch

from faceoff_amd import distributed as dist
from faceoff_amd.synth import make_state_dict, make_batch

pytestmark = pytest.mark.gpu
B, T, H, W = 2, 2, 64, 64


def _worker(outdir, own_device=False):
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.trainer import FaceOffTrainer
    rank = dist.get_rank()
    devno = rank if own_device else 0
    torch.cuda.set_device(devno)
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    eng = VQVAEEngine(sd, f"cuda:{devno}")
    tr = FaceOffTrainer(eng, lr=3e-4, bucket_bytes=2 << 20)
    assert tr.reducer is not None and len(tr.reducer.buckets) >= 4
    img, gt = make_batch(100, 2 * B, T, H, W)
    sl = slice(rank * B, (rank + 1) * B)
    recon, latent, _ = tr.step(torch.from_numpy(img[sl]).cuda(), torch.from_numpy(gt[sl]).cuda())
    torch.cuda.synchronize()
    torch.save({"params": eng.flat_params.cpu(), "grads": eng.flat_grads.cpu(), "offsets": dict(eng.offsets),
                "buffers": {k: v.cpu() for k, v in eng.buffers.items()}, "ids": tuple(t.cpu() for t in tr.last_ids),
                "embed_b": eng.buffers["quantize_b.embed"].cpu(), "recon": recon.cpu(), "latent": latent.cpu()},
               os.path.join(outdir, f"rank{rank}.pt"))


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    _two_ranks_vs_serial("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_on_two_gpus_equal_one_process_on_the_concatenated_batch():
    """Real multi-rank RCCL (backend "nccl", one rank per GPU, xGMI): bucketed side-stream gradient all-reduce and the
    in-forward VQ-statistics all-reduce against ONE process stepping on the concatenated batch."""
    _two_ranks_vs_serial("nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_launches_its_own_two_ranks():
    """`python bench.py --gpus 2` without torchrun: two RCCL ranks, one JSON line with n_gpus == 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-kernel-events", "--no-c3"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["frames_per_step"] == 320
    assert abs(d["value"] - 320 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]


def _two_ranks_vs_serial(backend):
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.trainer import FaceOffTrainer
    with tempfile.TemporaryDirectory() as td:
        dist.launch(_worker, 2, 1, 0, "auto", args=(td, backend == "nccl"), backend=backend)
        r = [torch.load(os.path.join(td, f"rank{i}.pt")) for i in range(2)]
    # both ranks hold identical parameters and codebooks after the step
    assert torch.equal(r[0]["params"], r[1]["params"]) and torch.equal(r[0]["embed_b"], r[1]["embed_b"])
    assert torch.equal(r[0]["grads"], r[1]["grads"])                 # arena holds the SUM over ranks on every rank
    # serial reference: one process, 2B clips
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    eng = VQVAEEngine(sd, "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4)
    img, gt = make_batch(100, 2 * B, T, H, W)
    recon, latent, _ = tr.step(torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda())
    g_serial = eng.flat_grads.cpu()
    g_ddp = r[0]["grads"] / 2                                         # DDP averages
    assert (g_ddp - g_serial).abs().max().item() <= 1e-3 * g_serial.abs().max().item()
    np.testing.assert_allclose((r[0]["recon"] + r[1]["recon"]).item() / 2, recon.item(), rtol=1e-4)
    np.testing.assert_allclose((r[0]["latent"] + r[1]["latent"]).item() / 2, latent.item(), rtol=1e-4)
    assert (r[0]["embed_b"] - eng.buffers["quantize_b.embed"].cpu()).abs().max().item() <= \
        2e-3 * eng.buffers["quantize_b.embed"].abs().max().item()
    # one Adam step of lr 3e-4 moves every parameter by ~lr * sign(g): the two runs may pick different (equally valid)
    # conv algorithms for their batch sizes, so a gradient that is zero up to rounding can change sign and its parameter
    # end up 2 lr apart.  Everything else must agree, and such elements must be rare and have negligible gradients.
    dpar = (r[0]["params"] - eng.flat_params.cpu()).abs()
    off = dpar > 1e-4
    assert dpar.max().item() <= 2.1 * 3e-4, dpar.max().item()
    assert off.float().mean().item() < 1e-3, off.float().mean().item()
    assert (g_serial.abs()[off] <= 1e-3 * g_serial.abs().max()).all()
    _two_ranks_vs_oracle(r, sd, img, gt)


def _two_ranks_vs_oracle(r, sd, img, gt):
    """The third party: the data-parallel step against the CPU ORACLE, not only against the engine's own serial step.
    What two ranks compute is, by the reference's semantics, one step on the concatenated batch: DDP averages the per-rank gradients of
    per-rank MEAN losses over equal shards (= the gradient of the global mean, train_faceoff_perceptual.py:164-169) and Quantize sums its
    EMA statistics over the ranks before the update (vqvae_conv3d_latent.py:59-64: oracle.quantize_forward's `all_reduce` argument is that
    sum; on the concatenated batch the oracle forms the same global statistics directly).  So oracle.train_step on all 2B clips gives the
    gradients (arena / 2), the six EMA buffers and the mean of the two ranks' losses, each at the north-star 1e-3."""
    from oracle import faceoff_oracle as O
    p = O.to_torch_state(sd)
    o = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), p)
    ids = [torch.cat([r[i]["ids"][l] for i in range(2)]) for l in range(2)]          # rank-major = batch order
    assert torch.equal(ids[0], o["fw"]["id_t"]) and torch.equal(ids[1], o["fw"]["id_b"]), "two-rank code indices differ from the oracle's"
    np.testing.assert_allclose((r[0]["recon"] + r[1]["recon"]).item() / 2, o["recon"].item(), rtol=1e-3)
    np.testing.assert_allclose((r[0]["latent"] + r[1]["latent"]).item() / 2, o["latent"].item(), rtol=1e-3)
    worst = (0.0, "")
    for n, g in o["grads"].items():
        off, cnt = r[0]["offsets"][n]
        got = r[0]["grads"][off:off + cnt].reshape(g.shape) / 2
        err = (got - g).abs().max().item() / (g.abs().max().item() + 1e-30)
        worst = max(worst, (err, n))
        assert err <= 1e-3, (n, err)
    for k, v in r[0]["buffers"].items():                      # `p` holds the oracle's post-EMA buffers (global statistics)
        assert torch.equal(v, r[1]["buffers"][k]), k
        err = (v - p[k]).abs().max().item() / (p[k].abs().max().item() + 1e-30)
        assert err <= 1e-3, (k, err)
    print(f"[two ranks vs CPU oracle on the concatenated batch] code indices equal; worst gradient rel err {worst}")


_RCCL_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(sys.argv[1]), rank=0, world_size=1,
                                     device_id=torch.device("cuda", 0))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict, make_batch
from faceoff_amd.trainer import FaceOffTrainer
img, gt = make_batch(100, 2, 2, 64, 64)
img, gt = torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda()
out = []
for force in (True, False):
    eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4, bucket_bytes=2 << 20, force_collectives=force)
    assert (tr.reducer is not None) == force
    recon, latent, _ = tr.step(img, gt)
    torch.distributed.barrier()
    torch.cuda.synchronize()
    if force:
        assert len(tr.reducer.buckets) >= 4
    out.append((eng.flat_params.clone(), eng.flat_grads.clone(), eng.buffers["quantize_b.embed"].clone(), recon.item()))
    tr.step(img, gt)                      # a second step re-arms the buckets (EMA rounding makes it non-bitwise)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.flat_params).all()
assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), "one-rank RCCL all-reduce changed the result"
assert (out[0][2] - out[1][2]).abs().max().item() <= 1e-5 * out[1][2].abs().max().item()
torch.distributed.destroy_process_group()
print("RCCL_PATH_OK")
"""


def test_rccl_code_path_in_a_one_rank_group():
    """The GPU box has ONE device and RCCL refuses two ranks on it, so the RCCL-specific plumbing (process group with
    device_id, async all-reduce of arena slices issued under the side stream, work.wait() on that stream, the in-forward
    VQ-statistics all-reduce, barrier) is exercised with a one-rank "nccl" group and forced collectives: a training
    step must give exactly the parameters and gradients of the plain path."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT, str(port)], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "RCCL_PATH_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


_ABI_COMM_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
from faceoff_amd.distributed.comm import AbiComm
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict, make_batch
from faceoff_amd.trainer import FaceOffTrainer
comm = AbiComm.create(0, 1, "cuda:0")
# the collective itself: in place, behind the producing stream, awaited on the device
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device="cuda")
with torch.cuda.stream(side):
    x.add_(3.0)
    comm.allreduce_async(x)                 # behind `side`
comm.wait()                                 # the current (default) stream waits for it
y = x * 2
torch.cuda.synchronize()
assert comm.issued == 1 and torch.equal(y, torch.full_like(y, 6.0))
img, gt = make_batch(100, 2, 2, 64, 64)
img, gt = torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda()
out = []
for use in (True, False):
    eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4, bucket_bytes=2 << 20, force_collectives=use, comm=comm if use else None)
    assert (tr.reducer is not None) == use
    before = comm.issued
    recon, latent, _ = tr.step(img, gt)
    torch.cuda.synchronize()
    if use:
        assert len(tr.reducer.buckets) >= 4
        assert comm.issued - before == len(tr.reducer.buckets) + 2      # every bucket + the two quantisers' statistics
    out.append((eng.flat_params.clone(), eng.flat_grads.clone(), eng.buffers["quantize_b.embed"].clone()))
    tr.step(img, gt)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.flat_params).all()
assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), "one-rank all-reduce through fo_comm changed the result"
assert (out[0][2] - out[1][2]).abs().max().item() <= 1e-5 * out[1][2].abs().max().item()
comm.destroy()
print("ABI_COMM_OK")
"""


def test_c_abi_communicator_in_a_one_rank_world():
    """fo_comm_{unique_id,init,allreduce_async,wait,destroy} (csrc/comm.cpp: RCCL behind the C-ABI) with one rank -- all this box
    can host: the collective is ordered behind its producer stream and awaited on the device; a training step whose gradient buckets and
    VQ statistics travel through it gives exactly the plain step's parameters and gradients, and issues one all-reduce per bucket + two."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _ABI_COMM_SCRIPT], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "ABI_COMM_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


_ABI_TWO_RANK_SCRIPT = r"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank, idfile = int(sys.argv[1]), sys.argv[2]
torch.cuda.set_device(rank)
from faceoff_amd.distributed.comm import AbiComm
def exchange(idb):                          # the "job's own rendezvous": a file rank 0 writes and rank 1 polls
    if idb is not None:
        with open(idfile + ".tmp", "wb") as f: f.write(idb)
        os.replace(idfile + ".tmp", idfile)
        return idb
    for _ in range(600):
        if os.path.exists(idfile):
            return open(idfile, "rb").read()
        time.sleep(0.1)
    raise RuntimeError("rank 0 never published the id")
comm = AbiComm.create(rank, 2, f"cuda:{rank}", exchange)
x = torch.full((1 << 20,), float(rank + 1), device=f"cuda:{rank}")
for _ in range(3):
    comm.allreduce_async(x)                 # 1, 2 -> 3, 3 -> 6, 6 -> 12, 12
comm.wait()
torch.cuda.synchronize()
assert torch.equal(x, torch.full_like(x, 12.0)), x[:4]
comm.destroy()
print("ABI_TWO_RANK_OK")
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_c_abi_communicator_two_ranks_on_two_gpus(tmp_path):
    """Two processes, one GPU each, id handed over through a file: three chained in-place SUMs through fo_comm_allreduce_async."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idfile = str(tmp_path / "fo_comm_id")
    procs = [subprocess.Popen([sys.executable, "-c", _ABI_TWO_RANK_SCRIPT, str(r), idfile], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, cwd=root) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "ABI_TWO_RANK_OK" in so, (so[-1000:], se[-3000:])
