"""N>1 path on CPU: world_size-2 `gloo` processes exercise the reference-compatible helpers
(all_reduce / all_gather / reduce_dict / launch) and the gradient bucket reducer that replaces DDP
(buckets fire in backward order, SUM over ranks equals the serial sum, VQ statistics in one message)."""
import os
import tempfile

import numpy as np
import torch

from faceoff_amd import distributed as dist


def _worker(outdir):
    import torch.distributed as td
    from faceoff_amd.distributed import GradBucketReducer, fused_vq_allreduce
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {"rank": rank, "world": world, "primary": dist.is_primary(), "local_rank": dist.get_local_rank()}
    # helpers (reference distributed/distributed.py:64-132)
    t = torch.full((4,), float(rank + 1))
    dist.all_reduce(t)
    res["all_reduce"] = t.tolist()
    res["all_gather"] = dist.all_gather({"mse_sum": 1.5 * (rank + 1), "mse_n": rank + 2})
    rd = dist.reduce_dict({"a": torch.tensor(float(rank)), "b": torch.tensor(2.0)})
    res["reduce_dict"] = {k: float(v) for k, v in rd.items()}
    dist.synchronize()
    # bucketed gradient exchange over a toy arena: 5 "layers" of 1000 floats, arena order L0..L4,
    # backward completes L4 first
    order = [f"L{i}" for i in range(5)]
    offsets = {}
    for i, n in enumerate(order):
        offsets[n + ".weight"] = (i * 1000, 996)
        offsets[n + ".bias"] = (i * 1000 + 996, 4)
    flat = torch.arange(5000, dtype=torch.float32) * (rank + 1)
    red = GradBucketReducer(flat, order, offsets, bucket_bytes=2000 * 4)
    res["buckets"] = [(lo, hi, trig) for lo, hi, trig in red.buckets]
    fired = []
    for n in reversed(order):            # backward order
        red.layer_done(n)
        fired.append(list(red.launched))
    res["fired"] = fired
    red.finish()
    res["flat_sum_ok"] = bool(torch.equal(flat, torch.arange(5000, dtype=torch.float32) * sum(range(1, world + 1))))
    # the engine overlaps independent chains, so layers may report out of arena order: a bucket must wait for ALL members
    flat.copy_(torch.arange(5000, dtype=torch.float32) * (rank + 1))
    fired2 = []
    for n in ("L2", "L4", "L0", "L3", "L1"):
        red.layer_done(n)
        fired2.append(list(red.launched))
    red.finish()
    res["fired_out_of_order"] = fired2
    res["flat_sum_ok2"] = bool(torch.equal(flat, torch.arange(5000, dtype=torch.float32) * sum(range(1, world + 1))))
    stats = torch.ones(512 + 512 * 64) * (rank + 1)
    fused_vq_allreduce()(stats)
    res["vq_stats"] = float(stats[0])
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))


def test_world2_gloo_helpers_and_bucket_reducer():
    with tempfile.TemporaryDirectory() as td:
        dist.launch(_worker, 2, 1, 0, "auto", args=(td,), backend="gloo")
        r = [torch.load(os.path.join(td, f"rank{i}.pt")) for i in range(2)]
    assert [x["rank"] for x in r] == [0, 1] and all(x["world"] == 2 for x in r)
    assert r[0]["primary"] and not r[1]["primary"]
    assert [x["local_rank"] for x in r] == [0, 1]
    for x in r:
        assert x["all_reduce"] == [3.0] * 4
        assert x["all_gather"] == [{"mse_sum": 1.5, "mse_n": 2}, {"mse_sum": 3.0, "mse_n": 3}]
        assert x["flat_sum_ok"] and x["vq_stats"] == 3.0
        # 2000-float buckets over a 5000-float arena, built from the END of the arena
        assert x["buckets"] == [(3000, 5000, "L3"), (1000, 3000, "L1"), (0, 1000, "L0")]
        # L4 done -> nothing; L3 done -> bucket 0; L2 -> nothing new; L1 -> bucket 1; L0 -> bucket 2
        assert x["fired"] == [[], [0], [0], [0, 1], [0, 1, 2]]
        # L2, L4 done -> nothing complete; L0 -> bucket 2 (= {L0}); L3 -> bucket 0 (= {L4, L3}); L1 -> bucket 1 (= {L2, L1})
        assert x["fired_out_of_order"] == [[], [], [2], [2, 0], [2, 0, 1]] and x["flat_sum_ok2"]
    assert r[0]["reduce_dict"] == {"a": 0.5, "b": 2.0}          # averaged on rank 0 (reference :127-128)


def test_single_process_degrades_like_the_reference():
    """No process group: helpers are no-ops (reference distributed.py:16-23,54-68)."""
    assert dist.get_rank() == 0 and dist.get_world_size() == 1 and dist.is_primary()
    assert dist.get_local_rank() == 0
    t = torch.ones(3)
    assert dist.all_reduce(t) is t and t.tolist() == [1, 1, 1]
    assert dist.all_gather({"x": 1}) == [{"x": 1}]
    d = {"a": torch.tensor(1.0)}
    assert dist.reduce_dict(d) is d
    dist.synchronize()
    out = []
    dist.launch(lambda v: out.append(v), 1, 1, 0, None, args=(7,))     # world 1: runs in-process (launch.py:48-49)
    assert out == [7]
