"""CPU-side checks (no GPU): C-ABI library loads and exports every declared symbol, host logic
(synthetic weights, state_dict layout, LR schedule, bucket planning) and the product path's refusal
to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from faceoff_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "faceoff_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fo_[a-zA-Z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    assert os.path.exists(_lib.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared, set(declared) ^ set(_lib.SIGNATURES)
    # ... and nothing else: the dynamic symbol table IS the header (csrc/exports.map keeps cross-file helpers local)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in nm.splitlines() if ln.strip())
    assert exported == declared, set(exported) ^ set(declared)
    assert _lib.load().fo_version() == _lib.ABI_VERSION            # no compute calls without a GPU
    assert int(re.search(r"#define\s+FO_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(tmp_path):
    """signatures changed (the loss kernels' `ws` argument) while fo_version() still said 100 -- an older .so handed in through
    FACEOFF_HIP_LIB would have read a stream pointer as a workspace.  Now the version is part of the contract: a library whose
    fo_version() is not the binding's ABI_VERSION is refused at load time (checked here with a stub library, in a fresh interpreter)."""
    import subprocess
    import sys
    from faceoff_amd import _lib
    src = tmp_path / "stub.c"
    names = [n for n in _lib.SIGNATURES if n != "fo_version"]
    src.write_text("int fo_version(void) { return 100; }\n" + "".join(f"int {n}(void) {{ return 0; }}\n" for n in names))
    so = tmp_path / "libstub.so"
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    code = ("import sys; sys.path.insert(0, %r)\nfrom faceoff_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.FaceoffHipError as e:\n    print('REFUSED', e)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, FACEOFF_HIP_LIB=str(so)), timeout=300)
    assert "REFUSED" in out.stdout and "ABI version 100" in out.stdout, (out.stdout, out.stderr[-500:])


def test_state_dict_layout_matches_reference_inventory():
    """70 parameters + 6 buffers, reference names and NCHW/OIHW shapes."""
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    from faceoff_amd.synth import make_state_dict
    m = VQVAE(in_channel=6)
    sd = m.state_dict()
    ref = make_state_dict(0)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == ref[k].shape for k in ref)
    assert sum(p.numel() for p in m.parameters()) == 4049990 and len(list(m.parameters())) == 70
    assert sd["dec_t.blocks.4.weight"].shape == (128, 64, 4, 4)            # ConvTranspose2d: [Cin, Cout, kh, kw]
    assert sd["conv3d_encoded_b.conv3d.0.0.weight"].shape == (128, 128, 3, 3, 3)
    # checkpoints saved from a DDP-wrapped reference model carry "module." (train_faceoff_perceptual.py:178-185)
    m.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in ref.items()})
    assert torch.equal(m.state_dict()["quantize_t.embed"], torch.from_numpy(ref["quantize_t.embed"]))


def test_product_path_refuses_cpu():
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    m = VQVAE(in_channel=6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 6, 64, 64))


def test_cycle_scheduler_restatement():
    """CycleScheduler (reference scheduler.py:251-320, as configured at train_faceoff_perceptual.py:194-201)."""
    from faceoff_amd.scheduler import CycleScheduler

    class Opt:
        param_groups = [{"lr": 0.0}]

    n_iter, lr = 200, 3e-4
    s = CycleScheduler(Opt(), lr, n_iter=n_iter, momentum=None, warmup_proportion=0.05)
    got = [s.step()[0] for _ in range(n_iter)]
    warm = int(n_iter * 0.05)
    want = []
    for i in range(1, warm + 1):                       # linear lr/25 -> lr
        want.append(lr / 25 + (i / warm) * (lr - lr / 25))
    for i in range(1, n_iter - warm + 1):              # cosine lr -> lr/25/1e4
        end = lr / 25 / 1e4
        want.append(end + (lr - end) / 2 * (np.cos(np.pi * i / (n_iter - warm)) + 1))
    np.testing.assert_allclose(got, want, rtol=1e-12)
    assert Opt.param_groups[0]["lr"] == got[-1]
    ref_path = "/root/reference/scheduler.py"
    if os.path.exists(ref_path):                        # live cross-check where the reference exists
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_scheduler", ref_path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        o = Opt()
        r = mod.CycleScheduler(o, lr, n_iter=n_iter, momentum=None, warmup_proportion=0.05)
        np.testing.assert_allclose([r.step()[0] for _ in range(n_iter)], got, rtol=1e-12)
        # default arguments (momentum cycling through betas[0] / momentum, 30 % warm-up), over two and a half cycles
        for groups in ([{"lr": 0.0, "betas": (0.9, 0.999)}], [{"lr": 0.0, "momentum": 0.9}]):
            class A:
                param_groups = [dict(g) for g in groups]

            class B:
                param_groups = [dict(g) for g in groups]
            mine, ref = CycleScheduler(A(), 1e-3, n_iter=37), mod.CycleScheduler(B(), 1e-3, n_iter=37)
            for _ in range(93):
                assert mine.step() == ref.step()
                assert A.param_groups == B.param_groups


def test_bucket_plan_covers_arena_in_backward_order():
    from faceoff_amd.distributed import GradBucketReducer
    from faceoff_amd.engine import BACKWARD_ORDER
    from faceoff_amd.synth import vqvae_param_specs
    # replicate the engine's arena layout without a device
    shapes = {n: (k, s) for n, k, s in vqvae_param_specs(in_channel=6) if k != "vq"}
    order = list(reversed(BACKWARD_ORDER))
    offsets, off = {}, 0
    for name in order:
        kind, shape = shapes[name]
        nw = int(np.prod(shape))
        nb = shape[1] if kind == "convT" else shape[0]
        for key, n in ((name + ".weight", nw), (name + ".bias", nb)):
            offsets[key] = (off, n)
            off += (n + 3) // 4 * 4
    flat = torch.zeros(off)
    red = GradBucketReducer(flat, order, offsets, bucket_bytes=4 << 20)
    spans = sorted((lo, hi) for lo, hi, _ in red.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == off
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))              # contiguous, no overlap
    assert 3 <= len(red.buckets) <= 6
    # buckets fire in the order backward completes them: their trigger layers appear in BACKWARD_ORDER order
    trig = [BACKWARD_ORDER.index(t) for _, _, t in red.buckets]
    assert trig == sorted(trig)
    assert red.buckets[0][1] == off                                          # first bucket = end of the arena (dec.*)


def test_process_data_mirrors_reference_contract():
    """utils.py:29-38: loader 5-tuple of [1,T,3,H,W] -> (img [T,6,H,W] = source||background on channels, S = T,
    ground_truth = source_images, source_images_original), and the zero-copy split the trainer uses."""
    from faceoff_amd.utils import process_data, split_batch
    g = torch.Generator().manual_seed(3)
    T, H, W = 3, 8, 8
    data = tuple(torch.rand((1, T, 3, H, W), generator=g) for _ in range(5))
    img, S, gt, orig = process_data(data, "cpu", None)
    assert img.shape == (T, 6, H, W) and S == T
    assert torch.equal(img[:, :3], data[0][0]) and torch.equal(img[:, 3:], data[2][0])
    assert torch.equal(gt, data[3][0]) and torch.equal(orig, data[4][0])
    (src, bg), T2, gt2 = split_batch(data, "cpu")
    assert T2 == T and torch.equal(torch.cat([src, bg], 1), img) and torch.equal(gt2, gt)
    # clip batches [B,T,3,H,W] flatten to B*T frames
    datab = tuple(torch.rand((2, T, 3, H, W), generator=g) for _ in range(5))
    (src, bg), T3, gt3 = split_batch(datab, "cpu")
    assert T3 == T and src.shape == (2 * T, 3, H, W) and gt3.shape == (2 * T, 3, H, W)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under faceoff_amd/, bench.py's product leg or __graft_entry__.build may
    import it (bench.py's cpu_baseline leg and smoke() are the two permitted checkers)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dirpath, _, files in os.walk(os.path.join(root, "faceoff_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), os.path.join(dirpath, f)
    bench = open(os.path.join(root, "bench.py")).read()
    uses = [m.start() for m in pat.finditer(bench)]
    # every import of the oracle in bench.py sits inside the cpu_baseline leg: cpu_baseline() itself or its timing helper _cpu_oracle_rate()
    def enclosing(u):
        return re.findall(r"^def (\w+)", bench[:u], re.M)[-1]
    assert uses and all(enclosing(u) in ("cpu_baseline", "_cpu_oracle_rate") for u in uses), [enclosing(u) for u in uses]


def test_bench_refuses_to_run_fewer_gpus_than_asked():
    """`python bench.py --gpus N` starts its own ranks; with fewer than N devices it must fail, not measure one GPU
    and report n_gpus 1 (this container has no GPU at all).  A WORLD_SIZE that contradicts --gpus is an error too."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                             text=True, timeout=300, env=env)
        assert out.returncode != 0 and out.stdout.strip() == "" and "refusing" in out.stderr
    out = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and out.stdout.strip() == "" and "WORLD_SIZE" in out.stderr


def test_counted_vmcnt_kernels_have_no_scratch_and_the_ring_kernel_its_verified_instruction_mix():
    """tools/check_isa.py: conv_rgb_dgrad_ring_bf16_kernel / conv_halo64_bf16_kernel / conv_bf16_pph_kernel wait on hand-counted
    `s_waitcnt vmcnt(N)`; a compiler bump that adds a scratch spill or splits a store between the counted operations must fail HERE, at build time,
    not as a flaky bit-equality test on the GPU.  (Cross-compiles csrc/conv_bf16.hip to assembly: ~25 s, no GPU.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
