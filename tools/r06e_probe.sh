O=gpurun_out/r06e; mkdir -p $O
FACEOFF_HIP_LIB=$PWD/faceoff_amd/csrc/variants/lib_stamp.so python tools/stamp_h64.py 2>&1 | grep -v amdgpu.ids | tee $O/stamp_h64.txt
python -m pytest tests/test_lpips_gpu.py tests/test_bf16_ops_gpu.py tests/test_c3_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 50 --warmup 5 --no-x6-leg --no-direct-leg --no-h2d-leg --no-kernel-events --c3-sustained 0 --no-cpu-baseline > $O/bench_free.json 2> $O/bench_free.err; echo "free rc=$?"
python bench.py --steps 50 --warmup 5 --no-x6-leg --no-direct-leg --no-h2d-leg --no-kernel-events --c3-sustained 0 --host-cpus 2 > $O/bench_2cpu.json 2> $O/bench_2cpu.err; echo "2cpu rc=$?"
python - <<'PY'
import json
for n in ("free","2cpu"):
    d=json.loads(open(f"gpurun_out/r06e/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, "C2", d["ms_per_step"], "C3", d["c3"]["ms_per_step"], "C3 vqvae-only bf16", d["c3"]["vqvae_only_bf16"]["ms_per_step"], "C5", d["c5"]["ms_per_iteration"], d.get("host"))
PY
