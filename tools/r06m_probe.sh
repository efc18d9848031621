O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests/test_lpips_gpu.py tests/test_c3_gpu.py tests/test_fp64_anchor_gpu.py::test_c3_free_running_branches_vs_fp64_accumulation tests/test_timed_size_oracle_gpu.py::test_c3_as_timed_full_size_teacher_forced_vs_cpu_oracle tests/test_bf16_engine_gpu.py -m gpu -q -s > $O/t.log 2>&1; grep -E "^\[C3|^\[late|^\[fp64|passed|failed|Error|assert " $O/t.log | cut -c1-700 | tail -14
python tools/soak_c3.py 100 2>&1 | grep "step 50\|step 100"
FACEOFF_LPIPS_LATE_HEADS=0 python tools/soak_c3.py 100 2>&1 | grep "step 50\|step 100"
