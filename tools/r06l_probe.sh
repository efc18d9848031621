O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests/test_lpips_gpu.py -m gpu -q -x -k "unpool" > $O/t1.log 2>&1; tail -3 $O/t1.log
FACEOFF_LPIPS_LATE_HEADS=1 python -m pytest tests/test_lpips_gpu.py tests/test_c3_gpu.py tests/test_timed_size_oracle_gpu.py::test_c3_as_timed_full_size_teacher_forced_vs_cpu_oracle -m gpu -q -x -s > $O/t2.log 2>&1; grep -E "^\[C3|passed|failed|Error" $O/t2.log | cut -c1-600 | tail -8
bash tools/ab_env.sh 3 FACEOFF_LPIPS_LATE_HEADS=1 2>&1 | tee $O/ab.txt
