#!/bin/bash
OUT=gpurun_out/r05c; mkdir -p $OUT
python -m pytest tests/test_bf16_ops_gpu.py tests/test_lpips_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for l in "conv1_2" "conv2_1 fwd"; do ROUNDS=7 python tools/ab_bf16.py "$l" lines old:FACEOFF_H64_NO_LINES=1 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ab_kernels.txt
bash tools/ab_env.sh 3 "FACEOFF_H64_NO_LINES=1" 2>&1 | tee $OUT/ab_step.txt
