#!/bin/bash
OUT=gpurun_out/r05b; mkdir -p $OUT
python -m pytest tests/test_fullsize_gpu.py::test_winograd_plane_split_experiment_is_bit_equal_to_the_default_path tests/test_lpips_gpu.py tests/test_gan_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
bash tools/ab_env_c2.sh 3 "FACEOFF_WINO_PLANE_SPLIT=1" > $OUT/ab_plane_split.txt 2>&1; cat $OUT/ab_plane_split.txt
