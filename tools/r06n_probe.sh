O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests/test_lpips_gpu.py -m gpu -q -k "unpool or late" -s > $O/t.log 2>&1; grep -E "^\[late|passed|failed|Error" $O/t.log | cut -c1-300
bash tools/ab_env.sh 3 FACEOFF_DIAG_NO_REPACK=1 2>&1 | tee $O/ab_norepack.txt
