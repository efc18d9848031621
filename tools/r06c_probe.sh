O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_disc_gpu.py tests/test_gan_gpu.py tests/test_fp64_anchor_gpu.py tests/test_e2e_gpu.py tests/test_module_gpu.py tests/test_dropin_gpu.py -m gpu -q -s --durations=8 -x > $O/tests.log 2>&1; grep -E "^\[|passed|failed|Error|assert|^[0-9.]+s " $O/tests.log | cut -c1-1200 | tail -40
python tools/bench_gan.py 8 > $O/bench_gan.txt 2>&1; cat $O/bench_gan.txt
FACEOFF_INSTNORM_ONE_LAUNCH=1 python tools/bench_gan.py 8 > $O/bench_gan_old_in.txt 2>&1; cat $O/bench_gan_old_in.txt
FACEOFF_ALWAYS_PACK=1 python tools/bench_gan.py 8 > $O/bench_gan_always_pack.txt 2>&1; cat $O/bench_gan_always_pack.txt
python tools/step_ledger.py --gan --top 30 > $O/ledger_c5.txt 2>&1; head -34 $O/ledger_c5.txt
