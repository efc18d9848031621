O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_gan_gpu.py::test_two_rank_gan_iterations_vs_oracle tests/test_fp64_anchor_gpu.py tests/test_w42_gpu.py -m gpu -q -s --durations=8 > $O/tests.log 2>&1; grep -E "^\[|passed|failed|Error|assert|^[0-9.]+s " $O/tests.log | cut -c1-1200 | tail -40
python tools/step_ledger.py --bf16 --lpips > $O/ledger_c3.txt 2>&1; head -3 $O/ledger_c3.txt
python tools/step_ledger.py --gan > $O/ledger_c5.txt 2>&1; head -3 $O/ledger_c5.txt
python tools/step_breakdown.py --bf16 --lpips > $O/breakdown_c3.txt 2>&1; head -40 $O/breakdown_c3.txt
python tools/bench_gan.py 8 > $O/bench_gan.txt 2>&1; cat $O/bench_gan.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/tl5 -o tl -- python3 tools/bench_gan.py 8 > $O/tl5.log 2>&1; python tools/timeline.py $O/tl5 3 > $O/timeline_c5.txt 2>&1; cat $O/timeline_c5.txt
rm -rf $O/tl5
