O=gpurun_out/r06o; mkdir -p $O
python -m pytest tests -m gpu -q -s --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|^FAILED" $O/pytest.log | tail -8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CONFIG=c3 bash profiles/collect.sh r06_c3 all > $O/collect_c3.log 2>&1; grep "summarize rc" $O/collect_c3.log
cp gpurun_out/profiles_r06_c3/pmc_traffic_c3.json profiles/pmc_traffic_c3.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json
