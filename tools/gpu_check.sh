#!/bin/bash
# On the GPU box (via gpurun): GPU parity tests, then the bench line as the driver runs it.  bash tools/gpu_check.sh <tag> [pytest args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q -s --durations=10 "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^\[|passed|failed|error|rc=" $OUT/pytest.log | tail -60
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err; head -c 600 $OUT/bench.json
