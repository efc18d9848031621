#!/bin/bash
# On the GPU box (via gpurun): GPU parity tests, then the bench line.  bash tools/gpu_check.sh <tag> [pytest args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q -s "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^\[|passed|failed|error|rc=" $OUT/pytest.log | tail -40
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err; head -c 600 $OUT/bench.json
