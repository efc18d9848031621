#!/bin/bash
# round 5, first GPU call: box facts, the new timed-size oracle tests (recording the bf16 bounds), the changed parity / DDP / bench tests, the bench line
OUT=gpurun_out/r05a; mkdir -p $OUT
{ nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max; free -g; } > $OUT/box.txt 2>&1
FACEOFF_RECORD_OBSERVED=$OUT/observed.json python -m pytest tests/test_timed_size_oracle_gpu.py -m gpu -q -s -x --durations=5 > $OUT/timed_size.log 2>&1
echo "timed-size rc=$?" >> $OUT/timed_size.log
grep -E "^\[|passed|failed|rc=|Error|assert" $OUT/timed_size.log | tail -20
python -m pytest tests/test_e2e_gpu.py tests/test_ddp_gpu.py tests/test_dropin_gpu.py tests/test_bench_gpu.py -m gpu -q -s --durations=8 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^\[|passed|failed|rc=|Error" $OUT/pytest.log | tail -30
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err; head -c 400 $OUT/bench.json
