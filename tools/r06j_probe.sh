O=gpurun_out/r06j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r06 all > $O/collect_c2.log 2>&1; tail -3 $O/collect_c2.log
CONFIG=c5 bash profiles/collect.sh r06_c5 all > $O/collect_c5.log 2>&1; tail -3 $O/collect_c5.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json
