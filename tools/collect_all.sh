#!/bin/bash
# rocprofv3 evidence for the three benched configurations in one gpurun call:  bash tools/collect_all.sh r05
# (then copy gpurun_out/profiles_<tag>*/ into profiles/)
TAG=${1:-r05}
bash profiles/collect.sh $TAG all > gpurun_out/collect_$TAG.log 2>&1; tail -2 gpurun_out/collect_$TAG.log
CONFIG=c3 bash profiles/collect.sh ${TAG}_c3 all > gpurun_out/collect_${TAG}_c3.log 2>&1; tail -2 gpurun_out/collect_${TAG}_c3.log
CONFIG=c5 bash profiles/collect.sh ${TAG}_c5 all > gpurun_out/collect_${TAG}_c5.log 2>&1; tail -2 gpurun_out/collect_${TAG}_c5.log
