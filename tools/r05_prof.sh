#!/bin/bash
# round 5: rocprofv3 evidence for config 3 (head stream folded now) and config 5 (first counter passes)
CONFIG=c3 bash profiles/collect.sh r05_c3 all > gpurun_out/collect_r05_c3.log 2>&1; tail -3 gpurun_out/collect_r05_c3.log
CONFIG=c5 bash profiles/collect.sh r05_c5 all > gpurun_out/collect_r05_c5.log 2>&1; tail -3 gpurun_out/collect_r05_c5.log
