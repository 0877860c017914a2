#!/bin/bash
# Unprofiled experiments on the two-bucket stall (tools/bench_ddp_one_rank.py, EXPERIMENT=...), 16 hardware queues.
set -o pipefail
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/ddp_experiments.txt
: > $out
run() { echo "== $*" | tee -a $out; env "$@" timeout -k 10 100 python3 tools/bench_ddp_one_rank.py 200 2>&1 | grep "ms/step" | tee -a $out || exit 3; }
for a in "$@"; do run $a; done
