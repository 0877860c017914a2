#!/bin/bash
# The communication-stream stall of the two-bucket data-parallel step, traced: a one-rank RCCL group on the
# box's one GPU, 16 hardware queues, with 2 buckets (stalls) and with 1 (does not).  Kernel trace alone (the HIP API
# trace makes both forms host-bound at ~2 ms per step and hides the difference).
set -o pipefail
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
out=gpurun_out/ddp_trace
rm -rf $out && mkdir -p $out
for nb in 2 1; do
  export CRADLE_BUCKETS=$nb
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/k$nb -o t -- python3 tools/bench_ddp_one_rank.py 100 > $out/kprof_b$nb.txt 2>&1 || { tail -5 $out/kprof_b$nb.txt; exit 4; }
  grep "ms/step" $out/kprof_b$nb.txt
done
