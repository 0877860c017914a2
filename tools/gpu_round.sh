#!/bin/bash
# One GPU visit: tests, bench line, kernel-trace profile.  Run from the repo root on the GPU box.
# SKIP_TESTS=1 skips the tests; TESTS="tests/a.py tests/b.py" runs only those.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -u -m pytest ${TESTS:-tests} -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
  tail -3 gpurun_out/gpu_tests.log
fi
[ -n "$SKIP_BENCH" ] && exit 0
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 2; }
cat gpurun_out/bench.json
rm -rf gpurun_out/prof && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof.json 2> gpurun_out/prof.err || { tail -20 gpurun_out/prof.err; exit 3; }
cat gpurun_out/bench_prof.json
python tools/step_timeline.py gpurun_out/prof/r_kernel_trace.csv > gpurun_out/timeline.txt && cat gpurun_out/timeline.txt
