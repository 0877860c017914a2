#!/bin/bash
# One GPU visit: tests, bench line, kernel-trace profile, PMC passes for K1.  Run from the repo root on the GPU box.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -3 gpurun_out/gpu_tests.log
timeout -k 10 240 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 2; }
cat gpurun_out/bench.json
rm -rf gpurun_out/prof && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof.json 2> gpurun_out/prof.err || { tail -20 gpurun_out/prof.err; exit 3; }
cat gpurun_out/bench_prof.json
rm -rf gpurun_out/pmc_fetch && timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o r -- python3 tools/pmc_nb.py > gpurun_out/pmc_fetch.log 2>&1 || { tail -20 gpurun_out/pmc_fetch.log; exit 4; }
rm -rf gpurun_out/pmc_write && timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o r -- python3 tools/pmc_nb.py > gpurun_out/pmc_write.log 2>&1 || { tail -20 gpurun_out/pmc_write.log; exit 5; }
python tools/parse_pmc.py gpurun_out/pmc_fetch nb_rows > gpurun_out/pmc_summary.txt
python tools/parse_pmc.py gpurun_out/pmc_write nb_rows >> gpurun_out/pmc_summary.txt
cat gpurun_out/pmc_summary.txt
