set -o pipefail
export TMPDIR=/tmp
timeout -k 10 200 python tools/bench_iwelbo.py 2>&1 | grep IWELBO
rm -rf gpurun_out/prof_iw && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_iw -o a -- python3 tools/bench_iwelbo.py > gpurun_out/prof_iw.log 2>&1 || { tail -20 gpurun_out/prof_iw.log; exit 3; }
grep IWELBO gpurun_out/prof_iw.log
