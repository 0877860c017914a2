#!/bin/bash
# bench.py under each VAR=value given as arguments, e.g. tools/env_sweep.sh CRADLE_EVFLAGS=0x40000000 CRADLE_SCHED=1
set -o pipefail
for kv in "$@"; do
  echo "== $kv"
  env $kv timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f  cells/s %.0f  k1 %.1f us  gemm %.1f us' % (d['ms_per_step'], d['value'], d['roofline']['launch_us'], d['mfma']['largest_gemm']['launch_us']))" || exit 1
done
