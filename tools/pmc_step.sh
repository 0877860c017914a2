#!/bin/bash
# HBM traffic of K1 as the training step runs it (nb_count_terms_kernel on a side stream, then nb_rows_kernel<..., PRE>):
# FETCH_SIZE and WRITE_SIZE of a short bench.py run, one pass each, nothing but --pmc.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_step
rm -rf $out && mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $out/$ctr -o p -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-ate > $out/$ctr.log 2>&1 || { tail -5 $out/$ctr.log; exit 3; }
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  python tools/parse_pmc.py $out/$ctr nb_rows
  python tools/parse_pmc.py $out/$ctr nb_count_terms
  python tools/parse_pmc.py $out/$ctr colsum
done > $out/summary.txt
cat $out/summary.txt
