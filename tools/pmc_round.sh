#!/bin/bash
# HBM traffic of K1 and K9 from the PMC counters: FETCH_SIZE and WRITE_SIZE in passes of their own (4 TCC counters per
# pass; FETCH_SIZE takes 3, WRITE_SIZE 2), each a run of its own with nothing but --pmc.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc
rm -rf $out && mkdir -p $out
for prog in pmc_nb pmc_sampler; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 90 rocprofv3 --pmc $ctr --output-format csv -d $out/${prog}_$ctr -o p -- python3 tools/$prog.py > $out/${prog}_$ctr.log 2>&1 || { tail -5 $out/${prog}_$ctr.log; exit 3; }
  done
done
python tools/parse_pmc.py $out/pmc_nb_FETCH_SIZE nb_rows > $out/summary.txt
python tools/parse_pmc.py $out/pmc_nb_WRITE_SIZE nb_rows >> $out/summary.txt
python tools/parse_pmc.py $out/pmc_sampler_FETCH_SIZE nb_sample >> $out/summary.txt
python tools/parse_pmc.py $out/pmc_sampler_FETCH_SIZE partial_sums >> $out/summary.txt
python tools/parse_pmc.py $out/pmc_sampler_WRITE_SIZE nb_sample >> $out/summary.txt
python tools/parse_pmc.py $out/pmc_sampler_WRITE_SIZE partial_sums >> $out/summary.txt
cat $out/summary.txt
