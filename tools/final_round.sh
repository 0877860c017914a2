#!/bin/bash
# The side measurements of a round, after tools/gpu_round.sh: other shapes, phase probes, PMC traffic, sampler, ATE.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/other_shapes.jsonl
for sh in replogle adamson scaleup; do
  timeout -k 10 300 python bench.py --shape $sh --steps 50 --warmup 10 --no-ate --no-secondary >> gpurun_out/other_shapes.jsonl 2> gpurun_out/other.err || { tail -5 gpurun_out/other.err; exit 2; }
done
timeout -k 10 600 python tools/probe_sweep.py 200 > gpurun_out/phase_probes.txt 2>&1 || { tail -5 gpurun_out/phase_probes.txt; exit 3; }
bash tools/pmc_round.sh > gpurun_out/pmc_round.txt 2>&1 || { tail -5 gpurun_out/pmc_round.txt; exit 4; }
bash tools/pmc_step.sh > gpurun_out/pmc_step.txt 2>&1 || { tail -5 gpurun_out/pmc_step.txt; exit 5; }
{ timeout -k 10 120 python tools/bench_sampler.py 5000 237 128 0.0; timeout -k 10 120 python tools/bench_sampler.py 5000 237 128 3.0; timeout -k 10 120 python tools/bench_sampler.py 8000 237 128 0.0; } 2>&1 | grep -v amdgpu.ids > gpurun_out/sampler.txt
timeout -k 10 200 python tools/ate_gemm_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ate_gemm_shapes.txt
timeout -k 10 200 python tools/bench_ate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_ate.txt
timeout -k 10 120 python tools/gemm_ksweep2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_ksweep.txt
timeout -k 10 120 python tools/gemm_quant.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/gemm_ksweep.txt
timeout -k 10 120 python tools/bench_nb.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_nb.txt
echo finished
