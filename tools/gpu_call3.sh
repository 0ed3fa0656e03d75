#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/r02/t_model_phase.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02/t_model_phase.log
for ph in 1 0; do
  AMT_DECODE_PHASE=$ph timeout -k 10 300 python bench.py --no_roofline --no_cpu_baseline --steps 3 > gpurun_out/r02/bench_phase$ph.json 2> gpurun_out/r02/bench_phase$ph.err; echo "bench phase=$ph rc=$?"
  cut -c1-220 gpurun_out/r02/bench_phase$ph.json
done
