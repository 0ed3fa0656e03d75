#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for ph in 0 1 2 4 7 9 12 15 31; do
  AMT_DECODE_PHASE=$ph timeout -k 10 300 python bench.py --no_roofline --no_cpu_baseline --steps 3 > gpurun_out/r02/bench_pm$ph.json 2> gpurun_out/r02/bench_pm$ph.err; echo "mask=$ph rc=$? $(cut -c50-140 gpurun_out/r02/bench_pm$ph.json)"
done
AMT_DECODE_PHASE=15 timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
