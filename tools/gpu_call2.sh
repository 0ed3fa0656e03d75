#!/bin/bash
# round-2 second GPU call: PMC passes (csv), overlap experiment, skinny-GEMM in-kernel timeline
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 120 ./tools/ubench_overlap.bin > gpurun_out/r02/ubench_overlap.log 2>&1; echo "overlap rc=$?"
cat gpurun_out/r02/ubench_overlap.log
timeout -k 10 120 ./tools/ubench_chain.bin > gpurun_out/r02/ubench_chain.log 2>&1; echo "chain rc=$?"
cat gpurun_out/r02/ubench_chain.log
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r02/pmc_$tag -o pmc -- python3 bench.py --seq 192 --steps 1 --warmup 0 --no_roofline --no_cpu_baseline > gpurun_out/r02/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
  python tools/pmc_summary.py gpurun_out/r02/pmc_$tag gpurun_out/r02/pmc_$tag.json > gpurun_out/r02/pmc_${tag}_summary.txt 2>&1
  find gpurun_out/r02/pmc_$tag -name "*.csv" | head -3
  rm -rf gpurun_out/r02/pmc_$tag
done
cat gpurun_out/r02/pmc_*_summary.txt
