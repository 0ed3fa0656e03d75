#!/bin/bash
# round-2 first GPU call: full GPU suite, baseline bench, self-launched 2-rank rehearsal, PMC passes on the decode step
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r02/tests_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/tests_gpu.log
tail -5 gpurun_out/r02/tests_gpu.log
python bench.py > gpurun_out/r02/bench_base.json 2> gpurun_out/r02/bench_base.err; echo "bench rc=$?"
AMT_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r02/bench_2rank_gloo.json 2> gpurun_out/r02/bench_2rank_gloo.err; echo "bench2 rc=$?"
cat gpurun_out/r02/bench_2rank_gloo.json
# PMC passes (short generate: 192 tokens) — separate passes per counter group
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/r02/pmc_$tag -o pmc -- python3 bench.py --seq 192 --steps 1 --warmup 0 --no_roofline --no_cpu_baseline > gpurun_out/r02/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
  python tools/pmc_summary.py gpurun_out/r02/pmc_$tag gpurun_out/r02/pmc_$tag.json > gpurun_out/r02/pmc_${tag}_summary.txt 2>&1
  rm -rf gpurun_out/r02/pmc_$tag
done
cat gpurun_out/r02/bench_base.json | head -c 1500
