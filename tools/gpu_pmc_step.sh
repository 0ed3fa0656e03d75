#!/bin/bash
# Round 3: hardware counters of the REAL decode step at config 2's full shape (32 clips, T = 1024), one rocprofv3 pass per
# counter group (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never combined with --stats / sys traces).
# Usage on the GPU box:  bash tools/gpu_pmc_step.sh   -> gpurun_out/r03/pmc_step_{SQ,FETCH_SIZE,WRITE_SIZE}.json
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03; export TMPDIR=/tmp
run() {   # $1 = tag, $2... = counters
  tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/r03/pmc_$tag -o p -- python3 bench.py --steps 1 --warmup 0 --no_roofline --no_cpu_baseline \
      > gpurun_out/r03/pmc_$tag.line 2> gpurun_out/r03/pmc_$tag.err || { echo "pass $tag failed"; tail -5 gpurun_out/r03/pmc_$tag.err; return 1; }
  python tools/pmc_summary.py gpurun_out/r03/pmc_$tag gpurun_out/r03/pmc_step_$tag.json && rm -rf gpurun_out/r03/pmc_$tag
}
run SQ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD && \
run FETCH_SIZE FETCH_SIZE && \
run WRITE_SIZE WRITE_SIZE
