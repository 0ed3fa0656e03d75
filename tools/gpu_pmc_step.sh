#!/bin/bash
# Round 3: hardware counters of the decode step's kernels, one rocprofv3 pass per counter group (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; --pmc never combined with --stats / sys traces).
#  * the skinny GEMMs and the sampling head: the REAL step of bench.py at config 2's launch shapes (32 clips, d_model 512); their
#    shapes do not depend on the sequence length, and rocprofv3 --pmc crashes (SIGSEGV in the tool right after HSA init, before the
#    first kernel) when bench.py runs at --seq 1024, so the passes run at --seq 192;
#  * the decode attention, whose bytes DO depend on the length: tools/pmc_attn.py, the kernels alone at config 2's launch shape
#    over positions t = 7, 15, ..., 1023 with six layer-sized K/V caches cycled.
# Usage on the GPU box:  bash tools/gpu_pmc_step.sh   -> gpurun_out/r03/pmc_{step,attn}_{FETCH_SIZE,WRITE_SIZE,SQ}.json
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03; export TMPDIR=/tmp
run() {   # $1 = tag, $2 = what (step | attn), $3... = counters
  tag=$1; what=$2; shift; shift
  if [ "$what" = step ]; then prog="bench.py --seq 192 --steps 1 --warmup 0 --no_roofline --no_cpu_baseline"; else prog="tools/pmc_attn.py"; fi
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/r03/pmc_${what}_$tag -o p -- python3 $prog \
      > gpurun_out/r03/pmc_${what}_$tag.line 2> gpurun_out/r03/pmc_${what}_$tag.err || { echo "pass $what $tag failed"; tail -5 gpurun_out/r03/pmc_${what}_$tag.err; return 1; }
  python tools/pmc_summary.py gpurun_out/r03/pmc_${what}_$tag gpurun_out/r03/pmc_${what}_$tag.json > /dev/null && rm -rf gpurun_out/r03/pmc_${what}_$tag
}
for what in step attn; do
  run FETCH_SIZE $what FETCH_SIZE && run WRITE_SIZE $what WRITE_SIZE && \
  run SQ $what SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
done
ls gpurun_out/r03/pmc_*.json
