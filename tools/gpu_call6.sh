#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
run() { echo -n "$1: "; env $1 timeout -k 10 200 python bench.py --no_roofline --no_cpu_baseline --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "X=1"
run "AMD_OPT_FLUSH=0"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=32"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=256"
run "DEBUG_CLR_MAX_BATCH_SIZE=1000"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"
run "DEBUG_HIP_KERNARG_COPY_OPT=0"
run "ROC_SYSTEM_SCOPE_SIGNAL=0"
run "AMT_STEPS_PER_GRAPH=32"
run "AMT_STEPS_PER_GRAPH=2"
run "X=2"
