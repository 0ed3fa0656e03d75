#!/bin/bash
# round 2: full GPU suite + the measurements that go to profiles/
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r02/tests_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02/tests_gpu_final.log
python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_bench -o b -- python3 bench.py --steps 3 --no_cpu_baseline > gpurun_out/r02/bench_under_rocprof.json 2> gpurun_out/r02/bench_under_rocprof.err; echo "rocprof bench rc=$?"
cp $(find gpurun_out/r02/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r02/r02_bench_kernel_stats.csv
python tools/trace_by_grid.py gpurun_out/r02/prof_bench gpurun_out/r02/r02_bench_by_grid.json; rm -rf gpurun_out/r02/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_fwd -o f -- python3 tools/bench_forward.py > gpurun_out/r02/forward_under_rocprof.json 2> gpurun_out/r02/forward_under_rocprof.err; echo "rocprof fwd rc=$?"
cp $(find gpurun_out/r02/prof_fwd -name "*kernel_stats.csv" | head -1) gpurun_out/r02/r02_forward_kernel_stats.csv
python tools/trace_by_grid.py gpurun_out/r02/prof_fwd gpurun_out/r02/r02_forward_by_grid.json; rm -rf gpurun_out/r02/prof_fwd
python tools/bench_forward.py 2>/dev/null | tail -1 > gpurun_out/r02/forward_unprofiled.json
python tools/bench_modules.py 2>/dev/null | tail -1 > gpurun_out/r02/r02_modules_bench.json
python tools/bench_v2.py 2>/dev/null | tail -1 > gpurun_out/r02/r02_v2_bench.json
python tools/bench_rpr_prefill.py 2>/dev/null | tail -1 > gpurun_out/r02/r02_rpr_prefill.json
NB=32 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_v2 -o v -- python3 tools/prof_v2_batch.py > gpurun_out/r02/prof_v2.log 2>&1
python tools/trace_by_grid.py gpurun_out/r02/prof_v2 gpurun_out/r02/r02_v2_lockstep_B32_by_grid.json; cp $(find gpurun_out/r02/prof_v2 -name "*kernel_stats.csv" | head -1) gpurun_out/r02/r02_v2_lockstep_B32_kernel_stats.csv; rm -rf gpurun_out/r02/prof_v2
python tools/bench_v2_phases.py 2>/dev/null | tail -1 > gpurun_out/r02/r02_v2_phases.json
NB=32 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/prof_v2t -o v -- python3 tools/prof_v2_batch.py > gpurun_out/r02/prof_v2t.log 2>&1
python tools/trace_v2_steps.py gpurun_out/r02/prof_v2t gpurun_out/r02/r02_v2_step_timeline.json > gpurun_out/r02/r02_v2_step_timeline.txt; rm -rf gpurun_out/r02/prof_v2t
python tools/bench_families.py 2>/dev/null | tail -1 > gpurun_out/r02/r02_families_bench.json
cat gpurun_out/r02/forward_unprofiled.json gpurun_out/r02/r02_rpr_prefill.json
head -c 2500 gpurun_out/r02/bench_final.json
