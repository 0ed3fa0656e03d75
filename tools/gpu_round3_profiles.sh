#!/bin/bash
# round 3: the measurements that go to profiles/ (one gpurun call; every step writes under gpurun_out/r03/)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03; export TMPDIR=/tmp
O=gpurun_out/r03
python bench.py > $O/r03_bench_line_unprofiled.json 2> $O/bench_final.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 bench.py --steps 3 --no_cpu_baseline > $O/r03_bench_line_under_rocprofv3.json 2> $O/bench_under_rocprof.err; echo "rocprof bench rc=$?"
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
python tools/trace_by_grid.py $O/prof_bench $O/r03_bench_by_grid.json > /dev/null; rm -rf $O/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o f -- python3 tools/bench_forward.py > $O/forward_under_rocprof.json 2> $O/forward_under_rocprof.err; echo "rocprof fwd rc=$?"
cp $(find $O/prof_fwd -name "*kernel_stats.csv" | head -1) $O/r03_forward_kernel_stats.csv
python tools/trace_by_grid.py $O/prof_fwd $O/r03_forward_by_grid.json > /dev/null; rm -rf $O/prof_fwd
python tools/bench_modules.py 2>/dev/null | tail -1 > $O/r03_modules_bench.json
python tools/bench_rpr_prefill.py 2>/dev/null | tail -1 > $O/r03_rpr_prefill.json
python tools/bench_v2_lockstep.py 2>/dev/null | tail -1 > $O/r03_v2_lockstep.json
NB=32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v2 -o v -- python3 tools/prof_v2_batch.py > $O/prof_v2.log 2>&1
python tools/trace_by_grid.py $O/prof_v2 $O/r03_v2_lockstep_B32_by_grid.json > /dev/null; cp $(find $O/prof_v2 -name "*kernel_stats.csv" | head -1) $O/r03_v2_lockstep_B32_kernel_stats.csv; rm -rf $O/prof_v2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ep -o e -- python3 tools/bench_moe_ep.py --gpus 1 > $O/r03_moe_ep_1rank_rccl_under_rocprofv3.json 2> $O/prof_ep.err
cp $(find $O/prof_ep -name "*kernel_stats.csv" | head -1) $O/r03_moe_ep_1rank_kernel_stats.csv; rm -rf $O/prof_ep
python tools/bench_moe_ep.py --gpus 1 2>/dev/null | tail -1 > $O/r03_moe_ep_1rank_rccl.json
python tools/bench_moe_ep.py --gpus 1 --shared 2>/dev/null | tail -1 > $O/r03_moe_ep_1rank_rccl_shared.json
python tools/cpu_thread_sweep.py 512 2>/dev/null | tail -1 > $O/r03_cpu_baseline_thread_sweep.json
python tools/bench_families.py 2>/dev/null | tail -1 > $O/r03_families_bench.json
head -c 1500 $O/r03_bench_line_unprofiled.json; echo; cat $O/r03_v2_lockstep.json $O/r03_rpr_prefill.json $O/r03_cpu_baseline_thread_sweep.json
