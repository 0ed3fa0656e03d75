#!/bin/bash
# Round 3: matrix-pipe occupancy of the MFMA kernels of the teacher-forced path (the north star's "MFMA-busy against MI355X peak"):
# the prefill attention kernels alone at config 2's shapes (tools/bench_rpr_prefill.py: relative-position causal, plain causal, cross)
# and the whole forward (tools/bench_forward.py: the dense fp32 GEMM).  One counter pass, --pmc with --kernel-trace only.
# Usage on the GPU box:  bash tools/gpu_pmc_prefill.sh  -> gpurun_out/r03/pmc_prefill_MFMA.json, pmc_forward_MFMA.json
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03; export TMPDIR=/tmp
for what in prefill forward; do
  prog=tools/bench_rpr_prefill.py; [ $what = forward ] && prog=tools/bench_forward.py
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_${what}_MFMA -o p -- python3 $prog \
      > gpurun_out/r03/pmc_${what}_MFMA.line 2> gpurun_out/r03/pmc_${what}_MFMA.err || { echo "pass $what failed"; tail -5 gpurun_out/r03/pmc_${what}_MFMA.err; exit 1; }
  python tools/pmc_summary.py gpurun_out/r03/pmc_${what}_MFMA gpurun_out/r03/pmc_${what}_MFMA.json > /dev/null && rm -rf gpurun_out/r03/pmc_${what}_MFMA
done
python - <<'PY'
import json
for what in ("prefill", "forward"):
    d = json.load(open(f"gpurun_out/r03/pmc_{what}_MFMA.json"))
    for k, c in d.items():
        if "MFMA" in "".join(c) and c.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean", 0) > 0 and ("prefill" in k or "gemm_f32" in k):
            print(what, k[:70], {n: round(v["mean"]) for n, v in c.items()}, "dispatches", c["SQ_BUSY_CYCLES"]["dispatches"])
PY
