#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
python tools/bench_rpr_prefill.py 2>&1 | tail -1 | tee gpurun_out/r02/rpr_prefill_lpt.json
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "prefill or attention" 2>&1 | tail -2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r02/pmc_prefill -o pmc -- python3 tools/bench_rpr_prefill.py > gpurun_out/r02/pmc_prefill.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py gpurun_out/r02/pmc_prefill gpurun_out/r02/pmc_prefill.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02/pmc_prefill.json"))
for k,v in d.items():
    if "attn_prefill" in k:
        print(k, {c: round(x["mean"]) for c,x in v.items()})
PY
rm -rf gpurun_out/r02/pmc_prefill
