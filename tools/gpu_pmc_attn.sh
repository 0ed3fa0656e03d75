set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r02/pmc_$c -o p -- python3 tools/pmc_attn.py > gpurun_out/r02/pmc_attn_alg.json 2> gpurun_out/r02/pmc_$c.err && python tools/pmc_summary.py gpurun_out/r02/pmc_$c gpurun_out/r02/pmc_attn_$c.json && rm -rf gpurun_out/r02/pmc_$c || exit 1
  PMC_PLAIN=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r02/pmcp_$c -o p -- python3 tools/pmc_attn.py > gpurun_out/r02/pmc_attn_alg_plain.json 2> gpurun_out/r02/pmcp_$c.err && python tools/pmc_summary.py gpurun_out/r02/pmcp_$c gpurun_out/r02/pmc_attn_plain_$c.json && rm -rf gpurun_out/r02/pmcp_$c || exit 1
done
cat gpurun_out/r02/pmc_attn_alg.json
