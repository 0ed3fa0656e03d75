"""Builds profiles/r03_pmc_decode_step.json from the passes of tools/gpu_pmc_step.sh (gpurun_out/r03/pmc_step_*.json): per kernel
of the decode step its HBM-side traffic per launch (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, MI355X_MICROARCH.md HBM
section), the wave-cycle split of the SQ pass, and the sha256 of the kernel sources the passes were collected on -- bench.py reports
`traffic` for a kernel only while its source still hashes to the recorded value."""
import hashlib
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out", "r03")
CS = os.path.join(R, "video2music_amd", "csrc")


def sha(name):
    return hashlib.sha256(open(os.path.join(CS, name), "rb").read()).hexdigest()


def main():
    ld = lambda n: json.load(open(os.path.join(G, n)))
    F, W, S = ld("pmc_step_FETCH_SIZE.json"), ld("pmc_step_WRITE_SIZE.json"), ld("pmc_step_SQ.json")
    kernels = {}
    for k in sorted(F):
        if not any(t in k for t in ("attn_decode_kernel", "decode_gemm_kernel", "sample_fold_kernel", "sample_kernel")):
            continue
        fe, wr = F[k]["FETCH_SIZE"], W.get(k, {}).get("WRITE_SIZE", {"mean": 0.0})
        e = {"dispatches": fe["dispatches"], "FETCH_SIZE_KiB_mean": round(fe["mean"], 2), "WRITE_SIZE_KiB_mean": round(wr["mean"], 2),
             "traffic_bytes_per_launch": round(2 * fe["mean"] * 1024 + wr["mean"] * 1024)}
        sq = S.get(k)
        if sq and "SQ_WAVE_CYCLES" in sq:
            wc = sq["SQ_WAVE_CYCLES"]["mean"]
            e["wave_cycle_split"] = {c: round(sq[c]["mean"] / wc, 4) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c in sq}
            e["SQ_mean"] = {c: round(v["mean"], 1) for c, v in sq.items()}
        kernels[k] = e
    out = {
        "source": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no_roofline --no_cpu_baseline (config 2: 32 clips, "
                  "T = 1024, the captured decode step replayed 1023 times), one pass per group: SQ | FETCH_SIZE | WRITE_SIZE (tools/gpu_pmc_step.sh)",
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> traffic = "
                      "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per launch (counter unit KiB); means over all launches of a kernel in the generate",
        "kernels": kernels,
        "kernel_source_sha256": {n: sha(n) for n in ("attn_decode.hip", "decode_gemm.hip", "sample.hip")},
    }
    dst = os.path.join(R, "profiles", "r03_pmc_decode_step.json")
    json.dump(out, open(dst, "w"), indent=1)
    for k, e in kernels.items():
        print(f"{k[:60]:60s} n={e['dispatches']:6d} traffic/launch {e['traffic_bytes_per_launch'] / 1e6:8.3f} MB  {e.get('wave_cycle_split', '')}")


if __name__ == "__main__":
    main()
