"""Builds profiles/r03_pmc_decode_step.json from the passes of tools/gpu_pmc_step.sh (gpurun_out/r03/pmc_{step,attn}_*.json): per
kernel of the decode step its HBM-side traffic per launch (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, MI355X_MICROARCH.md
HBM section), the wave-cycle split of the SQ pass, and the sha256 of the kernel sources the passes were collected on -- bench.py
reports `traffic` for a kernel only while its source still hashes to the recorded value."""
import hashlib
import json
import os

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out", "r03")
CS = os.path.join(R, "video2music_amd", "csrc")


def sha(name):
    return hashlib.sha256(open(os.path.join(CS, name), "rb").read()).hexdigest()


def entries(what, keep):
    ld = lambda n: json.load(open(os.path.join(G, n)))
    F, W, S = ld(f"pmc_{what}_FETCH_SIZE.json"), ld(f"pmc_{what}_WRITE_SIZE.json"), ld(f"pmc_{what}_SQ.json")
    out = {}
    for k in sorted(F):
        if not any(t in k for t in keep):
            continue
        fe, wr = F[k]["FETCH_SIZE"], W.get(k, {}).get("WRITE_SIZE", {"mean": 0.0})
        e = {"dispatches": fe["dispatches"], "FETCH_SIZE_KiB_mean": round(fe["mean"], 2), "WRITE_SIZE_KiB_mean": round(wr["mean"], 2),
             "traffic_bytes_per_launch": round(2 * fe["mean"] * 1024 + wr["mean"] * 1024)}
        sq = S.get(k)
        if sq and "SQ_WAVE_CYCLES" in sq:
            wc = sq["SQ_WAVE_CYCLES"]["mean"]
            e["wave_cycle_split"] = {c: round(sq[c]["mean"] / wc, 4) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c in sq}
            e["SQ_mean"] = {c: round(v["mean"], 1) for c, v in sq.items()}
        out[k] = e
    return out


def main():
    step = entries("step", ("decode_gemm_kernel", "sample_fold_kernel", "sample_kernel"))
    attn = entries("attn", ("attn_decode_kernel",))
    alg = json.loads(open(os.path.join(G, "pmc_attn_FETCH_SIZE.line")).read().strip().splitlines()[-1])
    for k, e in attn.items():
        a = alg["self_algorithmic_bytes_per_launch"] if "<64, true" in k else alg["cross_algorithmic_bytes_per_launch"]
        e["algorithmic_bytes_per_launch"] = a
        e["traffic_over_algorithmic"] = round(e["traffic_bytes_per_launch"] / a, 4)
    out = {
        "source": {
            "skinny GEMMs, sampling head": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --seq 192 --steps 1 --warmup 0 --no_roofline "
                                           "--no_cpu_baseline: the captured decode step of config 2's model (32 clips, d_model 512, 6 layers); the launch "
                                           "shapes of these kernels do not depend on the sequence length (rocprofv3 --pmc crashes inside the tool when "
                                           "bench.py runs at --seq 1024)",
            "decode attention": "rocprofv3 --pmc <group> --kernel-trace -- python3 tools/pmc_attn.py: the shipped (LayerNorm-folded) kernels alone at config "
                                "2's launch shape (B=32, H=8, hd=64), positions t = 7, 15, ..., 1023, six layer-sized K/V caches cycled",
            "passes": "one per group: FETCH_SIZE | WRITE_SIZE | SQ (tools/gpu_pmc_step.sh)"},
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> traffic = "
                      "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per launch (counter unit KiB); means over all launches of a kernel",
        "kernels": dict(step, **attn),
        "kernel_source_sha256": {n: sha(n) for n in ("attn_decode.hip", "decode_gemm.hip", "sample.hip")},
    }
    json.dump(out, open(os.path.join(R, "profiles", "r03_pmc_decode_step.json"), "w"), indent=1)
    for k, e in out["kernels"].items():
        print(f"{k[:58]:58s} n={e['dispatches']:6d} traffic/launch {e['traffic_bytes_per_launch'] / 1e6:8.3f} MB  {e.get('traffic_over_algorithmic', '')}  {e.get('wave_cycle_split', '')}")


if __name__ == "__main__":
    main()
