"""Builds profiles/r02_pmc_attn_traffic.json from the outputs of tools/gpu_pmc_attn.sh (gpurun_out/r02/pmc_attn_*.json) and
stamps it with the sha256 of the attn_decode.hip it was collected on (bench.py reports `traffic` only while that still matches)."""
import hashlib, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out", "r02")
ld = lambda n: json.load(open(os.path.join(G, n)))
f, w, fp, wp, alg = ld("pmc_attn_FETCH_SIZE.json"), ld("pmc_attn_WRITE_SIZE.json"), ld("pmc_attn_plain_FETCH_SIZE.json"), ld("pmc_attn_plain_WRITE_SIZE.json"), ld("pmc_attn_alg.json")


def ent(F, W, k, algb):
    fe, wr = F[k]["FETCH_SIZE"], W[k]["WRITE_SIZE"]
    tr = 2 * fe["mean"] * 1024 + wr["mean"] * 1024
    return {"kernel": k, "dispatches": fe["dispatches"], "FETCH_SIZE_KiB_mean": fe["mean"], "WRITE_SIZE_KiB_mean": wr["mean"],
            "traffic_bytes_per_launch": tr, "algorithmic_bytes_per_launch": algb, "traffic_over_algorithmic": tr / algb}


sa, ca = alg["self_algorithmic_bytes_per_launch"], alg["cross_algorithmic_bytes_per_launch"]
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/gpu_pmc_attn.sh) of tools/pmc_attn.py: the decode-attention kernels of the shipped (LayerNorm-folded) step at the config-2 launch shape (B=32,H=8,hd=64), positions t=7,15,...,1023, six layer-sized K/V caches cycled",
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes (counter unit KiB)",
    "self_attn": ent(f, w, "attn_decode_kernel<64, true, true, 2, 2>", sa),
    "cross_attn": ent(f, w, "attn_decode_kernel<64, false, true, 1, 2>", ca),
    "plain_variant_same_method": {"self_attn": ent(fp, wp, "attn_decode_kernel<64, true, true, 0, 2>", sa), "cross_attn": ent(fp, wp, "attn_decode_kernel<64, false, true, 0, 2>", ca)},
    "kernel_source_sha256": hashlib.sha256(open(os.path.join(R, "video2music_amd", "csrc", "attn_decode.hip"), "rb").read()).hexdigest(),
    "round2_note": "re-collected after the last round-2 change of attn_decode.hip (tools/pmc_attn_json.py).  The PMC pass over the REAL decode step (profiles/r02_pmc_decode_step_FETCH_SIZE.json, bench.py --seq 192 under rocprofv3 --pmc FETCH_SIZE, an earlier source) gave 19555.6 KiB per cross-attention launch.  bench.py reports `traffic` from this file only while the sha256 of video2music_amd/csrc/attn_decode.hip matches kernel_source_sha256, null otherwise.",
}
json.dump(out, open(os.path.join(R, "profiles", "r02_pmc_attn_traffic.json"), "w"), indent=1)
print(out["self_attn"]["traffic_over_algorithmic"], out["cross_attn"]["traffic_over_algorithmic"], out["kernel_source_sha256"][:12])
