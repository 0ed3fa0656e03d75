"""Summarise a rocprofv3 kernel_trace.csv of bench.py into a small JSON (runs on the GPU box)."""
import csv, glob, json, sys
import numpy as np
src, out = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
last = [i for i, n in enumerate(names) if "init_sequences" in n][-1]
seg = rows[last:]
res = {"n_kernels": len(seg), "span_ms": (seg[-1][1] - seg[0][0]) / 1e6}
agg = {}
for s, e, n in seg:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
res["kernels"] = {k: {"calls": v[0], "avg_us": round(v[1] / v[0], 3), "total_ms": round(v[1] / 1e3, 2)} for k, v in agg.items()}
selfk = [(e - s) / 1e3 for s, e, n in seg if "attn_decode_kernel<64, true" in n]
nl = 6
steps = len(selfk) // nl
a = np.array(selfk[: steps * nl]).reshape(steps, nl).mean(1)
res["self_attn_us_by_t"] = {str(t): round(float(a[t]), 2) for t in (0, 16, 64, 128, 200, 256, 384, 512, 768, steps - 1) if t < steps}
crossk = [(e - s) / 1e3 for s, e, n in seg if "attn_decode_kernel<64, false" in n]
res["cross_attn_us_median"] = float(np.median(crossk))
gk = [(e - s) / 1e3 for s, e, n in seg if "decode_gemm" in n]
res["decode_gemm_us_median"] = float(np.median(gk))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
