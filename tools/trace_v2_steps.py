"""Timeline of the lockstep V1/V2 step from a rocprofv3 kernel_trace.csv (runs on the GPU box): a step = the launches from one
v2_decide_kernel to the next; per step span, summed kernel time and summed gaps; per launch slot of the step its kernel, mean
duration and mean gap to the previous launch (graph replays only: the median step is reported)."""
import csv, glob, json, sys
import numpy as np
src, out = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
                     r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
cuts = [i for i, r in enumerate(rows) if "v2_decide" in r[2]]
steps = [rows[cuts[i] + 1: cuts[i + 1] + 1] for i in range(len(cuts) - 1)]
n = int(np.median([len(s) for s in steps]))
steps = [s for s in steps if len(s) == n]
spans = np.array([(s[-1][1] - s[0][0]) / 1e3 for s in steps])
order = np.argsort(spans)
keep = [steps[i] for i in order[: max(1, len(order) * 3 // 4)]]           # drop the slowest quarter (eager steps, captures)
busy = np.array([sum(e - b for b, e, _, _ in s) / 1e3 for s in keep])
span = np.array([(s[-1][1] - s[0][0]) / 1e3 for s in keep])
res = {"launches_per_step": n, "steps_used": len(keep), "span_us_median": float(np.median(span)), "kernel_time_us_median": float(np.median(busy)),
       "gaps_us_median": float(np.median(span - busy)), "slots": []}
for k in range(n):
    d = np.array([(s[k][1] - s[k][0]) / 1e3 for s in keep])
    g = np.array([(s[k][0] - s[k - 1][1]) / 1e3 for s in keep]) if k else np.zeros(len(keep))
    res["slots"].append({"kernel": keep[0][k][2], "grid": keep[0][k][3], "us": round(float(np.median(d)), 2), "gap_before_us": round(float(np.median(g)), 2)})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "slots"}))
for s in res["slots"]:
    print(f'{s["kernel"][:60]:60s} grid {s["grid"]:>8s}  {s["us"]:6.2f} us  gap {s["gap_before_us"]:5.2f}')
