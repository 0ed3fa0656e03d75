"""Group a rocprofv3 kernel_trace.csv by (kernel, grid) -> calls / avg us (runs on the GPU box)."""
import csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
agg = {}
with open(f) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = f'{name} grid=({r["Grid_Size_X"]},{r["Grid_Size_Y"]},{r["Grid_Size_Z"]}) wg={r["Workgroup_Size_X"]}'
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
res = {k: {"calls": v[0], "avg_us": round(v[1] / v[0], 2), "total_ms": round(v[1] / 1e3, 3)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
json.dump(res, open(out, "w"), indent=1)
