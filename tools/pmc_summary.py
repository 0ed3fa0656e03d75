"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (runs on the GPU box)."""
import csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            k = (name.split("(")[0], r["Counter_Name"])
            a = res.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
js = {}
for (k, c), (n, tot) in sorted(res.items()):
    js.setdefault(k, {})[c] = {"dispatches": n, "mean": tot / n, "sum": tot}
json.dump(js, open(out, "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"], 1) for c, v in d.items()} for k, d in js.items() if "decode" in k or "sample" in k}))
