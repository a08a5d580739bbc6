"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel class (run ON the GPU box; the raw CSVs are too big to ship).
usage: python tools/pmc_summarize.py <dir> <out.json> [COUNTER]"""
import csv, glob, json, sys, collections

d, out = sys.argv[1], sys.argv[2]
counter = sys.argv[3] if len(sys.argv) > 3 else "FETCH_SIZE"
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: [0, 0.0])
cols = None
with open(f) as fh:
    for r in csv.DictReader(fh):
        if cols is None:
            cols = list(r.keys())
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t3::", "")
        gy = r.get("Grid_Size_Y") or r.get("Grid_Size", "")
        key = f"{name}|gy={gy}"
        a = agg[key]; a[0] += 1; a[1] += float(r["Counter_Value"])
res = {"counter": counter, "columns": cols, "kernels": {k: {"launches": v[0], "sum": v[1], "avg": v[1] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in list(res["kernels"].items())[:8]}, indent=0)[:1500])
