"""Summarise one decode step of a rocprofv3 kernel trace: per-kernel durations and inter-kernel gaps (run on the box)."""
import csv, glob, sys, json, collections
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
samp = [i for i, r in enumerate(rows) if "sampler" in r["Kernel_Name"]]
out = {}
for which in (len(samp) // 4, len(samp) // 2):
    i0, i1 = samp[which], samp[which + 1]
    seg = rows[i0 + 1:i1 + 1]
    kt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    span = int(rows[i1]["End_Timestamp"]) - int(rows[i0]["End_Timestamp"])
    gaps = [int(seg[j]["Start_Timestamp"]) - int(seg[j - 1]["End_Timestamp"]) for j in range(1, len(seg))]
    first_gap = int(seg[0]["Start_Timestamp"]) - int(rows[i0]["End_Timestamp"])
    per = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t3::", "")[:40]
        per[n][0] += 1; per[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out[f"step@{which}"] = {"kernels": len(seg), "span_us": span / 1e3, "kernel_time_us": kt / 1e3, "sum_gaps_us": sum(gaps) / 1e3,
                             "gap_after_prev_sampler_us": first_gap / 1e3, "max_gap_us": max(gaps) / 1e3,
                             "per_kernel_us": {k: [v[0], round(v[1] / 1e3, 1)] for k, v in per.items()}}
print(json.dumps(out, indent=1))
