"""Print the prefill-side kernels of a rocprofv3 --stats run (run on the box): python tools/prefill_stats.py <dir>"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])))
for r in rows:
    n = r["Name"]
    if "pgemm" in n or "false>(t3::AttnArgs" in n or "rope_kv" in n or "row_rstd" in n or ("gemm_kernel" in n and int(r["Calls"]) <= 130):
        print("%-86s calls=%5s avg_us=%9.1f total_ms=%8.2f" % (n[:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
