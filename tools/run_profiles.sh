#!/bin/bash
# Round profiles (run ON the GPU box from the repo root; every PMC pass is a run of its own with --kernel-trace only):
#   tools/run_profiles.sh <tag>   ->  gpurun_out/<tag>_*  (copy what is to be kept into profiles/)
set -o pipefail
TAG=${1:-r02}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace + stats of the default bench workload, and two single steps of it as a timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile-pass > $O/${TAG}_trace_bench.log 2>&1
cp $(ls $O/p1/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_c3.csv
python3 $R/tools/trace_gaps.py $O/p1 > $O/${TAG}_step_timeline.json
rm -rf $O/p1
# (2) HBM traffic: FETCH_SIZE per kernel class
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p2 -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile-pass > $O/${TAG}_pmc_fetch.log 2>&1
python3 $R/tools/pmc_summarize.py $O/p2 $O/${TAG}_pmc_fetch_size_by_kernel.json FETCH_SIZE > /dev/null   # then, in the build container: python tools/update_traffic.py profiles/${TAG}_pmc_fetch_size_by_kernel.json
rm -rf $O/p2
# (3) MFMA counters of the prefill GEMMs (prefill-only program)
rocprofv3 -L > $O/${TAG}_counters_available.txt 2>&1
C=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do grep -q "$c" $O/${TAG}_counters_available.txt && C="$C $c"; done
echo "counters:$C" > $O/${TAG}_pmc_mfma.log
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p3 -- python3 $R/tools/prefill_only.py 3 >> $O/${TAG}_pmc_mfma.log 2>&1
for c in $C; do python3 $R/tools/pmc_summarize.py $O/p3 $O/${TAG}_pmc_${c}.json $c > /dev/null; done
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$O/p3/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t3::", "")
    agg[n][0] += 1; agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
json.dump({k: {"launches": v[0], "avg_us": v[1] / v[0] / 1e3} for k, v in agg.items()}, open("$O/${TAG}_pmc_mfma_kernel_durations.json", "w"), indent=1)
PY
rm -rf $O/p3
head -c 600 $O/${TAG}_pmc_mfma.log; ls $O | grep ${TAG}_
