#!/bin/bash
# C4 profiles (run ON the GPU box from the repo root): kernel stats of the continuous-batching stream and FETCH_SIZE of its kernels.
#   tools/run_profiles_c4.sh <tag>  ->  gpurun_out/<tag>_c4_*
set -o pipefail
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc1 -- python3 $R/bench.py --workload c4 --steps 600 --warmup 20 --no-cpu-baseline > $O/${TAG}_c4_trace_bench.log 2>&1
cp $(ls $O/pc1/*/*kernel_stats.csv | head -1) $O/${TAG}_c4_kernel_stats.csv
rm -rf $O/pc1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pc2 -- python3 $R/bench.py --workload c4 --steps 300 --warmup 20 --no-cpu-baseline > $O/${TAG}_c4_pmc_fetch.log 2>&1
python3 $R/tools/pmc_summarize.py $O/pc2 $O/${TAG}_c4_pmc_fetch_size_by_kernel.json FETCH_SIZE > /dev/null
rm -rf $O/pc2
head -12 $O/${TAG}_c4_kernel_stats.csv
