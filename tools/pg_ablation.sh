cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in nomfma nodma; do
  T3_ENGINE_LIB=$R/build_diag/libt3_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$v -- python3 $R/tools/prefill_only.py 2 > $R/gpurun_out/pg_$v.log 2>&1
  f=$(find $R/gpurun_out/prof_$v -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/pg_${v}_stats.csv; rm -rf $R/gpurun_out/prof_$v
done
