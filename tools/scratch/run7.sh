mkdir -p gpurun_out
for v in prod dry dry_nokv nokv prod; do
  if [ $v = prod ]; then L=""; else L="T3_ENGINE_LIB=$PWD/build_diag/$v/libt3engine.so"; fi
  env $L python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b7_$v.json 2>&1
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/b7_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))
PY
done
