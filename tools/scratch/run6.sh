set -x
mkdir -p gpurun_out
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b_prod.json 2>&1
T3_ENGINE_LIB=$PWD/build_diag/dry/libt3engine.so python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b_dry.json 2>&1
T3_ATTN_WAVES=8 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b_w8.json 2>&1
python - <<'PY'
import json,glob
for f in ['b_prod','b_dry','b_w8']:
    d=json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1]); print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'), d.get('kernel_ms_per_step_all_classes_evented'))
PY
