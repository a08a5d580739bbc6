set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -x -q -k "attention or single_utterance or block_boundaries or long_prompt or batch_invariance" > gpurun_out/t_k5.log 2>&1; tail -4 gpurun_out/t_k5.log
for i in 1 2; do
python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b_dyn_$i.json 2>&1
T3_ENGINE_LIB=$PWD/build_diag/static/libt3engine.so python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b_static_$i.json 2>&1
done
python bench.py --workload c4 --steps 1500 --warmup 50 --no-cpu-baseline > gpurun_out/b_c4_dyn.json 2>&1
T3_ENGINE_LIB=$PWD/build_diag/static/libt3engine.so python bench.py --workload c4 --steps 1500 --warmup 50 --no-cpu-baseline > gpurun_out/b_c4_static.json 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_dyn_*.json')+glob.glob('gpurun_out/b_static_*.json')+glob.glob('gpurun_out/b_c4_dyn.json')+glob.glob('gpurun_out/b_c4_static.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))
PY
