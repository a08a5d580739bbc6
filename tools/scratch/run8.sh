mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -x -q -k "decode_attention or single_utterance or block_boundaries or long_prompt or batch_invariance or stop_token" > gpurun_out/t_k8.log 2>&1; tail -3 gpurun_out/t_k8.log
for v in prod kpro r2form prod kpro r2form; do
  if [ $v = prod ]; then L=""; else L="T3_ENGINE_LIB=$PWD/build_diag/$v/libt3engine.so"; fi
  env $L python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/b8_$v.json 2>&1
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/b8_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))
PY
done
