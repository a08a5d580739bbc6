set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "small_step or decode_attention" > gpurun_out/t_k6.log 2>&1; tail -6 gpurun_out/t_k6.log
python -m pytest tests/test_gpu_engine.py tests/test_gpu_golden.py -m gpu -x -q -k "single_utterance or stop_token or c1_english or c2_ or block_boundaries or long_prompt or llm_surface or run_ahead" > gpurun_out/t_k7.log 2>&1; tail -6 gpurun_out/t_k7.log
for f in 1 0; do
T3_FUSE_QKV_SMALL=$f python bench.py --batch 1 --max-model-len 400 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/b_b1_fuse$f.json 2>&1
T3_FUSE_QKV_SMALL=$f python bench.py --batch 2 --max-model-len 400 --steps 200 --warmup 10 --no-cpu-baseline --no-profile-pass > gpurun_out/b_b2_fuse$f.json 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_b?_fuse?.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'), d.get('kernel_ms_per_step_all_classes_evented'), d['e2e']['value'], d['e2e']['rtf_p50'])
PY
