mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_golden.py -m gpu -x -q -k "gemm or gate_up or silu or norm_folded or single_utterance or batch_invariance or c2_full_depth or c3_multilingual" > gpurun_out/t_k9.log 2>&1; tail -3 gpurun_out/t_k9.log
for v in prod ew0 prod ew0; do
  if [ $v = prod ]; then L=""; else L="T3_ENGINE_LIB=$PWD/build_diag/$v/libt3engine.so"; fi
  env $L python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-e2e --no-profile-pass > gpurun_out/b9_$v.json 2>&1
  env $L python bench.py --batch 1 --max-model-len 400 --steps 150 --warmup 10 --no-cpu-baseline --no-e2e --no-profile-pass > gpurun_out/b9_b1_$v.json 2>&1
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/b9_$v.json') if l.startswith('{')][-1]); e=json.loads([l for l in open('gpurun_out/b9_b1_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], 'b1', e['value'], e['ms_per_step'])
PY
done
./tools/gemm_clk 64 > gpurun_out/gemm_clk_64_ew4.txt 2>&1; grep -E "^M=|workgroups|epilogue|lifetime" gpurun_out/gemm_clk_64_ew4.txt
