set -x
python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "groups or run_ahead or batch_invariance" > gpurun_out/t_k3.log 2>&1; tail -5 gpurun_out/t_k3.log
for g in 1 2; do
  T3_GROUPS=$g python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/b_c3_g$g.json 2> gpurun_out/b_c3_g$g.err; tail -c 1500 gpurun_out/b_c3_g$g.json
  T3_GROUPS=$g python bench.py --workload c4 --steps 1500 --warmup 50 --no-cpu-baseline > gpurun_out/b_c4_g$g.json 2> gpurun_out/b_c4_g$g.err; tail -c 1500 gpurun_out/b_c4_g$g.json
done
