set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -x -q -k "attention or llm_surface or single_utterance or block_boundaries or long_prompt" > gpurun_out/t_k4.log 2>&1; tail -4 gpurun_out/t_k4.log
python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/b_c3_partial.json 2> gpurun_out/b_c3_partial.err; tail -c 900 gpurun_out/b_c3_partial.json
python bench.py --batch 1 --max-model-len 400 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/b_b1_partial.json 2>&1; tail -c 600 gpurun_out/b_b1_partial.json
T3_EAGER=1 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-profile-pass > gpurun_out/b_c3_eager.json 2>&1; head -c 300 gpurun_out/b_c3_eager.json
./tools/scratch/conc_probe 1024 146 4 > gpurun_out/conc_probe.txt 2>&1
./tools/scratch/conc_probe 512 73 4 >> gpurun_out/conc_probe.txt 2>&1
./tools/scratch/conc_probe 512 73 2 >> gpurun_out/conc_probe.txt 2>&1
cat gpurun_out/conc_probe.txt
./tools/gemm_clk 64 > gpurun_out/gemm_clk_64.txt 2>&1; ./tools/gemm_clk 2 > gpurun_out/gemm_clk_2.txt 2>&1; head -50 gpurun_out/gemm_clk_64.txt
