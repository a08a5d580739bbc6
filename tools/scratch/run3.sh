set -x
mkdir -p gpurun_out
bash tools/run_profiles.sh r03a > gpurun_out/r03a_run.log 2>&1; tail -5 gpurun_out/r03a_run.log
T3_ENGINE_LIB=$PWD/build_diag/attn_clk/libt3engine.so python tools/attn_clk.py 250 1 > gpurun_out/r03a_attn_clk_b1.txt 2>&1; cat gpurun_out/r03a_attn_clk_b1.txt
T3_ENGINE_LIB=$PWD/build_diag/attn_clk/libt3engine.so python tools/attn_clk.py 560 32 > gpurun_out/r03a_attn_clk_b32.txt 2>&1; cat gpurun_out/r03a_attn_clk_b32.txt
python bench.py > gpurun_out/r03a_bench_default.json 2> gpurun_out/r03a_bench_default.err; tail -c 700 gpurun_out/r03a_bench_default.json
python bench.py --workload c4 --steps 1500 --warmup 50 --no-cpu-baseline > gpurun_out/r03a_bench_c4.json 2>&1; tail -c 300 gpurun_out/r03a_bench_c4.json
