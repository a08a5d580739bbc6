"""Which steps of a C3 window are slow?  Runs bench.py's C3 pass (fast-forward, warmup, K timed steps) and prints the timed steps whose
duration exceeds 1.3 x the median, with their index in the window (t3_step_times).  usage: python tools/step_times.py [steps] [warmup]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from chatterbox_vllm2_amd import engine as E
from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
args = type("A", (), dict(workload="c3", batch=32, max_model_len=1000, layers=30, steps=steps, warmup=warm, vocab=2454))()
ff, first, last = bench.plan_window(args)
eng = E.T3Engine(n_layers=30, text_vocab=2454, max_model_len=1000, max_seqs=32, gpu_memory_utilization=0.5, max_batched_rows=8192)
eng.load_tensors(list(synthetic_tensors(30, 2454, 1234))); eng.finalize()
res = bench.run_pass(eng, bench.build_requests(E, args, 0), synthetic_cond_emb(1), ff, warm, steps, torch.cuda.synchronize)
ms = np.asarray(res["step_ms"]); med = float(np.median(ms))
slow = [(int(i), round(float(ms[i]), 3)) for i in np.nonzero(ms > 1.3 * med)[0]]
print(json.dumps({"steps": steps, "fast_forward": ff, "median_ms": round(med, 4), "mean_ms": round(float(ms.mean()), 4), "slow_steps": slow,
                  "decode_step_of_first_timed": first}))
eng.close()
