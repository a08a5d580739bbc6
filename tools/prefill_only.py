"""Prefill-only run for the MFMA-counter pass: R prefill steps of 29 es prompts (T = 141, 2 CFG streams) = 8178
rows through the 30 layers -- the prefill-sized GEMM schedule (pgemm_kernel) and the unfused attention -- and nothing else.
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <out> -- python3 tools/prefill_only.py
(the program directly after `--`; a counter pass of its own, no other trace domains)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chatterbox_vllm2_amd import engine as E
from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tok = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer.json")))
p_es = assemble_prompt_ids(tok["es_mtl_ids"])
eng = E.T3Engine(n_layers=layers, text_vocab=2454, max_model_len=400, max_seqs=32, gpu_memory_utilization=0.3, max_batched_rows=8192, enforce_eager=True)
eng.load_tensors(synthetic_tensors(layers, 2454, 1234)); eng.finalize()
cond = synthetic_cond_emb(1)
n = 8192 // (2 * len(p_es))
times = []
for r in range(reps + 1):
    for i in range(n):
        eng.add_request(r * 100 + i, p_es, cond, E.make_sampling(max_tokens=2, ignore_eos=True, uid=i))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.step()
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    assert res.n_prefill_rows == 2 * n * len(p_es), res.n_prefill_rows
    for i in range(n):
        eng.abort(r * 100 + i)
rows = 2 * n * len(p_es)
flops = rows * 2.0 * (4 * 1024 * 1024 + 3 * 4096 * 1024) * layers
ms = min(times[1:])
print(json.dumps({"prefill_rows": rows, "layers": layers, "ms_best": round(ms, 3), "ms_all": [round(t, 2) for t in times],
                  "gemm_tflops": round(flops / ms / 1e9, 1), "frac_of_2500_tflops": round(flops / ms / 1e9 / 2500.0, 4)}))
