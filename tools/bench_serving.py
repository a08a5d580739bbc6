"""C4-shaped measurement (BASELINE.json configs: continuous batching): N requests with ragged prompt and output lengths through
B slots, everything timed end to end (admission, chunked prefill interleaved with decode, retirement).  One JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chatterbox_vllm2_amd import engine as E
from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=128); ap.add_argument("--requests", type=int, default=512)
ap.add_argument("--layers", type=int, default=30); ap.add_argument("--max-model-len", type=int, default=1000)
a = ap.parse_args()
eng = E.T3Engine(n_layers=a.layers, text_vocab=2454, max_model_len=a.max_model_len, max_seqs=a.slots, gpu_memory_utilization=0.6,
                 max_batched_rows=8192, enforce_eager=False)
eng.load_tensors(synthetic_tensors(a.layers, 2454, 1234)); eng.finalize()
cond = synthetic_cond_emb(1)
rs = np.random.RandomState(7)
total = 0
for i in range(a.requests):
    n_text = int(rs.randint(20, 140)); g = int(rs.randint(200, 801))
    ids = assemble_prompt_ids([int(x) for x in rs.randint(3, 690, size=n_text)])
    g = min(g, a.max_model_len - len(ids) - 1); total += g
    eng.add_request(i, ids, cond, E.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, max_tokens=g, ignore_eos=True, uid=i))
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.run_until_done()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = eng.stats()
assert st.tokens_generated == total
print(json.dumps({"workload": f"{a.requests} requests (20-139 text ids, 200-800 tokens each) through {a.slots} slots, {a.layers} layers, end to end",
                  "speech_tokens": total, "seconds": round(dt, 3), "speech_tokens_per_s": round(total / dt, 1), "steps": st.steps,
                  "decode_only_steps": st.decode_steps, "prefill_rows": st.prefill_rows, "audio_seconds_per_second": round(total / 25.0 / dt, 1)}))
