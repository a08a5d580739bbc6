"""C4 measurement (BASELINE.json configs[3]: benchmark-text-1/2/fr/zh continuous batching, batch = 128): the committed request stream
tests/golden/c4_requests.json (499 sentences of the reference's docs/benchmark-text-*.txt tokenised with the f2 tokenizer, per-request
output length G ~ U{200..800}, seed 7) through B slots, everything timed end to end (admission, chunked prefill interleaved with
decode, retirement).  One JSON line.  `--repeat n` submits the stream n times (distinct uids) for a longer steady state."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chatterbox_vllm2_amd import engine as E
from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=0, help="0 = the fixture's 128"); ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--layers", type=int, default=30); ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
c4 = json.load(open(os.path.join(ROOT, "tests", "golden", "c4_requests.json")))
slots = a.slots or c4["slots"]
eng = E.T3Engine(n_layers=a.layers, text_vocab=2454, max_model_len=c4["max_model_len"], max_seqs=slots, gpu_memory_utilization=0.6,
                 max_batched_rows=8192, enforce_eager=a.eager)
eng.load_tensors(synthetic_tensors(a.layers, 2454, 1234)); eng.finalize()
cond = synthetic_cond_emb(1)
total = 0; n = 0
for rep in range(a.repeat):
    for i, r in enumerate(c4["requests"]):
        eng.add_request(n, assemble_prompt_ids(r["text_ids"]), cond, E.make_sampling(max_tokens=r["max_tokens"], ignore_eos=True, uid=n, **c4["sampling"]))
        total += r["max_tokens"]; n += 1
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.run_until_done()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = eng.stats()
assert st.tokens_generated == total and st.kv_blocks_free == st.kv_blocks_total
print(json.dumps({"workload": f"C4: {n} requests (tests/golden/c4_requests.json x {a.repeat}: benchmark-text-1/2/fr/zh sentences, 13-387 text ids, "
                              f"200-800 output tokens each) through {slots} slots, {a.layers} layers, end to end",
                  "speech_tokens": total, "seconds": round(dt, 3), "speech_tokens_per_s": round(total / dt, 1), "steps": st.steps,
                  "decode_only_steps": st.decode_steps, "prefill_rows": st.prefill_rows, "audio_seconds_per_second": round(total / 25.0 / dt, 1),
                  "mean_ctx_decode": round(st.sum_ctx_decode / max(1, st.decode_rows), 1),
                  "step_hbm_frac": round(st.algo_bytes_decode / max(1e-9, st.gpu_ms_decode * 1e-3) / 8e12, 4),
                  "ms_per_decode_only_step": round(st.gpu_ms_decode / max(1, st.decode_steps), 3),
                  "ms_per_mixed_step": round((st.gpu_ms_total - st.gpu_ms_decode) / max(1, st.steps - st.decode_steps), 3),
                  "prefill_rows_per_mixed_step": round(st.prefill_rows / max(1, st.steps - st.decode_steps), 1)}))
eng.close()
