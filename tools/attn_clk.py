"""Diagnostic: phase timeline of the attention kernel inside the real engine (needs a -DT3_ATTN_CLK build: T3_ENGINE_LIB=...)."""
import sys, os, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import chatterbox_vllm2_amd.engine as E
from chatterbox_vllm2_amd.weights import synthetic_tensors, synthetic_cond_emb
from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 560
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = E.T3Engine(n_layers=4, text_vocab=2454, max_model_len=1000, max_seqs=B, kv_bytes=0, enforce_eager=False)
eng.load_tensors(synthetic_tensors(4, 2454, 1234)); eng.finalize()
cond = synthetic_cond_emb(1)
rs = np.random.RandomState(0)
for i in range(B):
    n_text = 81 if i < B // 2 else 106
    ids = assemble_prompt_ids([int(x) for x in rs.randint(3, 600, size=n_text)])
    eng.add_request(i, ids, cond, E.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, max_tokens=900, ignore_eos=True, uid=i))
steps = max(ctx - 128, 10)
eng.run_steps(steps + 2)
buf = (ct.c_uint64 * (4096 * 6))()
rc = eng.lib.t3_debug_attn_clk(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 6).astype(np.float64)[:32 * B] / 100.0
t0 = a[:, 0].min()
def st(v, name): v = np.sort(v); print(f"  {name:38s} min {v[0]:6.2f}  median {v[len(v)//2]:6.2f}  p90 {v[int(len(v)*.9)]:6.2f}  max {v[-1]:6.2f} us")
print(f"attention, {2 * B} rows x 16 heads, context ~{ctx}: first entry -> last exit {a[:,5].max() - t0:.2f} us (rc={rc})")
st(a[:, 0] - t0, "entry after first workgroup")
st(a[:, 1] - a[:, 0], "prologue (rowrec, q, RoPE, KV write)")
st(a[:, 2] - a[:, 1], "wave 0: first chunk (tile wait + math)")
st(a[:, 3] - a[:, 2], "wave 0: remaining chunks")
st(a[:, 4] - a[:, 3], "barrier (slowest wave)")
st(a[:, 5] - a[:, 4], "cross-chunk fold + store")
st(a[:, 5] - a[:, 0], "workgroup lifetime")
st(a[:, 5] - t0, "exit time since kernel start")
# finish time by XCD (workgroup id % 8: the dispatcher deals workgroups round-robin over the 8 XCDs), by head and by row
n = 32 * B
wid = np.arange(n)                      # stamp slot = blockIdx.y * gridDim.x + blockIdx.x  (x = head, y = row)
fin = a[:, 5] - t0
x, y = wid % 16, wid // 16
head = x
for name, key in (("XCD (id % 8)", wid % 8), ("head", head), ("row % 8", y % 8)):
    print(f"  finish time by {name}: " + "  ".join(f"{k}:{np.median(fin[key == k]):5.1f}/{fin[key == k].max():5.1f}" for k in sorted(set(key.tolist()))))
