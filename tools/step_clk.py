"""Diagnostic: where the boundaries of a decode layer sit on ONE clock (needs a build with -DT3_ATTN_CLK -DT3_GEMM_CLK: T3_ENGINE_LIB=...).
Every workgroup of the fused attention and of the gemm2 forms stamps s_memrealtime (100 MHz, the same counter everywhere) at its phases;
the arrays keep the LAST launch of each class, i.e. the last layer's attention, o_proj, gate/up, down_proj (class 0 holds the speech head,
which overwrites the last qkv).  Printed: per kernel first entry / last exit on a common axis, and the gaps between them.
usage: python tools/step_clk.py [context] [utterances]"""
import sys, os, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import chatterbox_vllm2_amd.engine as E
from chatterbox_vllm2_amd.weights import synthetic_tensors, synthetic_cond_emb
from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 560
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NL = 4
eng = E.T3Engine(n_layers=NL, text_vocab=2454, max_model_len=1000, max_seqs=B, kv_bytes=0, enforce_eager=False)
eng.load_tensors(synthetic_tensors(NL, 2454, 1234)); eng.finalize()
cond = synthetic_cond_emb(1)
rs = np.random.RandomState(0)
for i in range(B):
    n_text = 81 if i < B // 2 else 106
    ids = assemble_prompt_ids([int(x) for x in rs.randint(3, 600, size=n_text)])
    eng.add_request(i, ids, cond, E.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, max_tokens=900, ignore_eos=True, uid=i))
eng.run_steps(max(ctx - 128, 10) + 2)
ab = (ct.c_uint64 * (4096 * 6))(); eng.lib.t3_debug_attn_clk(ab)
gb = (ct.c_uint64 * (4 * 2048 * 6))(); eng.lib.t3_debug_gemm2_clk(gb)
att = np.frombuffer(ab, dtype=np.uint64).reshape(4096, 6).astype(np.float64)[:32 * B] / 100.0
g = np.frombuffer(gb, dtype=np.uint64).reshape(4, 2048, 6).astype(np.float64) / 100.0
rows = []
rows.append(("attention (last layer)", att[:, 0].min(), att[:, 5].max(), att[:, 0].max(), None))
for k, name in ((2, "o_proj"), (1, "gate/up"), (3, "down_proj")):
    v = g[k][g[k][:, 0] > 0]
    rows.append((name, v[:, 0].min(), v[:, 5].max(), v[:, 0].max(), np.median(v[:, 1] - v[:, 0])))
t0 = rows[0][1]
print(f"{2 * B} rows, context ~{ctx}, {NL} layers; us on one clock, 0 = first attention workgroup of the last layer")
prev_end = None
for name, a0, a1, alast, aland in rows:
    gap = "" if prev_end is None else f"   gap to the previous kernel's last exit {a0 - prev_end:6.2f}"
    extra = "" if aland is None else f"   entry -> operand rows in LDS (median) {aland:5.2f}"
    print(f"  {name:24s} first entry {a0 - t0:8.2f}  last entry {alast - t0:8.2f}  last exit {a1 - t0:8.2f}  (in-kernel {a1 - a0:6.2f}){gap}{extra}")
    prev_end = a1
