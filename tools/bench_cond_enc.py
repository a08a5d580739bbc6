"""f3 measurement: latency of one conditioning-encoder call (once per voice) on the device vs the oracle on the host cores.
Prints one JSON line.  `rocprofv3 --kernel-trace --stats -- python3 tools/bench_cond_enc.py` gives the per-kernel split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_vllm2_amd.cond_enc import T3CondEnc
from chatterbox_vllm2_amd.weights import synthetic_cond_enc_tensors, synthetic_cond_inputs

params = dict(synthetic_cond_enc_tensors(4321))
spk, prompt, emo = synthetic_cond_inputs(7, 150)
enc = T3CondEnc(); enc.load_state_dict(params)
for _ in range(5):
    out = enc(spk, prompt, emo)
n = 200
t0 = time.perf_counter()
for _ in range(n):
    out = enc(spk, prompt, emo)
gpu_ms = (time.perf_counter() - t0) / n * 1e3
res = {"what": "T3CondEnc.forward, 150 prompt rows, fp32 (host buffers in, host buffers out, synchronous)", "calls": n,
       "device_ms_per_call": round(gpu_ms, 4), "weight_bytes_read_per_call": 2 * 4 * 1024 * 1024 * 4 + 256 * 1024 * 4,
       "flop_per_call": 2 * 1024 * 1024 * (32 + 150 + 150 + 32 + 32 * 3 + 32) + 4 * 2 * 256 * (32 * 150 + 32 * 32) * 2}
if "--no-cpu" not in sys.argv:
    from oracle import oracle as O
    O.set_threads(min(len(os.sched_getaffinity(0)), 16))
    O.cond_enc(params, spk, prompt, emo)
    t0 = time.perf_counter()
    for _ in range(3):
        ref = O.cond_enc(params, spk, prompt, emo)
    res["cpu_oracle_ms_per_call"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    res["cpu_threads"] = min(len(os.sched_getaffinity(0)), 16)
    res["bit_identical_to_oracle"] = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
print(json.dumps(res))
