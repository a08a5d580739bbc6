"""Diagnostic: phase timestamps of the sampler kernel (needs a -DT3_SAMPLER_CLK build, path in argv[1])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import chatterbox_vllm2_amd.engine as E
E.LIB_PATH = os.path.abspath(sys.argv[1])
g = torch.Generator().manual_seed(1)
names = ["start", "loads issued", "phase A done", "max + weights", "wmax + W / minp / topk", "Tm", "radix select", "ties", "draw scan", "philox", "id written"]   # stamps of the thread that writes the id
for scale in (2.0, 6.0):
    lg = (torch.randn(2, 8208, generator=g) * scale).to(torch.bfloat16)
    counts = torch.zeros(8194, dtype=torch.int32); counts[torch.randint(0, 8194, (300,), generator=g)] = 1
    counts = counts.to(torch.uint16)
    sp = E.T3Sampling(0.8, 0.8, 0.0, 2.0, 0, 0, 0, 1000, 1, 6562, 7, 3, 0, 0)
    for rep in range(3):
        tok, d = E.k_sample(lg, counts, sp, 0.5, 5 + rep)
    t = d[:11].tolist()
    print("logit scale", scale, "token", tok)
    for i in range(1, 11):
        print(f"  {names[i]:16s} +{(t[i] - t[i-1]) / 100.0:7.2f} us   (t={t[i] / 100.0:.2f})")
