"""Random staged inputs for mfma_probe2 bf16_16x16x32: tiles 0-15 only block 0 non-zero & C=0; 16-31 block 0 + random C;
32-47 blocks 0,1 & C=0; 48-63 all blocks, C=0; 64-79 all blocks + C.  Narrow (|exp| <= 2) and wide (<= 8) exponent spreads alternate."""
import numpy as np
rng = np.random.default_rng(7)
T, M, N, K = 80, 16, 16, 32
def bf16(x):
    u = np.float32(x).view(np.uint32); u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)
A = np.zeros((T, M, K), np.float32); B = np.zeros((T, K, N), np.float32); C = np.zeros((T, M, N), np.float32)
for t in range(T):
    spread = 2 if t % 2 == 0 else 8
    a = rng.standard_normal((M, K)) * np.exp2(rng.integers(-spread, spread + 1, (M, K)))
    b = rng.standard_normal((K, N)) * np.exp2(rng.integers(-spread, spread + 1, (K, N)))
    stage = t // 16
    nblk = {0: 1, 1: 1, 2: 2, 3: 4, 4: 4}[stage]
    a[:, 8 * nblk:] = 0; b[8 * nblk:, :] = 0
    A[t], B[t] = a, b
    if stage in (1, 4):
        C[t] = rng.standard_normal((M, N)) * np.exp2(rng.integers(-spread, spread + 3, (M, N)))
import os
os.makedirs("tools/probe_in", exist_ok=True)
bf16(A).tofile("tools/probe_in/A3.bin"); bf16(B).tofile("tools/probe_in/B3.bin"); C.tofile("tools/probe_in/C3.bin")
print("ok")
