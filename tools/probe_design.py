"""Crafted inputs for mfma_probe2 (bf16 16x16x32) + analysis of the returned D.
   python tools/probe_design.py make   -> tools/probe_in/{A,B,C}.bin
   python tools/probe_design.py read   -> interprets gpurun_out/probe2_D.bin
Every row of every tile is one experiment: D[row][col] = C[row][col] + sum_k A[row][k] * B[k][col], B = all ones.
"""
import sys
from fractions import Fraction

import numpy as np

K, M, N = 32, 16, 16


def f2bf(x):
    u = np.float32(x).view(np.uint32)
    assert (u & 0xFFFF) == 0, f"{x} is not bf16-representable"
    return np.uint16(u >> 16)


def experiments():
    """yield (label, {k: value}, c) ; values must be bf16-representable"""
    small = lambda s: float(2.0 ** -s * (1 + 2.0 ** -7))
    for (k1, k2) in [(0, 1), (0, 7), (0, 8), (0, 16), (0, 24), (0, 31), (1, 0), (8, 0), (31, 0), (3, 12)]:
        for s in range(0, 40):
            yield (f"two k1={k1} k2={k2} s={s}", {k1: 1.0, k2: small(s)}, 0.0)
    for s in range(0, 40):
        yield (f"Cbig s={s}", {0: small(s)}, 1.0)
        yield (f"Csmall s={s}", {0: 1.0}, small(s))
    for n in range(1, 32):
        for e in (24, 25, 26, 23):
            yield (f"many n={n} e={e}", {0: 1.0, **{k: 2.0 ** -e for k in range(1, n + 1)}}, 0.0)
    for n in range(1, 32):
        yield (f"manyC n={n} e=24", {k: 2.0 ** -24 for k in range(0, n)}, 1.0)
    for s in range(1, 40):
        yield (f"neg s={s}", {0: 1.0, 1: -(2.0 ** -s)}, 0.0)
        yield (f"neg8 s={s}", {0: 1.0, 8: -(2.0 ** -s)}, 0.0)
        yield (f"negC s={s}", {0: -(2.0 ** -s)}, 1.0)
    # three terms: big, -big, small (cancellation): exact = small
    for s in range(0, 40):
        yield (f"cancel s={s}", {0: 1.0, 1: -1.0, 2: small(s)}, 0.0)
        yield (f"cancel8 s={s}", {0: 1.0, 8: -1.0, 16: small(s)}, 0.0)
        yield (f"cancelC s={s}", {0: -1.0, 2: small(s)}, 1.0)


def make():
    import os
    ex = list(experiments())
    T = (len(ex) + M - 1) // M
    A = np.zeros((T, M, K), np.uint16); C = np.zeros((T, M, N), np.float32)
    B = np.full((T, K, N), f2bf(1.0), np.uint16)
    for i, (_, a, c) in enumerate(ex):
        t, r = divmod(i, M)
        for k, v in a.items():
            A[t, r, k] = f2bf(v)
        C[t, r, :] = c
    os.makedirs("tools/probe_in", exist_ok=True)
    A.tofile("tools/probe_in/A.bin"); B.tofile("tools/probe_in/B.bin"); C.tofile("tools/probe_in/C.bin")
    print(len(ex), "experiments,", T, "tiles")


def read(path="gpurun_out/probe2_D.bin"):
    ex = list(experiments())
    D = np.fromfile(path, np.float32).reshape(-1, M, N)
    last = None
    for i, (label, a, c) in enumerate(ex):
        t, r = divmod(i, M)
        got = D[t, r]
        assert (got == got[0]).all(), (label, got)
        exact = Fraction(c) + sum(Fraction(v) for v in a.values())
        rne = np.float32(float(exact))      # float(Fraction) is correctly rounded to f64; f64->f32 double rounding is harmless for these inputs
        fam = label.split(" s=")[0].split(" n=")[0]
        if fam != last:
            print(f"--- {fam}"); last = fam
        flag = "" if got[0] == rne else "   <-- differs from RNE(exact)"
        print(f"{label:34s} got {float(got[0])!r:24} = 1{(Fraction(float(got[0])) - 1) * 2**23!s:>12}/2^23   exact-RNE {float(rne)!r}{flag}")


if __name__ == "__main__":
    make() if sys.argv[1] == "make" else read(*sys.argv[2:])
