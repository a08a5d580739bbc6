"""Offline analysis of tools/mfma_probe dumps: which summation order / rounding do the MFMA instructions
use?  (Numerics research for the bf16-MFMA path; not part of the product.)"""
import sys
from fractions import Fraction

import numpy as np

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/probe"


def bf(a):
    return (a.astype(np.uint32) << 16).view(np.float32)


def rne32(fr: Fraction) -> np.float32:
    """exact Fraction -> float32, round to nearest even"""
    if fr == 0:
        return np.float32(0.0)
    s = -1 if fr < 0 else 1
    fr = abs(fr)
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    if Fraction(2) ** e > fr:
        e -= 1
    e = max(e, -126)
    q = fr / Fraction(2) ** (e - 23)
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(s * float(n) * 2.0 ** (e - 23))


def rz32(fr: Fraction) -> np.float32:
    if fr == 0:
        return np.float32(0.0)
    s = -1 if fr < 0 else 1
    fr = abs(fr)
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    if Fraction(2) ** e > fr:
        e -= 1
    e = max(e, -126)
    q = fr / Fraction(2) ** (e - 23)
    n = q.numerator // q.denominator
    return np.float32(s * float(n) * 2.0 ** (e - 23))


def report(name, got, cand):
    for k, v in cand.items():
        eq = (v.view(np.uint32) == got.view(np.uint32)).mean()
        print(f"  {name:16s} {k:40s} match {eq*100:7.3f}%")


# ---------------- f32 16x16x4
A = np.fromfile(f"{d}/f32_16x16x4_A.bin", np.float32).reshape(-1, 16, 4)
B = np.fromfile(f"{d}/f32_16x16x4_B.bin", np.float32).reshape(-1, 4, 16)
C = np.fromfile(f"{d}/f32_16x16x4_C.bin", np.float32).reshape(-1, 16, 16)
D = np.fromfile(f"{d}/f32_16x16x4_D.bin", np.float32).reshape(-1, 16, 16)
print("f32 16x16x4 (random fp32 inputs, products NOT exact):")


def fma32(a, b, c):   # exact fma via float64 (a*b exact in f64; one more rounding to f64 on the add: use Fraction-free check)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


for order in ([0, 1, 2, 3], [3, 2, 1, 0]):
    acc = C.copy()
    for k in order:
        acc = fma32(A[:, :, k][:, :, None], B[:, k, :][:, None, :], acc)
    report("f32", D, {f"fma chain order {order} (f64-emulated fma)": acc})

# ---------------- bf16
for name, (m, n, k) in {"bf16_16x16x32": (16, 16, 32), "bf16_32x32x16": (32, 32, 16)}.items():
    A = bf(np.fromfile(f"{d}/{name}_A.bin", np.uint16)).reshape(-1, m, k)
    B = bf(np.fromfile(f"{d}/{name}_B.bin", np.uint16)).reshape(-1, k, n)
    C = np.fromfile(f"{d}/{name}_C.bin", np.float32).reshape(-1, m, n)
    D = np.fromfile(f"{d}/{name}_D.bin", np.float32).reshape(-1, m, n)
    T = min(A.shape[0], 24)
    A, B, C, D = A[:T], B[:T], C[:T], D[:T]
    print(f"{name}: {T} tiles")
    P = A[:, :, :, None].astype(np.float64) * B[:, None, :, :].astype(np.float64)     # [T, m, k, n] exact products
    P = np.transpose(P, (0, 1, 3, 2))                                                 # [T, m, n, k]
    cand = {}
    # sequential fp32 adds, ascending k
    acc = C.copy()
    for kk in range(k):
        acc = (acc.astype(np.float64) + P[..., kk]).astype(np.float32)
    cand["sequential k asc, RNE each add"] = acc
    # exact sum, single rounding (RNE / RZ)
    flatP = P.reshape(-1, k); flatC = C.reshape(-1)
    for blk in (k, k // 2, 8, 4, 2):
        for rnd_name, rnd in (("RNE", rne32), ("RZ", rz32)):
            out = np.empty(flatC.shape, np.float32)
            for i in range(flatC.size):
                acc = Fraction(float(flatC[i]))
                for b0 in range(0, k, blk):
                    s = acc
                    for kk in range(b0, b0 + blk):
                        s += Fraction(float(flatP[i, kk]))
                    acc = Fraction(float(rnd(s)))
                out[i] = np.float32(float(acc))
            cand[f"blocks of {blk}: exact(C+block) then {rnd_name}"] = out.reshape(C.shape)
    report(name, D, cand)
