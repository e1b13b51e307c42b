"""Measurement helper (not product): which summation model reproduces v_mfma_f32_16x16x32_bf16 bit for bit?
Reads scripts/mfma_probe_in.bin (+ .fam.npy) and the hardware's gpurun_out/mfma/out.bin; evaluates candidate models in
exact integer arithmetic and prints, per test family, the share of the 256-dot outputs each model reproduces exactly."""
import sys
from fractions import Fraction

import numpy as np


def bf16_to_frac(bits):
    f = np.array([int(bits) << 16], dtype=np.uint32).view(np.float32)[0]
    return Fraction(float(f))


def rn32(x, mode="rne"):
    """Fraction -> nearest float32 value (as Fraction); ties to even; subnormals kept; no overflow handling needed here."""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    a = -x if x < 0 else x
    # exponent e with 2^e <= a < 2^(e+1)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    q = max(e - 23, -149)
    scaled = a / (Fraction(2) ** q)
    fl = scaled.numerator // scaled.denominator
    rem = scaled - fl
    if mode == "rne":
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
    elif mode == "trunc":
        pass
    return s * fl * (Fraction(2) ** q)


def seq(c, p, order=None, mode="rne"):
    acc = c
    for k in (order or range(len(p))):
        acc = rn32(acc + p[k], mode)
    return acc


def exact_once(c, p, mode="rne"):
    return rn32(c + sum(p), mode)


def grouped(c, p, g, c_first=True, mode="rne", strided=False):
    n = len(p)
    if strided:
        blocks = [[p[k] for k in range(i, n, n // g)] for i in range(n // g)]
    else:
        blocks = [p[i:i + g] for i in range(0, n, g)]
    acc = c if c_first else Fraction(0)
    for b in blocks:
        acc = rn32(acc + sum(b), mode)
    if not c_first:
        acc = rn32(acc + c, mode)
    return acc


def aligned(c, p, keep_bits, mode="trunc", final="rne"):
    """All addends aligned to the largest exponent among (c, products); each is truncated to `keep_bits` fractional bits
    below that exponent's leading bit before the exact sum; one final rounding."""
    terms = [c] + list(p)
    nz = [t for t in terms if t != 0]
    if not nz:
        return Fraction(0)
    emax = max((abs(t).numerator.bit_length() - abs(t).denominator.bit_length()) for t in nz) + 1
    ulp = Fraction(2) ** (emax - keep_bits)
    tot = Fraction(0)
    for t in terms:
        qv = t / ulp
        fl = qv.numerator // qv.denominator  # floor
        if mode == "trunc" and qv < 0 and fl != qv:
            fl += 1  # toward zero
        tot += fl * ulp
    return rn32(tot, final)


MODELS = {
    "seq k-ascending rne": lambda c, p: seq(c, p),
    "exact, one rne": lambda c, p: exact_once(c, p),
    "exact, one trunc": lambda c, p: exact_once(c, p, "trunc"),
    "blocks of 2 exact, rne chain": lambda c, p: grouped(c, p, 2),
    "blocks of 4 exact, rne chain": lambda c, p: grouped(c, p, 4),
    "blocks of 8 exact, rne chain": lambda c, p: grouped(c, p, 8),
    "blocks of 16 exact, rne chain": lambda c, p: grouped(c, p, 16),
    "blocks of 8 exact, c last": lambda c, p: grouped(c, p, 8, c_first=False),
    "blocks of 16 exact, c last": lambda c, p: grouped(c, p, 16, c_first=False),
    "strided 8 (k%4) exact, rne chain": lambda c, p: grouped(c, p, 8, strided=True),
    "aligned 24 trunc": lambda c, p: aligned(c, p, 24),
    "aligned 26 trunc": lambda c, p: aligned(c, p, 26),
    "aligned 27 trunc": lambda c, p: aligned(c, p, 27),
    "aligned 28 trunc": lambda c, p: aligned(c, p, 28),
    "aligned 32 trunc": lambda c, p: aligned(c, p, 32),
}


def main(inp="scripts/mfma_probe_in.bin", outp="gpurun_out/mfma/out.bin", dots_per_family=160):
    fam = np.load(inp + ".fam.npy")
    T = len(fam)
    raw = np.fromfile(inp, dtype=np.uint8).reshape(T, -1)
    A = raw[:, :1024].copy().view(np.uint16).reshape(T, 16, 32)
    B = raw[:, 1024:2048].copy().view(np.uint16).reshape(T, 16, 32)
    C = raw[:, 2048:].copy().view(np.float32).reshape(T, 16, 16)
    D = np.fromfile(outp, dtype=np.float32).reshape(T, 16, 16)
    rng = np.random.default_rng(1)
    for f in sorted(set(fam.tolist())):
        ts = np.nonzero(fam == f)[0]
        hits = {m: 0 for m in MODELS}
        n = 0
        for _ in range(dots_per_family):
            t = int(rng.choice(ts)); r = int(rng.integers(16)); cidx = int(rng.integers(16))
            p = [bf16_to_frac(A[t, r, k]) * bf16_to_frac(B[t, cidx, k]) for k in range(32)]
            c = Fraction(float(C[t, r, cidx]))
            d = Fraction(float(D[t, r, cidx]))
            n += 1
            for m, fn in MODELS.items():
                hits[m] += int(fn(c, p) == d)
        print(f"family {f}: " + "; ".join(f"{m}: {hits[m] / n:.2f}" for m in MODELS))


if __name__ == "__main__":
    main(*sys.argv[1:])


def blk_aligned(c, p, g, keep_bits, mode="trunc", final="rne"):
    acc = c
    for i in range(0, len(p), g):
        acc = aligned(acc, p[i:i + g], keep_bits, mode, final)
    return acc


def blk_aligned_floor(c, p, g, keep_bits, final="rne"):
    acc = c
    for i in range(0, len(p), g):
        acc = aligned(acc, p[i:i + g], keep_bits, "floor", final)
    return acc
