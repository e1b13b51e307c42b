"""Measurement helper (not product): inputs for scripts/mfma_probe.hip.  Families of tests, 16x16 dot products each:
  0  small integers (exact in any order): checks the operand layout
  1..5  random bf16 operands with exponent spreads 0, 4, 8, 16, 30 (products spread twice that), c of matching size
  6  cancellation probes: b = 1, a = one +2^k and one -2^k among ones
"""
import sys

import numpy as np


def bf16_bits(x):
    """fp32 array -> bf16 bit patterns (round to nearest even)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def main(path, per_family=40, seed=0):
    rng = np.random.default_rng(seed)
    tests = []
    fam = []

    def add(A, B, C, f):
        tests.append(bf16_bits(A).tobytes() + bf16_bits(B).tobytes() + np.asarray(C, dtype=np.float32).tobytes())
        fam.append(f)

    for _ in range(per_family):
        add(rng.integers(-8, 9, (16, 32)), rng.integers(-8, 9, (16, 32)), rng.integers(-100, 100, (16, 16)), 0)
    for f, spread in enumerate((0, 4, 8, 16, 30), start=1):
        for _ in range(per_family):
            ea = rng.integers(-spread, spread + 1, (16, 32))
            eb = rng.integers(-spread, spread + 1, (16, 32))
            A = rng.uniform(1, 2, (16, 32)) * rng.choice([-1, 1], (16, 32)) * 2.0 ** ea
            B = rng.uniform(1, 2, (16, 32)) * rng.choice([-1, 1], (16, 32)) * 2.0 ** eb
            C = rng.uniform(1, 2, (16, 16)) * rng.choice([-1, 1], (16, 16)) * 2.0 ** rng.integers(-2 * spread, 2 * spread + 1, (16, 16))
            add(A, B, C, f)
    for _ in range(per_family):
        A = np.ones((16, 32))
        for r in range(16):
            p, q = rng.choice(32, 2, replace=False)
            k = rng.integers(10, 40)
            A[r, p], A[r, q] = 2.0 ** k, -(2.0 ** k)
        add(A, np.ones((16, 32)), rng.integers(0, 4, (16, 16)), 6)
    with open(path, "wb") as f:
        f.write(b"".join(tests))
    np.save(path + ".fam.npy", np.array(fam))
    print(len(tests))


if __name__ == "__main__":
    main(sys.argv[1])
