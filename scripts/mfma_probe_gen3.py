"""Measurement helper: VQ-like operands for scripts/mfma_probe.hip - z ~ N(0,1), codebook ~ N(0,1) or U(+-1/512),
c = a running score of the size the quantiser kernel carries (-|e|^2/2 plus partial dots)."""
import sys
import numpy as np
from mfma_probe_gen import bf16_bits


def main(path, n=120, seed=5):
    rng = np.random.default_rng(seed)
    blobs = []
    for t in range(n):
        A = rng.standard_normal((16, 32))
        if t % 2 == 0:
            B = rng.standard_normal((16, 32)); C = -128.0 + 16.0 * rng.standard_normal((16, 16))
        else:
            B = rng.uniform(-1 / 512, 1 / 512, (16, 32)); C = -1.6e-4 + 0.02 * rng.standard_normal((16, 16))
        blobs.append(bf16_bits(A).tobytes() + bf16_bits(B).tobytes() + C.astype(np.float32).tobytes())
    open(path, "wb").write(b"".join(blobs))
    np.save(path + ".fam.npy", np.arange(n) % 2)
    print(n)


if __name__ == "__main__":
    main(sys.argv[1])
