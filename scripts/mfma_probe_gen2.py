"""Measurement helper: targeted cancellation probes for v_mfma_f32_16x16x32_bf16 (analysed by hand / mfma_probe_fit.py).
Each dot product: +big at k0, -big at k2, small = (+-x) * (y 2^-s) at k1, everything else 0, C = c0.
One position pattern per test; rows carry (sign, mantissa x, big exponent), columns the shift s."""
import sys
import numpy as np
from mfma_probe_gen import bf16_bits

PATTERNS = [(0, 1, 2), (0, 1, 7), (0, 7, 3), (0, 1, 8), (0, 8, 1), (0, 8, 16), (0, 1, 31), (0, 31, 1),
            (8, 9, 0), (8, 9, 10), (24, 25, 0), (24, 31, 26), (3, 4, 5), (4, 3, 11), (16, 0, 8), (1, 0, 2)]
ROWS = [(sg, x, E) for E in (0, 5) for x in (1.9921875, 1.0, 1.5, 1.0078125) for sg in (1.0, -1.0)]  # 16 rows
Y = 1.9921875


def main(path):
    blobs, meta = [], []
    for pi, (k0, k2, k1) in enumerate(PATTERNS):
        for cmode in (0, 1, 2):   # 0: C = 0; 1: C = +big replaces the k0 product; 2: C = +big, both products cancel (k0,k2) too
            for s0 in (0, 16, 32, 48):
                A = np.zeros((16, 32)); B = np.zeros((16, 32)); C = np.zeros((16, 16))
                for r, (sg, x, E) in enumerate(ROWS):
                    A[r, k0] = 0.0 if cmode == 1 else 2.0 ** E
                    A[r, k2] = -(2.0 ** E)
                    A[r, k1] = sg * x
                    if cmode == 1:
                        C[r, :] = 2.0 ** E
                    if cmode == 2:
                        C[r, :] = 2.0 ** (E + 3)
                for c in range(16):
                    B[c, k0] = B[c, k2] = 1.0
                    B[c, k1] = Y * 2.0 ** (-(s0 + c))
                blobs.append(bf16_bits(A).tobytes() + bf16_bits(B).tobytes() + C.astype(np.float32).tobytes())
                meta.append((pi, cmode, s0))
    with open(path, "wb") as f:
        f.write(b"".join(blobs))
    np.save(path + ".meta.npy", np.array(meta))
    np.save(path + ".fam.npy", np.zeros(len(blobs), dtype=np.int64))
    print(len(blobs))


if __name__ == "__main__":
    main(sys.argv[1])
