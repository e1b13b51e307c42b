#!/usr/bin/env python3
"""Generate golden vectors for the VectorQuantizer from the LIVE reference.

Run ONLY in the build container (the reference never travels to the GPU box):

    python tests/golden/make_vq_golden.py

It imports /root/reference/scripts/quantizer.py (VectorQuantizer, quantizer.py:6-67),
runs forward + backward of  loss + (z_q * r).sum()  on seeded inputs and writes one
.npz per case into tests/golden/.  The .npz files hold DATA only (inputs, outputs,
gradients, and an fp64 top-2 distance gap used by the tests to explain rounding-
sensitive rows); no reference source is copied.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/scripts"
OUT = os.path.dirname(os.path.abspath(__file__))


def run_case(name, B, n_e, D, beta, seed, codebook, ties=None):
    sys.path.insert(0, REF)
    import quantizer  # the live reference module

    torch.manual_seed(seed)
    vq = quantizer.VectorQuantizer(n_e, D, beta)
    if codebook == "normal":
        with torch.no_grad():
            vq.embedding.weight.normal_()
    elif codebook == "ref_init":
        pass  # U(-1/n_e, 1/n_e) from the reference constructor (quantizer.py:23)
    else:
        raise ValueError(codebook)
    z = torch.randn(B, D)
    if ties == "dup_rows":
        # duplicated codebook rows: exact ties, the LOWER index must win
        with torch.no_grad():
            vq.embedding.weight[n_e // 2:] = vq.embedding.weight[: n_e // 2]
    elif ties == "all_equal":
        # every code identical: all distances equal, index 0 must win
        with torch.no_grad():
            vq.embedding.weight[:] = vq.embedding.weight[0:1]
    elif ties == "exact_hit":
        # latents exactly equal to codebook rows (distance 0 up to rounding)
        with torch.no_grad():
            sel = torch.arange(B) % n_e
            z = vq.embedding.weight[sel].clone()
    z.requires_grad_(True)
    r = torch.randn(B, D)
    loss, z_q, ppl, onehot, idx = vq(z)
    (loss + (z_q * r).sum()).backward()

    W = vq.embedding.weight.detach()
    # fp64 distances -> top-2 gap per row (rounding-sensitivity witness)
    zd, Wd = z.detach().double(), W.double()
    d64 = (zd ** 2).sum(1, keepdim=True) + (Wd ** 2).sum(1) - 2 * zd @ Wd.t()
    top2 = torch.topk(d64, 2, dim=1, largest=False).values
    gap = (top2[:, 1] - top2[:, 0]).numpy()
    idx64 = torch.argmin(d64, dim=1).numpy()

    assert onehot.shape == (B, n_e) and onehot.dtype == torch.float32
    assert idx.shape == (B, 1) and idx.dtype == torch.int64
    assert torch.equal(onehot.argmax(1, keepdim=True), idx)
    np.savez_compressed(
        os.path.join(OUT, f"vq_{name}.npz"),
        z=z.detach().numpy(), W=W.numpy(), r=r.numpy(), beta=np.float32(beta),
        loss=loss.detach().numpy(), z_q=z_q.detach().numpy(),
        perplexity=ppl.detach().numpy(), idx=idx.numpy(),
        dz=z.grad.numpy(), dW=vq.embedding.weight.grad.numpy(),
        gap64=gap, idx64=idx64,
    )
    print(f"{name}: B={B} n_e={n_e} D={D} loss={float(loss):.6g} ppl={float(ppl):.6g} "
          f"min_gap64={gap.min():.3g} ref_vs_fp64_mismatch={(idx64 != idx.numpy()[:, 0]).sum()}")


def main():
    # the reference's own instantiation: (512, 8, 0.25)  (run.py:296-298)
    run_case("ref8_refinit_s0", 64, 512, 8, 0.25, 0, "ref_init")
    run_case("ref8_normal_s1", 64, 512, 8, 0.25, 1, "normal")
    run_case("d128_normal_s0", 257, 512, 128, 0.25, 0, "normal")
    run_case("d128_refinit_s2", 257, 512, 128, 0.25, 2, "ref_init")
    run_case("d256_normal_s0", 160, 512, 256, 0.25, 0, "normal")
    run_case("d256_refinit_s1", 160, 512, 256, 0.25, 1, "ref_init")
    run_case("ne96_d40_normal_s3", 77, 96, 40, 0.5, 3, "normal")  # ragged n_e / D
    run_case("tie_dup_rows", 96, 64, 32, 0.25, 4, "normal", ties="dup_rows")
    run_case("tie_all_equal", 33, 64, 16, 0.25, 5, "normal", ties="all_equal")
    run_case("tie_exact_hit", 80, 64, 32, 0.25, 6, "normal", ties="exact_hit")
    run_case("single_row", 1, 512, 8, 0.25, 7, "normal")


if __name__ == "__main__":
    main()
