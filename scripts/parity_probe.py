"""Measurement helper (not product): error of the GPU path against the fp64-accumulated oracle, in units of the north-star
tolerance  tol = 1e-5 |ref| + 1e-6 max|ref|.  A ratio <= 1 means the tensor passes rtol 1e-5 outright."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_torch  # noqa: E402


def ratio(got, ref):
    g = got.detach().cpu().double()
    r = ref.detach().cpu().double()
    tol = 1e-5 * r.abs() + 1e-6 * r.abs().max().clamp(min=1e-300)
    return float(((g - r).abs() / tol).max())


def graph(N, E, seed):
    rng = np.random.default_rng(seed)
    dst = rng.integers(1, N, 2 * E)
    src = (dst * rng.random(2 * E) ** 2).astype(np.int64)
    key = np.unique(dst * N + src)[:E]
    rng.shuffle(key)
    return np.stack([key % N, key // N]).astype(np.int64)


def full_step(N, E, Fi, D, n_e):
    from causal_eeg_vae_amd import VGAEVQ, EventGraph, sigma_params
    ei_np = graph(N, E, 2)
    torch.manual_seed(0)
    ref = ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = VGAEVQ(Fi, D, n_e, 0.25).cuda()
    model.load_state_dict(ref.state_dict())
    x, eps = torch.randn(N, Fi), torch.randn(N, D)
    sig = sigma_params(N, 0, 5)
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, sg, valid = ref_torch.sigma_negatives_np(rowptr, col, N, sig[0], sig[1])
    neg_ei = torch.from_numpy(np.stack([s, sg]).astype(np.int64))
    ei = torch.from_numpy(ei_np)
    loss, aux = model(x.cuda(), EventGraph(ei.cuda(), N), step=5, eps=eps.cuda(), sigma=sig)
    loss.backward()
    gi = aux["idx"].cpu().view(-1)
    for name, r in (("fp32", ref), ("fp64", ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25).double())):
        if name == "fp64":
            r.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
            xs, es = x.double(), eps.double()
        else:
            xs, es = x, eps
        r.zero_grad()
        l0, a0 = r(xs, ei, es, neg_ei, torch.from_numpy(valid))
        flips = int((a0["idx"].view(-1) != gi).sum())
        d = r.vector_quantization.last_d
        gap = (d.gather(1, gi[:, None]) - d.min(1, keepdim=True).values).view(-1)
        l1, a1 = r(xs, ei, es, neg_ei, torch.from_numpy(valid), force_idx=gi)
        l1.backward()
        print(f"  [{name} oracle] flips {flips}, max gap of flipped rows {float(gap.max()):.3e} (d scale {float(d.abs().max()):.2f})")
        print(f"    loss {ratio(loss, l1):.2f} z {ratio(aux['z'], a1['z']):.2f} kl {ratio(aux['kl'], a1['kl']):.2f} rec {ratio(aux['rec'], a1['rec']):.2f} vq {ratio(aux['vq'], a1['vq']):.2f}")
        rp = dict(r.named_parameters())
        for n_, p in model.named_parameters():
            print(f"    grad {n_:40s} {ratio(p.grad, rp[n_].grad):8.2f}")


if __name__ == "__main__":
    for cfg in ((4000, 64000, 32, 16, 64), (20000, 320000, 64, 64, 512)):
        print("full step", cfg)
        full_step(*cfg)
