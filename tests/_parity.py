"""Shared parity helpers of the GPU tests.

North-star tolerance (BASELINE.json): embeddings, losses and gradients within rtol 1e-5 (fp32) of the
reference, with the absolute floor tied to the tensor's scale (1e-6 max|ref|).  The thing compared against is the oracle
run in FLOAT64 on the same fp32 inputs (`to64`): the exact value of the reference's formulas.  The reference's own fp32
CPU arithmetic sits up to 10x this tolerance away from that exact value on long sums (measured: scripts/parity_probe.py),
so comparing two fp32 results would need a looser bound than comparing each with the exact one.
"""
import copy

import numpy as np
import torch

RTOL, ATOL_REL = 1e-5, 1e-6


def _np64(a):
    return a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)


def close(got, ref, rtol=RTOL, atol_rel=ATOL_REL, what=""):
    """|got - ref| <= rtol |ref| + atol_rel max|ref| elementwise; `ref` is the fp64 oracle's value."""
    g, r = _np64(got), _np64(ref)
    np.testing.assert_allclose(g, r, rtol=rtol, atol=atol_rel * max(np.abs(r).max(initial=0.0), 1e-300), err_msg=what)


def to64(module):
    """A float64 copy of an oracle module (same parameter values, exactly)."""
    return copy.deepcopy(module).double()


def grads_close(model, ref64, **kw):
    want = dict(ref64.named_parameters())
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        close(p.grad, want[name].grad, what=name, **kw)


def flipped_rows_are_near_ties(gpu_idx, ref_vq, bound_rel=4e-5):
    """After an UNFORCED fp64 oracle forward: every row whose GPU code differs from the exact argmin must be a near tie -
    its exact distance to the GPU's code exceeds the minimum by less than `bound_rel` x the distance scale of the row
    (|z|^2 + |e|^2: an rtol-1e-5 perturbation of z moves a distance by up to ~4e-5 of that scale, and the fp32 rounding of
    the distance itself is ~D 2^-24 of it).  Returns the number of such rows."""
    d = ref_vq.last_d
    gi = gpu_idx.view(-1).cpu().to(torch.int64)
    best = d.min(1).values
    got = d.gather(1, gi[:, None])[:, 0]
    flipped = d.argmin(1) != gi
    scale = d.abs().max(1).values
    assert bool(((got - best)[flipped] <= bound_rel * scale[flipped]).all()), \
        f"{int(flipped.sum())} rows differ from the exact argmin, worst gap/scale {float(((got - best) / scale).max()):.3e}"
    return int(flipped.sum())
