"""Validation sweep (GPU box): screened fp32 forward vs the full kernel (CEEG_VQ_SCREEN=0) on large random batches - codes and z_q
must be identical bit for bit whatever the scales; prints one line per configuration with which form decided."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")


def fwd(z, W):
    B, D = z.shape
    n_e = W.shape[0]
    zq = torch.empty_like(z)
    idx = torch.empty(B, dtype=torch.int64, device=dev)
    sc = torch.empty(2, device=dev)
    ws = _lib.workspace(L.ceeg_vq_forward_ws_bytes(B, n_e, D, _lib.CEEG_F32), dev)
    _lib.check(L.ceeg_vq_forward(z.data_ptr(), W.data_ptr(), B, n_e, D, _lib.CEEG_F32, 0.25, zq.data_ptr(), idx.data_ptr(), sc[0:].data_ptr(),
                                 sc[1:].data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream()), "vq")
    how = ctypes.c_int(-7)
    _lib.check(L.ceeg_vq_forward_screened(ws.data_ptr(), ws.numel(), B, n_e, D, _lib.CEEG_F32, ctypes.addressof(how), _lib.stream()), "how")
    return idx, zq, how.value


def main():
    bad = 0
    for seed, (B, n_e, D, zs, ws_, kind) in enumerate([
            (1_000_000, 512, 128, 1.0, 0.1, "normal"), (1_000_000, 512, 128, 1.0, 1.0, "normal"), (1_000_000, 512, 128, 3.0, 0.02, "normal"),
            (1_000_000, 512, 128, 1.0, 1.0, "uniform_ref"), (1_000_000, 512, 128, 1.0, 1.0, "trained_like"), (2_000_000, 256, 64, 1.0, 0.3, "normal"),
            (500_000, 256, 256, 1.0, 0.1, "normal"), (1_000_000, 100, 32, 0.5, 0.5, "normal"), (1_000_000, 512, 128, 1e-3, 1e-3, "normal"),
            (1_000_000, 512, 128, 30.0, 5.0, "normal")]):
        g = torch.Generator(device=dev).manual_seed(100 + seed)
        z = torch.randn(B, D, device=dev, generator=g) * zs
        if kind == "uniform_ref":
            W = (torch.rand(n_e, D, device=dev, generator=g) * 2 - 1) / n_e
        elif kind == "trained_like":      # codes are cluster centres of z: rows sit near their codes
            W = z[torch.randint(0, B, (n_e,), device=dev, generator=g)] + 0.05 * torch.randn(n_e, D, device=dev, generator=g)
        else:
            W = torch.randn(n_e, D, device=dev, generator=g) * ws_
        os.environ["CEEG_VQ_SCREEN"] = "1"
        i1, q1, how = fwd(z, W)
        os.environ["CEEG_VQ_SCREEN"] = "0"
        i0, q0, _ = fwd(z, W)
        same = bool(torch.equal(i1, i0)) and bool(torch.equal(q1, q0))
        bad += not same
        print(f"B={B} n_e={n_e} D={D} z*{zs} W*{ws_} {kind}: decided by {'screen' if how == 1 else 'full kernel'}; identical {same}; "
              f"codes used {int(torch.unique(i1).numel())}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
