"""Fuzz (GPU box): screened fp32 forward vs the full kernel on codebooks built to sit AT the screening bound - clusters of codes
whose spread ranges from far above to far below Delta_row, rows near the cluster centres, random shapes and scales.  Codes and z_q
must be identical in every trial.   python scripts/vq_screen_fuzz.py [trials]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vq_screen_sweep import dev, fwd  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device=dev).manual_seed(2024)
cpu = torch.Generator().manual_seed(7)
bad = screened = 0
for t in range(trials):
    D = [32, 64, 128, 256][int(torch.randint(0, 4, (1,), generator=cpu))]
    n_max = min(65536 // D, 1024)
    n_e = int(torch.randint(2, n_max + 1, (1,), generator=cpu))
    B = int(torch.randint(1, 30000, (1,), generator=cpu))
    scale = 10.0 ** float(torch.empty(1).uniform_(-4, 4, generator=cpu))
    spread = 10.0 ** float(torch.empty(1).uniform_(-7, 0, generator=cpu))       # relative spread of the codes inside a cluster
    n_cl = int(torch.randint(1, max(2, n_e // 2), (1,), generator=cpu))
    centres = torch.randn(n_cl, D, device=dev, generator=g)
    member = torch.randint(0, n_cl, (n_e,), device=dev, generator=g)
    W = (centres[member] * (1 + spread * torch.randn(n_e, D, device=dev, generator=g))) * scale
    zsrc = torch.randint(0, n_cl, (B,), device=dev, generator=g)
    znoise = 10.0 ** float(torch.empty(1).uniform_(-6, 0.5, generator=cpu))
    z = (centres[zsrc] + znoise * torch.randn(B, D, device=dev, generator=g)) * scale * (10.0 ** float(torch.empty(1).uniform_(-1, 1, generator=cpu)))
    os.environ["CEEG_VQ_SCREEN"] = "1"
    i1, q1, how = fwd(z.contiguous(), W.contiguous())
    os.environ["CEEG_VQ_SCREEN"] = "0"
    i0, q0, _ = fwd(z.contiguous(), W.contiguous())
    same = bool(torch.equal(i1, i0)) and bool(torch.equal(q1, q0))
    screened += how == 1
    if not same:
        bad += 1
        print(f"MISMATCH trial {t}: B={B} n_e={n_e} D={D} scale={scale:.3g} spread={spread:.3g} znoise={znoise:.3g} how={how} "
              f"rows differing {int((i1 != i0).sum())}")
print(f"{trials} trials: {bad} mismatches; {screened} decided by the screen, {trials - screened} handed over")
sys.exit(1 if bad else 0)
