"""Degree statistics of the bench graph (attention CSR): longest rows of the CSR and of its transpose."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from causal_eeg_vae_amd import GATConv  # noqa: E402
from causal_eeg_vae_amd.synth import synth_event_graph  # noqa: E402

ei = synth_event_graph(1_000_000, 16_000_000, seed=1000, device="cuda")
pair = GATConv.attention_graph(ei, 1_000_000)
for name, rp in (("rows (by destination)", pair.rowptr), ("transposed rows (by source)", pair.t_rowptr)):
    d = (rp[1:] - rp[:-1]).to(torch.int64)
    top = torch.sort(d, descending=True).values[:8].tolist()
    print(name, "max", int(d.max()), "top8", top, "entries in rows > 256:", int(d[d > 256].sum()), "rows > 256:", int((d > 256).sum()),
          "entries in rows > 4096:", int(d[d > 4096].sum()))
