#!/usr/bin/env python3
"""Feature-slab experiment: one SpMM pass of width F vs F/S passes over column slabs of width S (views, same ld)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_eeg_vae_amd import EventGraph
from causal_eeg_vae_amd.models import _spmm
from causal_eeg_vae_amd.synth import synth_event_graph

N, E = 1_000_000, 16_000_000
dev = torch.device("cuda")
ei = synth_event_graph(N, E, seed=1000, device=dev)
g = EventGraph(ei, N)
def run(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for F in (128, 256):
    x = torch.randn(N, F, device=dev)
    out = torch.empty(N, F, device=dev)
    for S in (F, 128, 64, 32, 16):
        if S > F: continue
        def fwd():
            for c in range(0, F, S):
                _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x[:, c:c + S], N, out=out[:, c:c + S])
        def bwd():
            for c in range(0, F, S):
                _spmm(g.gcn.t_rowptr, g.gcn.t_col, g.gcn_val, g.gcn.t_perm, x[:, c:c + S], N, out=out[:, c:c + S])
        print(f"F={F} slab={S}: fwd {run(fwd):.3f} ms  transposed {run(bwd):.3f} ms", flush=True)
