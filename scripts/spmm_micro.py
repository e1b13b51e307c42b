#!/usr/bin/env python3
"""Micro-benchmark of ceeg_spmm_csr on the bench graph (forward CSR and transposed pass), HIP-event timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_eeg_vae_amd import EventGraph
from causal_eeg_vae_amd.models import _spmm
from causal_eeg_vae_amd.synth import synth_event_graph

N, E = 1_000_000, 16_000_000
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
ei = synth_event_graph(N, E, seed=1000, device=dev)
g = EventGraph(ei, N)
x = torch.randn(N, F, device=dev)
def run(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
nnz = g.gcn.nnz
bytes_alg = nnz * 8 + (N + 1) * 4 + 2 * N * F * 4
fwd = run(lambda: _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x, N))
bwd = run(lambda: _spmm(g.gcn.t_rowptr, g.gcn.t_col, g.gcn_val, g.gcn.t_perm, x, N))
print(f"F={F} BPX={os.environ.get('CEEG_SPMM_BPX','default')} fwd {fwd:.3f} ms ({bytes_alg/fwd/1e6:.0f} GB/s alg)  transposed {bwd:.3f} ms ({bytes_alg/bwd/1e6:.0f} GB/s alg)")
