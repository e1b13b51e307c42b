"""Does a hipBLASLt fp32 product overlap with a gather-bound SpMM launch when they sit on two streams?
Measurement helper for DESIGN.md section 7 (0): serial vs concurrent time of one F=256 aggregation and one
[N,256]x[256,256] product on the config-2 graph."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import EventGraph
from causal_eeg_vae_amd.models import _spmm
from causal_eeg_vae_amd.synth import synth_event_graph

N, E = 1_000_000, 16_000_000
ei = synth_event_graph(N, E, seed=0, device="cuda")
g = EventGraph(ei, N)
x = torch.randn(N, 256, device="cuda")
h = torch.randn(N, 256, device="cuda")
w = torch.randn(256, 256, device="cuda")
out = torch.empty(N, 256, device="cuda")
y = torch.empty(N, 256, device="cuda")
li = g.gcn.long_items(False)
s2 = torch.cuda.Stream()

def spmm():
    _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x, N, out=out, max_row=g.gcn.max_row, items=li)

def gemm():
    torch.mm(h, w, out=y)

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        gemm()
    spmm()
    torch.cuda.current_stream().wait_stream(s2)

def both_rev():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        spmm()
    gemm()
    torch.cuda.current_stream().wait_stream(s2)

res = dict(spmm_ms=timed(spmm), gemm_ms=timed(gemm), serial_ms=timed(lambda: (spmm(), gemm())), concurrent_ms=timed(both),
           concurrent_rev_ms=timed(both_rev))
print(json.dumps(res))
