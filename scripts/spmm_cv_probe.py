"""Measurement helper (GPU box): ceeg_spmm_csr_items vs ceeg_spmm_csr_cv (interleaved (col, val) stream) on the config-2 graph,
forward and transposed, F = 256 and 128; checks bitwise equality."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import EventGraph, _lib  # noqa: E402
from causal_eeg_vae_amd.models import _spmm  # noqa: E402
from causal_eeg_vae_amd.synth import synth_event_graph  # noqa: E402

N, E = 1_000_000, 16_000_000
dev = torch.device("cuda")
torch.manual_seed(0)
L = _lib.lib()
g = EventGraph(synth_event_graph(N, E, seed=1000, device=dev), N)
p = g.gcn


def cv(col, val):
    return torch.stack([col, val.view(torch.int32)], dim=1).contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for F in (256, 128):
    x = torch.randn(N, F, device=dev)
    for name, rp, col, val, mr, items in (("forward", p.rowptr, p.col, g.gcn_val, p.max_row, p.long_items(False)),
                                          ("transposed", p.t_rowptr, p.t_col, g.gcn_tval, p.max_trow, p.long_items(True))):
        packed = cv(col, val)
        out_a = torch.empty(N, F, device=dev)
        out_b = torch.empty(N, F, device=dev)
        ws = _lib.workspace(L.ceeg_spmm_ws_bytes(N, p.nnz, F), dev)

        def a():
            _spmm(rp, col, val, None, x, N, out=out_a, max_row=mr, items=items)

        def b():
            _lib.check(L.ceeg_spmm_csr_cv(rp.data_ptr(), packed.data_ptr(), x.data_ptr(), x.stride(0), N, p.nnz, F, None, 0, None, 0, out_b.data_ptr(),
                                          out_b.stride(0), mr, _lib.ptr(items), 0 if items is None else items.shape[0], ws.data_ptr(), ws.numel(),
                                          _lib.stream()), "cv")

        ta, tb = timeit(a), timeit(b)
        chk = int(out_b.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff  # compares runs with different CEEG_SPMM_* switches
        print(f"F={F} {name}: separate streams {ta:.3f} ms, interleaved {tb:.3f} ms, equal {torch.equal(out_a, out_b)}, bits {chk:012x}")
