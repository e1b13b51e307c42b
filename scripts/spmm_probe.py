"""Measurement helper (not product): the SpMM on BASELINE config 2's graph - what bounds it?
Times (HIP events on the launch stream, median of `reps`) of
  * ceeg_spmm_csr forward and transposed at F = 256 (the roofline launches of bench.py),
  * the pure-gather probe (ceeg_spmm_gather_probe): all entries / only near ones (|col-row| <= window) / only far ones,
  * the same with CEEG_SPMM_NT = 1, 2, 3 (non-temporal out stores / col-val loads),
  * optionally after a hub-first relabelling of the nodes (--relabel): nodes sorted by how often they are gathered.
Prints one JSON line per measurement.  Usage: python scripts/spmm_probe.py [--relabel] [--N 1000000 --E 16000000]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import EventGraph, _lib  # noqa: E402
from causal_eeg_vae_amd.models import _spmm  # noqa: E402
from causal_eeg_vae_amd.synth import synth_event_graph  # noqa: E402


def timeit(fn, reps=15, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--E", type=int, default=16_000_000)
    ap.add_argument("--F", type=int, default=256)
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--relabel", action="store_true")
    ap.add_argument("--only", default="", help="'probe' or 'spmm': run only that group (short PMC runs)")
    ap.add_argument("--reps", type=int, default=15)
    a = ap.parse_args()
    N, E, F = a.N, a.E, a.F
    dev = torch.device("cuda:0")
    ei = synth_event_graph(N, E, seed=0, device=dev)
    if a.relabel:  # hub-first: new id = rank by descending number of times the node is gathered (as a source)
        cnt = torch.bincount(ei[0], minlength=N)
        order = torch.argsort(cnt, descending=True, stable=True)
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(N, device=dev)
        ei = new_id[ei]
    g = EventGraph(ei, N)
    p = g.gcn
    x = torch.randn(N, F, device=dev)
    L = _lib.lib()
    ws = _lib.workspace(L.ceeg_spmm_ws_bytes(N, p.nnz, F) + 4096, dev)
    alg = p.nnz * 8 + (N + 1) * 4 + 2 * N * F * 4
    near_f = int(((p.col.long() - torch.repeat_interleave(torch.arange(N, device=dev), (p.rowptr[1:] - p.rowptr[:-1]).long())).abs() <= a.window).sum())
    print(json.dumps(dict(what="graph", N=N, nnz=p.nnz, F=F, relabel=a.relabel, near_entries=near_f, far_entries=p.nnz - near_f, window=a.window)))

    def spmm(tr):
        if tr:
            return lambda: _spmm(p.t_rowptr, p.t_col, g.gcn_tval, None, x, N, max_row=p.max_trow, items=p.long_items(True))
        return lambda: _spmm(p.rowptr, p.col, g.gcn_val, None, x, N, max_row=p.max_row, items=p.long_items(False))

    def probe(tr, mode):
        rp, col = (p.t_rowptr, p.t_col) if tr else (p.rowptr, p.col)
        return lambda: _lib.check(L.ceeg_spmm_gather_probe(rp.data_ptr(), col.data_ptr(), x.data_ptr(), x.stride(0), N, p.nnz, F, mode, a.window,
                                                           ws.data_ptr(), ws.numel(), _lib.stream()), "probe")

    for nt in (("0", "1", "2", "3") if a.only in ("", "spmm") else ()):
        os.environ["CEEG_SPMM_NT"] = nt
        for tr in (False, True):
            ms = timeit(spmm(tr), a.reps)
            print(json.dumps(dict(what="spmm", transposed=tr, nt=int(nt), ms=round(ms, 4), alg_TBps=round(alg / ms / 1e9, 3))))
    os.environ["CEEG_SPMM_NT"] = "0"
    for tr in ((False, True) if a.only in ("", "probe") else ()):
        for mode, name in ((3, "all"), (1, "near"), (2, "far"), (7, "all, feature-sliced per XCD"), (5, "near, feature-sliced per XCD"),
                           (6, "far, feature-sliced per XCD")):
            ms = timeit(probe(tr, mode), a.reps)
            n_rows = p.nnz if (mode & 3) == 3 else (near_f if (mode & 3) == 1 else p.nnz - near_f)
            print(json.dumps(dict(what="gather_probe", transposed=tr, entries=name, ms=round(ms, 4), gathered_GB=round(n_rows * F * 4 / 1e9, 2),
                                  gather_TBps=round(n_rows * F * 4 / ms / 1e9, 2))))


if __name__ == "__main__":
    main()
