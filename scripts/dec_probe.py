"""Measurement helper: how much of the code-form decoder backward (dc_grad_rows_kernel) is the random 16-byte record gathers of
the transposed list?  Times ceeg_decoder_codes_bwd on the config-2 graph with the real t_perm and with t_perm = identity
(sequential records: wrong gradient, same work otherwise)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import EventGraph, _lib, sigma_params
from causal_eeg_vae_amd.synth import synth_event_graph

N, E, D, K = 1_000_000, 16_000_000, 128, 512
ei = synth_event_graph(N, E, seed=0, device="cuda")
g = EventGraph(ei, N)
p = g.dec
L = _lib.lib()
gen = torch.Generator(device="cuda").manual_seed(1)
W = torch.randn(K, D, device="cuda", generator=gen) * 0.2
code = torch.randint(0, K, (N,), device="cuda", generator=gen, dtype=torch.int32)
a, b, ainv = sigma_params(N, 0, 3)
li, lt = p.long_items(False), p.long_items(True)
nl, nt = (0 if li is None else li.shape[0]), (0 if lt is None else lt.shape[0])
rec = torch.empty(p.nnz, 4, dtype=torch.int32, device="cuda")
sums = torch.empty(3, device="cuda")
ws = _lib.workspace(L.ceeg_decoder_codes_ws_bytes(N, p.nnz, K, D, nl, nt), "cuda")
_lib.check(L.ceeg_decoder_codes_fwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), p.rowptr.data_ptr(), p.col.data_ptr(), N, p.nnz, a, b, N,
                                    _lib.ptr(li), nl, p.max_row, None, rec.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "fwd")
gs = torch.tensor([1.0 / E, 1.0 / E], device="cuda")
dz = torch.empty(N, D, device="cuda")
ident = torch.arange(p.nnz, dtype=torch.int32, device="cuda")

def run(tperm, sig_a):
    _lib.check(L.ceeg_decoder_codes_bwd(W.data_ptr(), W.stride(0), K, D, p.rowptr.data_ptr(), p.t_rowptr.data_ptr(), tperm.data_ptr(), N, p.nnz,
                                        sig_a, b if sig_a else 0, ainv if sig_a else 1, N, _lib.ptr(li), nl, _lib.ptr(lt), nt, p.max_row, p.max_trow,
                                        rec.data_ptr(), gs.data_ptr(), dz.data_ptr(), dz.stride(0), None, ws.data_ptr(), ws.numel(), _lib.stream()), "bwd")

def timeit(fn, n=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]

print(json.dumps({"bwd_ms": timeit(lambda: run(p.t_perm, a)), "bwd_identity_tperm_ms": timeit(lambda: run(ident, a)),
                  "bwd_posonly_ms": timeit(lambda: run(p.t_perm, 0)), "bwd_posonly_identity_tperm_ms": timeit(lambda: run(ident, 0))}))
