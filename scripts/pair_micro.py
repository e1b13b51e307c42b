"""Measurement helper (GPU box): the literal sampler path's forward on the config-2 graph - stand-alone sampler + pair-list forward
against the fused launch (ceeg_neg_sample_pair_bce_codes_fwd); checks that samples, keys and sums are identical.
Usage: python scripts/pair_micro.py [nodes edges]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import EventGraph, _lib  # noqa: E402
from causal_eeg_vae_amd.models import _neg_filter, negative_sampling_raw  # noqa: E402
from causal_eeg_vae_amd.synth import synth_event_graph  # noqa: E402

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 16_000_000)
K, D = 512, 128
dev = torch.device("cuda")
L = _lib.lib()
g = EventGraph(synth_event_graph(N, E, seed=1000, device=dev), N, with_dec=True)
p = g.dec
torch.manual_seed(0)
W = torch.randn(K, D, device=dev) * 0.1
code = torch.randint(0, K, (N,), device=dev, dtype=torch.int32)
filt, bits = _neg_filter(p)
ws = _lib.workspace(L.ceeg_pair_bce_codes_ws_bytes(E, N, K), dev)


def timeit(fn, n=10):
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


neg = [None]
keys_a = torch.empty(2 * E, dtype=torch.int32, device=dev)
keys_b = torch.empty(2 * E, dtype=torch.int32, device=dev)
ctab = torch.empty(K * K, device=dev)
out_a, out_b = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
neg_b = torch.empty(2, E, dtype=torch.int32, device=dev)


def sampler():
    neg[0] = negative_sampling_raw(g, N, E, 12345)


def pair_fwd():
    i, j = neg[0][0], neg[0][1]
    _lib.check(L.ceeg_pair_bce_codes_fwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), i.data_ptr(), j.data_ptr(), E, N, 1, keys_a.data_ptr(),
                                         ctab.data_ptr(), out_a.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "pair fwd")


def fused():
    _lib.check(L.ceeg_neg_sample_pair_bce_codes_fwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), p.rowptr.data_ptr(), p.col.data_ptr(), N, E, 12345,
                                                    _lib.ptr(filt), bits, neg_b[0].data_ptr(), neg_b[1].data_ptr(), keys_b.data_ptr(), ctab.data_ptr(),
                                                    out_b.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "fused")


t_s, t_p, t_f = timeit(sampler), timeit(pair_fwd), timeit(fused)
same = torch.equal(neg[0], neg_b) and torch.equal(keys_a, keys_b) and torch.equal(out_a, out_b)
print(f"sampler {t_s:.3f} ms + pair forward {t_p:.3f} ms = {t_s + t_p:.3f} ms; fused {t_f:.3f} ms; identical: {same}; pairs counted {int(out_b[1])}")
