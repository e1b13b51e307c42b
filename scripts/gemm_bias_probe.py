"""Probe (GPU box): the two forward products of config 2 ([1 M, 256] x [256, 256] and [1 M, 128] x [128, 256]) as plain mm, addmm with
bias and _addmm_activation (bias + ReLU in the epilogue), with the tuned GEMM table loaded.  python scripts/gemm_bias_probe.py"""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import _lib
_lib.enable_tuned_gemms()
N = 1_000_000
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for K, M in ((256, 256), (128, 256)):
    x = torch.randn(N, K, device="cuda"); w = torch.randn(M, K, device="cuda"); b = torch.randn(M, device="cuda")
    wt = w.t().contiguous()
    out = torch.empty(N, M, device="cuda")
    print(K, M, "mm(x, wt)", round(t(lambda: torch.mm(x, wt, out=out)), 3),
          "addmm(b, x, wt)", round(t(lambda: torch.addmm(b, x, wt, out=out)), 3),
          "addmm_act relu", round(t(lambda: torch._addmm_activation(b, x, wt, use_gelu=False)), 3),
          "mm(x, w.t())", round(t(lambda: torch.mm(x, w.t(), out=out)), 3),
          "linear", round(t(lambda: torch.nn.functional.linear(x, w, b)), 3))
