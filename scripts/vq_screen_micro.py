"""Measurement helper (GPU box): ceeg_vq_forward fp32 at the in-step shape (1 M x 512 x 128), screened vs full kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import _lib  # noqa: E402

B, n_e, D = 1_000_000, 512, 128
dev = torch.device("cuda")
L = _lib.lib()
torch.manual_seed(0)
z = torch.randn(B, D, device=dev)
W = torch.randn(n_e, D, device=dev) * float(os.environ.get("W_SCALE", "0.1"))
zq = torch.empty_like(z)
idx = torch.empty(B, dtype=torch.int64, device=dev)
sc = torch.empty(2, device=dev)
ws = _lib.workspace(L.ceeg_vq_forward_ws_bytes(B, n_e, D, _lib.CEEG_F32), dev)


def run():
    _lib.check(L.ceeg_vq_forward(z.data_ptr(), W.data_ptr(), B, n_e, D, _lib.CEEG_F32, 0.25, zq.data_ptr(), idx.data_ptr(), sc[0:].data_ptr(),
                                 sc[1:].data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream()), "vq")


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    run()
e.record()
torch.cuda.synchronize()
print(f"CEEG_VQ_SCREEN={os.environ.get('CEEG_VQ_SCREEN', '1')}: {s.elapsed_time(e) / 10:.3f} ms per forward (prep + main + histogram + finalize)  "
      f"idxsum {int(idx.sum())} loss {float(sc[0]):.6f}")
