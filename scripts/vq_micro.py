"""VectorQuantizer forward at the north-star step's shape (1M x 512 x 128 fp32 by default) for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from causal_eeg_vae_amd import VectorQuantizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
NE = int(sys.argv[4]) if len(sys.argv) > 4 else 512
dev = torch.device("cuda:0")
torch.manual_seed(0)
z = torch.randn(B, D, device=dev).to(dt)
vq = VectorQuantizer(NE, D, 0.25, return_one_hot=False).to(dev)
with torch.no_grad():
    vq.embedding.weight.normal_()
    for _ in range(6):
        vq(z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    e0.record()
    for _ in range(10):
        vq(z)
    e1.record()
torch.cuda.synchronize()
print("ms/call", e0.elapsed_time(e1) / 10)
