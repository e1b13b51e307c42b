"""Two ranks on ONE GPU over gloo: rehearses the multi-process training path (everything but RCCL itself)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import faulthandler
faulthandler.dump_traceback_later(45, exit=True)
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
def note(m): print(f"[r{rank}] {m}", flush=True)
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
t = torch.full((4,), float(rank + 1), device=dev)
dist.all_reduce(t); torch.cuda.synchronize(); note(f"allreduce cuda tensor ok {t.tolist()}")
from causal_eeg_vae_amd import EventGraph, VGAEVQ
from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
from causal_eeg_vae_amd.trainer import Trainer, allreduce_grads_
N, E = int(os.environ.get("RN", 50000)), int(os.environ.get("RE", 800000))
ei = synth_event_graph(N, E, seed=rank, device=dev); x = synth_features(N, 128, seed=rank, device=dev)
g = EventGraph(ei, N); g.dec
torch.manual_seed(0)
m = VGAEVQ(128, 128, 512, 0.25, seed=rank).to(dev)
tr = Trainer(m, world_size=world); note("trainer built")
loss, _ = m(x, g, step=0); torch.cuda.synchronize(); note(f"forward ok {float(loss):.4f}")
loss.backward(); torch.cuda.synchronize(); note("backward ok")
allreduce_grads_(tr.params, world); torch.cuda.synchronize(); note("grad allreduce ok")
for i in range(3):
    l, _ = tr.step(x, g); torch.cuda.synchronize(); note(f"step {i} loss {float(l):.4f}")
w = [p.detach().double().sum().item() for p in m.parameters()]
ws = torch.tensor(w, dtype=torch.float64, device=dev); w0 = ws.clone(); dist.broadcast(w0, 0)
note(f"replicas identical: {bool(torch.equal(ws, w0))}")
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize(); note("barrier ok")
dist.destroy_process_group()
