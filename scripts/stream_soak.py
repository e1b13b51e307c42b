"""Soak (GPU box): 120 minibatches of random graph counts and sizes through MinibatchStream with the side-stream loader and with
the plain loop - every loss and the final parameters must be bitwise equal.  python scripts/stream_soak.py"""
import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causal_eeg_vae_amd import ConditionalVGAEVQ
from causal_eeg_vae_amd.collate import MinibatchStream
from causal_eeg_vae_amd.synth import synth_event_graph
from causal_eeg_vae_amd.trainer import Trainer
dev = torch.device("cuda")
rng = np.random.default_rng(0)
Fi, C, D, n_e = 28, 4, 64, 128
def minibatches(seed):
    r = np.random.default_rng(seed)
    out = []
    for k in range(120):
        G = int(r.integers(1, 9))
        n = int(r.integers(50, 3000))
        eis = [synth_event_graph(n, int(r.integers(n, 12 * n)), seed=1000 * k + g, device=dev) for g in range(G)]
        out.append((eis, [n] * G))
    return out
mbs = minibatches(1)
res = {}
for prefetch in (True, False):
    torch.manual_seed(0)
    model = ConditionalVGAEVQ(Fi, C, D, n_e, 0.25).to(dev)
    tr = Trainer(model, lr=1e-3)
    losses = []
    for k, (cb, g) in enumerate(MinibatchStream(iter(mbs), prefetch=prefetch)):
        gen = torch.Generator(device=dev).manual_seed(k)
        x = torch.randn(cb.num_nodes, Fi, device=dev, generator=gen)
        cond = torch.randn(cb.num_graphs, C, device=dev, generator=gen)
        l, _ = tr.step(x, g, cb.batch, cond, nodes_per_graph=cb.nodes_per_graph[0])
        losses.append(l)
    torch.cuda.synchronize()
    res[prefetch] = (torch.stack(losses).cpu(), [p.detach().clone() for p in model.parameters()])
same = torch.equal(res[True][0], res[False][0]) and all(torch.equal(a, b) for a, b in zip(res[True][1], res[False][1]))
print("120 minibatches of random sizes: prefetch == serial bitwise:", same, "last loss", float(res[True][0][-1]))
sys.exit(0 if same else 1)
