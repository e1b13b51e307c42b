"""One-rank RCCL rehearsal on one GPU (a two-rank group needs two devices): the `nccl` branch of trainer.allreduce_grads_ /
broadcast_params_ and bench.rank_report on device tensors, stream-ordered behind real backward kernels.  Prints one JSON line.
    python scripts/rccl_single_rank.py"""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    from causal_eeg_vae_amd import EventGraph, VGAEVQ
    from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
    from causal_eeg_vae_amd.trainer import Trainer, allreduce_grads_, broadcast_params_
    N, E = 20000, 300000
    graph = EventGraph(synth_event_graph(N, E, seed=3, device=dev), N)
    x = synth_features(N, 64, seed=4, device=dev)
    torch.manual_seed(0)
    model = VGAEVQ(64, 32, 64, 0.25, negatives="uniform").to(dev)
    before = [p.detach().clone() for p in model.parameters()]
    broadcast_params_(model)                                   # rank 0 -> rank 0 over RCCL: values unchanged
    same = all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    trainer = Trainer(model, lr=1e-3, world_size=1)
    trainer.opt.zero_grad(set_to_none=True)
    loss, _ = model(x, graph, step=0)
    loss.backward()
    grads = [p.grad.clone() for p in trainer.params]
    # the collective is queued right behind the backward's kernels; "world_size" 2 with one rank: sum over one rank, then / 2
    allreduce_grads_(trainer.params, 2)
    halved = all(torch.equal(p.grad, g / 2) for p, g in zip(trainer.params, grads))
    t = torch.tensor([0.125], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "broadcast_unchanged": bool(same), "allreduce_sum_div2_exact": bool(halved),
           "max_reduce": float(t.item()), "loss": float(loss.item()), "seconds": time.perf_counter() - t0}
    dist.destroy_process_group()
    print(json.dumps(out))
    return 0 if same and halved else 1


if __name__ == "__main__":
    sys.exit(main())
