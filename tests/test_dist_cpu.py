"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (one flat all-reduce per step) and the
parameter broadcast used by bench.py / trainer.py on RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from causal_eeg_vae_amd.trainer import allreduce_grads_, broadcast_params_
    torch.manual_seed(rank)  # different init per rank on purpose
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    broadcast_params_(m)
    w0 = torch.cat([p.detach().flatten() for p in m.parameters()])
    x = torch.full((4, 5), float(rank + 1))
    m(x).sum().backward()
    g_local = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    allreduce_grads_(list(m.parameters()), world)
    g_avg = torch.cat([p.grad.flatten() for p in m.parameters()])
    q.put((rank, w0, g_local, g_avg))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_averages_gradients_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w_a, gl_a, ga_a), (_, w_b, gl_b, ga_b) = res
    assert torch.equal(w_a, w_b)                                   # broadcast made the replicas identical
    assert not torch.equal(gl_a, gl_b)                             # different shards -> different local grads
    torch.testing.assert_close(ga_a, (gl_a + gl_b) / 2)            # one all-reduce(sum) / world
    assert torch.equal(ga_a, ga_b)                                 # every rank steps with the same gradient
