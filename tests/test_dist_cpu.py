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


class _Toy(torch.nn.Module):
    """A CPU stand-in with the call convention Trainer.step drives (model(x, graph, step=, eps=) -> (loss, aux))."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 8)
        self.b = torch.nn.Linear(8, 2)

    def forward(self, x, graph, step=0, eps=None):
        h = torch.tanh(self.a(x)) * (1.0 + 0.1 * step)
        return self.b(h).square().mean(), {}


def _trainer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from causal_eeg_vae_amd.trainer import Trainer
    torch.manual_seed(100 + rank)            # different init per rank: the Trainer's broadcast must make them equal
    tr = Trainer(_Toy(), lr=1e-2, world_size=world)
    gen = torch.Generator().manual_seed(7 + rank)   # a different shard per rank
    snaps, losses = [], []
    for _ in range(4):                        # zero_grad -> forward -> backward -> all-reduce -> Adam, four times
        x = torch.randn(16, 6, generator=gen)
        loss, _ = tr.step(x, None)
        losses.append(float(loss))
        snaps.append(torch.cat([p.detach().flatten() for p in tr.model.parameters()]).clone())
    q.put((rank, torch.stack(snaps).numpy(), losses))   # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_step_exchange_keeps_replicas_bit_identical_world2():
    """Trainer.step's exchange order over 4 optimizer steps on 2 ranks with different shards: after every step the replicas
    are bit-identical, and they equal a single process that steps on the average of the two shards' gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, snaps_a, losses_a), (_, snaps_b, losses_b) = res
    assert losses_a != losses_b                                  # different shards
    snaps_a, snaps_b = torch.from_numpy(snaps_a), torch.from_numpy(snaps_b)
    for sa, sb in zip(snaps_a, snaps_b):
        assert torch.equal(sa, sb)                               # every step: identical replicas
    # single-process restatement: rank 0's initial weights, gradients of both shards averaged
    torch.manual_seed(100)
    m = _Toy()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    gens = [torch.Generator().manual_seed(7 + r) for r in range(2)]
    for step in range(4):
        gs = []
        for g in gens:
            m.zero_grad()
            m(torch.randn(16, 6, generator=g), None, step=step)[0].backward()
            gs.append([p.grad.clone() for p in m.parameters()])
        for p, g0, g1 in zip(m.parameters(), *gs):
            p.grad = (g0 + g1) / 2
        opt.step()
        torch.testing.assert_close(torch.cat([p.detach().flatten() for p in m.parameters()]), snaps_a[step], rtol=1e-6, atol=1e-7)
