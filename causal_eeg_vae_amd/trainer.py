"""Training step of the VGAE + VQ model: the host loop of /root/reference/scripts/run.py:54-98 restated
(Adam lr 1e-3, zero_grad -> forward -> sum of losses -> backward -> step), plus data-parallel gradient
averaging over RCCL (one flat all-reduce per step; one process per GPU, graph shards never exchange data)."""
import torch
import torch.distributed as dist


def allreduce_grads_(params, world_size, force=False):
    """Average gradients across ranks with ONE flat all-reduce (sum) followed by 1/world_size.  force=True runs the exchange
    on a one-rank group too (rehearsal of the collective path on a single GPU).

    Correct by construction for either backend:
      * `nccl` (= RCCL): the collective is enqueued behind the backward's kernels (ProcessGroupNCCL makes its stream wait on
        an event of the current stream, and `wait()` - implied by the synchronous call - makes the current stream wait for
        the collective), `flat` stays referenced until after that wait, and the division and the copies back are ordinary
        kernels on the current stream, i.e. ordered after it;
      * any other backend (gloo: CPU tests, one-GPU rehearsals) never sees device memory: the current stream is
        synchronised, the flat bucket crosses to the host with a blocking copy, is reduced there and copied back.
        (Round 1: two ranks sharing one GPU dead-locked in the SECOND gloo all-reduce of a device tensor, both main threads
        inside work.wait(); DESIGN.md section 6 has what the trace does and does not show.)"""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or (world_size <= 1 and not force):
        return
    if len(grads) != len(params):
        # every rank must reduce the same bucket layout: a parameter without a gradient on one rank only would silently
        # misalign (or hang) the exchange
        raise RuntimeError("allreduce_grads_: some parameters have no gradient; the flat bucket would differ between ranks")
    flat = torch._utils._flatten_dense_tensors(grads)
    if flat.is_cuda and dist.get_backend() != "nccl":
        torch.cuda.current_stream(flat.device).synchronize()
        host = flat.to("cpu", non_blocking=False)
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host, non_blocking=False)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)  # RCCL: in place on the device, stream-ordered
    flat.div_(world_size)
    # back into the parameters' gradients with one multi-tensor copy (one launch instead of one per parameter)
    torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))


def broadcast_params_(module, src=0):
    """Rank `src`'s parameters and buffers to every rank.  As in allreduce_grads_, only `nccl` is handed device memory:
    under any other backend a device tensor crosses through a host copy."""
    staged = dist.get_backend() != "nccl"
    for t in list(module.parameters()) + list(module.buffers()):
        if staged and t.is_cuda:
            host = t.data.to("cpu")
            dist.broadcast(host, src=src)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src)


class Trainer:
    def __init__(self, model, lr=1e-3, world_size=1, force_exchange=False):
        self.model = model
        self.world_size = int(world_size)
        self.force_exchange = bool(force_exchange)   # broadcast + all-reduce even with one rank (one-GPU RCCL rehearsal)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if any(p.is_cuda for p in self.params):
            from . import _lib
            _lib.enable_tuned_gemms()   # library kernels picked for the step's GEMM shapes on an MI355X (lookup table, no tuning)
        # scripts/run.py:56; on the GPU the single-kernel (fused) form of the same update: 7 launches -> 1 per step
        self.opt = torch.optim.Adam(self.params, lr=lr, fused=all(p.is_cuda for p in self.params))
        self.sched = torch.optim.lr_scheduler.ReduceLROnPlateau(self.opt)  # scripts/run.py:57 (defaults)
        self.step_idx = 0
        if self.world_size > 1 or self.force_exchange:
            broadcast_params_(model)

    def end_epoch(self, val_loss):
        """scheduler.step(val_loss) once per epoch (scripts/run.py:131); every rank must pass the same value."""
        self.sched.step(float(val_loss))

    def step(self, x, graph, *model_args, eps=None, **model_kwargs):
        """One optimizer step.  Extra positional / keyword arguments go to the model (ConditionalVGAEVQ takes
        `batch, condition` and `nodes_per_graph=`)."""
        self.opt.zero_grad(set_to_none=True)                 # scripts/run.py:87
        loss, aux = self.model(x, graph, *model_args, step=self.step_idx, eps=eps, **model_kwargs)  # scripts/run.py:89-95
        loss.backward()                                      # scripts/run.py:96
        allreduce_grads_(self.params, self.world_size, self.force_exchange)
        self.opt.step()                                      # scripts/run.py:97
        self.step_idx += 1
        return loss.detach(), aux                            # no host sync here (run.py:98 syncs every step)
