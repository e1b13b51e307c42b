"""Per-graph reductions over a collated batch (`batch` = sorted graph id per node, as PyG's DataLoader builds it):
GlobalAttention pooling (scripts/models.py:112-113,122) and graph-mode LayerNorm (scripts/models.py:107, PyG
`LayerNorm(mode='graph')`).  The segment sums run through libceeg's fixed-order kernels (no atomics): the segment
softmax is ceeg_gat_softmax_fwd with rows = graphs, the weighted sum is ceeg_spmm_csr with rows = graphs."""
import torch
import torch.nn as nn

from . import _lib
from .models import _spmm


def graph_ptr(batch, num_graphs=None):
    """[G+1] int32 offsets of the graphs in a sorted `batch` vector."""
    if batch.numel() > 1 and bool((batch[1:] < batch[:-1]).any()):
        raise ValueError("batch must be sorted (nodes of a graph contiguous), as collate_graphs / PyG produce it")
    G = int(batch[-1]) + 1 if num_graphs is None else int(num_graphs)
    counts = torch.bincount(batch, minlength=G)
    ptr = torch.zeros(G + 1, dtype=torch.int32, device=batch.device)
    ptr[1:] = counts.cumsum(0)
    return ptr


def segment_sum(v, ptr):
    """out[g] = sum of v over graph g (v: [N] fp32), fixed order."""
    L = _lib.lib()
    v = v.contiguous()
    G = ptr.shape[0] - 1
    out = torch.empty(G, dtype=torch.float32, device=v.device)
    _lib.check(L.ceeg_segment_sum_perm(ptr.data_ptr(), None, v.data_ptr(), G, out.data_ptr(), _lib.stream()), "ceeg_segment_sum_perm")
    return out


class _AttentionPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate, ptr, batch):
        _lib.require_cuda(x)
        L = _lib.lib()
        x, gate = x.contiguous(), gate.contiguous()
        N, F = x.shape
        G = ptr.shape[0] - 1
        ident = torch.arange(N, dtype=torch.int32, device=x.device)
        zeros = torch.zeros(G, dtype=torch.float32, device=x.device)
        w = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)[:N]
        # softmax of the gate inside each graph: rows = graphs, entries = their nodes, slope 1 = no leaky relu
        _lib.check(L.ceeg_gat_softmax_fwd(ptr.data_ptr(), ident.data_ptr(), gate.data_ptr(), zeros.data_ptr(), G, N, 1.0,
                                          w.data_ptr(), _lib.stream()), "ceeg_gat_softmax_fwd")
        out = torch.empty(G, F, dtype=torch.float32, device=x.device)
        for f0 in range(0, F, 1024):   # ceeg_spmm_csr takes up to 1024 columns per call
            f1 = min(F, f0 + 1024)
            _spmm(ptr, ident, w, None, x[:, f0:f1], G, out=out[:, f0:f1])
        ctx.save_for_backward(x, w, ptr, batch)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, ptr, batch = ctx.saved_tensors
        gn = g.contiguous().index_select(0, batch)            # g[batch[n]]
        dx = gn * w[:, None]
        dw = (gn * x).sum(dim=1)
        s = segment_sum(w * dw, ptr)                          # sum_m w_m dw_m per graph
        dgate = w * (dw - s.index_select(0, batch))
        return dx, dgate, None, None


class GlobalAttention(nn.Module):
    """PyG GlobalAttention(gate_nn): out[g] = sum_{n in g} softmax_g(gate_nn(x_n)) * x_n."""

    def __init__(self, gate_nn):
        super().__init__()
        self.gate_nn = gate_nn

    def forward(self, x, batch, size=None):
        ptr = graph_ptr(batch, size)
        gate = self.gate_nn(x).view(-1)
        return _AttentionPool.apply(x, gate, ptr, batch)


class GraphLayerNorm(nn.Module):
    """PyG LayerNorm(in_channels, mode='graph'): statistics over ALL nodes and channels of each graph;
    out = (x - mean_g) / sqrt(var_g + eps) * weight + bias (eps 1e-5, biased variance)."""

    def __init__(self, in_channels, eps=1e-5, affine=True):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels)) if affine else None
        self.bias = nn.Parameter(torch.zeros(in_channels)) if affine else None

    def forward(self, x, batch, size=None):
        ptr = graph_ptr(batch, size)
        cnt = (ptr[1:] - ptr[:-1]).to(torch.float32).clamp(min=1) * x.shape[1]
        mean = _SegRowSum.apply(x, ptr, batch) / cnt
        xc = x - mean.index_select(0, batch)[:, None]
        var = _SegRowSum.apply(xc * xc, ptr, batch) / cnt
        out = xc / (var + self.eps).sqrt().index_select(0, batch)[:, None]
        if self.weight is not None:
            out = out * self.weight + self.bias
        return out


class _SegRowSum(torch.autograd.Function):
    """[N, F] -> [G]: sum of all entries of each graph's rows (row sums by torch, graph sums by the fixed-order kernel)."""

    @staticmethod
    def forward(ctx, x, ptr, batch):
        ctx.save_for_backward(batch)
        ctx.F = x.shape[1]
        return segment_sum(x.sum(dim=1), ptr)

    @staticmethod
    def backward(ctx, g):
        (batch,) = ctx.saved_tensors
        return g.index_select(0, batch)[:, None].expand(-1, ctx.F), None, None
