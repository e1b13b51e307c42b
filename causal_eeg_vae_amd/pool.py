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


class _GraphLayerNormFn(torch.autograd.Function):
    """ceeg_graph_layernorm_fwd / _bwd: per-graph statistics, normalise + affine (+ ReLU) and the whole backward as fused
    streaming kernels (two reads of x and one write each way)."""

    @staticmethod
    def forward(ctx, x, ptr, weight, bias, eps, relu):
        _lib.require_cuda(x)
        L = _lib.lib()
        x = x.contiguous()
        N, F = x.shape
        G = ptr.shape[0] - 1
        out = torch.empty_like(x)
        stats = torch.empty(2, G, dtype=torch.float64, device=x.device)
        w = None if weight is None else weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        ws = _lib.workspace(L.ceeg_graph_layernorm_ws_bytes(N, F, G), x.device)
        with _lib.timed("graph_layernorm_fwd"):
            _lib.check(L.ceeg_graph_layernorm_fwd(x.data_ptr(), x.stride(0), ptr.data_ptr(), N, F, G, _lib.ptr(w), _lib.ptr(b), float(eps), int(relu),
                                                  out.data_ptr(), out.stride(0), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(), ws.numel(),
                                                  _lib.stream()), "ceeg_graph_layernorm_fwd")
        ctx.save_for_backward(x, out if relu else None, ptr, w, stats)
        ctx.relu, ctx.affine = bool(relu), weight is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, ptr, w, stats = ctx.saved_tensors
        L = _lib.lib()
        g = g.contiguous()
        N, F = x.shape
        G = ptr.shape[0] - 1
        dx = torch.empty_like(x)
        dwb = torch.empty(2, F, dtype=torch.float32, device=x.device) if ctx.affine else None
        ws = _lib.workspace(L.ceeg_graph_layernorm_ws_bytes(N, F, G), x.device)
        with _lib.timed("graph_layernorm_bwd"):
            _lib.check(L.ceeg_graph_layernorm_bwd(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), _lib.ptr(y), 0 if y is None else y.stride(0),
                                                  ptr.data_ptr(), N, F, G, _lib.ptr(w), stats[0].data_ptr(), stats[1].data_ptr(), int(ctx.relu),
                                                  dx.data_ptr(), dx.stride(0), None if dwb is None else dwb[0].data_ptr(),
                                                  None if dwb is None else dwb[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()),
                       "ceeg_graph_layernorm_bwd")
        return dx, None, (dwb[0] if ctx.affine else None), (dwb[1] if ctx.affine else None), None, None


class GraphLayerNorm(nn.Module):
    """PyG LayerNorm(in_channels, mode='graph'): statistics over ALL nodes and channels of each graph;
    out = (x - mean_g) / sqrt(var_g + eps) * weight + bias (eps 1e-5, biased variance).  `relu=True` fuses the ReLU the
    reference puts behind it (scripts/models.py:108) into the same pass.  in_channels % 4 == 0, <= 2048."""

    def __init__(self, in_channels, eps=1e-5, affine=True):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels)) if affine else None
        self.bias = nn.Parameter(torch.zeros(in_channels)) if affine else None

    def forward(self, x, batch, size=None, relu=False):
        _lib.require_cuda(x)
        F = x.shape[1]
        if F % 4 or F > 2048:
            # widths the fused kernels do not take: the same formula in torch ops on the device (float64 statistics)
            return _graph_layernorm_torch(x, batch, size, self.weight, self.bias, self.eps, bool(relu))
        ptr = graph_ptr(batch, size)
        if x.dtype != torch.float32:
            # the kernels read fp32: a half / bf16 activation (autocast) is widened first and the result narrowed back
            if not x.is_floating_point():
                raise TypeError(f"GraphLayerNorm: floating-point input expected, got {x.dtype}")
            return _GraphLayerNormFn.apply(x.float(), ptr, self.weight, self.bias, self.eps, bool(relu)).to(x.dtype)
        return _GraphLayerNormFn.apply(x, ptr, self.weight, self.bias, self.eps, bool(relu))


def _graph_layernorm_torch(x, batch, size, weight, bias, eps, relu):
    """Graph-mode LayerNorm in torch ops (device tensors; used for widths outside the fused kernels' range)."""
    G = int(batch[-1]) + 1 if size is None else int(size)
    xf = x.double()
    n = torch.bincount(batch, minlength=G).clamp(min=1).double() * x.shape[1]
    mean = torch.zeros(G, dtype=torch.float64, device=x.device).index_add_(0, batch, xf.sum(dim=1)) / n
    xc = xf - mean.index_select(0, batch)[:, None]
    var = torch.zeros(G, dtype=torch.float64, device=x.device).index_add_(0, batch, (xc * xc).sum(dim=1)) / n
    out = (xc / torch.sqrt(var + eps).index_select(0, batch)[:, None]).to(x.dtype)
    if weight is not None:
        out = out * weight + bias
    return torch.relu(out) if relu else out


class _TokenPool(torch.autograd.Function):
    """x[n] = sum_t (<hidden[n,t], w> + b) hidden[n,t]  (ceeg_token_pool_fwd / _bwd): the reference's
    `torch.bmm(hidden.permute(0, 2, 1), self.attn(hidden)).squeeze(2)` (scripts/models.py:118-119) in one pass over hidden."""

    @staticmethod
    def forward(ctx, hidden, weight, bias):
        _lib.require_cuda(hidden, weight)
        L = _lib.lib()
        hidden = hidden.contiguous()
        N, T, H = hidden.shape
        w = weight.detach().reshape(-1).contiguous()
        b = None if bias is None else bias.detach().reshape(-1).contiguous()
        a = torch.empty(N, T, dtype=torch.float32, device=hidden.device)
        x = torch.empty(N, H, dtype=torch.float32, device=hidden.device)
        with _lib.timed("token_pool_fwd"):
            _lib.check(L.ceeg_token_pool_fwd(hidden.data_ptr(), N, T, H, w.data_ptr(), _lib.ptr(b), a.data_ptr(), x.data_ptr(), _lib.stream()),
                       "ceeg_token_pool_fwd")
        ctx.save_for_backward(hidden, w, a)
        ctx.wshape, ctx.has_bias = weight.shape, bias is not None
        return x

    @staticmethod
    def backward(ctx, g):
        from .models import _colsum
        hidden, w, a = ctx.saved_tensors
        L = _lib.lib()
        N, T, H = hidden.shape
        g = g.contiguous()
        dh = torch.empty_like(hidden) if ctx.needs_input_grad[0] else None
        dw_part = torch.empty(N, H, dtype=torch.float32, device=hidden.device)
        db_part = torch.empty(N, 4, dtype=torch.float32, device=hidden.device)
        with _lib.timed("token_pool_bwd"):
            _lib.check(L.ceeg_token_pool_bwd(hidden.data_ptr(), N, T, H, w.data_ptr(), a.data_ptr(), g.data_ptr(), _lib.ptr(dh), dw_part.data_ptr(),
                                             db_part.data_ptr(), _lib.stream()), "ceeg_token_pool_bwd")
        dw = torch.cat([_colsum(dw_part[:, c0:c0 + 1024]) for c0 in range(0, H, 1024)]).view(ctx.wshape)
        db = _colsum(db_part)[:1] if ctx.has_bias else None
        return dh, dw, db


def token_attention_pool(hidden, attn):
    """`attn`: the nn.Linear(H, 1) of the reference's conditioners; hidden [N, T, H] -> [N, H].
    The kernels read fp32: a half / bf16 `hidden` (an autocast token encoder) is widened first and the result narrowed back
    to its dtype - never reinterpreted."""
    _lib.require_cuda(hidden)
    if not hidden.is_floating_point():
        raise TypeError(f"token_attention_pool: floating-point hidden states expected, got {hidden.dtype}")
    if hidden.dtype != torch.float32 or attn.weight.dtype != torch.float32:
        b = None if attn.bias is None else attn.bias.float()
        return _TokenPool.apply(hidden.float(), attn.weight.float(), b).to(hidden.dtype)
    return _TokenPool.apply(hidden, attn.weight, attn.bias)
