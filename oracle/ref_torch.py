"""CPU ORACLE (test infrastructure only) — PyTorch/NumPy restatement of the hot path's semantics.

Pinned:   RefVectorQuantizer  <- tests/golden/vq_*.npz generated from the live reference
          (scripts/quantizer.py:6-67) by tests/golden/make_vq_golden.py.
Unpinned: everything PyG-derived (gcn_norm/GCNConv, VGAE, InnerProductDecoder, recon_loss).  The
          reference does not vendor or pin torch_geometric (environment.yml:13) and holds no tests or
          fixtures for it, so these follow the formulas of SURVEY.md §8(a) rows A3-A6 and are checked
          by hand-computed known-answer tests only: "parity unpinned".

Nothing under causal_eeg_vae_amd/ imports this module.
"""
import math

import numpy as np
import torch
import torch.nn as nn

MAX_LOGSTD = 10.0  # scripts/run.py:13
EPS = 1e-15        # upstream PyG autoencoder EPS


# --------------------------------------------------------------------------------------------------
# A1/A2  VectorQuantizer — restated from scripts/quantizer.py:16-67 (same op order, same returns)
# --------------------------------------------------------------------------------------------------
class RefVectorQuantizer(nn.Module):
    def __init__(self, n_e, e_dim, beta):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = nn.Embedding(n_e, e_dim)                       # quantizer.py:22
        with torch.no_grad():
            self.embedding.weight.uniform_(-1.0 / n_e, 1.0 / n_e)       # quantizer.py:23

    def forward(self, z, force_idx=None):
        """force_idx ([B] or [B,1] int64): take these codes instead of the argmin - the checker's way to compare losses and
        gradients on rows whose top-2 gap is below the rounding of the distances (the caller asserts that gap first);
        `self.last_d` keeps the distance matrix for that assertion."""
        W = self.embedding.weight
        zz = (z * z).sum(dim=1, keepdim=True)                           # quantizer.py:43
        ee = (W * W).sum(dim=1)                                         # quantizer.py:44
        d = zz + ee - 2.0 * (z @ W.t())                                 # quantizer.py:43-45
        self.last_d = d.detach()
        idx = d.argmin(dim=1, keepdim=True)                             # quantizer.py:48
        if force_idx is not None:
            idx = force_idx.view(-1, 1).to(torch.int64)
        onehot = torch.zeros(z.shape[0], self.n_e, dtype=z.dtype, device=z.device)
        onehot.scatter_(1, idx, 1.0)                                    # quantizer.py:49-51
        zq = onehot @ W                                                 # quantizer.py:54
        loss = ((zq.detach() - z) ** 2).mean() + self.beta * ((zq - z.detach()) ** 2).mean()  # :57-58
        zq = z + (zq - z).detach()                                      # quantizer.py:61
        p = onehot.mean(dim=0)                                          # quantizer.py:64
        ppl = torch.exp(-(p * torch.log(p + 1e-10)).sum())              # quantizer.py:65
        return loss, zq, ppl, onehot, idx                               # quantizer.py:67


def vq_bf16_scores64(z_bf16, W):
    """bf16 mode oracle: exact (fp64) score = <z, bf16(W)> - |bf16(W)|^2/2 per (row, code)."""
    zb = z_bf16.to(torch.float64)
    Wb = W.to(torch.bfloat16).to(torch.float64)
    return zb @ Wb.t() - 0.5 * (Wb * Wb).sum(1)


# --------------------------------------------------------------------------------------------------
# A5  CSR build, gcn_norm, GCNConv  (upstream PyG semantics)
# --------------------------------------------------------------------------------------------------
def csr_build_np(src, dst, N, self_loops):
    """Edge list -> CSR grouped by dst, entries ordered by (dst, src, original edge id).

    self_loops=True applies add_remaining_self_loops: existing self-loops are dropped and exactly one
    (i,i) entry per node is added; its perm is the LAST existing self-loop's edge id (whose weight PyG
    keeps) or -1 when the node had none.  Returns rowptr[N+1], col[nnz], perm[nnz] (int32).
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    eid = np.arange(src.shape[0], dtype=np.int64)
    if self_loops:
        loop = src == dst
        loop_eid = np.full(N, -1, dtype=np.int64)
        if loop.any():
            np.maximum.at(loop_eid, src[loop], eid[loop])
        keep = ~loop
        src = np.concatenate([src[keep], np.arange(N, dtype=np.int64)])
        dst = np.concatenate([dst[keep], np.arange(N, dtype=np.int64)])
        eid = np.concatenate([eid[keep], loop_eid])
    order = np.lexsort((eid, src, dst))  # primary dst, then src, then edge id (stable)
    src, dst, eid = src[order], dst[order], eid[order]
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), src.astype(np.int32), eid.astype(np.int32)


def csr_transpose_np(rowptr, col, N):
    """CSR (rows=dst) -> CSR of the transpose (rows=src) ordered by (src, dst, position); t_perm = position."""
    nnz = int(rowptr[-1])
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    pos = np.arange(nnz, dtype=np.int64)
    c = col[:nnz].astype(np.int64)
    order = np.lexsort((pos, rows, c))
    t_rowptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(t_rowptr, c + 1, 1)
    t_rowptr = np.cumsum(t_rowptr)
    return t_rowptr.astype(np.int32), rows[order].astype(np.int32), pos[order].astype(np.int32)


def gcn_norm_np(rowptr, col, perm, ew, N):
    """deg[i] = sum of weights over row i (in-degree at the target incl. self-loop), float64 math."""
    nnz = int(rowptr[-1])
    if ew is None:
        w = np.ones(nnz, dtype=np.float64)
    else:
        ew = np.asarray(ew, dtype=np.float64)
        w = np.where(perm[:nnz] >= 0, ew[np.maximum(perm[:nnz], 0)], 1.0)
    rows = np.repeat(np.arange(N), np.diff(rowptr.astype(np.int64)))
    deg = np.zeros(N, dtype=np.float64)
    np.add.at(deg, rows, w)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    val = dinv[col[:nnz]] * w * dinv[rows]
    return dinv, val


def spmm_np(rowptr, col, val, X):
    """out[i] = sum_p val[p] X[col[p]] in float64."""
    N = rowptr.shape[0] - 1
    nnz = int(rowptr[-1])
    rows = np.repeat(np.arange(N), np.diff(rowptr.astype(np.int64)))
    out = np.zeros((N, X.shape[1]), dtype=np.float64)
    np.add.at(out, rows, np.asarray(val[:nnz], dtype=np.float64)[:, None] * np.asarray(X, dtype=np.float64)[col[:nnz]])
    return out


def glorot_(w):
    a = math.sqrt(6.0 / (w.shape[0] + w.shape[1]))
    with torch.no_grad():
        w.uniform_(-a, a)


class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        glorot_(self.weight)


class RefGCNConv(nn.Module):
    """out = A_hat (x W^T) + b, A_hat = D^-1/2 (A + I) D^-1/2 — PyG GCNConv: linear first (no bias inside,
    glorot), sum-aggregate norm_e * (xW)[src] at dst, then + b (zeros init)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = _Lin(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, edge_weight=None):
        N = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        w = torch.ones(src.shape[0], dtype=x.dtype) if edge_weight is None else edge_weight
        keep = src != dst
        loop_w = torch.ones(N, dtype=x.dtype)
        if edge_weight is not None and (~keep).any():
            loop_w[src[~keep]] = w[~keep]  # existing loops keep their weight (last one wins)
        ar = torch.arange(N)
        src = torch.cat([src[keep], ar])
        dst = torch.cat([dst[keep], ar])
        w = torch.cat([w[keep], loop_w])
        deg = torch.zeros(N, dtype=x.dtype).index_add_(0, dst, w)
        dinv = deg.pow(-0.5)
        dinv[torch.isinf(dinv)] = 0
        norm = dinv[src] * w * dinv[dst]
        h = x @ self.lin.weight.t()
        out = torch.zeros(N, h.shape[1], dtype=x.dtype).index_add_(0, dst, norm[:, None] * h[src])
        return out + self.bias


# --------------------------------------------------------------------------------------------------
# A3/A4/A6  VGAE pieces (upstream PyG semantics)
# --------------------------------------------------------------------------------------------------
def reparametrize(mu, logstd, eps, log_var=False):
    """z = mu + eps*exp(logstd) (PyG log-std form) or mu + eps*exp(0.5*log_var) (scripts/models.py:38-43)."""
    return mu + eps * torch.exp(0.5 * logstd if log_var else logstd)


def kl_loss(mu, logstd):
    logstd = logstd.clamp(max=MAX_LOGSTD)
    return -0.5 * torch.mean(torch.sum(1 + 2 * logstd - mu ** 2 - logstd.exp() ** 2, dim=1))


class _SigmoidOnF32Grid(torch.autograd.Function):
    """float64 form of the reference's fp32 `torch.sigmoid` inside the literal `+EPS` loss.  torch evaluates the fp32
    sigmoid as 1 / (1 + exp(-v)) (CPU and GPU kernels alike); the three fp32 roundings of that formula are reproduced
    here - exp(-v) exact then rounded, the sum rounded, the quotient rounded - and the derivative p(1-p) is taken from the
    ROUNDED p (torch's sigmoid backward uses the saved output).  These roundings are part of the semantics, not noise:
    for v > ~17 the sum 1 + exp(-v) IS 1.0f, which is what caps a term at -log(1e-15) = 34.54 and zeroes its gradient
    (SURVEY.md section 8 A6), and between v ~ 9 and 17 the complement 1-p is a small multiple of 2^-24, so a term
    -log(1-p+EPS) moves by percents with the rounding of p.  Everything else (dot products, logs, means, the backward
    sums) stays exact float64."""

    @staticmethod
    def forward(ctx, v):
        e = torch.exp(-v).float()                 # correctly rounded expf
        p = (1.0 / (1.0 + e)).double()            # fp32 add, fp32 divide (both correctly rounded by IEEE)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return g * p * (1 - p)


def decode(z, edge_index, sigmoid=True):
    v = (z[edge_index[0]] * z[edge_index[1]]).sum(dim=1)
    if not sigmoid:
        return v
    return _SigmoidOnF32Grid.apply(v) if v.dtype == torch.float64 else torch.sigmoid(v)


def recon_loss(z, pos_edge_index, neg_edge_index):
    pos = -torch.log(decode(z, pos_edge_index) + EPS).mean()
    neg = -torch.log(1 - decode(z, neg_edge_index) + EPS).mean()
    return pos + neg


def sigma_negatives_np(rowptr, col, N, a, b, mod=None):
    """Permutation-structured negatives (ceeg_decoder_fused_fwd): for CSR (no self-loops) entry p in row r:
    pair (col[p], sigma(r)), sigma(x) = (x - x % m) + (a (x % m) + b) % m with m = mod (default N: one permutation
    of all nodes; m = nodes per graph: inside each graph of a collated batch); invalid when equal or when
    col[p]->sigma(r) is an edge."""
    nnz = int(rowptr[-1])
    m = int(N if mod is None else mod)
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    sig = rows - rows % m + (a * (rows % m) + b) % m
    s = col[:nnz].astype(np.int64)
    edges = set(zip(s.tolist(), rows.tolist()))
    valid = np.array([(si != di) and ((si, di) not in edges) for si, di in zip(s.tolist(), sig.tolist())], dtype=bool)
    return s, sig, valid


class RefEncoder(nn.Module):
    """2-layer variational GCN encoder (PyG example convention): conv1 F_in -> 2D, conv_mu / conv_logstd 2D -> D."""

    def __init__(self, f_in, latent):
        super().__init__()
        self.conv1 = RefGCNConv(f_in, 2 * latent)
        self.conv_mu = RefGCNConv(2 * latent, latent)
        self.conv_logstd = RefGCNConv(2 * latent, latent)

    def forward(self, x, edge_index):
        h = self.conv1(x, edge_index).relu()
        return self.conv_mu(h, edge_index), self.conv_logstd(h, edge_index)


class RefVGAEVQ(nn.Module):
    """GCN-VGAE + VQ bottleneck: loss = recon_loss(z_q) + kl/N + embedding_loss (SURVEY.md §3.5;
    loss summation as scripts/run.py:95, VQ placement as scripts/models.py:32-34)."""

    def __init__(self, f_in, latent, n_e=512, beta=0.25):
        super().__init__()
        self.encoder = RefEncoder(f_in, latent)
        self.vector_quantization = RefVectorQuantizer(n_e, latent, beta)

    def forward(self, x, edge_index, eps, neg_edge_index, neg_valid=None, batch=None, condition=None, force_idx=None):
        """batch / condition: config 4's conditional form - the per-graph condition c[batch] is concatenated to the
        encoder input (scripts/models.py:162) and to the latent before decoding (scripts/models.py:259).
        force_idx: see RefVectorQuantizer.forward.  Runs in the dtype of its parameters and inputs (call .double() on the
        module and the inputs for the fp64-accumulated form the GPU tests compare against)."""
        cn = None if condition is None else condition[batch]
        if cn is not None:
            x = torch.cat((x, cn), dim=1)
        mu, logstd = self.encoder(x, edge_index)
        logstd = logstd.clamp(max=MAX_LOGSTD)
        z = reparametrize(mu, logstd, eps)
        kl = kl_loss(mu, logstd)
        vq_loss, z_q, ppl, _, idx = self.vector_quantization(z, force_idx)
        if cn is not None:
            z_q = torch.cat((z_q, cn), dim=1)
        pos = -torch.log(decode(z_q, edge_index) + EPS).mean()
        nterm = -torch.log(1 - decode(z_q, neg_edge_index) + EPS)
        if neg_valid is not None:
            nterm = nterm[neg_valid]
        rec = pos + nterm.mean()
        loss = rec + kl / x.shape[0] + vq_loss
        return loss, dict(rec=rec, kl=kl, vq=vq_loss, ppl=ppl, idx=idx, z=z, mu=mu, logstd=logstd)


class RefGATConv(nn.Module):
    """Upstream PyG GATConv (heads, concat, negative_slope 0.2, add_self_loops, bias; dropout 0), written out:
    x' = lin_src(x) -> [N,H,C]; a_s = <x', att_src>, a_d = <x', att_dst>; self-loops of the input removed and one per
    node added; alpha = softmax over incoming edges of leaky_relu(a_s[src] + a_d[dst]) with PyG's
    exp(e - max) / (sum + 1e-16); out = sum alpha * x'[src] at dst; concat (or mean over heads) + bias."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2):
        super().__init__()
        self.heads, self.out_channels, self.concat, self.negative_slope = heads, out_channels, concat, negative_slope
        self.lin_src = _Lin(in_channels, heads * out_channels)
        self.att_src = nn.Parameter(torch.randn(1, heads, out_channels) * 0.3)
        self.att_dst = nn.Parameter(torch.randn(1, heads, out_channels) * 0.3)
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))

    def forward(self, x, edge_index):
        N, H, C = x.shape[0], self.heads, self.out_channels
        xp = (x @ self.lin_src.weight.t()).view(N, H, C)
        a_s, a_d = (xp * self.att_src).sum(-1), (xp * self.att_dst).sum(-1)
        keep = edge_index[0] != edge_index[1]
        ar = torch.arange(N)
        src, dst = torch.cat([edge_index[0][keep], ar]), torch.cat([edge_index[1][keep], ar])
        e = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], self.negative_slope)            # [E, H]
        m = torch.full((N, H), -float("inf"), dtype=x.dtype).scatter_reduce(0, dst[:, None].expand(-1, H), e, "amax", include_self=True)
        ex = (e - m[dst]).exp()
        den = torch.zeros(N, H, dtype=x.dtype).index_add_(0, dst, ex)
        alpha = ex / (den[dst] + 1e-16)
        out = torch.zeros(N, H, C, dtype=x.dtype).index_add_(0, dst, alpha[:, :, None] * xp[src])
        out = out.reshape(N, H * C) if self.concat else out.mean(1)
        return out + self.bias


def ref_global_attention(x, gate, batch, num_graphs):
    """PyG GlobalAttention: softmax of the gate inside each graph (exp(g - max) / (sum + 1e-16)), weighted node sum."""
    m = torch.full((num_graphs,), -float("inf"), dtype=x.dtype).scatter_reduce(0, batch, gate, "amax", include_self=True)
    ex = (gate - m[batch]).exp()
    den = torch.zeros(num_graphs, dtype=x.dtype).index_add_(0, batch, ex)
    w = ex / (den[batch] + 1e-16)
    return torch.zeros(num_graphs, x.shape[1], dtype=x.dtype).index_add_(0, batch, w[:, None] * x)


def ref_graph_layer_norm(x, batch, num_graphs, weight=None, bias=None, eps=1e-5):
    """PyG LayerNorm(mode='graph'): per-graph mean / biased variance over all nodes and channels."""
    cnt = torch.bincount(batch, minlength=num_graphs).clamp(min=1).to(x.dtype) * x.shape[1]
    mean = torch.zeros(num_graphs, dtype=x.dtype).index_add_(0, batch, x.sum(1)) / cnt
    xc = x - mean[batch][:, None]
    var = torch.zeros(num_graphs, dtype=x.dtype).index_add_(0, batch, (xc * xc).sum(1)) / cnt
    out = xc / (var + eps).sqrt()[batch][:, None]
    return out if weight is None else out * weight + bias


class RefGraphConditioner(nn.Module):
    """Restatement of the reference's GraphConditioner (scripts/models.py:95-122) on the written-out PyG formulas above:
    token attention pooling bmm(hidden^T, attn(hidden)) (:118-119), then per layer GATConv -> LayerNorm(mode='graph') ->
    ReLU (:105-110), then GlobalAttention (:112-113,122).  Parameter names follow the reference's state dict
    (torch_geometric.nn.Sequential numbers every module, the parameter-free ReLUs included).  Parity unpinned
    (torch_geometric is not installed; the token encoder is a plug-in and out of scope)."""

    def __init__(self, embedding, embedding_hidden_size, graph_layer_sizes):
        super().__init__()
        self.embedding = embedding
        self.attn = nn.Linear(embedding_hidden_size, 1)
        seq = nn.Module()
        self.n_layers = len(graph_layer_sizes)
        sizes = [embedding_hidden_size] + list(graph_layer_sizes)
        for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
            seq.add_module(f"module_{3 * i}", RefGATConv(a, b))
            ln = nn.Module()
            ln.weight, ln.bias = nn.Parameter(torch.ones(b)), nn.Parameter(torch.zeros(b))
            seq.add_module(f"module_{3 * i + 1}", ln)
        self.graph_sequential_model = seq
        pool = nn.Module()
        pool.gate_nn = nn.Linear(graph_layer_sizes[-1], 1)
        self.pooling_layer = pool

    def forward(self, input):
        hidden = self.embedding(input_ids=input.input_ids, attention_mask=input.attention_mask).last_hidden_state
        x = torch.bmm(hidden.permute(0, 2, 1), self.attn(hidden)).squeeze(2)
        batch, G = input.input_ids_batch, input.num_graphs
        for i in range(self.n_layers):
            conv = getattr(self.graph_sequential_model, f"module_{3 * i}")
            ln = getattr(self.graph_sequential_model, f"module_{3 * i + 1}")
            x = torch.relu(ref_graph_layer_norm(conv(x, input.edge_index), batch, G, ln.weight, ln.bias))
        return ref_global_attention(x, self.pooling_layer.gate_nn(x).view(-1), batch, G)
