"""Host-side model API of the VGAE + VQ training step on MI355X.

Mirrors the surface the north star names: PyG-style `GCNConv`, `VGAE` with
`encode / reparametrize / kl_loss / recon_loss`, `InnerProductDecoder`, `negative_sampling`, and the
reference's `VectorQuantizer` (scripts/quantizer.py) placed after the encoder exactly where
`BaseCVAE.forward` puts it (scripts/models.py:30-34).  torch_geometric is NOT a dependency: the
semantics are restated from SURVEY.md §8(a) rows A3-A6 ("parity unpinned": the reference neither
vendors nor pins PyG, environment.yml:13).

Everything numeric goes through libceeg.so (include/ceeg.h); PyTorch supplies device memory, the
stream, autograd bookkeeping and the dense feature transforms x @ W^T (hipBLASLt via torch.mm —
"plain library GEMMs").  There is no CPU fallback.
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib
from .quantizer import VectorQuantizer

MAX_LOGSTD = 10  # /root/reference/scripts/run.py:13 (== upstream PyG autoencoder.MAX_LOGSTD)
EPS = 1e-15


# ==================================================================================================
# Graph structure: canonical CSR (by dst, then src, then edge id) + its transpose, built once per graph
# ==================================================================================================
class CsrPair:
    """A CSR grouped by destination and its transpose (grouped by source) with the value permutation.

    Everything is built by kernels launched back to back - canonical CSR, transpose, the longest row of each orientation and
    the piece lists of the long rows (ceeg_csr_long_items) - and what the host has to know (entry count, longest rows, piece
    counts) arrives in ONE small device record read once by `finalize()`.  `defer=True` leaves that read to the caller, so that
    several structures (EventGraph's two pairs, a collation's layout check) share a single host synchronisation."""

    def __init__(self, edge_index, num_nodes, self_loops, defer=False, raw=None, blocks=None):
        """blocks = (batch [N] int64, node_ptr [G+1] int64, largest graph): the edge list is a collated batch (every edge inside its
        graph's block of node ids) - the raw-edge CSR is then sorted on 32-bit keys that hold the source relative to its graph
        (ceeg_csr_build_blocks).  `blocks_ok` (0-d bool tensor on the device, None without blocks) says whether the promise held; if
        not, the structure lacks the offending edges and must be rebuilt without `blocks`."""
        _lib.require_cuda(edge_index)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        L = _lib.lib()
        dev = edge_index.device
        E, N = edge_index.shape[1], int(num_nodes)
        cap = max(E + (N if self_loops else 0), 1)
        self.num_nodes, self.self_loops = N, bool(self_loops)
        self._cap = cap
        self.blocks_ok = None
        self.rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.col = torch.empty(cap, dtype=torch.int32, device=dev)
        self.perm = torch.empty(cap, dtype=torch.int32, device=dev)
        self.t_rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.t_col = torch.empty(cap, dtype=torch.int32, device=dev)
        self.t_perm = torch.empty(cap, dtype=torch.int32, device=dev)
        if raw is not None:
            # `raw` = the not yet finalised raw-edge pair of the SAME edge list: A + I is that structure with every row's run of
            # self-loops replaced by one loop entry (ceeg_csr_add_self_loops) - no second sort, no second transpose
            if not self_loops or raw.self_loops or raw.num_nodes != N or raw._items is None or raw._cap != max(E, 1):
                raise ValueError("raw must be the unfinalised raw-edge CsrPair (self_loops=False) of the same edge list")
            ws = _lib.workspace(L.ceeg_csr_add_self_loops_ws_bytes(N), dev)
            _lib.check(L.ceeg_csr_add_self_loops(raw.rowptr.data_ptr(), raw.col.data_ptr(), raw.perm.data_ptr(), raw.t_rowptr.data_ptr(),
                                                 raw.t_col.data_ptr(), raw.t_perm.data_ptr(), N, self.rowptr.data_ptr(), self.col.data_ptr(),
                                                 self.perm.data_ptr(), self.t_rowptr.data_ptr(), self.t_col.data_ptr(),
                                                 self.t_perm.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_csr_add_self_loops")
        else:
            ei = edge_index.to(torch.int64).contiguous()
            if (blocks is not None and not self_loops and E > 0 and os.environ.get("CEEG_CSR_BLOCKS", "1") == "1"
                    and L.ceeg_csr_build_blocks_supported(N, int(blocks[2])) == 1):
                b_batch, b_ptr, b_max = blocks
                flag = torch.empty(1, dtype=torch.int32, device=dev)
                ws = _lib.workspace(L.ceeg_csr_build_blocks_ws_bytes(E, N, int(b_max)), dev)
                _lib.check(L.ceeg_csr_build_blocks(ei[0].data_ptr(), ei[1].data_ptr(), E, N, b_batch.data_ptr(), b_ptr.data_ptr(),
                                                   b_ptr.shape[0] - 1, int(b_max), self.rowptr.data_ptr(), self.col.data_ptr(),
                                                   self.perm.data_ptr(), None, flag.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()),
                           "ceeg_csr_build_blocks")
                self.blocks_ok = flag[0] == 0
            else:
                ws = _lib.workspace(L.ceeg_csr_build_ws_bytes(E, N, int(self_loops)), dev)
                _lib.check(L.ceeg_csr_build(ei[0].data_ptr(), ei[1].data_ptr(), E, N, int(self_loops), self.rowptr.data_ptr(),
                                            self.col.data_ptr(), self.perm.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                            _lib.stream()), "ceeg_csr_build")
            # the transpose runs on the CAPACITY: the true entry count (input self-loops dropped, out-of-range ids skipped) stays
            # on the device until finalize()
            ws = _lib.workspace(L.ceeg_csr_transpose_ws_bytes(cap, N), dev)
            _lib.check(L.ceeg_csr_transpose(self.rowptr.data_ptr(), self.col.data_ptr(), cap, N, self.t_rowptr.data_ptr(),
                                            self.t_col.data_ptr(), self.t_perm.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _lib.stream()), "ceeg_csr_transpose")
        # row statistics + piece lists of both orientations -> one record [nnz, max_row, n_long, nnz, max_trow, n_tlong]
        chunk = int(L.ceeg_decoder_codes_chunk())
        icap = 3 * cap // chunk + 4
        self._items = [torch.empty(icap, 4, dtype=torch.int32, device=dev) for _ in range(2)]
        self._rec = torch.empty(6, dtype=torch.int64, device=dev)
        ws = _lib.workspace(L.ceeg_csr_long_items_ws_bytes(N), dev)
        for k, rp in enumerate((self.rowptr, self.t_rowptr)):
            _lib.check(L.ceeg_csr_long_items(rp.data_ptr(), N, chunk, self._items[k].data_ptr(), icap, self._rec[3 * k:].data_ptr(),
                                             ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_csr_long_items")
        self._long = None
        if not defer:
            self.finalize()

    def finalize(self, record=None):
        """Take the host copy of the record (read here - the ONE host synchronisation of the construction - unless the caller
        read it together with other records and passes it in) and cut the arrays to the true entry count."""
        if self._long is not None:
            return self
        r = (self._rec.cpu() if record is None else record).tolist()
        icap = self._items[0].shape[0]
        if r[2] > icap or r[5] > icap:
            raise _lib.CeegError("piece list capacity exceeded (internal bound 3 * nnz / chunk)")
        self.nnz, self.max_row, self.max_trow = int(r[0]), int(r[1]), int(r[4])
        self.col, self.perm = self.col[: self.nnz], self.perm[: self.nnz]
        self.t_col, self.t_perm = self.t_col[: self.nnz], self.t_perm[: self.nnz]
        self._long = {False: self._items[0][: r[2]] if r[2] else None, True: self._items[1][: r[5]] if r[5] else None}
        self._items = None
        return self

    @property
    def device(self):
        return self.rowptr.device

    def long_items(self, transposed=False):
        """Piece list of the rows longer than ceeg_decoder_codes_chunk() entries (include/ceeg.h, ceeg_decoder_codes_* and
        ceeg_spmm_csr_items); None when no row is that long.  Static per graph, filled on the device at construction."""
        return self.finalize()._long[bool(transposed)]


def build_long_items(rowptr, chunk):
    """int32 [n][4] = (lo, hi, chunk_index*2 + slot, row): every row longer than `chunk` entries cut at the GLOBAL positions
    k*chunk, rows ascending, pieces ascending; slot = 1 for the piece that starts its row in the middle of a chunk (a chunk
    can hold the tail of one long row and the head of the next), else 0.  None when no row is that long.
    Host-side (torch ops, synchronising) statement of what ceeg_csr_long_items fills on the device; kept as its checker and for
    row-pointer arrays that do not come from a CsrPair."""
    rp = rowptr.long()
    rows = torch.nonzero((rp[1:] - rp[:-1]) > chunk).view(-1)
    if rows.numel() == 0:
        return None
    lo_r, hi_r = rp[rows], rp[rows + 1]
    c_lo, c_hi = lo_r // chunk, (hi_r - 1) // chunk
    cnt = c_hi - c_lo + 1
    rid = torch.repeat_interleave(torch.arange(rows.numel(), device=rp.device), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    c = c_lo[rid] + (torch.arange(rid.numel(), device=rp.device) - first[rid])
    lo = torch.maximum(lo_r[rid], c * chunk)
    hi = torch.minimum(hi_r[rid], (c + 1) * chunk)
    slot = ((c == c_lo[rid]) & (lo_r[rid] != c * chunk)).long()
    return torch.stack([lo, hi, c * 2 + slot, rows[rid]], dim=1).to(torch.int32).contiguous()


class EventGraph:
    """An event graph (edges old_id -> new_id, /root/reference/scripts/dataset.py:147-153) prepared for the
    GCN encoder (`gcn`: A + I with symmetric normalisation) and the decoder (`dec`: the raw edges).
    Build it once per graph / minibatch; every step reuses the structures.

    with_dec=True builds the decoder's pair in the same pass, and `checks` (name -> 0-d bool tensor on the device, e.g. a
    collation's "no edge leaves its graph") is read with the same record: the whole construction then costs ONE host
    synchronisation - what a loop that sees a new minibatch every step (scripts/run.py:310-326) wants.  Without them the decoder
    pair is built on first use (a second synchronisation)."""

    def __init__(self, edge_index, num_nodes, edge_weight=None, with_dec=False, checks=None, defer=False, blocks=None):
        """blocks: see CsrPair - a collation's promise that every edge stays inside its graph; checked on the device, and a broken
        promise only costs a second construction in finish()."""
        self.edge_index = edge_index
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.shape[1])
        self._dec = CsrPair(edge_index, num_nodes, self_loops=False, defer=True, blocks=blocks) if with_dec else None
        if self._dec is not None and self._dec.blocks_ok is not None:
            checks = dict(checks or {})
            checks[("csr_blocks",)] = self._dec.blocks_ok
        derive = with_dec and os.environ.get("CEEG_GCN_FROM_RAW", "1") == "1"
        self.gcn = CsrPair(edge_index, num_nodes, self_loops=True, defer=True, raw=self._dec if derive else None)
        names = list(checks or {})
        parts = [self.gcn._rec] + ([self._dec._rec] if with_dec else [])
        if names:
            parts.append(torch.stack([checks[k].reshape(()).to(torch.int64) for k in names]))
        # the record starts its way to the host now (pinned buffer, asynchronous copy); finish() waits for it
        rec_dev = torch.cat(parts)
        rec_host = torch.empty(rec_dev.shape, dtype=torch.int64, pin_memory=True)
        rec_host.copy_(rec_dev, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record()
        self._pending = (rec_host, landed, rec_dev, names, edge_weight, bool(with_dec))
        if not defer:
            self.finish()

    def finish(self):
        """Second half of the construction: wait for the record (the ONE host synchronisation), cut the arrays to the true entry
        counts, gcn_norm.  `defer=True` leaves the call to a loader that has other work to queue first (collate.MinibatchStream);
        it must run on the stream the construction was started on."""
        if self._pending is None:
            return self
        rec_host, landed, _rec_dev, names, edge_weight, with_dec = self._pending
        self._pending = None
        landed.synchronize()
        rec = rec_host
        self.gcn.finalize(rec[:6])
        if with_dec:
            self._dec.finalize(rec[6:12])
        self.checks = {k: bool(v) for k, v in zip(names, rec[len(rec) - len(names):].tolist())} if names else {}
        if not self.checks.pop(("csr_blocks",), True):
            # an edge left its graph's block: the 32-bit keyed structure lacks it - build again on the general keys (synchronously)
            other = EventGraph(self.edge_index, self.num_nodes, edge_weight, with_dec=True)
            other.checks.update(self.checks)
            self.__dict__.update(other.__dict__)
            return self
        L = _lib.lib()
        dev = self.edge_index.device
        ew = None if edge_weight is None else edge_weight.to(torch.float32).contiguous()
        self.dinv = torch.empty(self.num_nodes, dtype=torch.float32, device=dev)
        self.gcn_val = torch.empty(max(self.gcn.nnz, 1), dtype=torch.float32, device=dev)[: self.gcn.nnz]
        # values also in the transposed order, gathered once (same call): the backward SpMM then streams them instead of chasing
        # val[t_perm[q]] (a dependent load in front of every row's gathers)
        self.gcn_tval = torch.empty(max(self.gcn.nnz, 1), dtype=torch.float32, device=dev)[: self.gcn.nnz]
        _lib.check(L.ceeg_gcn_norm_t(self.gcn.rowptr.data_ptr(), self.gcn.col.data_ptr(), self.gcn.perm.data_ptr(), _lib.ptr(ew),
                                     self.gcn.nnz, self.num_nodes, self.dinv.data_ptr(), self.gcn_val.data_ptr(), self.gcn.t_rowptr.data_ptr(),
                                     self.gcn.t_perm.data_ptr(), self.gcn_tval.data_ptr(), _lib.stream()), "ceeg_gcn_norm_t")
        self.reuse, self._gcn_cv, self._gcn_tcv = True, None, None
        return self

    def __getattr__(self, name):   # only reached for attributes that are not there
        if name in ("gcn_val", "gcn_tval", "dinv", "checks", "reuse") and self.__dict__.get("_pending") is not None:
            raise _lib.CeegError(f"EventGraph.{name}: the graph was built with defer=True and finish() has not been called")
        raise AttributeError(name)

    def cv(self, transposed=False):
        """(column, value) pairs of the normalised adjacency (or its transpose) interleaved into one int32 [nnz, 2] stream."""
        if transposed:
            if self._gcn_tcv is None:
                self._gcn_tcv = torch.stack([self.gcn.t_col, self.gcn_tval.view(torch.int32)], dim=1).contiguous()
            return self._gcn_tcv
        if self._gcn_cv is None:
            self._gcn_cv = torch.stack([self.gcn.col, self.gcn_val.view(torch.int32)], dim=1).contiguous()
        return self._gcn_cv

    def aggregate16(self, h, transposed=False, mask=None):
        """A_hat @ h (transposed: A_hat^T @ h, optionally times (mask > 0)) with bf16 storage and fp32 sums."""
        g = self.gcn
        rp, mr, items = (g.t_rowptr, g.max_trow, g.long_items(True)) if transposed else (g.rowptr, g.max_row, g.long_items(False))
        return _spmm_cv(rp, self.cv(transposed), h.contiguous(), self.num_nodes, g.nnz, mask=mask, max_row=mr, items=items)

    def aggregate(self, h, bias=None, relu=False):
        """act(A_hat @ h + b).  A graph that is reused across steps (`reuse`, the default; a minibatch-stream graph is not)
        keeps its (column, value) pairs interleaved in one stream (built on first use): 7 % off the F = 128 launch."""
        g = self.gcn
        if self.reuse and g.nnz and os.environ.get("CEEG_SPMM_CV", "1") == "1":
            return _spmm_cv(g.rowptr, self.cv(False), h, self.num_nodes, g.nnz, bias=bias, relu=relu, max_row=g.max_row, items=g.long_items(False))
        return _spmm(g.rowptr, g.col, self.gcn_val, None, h, self.num_nodes, bias=bias, relu=relu, max_row=g.max_row, items=g.long_items(False))

    def edge_index_i32(self):
        """int32 copy of edge_index for the edge-list kernels (made once per graph)."""
        if getattr(self, "_ei32", None) is None:
            self._ei32 = self.edge_index.to(torch.int32).contiguous()
        return self._ei32

    @property
    def dec(self):
        if self._dec is None:
            self._dec = CsrPair(self.edge_index, self.num_nodes, self_loops=False)
        return self._dec


def _as_graph(g, num_nodes):
    if isinstance(g, EventGraph):
        return g
    return EventGraph(g, num_nodes)


def _spmm_cv(rowptr, colval, x, n_rows, nnz, bias=None, relu=False, mask=None, max_row=0, items=None):
    """ceeg_spmm_csr_cv: the aggregation over a CSR whose (column, value) pairs are interleaved into one 8-byte stream."""
    L = _lib.lib()
    F = x.shape[1]
    out = torch.empty(n_rows, F, dtype=x.dtype, device=x.device)
    ws = _lib.workspace(L.ceeg_spmm_ws_bytes(n_rows, nnz, F), x.device)
    if x.dtype == torch.bfloat16:   # bf16 storage of the gathered matrix, the mask and the result; fp32 sums
        if mask is not None and mask.dtype != torch.bfloat16:
            raise ValueError("the bf16 aggregation takes a bf16 mask")
        with _lib.timed(f"spmm16_masked_F{F}" if mask is not None else f"spmm16_F{F}"):
            _lib.check(L.ceeg_spmm_csr_cv_bf16(rowptr.data_ptr(), colval.data_ptr(), x.data_ptr(), x.stride(0), n_rows, nnz, F, _lib.ptr(bias), int(relu),
                                               _lib.ptr(mask), 0 if mask is None else mask.stride(0), out.data_ptr(), out.stride(0), int(max_row),
                                               _lib.ptr(items), 0 if items is None else items.shape[0], ws.data_ptr(), ws.numel(), _lib.stream()),
                       "ceeg_spmm_csr_cv_bf16")
        return out
    with _lib.timed(f"spmm_masked_F{F}" if mask is not None else f"spmm_F{F}"):
        _lib.check(L.ceeg_spmm_csr_cv(rowptr.data_ptr(), colval.data_ptr(), x.data_ptr(), x.stride(0), n_rows, nnz, F, _lib.ptr(bias), int(relu),
                                      _lib.ptr(mask), 0 if mask is None else mask.stride(0), out.data_ptr(), out.stride(0), int(max_row),
                                      _lib.ptr(items), 0 if items is None else items.shape[0], ws.data_ptr(), ws.numel(), _lib.stream()),
                   "ceeg_spmm_csr_cv")
    return out


def _spmm(rowptr, col, val, vperm, x, n_rows, bias=None, relu=False, out=None, accumulate=False, max_row=-1, items=False, xidx=None, mask=None):
    """items: False = let the kernel find the long rows; None / int32 [n,4] = the graph's piece list (CsrPair.long_items).
    xidx (int32 [n_rows]): x is a TABLE and node c's row is x[xidx[c]] (ceeg_spmm_csr_table; no bias / relu there).
    mask ([n_rows, F] fp32, with a piece list): the finished rows are multiplied by (mask > 0) (ceeg_spmm_csr_masked)."""
    L = _lib.lib()
    nnz = col.shape[0]
    F = x.shape[1]
    if out is None:
        out = torch.empty(n_rows, F, dtype=torch.float32, device=x.device)
    ws = _lib.workspace(L.ceeg_spmm_ws_bytes(n_rows, nnz, F), x.device)
    if mask is not None:
        if items is False or max_row < 0 or bias is not None or relu or accumulate or xidx is not None:
            raise ValueError("the masked SpMM takes the graph's piece list and no bias / relu / accumulate / table")
        with _lib.timed(f"spmm_masked_F{F}"):
            _lib.check(L.ceeg_spmm_csr_masked(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), _lib.ptr(vperm), x.data_ptr(), x.stride(0),
                                              n_rows, nnz, F, mask.data_ptr(), mask.stride(0), out.data_ptr(), out.stride(0), int(max_row),
                                              _lib.ptr(items), 0 if items is None else items.shape[0], ws.data_ptr(), ws.numel(),
                                              _lib.stream()), "ceeg_spmm_csr_masked")
        return out
    if xidx is not None:
        with _lib.timed(f"spmm_table_F{F}"):
            _lib.check(L.ceeg_spmm_csr_table(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), _lib.ptr(vperm), x.data_ptr(), x.stride(0),
                                             xidx.data_ptr(), n_rows, nnz, F, int(accumulate), out.data_ptr(), out.stride(0), int(max_row),
                                             ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_spmm_csr_table")
        return out
    if items is not False and max_row >= 0:
        with _lib.timed(f"spmm_F{F}"):
            _lib.check(L.ceeg_spmm_csr_items(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), _lib.ptr(vperm), x.data_ptr(), x.stride(0),
                                             n_rows, nnz, F, _lib.ptr(bias), int(relu), int(accumulate), out.data_ptr(), out.stride(0),
                                             int(max_row), _lib.ptr(items), 0 if items is None else items.shape[0], ws.data_ptr(), ws.numel(),
                                             _lib.stream()), "ceeg_spmm_csr_items")
        return out
    with _lib.timed(f"spmm_F{F}"):
        _lib.check(L.ceeg_spmm_csr(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), _lib.ptr(vperm), x.data_ptr(), x.stride(0),
                                   n_rows, nnz, F, _lib.ptr(bias), int(relu), int(accumulate), out.data_ptr(), out.stride(0),
                                   int(max_row), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_spmm_csr")
    return out


def _colsum(g):
    """Deterministic column sums of a [N, F] fp32 matrix (bias gradients)."""
    L = _lib.lib()
    n, F = g.shape
    out = torch.empty(F, dtype=torch.float32, device=g.device)
    ws = _lib.workspace(L.ceeg_colsum_ws_bytes(n, F), g.device)
    with _lib.timed("colsum"):
        _lib.check(L.ceeg_colsum(g.data_ptr(), g.stride(0), n, F, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_colsum")
    return out


def _relu_bwd_colsum(g, y):
    """One pass: gm = g * (y > 0) and its column sums (autograd's incoming gradient is left untouched)."""
    L = _lib.lib()
    n, F = g.shape
    gm = torch.empty(n, F, dtype=torch.float32, device=g.device)
    out = torch.empty(F, dtype=torch.float32, device=g.device)
    ws = _lib.workspace(L.ceeg_colsum_ws_bytes(n, F), g.device)
    with _lib.timed("relu_bwd_colsum"):
        _lib.check(L.ceeg_relu_bwd_colsum(g.data_ptr(), g.stride(0), y.data_ptr(), y.stride(0), gm.data_ptr(), gm.stride(0), n, F, out.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_relu_bwd_colsum")
    return gm, out


class _GcnAggregate(torch.autograd.Function):
    """out = act(A_hat @ h + b); backward dh = A_hat^T @ g through the transposed CSR (values via t_perm)."""

    @staticmethod
    def forward(ctx, h, bias, graph, relu):
        _lib.require_cuda(h)
        h = h.contiguous()
        out = graph.aggregate(h, bias=bias, relu=relu)
        ctx.graph, ctx.relu, ctx.has_bias = graph, bool(relu), bias is not None
        if relu:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        graph, g = ctx.graph, ctx.graph.gcn
        grad = grad.contiguous()
        if ctx.relu:
            (out,) = ctx.saved_tensors
            grad, dbias = _relu_bwd_colsum(grad, out)
        else:
            dbias = _colsum(grad) if ctx.has_bias else None
        dh = _spmm(g.t_rowptr, g.t_col, graph.gcn_tval, None, grad, graph.num_nodes, max_row=g.max_trow, items=g.long_items(True)) if ctx.needs_input_grad[0] else None
        return dh, dbias, None, None


def glorot_(w):
    a = math.sqrt(6.0 / (w.shape[0] + w.shape[1]))
    with torch.no_grad():
        w.uniform_(-a, a)


def _split_k(n):
    """Largest S <= 64 dividing n (S = 1 when n is prime-ish): the K-split of the weight-gradient product."""
    for s in (64, 32, 16, 8, 4, 2):
        if n % s == 0:
            return s
    return 1


def _linear_fwd(x, weight, bias, relu):
    """y = act(x @ W^T + b) in NN form; bias + ReLU in the GEMM epilogue when both are there."""
    wt = weight.t().contiguous()  # [in, out]; out*in floats
    if relu and bias is not None and os.environ.get("CEEG_GEMM_EPILOGUE", "1") == "1":
        return torch._addmm_activation(bias, x, wt)  # hipBLASLt bias+ReLU epilogue: no extra pass over [N, out]
    y = torch.mm(x, wt) if bias is None else torch.addmm(bias, x, wt)
    if relu:
        L = _lib.lib()
        with _lib.timed("bias_relu"):
            _lib.check(L.ceeg_bias_relu(y.data_ptr(), y.stride(0), y.shape[0], y.shape[1], None, 1, _lib.stream()), "ceeg_bias_relu")
    return y


def _weight_grad(x, g):
    """dW = g^T x for a tall x [N, in] and g [N, out]: the N-deep reduction split over K into a batched product whose partial
    results are summed in a fixed order (deterministic)."""
    n = x.shape[0]
    s = _split_k(n)
    if s > 1:
        # x^T g per slice (the faster of the two orientations in hipBLASLt for these shapes), transposed at the end
        return torch.bmm(x.contiguous().view(s, n // s, -1).transpose(1, 2), g.view(s, n // s, -1)).sum(dim=0).t()
    return g.t() @ x


class _Linear(torch.autograd.Function):
    """y = x @ W^T (+ b) for a tall-skinny x [N, in] and a small W [out, in], arranged for hipBLASLt's fast
    fp32 kernels (measured on MI355X, N = 1M, 256x256): every product is issued in NN form, and the weight
    gradient g^T x (a 1M-deep reduction into a 256x256 tile: 64 workgroups on 256 CUs as one GEMM) is split
    over K into a batched product whose S partial results are summed in a fixed order (deterministic)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        y = _linear_fwd(x, weight, bias, relu)
        if relu:
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.relu = bias is not None, bool(relu)
        return y

    @staticmethod
    def backward(ctx, g):
        db = None
        if ctx.relu:
            x, weight, y = ctx.saved_tensors
            g, db = _relu_bwd_colsum(g.contiguous(), y)
        else:
            x, weight = ctx.saved_tensors
            g = g.contiguous()
        dx = torch.mm(g, weight) if ctx.needs_input_grad[0] else None  # [N,out] @ [out,in]: NN
        dw = _weight_grad(x, g) if ctx.needs_input_grad[1] else None
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(g)
        return dx, dw, (db if ctx.has_bias else None), None


class _LinearReluAggregate(torch.autograd.Function):
    """ah = A_hat @ relu(ax @ W^T + b): the first layer's dense half (aggregate-first form, ax = A_hat x) and the SECOND
    layer's aggregation as one autograd node, so that the ReLU's backward mask is applied in the epilogue of the transposed
    pass that produces dh (ceeg_spmm_csr_masked) instead of a separate read-modify-write of the [N, H] gradient."""

    @staticmethod
    def forward(ctx, ax, weight, bias, graph):
        _lib.require_cuda(ax)
        h = _linear_fwd(ax, weight, bias, True)
        ah = graph.aggregate(h)
        ctx.save_for_backward(ax, weight, h)
        ctx.graph = graph
        return ah

    @staticmethod
    def backward(ctx, dah):
        ax, weight, h = ctx.saved_tensors
        graph, g = ctx.graph, ctx.graph.gcn
        gm = _spmm(g.t_rowptr, g.t_col, graph.gcn_tval, None, dah.contiguous(), graph.num_nodes, max_row=g.max_trow, items=g.long_items(True),
                   mask=h)                                            # (A_hat^T dah) * (h > 0)
        db = _colsum(gm) if ctx.needs_input_grad[2] else None
        dw = _weight_grad(ax, gm) if ctx.needs_input_grad[1] else None
        dax = torch.mm(gm, weight) if ctx.needs_input_grad[0] else None
        return dax, dw, db, None


# ---- opt-in reduced-precision encoder (bf16 storage and bf16 MFMA products, fp32 accumulation everywhere) -------------------------
_F32_OUT = {}


def _mm16(a, b, bias=None):
    """fp32 result of a bf16 product (+ fp32 bias): the library's fp32-output form where this PyTorch build has it (no rounding
    of the accumulator), else the bf16 product widened."""
    mode = _F32_OUT.get("mm")
    if mode is None:
        try:
            torch.mm(a[:1], b, out_dtype=torch.float32)
            mode = True
        except Exception:   # noqa: BLE001 - an older / different build: fall back
            mode = False
        _F32_OUT["mm"] = mode
    y = torch.mm(a, b, out_dtype=torch.float32) if mode else torch.mm(a, b).float()
    return y if bias is None else y.add_(bias)


def _weight_grad16(x, g):
    """dW = g^T x (fp32 [out, in]) for bf16 x [N, in], g [N, out]: split over K as _weight_grad, the per-slice products leave the
    matrix cores in fp32 where the build allows, and are summed in fp32 in a fixed order."""
    n = x.shape[0]
    s = _split_k(n)
    xs, gs = x.contiguous().view(s, n // s, -1).transpose(1, 2), g.contiguous().view(s, n // s, -1)
    mode = _F32_OUT.get("bmm")
    if mode is None:
        try:
            torch.bmm(xs[:1, :, :1], gs[:1, :1], out_dtype=torch.float32)
            mode = True
        except Exception:   # noqa: BLE001
            mode = False
        _F32_OUT["bmm"] = mode
    part = torch.bmm(xs, gs, out_dtype=torch.float32) if mode else torch.bmm(xs, gs).float()
    return part.sum(dim=0).t()


class _LinearReluAggregate16(torch.autograd.Function):
    """_LinearReluAggregate on bf16 activations: h = relu(ax W^T + b) from the bf16 MFMA GEMM (bias + ReLU epilogue), ah = A_hat h
    gathered as 512-byte rows; backward: gm = (A_hat^T dah) * (h > 0) in one pass, db = column sums (fp32), dW = gm^T ax."""

    @staticmethod
    def forward(ctx, ax, weight, bias, graph):
        w16 = weight.detach().to(torch.bfloat16)
        h = torch._addmm_activation(bias.detach().to(torch.bfloat16), ax, w16.t().contiguous())
        ah = graph.aggregate16(h)
        ctx.save_for_backward(ax, h)
        ctx.graph = graph
        return ah

    @staticmethod
    def backward(ctx, dah):
        ax, h = ctx.saved_tensors
        gm = ctx.graph.aggregate16(dah.contiguous(), transposed=True, mask=h)
        db = torch.sum(gm, dim=0, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        dw = _weight_grad16(ax, gm) if ctx.needs_input_grad[1] else None
        return None, dw, db, None


class _HeadReparamVQ16(torch.autograd.Function):
    """_HeadReparamVQ with the head product on bf16 operands: ml = ah W^T + b leaves the GEMM in fp32 (mu | logstd, the noise and
    the quantiser stay fp32 - the codes decide the decoder's gathers); backward: dml (fp32, from the fused kernel) is rounded to
    bf16 once for the two products dah = dml W and dW = dml^T ah; the bias gradient is the fused kernel's fp32 column sums."""

    @staticmethod
    def forward(ctx, ah, wcat, bcat, eps, seed, training, log_var, weight, beta):
        _lib.require_cuda(ah, weight)
        w16 = wcat.detach().to(torch.bfloat16)
        ml = _mm16(ah, w16.t().contiguous(), bcat.detach())
        z, kl, scal, z_q, idx, epsc, wc = _rvq_forward(ml, eps, seed, training, log_var, weight, beta)
        ctx.save_for_backward(ah, w16, ml, epsc, wc, idx, z)
        ctx.cfg = (int(seed), int(training), int(log_var), float(beta))
        loss_t, ppl_t = scal[0], scal[1]
        ctx.mark_non_differentiable(ppl_t, idx)
        ctx.set_materialize_grads(False)
        return ml, z, kl, loss_t, z_q, ppl_t, idx

    @staticmethod
    def backward(ctx, g_ml, g_z, g_kl, g_loss, g_zq, _g_ppl, _g_idx):
        ah, w16, ml, epsc, wc, idx, z = ctx.saved_tensors
        need = ctx.needs_input_grad
        dml, colsum, dW = _rvq_backward(ml, epsc, wc, idx, z, ctx.cfg, g_z, g_kl, g_loss, g_zq, need[7], need[0] or need[1] or need[2],
                                        need[2] and g_ml is None)
        dah = dwcat = dbcat = None
        if dml is not None:
            if g_ml is not None:
                dml = dml + g_ml
                colsum = _colsum(dml) if need[2] else None
            dbcat = colsum
            d16 = dml.to(torch.bfloat16)
            dwcat = _weight_grad16(ah, d16) if need[1] else None
            dah = torch.mm(d16, w16) if need[0] else None
        return dah, dwcat, dbcat, None, None, None, None, dW, None


class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        glorot_(self.weight)


class GCNConv(nn.Module):
    """PyG-style GCNConv(in_channels, out_channels): out = D^-1/2 (A+I) D^-1/2 (x W^T) + b.
    Linear first (no bias inside, glorot), then the deterministic CSR SpMM, then + b (zeros init).
    State-dict keys `lin.weight`, `bias` as upstream."""

    def __init__(self, in_channels, out_channels, aggregate_first=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregate_first = bool(aggregate_first)
        self.lin = _Lin(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, graph, edge_weight=None, relu=False):
        """aggregate_first=True evaluates (A_hat x) W^T + b instead of A_hat (x W^T) + b: the same product,
        re-associated so the gather runs at width in_channels (< out_channels) and needs no transposed pass
        when x carries no gradient.  Default False keeps upstream's linear-first order."""
        if not isinstance(graph, EventGraph):
            graph = EventGraph(graph, x.shape[0], edge_weight)
        if self.aggregate_first:
            ax = _GcnAggregate.apply(x, None, graph, False)
            return _Linear.apply(ax, self.lin.weight, self.bias, bool(relu))
        return _GcnAggregate.apply(_Linear.apply(x, self.lin.weight, None, False), self.bias, graph, relu)


# ==================================================================================================
# Reparametrisation + KL (fused), decoder
# ==================================================================================================
class _ReparamKL(torch.autograd.Function):
    """(z, kl) = fused reparametrise + KL.  `logstd is None` means `mu` is the fused [N, 2D] encoder output
    (mu | logstd), read in place through row strides; its gradient then comes back as ONE [N, 2D] tensor."""

    @staticmethod
    def forward(ctx, mu, logstd, eps, seed, training, log_var):
        _lib.require_cuda(mu, logstd)
        L = _lib.lib()
        fused = logstd is None
        if fused:
            ml = mu.contiguous()
            N, D = ml.shape[0], ml.shape[1] // 2
            mu_v, ls_v = ml[:, :D], ml[:, D:]
        else:
            N, D = mu.shape
            mu_v = mu if mu.stride(1) == 1 else mu.contiguous()
            ls_v = logstd if logstd.stride(1) == 1 else logstd.contiguous()
        epsc = None if eps is None else eps.contiguous()
        z = torch.empty(N, D, dtype=torch.float32, device=mu.device)
        kl = torch.empty((), dtype=torch.float32, device=mu.device)
        ws = _lib.workspace(L.ceeg_reparam_kl_ws_bytes(N, D), mu.device)
        _lib.check(L.ceeg_reparam_kl_fwd(mu_v.data_ptr(), mu_v.stride(0), ls_v.data_ptr(), ls_v.stride(0), _lib.ptr(epsc), int(seed),
                                         N, D, float(MAX_LOGSTD), int(log_var), int(training), z.data_ptr(), kl.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_reparam_kl_fwd")
        ctx.save_for_backward(mu_v, ls_v, epsc)
        ctx.seed, ctx.training, ctx.log_var, ctx.fused = int(seed), int(training), int(log_var), fused
        ctx.set_materialize_grads(False)  # an unused output's gradient arrives as None (the kernel takes NULL), not as zeros
        return z, kl

    @staticmethod
    def backward(ctx, gz, gkl):
        mu, logstd, epsc = ctx.saved_tensors
        L = _lib.lib()
        N, D = mu.shape
        dml = torch.empty(N, 2 * D, dtype=torch.float32, device=mu.device)
        gzc = None if gz is None else gz.contiguous()
        gk = None if gkl is None else gkl.reshape(1).contiguous()
        _lib.check(L.ceeg_reparam_kl_bwd(mu.data_ptr(), mu.stride(0), logstd.data_ptr(), logstd.stride(0), _lib.ptr(epsc), ctx.seed,
                                         _lib.ptr(gzc), _lib.ptr(gk), N, D, float(MAX_LOGSTD), ctx.log_var, ctx.training,
                                         dml.data_ptr(), 2 * D, dml[:, D:].data_ptr(), 2 * D, _lib.stream()), "ceeg_reparam_kl_bwd")
        if ctx.fused:
            return dml, None, None, None, None, None
        return dml[:, :D], dml[:, D:], None, None, None, None


def _rvq_forward(ml, eps, seed, training, log_var, weight, beta):
    """ceeg_reparam_kl_fwd + ceeg_vq_forward on the fused [N, 2D] encoder output -> (z, kl, scal[loss, ppl], z_q, idx, epsc, wc)."""
    L = _lib.lib()
    N, D = ml.shape[0], ml.shape[1] // 2
    mu_v, ls_v = ml[:, :D], ml[:, D:]
    dev = ml.device
    epsc = None if eps is None else eps.contiguous()
    wc = weight.contiguous()
    n_e = wc.shape[0]
    if wc.shape[1] != D or wc.dtype != torch.float32:
        raise ValueError("the codebook must be float32 [n_e, latent]")
    z = torch.empty(N, D, dtype=torch.float32, device=dev)
    kl = torch.empty((), dtype=torch.float32, device=dev)
    ws = _lib.workspace(L.ceeg_reparam_kl_ws_bytes(N, D), dev)
    _lib.check(L.ceeg_reparam_kl_fwd(mu_v.data_ptr(), mu_v.stride(0), ls_v.data_ptr(), ls_v.stride(0), _lib.ptr(epsc), int(seed),
                                     N, D, float(MAX_LOGSTD), int(log_var), int(training), z.data_ptr(), kl.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_reparam_kl_fwd")
    z_q = torch.empty_like(z)
    idx = torch.empty(N, 1, dtype=torch.int64, device=dev)
    scal = torch.empty(2, dtype=torch.float32, device=dev)
    ws = _lib.workspace(L.ceeg_vq_forward_ws_bytes(N, n_e, D, _lib.CEEG_F32), dev)
    _lib.check(L.ceeg_vq_forward(z.data_ptr(), wc.data_ptr(), N, n_e, D, _lib.CEEG_F32, float(beta), z_q.data_ptr(), idx.data_ptr(),
                                 scal[0:].data_ptr(), scal[1:].data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream()),
               "ceeg_vq_forward")
    return z, kl, scal, z_q, idx, epsc, wc


def _rvq_backward(ml, epsc, wc, idx, z, cfg, g_z, g_kl, g_loss, g_zq, want_dW, want_dml, want_colsum):
    """ceeg_vq_backward (codebook gradient) + ceeg_reparam_vq_bwd -> (dml [N, 2D], its column sums or None, dW)."""
    seed, training, log_var, beta = cfg
    L = _lib.lib()
    N, D = z.shape
    n_e = wc.shape[0]
    dev = z.device
    s = None if g_loss is None else g_loss.reshape(1).to(torch.float32).contiguous()
    gk = None if g_kl is None else g_kl.reshape(1).contiguous()
    dW = None
    if want_dW:
        dW = torch.empty_like(wc)
        sw = s if s is not None else torch.zeros(1, dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.ceeg_vq_backward_ws_bytes(N, n_e, D, _lib.CEEG_F32), dev)
        _lib.check(L.ceeg_vq_backward(z.data_ptr(), wc.data_ptr(), idx.data_ptr(), None, sw.data_ptr(), N, n_e, D, _lib.CEEG_F32, beta,
                                      None, dW.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_vq_backward")
    dml = colsum = None
    if want_dml:
        dml = torch.empty(N, 2 * D, dtype=torch.float32, device=dev)
        colsum = torch.empty(2 * D, dtype=torch.float32, device=dev) if want_colsum else None
        gq = None if g_zq is None else g_zq.contiguous()
        gz = None if g_z is None else g_z.contiguous()
        ws = _lib.workspace(L.ceeg_reparam_vq_bwd_ws_bytes(N, D), dev)
        with _lib.timed("reparam_vq_bwd"):
            _lib.check(L.ceeg_reparam_vq_bwd(ml.data_ptr(), ml.stride(0), ml[:, D:].data_ptr(), ml.stride(0), _lib.ptr(epsc), seed,
                                             _lib.ptr(gq), _lib.ptr(gz), wc.data_ptr(), idx.data_ptr(), n_e, _lib.ptr(s), _lib.ptr(gk),
                                             N, D, float(MAX_LOGSTD), log_var, training, dml.data_ptr(), 2 * D, dml[:, D:].data_ptr(),
                                             2 * D, _lib.ptr(colsum), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_reparam_vq_bwd")
    return dml, colsum, dW


class _ReparamVQ(torch.autograd.Function):
    """(z, kl, vq_loss, z_q, perplexity, idx) = VectorQuantizer(reparametrize(mu | logstd)) with ONE fused backward pass
    (ceeg_reparam_vq_bwd): dz of the quantiser is never written and z is recomputed from (mu, logstd, eps/seed).
    `ml` is the fused [N, 2D] encoder output (mu | logstd); forward = ceeg_reparam_kl_fwd + ceeg_vq_forward as
    _ReparamKL / _VQFunction (scripts/models.py:30-32 order: encoder -> latent -> vector_quantization)."""

    @staticmethod
    def forward(ctx, ml, eps, seed, training, log_var, weight, beta):
        _lib.require_cuda(ml, weight)
        ml = ml.contiguous()
        z, kl, scal, z_q, idx, epsc, wc = _rvq_forward(ml, eps, seed, training, log_var, weight, beta)
        ctx.save_for_backward(ml, epsc, wc, idx, z)
        ctx.cfg = (int(seed), int(training), int(log_var), float(beta))
        loss_t, ppl_t = scal[0], scal[1]
        ctx.mark_non_differentiable(ppl_t, idx)
        # an output nobody differentiates (z in the training step) must not come back as an [N, D] tensor of zeros: that is a
        # 0.5 GB fill plus a 0.5 GB read inside the fused backward; the backward takes None for it
        ctx.set_materialize_grads(False)
        return z, kl, loss_t, z_q, ppl_t, idx

    @staticmethod
    def backward(ctx, g_z, g_kl, g_loss, g_zq, _g_ppl, _g_idx):
        ml, epsc, wc, idx, z = ctx.saved_tensors
        dml, _, dW = _rvq_backward(ml, epsc, wc, idx, z, ctx.cfg, g_z, g_kl, g_loss, g_zq, ctx.needs_input_grad[5], ctx.needs_input_grad[0], False)
        return dml, None, None, None, None, dW, None


class _HeadReparamVQ(torch.autograd.Function):
    """The encoder head's dense half and everything up to the quantiser as ONE autograd node:
        ml = ah @ [W_mu | W_sigma]^T + [b_mu | b_sigma]      (ah = A_hat h: the head in aggregate-first form)
        (z, kl) = reparametrize + KL;  (vq_loss, z_q, perplexity, idx) = VectorQuantizer(z)
    Its backward takes the bias gradient from the column sums ceeg_reparam_vq_bwd leaves in the same pass - a real output of
    this function, not a hint passed between nodes - and returns dah = dml W, dW = dml^T ah (split-K, fixed order).
    `ml` is also an output (the caller may attach a loss to mu | logstd): a gradient arriving for it is added to dml and the
    column sums are then recomputed."""

    @staticmethod
    def forward(ctx, ah, wcat, bcat, eps, seed, training, log_var, weight, beta):
        _lib.require_cuda(ah, weight)
        ml = _linear_fwd(ah, wcat, bcat, False)
        z, kl, scal, z_q, idx, epsc, wc = _rvq_forward(ml, eps, seed, training, log_var, weight, beta)
        ctx.save_for_backward(ah, wcat, ml, epsc, wc, idx, z)
        ctx.cfg = (int(seed), int(training), int(log_var), float(beta))
        loss_t, ppl_t = scal[0], scal[1]
        ctx.mark_non_differentiable(ppl_t, idx)
        ctx.set_materialize_grads(False)
        return ml, z, kl, loss_t, z_q, ppl_t, idx

    @staticmethod
    def backward(ctx, g_ml, g_z, g_kl, g_loss, g_zq, _g_ppl, _g_idx):
        ah, wcat, ml, epsc, wc, idx, z = ctx.saved_tensors
        need = ctx.needs_input_grad
        dml, colsum, dW = _rvq_backward(ml, epsc, wc, idx, z, ctx.cfg, g_z, g_kl, g_loss, g_zq, need[7], need[0] or need[1] or need[2],
                                        need[2] and g_ml is None)
        dah = dwcat = dbcat = None
        if dml is not None:
            if g_ml is not None:                      # a second loss on mu | logstd
                dml = dml + g_ml
                colsum = _colsum(dml) if need[2] else None
            dbcat = colsum
            dwcat = _weight_grad(ah, dml) if need[1] else None
            dah = torch.mm(dml, wcat) if need[0] else None
        return dah, dwcat, dbcat, None, None, None, None, dW, None


def _reparam_vq(ml, eps, seed, training, vq):
    """reparametrise + KL + VectorQuantizer on the fused [N, 2D] encoder output; the fused-backward form when the shapes
    allow it (fp32, latent % 4 == 0), else the two separate autograd functions."""
    D = ml.shape[1] // 2
    if D % 4 == 0 and os.environ.get("CEEG_FUSED_RVQ", "1") == "1":
        z, kl, vq_loss, z_q, ppl, idx = _ReparamVQ.apply(ml, eps, seed, training, False, vq.embedding.weight, vq.beta)
        return z, kl, vq_loss, z_q, ppl, idx
    z, kl = _ReparamKL.apply(ml, None, eps, seed, training, False)
    vq_loss, z_q, ppl, _, idx = vq(z)
    return z, kl, vq_loss, z_q, ppl, idx


def _edge_lists_i32(edge_index):
    ei = edge_index.to(torch.int32)
    return ei[0].contiguous(), ei[1].contiguous()


class _EdgeBCE(torch.autograd.Function):
    """(sum_e term_e, number of pairs counted) over an arbitrary edge list; backward = two SpMMs over the list's CSR and
    transpose.  Pairs with a negative id (samples the sampler gave up on) are skipped.  table / table_idx: a quantised
    latent given as (codebook, int32 code per node) with z[n] == table[table_idx[n]] bit for bit - the forward then gathers
    codebook rows (L2-resident) through the 4-byte index; the backward's SpMMs gather rows of `z` itself."""

    @staticmethod
    def forward(ctx, z, edge_index, negative, pair, table=None, table_idx=None):
        _lib.require_cuda(z, edge_index)
        L = _lib.lib()
        z = z.contiguous()
        N, D = z.shape
        i, j = _edge_lists_i32(edge_index)
        E = i.shape[0]
        coef = torch.empty(max(E, 1), dtype=torch.float32, device=z.device)[:E]
        out = torch.zeros(2, dtype=torch.float32, device=z.device)
        ws = _lib.workspace(L.ceeg_edge_bce_ws_bytes(E, D), z.device)
        src, sidx = (z, None) if table is None else (table.detach().contiguous(), table_idx.contiguous())
        if table is not None and src.shape[0] <= 4096 and os.environ.get("CEEG_EDGE_CODES", "1") == "1":
            # code form: the dot product of a pair is an entry of the codebook's Gram matrix
            ws = _lib.workspace(L.ceeg_edge_bce_codes_ws_bytes(E, src.shape[0]), z.device)
            with _lib.timed("edge_bce_fwd"):
                _lib.check(L.ceeg_edge_bce_codes_fwd(src.data_ptr(), src.stride(0), src.shape[0], D, sidx.data_ptr(), i.data_ptr(), j.data_ptr(), E,
                                                     int(negative), coef.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()),
                           "ceeg_edge_bce_codes_fwd")
        else:
            with _lib.timed("edge_bce_fwd"):
                _lib.check(L.ceeg_edge_bce_fwd(src.data_ptr(), src.stride(0), _lib.ptr(sidx), i.data_ptr(), j.data_ptr(), E, D, int(negative),
                                               coef.data_ptr(), None, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_edge_bce_fwd")
        if table is not None:
            ctx.save_for_backward(src, coef, sidx)
        else:
            ctx.save_for_backward(z, coef)
        ctx.shape = (N, D)
        ctx.edge_index, ctx.pair = edge_index, pair
        ctx.mark_non_differentiable(out[1])
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g, _gc):
        if len(ctx.saved_tensors) == 3:
            z, coef, xidx = ctx.saved_tensors   # z = the codebook, rows through the code index (bitwise the same sums)
        else:
            (z, coef), xidx = ctx.saved_tensors, None
        N = ctx.shape[0]
        # the CSR of a fresh edge list (sampled negatives) is built here, once per step: a stable sort by (dst, src) on the
        # device; skipped pairs (negative ids) carry coef 0, so any in-range endpoint will do for them
        pair = ctx.pair if ctx.pair is not None else CsrPair(ctx.edge_index.clamp(min=0), N, self_loops=False)
        # CSR rows = edge_index[1] (j), cols = edge_index[0] (i); perm maps CSR position -> edge id
        cv = coef[pair.perm.long()] * g
        dz = _spmm(pair.rowptr, pair.col, cv, None, z, N, max_row=pair.max_row, xidx=xidx)
        _spmm(pair.t_rowptr, pair.t_col, cv, pair.t_perm, z, N, out=dz, accumulate=True, max_row=pair.max_trow, xidx=xidx)
        return dz, None, None, None, None, None


class _PairBCECodes(torch.autograd.Function):
    """(sum_k term_k, pairs counted) over an arbitrary pair list on a QUANTISED latent (ceeg_pair_bce_codes_fwd / _bwd): dots
    from the codebook's Gram matrix; the backward groups the 2E incidences by node with one device radix sort and sums
    c * W[code] from an LDS codebook slab - no CSR pair of the list, no host sync, no float atomics."""

    @staticmethod
    def supported(table, num_nodes, num_pairs):
        return _lib.lib().ceeg_pair_bce_codes_supported(int(table.shape[0]), int(table.shape[1]), int(num_nodes), int(num_pairs)) == 1

    @staticmethod
    def forward(ctx, z, edge_index, negative, table, table_idx):
        _lib.require_cuda(z, edge_index)
        L = _lib.lib()
        N, D = z.shape
        W, code = table.detach().contiguous(), table_idx.contiguous()
        i, j = _edge_lists_i32(edge_index)
        E = i.shape[0]
        out = torch.zeros(2, dtype=torch.float32, device=z.device)
        want = ctx.needs_input_grad[0]
        keys = torch.empty(max(2 * E, 2), dtype=torch.int32, device=z.device) if want else None
        ctab = torch.empty(W.shape[0] * W.shape[0], dtype=torch.float32, device=z.device)
        ws = _lib.workspace(L.ceeg_pair_bce_codes_ws_bytes(E, N, W.shape[0]), z.device)
        with _lib.timed("pair_bce_fwd"):
            _lib.check(L.ceeg_pair_bce_codes_fwd(W.data_ptr(), W.stride(0), W.shape[0], D, code.data_ptr(), i.data_ptr(), j.data_ptr(), E, N,
                                                 int(negative), _lib.ptr(keys), ctab.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _lib.stream()), "ceeg_pair_bce_codes_fwd")
        if want:
            ctx.save_for_backward(W, keys, ctab, code)
        ctx.shape, ctx.E = (N, D), E
        ctx.mark_non_differentiable(out[1])
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g, _gc):
        W, keys, ctab, code = ctx.saved_tensors
        L = _lib.lib()
        N, D = ctx.shape
        dz = torch.empty(N, D, dtype=torch.float32, device=W.device)
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        ws = _lib.workspace(L.ceeg_pair_bce_codes_ws_bytes(ctx.E, N, W.shape[0]), W.device)
        with _lib.timed("pair_bce_bwd"):
            _lib.check(L.ceeg_pair_bce_codes_bwd(W.data_ptr(), W.stride(0), W.shape[0], D, code.data_ptr(), keys.data_ptr(), ctab.data_ptr(), ctx.E, N,
                                                 gs.data_ptr(), 0, dz.data_ptr(), dz.stride(0), ws.data_ptr(), ws.numel(), _lib.stream()),
                       "ceeg_pair_bce_codes_bwd")
        return dz, None, None, None, None


def _neg_filter(pair):
    """(bitmap, bits) of the sampler's edge-set prefilter, built on first use and kept with the CSR pair; (None, 0) when
    CEEG_NEG_FILTER=0 (the plain search: the checker form of the tests)."""
    if os.environ.get("CEEG_NEG_FILTER", "1") != "1":
        return None, 0
    if getattr(pair, "neg_filter", None) is None:
        L = _lib.lib()
        nbits = int(L.ceeg_neg_filter_bits(pair.nnz))
        bitmap = torch.empty((1 << nbits) // 32, dtype=torch.int32, device=pair.device)
        _lib.check(L.ceeg_neg_filter_build(pair.rowptr.data_ptr(), pair.col.data_ptr(), pair.num_nodes, nbits, bitmap.data_ptr(),
                                           _lib.stream()), "ceeg_neg_filter_build")
        pair.neg_filter = (bitmap, nbits)
    return pair.neg_filter


class _UniformRecon(torch.autograd.Function):
    """recon_loss(z_q, pos_edge_index, neg_edge_index) of the literal sampler path in ONE autograd node, code form throughout:
    positives through ceeg_decoder_codes_fwd with the structured negatives off (sig_a = 0), the sampled pairs through
    ceeg_pair_bce_codes_fwd - or, when the pairs are to be drawn here (neg_edge_index None, `seed` given), through
    ceeg_neg_sample_pair_bce_codes_fwd, which draws pair k inside the kernel (same samples bit for bit) - and ONE gradient kernel for
    both pair sets (ceeg_decoder_codes_bwd_pairs: the sampled pairs' node-sorted incidences are a fourth list of every node).
    CEEG_PAIR_MERGED=0 keeps the two-launch forms (separate sampler; second gradient kernel accumulating into dz): bitwise equal.
    Returns (pos_mean + neg_sum / neg_count, [pos_sum, neg_sum, neg_count], neg_edge_index int32 [2, E])."""

    @staticmethod
    def supported(table, num_nodes, num_pairs):
        L = _lib.lib()
        return (L.ceeg_decoder_codes_supported(int(table.shape[0]), int(table.shape[1])) == 1
                and L.ceeg_pair_bce_codes_supported(int(table.shape[0]), int(table.shape[1]), int(num_nodes), int(num_pairs)) == 1)

    @staticmethod
    def forward(ctx, z, graph, neg_edge_index, table, table_idx, seed=None, num_neg=None):
        _lib.require_cuda(z, neg_edge_index)
        L = _lib.lib()
        N, D = z.shape
        W, code = table.detach().contiguous(), table_idx.contiguous()
        K = W.shape[0]
        p = graph.dec
        dev = z.device
        merged = os.environ.get("CEEG_PAIR_MERGED", "1") == "1"
        li, lt = p.long_items(False), p.long_items(True)
        nl = 0 if li is None else li.shape[0]
        rec = torch.empty(max(p.nnz, 1), 4, dtype=torch.int32, device=dev)
        psums = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.ceeg_decoder_codes_ws_bytes(N, p.nnz, K, D, nl, 0), dev)
        with _lib.timed("decoder_fwd"):
            _lib.check(L.ceeg_decoder_codes_fwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), p.rowptr.data_ptr(), p.col.data_ptr(), N, p.nnz,
                                                0, 0, N, _lib.ptr(li), nl, p.max_row, None, rec.data_ptr(), psums.data_ptr(), ws.data_ptr(),
                                                ws.numel(), _lib.stream()), "ceeg_decoder_codes_fwd")
        if neg_edge_index is None and not merged:
            neg_edge_index = negative_sampling_raw(graph, N, int(num_neg), seed)
        E = int(num_neg) if neg_edge_index is None else int(neg_edge_index.shape[1])
        want = ctx.needs_input_grad[0]
        nkeys = torch.empty(max(2 * E, 2), dtype=torch.int32, device=dev) if want else None
        ctab = torch.empty(K * K, dtype=torch.float32, device=dev)
        nout = torch.zeros(2, dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.ceeg_pair_bce_codes_ws_bytes(E, N, K), dev)
        if neg_edge_index is None:   # draw the pairs inside the forward kernel
            neg_edge_index = torch.empty(2, max(E, 1), dtype=torch.int32, device=dev)[:, :E]
            filt, bits = _neg_filter(p)
            with _lib.timed("pair_bce_fwd"):
                _lib.check(L.ceeg_neg_sample_pair_bce_codes_fwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), p.rowptr.data_ptr(), p.col.data_ptr(),
                                                                N, E, int(seed), _lib.ptr(filt), bits, neg_edge_index[0].data_ptr(),
                                                                neg_edge_index[1].data_ptr(), _lib.ptr(nkeys), ctab.data_ptr(), nout.data_ptr(),
                                                                ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_neg_sample_pair_bce_codes_fwd")
        else:
            i, j = _edge_lists_i32(neg_edge_index)
            with _lib.timed("pair_bce_fwd"):
                _lib.check(L.ceeg_pair_bce_codes_fwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), i.data_ptr(), j.data_ptr(), E, N, 1,
                                                     _lib.ptr(nkeys), ctab.data_ptr(), nout.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()),
                           "ceeg_pair_bce_codes_fwd")
        out = torch.empty(6, dtype=torch.float32, device=dev)   # rec, 1/E_pos, 1/pairs counted, pos_sum, neg_sum, neg_count
        _lib.check(L.ceeg_recon_finish(psums.data_ptr(), nout.data_ptr(), nout[1:].data_ptr(), float(max(p.nnz, 1)), out.data_ptr(),
                                       _lib.stream()), "ceeg_recon_finish")
        if want:
            ctx.save_for_backward(W, rec, nkeys, ctab, code, out)
        ctx.graph, ctx.items, ctx.shape, ctx.E, ctx.merged = graph, (li, lt), (N, D), E, merged
        sums = out[3:6]
        ctx.mark_non_differentiable(sums, neg_edge_index)
        ctx.set_materialize_grads(False)   # no zero "gradient" tensors for the integer / statistics outputs (a 128 MB fill per step)
        return out[0], sums, neg_edge_index

    @staticmethod
    def backward(ctx, g, _gs, _gn):
        W, rec, nkeys, ctab, code, out = ctx.saved_tensors
        L = _lib.lib()
        N, D = ctx.shape
        K = W.shape[0]
        p = ctx.graph.dec
        li, lt = ctx.items
        nl, nt = (0 if li is None else li.shape[0]), (0 if lt is None else lt.shape[0])
        gscale = (out[1:3] * g).contiguous()                 # [g / E_pos, g / pairs counted]
        dz = torch.empty(N, D, dtype=torch.float32, device=W.device)
        if ctx.merged:
            ws = _lib.workspace(L.ceeg_decoder_codes_bwd_pairs_ws_bytes(N, p.nnz, K, D, nl, nt, ctx.E), W.device)
            with _lib.timed("decoder_bwd"):
                _lib.check(L.ceeg_decoder_codes_bwd_pairs(W.data_ptr(), W.stride(0), K, D, p.rowptr.data_ptr(), p.t_rowptr.data_ptr(),
                                                          p.t_perm.data_ptr(), N, p.nnz, _lib.ptr(li), nl, _lib.ptr(lt), nt, p.max_row, p.max_trow,
                                                          rec.data_ptr(), code.data_ptr(), nkeys.data_ptr(), ctab.data_ptr(), ctx.E,
                                                          gscale.data_ptr(), dz.data_ptr(), dz.stride(0), ws.data_ptr(), ws.numel(), _lib.stream()),
                           "ceeg_decoder_codes_bwd_pairs")
            return dz, None, None, None, None, None, None
        ws = _lib.workspace(L.ceeg_decoder_codes_ws_bytes(N, p.nnz, K, D, nl, nt), W.device)
        with _lib.timed("decoder_bwd"):
            _lib.check(L.ceeg_decoder_codes_bwd(W.data_ptr(), W.stride(0), K, D, p.rowptr.data_ptr(), p.t_rowptr.data_ptr(), p.t_perm.data_ptr(),
                                                N, p.nnz, 0, 0, 1, N, _lib.ptr(li), nl, _lib.ptr(lt), nt, p.max_row, p.max_trow, rec.data_ptr(),
                                                gscale.data_ptr(), dz.data_ptr(), dz.stride(0), None, ws.data_ptr(), ws.numel(), _lib.stream()),
                       "ceeg_decoder_codes_bwd")
        ws = _lib.workspace(L.ceeg_pair_bce_codes_ws_bytes(ctx.E, N, K), W.device)
        with _lib.timed("pair_bce_bwd"):
            _lib.check(L.ceeg_pair_bce_codes_bwd(W.data_ptr(), W.stride(0), K, D, code.data_ptr(), nkeys.data_ptr(), ctab.data_ptr(), ctx.E, N,
                                                 gscale[1:].data_ptr(), 1,
                                                 dz.data_ptr(), dz.stride(0), ws.data_ptr(), ws.numel(), _lib.stream()),
                       "ceeg_pair_bce_codes_bwd")
        return dz, None, None, None, None, None, None


class _LossSum(torch.autograd.Function):
    """loss = _loss_sum(rec, kl, vq_loss, N) (scripts/run.py:95 plus the VGAE example's (1/N) kl_loss) as one launch each way instead of
    a chain of scalar torch kernels and their autograd nodes."""

    @staticmethod
    def forward(ctx, rec, kl, vq_loss, kl_scale):
        _lib.require_cuda(rec, kl, vq_loss)
        out = torch.empty((), dtype=torch.float32, device=rec.device)
        _lib.check(_lib.lib().ceeg_loss_sum(rec.data_ptr(), kl.data_ptr(), vq_loss.data_ptr(), float(kl_scale), None, out.data_ptr(), _lib.stream()),
                   "ceeg_loss_sum")
        ctx.kl_scale = float(kl_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        gs = torch.empty(2, dtype=torch.float32, device=g.device)
        gc = g.contiguous()
        _lib.check(_lib.lib().ceeg_loss_sum(None, None, None, ctx.kl_scale, gc.data_ptr(), gs.data_ptr(), _lib.stream()), "ceeg_loss_sum")
        return gs[0], gs[1], gs[0], None


def _loss_sum(rec, kl, vq_loss, n):
    if rec.dtype == kl.dtype == vq_loss.dtype == torch.float32 and rec.dim() == kl.dim() == vq_loss.dim() == 0:
        return _LossSum.apply(rec, kl, vq_loss, 1.0 / n)
    return rec + kl / n + vq_loss


class InnerProductDecoder(nn.Module):
    """PyG InnerProductDecoder: sigma(<z_i, z_j>) per edge (sigmoid=False: the raw inner product).  Inference-only: it
    carries no autograd path - training goes through VGAE.recon_loss, whose loss and gradient are fused kernels."""

    def forward(self, z, edge_index, sigmoid=True):
        _lib.require_cuda(z, edge_index)
        L = _lib.lib()
        zc = z.detach().contiguous()
        i, j = _edge_lists_i32(edge_index)
        E, D = i.shape[0], zc.shape[1]
        prob = torch.empty(max(E, 1), dtype=torch.float32, device=z.device)[:E]
        out = torch.zeros(2, dtype=torch.float32, device=z.device)
        ws = _lib.workspace(L.ceeg_edge_bce_ws_bytes(E, D), z.device)
        _lib.check(L.ceeg_edge_bce_fwd(zc.data_ptr(), zc.stride(0), None, i.data_ptr(), j.data_ptr(), E, D, 0 if sigmoid else 2, None,
                                       prob.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_edge_bce_fwd")
        return prob

    def forward_all(self, z, sigmoid=True):
        adj = z @ z.t()  # dense N x N: small graphs only (out of scope at N = 1M)
        return torch.sigmoid(adj) if sigmoid else adj


def negative_sampling_raw(graph_or_pair, num_nodes, num, seed):
    """ceeg_neg_sample_uniform_filtered: int32 [2, num] on the device, (-1, -1) where 16 attempts all collided.  No host sync.
    The edge-set prefilter (a bitmap, >= 16 bits per edge) is built on the first call and kept with the graph's CSR."""
    L = _lib.lib()
    pair = graph_or_pair.dec if isinstance(graph_or_pair, EventGraph) else graph_or_pair
    out = torch.empty(2, max(int(num), 1), dtype=torch.int32, device=pair.device)[:, : int(num)]
    filt, bits = _neg_filter(pair)
    with _lib.timed("neg_sample"):
        _lib.check(L.ceeg_neg_sample_uniform_filtered(pair.rowptr.data_ptr(), pair.col.data_ptr(), int(num_nodes), int(num), int(seed), _lib.ptr(filt),
                                                      bits, out[0].data_ptr(), out[1].data_ptr(), _lib.stream()), "ceeg_neg_sample_uniform_filtered")
    return out


def negative_sampling(edge_index, num_nodes, num_neg_samples=None, seed=0, graph=None):
    """Uniform negative sampler (PyG negative_sampling surface).  Deterministic under `seed`
    (Philox4x32-10, restated in oracle/ceeg_oracle.c); samples that collide 16 times are dropped."""
    _lib.require_cuda(edge_index)
    pair = graph.dec if isinstance(graph, EventGraph) else CsrPair(edge_index, num_nodes, self_loops=False)
    num = int(edge_index.shape[1] if num_neg_samples is None else num_neg_samples)
    out = negative_sampling_raw(pair, num_nodes, num, seed).to(torch.int64)
    return out[:, out[0] >= 0]


def sigma_params(num_nodes, seed, step):
    """(a, b, a^-1) of the per-step node permutation sigma(x) = (a x + b) mod N used by the fused decoder's
    permutation-structured negatives; splitmix64 of (seed, step), a bumped until gcd(a, N) == 1.
    For a collated batch of equal-sized graphs pass the nodes PER GRAPH as `num_nodes` and hand the same number to
    the decoder as `sig_mod`: sigma then permutes inside every graph (include/ceeg.h, ceeg_decoder_fused_fwd)."""
    def mix(x):
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return x ^ (x >> 31)
    N = int(num_nodes)
    if N < 2:
        return 1, 0, 1
    h = mix(mix(int(seed)) ^ (int(step) * 0xD1342543DE82EF95 & (2**64 - 1)))
    a = 1 + h % (N - 1)
    while math.gcd(a, N) != 1:
        a = 1 + a % (N - 1)
    b = mix(h) % N
    return a, b, pow(a, -1, N)


class _FusedDecoder(torch.autograd.Function):
    """recon_loss with permutation-structured negatives: one gather pass for both terms (ceeg_decoder_fused_*)."""

    @staticmethod
    def forward(ctx, z, graph, a, b, ainv, row_sums=True, sig_mod=None, table=None, table_idx=None, cond_rows=None):
        """table / table_idx: a quantised latent given as (codebook [n_e, D], int32 code per node) with
        z[n] == table[table_idx[n]] bit for bit (what VectorQuantizer's forward returns in fp32): the kernels then
        gather codebook rows through the 4-byte index instead of D*4-byte rows of z.  Gradients still flow to z."""
        _lib.require_cuda(z)
        L = _lib.lib()
        N, D = z.shape
        mod = int(N if sig_mod is None else sig_mod)
        # cond_rows [N, C]: extra decoder-input columns that are constant over every pair of a node (a per-graph condition,
        # block-diagonal batch, sigma inside the graphs): the code form folds them into a per-row addend; any other form
        # decodes the concatenation cat((z, cond_rows)) (scripts/models.py:259)
        # cond_rows may also be (condition [G, C], batch [N]) = the rows condition[batch] without materialising them (no gradient
        # reaches the condition through this form)
        ctx.cond_split = None
        use_codes = (table is not None and os.environ.get("CEEG_DEC_CODES", "1") == "1"
                     and L.ceeg_decoder_codes_supported(int(table.shape[0]), D) == 1)
        cond_graphs = None
        ctx.cond_is_tensor = torch.is_tensor(cond_rows)
        if isinstance(cond_rows, tuple):
            cg, cb = cond_rows
            if use_codes and cg.dtype == torch.float32 and cg.shape[1] % 4 == 0:
                cond_graphs, cond_rows = (cg.detach().contiguous(), cb), None
            else:
                cond_rows = cg.detach().index_select(0, cb)
        if cond_rows is not None and not use_codes:
            ctx.cond_split = (D, cond_rows.shape[1])
            z = torch.cat((z, cond_rows), dim=1)
            N, D = z.shape
            table = table_idx = None
        if table is not None:
            src, sidx = table.detach().contiguous(), table_idx.contiguous()
            if src.shape[1] != D or sidx.shape[0] != N or sidx.dtype != torch.int32:
                raise ValueError("table must be [n_e, D] and table_idx int32 [N]")
        else:
            src, sidx = z.contiguous(), None
        z = src
        p = graph.dec
        sums = torch.empty(3, dtype=torch.float32, device=z.device)
        ctx.codes = use_codes and table is not None
        ctx.cond = None
        if ctx.codes:
            # code form: dots from the Gram matrix of the codebook, gradient from an LDS-resident codebook slab
            li, lt = p.long_items(False), p.long_items(True)
            ctx.items = (li, lt)
            rec = torch.empty(max(p.nnz, 1), 4, dtype=torch.int32, device=z.device)
            row_add = None
            if cond_graphs is not None:   # |c_graph|^2 per graph, broadcast to the rows through batch
                cg, cb = cond_graphs
                per_graph = torch.empty(cg.shape[0], dtype=torch.float32, device=z.device)
                _lib.check(L.ceeg_row_sqnorm(cg.data_ptr(), cg.stride(0), cg.shape[0], cg.shape[1], per_graph.data_ptr(), _lib.stream()),
                           "ceeg_row_sqnorm")
                row_add = per_graph.index_select(0, cb)
            elif cond_rows is not None:
                ctx.cond = cond_rows.detach().contiguous()
                # |c_graph(r)|^2: the condition's share of every dot of row r
                if ctx.cond.shape[1] % 4 == 0 and ctx.cond.dtype == torch.float32 and ctx.cond.data_ptr() % 16 == 0:
                    row_add = torch.empty(ctx.cond.shape[0], dtype=torch.float32, device=z.device)
                    _lib.check(L.ceeg_row_sqnorm(ctx.cond.data_ptr(), ctx.cond.stride(0), ctx.cond.shape[0], ctx.cond.shape[1],
                                                 row_add.data_ptr(), _lib.stream()), "ceeg_row_sqnorm")
                else:
                    row_add = (ctx.cond * ctx.cond).sum(dim=1).contiguous()
            ws = _lib.workspace(L.ceeg_decoder_codes_ws_bytes(N, p.nnz, src.shape[0], D, 0 if li is None else li.shape[0], 0), z.device)
            with _lib.timed("decoder_fwd"):
                _lib.check(L.ceeg_decoder_codes_fwd(src.data_ptr(), src.stride(0), src.shape[0], D, sidx.data_ptr(), p.rowptr.data_ptr(),
                                                    p.col.data_ptr(), N, p.nnz, a, b, mod, _lib.ptr(li), 0 if li is None else li.shape[0],
                                                    p.max_row, _lib.ptr(row_add), rec.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                                    _lib.stream()), "ceeg_decoder_codes_fwd")
            coef = rec
            ctx.grow = None
        else:
            coef = torch.empty(max(p.nnz, 1), 2, dtype=torch.float32, device=z.device)  # (d pos / dv, d neg / dv) per entry, interleaved
            cp, cn = coef[:, 0], coef[:, 1]
            # with a gradient wanted, the forward also leaves the row-side halves of dz (two [N, D] tables)
            grow = torch.empty(2, N, D, dtype=torch.float32, device=z.device) if (ctx.needs_input_grad[0] and row_sums) else None
            ws = _lib.workspace(L.ceeg_decoder_fused_ws_bytes(N, p.nnz, D), z.device)
            with _lib.timed("decoder_fwd"):
                _lib.check(L.ceeg_decoder_fused_fwd(z.data_ptr(), z.stride(0), _lib.ptr(sidx), p.rowptr.data_ptr(), p.col.data_ptr(), N, p.nnz, D, a, b, mod,
                                                    cp.data_ptr(), cn.data_ptr(), 2, _lib.ptr(grow), None if grow is None else grow[1].data_ptr(), D,
                                                    sums.data_ptr(), p.max_row, ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_decoder_fused_fwd")
            ctx.grow = grow
        out = torch.empty(6, dtype=torch.float32, device=z.device)   # rec, 1/E_pos, 1/valid negatives, then the three sums again
        _lib.check(L.ceeg_recon_finish(sums.data_ptr(), sums[1:].data_ptr(), sums[2:].data_ptr(), float(max(p.nnz, 1)), out.data_ptr(),
                                       _lib.stream()), "ceeg_recon_finish")
        ctx.save_for_backward(z, coef, out)
        ctx.sidx, ctx.shape = sidx, (N, D)
        ctx.graph, ctx.sig = graph, (a, b, ainv, mod)
        ctx.mark_non_differentiable(sums)
        return out[0], sums

    @staticmethod
    def backward(ctx, g, _gs):
        z, coef, out = ctx.saved_tensors
        L = _lib.lib()
        N, D = ctx.shape
        p = ctx.graph.dec
        sidx = ctx.sidx
        a, b, ainv, mod = ctx.sig
        if a == 0 and not ctx.codes:  # positives only: every cn is 0; sigma is the identity for the backward's addressing
            a, b, ainv = 1, 0, 1          # (the code form takes a == 0 itself and walks only the positive lists)
        gscale = (out[1:3] * g).contiguous()
        dz = torch.empty(N, D, dtype=torch.float32, device=z.device)
        if ctx.codes:
            li, lt = ctx.items
            nl, nt = (0 if li is None else li.shape[0]), (0 if lt is None else lt.shape[0])
            ws = _lib.workspace(L.ceeg_decoder_codes_ws_bytes(N, p.nnz, z.shape[0], D, nl, nt), z.device)
            want_dcond = ctx.cond is not None and ctx.needs_input_grad[9]
            csum = torch.empty(N, dtype=torch.float32, device=z.device) if want_dcond else None
            with _lib.timed("decoder_bwd"):
                _lib.check(L.ceeg_decoder_codes_bwd(z.data_ptr(), z.stride(0), z.shape[0], D, p.rowptr.data_ptr(), p.t_rowptr.data_ptr(),
                                                    p.t_perm.data_ptr(), N, p.nnz, a, b, ainv, mod, _lib.ptr(li), nl, _lib.ptr(lt), nt,
                                                    p.max_row, p.max_trow, coef.data_ptr(), gscale.data_ptr(), dz.data_ptr(), dz.stride(0),
                                                    _lib.ptr(csum), ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_decoder_codes_bwd")
            dcond = None if csum is None else csum[:, None] * ctx.cond
            return dz, None, None, None, None, None, None, None, None, dcond
        cp, cn = coef[:, 0], coef[:, 1]
        ws = _lib.workspace(L.ceeg_decoder_fused_ws_bytes(N, p.nnz, D), z.device)
        grow = ctx.grow
        with _lib.timed("decoder_bwd"):
            _lib.check(L.ceeg_decoder_fused_bwd(z.data_ptr(), z.stride(0), _lib.ptr(sidx), p.rowptr.data_ptr(), p.col.data_ptr(), p.t_rowptr.data_ptr(),
                                                p.t_col.data_ptr(), p.t_perm.data_ptr(), N, p.nnz, D, a, b, ainv, mod, cp.data_ptr(),
                                                cn.data_ptr(), 2, _lib.ptr(grow), None if grow is None else grow[1].data_ptr(), D,
                                                gscale.data_ptr(), dz.data_ptr(), dz.stride(0), p.max_row, p.max_trow, ws.data_ptr(), ws.numel(),
                                                _lib.stream()), "ceeg_decoder_fused_bwd")
        ctx.grow = None
        if ctx.cond_split is not None:
            d0, c0 = ctx.cond_split
            return dz[:, :d0], None, None, None, None, None, None, None, None, (dz[:, d0:] if ctx.cond_is_tensor else None)
        return dz, None, None, None, None, None, None, None, None, None


# ==================================================================================================
# VGAE (PyG surface) and the VGAE + VQ model of the north-star step
# ==================================================================================================
class VariationalGCNEncoder(nn.Module):
    """2-layer variational GCN encoder (PyG example convention): conv1 F_in -> 2*latent (ReLU),
    conv_mu / conv_logstd 2*latent -> latent.  The two heads run as ONE SpMM over h @ [W_mu | W_sigma]^T."""

    def __init__(self, in_channels, latent, compute_dtype=torch.float32):
        super().__init__()
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = compute_dtype
        self.conv1 = GCNConv(in_channels, 2 * latent, aggregate_first=in_channels < 2 * latent)
        self.conv_mu = GCNConv(2 * latent, latent)
        self.conv_logstd = GCNConv(2 * latent, latent)
        self.latent = latent

    def forward_fused(self, x, graph):
        h = self.conv1(x, graph, relu=True)
        wcat = torch.cat([self.conv_mu.lin.weight, self.conv_logstd.lin.weight], dim=0)
        bcat = torch.cat([self.conv_mu.bias, self.conv_logstd.bias], dim=0)
        return _GcnAggregate.apply(_Linear.apply(h, wcat, None), bcat, graph, False)  # [N, 2*latent] = mu | logstd

    def forward(self, x, graph):
        graph = _as_graph(graph, x.shape[0])
        ml = self.forward_fused(x, graph)
        return ml[:, : self.latent], ml[:, self.latent:]

    def encode_reparam_vq(self, x, graph, eps, seed, training, vq):
        """(ml, z, kl, vq_loss, z_q, perplexity, idx): encoder -> reparametrise + KL -> VectorQuantizer (scripts/models.py:30-32).
        When the first layer runs aggregate-first (in_channels < 2 * latent) the step is three autograd nodes -
        A_hat x;  A_hat relu((A_hat x) W1^T + b1) with the ReLU mask in the transposed pass' epilogue;  head product +
        reparametrise + quantiser with the head's bias gradient a real output of the fused backward - the same arithmetic
        re-associated: (A_hat h) W^T + b == A_hat (h W^T) + b.  CEEG_FUSED_HEAD=0 keeps the layer-by-layer form (the checker)."""
        D = self.latent
        if self.compute_dtype == torch.bfloat16:
            # opt-in reduced precision (north star: "rtol 1e-2 (bf16)"): activations stored as bf16 - every gathered row is half
            # the bytes -, dense products on the bf16 matrix cores, fp32 accumulation throughout; parameters, mu | logstd, the
            # noise, the quantiser and the decoder stay fp32.  Pass x already as bf16 to keep its conversion out of the step.
            if not self.conv1.aggregate_first or D % 4 or x.shape[1] % 8:
                raise ValueError("the bf16 encoder needs in_channels < 2 * latent, in_channels % 8 == 0 and latent % 4 == 0 (16-byte row pieces)")
            with torch.no_grad():
                ax = graph.aggregate16(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16))
            ah = _LinearReluAggregate16.apply(ax, self.conv1.lin.weight, self.conv1.bias, graph)
            wcat = torch.cat([self.conv_mu.lin.weight, self.conv_logstd.lin.weight], dim=0)
            bcat = torch.cat([self.conv_mu.bias, self.conv_logstd.bias], dim=0)
            return _HeadReparamVQ16.apply(ah, wcat, bcat, eps, seed, training, False, vq.embedding.weight, vq.beta)
        if (self.conv1.aggregate_first and D % 4 == 0 and os.environ.get("CEEG_FUSED_RVQ", "1") == "1"
                and os.environ.get("CEEG_FUSED_HEAD", "1") == "1"):
            ax = _GcnAggregate.apply(x, None, graph, False)
            ah = _LinearReluAggregate.apply(ax, self.conv1.lin.weight, self.conv1.bias, graph)
            wcat = torch.cat([self.conv_mu.lin.weight, self.conv_logstd.lin.weight], dim=0)
            bcat = torch.cat([self.conv_mu.bias, self.conv_logstd.bias], dim=0)
            return _HeadReparamVQ.apply(ah, wcat, bcat, eps, seed, training, False, vq.embedding.weight, vq.beta)
        ml = self.forward_fused(x, graph)
        return (ml,) + tuple(_reparam_vq(ml, eps, seed, training, vq))


class VGAE(nn.Module):
    """PyG VGAE surface: encode / reparametrize / kl_loss / recon_loss / decoder."""

    def __init__(self, encoder, decoder=None):
        super().__init__()
        self.encoder = encoder
        self.decoder = InnerProductDecoder() if decoder is None else decoder
        self.seed, self._step = 0, 0
        self.__mu__ = self.__logstd__ = self.__kl__ = None

    def reparametrize(self, mu, logstd, eps=None):
        """z = mu + eps * exp(min(logstd, MAX_LOGSTD)) in training, mu in eval; eps=None draws Philox noise
        keyed by (self.seed, step).  Also caches the fused KL for kl_loss()."""
        seed = (self.seed * 0x9E3779B1 + self._step) & (2**63 - 1)
        z, kl = _ReparamKL.apply(mu, logstd, eps, seed, self.training, False)
        self.__kl__ = kl
        return z

    def encode(self, *args, eps=None, **kwargs):
        self.__mu__, self.__logstd__ = self.encoder(*args, **kwargs)
        z = self.reparametrize(self.__mu__, self.__logstd__, eps=eps)
        self._step += 1
        return z

    def kl_loss(self, mu=None, logstd=None):
        if mu is None and logstd is None and self.__kl__ is not None:
            return self.__kl__
        mu = self.__mu__ if mu is None else mu
        logstd = self.__logstd__ if logstd is None else logstd
        _, kl = _ReparamKL.apply(mu, logstd, None, 0, False, False)
        return kl

    def recon_loss(self, z, pos_edge_index, neg_edge_index=None, graph=None):
        N = z.shape[0]
        pos_pair = graph.dec if isinstance(graph, EventGraph) else None
        E = pos_edge_index.shape[1]
        ps, pc = _EdgeBCE.apply(z, pos_edge_index, False, pos_pair)
        if neg_edge_index is None:
            neg_edge_index = negative_sampling(pos_edge_index, N, seed=self.seed + self._step, graph=graph)
        ns, nc = _EdgeBCE.apply(z, neg_edge_index, True, None)
        return ps / pc.clamp(min=1.0) + ns / nc.clamp(min=1.0)


class VGAEVQ(nn.Module):
    """GCN-VGAE + VQ bottleneck: the north-star training step.

        mu, logstd = encoder(x, graph);  z = reparametrize;  kl = kl_loss
        vq_loss, z_q, perplexity, _, idx = vector_quantization(z)      (scripts/models.py:32)
        rec = recon_loss(z_q) over the graph's edges and 1:1 permutation-structured negatives
        loss = _loss_sum(rec, kl, vq_loss, N)                                   (loss sum as scripts/run.py:95)
    """

    def __init__(self, in_channels, latent, num_latent_embeddings=512, beta=0.25, seed=0, negatives="sigma", compute_dtype=torch.float32):
        """negatives: "sigma" = permutation-structured negatives (col[p], sigma(r)) drawn inside the fused decoder (no per-step
        sort; colliding pairs masked) - the fast path; "uniform" = the literal recon_loss path: E pairs from
        negative_sampling (ceeg_neg_sample_uniform, rejection against the edge set) decoded through the edge-list kernel,
        their per-step CSR built by a device sort for the atomics-free backward."""
        super().__init__()
        if negatives not in ("sigma", "uniform"):
            raise ValueError("negatives must be 'sigma' or 'uniform'")
        self.encoder = VariationalGCNEncoder(in_channels, latent, compute_dtype)
        self.vector_quantization = VectorQuantizer(num_latent_embeddings, latent, beta, return_one_hot=False)
        self.latent, self.seed, self.negatives = latent, int(seed), negatives

    def negative_seed(self, step):
        return (self.seed * 0x9E3779B1 + 0x5851F42D * (int(step) + 1)) & (2**63 - 1)

    def forward(self, x, graph, step=0, eps=None, sigma=None, neg_edge_index=None):
        N = x.shape[0]
        D = self.latent
        seed = (self.seed * 0x9E3779B1 + step) & (2**63 - 1)
        ml, z, kl, vq_loss, z_q, ppl, idx = self.encoder.encode_reparam_vq(x, graph, eps, seed, self.training, self.vector_quantization)
        if self.negatives == "uniform" or neg_edge_index is not None:
            # literal path: recon_loss(z_q, pos_edge_index, negative_sampling(...)) - one fresh sample set per step
            W, code = self.vector_quantization.embedding.weight, idx.view(-1).to(torch.int32)
            codes_on = os.environ.get("CEEG_DEC_CODES", "1") == "1" and os.environ.get("CEEG_PAIR_CODES", "1") == "1"
            n_neg = graph.num_edges if neg_edge_index is None else neg_edge_index.shape[1]
            if codes_on and graph.num_edges > 0 and N > 1 and _UniformRecon.supported(W, N, n_neg):
                # both pair sets in code form, one dz; with no list given the pairs are drawn inside the forward kernel
                rec, sums, neg_edge_index = _UniformRecon.apply(z_q, graph, neg_edge_index, W, code, self.negative_seed(step), n_neg)
                loss = _loss_sum(rec, kl, vq_loss, N)
                return loss, dict(rec=rec, kl=kl, vq=vq_loss, ppl=ppl, idx=idx, z=z, ml=ml, dec_sums=sums, neg_edge_index=neg_edge_index)
            if neg_edge_index is None:
                neg_edge_index = negative_sampling_raw(graph, N, graph.num_edges, self.negative_seed(step))
            if (os.environ.get("CEEG_DEC_CODES", "1") == "1" and graph.num_edges > 0
                    and _lib.lib().ceeg_decoder_codes_supported(int(W.shape[0]), int(W.shape[1])) == 1):
                # positives: the CSR-ordered code-form decoder with its negatives switched off (sig_a = 0)
                pos, psums = _FusedDecoder.apply(z_q, graph, 0, 0, 1, True, None, W, code)
                ps = psums[0]
            else:
                ps, pc = _EdgeBCE.apply(z_q, graph.edge_index_i32(), False, graph.dec, W, code)
                pos = ps / pc.clamp(min=1.0)
            if os.environ.get("CEEG_PAIR_CODES", "1") == "1" and _PairBCECodes.supported(W, N, neg_edge_index.shape[1]):
                ns, nc = _PairBCECodes.apply(z_q, neg_edge_index, True, W, code)   # no CSR of the sampled list: one sort of incidences
            else:
                ns, nc = _EdgeBCE.apply(z_q, neg_edge_index, True, None, W, code)
            rec = pos + ns / nc.clamp(min=1.0)
            sums = torch.stack([ps.detach(), ns.detach(), nc])
        else:
            a, b, ainv = sigma_params(N, self.seed, step) if sigma is None else sigma
            # z_q is bit-for-bit codebook[idx] in fp32: let the decoder gather 512 codebook rows through idx
            rec, sums = _FusedDecoder.apply(z_q, graph, a, b, ainv, True, None, self.vector_quantization.embedding.weight,
                                            idx.view(-1).to(torch.int32))
        loss = _loss_sum(rec, kl, vq_loss, N)
        return loss, dict(rec=rec, kl=kl, vq=vq_loss, ppl=ppl, idx=idx, z=z, ml=ml, dec_sums=sums, neg_edge_index=neg_edge_index)


class ConditionalVGAEVQ(nn.Module):
    """Conditional GCN-VGAE + VQ on a PyG-style collated batch of graphs (BASELINE config 4).

    The per-graph condition c [G, C] is broadcast to the nodes as c[batch] and concatenated to the encoder input
    (scripts/models.py:162: `cat((x, condition))`) and to the latent before decoding (scripts/models.py:259:
    `cat((latent, condition))`); `batch` is the collated graph-id vector (scripts/run.py:310-311,
    scripts/models.py:121-122).  Everything else is the VGAEVQ step.  With `nodes_per_graph` (equal-sized graphs)
    the permutation-structured negatives stay inside their graph; without it sigma permutes all nodes of the batch.
    in_channels + cond_channels and latent + cond_channels must be multiples of 4 (16-byte gathers), the latter <= 256."""

    def __init__(self, in_channels, cond_channels, latent, num_latent_embeddings=512, beta=0.25, seed=0):
        super().__init__()
        if (in_channels + cond_channels) % 4 or (latent + cond_channels) % 4 or latent + cond_channels > 256:
            raise ValueError("in_channels + cond_channels and latent + cond_channels must be multiples of 4 (latent + cond <= 256)")
        self.encoder = VariationalGCNEncoder(in_channels + cond_channels, latent)
        self.vector_quantization = VectorQuantizer(num_latent_embeddings, latent, beta, return_one_hot=False)
        self.latent, self.cond_channels, self.seed = latent, cond_channels, int(seed)

    @staticmethod
    def _concat_condition(x, condition, batch, cn):
        """cat((x, c[batch]), dim=1): one pass (ceeg_concat_condition) when nothing in it carries a gradient, else torch.cat."""
        F, C = x.shape[1], condition.shape[1]
        if (x.is_cuda and not x.requires_grad and not condition.requires_grad and x.dtype == torch.float32 and condition.dtype == torch.float32
                and batch.dtype == torch.int64 and F % 4 == 0 and C % 4 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0):
            cond = condition.contiguous()
            out = torch.empty(x.shape[0], F + C, dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().ceeg_concat_condition(x.data_ptr(), x.stride(0), x.shape[0], F, cond.data_ptr(), C, cond.shape[0],
                                                        batch.contiguous().data_ptr(), out.data_ptr(), _lib.stream()), "ceeg_concat_condition")
            return out
        return torch.cat((x, condition.index_select(0, batch) if cn is None else cn), dim=1)

    @staticmethod
    def _intra_graph(graph, batch, mod):
        """True when `batch` is the block layout of equal-sized graphs (node n in graph n // mod) and no edge crosses
        graphs - checked once per graph object (graph-construction work, one host sync)."""
        pre = getattr(graph, "checks", {}).get(("intra", int(mod)))   # a collation already checked it on the device
        if pre is not None and getattr(graph, "batch", None) is batch:
            return pre
        key = (batch.data_ptr(), int(mod))
        cache = getattr(graph, "_intra", None)
        if cache is None or cache[0] != key:
            n = batch.shape[0]
            blocks = torch.arange(n, device=batch.device) // int(mod)
            ei = graph.edge_index
            ok = bool(torch.equal(batch.to(torch.int64), blocks)) and bool((blocks[ei[0]] == blocks[ei[1]]).all())
            graph._intra = (key, ok)
            cache = graph._intra
        return cache[1]

    def forward(self, x, graph, batch, condition, step=0, eps=None, sigma=None, nodes_per_graph=None):
        N = x.shape[0]
        if batch.shape[0] != N or condition.shape[1] != self.cond_channels:
            raise ValueError("batch must hold one graph id per node and condition must be [num_graphs, cond_channels]")
        # c[batch] is only materialised where a gradient has to reach the condition (or a kernel wants the rows)
        cn = condition.index_select(0, batch) if condition.requires_grad else None
        seed = (self.seed * 0x9E3779B1 + step) & (2**63 - 1)
        xc = self._concat_condition(x, condition, batch, cn)
        ml, z, kl, vq_loss, z_q, ppl, idx = self.encoder.encode_reparam_vq(xc, graph, eps, seed, self.training, self.vector_quantization)
        mod = N if nodes_per_graph is None else int(nodes_per_graph)
        if N % mod:
            raise ValueError("nodes_per_graph must divide the number of nodes of the batch")
        a, b, ainv = sigma_params(mod, self.seed, step) if sigma is None else sigma
        if nodes_per_graph is not None and self._intra_graph(graph, batch, mod):
            # every positive and negative pair joins two nodes of ONE graph: the condition's share of a dot product is
            # |c_graph|^2, so the decoder runs in code form on z_q with cn as constant extra columns
            rec, sums = _FusedDecoder.apply(z_q, graph, a, b, ainv, True, mod, self.vector_quantization.embedding.weight,
                                            idx.view(-1).to(torch.int32), (condition, batch) if cn is None else cn)
        else:
            cn = condition.index_select(0, batch) if cn is None else cn
            rec, sums = _FusedDecoder.apply(torch.cat((z_q, cn), dim=1), graph, a, b, ainv, True, mod)
        loss = _loss_sum(rec, kl, vq_loss, N)
        return loss, dict(rec=rec, kl=kl, vq=vq_loss, ppl=ppl, idx=idx, z=z, ml=ml, dec_sums=sums)


# ==================================================================================================
# GATConv (the graph layer the reference actually instantiates: scripts/models.py:106)
# ==================================================================================================
_SPMM_MAX_F = 1024  # widest feature slice of one ceeg_spmm_csr call (include/ceeg.h)


class _GATAggregate(torch.autograd.Function):
    """out[i, h] = sum_j softmax_j(leaky_relu(a_src[j,h] + a_dst[i,h])) * xp[j, h, :] over the CSR row of i (self-loops in),
    with a_src[n,h] = <xp[n,h,:], att_src[h,:]>, a_dst likewise (PyG GATConv: (x_proj * att).sum(-1)).
    Scores: ceeg_gat_scores_fwd (one pass over xp for both vectors and every head); softmax weights: ceeg_gat_softmax_fwd;
    aggregation and its transpose: ceeg_spmm_csr, one pass per head over column slices (no copies); attention backward:
    ceeg_gat_attn_bwd + ceeg_segment_sum_perm_items; the scores' backward lands in the same dxp (ceeg_gat_scores_bwd)."""

    @staticmethod
    def forward(ctx, xp, att_src, att_dst, bias, pair, heads, slope):
        """bias: None or the [heads * C] output bias of the concatenating layer, added in the aggregation's epilogue."""
        _lib.require_cuda(xp)
        L = _lib.lib()
        xp = xp.contiguous()
        N, HC = xp.shape
        C = HC // heads
        att_s, att_d = att_src.detach().reshape(-1).contiguous(), att_dst.detach().reshape(-1).contiguous()
        a_src = torch.empty(heads, N, dtype=torch.float32, device=xp.device)
        a_dst = torch.empty(heads, N, dtype=torch.float32, device=xp.device)
        _lib.check(L.ceeg_gat_scores_fwd(xp.data_ptr(), xp.stride(0), att_s.data_ptr(), att_d.data_ptr(), N, heads, C, a_src.data_ptr(),
                                         a_dst.data_ptr(), _lib.stream()), "ceeg_gat_scores_fwd")
        w = torch.empty(heads, max(pair.nnz, 1), dtype=torch.float32, device=xp.device)
        out = torch.empty_like(xp)
        li = pair.long_items(False)
        bias_v = None if bias is None else bias.detach().contiguous()
        ctx.has_bias = bias is not None
        for h in range(heads):
            _lib.check(L.ceeg_gat_softmax_fwd(pair.rowptr.data_ptr(), pair.col.data_ptr(), a_src[h].data_ptr(), a_dst[h].data_ptr(),
                                              N, pair.nnz, float(slope), w[h].data_ptr(), _lib.stream()), "ceeg_gat_softmax_fwd")
            for c0 in range(h * C, (h + 1) * C, _SPMM_MAX_F):  # a head wider than one SpMM call: column slices of it
                c1 = min(c0 + _SPMM_MAX_F, (h + 1) * C)
                _spmm(pair.rowptr, pair.col, w[h], None, xp[:, c0:c1], N, bias=None if bias_v is None else bias_v[c0:c1],
                      out=out[:, c0:c1], max_row=pair.max_row, items=li)
        ctx.save_for_backward(xp, a_src, a_dst, w, att_s, att_d)
        ctx.pair, ctx.heads, ctx.slope = pair, heads, float(slope)
        ctx.att_shape = att_src.shape
        return out

    @staticmethod
    def backward(ctx, g):
        xp, a_src, a_dst, w, att_s, att_d = ctx.saved_tensors
        pair, heads, slope = ctx.pair, ctx.heads, ctx.slope
        L = _lib.lib()
        g = g.contiguous()
        N, HC = xp.shape
        C = HC // heads
        dxp = torch.empty_like(xp)
        d_asrc = torch.empty(heads, N, dtype=torch.float32, device=xp.device)
        d_adst = torch.empty(heads, N, dtype=torch.float32, device=xp.device)
        dl = torch.empty(max(pair.nnz, 1), dtype=torch.float32, device=xp.device)
        lt = pair.long_items(True)  # hub sources: their scalar segment sums go by pieces
        n_lt = 0 if lt is None else lt.shape[0]
        chunk = int(L.ceeg_decoder_codes_chunk())
        ws_seg = _lib.workspace(L.ceeg_segment_sum_items_ws_bytes(n_lt), xp.device) if n_lt else None
        for h in range(heads):
            gs, xs = g[:, h * C:(h + 1) * C], xp[:, h * C:(h + 1) * C]
            for c0 in range(h * C, (h + 1) * C, _SPMM_MAX_F):
                c1 = min(c0 + _SPMM_MAX_F, (h + 1) * C)
                _spmm(pair.t_rowptr, pair.t_col, w[h], pair.t_perm, g[:, c0:c1], N, out=dxp[:, c0:c1], max_row=pair.max_trow, items=lt)
            _lib.check(L.ceeg_gat_attn_bwd(pair.rowptr.data_ptr(), pair.col.data_ptr(), w[h].data_ptr(), a_src[h].data_ptr(),
                                           a_dst[h].data_ptr(), gs.data_ptr(), gs.stride(0), xs.data_ptr(), xs.stride(0), N, pair.nnz, C,
                                           slope, dl.data_ptr(), d_adst[h].data_ptr(), _lib.stream()), "ceeg_gat_attn_bwd")
            _lib.check(L.ceeg_segment_sum_perm_items(pair.t_rowptr.data_ptr(), pair.t_perm.data_ptr(), dl.data_ptr(), N, _lib.ptr(lt), n_lt,
                                                     chunk, d_asrc[h].data_ptr(), _lib.ptr(ws_seg), 0 if ws_seg is None else ws_seg.numel(),
                                                     _lib.stream()), "ceeg_segment_sum_perm_items")
        d_att = torch.empty(2, HC, dtype=torch.float32, device=xp.device)
        ws = _lib.workspace(L.ceeg_gat_scores_bwd_ws_bytes(N, heads, C), xp.device)
        _lib.check(L.ceeg_gat_scores_bwd(xp.data_ptr(), xp.stride(0), att_s.data_ptr(), att_d.data_ptr(), d_asrc.data_ptr(), d_adst.data_ptr(),
                                         N, heads, C, dxp.data_ptr(), dxp.stride(0), d_att[0].data_ptr(), d_att[1].data_ptr(), ws.data_ptr(),
                                         ws.numel(), _lib.stream()), "ceeg_gat_scores_bwd")
        d_bias = torch.cat([_colsum(g[:, c0:c0 + _SPMM_MAX_F]) for c0 in range(0, HC, _SPMM_MAX_F)]) if ctx.has_bias else None
        return dxp, d_att[0].view(ctx.att_shape), d_att[1].view(ctx.att_shape), d_bias, None, None, None


class GATConv(nn.Module):
    """PyG-style GATConv(in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, add_self_loops=True,
    bias=True) for one node set: x' = x W^T;  alpha_ij = softmax_j(leaky_relu(<x'_j, att_src> + <x'_i, att_dst>));
    out_i = sum_j alpha_ij x'_j (+ bias).  State-dict keys `lin_src.weight` (shared source/target transform),
    `att_src`, `att_dst` [1, heads, out], `bias`.  out_channels must be a multiple of 4 and <= 2048 per head (the reference's
    conditioner layers are GATConv(768, 2048) and GATConv(2048, 2048), scripts/models.py:105-110, scripts/run.py:295)."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, add_self_loops=True, bias=True):
        super().__init__()
        if out_channels % 4 or out_channels > 2048 or heads * out_channels > 16384:
            raise ValueError("out_channels must be a multiple of 4 and <= 2048 (heads * out_channels <= 16384)")
        if not add_self_loops:
            raise NotImplementedError("the CSR built for attention always carries one self-loop per node")
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, int(heads), bool(concat)
        self.negative_slope = float(negative_slope)
        self.lin_src = _Lin(in_channels, heads * out_channels)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        glorot_(self.att_src.data.view(heads, out_channels))
        glorot_(self.att_dst.data.view(heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels)) if bias else None

    @staticmethod
    def attention_graph(edge_index, num_nodes):
        """CSR pair for attention: self-loops of the input removed, one fresh self-loop per node added (PyG)."""
        keep = edge_index[0] != edge_index[1]
        return CsrPair(edge_index[:, keep], num_nodes, self_loops=True)

    def forward(self, x, graph):
        N = x.shape[0]
        pair = graph if isinstance(graph, CsrPair) else self.attention_graph(graph.edge_index if isinstance(graph, EventGraph) else graph, N)
        H, C = self.heads, self.out_channels
        xp = _Linear.apply(x, self.lin_src.weight, None, False)                     # [N, H*C]
        if self.concat:  # the bias rides in the aggregation's epilogue
            return _GATAggregate.apply(xp, self.att_src, self.att_dst, self.bias, pair, H, self.negative_slope)
        out = _GATAggregate.apply(xp, self.att_src, self.att_dst, None, pair, H, self.negative_slope)
        out = out.view(N, H, C).mean(dim=1)
        return out if self.bias is None else out + self.bias
