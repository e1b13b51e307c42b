"""Event-graph ingestion: the on-disk CSV pair of the reference -> `edge_index` (+ `entities` weights), and the
ancestor-subgraph extraction its dataset builder runs per target node.

Follows /root/reference/scripts/dataset.py:
  * :122-153  nodes `dropna()` + `drop_duplicates(subset='title')`; edges whose endpoints were dropped are skipped;
              parallel (old_id, new_id) rows collapse into one directed edge whose `entities` counts them;
  * :22-33    `get_top_n_valid_predecessors`: predecessors with entities >= MIN_COMMON_ENTITIES, heaviest first
              (stable: ties keep the order in which the edges first appeared), at most n;
  * :35-46    `get_n_gen_ancestors`: the union of those predecessors over NUM_GENERATIONS generations.
Host-side preprocessing (runs once per dataset, numpy/pandas on the CPU like the reference's); the tensors it
returns feed `EventGraph(edge_index, num_nodes, edge_weight=entities)`.  Nodes are numbered 0..n-1 in the order
of the cleaned node table (the reference labels them with the table's index; `node_ids[i]` is the CSV `id`).
"""
import numpy as np

MAX_TOP_PREDECESSORS = 3   # dataset.py:16
NUM_GENERATIONS = 3        # dataset.py:17
MIN_COMMON_ENTITIES = 3    # dataset.py:18


def load_event_graph_csv(nodes_csv, edges_csv):
    """-> dict(edge_index [2,E] int64 (row 0 = old_id = source, row 1 = new_id = target), entities [E] int64,
    num_nodes, node_ids [n] (CSV ids), titles [n])."""
    import pandas as pd
    nodes = pd.read_csv(nodes_csv).dropna().drop_duplicates(subset="title")   # dataset.py:126-133
    edges = pd.read_csv(edges_csv)
    ids = nodes["id"].to_numpy()
    # CSV id -> position in the cleaned node table (-1: the node row was dropped), vectorised (dataset.py:145-156 loops in Python)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]

    def position(v):
        # a node id that appears on several (distinct-title) rows maps to its LAST row, as the reference's dict does
        # (dataset.py:137: node_mapping[id] = idx overwrites): stable sort + the right-most equal entry
        v = np.asarray(v)
        at = np.clip(np.searchsorted(sid, v, side="right") - 1, 0, max(len(sid) - 1, 0))
        ok = (sid[at] == v) if len(sid) else np.zeros(len(v), dtype=bool)
        return np.where(ok, order[at] if len(sid) else 0, -1)

    o, n = position(edges["old_id"].to_numpy()), position(edges["new_id"].to_numpy())
    keep = (o >= 0) & (n >= 0)
    ei, cnt = collapse_multigraph(o[keep], n[keep], len(ids))
    return dict(edge_index=ei, entities=cnt, num_nodes=len(ids), node_ids=ids.astype(np.int64), titles=nodes["title"].tolist())


def collapse_multigraph(src, dst, num_nodes):
    """Parallel (src, dst) rows -> one directed edge each, `entities` = how many rows it stood for, edges in the order of their
    FIRST appearance (networkx's insertion order, which `predecessors` iterates: dataset.py:147-153).  numpy arrays in, numpy
    out; torch CUDA tensors in, the same on the device (sort by pair, run lengths, re-order by first row - no host loop)."""
    if hasattr(src, "is_cuda"):
        import torch
        key = src.to(torch.int64) * int(num_nodes) + dst.to(torch.int64)
        uniq, inv, counts = torch.unique(key, return_inverse=True, return_counts=True)
        first = torch.full((uniq.numel(),), key.numel(), dtype=torch.int64, device=key.device)
        first.scatter_reduce_(0, inv, torch.arange(key.numel(), device=key.device), reduce="amin")
        by_first = torch.argsort(first)
        uniq, counts = uniq[by_first], counts[by_first]
        return torch.stack([uniq // int(num_nodes), uniq % int(num_nodes)]), counts
    src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
    key = src * int(num_nodes) + dst
    uniq, first, counts = np.unique(key, return_index=True, return_counts=True)
    by_first = np.argsort(first, kind="stable")
    uniq, counts = uniq[by_first], counts[by_first].astype(np.int64)
    return np.stack([uniq // int(num_nodes), uniq % int(num_nodes)]).astype(np.int64).reshape(2, -1), counts


class PredecessorIndex:
    """In-edges grouped by target, in first-appearance order (what networkx's `predecessors` iterates)."""

    def __init__(self, edge_index, entities, num_nodes):
        ei = np.asarray(edge_index)
        order = np.argsort(ei[1], kind="stable")
        self.src = ei[0][order]
        self.w = np.asarray(entities)[order]
        self.ptr = np.zeros(num_nodes + 1, dtype=np.int64)
        np.add.at(self.ptr, ei[1] + 1, 1)
        self.ptr = np.cumsum(self.ptr)

    def top_predecessors(self, node, n, min_entities=MIN_COMMON_ENTITIES):
        lo, hi = self.ptr[node], self.ptr[node + 1]
        s, w = self.src[lo:hi], self.w[lo:hi]
        keep = w >= min_entities
        s, w = s[keep], w[keep]
        order = np.argsort(-w, kind="stable")          # sorted(..., reverse=True) is stable
        return s[order][: min(n, hi - lo)].tolist()

    def ancestors(self, node, num_gens=NUM_GENERATIONS, num_predecessors=MAX_TOP_PREDECESSORS):
        """dataset.py:35-46 without the recursion: generation-by-generation frontier expansion."""
        if num_gens == 0:
            return {int(node)}
        out, frontier = set(), [int(node)]
        for g in range(num_gens):
            nxt = []
            for v in frontier:
                for p in self.top_predecessors(v, num_predecessors):
                    out.add(int(p))
                    nxt.append(int(p))
            if g == num_gens - 1:   # the reference's base case adds the last generation's nodes themselves
                out.update(nxt)
            frontier = nxt
        return out


def ancestor_subgraph(edge_index, entities, num_nodes, node, index=None):
    """Induced subgraph on the ancestors of `node` (dataset.py:201-225): (local edge_index [2,e], node list)."""
    idx = PredecessorIndex(edge_index, entities, num_nodes) if index is None else index
    keep = sorted(idx.ancestors(node))
    loc = {v: i for i, v in enumerate(keep)}
    ei = np.asarray(edge_index)
    m = np.isin(ei[0], keep) & np.isin(ei[1], keep)
    sub = np.array([[loc[int(a)] for a in ei[0][m]], [loc[int(b)] for b in ei[1][m]]], dtype=np.int64).reshape(2, -1)
    return sub, keep


class DeviceAncestors:
    """The ancestor extraction for EVERY target node at once on the GPU (ceeg_top_predecessors / ceeg_ancestors /
    ceeg_ancestor_subgraph_*): what `PredecessorIndex.ancestors` and `ancestor_subgraph` compute one node at a time.

        da = DeviceAncestors(edge_index, entities, num_nodes)            # device tensors; edges in first-appearance order
        da.anc [T, cap] int32 (sorted, -1 padded), da.count [T]
        sub_edge_index, edge_ptr = da.subgraphs()                        # local ids, edges of target t at [edge_ptr[t], edge_ptr[t+1])
    """

    def __init__(self, edge_index, entities, num_nodes, targets=None, num_gens=NUM_GENERATIONS, num_predecessors=MAX_TOP_PREDECESSORS,
                 min_entities=MIN_COMMON_ENTITIES):
        import torch
        from . import _lib
        from .models import CsrPair
        _lib.require_cuda(edge_index, entities)
        L = _lib.lib()
        dev = edge_index.device
        N, n = int(num_nodes), int(num_predecessors)
        self.pair = CsrPair(edge_index, N, self_loops=False)
        ent = entities.to(torch.int64).contiguous()
        if ent.numel() == 0:
            ent = torch.zeros(1, dtype=torch.int64, device=dev)
        col = self.pair.col if self.pair.nnz else torch.zeros(1, dtype=torch.int32, device=dev)
        perm = self.pair.perm if self.pair.nnz else torch.zeros(1, dtype=torch.int32, device=dev)
        self._col = col
        self.top = torch.empty(N, n, dtype=torch.int32, device=dev)
        _lib.check(L.ceeg_top_predecessors(self.pair.rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(), ent.data_ptr(), N, n, int(min_entities),
                                           self.top.data_ptr(), _lib.stream()), "ceeg_top_predecessors")
        cap = int(L.ceeg_ancestors_cap(n, int(num_gens)))
        if cap < 0:
            raise ValueError("num_predecessors + ... + num_predecessors^num_gens exceeds what the kernel holds (128)")
        self.cap = max(cap, 1)
        tg = None if targets is None else targets.to(torch.int32).contiguous()
        self.T = N if tg is None else int(tg.shape[0])
        self.anc = torch.empty(self.T, self.cap, dtype=torch.int32, device=dev)
        self.count = torch.empty(self.T, dtype=torch.int32, device=dev)
        _lib.check(L.ceeg_ancestors(self.top.data_ptr(), N, n, _lib.ptr(tg), self.T, int(num_gens), self.cap, self.anc.data_ptr(),
                                    self.count.data_ptr(), _lib.stream()), "ceeg_ancestors")

    def subgraphs(self):
        """(sub_edge_index [2, total] int32 with LOCAL ids, edge_ptr [T+1] int64): the induced subgraph of every target's
        ancestors, edges ordered by (target slot, source id) inside each subgraph."""
        import torch
        from . import _lib
        L = _lib.lib()
        dev = self.anc.device
        slots = torch.empty(self.T * self.cap, dtype=torch.int64, device=dev)
        args = (self.pair.rowptr.data_ptr(), self._col.data_ptr(), self.anc.data_ptr(), self.count.data_ptr(), self.T, self.cap)
        _lib.check(L.ceeg_ancestor_subgraph_count(*args, slots.data_ptr(), _lib.stream()), "ceeg_ancestor_subgraph_count")
        incl = slots.cumsum(0)
        off = (incl - slots).contiguous()
        total = int(incl[-1].item()) if slots.numel() else 0         # dataset-construction time: one host sync
        sub = torch.empty(2, max(total, 1), dtype=torch.int32, device=dev)[:, :total]
        if total:
            _lib.check(L.ceeg_ancestor_subgraph_fill(*args, off.data_ptr(), sub[0].data_ptr(), sub[1].data_ptr(), _lib.stream()),
                       "ceeg_ancestor_subgraph_fill")
        edge_ptr = torch.cat([off.view(self.T, self.cap)[:, 0], incl[-1:]]) if slots.numel() else torch.zeros(1, dtype=torch.int64, device=dev)
        return sub, edge_ptr
