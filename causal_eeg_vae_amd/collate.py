"""PyG-style collation of disjoint graphs on the device.

The reference batches its samples with `torch_geometric.loader.DataLoader(dataset, batch_size, follow_batch=[...])`
(scripts/run.py:310-326): node-level tensors concatenated, `edge_index` offset block-diagonally by the cumulative node
counts, a `batch` vector with the graph id of every node, one `<key>_batch` vector per followed key, and reads single samples
back out of the collated storage with `torch_geometric.data.separate` (scripts/dataset.py:66-92).  Here the per-element work
(offsetting, graph ids) is two HIP kernels over the already concatenated tensors (ceeg_collate_edges, ceeg_segment_ids);
only the list of per-graph sizes is host data, as it is in PyG's `slice_dict`.

`MinibatchStream` is the loader loop around it: minibatch k+1 is collated and its graph structures are built on a side HIP
stream while the caller's optimizer step k runs on the current one.
"""
import torch

from . import _lib


def _ptr(counts, device):
    p = torch.zeros(len(counts) + 1, dtype=torch.int64)
    if len(counts):
        p[1:] = torch.as_tensor(list(counts), dtype=torch.int64).cumsum(0)
    return p.to(device)


def segment_ids(ptr, total):
    """[total] int64: the segment (graph) of every element, from ascending offsets ptr [G+1]."""
    out = torch.empty(max(int(total), 1), dtype=torch.int64, device=ptr.device)[: int(total)]
    _lib.check(_lib.lib().ceeg_segment_ids(ptr.data_ptr(), ptr.shape[0] - 1, int(total), out.data_ptr(), _lib.stream()), "ceeg_segment_ids")
    return out


class CollatedBatch:
    """The collated form of a list of graphs: `edge_index` [2, sum E] (offset), `batch` [sum N], `ptr` / `edge_ptr` [G+1]
    (PyG's slice_dict for nodes and edges), `num_nodes`, `num_graphs`, and per followed key `<key>` (concatenated rows),
    `<key>_batch` and `<key>_ptr`."""

    def __init__(self, edge_indices, nodes_per_graph, follow=None):
        if len(edge_indices) != len(nodes_per_graph) or not len(edge_indices):
            raise ValueError("one edge_index and one node count per graph")
        dev = edge_indices[0].device
        _lib.require_cuda(*edge_indices)
        self.num_graphs = G = len(edge_indices)
        self.nodes_per_graph = [int(n) for n in nodes_per_graph]
        # both slice tables in ONE pinned buffer and one asynchronous copy (two pageable copies cost the loader loop ~0.1 ms of idle GPU)
        host = torch.zeros(2, G + 1, dtype=torch.int64, pin_memory=True)
        host[0, 1:] = torch.as_tensor(self.nodes_per_graph, dtype=torch.int64).cumsum(0)
        host[1, 1:] = torch.as_tensor([int(e.shape[1]) for e in edge_indices], dtype=torch.int64).cumsum(0)
        both = host.to(dev, non_blocking=True)
        self.ptr, self.edge_ptr = both[0], both[1]
        self.num_nodes = int(sum(int(n) for n in nodes_per_graph))
        raw = torch.cat([e.to(torch.int64) for e in edge_indices], dim=1).contiguous()   # one concatenation, no per-graph arithmetic
        E = raw.shape[1]
        self.edge_index = torch.empty_like(raw)
        self.edge_batch = torch.empty(max(E, 1), dtype=torch.int64, device=dev)[:E]
        # `outside` (device int32): 1 if some edge does not stay inside its graph's block of node ids - written by the same kernel
        self.outside = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().ceeg_collate_edges_checked(raw.data_ptr(), raw.stride(0), self.edge_ptr.data_ptr(), self.ptr.data_ptr(), G, E,
                                                         self.edge_index.data_ptr(), self.edge_index.stride(0), self.edge_batch.data_ptr(),
                                                         self.outside.data_ptr(), _lib.stream()), "ceeg_collate_edges_checked")
        self.batch = segment_ids(self.ptr, self.num_nodes)
        self.followed = {}
        for key, parts in (follow or {}).items():          # follow_batch=[key, ...]: rows of `key` belong to their sample's graph
            if len(parts) != G:
                raise ValueError(f"followed key {key!r}: one tensor per graph")
            kp = _ptr([int(t.shape[0]) for t in parts], dev)
            val = torch.cat(list(parts), dim=0)
            self.followed[key] = (val, segment_ids(kp, val.shape[0]), kp)
            setattr(self, key, val)
            setattr(self, key + "_batch", self.followed[key][1])
            setattr(self, key + "_ptr", kp)

    def event_graph(self, edge_weight=None, with_dec=True, defer=False):
        """The collated batch as an EventGraph, built with ONE host synchronisation (models.EventGraph: both CSR pairs, their
        piece lists and the layout check share one record).  For equal-sized graphs the check "no edge leaves its graph" is
        evaluated on the device and lands in graph.checks[("intra", nodes_per_graph)], which ConditionalVGAEVQ consults instead
        of testing (and synchronising) itself; `batch` is the block layout by construction."""
        from .models import EventGraph
        checks = None
        sizes = set(self.nodes_per_graph)
        if len(sizes) == 1 and self.nodes_per_graph[0] > 0:
            n = int(self.nodes_per_graph[0])
            checks = {("intra", n): self.outside[0] == 0}
        # every collated edge lies inside its graph's block of node ids by construction (local ids outside [0, n_graph) raise
        # `outside` AND fail the device-side check of the block-keyed build, which then falls back to the general one)
        blocks = (self.batch, self.ptr, max(self.nodes_per_graph)) if with_dec else None
        g = EventGraph(self.edge_index, self.num_nodes, edge_weight, with_dec=with_dec, checks=checks, defer=True, blocks=blocks)
        g.batch = self.batch
        return g if defer else self.finish_graph(g)

    @staticmethod
    def finish_graph(g):
        """Second half of event_graph(defer=True): the one host read + gcn_norm (models.EventGraph.finish)."""
        g.finish()
        g.reuse = False          # a minibatch graph lives for one step: no per-graph packing of the value stream
        return g

    def separate(self, idx):
        """The idx-th graph back out of the collated storage (torch_geometric.data.separate): its edge_index with the node
        offset subtracted again, its node count, and its slice of every followed key."""
        i = int(idx)
        if not 0 <= i < self.num_graphs:
            raise IndexError(i)
        lo, hi = int(self.edge_ptr[i]), int(self.edge_ptr[i + 1])     # slice_dict lookups: host values, as in PyG
        n_lo, n_hi = int(self.ptr[i]), int(self.ptr[i + 1])
        sl = self.edge_index[:, lo:hi]
        out = torch.empty(2, hi - lo, dtype=torch.int64, device=sl.device)
        if hi > lo:
            one_e = torch.tensor([0, hi - lo], dtype=torch.int64, device=sl.device)
            one_n = torch.tensor([n_lo, n_hi], dtype=torch.int64, device=sl.device)
            _lib.check(_lib.lib().ceeg_collate_edges(sl.data_ptr(), sl.stride(0), one_e.data_ptr(), one_n.data_ptr(), 1, hi - lo, -1,
                                                     out.data_ptr(), out.stride(0), None, _lib.stream()), "ceeg_collate_edges")
        parts = {}
        for key, (val, _, kp) in self.followed.items():
            parts[key] = val[int(kp[i]): int(kp[i + 1])]
        return out, n_hi - n_lo, parts


class MinibatchStream:
    """for collated, graph in MinibatchStream(minibatches): ... - the loop `for batch in train_dataloader`
    (scripts/run.py:87-98,310) for minibatches whose per-graph edge lists are resident on the device.

    `minibatches` yields (edge_indices, nodes_per_graph) or (edge_indices, nodes_per_graph, follow) (+ an optional 4th item,
    the collated edge weights).  prefetch=False is the plain loop on the current stream.  With prefetch=True (the default) the
    construction runs on a SIDE stream, `depth` minibatches ahead
    of the caller's optimizer steps on the current stream, in two halves: `begin` (collation, both CSR pairs, transposes, piece
    lists, layout check, the record's asynchronous copy to pinned host memory - no host wait) when a minibatch enters the
    window, `finish` (wait for the record, gcn_norm) right before it is handed out - by then the record has had a whole step to
    land, so the host does not block.  Ordering rules the iterator enforces:
      * the current stream waits for a minibatch's `finish` before the minibatch is handed out;
      * a `begin` waits for the work the caller had queued on the current stream when it asked for the minibatch BEFORE the
        one it is asking for now: the structures the caching allocator may hand to that `begin` (those of minibatches the
        caller has dropped) are then no longer being read;
      * the iterator keeps the minibatch it handed out last referenced until the next one has been handed out.
    The inputs (edge lists, follow tensors) must be complete on the current stream when iteration starts.  Which one is faster is a
    measurement: the overlap hides the host's one read and the launch gaps of the construction but its sorts compete with the
    step for the memory system (config 4 on an MI355X: 5.76 ms/step against 6.06 for the plain loop once the construction was down
    to 1.25 ms per minibatch; 6.88 against 6.79 when it took 1.5 ms - DESIGN.md section 4.7).  `build_events` (timing=True): (start, end) event pairs, two per minibatch."""

    def __init__(self, minibatches, with_dec=True, prefetch=True, depth=2, timing=False):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.minibatches, self.with_dec, self.prefetch, self.depth, self.timing = minibatches, bool(with_dec), bool(prefetch), int(depth), bool(timing)
        self.build_events = []

    def _timed(self, fn):
        if not self.timing:
            return fn()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        t.record()
        self.build_events.append((s, t))
        return out

    def _begin(self, item):
        def go():
            cb = CollatedBatch(item[0], item[1], item[2] if len(item) > 2 else None)
            return cb, cb.event_graph(edge_weight=item[3] if len(item) > 3 else None, with_dec=self.with_dec, defer=True)
        return self._timed(go)

    def _finish(self, cb, g):
        return self._timed(lambda: (cb, CollatedBatch.finish_graph(g)))

    def __iter__(self):
        it = iter(self.minibatches)
        if not self.prefetch:
            for item in it:
                yield self._finish(*self._begin(item))
            return
        from collections import deque
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=main.device)
        side.wait_stream(main)
        window = deque()
        behind = None               # current-stream event behind the step in flight

        for _ in range(self.depth):
            item = next(it, None)
            if item is None:
                break
            with torch.cuda.stream(side):
                window.append(self._begin(item))
        guard = None                # current-stream event behind the step BEFORE the one in flight
        while True:
            # Everything the side stream does from here on may allocate, and what the allocator hands out are the structures of
            # minibatches the caller has dropped - the last of them was read by the step in front of `guard`.  The wait comes
            # first and covers the `begin` and the `finish` below (also at the tail of the sequence, where there is no `begin`).
            with torch.cuda.stream(side):
                if guard is not None:
                    side.wait_event(guard)
                if behind is not None:
                    item = next(it, None)
                    if item is not None:
                        window.append(self._begin(item))
                if not window:
                    return
                cb, g = self._finish(*window.popleft())
                done = torch.cuda.Event()
                done.record(side)
            guard = behind
            main.wait_event(done)
            yield cb, g
            behind = torch.cuda.Event()
            behind.record(main)     # the caller's step on (cb, g) is queued in front of this event
