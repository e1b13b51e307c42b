"""GPU: ancestor extraction for all targets at once (ceeg_top_predecessors / ceeg_ancestors /
ceeg_ancestor_subgraph_*) against the CHECKER - the pandas + networkx restatement of scripts/dataset.py:22-46,122-156,201-225
in tests/test_ingest_cpu.py (`_restated`, `_top`, `_anc`, nx.induced_subgraph) - fed from the restated graph itself, not
from the product's host code; the product's host path (ingest.PredecessorIndex / ancestor_subgraph) is compared as a second
assertion.  Synthetic tables of the reference's CSV schema (parity unpinned: the reference holds no expected output here)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
pd = pytest.importorskip("pandas")

nx = pytest.importorskip("networkx")

from test_ingest_cpu import _anc, _restated, _tables, _top  # noqa: E402


@pytest.mark.parametrize("n_nodes,n_rows,seed", [(97, 234, 0), (400, 3000, 1), (30, 10, 2), (1500, 20000, 3)])
def test_device_ancestors_equal_networkx_restatement(tmp_path, n_nodes, n_rows, seed):
    """DeviceAncestors.top / anc / count / subgraphs() against the restated reference builder directly.  The device input is
    read off the restated networkx graph (in-edges of every node in `predecessors` order = first appearance, which is the only
    order the tie-break depends on), so nothing of causal_eeg_vae_amd's host code sits between checker and kernels."""
    from causal_eeg_vae_amd import ingest
    ncsv, ecsv = _tables(tmp_path, n_nodes, n_rows, seed)
    graph, order = _restated(ncsv, ecsv)
    pos = {lab: i for i, lab in enumerate(order)}
    N = len(order)
    src, dst, ent = [], [], []
    for lab in order:
        for p in graph.predecessors(lab):
            src.append(pos[p]); dst.append(pos[lab]); ent.append(graph[p][lab]["entities"])
    ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1).cuda()
    da = ingest.DeviceAncestors(ei, torch.tensor(ent, dtype=torch.int64).cuda(), N)
    top, anc, cnt = da.top.cpu().numpy(), da.anc.cpu().numpy(), da.count.cpu().numpy()
    sub, eptr = da.subgraphs()
    sub, eptr = sub.cpu().numpy(), eptr.cpu().numpy()
    assert eptr[0] == 0 and eptr[-1] == sub.shape[1]
    n_anc = n_sub = 0
    for lab in order:
        v = pos[lab]
        assert [int(x) for x in top[v] if x >= 0] == [pos[p] for p in _top(graph, lab, ingest.MAX_TOP_PREDECESSORS)]
        a_labs = _anc(graph, lab, ingest.NUM_GENERATIONS, ingest.MAX_TOP_PREDECESSORS)
        want = sorted(pos[p] for p in a_labs)
        assert cnt[v] == len(want) and anc[v, : cnt[v]].tolist() == want and (anc[v, cnt[v]:] == -1).all()
        loc = {g: i for i, g in enumerate(want)}
        ind = nx.induced_subgraph(graph, a_labs)
        want_edges = sorted((loc[pos[b]], loc[pos[a]]) for a, b in ind.edges())         # (target, source), local ids
        mine = sub[:, eptr[v]: eptr[v + 1]]
        assert list(zip(mine[1].tolist(), mine[0].tolist())) == want_edges
        n_anc += len(want)
        n_sub += len(want_edges)
    if n_rows > 100:
        assert n_anc > 0 and (n_sub > 0 or n_rows < 1000)
    tg = torch.arange(0, N, 3, dtype=torch.int64).cuda()
    d2 = ingest.DeviceAncestors(ei, torch.tensor(ent, dtype=torch.int64).cuda(), N, targets=tg, num_gens=2, num_predecessors=2)
    a2, c2 = d2.anc.cpu().numpy(), d2.count.cpu().numpy()
    for i, v in enumerate(tg.cpu().tolist()):
        assert a2[i, : c2[i]].tolist() == sorted(pos[p] for p in _anc(graph, order[v], 2, 2))


@pytest.mark.parametrize("n_nodes,n_rows,seed", [(97, 234, 0), (400, 3000, 1), (30, 10, 2), (3000, 40000, 3)])
def test_device_ancestors_equal_host_extraction(tmp_path, n_nodes, n_rows, seed):
    from causal_eeg_vae_amd import ingest
    ncsv, ecsv = _tables(tmp_path, n_nodes, n_rows, seed)
    g = ingest.load_event_graph_csv(ncsv, ecsv)
    N = g["num_nodes"]
    ei, ent = g["edge_index"], g["entities"]
    host = ingest.PredecessorIndex(ei, ent, N)
    da = ingest.DeviceAncestors(torch.from_numpy(ei).cuda(), torch.from_numpy(ent).cuda(), N)
    top = da.top.cpu().numpy()
    anc, cnt = da.anc.cpu().numpy(), da.count.cpu().numpy()
    sub, eptr = da.subgraphs()
    sub, eptr = sub.cpu().numpy(), eptr.cpu().numpy()
    assert eptr[0] == 0 and eptr[-1] == sub.shape[1] and (np.diff(eptr) >= 0).all()
    n_anc = 0
    for v in range(N):
        want_top = host.top_predecessors(v, ingest.MAX_TOP_PREDECESSORS)
        assert [int(x) for x in top[v] if x >= 0] == want_top               # order matters: heaviest first, ties by first appearance
        want = sorted(host.ancestors(v))
        assert cnt[v] == len(want) and anc[v, : cnt[v]].tolist() == want and (anc[v, cnt[v]:] == -1).all()
        n_anc += len(want)
        hsub, keep = ingest.ancestor_subgraph(ei, ent, N, v, host)
        assert keep == want
        mine = sub[:, eptr[v]: eptr[v + 1]]
        key = lambda a: sorted(zip(a[1].tolist(), a[0].tolist()))             # canonical (target, source) order
        assert key(mine) == key(hsub)
        assert mine.shape[1] == 0 or list(zip(mine[1].tolist(), mine[0].tolist())) == key(mine)   # already emitted in that order
    if n_rows > 100:
        assert n_anc > 0
    # a subset of targets, fewer generations / predecessors
    tg = torch.arange(0, N, 3, dtype=torch.int64).cuda()
    d2 = ingest.DeviceAncestors(torch.from_numpy(ei).cuda(), torch.from_numpy(ent).cuda(), N, targets=tg, num_gens=2, num_predecessors=2)
    a2, c2 = d2.anc.cpu().numpy(), d2.count.cpu().numpy()
    for i, v in enumerate(tg.cpu().tolist()):
        assert a2[i, : c2[i]].tolist() == sorted(host.ancestors(v, 2, 2))


def test_device_ancestors_argument_checks():
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    assert L.ceeg_ancestors_cap(3, 3) == 39 and L.ceeg_ancestors_cap(2, 2) == 6 and L.ceeg_ancestors_cap(8, 3) == -1
    t = torch.zeros(64, dtype=torch.int32, device="cuda")
    p = t.data_ptr()
    assert L.ceeg_top_predecessors(p, p, p, p, 4, 9, 3, p, _lib.stream()) == -2          # more predecessors than the kernel keeps
    assert L.ceeg_ancestors(p, 4, 3, None, 4, 3, 10, p, p, _lib.stream()) == -2          # cap below 39
    assert L.ceeg_ancestors(None, 4, 3, None, 4, 3, 39, p, p, _lib.stream()) == -1


@pytest.mark.parametrize("n,rows,seed", [(50, 400, 0), (1000, 30000, 1), (7, 0, 2), (200_000, 2_000_000, 3)])
def test_multigraph_collapse_on_device_equals_host(n, rows, seed):
    """ingest.collapse_multigraph (dataset.py:147-153: parallel rows -> one edge with `entities` = multiplicity, edges in
    first-appearance order) on the device against the numpy form and, for the small cases, a plain dict loop."""
    from causal_eeg_vae_amd import ingest
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, rows)
    dst = (src + rng.integers(0, 5, rows)) % n          # few distinct targets per source: many repeats
    ei_h, cnt_h = ingest.collapse_multigraph(src, dst, n)
    ei_d, cnt_d = ingest.collapse_multigraph(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n)
    np.testing.assert_array_equal(ei_d.cpu().numpy(), ei_h)
    np.testing.assert_array_equal(cnt_d.cpu().numpy(), cnt_h)
    assert int(cnt_h.sum()) == rows and ei_h.shape == (2, len(cnt_h))
    if rows <= 30000:
        first = {}
        for a, b in zip(src.tolist(), dst.tolist()):
            first[(a, b)] = first.get((a, b), 0) + 1      # dicts keep insertion order = first appearance
        assert list(first.keys()) == list(zip(ei_h[0].tolist(), ei_h[1].tolist())) and list(first.values()) == cnt_h.tolist()


@pytest.mark.parametrize("negatives", ["uniform", "sigma"])
def test_config1_plumbing_csv_to_training(tmp_path, negatives):
    """BASELINE configs[0] as plumbing: tables of the reference's CSV schema at its own size (97 node rows, 234 edge rows ->
    a graph of about 90 nodes / 200 collapsed edges, `entities` weights) through ingestion, the weighted GCN normalisation and
    20 optimizer steps of the GCN-VGAE + VQ model; the loss is finite and goes down.  Synthetic tables: the reference's own
    CSVs are not read."""
    from causal_eeg_vae_amd.run_csv import train_csv
    ncsv, ecsv = _tables(tmp_path, 97, 234, 0)
    lines = []
    model, hist = train_csv(ncsv, ecsv, epochs=4, steps_per_epoch=5, in_dim=16, latent=8, codes=32, negatives=negatives, log=lines.append)
    assert len(hist) == 4 and all(np.isfinite(hist)) and hist[-1] < hist[0]
    assert len(lines) == 4 and '"epoch": 3' in lines[-1]
    assert model.vector_quantization.embedding.weight.shape == (32, 8)
