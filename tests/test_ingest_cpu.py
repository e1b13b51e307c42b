"""Event-graph ingestion (causal_eeg_vae_amd/ingest.py) against a pandas + networkx restatement of the reference's
dataset builder (scripts/dataset.py:22-46,122-156) on SYNTHETIC tables of the reference's CSV schema.
Parity unpinned: the reference holds no test or expected output for this step, and its own data files are not used
(see DESIGN.md §2); the restatement below is the checker, not the product path."""
import numpy as np
import pytest

pd = pytest.importorskip("pandas")
nx = pytest.importorskip("networkx")

from causal_eeg_vae_amd import ingest  # noqa: E402


def _tables(tmp_path, n_nodes, n_rows, seed):
    rng = np.random.default_rng(seed)
    ids = rng.permutation(n_nodes * 2)[:n_nodes]          # arbitrary, non-contiguous ids
    titles = [f"title {i}" for i in range(n_nodes)]
    for i in rng.choice(n_nodes, n_nodes // 10, replace=False):   # duplicate titles (dropped by the builder)
        titles[i] = titles[int(rng.integers(0, n_nodes))]
    text = [f"multi\nline, \"quoted\" text {i}" for i in range(n_nodes)]
    date = ["2021-01-01"] * n_nodes
    url = [f"http://x/{i}" for i in range(n_nodes)]
    nodes = pd.DataFrame({"title": titles, "text": text, "publish_date": date, "url": url, "id": ids})
    for col in ("text", "publish_date", "url"):                   # missing values (rows dropped by dropna)
        nodes.loc[rng.choice(n_nodes, 3, replace=False), col] = np.nan
    dst = rng.integers(1, n_nodes, n_rows)
    src = (dst * rng.random(n_rows) ** 2).astype(np.int64)
    few = rng.integers(0, max(n_rows // 8, 1), n_rows)            # repeat a few pairs often: entities up to ~8
    src, dst = np.where(few < 6, src[few], src), np.where(few < 6, dst[few], dst)
    edges = pd.DataFrame({"old_id": ids[src], "new_id": ids[dst], "entity": [f"e{k}" for k in range(n_rows)],
                          "entity_type": "LOC", "date_diff": rng.integers(1, 1000, n_rows)})
    edges.loc[0, "old_id"] = 10 ** 6                               # endpoint that is in no node row
    npth, epth = tmp_path / "g_nodes.csv", tmp_path / "g_edges.csv"
    nodes.to_csv(npth, index=False)
    edges.to_csv(epth, index=False)
    return str(npth), str(epth)


def _restated(nodes_csv, edges_csv):
    nodes_df = pd.read_csv(nodes_csv).dropna().drop_duplicates(subset="title")
    edges_df = pd.read_csv(edges_csv)
    graph, mapping = nx.DiGraph(), {}
    for idx, row in nodes_df.iterrows():
        mapping[row["id"]] = idx
        graph.add_node(idx)
    for _, row in edges_df.iterrows():
        if row["old_id"] not in mapping or row["new_id"] not in mapping:
            continue
        o, n = mapping[row["old_id"]], mapping[row["new_id"]]
        if graph.has_edge(o, n):
            graph[o][n]["entities"] += 1
        else:
            graph.add_edge(o, n, entities=1)
    return graph, list(nodes_df.index)


def _top(graph, node, n):
    pw = [(p, graph[p][node]["entities"]) for p in graph.predecessors(node) if graph[p][node]["entities"] >= 3]
    return [p for p, _ in sorted(pw, key=lambda x: x[1], reverse=True)][: min(n, graph.in_degree(node))]


def _anc(graph, node, gens, npred):
    if gens == 0:
        return {node}
    out = set()
    for p in _top(graph, node, npred):
        out.add(p)
        out.update(_anc(graph, p, gens - 1, npred))
    return out


@pytest.mark.parametrize("n_nodes,n_rows,seed", [(97, 234, 0), (400, 3000, 1), (30, 10, 2)])
def test_csv_to_graph_and_ancestors(tmp_path, n_nodes, n_rows, seed):
    ncsv, ecsv = _tables(tmp_path, n_nodes, n_rows, seed)
    got = ingest.load_event_graph_csv(ncsv, ecsv)
    graph, order = _restated(ncsv, ecsv)
    pos = {lab: i for i, lab in enumerate(order)}
    assert got["num_nodes"] == len(order) < n_nodes            # something was dropped
    want = {(pos[a], pos[b]): d["entities"] for a, b, d in graph.edges(data=True)}
    have = {(int(a), int(b)): int(w) for a, b, w in zip(got["edge_index"][0], got["edge_index"][1], got["entities"])}
    assert have == want and len(have) == got["edge_index"].shape[1]
    idx = ingest.PredecessorIndex(got["edge_index"], got["entities"], got["num_nodes"])
    some_anc = 0
    for lab in order:
        v = pos[lab]
        assert idx.top_predecessors(v, 1) == [pos[p] for p in _top(graph, lab, 1)]
        assert idx.top_predecessors(v, 3) == [pos[p] for p in _top(graph, lab, 3)]
        a = {pos[p] for p in _anc(graph, lab, 3, 3)}
        assert idx.ancestors(v) == a
        some_anc += len(a)
        sub, keep = ingest.ancestor_subgraph(got["edge_index"], got["entities"], got["num_nodes"], v, idx)
        ind = nx.induced_subgraph(graph, [order[k] for k in keep])
        assert sub.shape[1] == ind.number_of_edges() and keep == sorted(a)
    if n_rows > 100:
        assert some_anc > 0


def test_duplicated_node_id_maps_to_its_last_row(tmp_path):
    """dataset.py:137 `node_mapping[node_row['id']] = idx`: when an id appears on two rows (distinct titles, so both survive
    the cleaning) the LAST row wins; edges naming that id attach there."""
    nodes = pd.DataFrame({"title": ["a", "b", "c", "d"], "text": ["t"] * 4, "publish_date": ["2021-01-01"] * 4, "url": ["u"] * 4,
                          "id": [7, 5, 7, 9]})
    edges = pd.DataFrame({"old_id": [5, 7, 7], "new_id": [7, 9, 9], "entity": ["x", "y", "z"], "entity_type": "LOC", "date_diff": [1, 2, 3]})
    npth, epth = tmp_path / "n.csv", tmp_path / "e.csv"
    nodes.to_csv(npth, index=False)
    edges.to_csv(epth, index=False)
    got = ingest.load_event_graph_csv(str(npth), str(epth))
    graph, order = _restated(str(npth), str(epth))
    pos = {lab: i for i, lab in enumerate(order)}
    want = {(pos[a], pos[b]): d["entities"] for a, b, d in graph.edges(data=True)}
    have = {(int(a), int(b)): int(w) for a, b, w in zip(got["edge_index"][0], got["edge_index"][1], got["entities"])}
    assert have == want == {(1, 2): 1, (2, 3): 2}
