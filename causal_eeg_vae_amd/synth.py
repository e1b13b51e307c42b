"""Synthetic power-law event graphs of BASELINE.json's shapes.

Event graphs are time-ordered DAGs: an article (dst, `new_id`) links back to earlier articles (src, `old_id`)
that share named entities (/root/reference/scripts/dataset.py:143-153).  The generator mixes two sources
per edge so that the degree distribution is heavy-tailed AND links are mostly recent:
  * with probability p_hub  : src = floor(dst * u^gamma)         (old "hub" events: power-law out-degree)
  * otherwise               : src = dst - 1 - floor(age), age ~ Pareto(alpha) * scale   (recent events)
Edges are unique, src < dst, no self-loops; exactly `num_edges` of them (oversample, unique, trim).
"""
import torch


def synth_event_graph(num_nodes, num_edges, seed=0, device="cpu", p_hub=0.35, gamma=3.0, alpha=1.2, scale=8.0):
    N, E = int(num_nodes), int(num_edges)
    if E > N * (N - 1) // 2:
        raise ValueError("more edges requested than a DAG on num_nodes nodes can hold")
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = E
    for _ in range(64):
        m = int((want - keys.numel()) * 1.15) + 1024
        dst = torch.randint(1, N, (m,), generator=gen, device=device)
        u = torch.rand(m, generator=gen, device=device, dtype=torch.float64)
        hub = torch.rand(m, generator=gen, device=device) < p_hub
        src_hub = (dst.to(torch.float64) * u.pow(gamma)).floor().to(torch.int64)
        age = ((1.0 - u).clamp_min(1e-12).pow(-1.0 / alpha) - 1.0) * scale
        src_loc = dst - 1 - age.floor().clamp(max=2.0**40).to(torch.int64)
        src = torch.where(hub, src_hub, src_loc)
        ok = (src >= 0) & (src < dst)
        keys = torch.unique(torch.cat([keys, dst[ok] * N + src[ok]]))
        if keys.numel() >= want:
            break
    if keys.numel() < want:
        raise RuntimeError("generator could not reach the requested edge count")
    # keep exactly E edges, chosen by a seeded permutation so the trim is not biased towards low dst
    sel = torch.randperm(keys.numel(), generator=gen, device=device)[:E]
    keys = keys[sel]
    return torch.stack([keys % N, keys // N])  # edge_index[0] = src (old), edge_index[1] = dst (new)


def synth_features(num_nodes, dim, seed=0, device="cpu"):
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) + 1_000_003)
    return torch.randn(num_nodes, dim, generator=gen, device=device)


def collate_graphs(edge_indices, nodes_per_graph):
    """PyG-style block-diagonal collation (scripts/run.py:310): offsets added to edge_index, `batch` vector.
    On the GPU this is causal_eeg_vae_amd.collate.CollatedBatch (two HIP kernels over the concatenated lists); CPU inputs
    (the oracle side of the tests) take the plain torch restatement below."""
    if edge_indices and edge_indices[0].is_cuda:
        from .collate import CollatedBatch
        c = CollatedBatch(edge_indices, nodes_per_graph)
        return c.edge_index, c.batch, c.num_nodes
    offs, eis, batch = 0, [], []
    for g, (ei, n) in enumerate(zip(edge_indices, nodes_per_graph)):
        eis.append(ei + offs)
        batch.append(torch.full((n,), g, dtype=torch.int64, device=ei.device))
        offs += n
    return torch.cat(eis, dim=1), torch.cat(batch), offs
