"""GraphConditioner of the reference (scripts/models.py:95-122) on the MI355X kernels: per-node token embeddings ->
learned-attention token pooling (bmm) -> [GATConv -> LayerNorm(mode='graph') -> ReLU] per layer -> GlobalAttention graph
pooling.  SURVEY.md section 8(f) rows N1 + N4.  The token encoder (DistilBERT in the reference, scripts/run.py:287-289)
is out of scope and stays a plug-in: any callable `embedding(input_ids=..., attention_mask=...)` whose result has
`.last_hidden_state` [num_nodes, tokens, hidden]."""
import torch
import torch.nn as nn

from .models import GATConv
from .pool import GlobalAttention, GraphLayerNorm, token_attention_pool


class GraphConditioner(nn.Module):
    """Same constructor, forward argument and parameter names as the reference class: `attn`, `graph_sequential_model.
    module_{0,1,3,4,...}` (torch_geometric.nn.Sequential numbers its modules, ReLUs included) and `pooling_layer.gate_nn`,
    so a state dict of the reference's conditioner loads (PyG GATConv keys lin_src.weight / att_src / att_dst / bias,
    LayerNorm weight / bias).  `input` needs input_ids, attention_mask, edge_index, input_ids_batch and num_graphs."""

    def __init__(self, embedding, embedding_hidden_size, graph_layer_sizes):
        super().__init__()
        self.embedding = embedding
        self.attn = nn.Linear(embedding_hidden_size, 1)
        seq = nn.Module()
        self._layers = []
        sizes = [embedding_hidden_size] + list(graph_layer_sizes)
        for i, (in_size, out_size) in enumerate(zip(sizes[:-1], sizes[1:])):
            seq.add_module(f"module_{3 * i}", GATConv(in_size, out_size))
            seq.add_module(f"module_{3 * i + 1}", GraphLayerNorm(out_size))
            seq.add_module(f"module_{3 * i + 2}", nn.ReLU(inplace=True))
            self._layers.append((f"module_{3 * i}", f"module_{3 * i + 1}", f"module_{3 * i + 2}"))
        self.graph_sequential_model = seq
        self.pooling_layer = GlobalAttention(nn.Linear(graph_layer_sizes[-1], 1))

    def forward(self, input):
        output = self.embedding(input_ids=input.input_ids, attention_mask=input.attention_mask)
        hidden = output.last_hidden_state                                        # [N, T, H]
        # attn_weights = self.attn(hidden); x = bmm(hidden^T, attn_weights).squeeze(2)   (scripts/models.py:118-119), one kernel
        x = token_attention_pool(hidden, self.attn)                              # [N, H]
        batch = input.input_ids_batch
        # one attention CSR (self-loops replaced by one per node) for every layer
        pair = GATConv.attention_graph(input.edge_index, x.shape[0])
        seq = self.graph_sequential_model
        for conv, norm, _act in self._layers:   # the ReLU module (kept for the state-dict numbering) runs inside the norm's kernel
            x = getattr(seq, norm)(getattr(seq, conv)(x.contiguous(), pair), batch, input.num_graphs, relu=True)
        return self.pooling_layer(x, batch=batch, size=input.num_graphs)
