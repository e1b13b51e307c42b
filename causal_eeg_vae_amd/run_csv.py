"""Config-1 plumbing (BASELINE.json configs[0]): the on-disk CSV pair of the reference -> event graph -> GCN-VGAE + VQ
training on the GPU.  The host loop of /root/reference/scripts/run.py:54-135 restated for this model: epochs of
optimizer steps (Adam 1e-3, scripts/run.py:56) and `ReduceLROnPlateau` (:57) stepped once per epoch.  Two differences from the
reference's loop, both because there is no validation split here: the scheduler sees the TRAINING epoch loss (the reference
steps it on the validation loss, :136), and no best-loss checkpointing / early stopping is kept (:139-149).

    python -m causal_eeg_vae_amd.run_csv NODES.csv EDGES.csv [--epochs 5 --steps-per-epoch 20 --latent 16 --codes 64]

Node features: the reference embeds the article text with DistilBERT (scripts/models.py:57-59), which is out of scope
(SURVEY.md section 2); pass `--features file.npy` ([num_nodes, F] float32, rows in the order of the cleaned node table) or
synthetic N(0, 1) features are drawn (seed 0).  Edge weights: the `entities` multiplicity of every collapsed edge
(scripts/dataset.py:147-153) weights the GCN normalisation; `--unweighted` switches that off."""
import argparse
import json
import sys

import numpy as np
import torch

from . import ingest
from .models import EventGraph, VGAEVQ
from .trainer import Trainer


def train_csv(nodes_csv, edges_csv, epochs=5, steps_per_epoch=20, in_dim=32, latent=16, codes=64, beta=0.25, features=None, weighted=True,
              negatives="uniform", seed=0, device="cuda", log=None):
    """Returns (model, history): history[e] = mean loss of epoch e.  Raises CeegError without a GPU / libceeg.so."""
    g = ingest.load_event_graph_csv(nodes_csv, edges_csv)
    N = g["num_nodes"]
    dev = torch.device(device)
    ei = torch.from_numpy(g["edge_index"]).to(dev)
    ew = torch.from_numpy(g["entities"]).to(dev, torch.float32) if weighted else None
    if features is not None:
        x = torch.from_numpy(np.load(features).astype(np.float32)).to(dev)
        if x.shape[0] != N:
            raise ValueError(f"features have {x.shape[0]} rows, the cleaned node table {N}")
    else:
        x = torch.randn(N, in_dim, generator=torch.Generator().manual_seed(seed)).to(dev)
    graph = EventGraph(ei, N, edge_weight=ew)
    torch.manual_seed(seed)
    model = VGAEVQ(x.shape[1], latent, codes, beta, seed=seed, negatives=negatives).to(dev)
    trainer = Trainer(model, lr=1e-3)
    history = []
    for epoch in range(epochs):
        model.train()
        total = torch.zeros((), device=dev)
        for _ in range(steps_per_epoch):
            loss, aux = trainer.step(x, graph)
            total += loss
        mean = float(total.item()) / max(steps_per_epoch, 1)     # one host sync per epoch
        trainer.end_epoch(mean)
        history.append(mean)
        if log is not None:
            log(json.dumps({"epoch": epoch, "loss": mean, "perplexity": float(aux["ppl"]), "lr": trainer.opt.param_groups[0]["lr"],
                            "nodes": N, "edges": int(ei.shape[1])}))
    return model, history


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("nodes_csv")
    ap.add_argument("edges_csv")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--steps-per-epoch", type=int, default=20)
    ap.add_argument("--in-dim", type=int, default=32)
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--codes", type=int, default=64)
    ap.add_argument("--features", default=None)
    ap.add_argument("--unweighted", action="store_true")
    ap.add_argument("--negatives", default="uniform", choices=["uniform", "sigma"])
    a = ap.parse_args(argv)
    train_csv(a.nodes_csv, a.edges_csv, a.epochs, a.steps_per_epoch, a.in_dim, a.latent, a.codes, features=a.features,
              weighted=not a.unweighted, negatives=a.negatives, log=print)
    return 0


if __name__ == "__main__":
    sys.exit(main())
