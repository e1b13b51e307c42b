#!/usr/bin/env python3
"""bench.py — the north-star metric on its named config.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): synthetic power-law event graph, 1 M nodes / 16 M directed edges per
GPU, 2-layer GCN-VGAE with a 128-d latent + VQ bottleneck (512 codes), fp32; one "step" = one full
optimizer step (encode -> reparam+KL -> VQ -> decode with 1:1 negatives -> backward -> [all-reduce] -> Adam)
with every input already resident in HBM.  value = positive directed edges processed per second, whole job.
Multi-GPU: one graph shard per rank (weak scaling), gradients averaged with one RCCL all-reduce per step.

Extra (my own measurements, not the driver's contract): --workload vq (BASELINE.json configs[2]) and
--workload cvgae (configs[3]: conditional VGAE + VQ on 64 collated graphs x 100k edges per rank), --workload gat
(next-row N1: GATConv forward + backward on the configs[1] graph).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense
MFMA_F32_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=16_000_000)
    ap.add_argument("--in-dim", type=int, default=128)
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--codes", type=int, default=512)
    ap.add_argument("--gat-heads", type=int, default=1)
    ap.add_argument("--workload", default="vgae", choices=["vgae", "vq", "cvgae", "gat"])
    ap.add_argument("--graphs", type=int, default=64)           # cvgae (configs[3]): graphs per rank
    ap.add_argument("--edges-per-graph", type=int, default=100_000)
    ap.add_argument("--nodes-per-graph", type=int, default=6_250)
    ap.add_argument("--cond-dim", type=int, default=32)
    ap.add_argument("--vq-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (CPU restatement, kind "port") timed on this box's host cores on a 1/16 sample of the workload."""
    from oracle import ref_torch  # checker / reported baseline only
    from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    n, e = args.nodes // 16, args.edges // 16
    ei = synth_event_graph(n, e, seed=0)
    x = synth_features(n, args.in_dim, seed=0)
    torch.manual_seed(0)
    model = ref_torch.RefVGAEVQ(args.in_dim, args.latent, args.codes, 0.25)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(1)
    best = float("inf")
    for it in range(3):
        eps = torch.randn(n, args.latent, generator=gen)
        neg = torch.randint(0, n, (2, e), generator=gen)
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = model(x, ei, eps, neg)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0:
            best = min(best, dt)
    return {"value": e / best, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"oracle/ref_torch.RefVGAEVQ full step (fwd+bwd+Adam), {n} nodes / {e} edges (1/16 of the workload), best of 2 after 1 warm-up"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; CEEG_DIST_BACKEND=gloo + CEEG_SHARE_GPU=1 is the one-GPU rehearsal of the multi-process path
    backend = os.environ.get("CEEG_DIST_BACKEND", "nccl")
    if os.environ.get("CEEG_SHARE_GPU") == "1":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    verbose = os.environ.get("CEEG_BENCH_VERBOSE") == "1"

    def note(msg):
        if verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    note("process group up")
    from causal_eeg_vae_amd import _lib, EventGraph, VGAEVQ, VectorQuantizer
    from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
    from causal_eeg_vae_amd.trainer import Trainer
    _lib.check(_lib.lib().ceeg_check_device(), "ceeg_check_device")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.workload == "vq":
        B, n_e, D = 4 * 1024 * 1024, 512, 256
        dt = torch.bfloat16 if args.vq_dtype == "bf16" else torch.float32
        torch.manual_seed(rank)
        z = torch.randn(B, D, device=dev).to(dt)
        vq = VectorQuantizer(n_e, D, 0.25, return_one_hot=False).to(dev)
        with torch.no_grad():
            vq.embedding.weight.normal_()
            for _ in range(args.warmup):
                vq(z)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                vq(z)
            barrier()
        el = time.perf_counter() - t0
        flops = 2.0 * B * n_e * D
        ms = el / args.steps * 1e3
        peak = MFMA_BF16_PEAK_TF if dt == torch.bfloat16 else MFMA_F32_PEAK_TF
        print(json.dumps({"metric": "VQ forward latents/s", "value": B * world / (el / args.steps), "unit": "latents/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.vq_dtype, "data": "synthetic",
                          "config": {"workload": f"VectorQuantizer fwd {B}x{n_e}x{D}"},
                          "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                       "frac": flops / (ms * 1e-3) / 1e12 / peak, "traffic": None}}))
        return

    if args.workload == "gat":
        # next-row N1: the graph layer the reference instantiates (scripts/models.py:106), forward +
        # backward on the config-2 graph; heads = 1, in_dim -> 2 * latent as the encoder's first layer
        from causal_eeg_vae_amd import GATConv
        N, E = args.nodes, args.edges
        ei = synth_event_graph(N, E, seed=1000 + rank, device=dev)
        x = synth_features(N, args.in_dim, seed=1000 + rank, device=dev).requires_grad_(True)
        graph = GATConv.attention_graph(ei, N)
        torch.manual_seed(0)
        conv = GATConv(args.in_dim, 2 * args.latent, heads=args.gat_heads).to(dev)

        gout = torch.ones(N, 2 * args.latent * args.gat_heads, device=dev)  # the incoming gradient, allocated once

        def once():
            out = conv(x, graph)
            out.backward(gout)
            x.grad = None
            conv.zero_grad(set_to_none=True)

        for _ in range(args.warmup):
            once()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            once()
        barrier()
        el = time.perf_counter() - t0
        print(json.dumps({"metric": "GATConv fwd+bwd edges/sec", "value": E * world / (el / args.steps), "unit": "edges/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"GATConv({args.in_dim}, {2 * args.latent}, heads={args.gat_heads}) forward + backward, "
                                                 f"synthetic power-law event graph {N} nodes / {E} edges per GPU"}}))
        return

    if args.workload == "cvgae":
        from causal_eeg_vae_amd import ConditionalVGAEVQ
        from causal_eeg_vae_amd.synth import collate_graphs
        G, n, e, C = args.graphs, args.nodes_per_graph, args.edges_per_graph, args.cond_dim
        eis = [synth_event_graph(n, e, seed=(1000 + rank) * 4096 + g, device=dev) for g in range(G)]
        ei, batch, N = collate_graphs(eis, [n] * G)
        x = synth_features(N, args.in_dim, seed=1000 + rank, device=dev)
        gen = torch.Generator(device=dev).manual_seed(77 + rank)
        cond = torch.randn(G, C, generator=gen, device=dev)
        graph = EventGraph(ei, N)
        graph.dec
        torch.manual_seed(0)
        model = ConditionalVGAEVQ(args.in_dim, C, args.latent, args.codes, 0.25, seed=rank).to(dev)
        with torch.no_grad():
            model.vector_quantization.embedding.weight.normal_(0, 0.1)
        trainer = Trainer(model, lr=1e-3, world_size=world)
        for _ in range(args.warmup):
            trainer.step(x, graph, batch, cond, nodes_per_graph=n)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, aux = trainer.step(x, graph, batch, cond, nodes_per_graph=n)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            if backend != "nccl":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        if rank == 0:
            print(json.dumps({"metric": "VGAE train edges/sec", "value": G * e * world / (el / args.steps), "unit": "edges/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"conditional GCN-VGAE+VQ train step, {G} collated graphs x {e} edges x {n} nodes per GPU, "
                                                     f"in {args.in_dim}+{C}, latent {args.latent}(+{C} decoding), {args.codes} codes, per-graph 1:1 negatives, Adam",
                                         "parallelism": f"dp{world}"}, "loss": float(loss.item())}))
        if world > 1:
            dist.destroy_process_group()
        return

    N, E = args.nodes, args.edges
    ei = synth_event_graph(N, E, seed=1000 + rank, device=dev)   # shard r: seed base + r
    x = synth_features(N, args.in_dim, seed=1000 + rank, device=dev)
    graph = EventGraph(ei, N)
    graph.dec  # build the decoder structure before timing
    torch.manual_seed(0)
    model = VGAEVQ(args.in_dim, args.latent, args.codes, 0.25, seed=rank).to(dev)
    with torch.no_grad():
        model.vector_quantization.embedding.weight.normal_(0, 0.1)
    note("graph built")
    trainer = Trainer(model, lr=1e-3, world_size=world)
    note("parameters broadcast")

    for _ in range(args.warmup):
        trainer.step(x, graph)
    barrier()
    note("warm-up done")
    _lib.PROFILE = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, aux = trainer.step(x, graph)
    barrier()
    el = time.perf_counter() - t0
    prof = _lib.profile_summary()
    _lib.PROFILE = None
    if not torch.isfinite(loss).item():
        print("bench.py: non-finite loss", file=sys.stderr)
        sys.exit(3)
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        if backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    if rank == 0:
        ms = el / args.steps * 1e3
        F = 2 * args.latent
        nnz = graph.gcn.nnz
        # compulsory bytes of one SpMM pass of width F
        spmm_bytes = nnz * 8 + (N + 1) * 4 + 2 * N * F * 4
        launches, spmm_ms = prof.get(f"spmm_F{F}", (0, float("nan")))  # layer-2 aggregation (fwd) and its transposed pass (bwd)
        achieved = spmm_bytes / (spmm_ms * 1e-3) / 1e9
        # HBM-side bytes per launch from the rocprofv3 PMC passes of this round (profiles/r01_step_pmc_hbm.txt);
        # counters cannot be read from inside the process, so the figure is reported only for the workload it was
        # measured on
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_spmm_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            if (t.get("nodes"), t.get("edges"), t.get("F")) == (N, E, F):
                traffic = t["traffic_bytes"]
        line = {"metric": "VGAE train edges/sec", "value": E * world / (el / args.steps), "unit": "edges/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"GCN-VGAE+VQ train step, synthetic power-law event graph {N} nodes / {E} edges per GPU, "
                                       f"in {args.in_dim}, latent {args.latent}, {args.codes} codes, 1:1 negatives, Adam",
                           "parallelism": f"dp{world}"},
                "roofline": {"bound": "hbm", "kernel": f"ceeg_spmm_csr F={F} (spmm_rows_kernel<64> + long-row pieces)",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": traffic, "launches_timed": launches, "avg_launch_ms": spmm_ms,
                             "algorithmic_bytes_per_launch": spmm_bytes},
                "loss": float(loss.item())}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
