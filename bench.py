#!/usr/bin/env python3
"""bench.py — the north-star metric on its named config.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): synthetic power-law event graph, 1 M nodes / 16 M directed edges per
GPU, 2-layer GCN-VGAE with a 128-d latent + VQ bottleneck (512 codes), fp32; one "step" = one full
optimizer step (encode -> reparam+KL -> VQ -> decode with 1:1 negatives -> backward -> [all-reduce] -> Adam)
with every input already resident in HBM.  value = positive directed edges processed per second, whole job.
Multi-GPU: one graph shard per rank (weak scaling), gradients averaged with one RCCL all-reduce per step.

At N = 1 the default run appends, AFTER the headline's timed region, extra keys on the same JSON line: `vq_bf16` / `vq_f32`
(BASELINE.json configs[2], MFMA and HBM fractions), `cvgae` (configs[3]), `other_negatives` (the same step with the other
negative sampler: uniform negative_sampling when the headline ran the permutation-structured one) and, inside `roofline`,
the pure-gather ceiling of the SpMM.  --no-extras skips them.  Stand-alone: --workload vq | cvgae | gat.
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (configs[2], configs[3], uniform negatives, gather ceiling)")
    ap.add_argument("--negatives", default="uniform", choices=["sigma", "uniform"],
                    help="uniform = the north star's sampler (negative_sampling drawn every step, SURVEY.md 8 A6): the headline; "
                         "sigma = permutation-structured negatives inside the fused decoder (reported as the other_negatives extra)")
    ap.add_argument("--fresh-batches", type=int, default=0,
                    help="cvgae only: K > 0 = a minibatch stream - K resident minibatches visited round-robin, collation and graph construction "
                         "inside the timed region of every step (scripts/run.py:310-326)")
    ap.add_argument("--prefetch", action="store_true",
                    help="with --fresh-batches: build minibatch k+1 on a side stream while step k runs (collate.MinibatchStream)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="cvgae only (SURVEY.md 8(d) config 4): weak = --graphs per rank; strong = the --graphs minibatch split over the ranks")
    return ap.parse_args()


def host_threads():
    """Threads the CPU baseline may use: the cores this process is allowed to run on.  On the GPU box a one-GPU job owns a
    16-core share of a much larger machine (os.cpu_count() reports the machine; OMP_NUM_THREADS / the affinity mask report
    the share), so the smallest of the three is the honest count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(args):
    """Oracle (CPU restatement, kind "port") timed on this box's host cores on an N/8, E/8 sample of the workload
    (same generator and seeds, all available cores, 1 warm-up + best of 3, CPU model stated)."""
    from oracle import ref_torch  # checker / reported baseline only
    from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
    threads = host_threads()
    torch.set_num_threads(threads)
    n, e = args.nodes // 8, args.edges // 8
    ei = synth_event_graph(n, e, seed=1000)
    x = synth_features(n, args.in_dim, seed=1000)
    torch.manual_seed(0)
    model = ref_torch.RefVGAEVQ(args.in_dim, args.latent, args.codes, 0.25)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(1)
    best = float("inf")
    for it in range(4):
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
            "sample": f"oracle/ref_torch.RefVGAEVQ full step (fwd+bwd+Adam, stock PyTorch CPU ops, fp32) on {cpu_model()}, {threads} threads "
                      f"(machine reports {os.cpu_count()}), {n} nodes / {e} edges (1/8 of the workload, same generator), best of 3 after 1 warm-up"}


def run_vq(dev, dtype_name, steps, warmup, rank=0, world=1, barrier=None):
    """BASELINE.json configs[2]: VectorQuantizer forward, 4 Mi latents x 512 codes x 256-d.  Reports BOTH roofline fractions
    (in bf16 the kernel sits at the ridge): MFMA flops 2 B n_e D against the dense peak, and the
    compulsory HBM bytes B (2 D s + 8) + n_e D 4 against 8 TB/s."""
    from causal_eeg_vae_amd import VectorQuantizer
    B, n_e, D = 4 * 1024 * 1024, 512, 256
    dt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    torch.manual_seed(rank)
    z = torch.randn(B, D, device=dev).to(dt)
    vq = VectorQuantizer(n_e, D, 0.25, return_one_hot=False).to(dev)
    sync = barrier or torch.cuda.synchronize
    with torch.no_grad():
        vq.embedding.weight.normal_()
        for _ in range(warmup):
            vq(z)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            vq(z)
        sync()
    el = time.perf_counter() - t0
    ms = el / steps * 1e3
    flops = 2.0 * B * n_e * D
    peak = MFMA_BF16_PEAK_TF if dt == torch.bfloat16 else MFMA_F32_PEAK_TF
    es = 2 if dt == torch.bfloat16 else 4
    hbm_bytes = B * (2 * D * es + 8) + n_e * D * 4
    tf = flops / (ms * 1e-3) / 1e12
    return {"ms": ms, "latents_per_s": B * world / (el / steps), "TFLOP/s": tf, "mfma_peak_TFLOP/s": peak, "mfma_frac": tf / peak,
            "compulsory_GB": hbm_bytes / 1e9, "hbm_GB/s": hbm_bytes / (ms * 1e-3) / 1e9, "hbm_frac": hbm_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "shape": f"{B}x{n_e}x{D}", "dtype": dtype_name}


def run_cvgae(args, dev, steps, warmup, rank=0, world=1, barrier=None):
    """BASELINE.json configs[3]: conditional GCN-VGAE + VQ on a collated batch of graphs per rank."""
    from causal_eeg_vae_amd import ConditionalVGAEVQ, EventGraph
    from causal_eeg_vae_amd.synth import collate_graphs, synth_event_graph, synth_features
    from causal_eeg_vae_amd.trainer import Trainer
    G, n, e, C = args.graphs, args.nodes_per_graph, args.edges_per_graph, args.cond_dim
    first = 0
    if getattr(args, "scaling", "weak") == "strong":   # the same G graphs whatever the world size; rank r takes a contiguous share
        first, G = rank * (G // world), G // world
        eis = [synth_event_graph(n, e, seed=1000 * 4096 + first + g, device=dev) for g in range(G)]
    else:
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
    trainer = Trainer(model, lr=1e-3, world_size=world, force_exchange=FORCE_DIST)
    sync = barrier or torch.cuda.synchronize
    for _ in range(warmup):
        trainer.step(x, graph, batch, cond, nodes_per_graph=n)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, aux = trainer.step(x, graph, batch, cond, nodes_per_graph=n)
    sync()
    el = time.perf_counter() - t0
    return el, float(loss.item()), (f"conditional GCN-VGAE+VQ train step, {G} collated graphs x {e} edges x {n} nodes per GPU, in {args.in_dim}+{C}, "
                                    f"latent {args.latent}(+{C} decoding), {args.codes} codes, per-graph 1:1 permutation-structured negatives, Adam")


def run_cvgae_stream(args, dev, steps, warmup, fresh, rank=0, world=1, barrier=None, prefetch=True):
    """configs[3] as the minibatch STREAM the reference's loader produces (scripts/run.py:310-326: a new batch every step):
    `fresh` pre-generated minibatches (per-graph edge lists, features and conditions resident in HBM) are visited round-robin and
    EVERYTHING per-batch happens inside the timed region - device collation (CollatedBatch), both CSR pairs with their
    transposes, gcn_norm, piece lists, the layout check (one host read for all of it) - followed by the optimizer step.
    Returns (seconds, graph_build_ms measured with HIP events around the construction, last loss, label)."""
    from causal_eeg_vae_amd import ConditionalVGAEVQ
    from causal_eeg_vae_amd.collate import CollatedBatch
    from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
    from causal_eeg_vae_amd.trainer import Trainer
    G, n, e, C = args.graphs, args.nodes_per_graph, args.edges_per_graph, args.cond_dim
    batches = []
    for k in range(fresh):
        eis = [synth_event_graph(n, e, seed=((1000 + rank) * 64 + k) * 4096 + g, device=dev) for g in range(G)]
        gen = torch.Generator(device=dev).manual_seed(77 + rank + 1000 * k)
        batches.append((eis, synth_features(G * n, args.in_dim, seed=1000 + rank + 31 * k, device=dev), torch.randn(G, C, generator=gen, device=dev)))
    torch.manual_seed(0)
    model = ConditionalVGAEVQ(args.in_dim, C, args.latent, args.codes, 0.25, seed=rank).to(dev)
    with torch.no_grad():
        model.vector_quantization.embedding.weight.normal_(0, 0.1)
    trainer = Trainer(model, lr=1e-3, world_size=world, force_exchange=FORCE_DIST)
    sync = barrier or torch.cuda.synchronize
    sizes = [n] * G
    from causal_eeg_vae_amd.collate import MinibatchStream

    def loop(k0, count, timing):
        # the loader loop: collation + graph construction of every minibatch inside the loop (prefetch: minibatch k+1 is built
        # on a side stream while step k runs; the construction is still part of the timed region, only overlapped)
        ms = MinibatchStream(((batches[k % fresh][0], sizes) for k in range(k0, k0 + count)), with_dec=True, prefetch=prefetch, timing=timing)
        loss = None
        for k, (cb, graph) in zip(range(k0, k0 + count), ms):
            _, x, cond = batches[k % fresh]
            loss, _ = trainer.step(x, graph, cb.batch, cond, nodes_per_graph=n)
        return loss, ms.build_events

    loop(0, warmup, False)
    sync()
    t0 = time.perf_counter()
    loss, evs = loop(warmup, steps, True)
    sync()
    el = time.perf_counter() - t0
    build_ms = sum(a.elapsed_time(b) for a, b in evs) / max(steps, 1)   # two event pairs per minibatch (begin, finish)
    how = ("minibatches collated and built two ahead on a side stream while the steps run (MinibatchStream)" if prefetch
           else "each step collates and builds on the step's own stream")
    return el, build_ms, float(loss.item()), (f"conditional GCN-VGAE+VQ train step on a minibatch stream: {fresh} resident minibatches round-robin, {how}: "
                                              f"{G} graphs x {e} edges x {n} nodes collated on the device, both CSR pairs / transposes / gcn_norm / "
                                              f"piece lists (one host read), then the optimizer step; in {args.in_dim}+{C}, latent {args.latent}, {args.codes} codes")


def gather_ceiling_ms(graph, x_like, F):
    """The pure-gather probe (ceeg_spmm_gather_probe) on the two structures the F-wide roofline launches walk (forward CSR and
    its transpose), median of 7 launches each, plus the measured hub-piece launch of the transposed pass: the empirical
    ceiling of a whole-row gather SpMM on this graph."""
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    p = graph.gcn
    N = graph.num_nodes
    x = torch.randn(N, F, device=x_like.device)
    ws = _lib.workspace(L.ceeg_spmm_ws_bytes(N, p.nnz, F) + 4096, x.device)
    out = []
    for rp, col in ((p.rowptr, p.col), (p.t_rowptr, p.t_col)):
        ts = []
        for it in range(10):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _lib.check(L.ceeg_spmm_gather_probe(rp.data_ptr(), col.data_ptr(), x.data_ptr(), x.stride(0), N, p.nnz, F, 3, 64, ws.data_ptr(),
                                                ws.numel(), _lib.stream()), "ceeg_spmm_gather_probe")
            e.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(s.elapsed_time(e))
        ts.sort()
        out.append(ts[len(ts) // 2])
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` without a launcher around it (N > 1, WORLD_SIZE unset): start N fresh ranks with
    torch.distributed.run as CHILD processes - this process has made no GPU call and makes none -, relay rank 0's JSON line
    as this process's last stdout line and return the children's exit code (non-zero when any rank failed or no line came).
    Everything else the ranks print goes to stderr."""
    port = _free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    try:
        for line in proc.stdout:
            t = line.strip()
            if t.startswith("{") and t.endswith("}"):
                try:
                    json.loads(t)
                    line_json = t
                    continue
                except ValueError:
                    pass
            sys.stderr.write(line)
        rc = proc.wait(timeout=timeout)
    except BaseException:
        proc.kill()
        proc.wait()
        raise
    if rc != 0:
        print(f"bench.py: the {n}-rank job exited with code {rc}", file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    if line_json is None:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        return 4
    print(line_json, flush=True)
    return 0


# CEEG_FORCE_DIST=1: a ONE-rank process group on the real backend (RCCL) with every collective of the N-rank path issued -
# parameter broadcast, the per-step flat all-reduce, barriers, the report's all-gather.  Two RCCL ranks cannot share a device, so
# on a one-GPU box this is as far as RCCL itself can be exercised (profiles/r03_rehearsal_rccl_1rank.json).
FORCE_DIST = os.environ.get("CEEG_FORCE_DIST") == "1"


def rank_report(el, dev, world, backend):
    """What the process group saw, so an N-rank record checks itself: ranks, backend, the device every rank ran on and the
    per-rank time of the timed region.  Returns (max el over ranks, dict) on every rank."""
    idx = dev.index if dev.index is not None else 0
    if world == 1 and not FORCE_DIST:
        return el, {"ranks": 1, "backend": None, "devices": [idx]}
    t = torch.tensor([el, float(idx)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    els = [float(o[0]) for o in out]
    return max(els), {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "devices": sorted({int(o[1]) for o in out}),
                      "devices_per_rank": [int(o[1]) for o in out],
                      "rank_ms_per_step": {"min": None, "max": None, "_els": els}}


def _finish_report(rep, steps):
    r = rep.get("rank_ms_per_step")
    if r:
        els = r.pop("_els")
        r["min"], r["max"] = min(els) / steps * 1e3, max(els) / steps * 1e3
    return rep


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's command form is `python bench.py --gpus N ...`: become the launcher (no GPU call has been made)
        if torch.cuda.device_count() < args.gpus and os.environ.get("CEEG_SHARE_GPU") != "1":
            print(f"bench.py: --gpus {args.gpus} but {torch.cuda.device_count()} GPU(s) visible "
                  "(CEEG_SHARE_GPU=1 CEEG_DIST_BACKEND=gloo rehearses the multi-process path on fewer)", file=sys.stderr)
            sys.exit(2)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; CEEG_DIST_BACKEND=gloo + CEEG_SHARE_GPU=1 is the one-GPU rehearsal of the multi-process path
    backend = os.environ.get("CEEG_DIST_BACKEND", "nccl")
    if os.environ.get("CEEG_SHARE_GPU") == "1":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or FORCE_DIST:
        kw = {}
        if "WORLD_SIZE" not in os.environ:      # CEEG_FORCE_DIST without a launcher: a one-rank group of our own
            kw = {"init_method": f"tcp://127.0.0.1:{_free_port()}", "world_size": 1, "rank": 0}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)

    verbose = os.environ.get("CEEG_BENCH_VERBOSE") == "1"

    def note(msg):
        if verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    note("process group up")
    from causal_eeg_vae_amd import _lib, EventGraph, VGAEVQ, VectorQuantizer
    from causal_eeg_vae_amd.synth import synth_event_graph, synth_features
    from causal_eeg_vae_amd.trainer import Trainer
    _lib.check(_lib.lib().ceeg_check_device(), "ceeg_check_device")
    tuned = _lib.enable_tuned_gemms()
    note(f"tuned GEMM table active: {tuned}")

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or FORCE_DIST:
            dist.barrier()
            torch.cuda.synchronize()

    if args.workload == "vq":
        r = run_vq(dev, args.vq_dtype, args.steps, args.warmup, rank, world, barrier)
        print(json.dumps({"metric": "VQ forward latents/s", "value": r["latents_per_s"], "unit": "latents/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms"], "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.vq_dtype, "data": "synthetic",
                          "config": {"workload": f"VectorQuantizer fwd {r['shape']}"},
                          "roofline": {"bound": "mfma", "achieved": r["TFLOP/s"], "peak": r["mfma_peak_TFLOP/s"], "unit": "TFLOP/s",
                                       "frac": r["mfma_frac"], "traffic": None, "hbm_frac": r["hbm_frac"], "hbm_GB/s": r["hbm_GB/s"]}}))
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

    if args.scaling == "strong" and (args.workload != "cvgae" or args.graphs < world):
        raise SystemExit("--scaling strong: the cvgae workload with at least one graph per rank (the other workloads are one shard per GPU)")
    if args.workload == "cvgae":
        build_ms = None
        if args.fresh_batches > 0:
            if args.scaling != "weak":
                raise SystemExit("--fresh-batches: weak scaling only (every rank streams its own minibatches)")
            el, build_ms, last_loss, wl = run_cvgae_stream(args, dev, args.steps, args.warmup, args.fresh_batches, rank, world, barrier,
                                                           prefetch=args.prefetch)
        else:
            el, last_loss, wl = run_cvgae(args, dev, args.steps, args.warmup, rank, world, barrier)
        el, rep = rank_report(el, dev, world, backend)
        if rank == 0:
            total_graphs = (args.graphs // world) * world if args.scaling == "strong" else args.graphs * world
            line = {"metric": "VGAE train edges/sec", "value": total_graphs * args.edges_per_graph / (el / args.steps), "unit": "edges/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
                    "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": wl, "parallelism": f"dp{world}"}, "loss": last_loss}
            if build_ms is not None:
                line["graph_build_ms"] = build_ms
            line.update(_finish_report(rep, args.steps))
            print(json.dumps(line))
        if world > 1 or FORCE_DIST:
            dist.destroy_process_group()
        return

    N, E = args.nodes, args.edges
    ei = synth_event_graph(N, E, seed=1000 + rank, device=dev)   # shard r: seed base + r
    x = synth_features(N, args.in_dim, seed=1000 + rank, device=dev)
    graph = EventGraph(ei, N)
    graph.dec  # build the decoder structure before timing
    torch.manual_seed(0)
    model = VGAEVQ(args.in_dim, args.latent, args.codes, 0.25, seed=rank, negatives=args.negatives).to(dev)
    with torch.no_grad():
        model.vector_quantization.embedding.weight.normal_(0, 0.1)
    note("graph built")
    trainer = Trainer(model, lr=1e-3, world_size=world, force_exchange=FORCE_DIST)
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
    el, rep = rank_report(el, dev, world, backend)
    if rank == 0:
        ms = el / args.steps * 1e3
        F = 2 * args.latent
        nnz = graph.gcn.nnz
        # compulsory bytes of one SpMM pass of width F.  The two F-wide launches of a step are the layer-2
        # aggregation and its transposed pass; the latter also applies the ReLU backward mask in its epilogue (it reads the
        # [N, F] activation once more: compulsory bytes of the fused operation, counted for that launch only).
        spmm_bytes = nnz * 8 + (N + 1) * 4 + 2 * N * F * 4
        n1, ms1 = prof.get(f"spmm_F{F}", (0, 0.0))
        n2, ms2 = prof.get(f"spmm_masked_F{F}", (0, 0.0))
        launches = n1 + n2
        total_bytes = n1 * spmm_bytes + n2 * (spmm_bytes + N * F * 4)
        spmm_ms = (n1 * ms1 + n2 * ms2) / launches if launches else float("nan")
        achieved = total_bytes / launches / (spmm_ms * 1e-3) / 1e9 if launches else float("nan")
        # HBM-side bytes per launch from the rocprofv3 PMC passes (profiles/r03_traffic.json, made by scripts/pmc_traffic.py from
        # separate --pmc runs of this command); counters cannot be read from inside the process, so the figures are reported
        # only for the workload they were measured on
        traffic = dec_traffic = None
        here = os.path.dirname(os.path.abspath(__file__))
        for name in ("r03_traffic.json", "r02_spmm_traffic.json"):
            tpath = os.path.join(here, "profiles", name)
            if os.path.exists(tpath):
                t = json.load(open(tpath))
                if (t.get("nodes"), t.get("edges"), t.get("F")) == (N, E, F):
                    t = t.get("by_negatives", {}).get(args.negatives, t)
                    traffic = t.get("traffic_bytes")
                    dec_traffic = t.get("decoder_traffic_bytes")
                break
        # the decoder of the step that ran (HBM-bound; north star: "HBM GB/s ... for SpMM and the decoder"):
        # algorithmic bytes = every list entry / record / key read or written ONCE + the [N, D] gradient, over the summed
        # HIP-event time of its launches
        D_lat, Ed = args.latent, graph.dec.nnz
        dec_keys = ["decoder_fwd", "decoder_bwd", "neg_sample", "pair_bce_fwd", "pair_bce_bwd"]
        dec_ms = sum(prof[k][1] * prof[k][0] for k in dec_keys if k in prof) / max(args.steps, 1)
        if args.negatives == "uniform":
            dec_bytes = (Ed * 20 + N * 12) + (Ed * (2 * 16 + 4) + N * D_lat * 4) + E * 8 + (E * 16 + N * 4) + (E * 8 + 2 * N * D_lat * 4)
            dec_what = ("code-form decoder, positives (ceeg_decoder_codes_fwd/_bwd) + negative_sampling (ceeg_neg_sample_uniform_filtered) + "
                        "sampled pairs (ceeg_pair_bce_codes_fwd/_bwd incl. the incidence sort)")
        else:
            dec_bytes = (Ed * 20 + N * 12) + (Ed * (3 * 16 + 4) + N * D_lat * 4)
            dec_what = "code-form decoder with permutation-structured negatives (ceeg_decoder_codes_fwd/_bwd)"
        neg_label = ("1:1 permutation-structured negatives (sigma affine per step, pairs colliding with an edge masked)" if args.negatives == "sigma"
                     else "1:1 uniform negatives (negative_sampling: E pairs from ceeg_neg_sample_uniform every step, their incidences grouped by one radix sort per step)")
        line = {"metric": "VGAE train edges/sec", "value": E * world / (el / args.steps), "unit": "edges/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"GCN-VGAE+VQ train step, synthetic power-law event graph {N} nodes / {E} edges per GPU, "
                                       f"in {args.in_dim}, latent {args.latent}, {args.codes} codes, {neg_label}, Adam",
                           "parallelism": f"dp{world}",
                           "dense_products": ("fp32 library GEMMs (hipBLASLt / rocBLAS), kernels chosen by the package's TunableOp table "
                                              "for these shapes" if tuned else "fp32 library GEMMs (hipBLASLt), library heuristic")},
                "roofline": {"bound": "hbm", "kernel": f"ceeg_spmm_csr_items / ceeg_spmm_csr_masked F={F} (spmm_rows_kernel<64> + long-row pieces)",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": traffic, "launches_timed": launches, "avg_launch_ms": spmm_ms,
                             "algorithmic_bytes_per_launch": total_bytes / launches if launches else None,
                             "launches": {"aggregation": [n1, ms1], "transposed_with_relu_mask": [n2, ms2]},
                             "hbm_side_GBps": (traffic / (spmm_ms * 1e-3) / 1e9) if traffic else None,
                             "hbm_side_frac": (traffic / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None},
                "roofline_decoder": {"bound": "hbm", "kernel": dec_what, "achieved": dec_bytes / (dec_ms * 1e-3) / 1e9 if dec_ms else None,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dec_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dec_ms else None,
                                     "traffic": dec_traffic, "ms_per_step": dec_ms, "algorithmic_bytes_per_step": dec_bytes,
                                     "hbm_side_GBps": (dec_traffic / (dec_ms * 1e-3) / 1e9) if (dec_traffic and dec_ms) else None},
                "loss": float(loss.item())}
        line.update(_finish_report(rep, args.steps))
        if world == 1 and not args.no_extras:
            # Extra legs, after the headline's timed region (BASELINE metric = "edges/s; SpMM HBM GB/s; VQ MFMA util").  Each leg
            # is on its own: a failure is recorded under the leg's key and the headline still prints.
            def leg(key, fn):
                try:
                    fn()
                except Exception as ex:  # noqa: BLE001 - recorded, never hidden
                    line.setdefault("extras_failed", {})[key] = f"{type(ex).__name__}: {ex}"[:400]
                    torch.cuda.empty_cache()

            def leg_gather():
                # (1) the empirical ceiling of the roofline kernel: the same traversal with only the row gathers left
                fwd_c, tr_c = gather_ceiling_ms(graph, x, F)
                line["roofline"]["gather_only_ms"] = {"forward": fwd_c, "transposed_rows_le_256": tr_c}
                line["roofline"]["launch_ms_over_gather_only_ms"] = spmm_ms / ((fwd_c + tr_c) / 2)

            def leg_other():
                # (2) the same step with the other sampler (sigma = permutation-structured negatives inside the fused decoder)
                other = "uniform" if args.negatives == "sigma" else "sigma"
                model_u = VGAEVQ(args.in_dim, args.latent, args.codes, 0.25, seed=rank, negatives=other).to(dev)
                model_u.load_state_dict(model.state_dict())
                tr_u = Trainer(model_u, lr=1e-3, world_size=1)
                for _ in range(args.warmup):
                    tr_u.step(x, graph)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    lu, _ = tr_u.step(x, graph)
                torch.cuda.synchronize()
                ms_u = (time.perf_counter() - t1) / args.steps * 1e3
                line["other_negatives"] = {"negatives": other, "ms_per_step": ms_u, "edges_per_s": E / (ms_u * 1e-3), "steps": args.steps,
                                           "loss": float(lu.item())}

            def leg_bf16():
                # (2b) the opt-in reduced-precision step beside the fp32 headline (north star: rtol 1e-2 for bf16): bf16 storage of the
                # encoder's activations (every gathered row half the bytes), bf16 MFMA products, fp32 sums; same sampler as the headline
                model_b = VGAEVQ(args.in_dim, args.latent, args.codes, 0.25, seed=rank, negatives=args.negatives, compute_dtype=torch.bfloat16).to(dev)
                model_b.load_state_dict(model.state_dict())
                tr_b = Trainer(model_b, lr=1e-3, world_size=1)
                x16 = x.to(torch.bfloat16)
                for _ in range(args.warmup):
                    tr_b.step(x16, graph)
                torch.cuda.synchronize()
                _lib.PROFILE = {}
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    lb, _ = tr_b.step(x16, graph)
                torch.cuda.synchronize()
                ms_b = (time.perf_counter() - t1) / args.steps * 1e3
                pb = _lib.profile_summary()
                _lib.PROFILE = None
                b1, t1_ = pb.get(f"spmm16_F{F}", (0, 0.0))
                b2, t2_ = pb.get(f"spmm16_masked_F{F}", (0, 0.0))
                by = nnz * 8 + (N + 1) * 4 + 2 * N * F * 2
                tot_b, tot_t = b1 * by + b2 * (by + N * F * 2), b1 * t1_ + b2 * t2_
                line["bf16_step"] = {"ms_per_step": ms_b, "edges_per_s": E / (ms_b * 1e-3), "steps": args.steps, "loss": float(lb.item()),
                                     "what": "encoder activations stored as bf16, bf16 MFMA products, fp32 sums; parameters, mu|logstd, quantiser and decoder fp32",
                                     "roofline": {"bound": "hbm", "kernel": f"ceeg_spmm_csr_cv_bf16 F={F} (aggregation + transposed pass with ReLU mask)",
                                                  "achieved": tot_b / (tot_t * 1e-3) / 1e9 if tot_t else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": tot_b / (tot_t * 1e-3) / 1e9 / HBM_PEAK_GBS if tot_t else None, "traffic": None,
                                                  "launches": {"aggregation": [b1, t1_], "transposed_with_relu_mask": [b2, t2_]},
                                                  "algorithmic_bytes_per_launch": tot_b / (b1 + b2) if b1 + b2 else None}}

            leg("gather_only", leg_gather)
            leg("other_negatives", leg_other)
            leg("bf16_step", leg_bf16)
            trainer = model = graph = None
            torch.cuda.empty_cache()
            # (3) BASELINE.json configs[2] (VQ 4 Mi x 512 x 256) in both dtypes and configs[3] (64 collated graphs x 100k edges)
            leg("vq_bf16", lambda: line.__setitem__("vq_bf16", run_vq(dev, "bf16", 20, 3)))
            leg("vq_f32", lambda: line.__setitem__("vq_f32", run_vq(dev, "f32", 10, 2)))
            torch.cuda.empty_cache()

            def leg_cvgae():
                el_c, loss_c, wl_c = run_cvgae(args, dev, 20, 3)
                line["cvgae"] = {"ms_per_step": el_c / 20 * 1e3, "edges_per_s": args.graphs * args.edges_per_graph / (el_c / 20), "steps": 20,
                                 "workload": wl_c, "loss": loss_c}

            leg("cvgae", leg_cvgae)

            def leg_stream():
                # the loader's default (MinibatchStream(prefetch=True): construction two minibatches ahead on a side stream) is the
                # figure; the plain loop on the step's own stream is timed beside it - its graph_build_ms is the construction alone
                el_p, _, loss_p, wl_p = run_cvgae_stream(args, dev, 20, 3, 4, prefetch=True)
                el_s, build_ms, _, _ = run_cvgae_stream(args, dev, 20, 3, 4, prefetch=False)
                line["cvgae_stream"] = {"ms_per_step": el_p / 20 * 1e3, "edges_per_s": args.graphs * args.edges_per_graph / (el_p / 20),
                                        "serial_loop_ms_per_step": el_s / 20 * 1e3, "graph_build_ms": build_ms,
                                        "steps": 20, "workload": wl_p, "loss": loss_p}

            leg("cvgae_stream", leg_stream)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args)
            except Exception as ex:  # noqa: BLE001
                line.setdefault("extras_failed", {})["cpu_baseline"] = f"{type(ex).__name__}: {ex}"[:400]
        print(json.dumps(line))
    if world > 1 or FORCE_DIST:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
