"""GPU parity: reparam+KL, decoder (edge-list and fused), GCNConv, and the full VGAE+VQ step vs the oracle.

Every float comparison is `_close` = rtol 1e-5, atol 1e-6 max|ref| (the north-star tolerance) against the oracle run in
float64 on the same fp32 inputs (tests/_parity.py); no assertion in this file is conditional on another one's outcome."""
import numpy as np
import pytest
import torch

from _parity import close as _close, flipped_rows_are_near_ties, grads_close, to64

pytestmark = pytest.mark.gpu


def _graph(N, E, seed):
    rng = np.random.default_rng(seed)
    dst = rng.integers(1, N, 2 * E)
    src = (dst * rng.random(2 * E) ** 2).astype(np.int64)  # DAG, src < dst, early nodes are hubs
    key = np.unique(dst * N + src)[:E]
    rng.shuffle(key)
    return np.stack([key % N, key // N]).astype(np.int64)


@pytest.mark.parametrize("N,D,fused,training", [(1000, 128, True, True), (333, 20, False, True), (257, 7, False, True), (500, 64, True, False)])
def test_reparam_kl_fwd_bwd(N, D, fused, training):
    from causal_eeg_vae_amd.models import _ReparamKL
    from oracle import ref_torch
    torch.manual_seed(N)
    ml = torch.randn(N, 2 * D) * 2
    ml[::7, D:] = 11.5  # exercise the MAX_LOGSTD clamp (zero gradient there)
    eps = torch.randn(N, D)
    w = torch.randn(N, D)
    a = ml.double().requires_grad_(True)
    mu, ls = a[:, :D], a[:, D:].clamp(max=ref_torch.MAX_LOGSTD)
    z_ref = ref_torch.reparametrize(mu, ls, eps.double()) if training else mu
    kl_ref = ref_torch.kl_loss(mu, ls)
    ((z_ref * w.double()).sum() + 3.0 * kl_ref).backward()
    b = ml.clone().cuda().requires_grad_(True)
    if fused:
        z, kl = _ReparamKL.apply(b, None, eps.cuda(), 0, training, False)
    else:
        z, kl = _ReparamKL.apply(b[:, :D], b[:, D:], eps.cuda(), 0, training, False)
    ((z * w.cuda()).sum() + 3.0 * kl).backward()
    _close(z, z_ref)
    _close(kl, kl_ref)
    _close(b.grad, a.grad)


def test_reparam_log_var_variant():
    from causal_eeg_vae_amd.models import _ReparamKL
    mu = torch.randn(64, 16); lv = torch.randn(64, 16); eps = torch.randn(64, 16)
    z, _ = _ReparamKL.apply(mu.cuda(), lv.cuda(), eps.cuda(), 0, True, True)
    _close(z, mu.double() + eps.double() * torch.exp(0.5 * lv.double()))  # scripts/models.py:38-43


@pytest.mark.parametrize("D", [128, 16, 256])
def test_edge_bce_and_recon_loss(D):
    from causal_eeg_vae_amd import VGAE, InnerProductDecoder
    from oracle import ref_torch
    N, E = 2000, 30000
    ei = torch.from_numpy(_graph(N, E, D))
    neg = torch.from_numpy(np.random.default_rng(D).integers(0, N, (2, E)))
    z0 = torch.randn(N, D) * (2.0 / D ** 0.5)
    z0[:5] *= 12  # saturate some sigmoids: the literal +EPS form must match
    a = z0.double().requires_grad_(True)
    ref = ref_torch.recon_loss(a, ei, neg)
    ref.backward()
    b = z0.clone().cuda().requires_grad_(True)
    model = VGAE(torch.nn.Identity())
    loss = model.recon_loss(b, ei.cuda(), neg.cuda())
    loss.backward()
    _close(loss, ref)
    _close(b.grad, a.grad)
    p = InnerProductDecoder()(b.detach(), ei.cuda())
    _close(p, ref_torch.decode(z0.double(), ei))
    v = InnerProductDecoder()(b.detach(), ei.cuda(), sigmoid=False)   # the raw inner product, finite where the sigmoid saturates
    _close(v, ref_torch.decode(z0.double(), ei, sigmoid=False))


@pytest.mark.parametrize("N,E,D", [(3000, 50000, 128), (500, 20000, 32), (2000, 3000, 256), (700, 30000, 64), (900, 9000, 4),
                                   (1500, 40000, 8), (1200, 20000, 16), (1000, 150000, 100), (96, 145, 8), (5, 6, 4), (17, 40, 128)])
def test_fused_decoder_vs_oracle(N, E, D):
    from causal_eeg_vae_amd import EventGraph, sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    from oracle import ref_torch
    ei_np = _graph(N, E, N)
    g = EventGraph(torch.from_numpy(ei_np).cuda(), N)
    a_, b_, ainv = sigma_params(N, 3, 17)
    assert (a_ * ainv) % N == 1
    rowptr, col, perm = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, sig, valid = ref_torch.sigma_negatives_np(rowptr, col, N, a_, b_)
    assert 0 < (~valid).sum() < len(valid) or N > 1000 or N < 100  # the dense case must exercise the collision mask
    z0 = torch.randn(N, D) * (2.0 / D ** 0.5)
    a = z0.double().requires_grad_(True)
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    pos_ei = torch.from_numpy(np.stack([col, rows]).astype(np.int64))
    neg_ei = torch.from_numpy(np.stack([s, sig]).astype(np.int64))[:, torch.from_numpy(valid)]
    ref = ref_torch.recon_loss(a, pos_ei, neg_ei)
    (ref * 1.7).backward()
    b = z0.clone().cuda().requires_grad_(True)
    rec, sums = _FusedDecoder.apply(b, g, a_, b_, ainv)
    (rec * 1.7).backward()
    assert int(sums[2].item()) == int(valid.sum())  # valid-negative count is exact
    _close(rec, ref)
    _close(b.grad, a.grad)
    rec2, _ = _FusedDecoder.apply(b, g, a_, b_, ainv)
    assert torch.equal(rec, rec2)
    # the four-pass backward (no row-side sums kept by the forward) gives the same gradient
    c = z0.clone().cuda().requires_grad_(True)
    rec3, _ = _FusedDecoder.apply(c, g, a_, b_, ainv, False)
    (rec3 * 1.7).backward()
    assert torch.equal(rec3, rec)
    _close(c.grad, a.grad)


def test_gcnconv_fwd_bwd_and_state_dict():
    from causal_eeg_vae_amd import GCNConv, EventGraph
    from oracle import ref_torch
    N, E, Fi, Fo = 3000, 40000, 48, 128
    ei = torch.from_numpy(_graph(N, E, 1))
    ei = torch.cat([ei, torch.tensor([[5, 5, 9], [5, 5, 9]])], dim=1)  # pre-existing (duplicated) self-loops
    ew = torch.rand(ei.shape[1]) + 0.5
    ref = ref_torch.RefGCNConv(Fi, Fo)
    with torch.no_grad():
        ref.bias.normal_()
    conv = GCNConv(Fi, Fo).cuda()
    conv.load_state_dict(ref.state_dict())  # same keys: lin.weight, bias
    ref = to64(ref)
    x = torch.randn(N, Fi)
    for w in (None, ew):
        a = x.double().requires_grad_(True)
        out_ref = ref(a, ei, None if w is None else w.double())
        out_ref.square().sum().backward()
        b = x.clone().cuda().requires_grad_(True)
        g = EventGraph(ei.cuda(), N, None if w is None else w.cuda())
        out = conv(b, g)
        out.square().sum().backward()
        _close(out, out_ref)
        _close(b.grad, a.grad)
        grads_close(conv, ref)
        conv.zero_grad(); ref.zero_grad()


def _check_step_against_fp64_oracle(ref, model, loss, aux, args, kwargs):
    """The fp64 oracle runs twice: unforced, to check that every code the GPU chose is the exact argmin or a near tie
    (flipped_rows_are_near_ties), then with the GPU's codes forced, so that losses and EVERY parameter gradient are compared
    unconditionally on identical discrete choices."""
    ref = to64(ref)
    x, ei, eps, neg_ei, valid = args
    kw = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kwargs.items()}
    with torch.no_grad():
        _, aux0 = ref(x.double(), ei, eps.double(), neg_ei, valid, **kw)
    _close(aux["z"], aux0["z"])
    n_flip = flipped_rows_are_near_ties(aux["idx"], ref.vector_quantization)
    assert n_flip <= max(1, x.shape[0] // 1000)
    loss_ref, aux_ref = ref(x.double(), ei, eps.double(), neg_ei, valid, force_idx=aux["idx"].cpu(), **kw)
    loss_ref.backward()
    for k in ("kl", "rec", "vq", "ppl"):
        _close(aux[k], aux_ref[k], what=k)
    _close(loss, loss_ref)
    grads_close(model, ref)


@pytest.mark.parametrize("Fi,D,fused_head", [(32, 16, "1"), (16, 16, "1"), (16, 16, "0"), (12, 64, "1")])
def test_full_step_vs_oracle(Fi, D, fused_head, monkeypatch):
    """GCN-VGAE + VQ training step: loss, code indices and every parameter gradient vs the CPU restatement.  Fi < 2 D runs the
    first layer aggregate-first and the step as three fused autograd nodes (ReLU mask in the transposed aggregation's epilogue,
    head bias gradient out of the fused reparam / VQ backward); CEEG_FUSED_HEAD=0 keeps the layer-by-layer form; Fi = 2 D is
    the linear-first first layer."""
    from causal_eeg_vae_amd import VGAEVQ, EventGraph, sigma_params
    from oracle import ref_torch
    monkeypatch.setenv("CEEG_FUSED_HEAD", fused_head)
    N, E, n_e = 4000, 64000, 64
    ei_np = _graph(N, E, 2)
    torch.manual_seed(0)
    ref = ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = VGAEVQ(Fi, D, n_e, 0.25).cuda()
    model.load_state_dict(ref.state_dict())
    x = torch.randn(N, Fi)
    eps = torch.randn(N, D)
    sig = sigma_params(N, 0, 5)
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, sg, valid = ref_torch.sigma_negatives_np(rowptr, col, N, sig[0], sig[1])
    neg_ei = torch.from_numpy(np.stack([s, sg]).astype(np.int64))
    ei = torch.from_numpy(ei_np)
    g = EventGraph(ei.cuda(), N)
    loss, aux = model(x.cuda(), g, step=5, eps=eps.cuda(), sigma=sig)
    loss.backward()
    _check_step_against_fp64_oracle(ref, model, loss, aux, (x, ei, eps, neg_ei, torch.from_numpy(valid)), {})


@pytest.mark.parametrize("forms", ["pairs", "pairs_two_launch", "codes", "rows"])
def test_full_step_uniform_negatives_vs_oracle(forms, monkeypatch):
    """forms = "pairs" (the default): the pairs are drawn INSIDE the pair-list forward (ceeg_neg_sample_pair_bce_codes_fwd) and both
    pair sets share one gradient kernel (ceeg_decoder_codes_bwd_pairs); "pairs_two_launch" (CEEG_PAIR_MERGED=0): separate sampler
    launch and a second gradient kernel accumulating into dz - loss, samples and every gradient must equal "pairs" BIT FOR BIT;
    both as "codes", but the sampled negatives go through ceeg_pair_bce_codes_* (one radix sort of
    the 2E incidences, LDS codebook slab; no CSR of the sampled list); "codes": positives through the code-form fused decoder with its negatives switched off, sampled negatives through
    the Gram-matrix edge kernel and table-form SpMMs; "rows": every pair list through the row-gather edge kernel (table rows)
    and SpMMs - the checker form.  Both against the fp64 oracle.
    The literal path of the north star: recon_loss over the positive edges and E pairs drawn by negative_sampling
    (ceeg_neg_sample_uniform) every step.  The GPU's samples are bit-exact with the C oracle's Philox stream; loss and every
    parameter gradient against the fp64 oracle fed the same explicit negatives; a second call with given negatives of
    which some are (-1, -1) (samples the sampler gave up on) checks that such pairs are skipped and not counted."""
    from causal_eeg_vae_amd import VGAEVQ, EventGraph
    from oracle import c_oracle, ref_torch
    if forms == "rows":
        monkeypatch.setenv("CEEG_DEC_CODES", "0")
        monkeypatch.setenv("CEEG_EDGE_CODES", "0")
    if not forms.startswith("pairs"):
        monkeypatch.setenv("CEEG_PAIR_CODES", "0")
    monkeypatch.setenv("CEEG_PAIR_MERGED", "0" if forms == "pairs_two_launch" else "1")
    N, E, Fi, D, n_e = 3000, 48000, 32, 64, 128
    ei_np = _graph(N, E, 7)
    torch.manual_seed(0)
    ref = ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = VGAEVQ(Fi, D, n_e, 0.25, seed=3, negatives="uniform").cuda()
    model.load_state_dict(ref.state_dict())
    x, eps = torch.randn(N, Fi), torch.randn(N, D)
    ei = torch.from_numpy(ei_np)
    g = EventGraph(ei.cuda(), N)
    loss, aux = model(x.cuda(), g, step=5, eps=eps.cuda())
    loss.backward()
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, d = c_oracle.neg_sample_uniform(rowptr, col, N, E, model.negative_seed(5))
    np.testing.assert_array_equal(aux["neg_edge_index"].cpu().numpy(), np.stack([s, d]))     # negative-sample indices bit-exact
    assert (s >= 0).all() and int(aux["dec_sums"][2]) == E
    neg_ei = torch.from_numpy(np.stack([s, d]).astype(np.int64))
    _check_step_against_fp64_oracle(ref, model, loss, aux, (x, ei, eps, neg_ei, None), {})
    if forms == "pairs_two_launch":   # the merged form on the same inputs: bitwise the same step
        grads = [p.grad.clone() for p in model.parameters()]
        model.zero_grad(set_to_none=True)
        monkeypatch.setenv("CEEG_PAIR_MERGED", "1")
        loss_m, aux_m = model(x.cuda(), g, step=5, eps=eps.cuda())
        loss_m.backward()
        assert torch.equal(loss_m, loss) and torch.equal(aux_m["neg_edge_index"], aux["neg_edge_index"])
        assert torch.equal(aux_m["dec_sums"], aux["dec_sums"])
        for p, g0 in zip(model.parameters(), grads):
            assert torch.equal(p.grad, g0)
        monkeypatch.setenv("CEEG_PAIR_MERGED", "0")
    # given negatives with holes
    model.zero_grad(set_to_none=True)
    s2, d2 = s.copy(), d.copy()
    s2[::7] = -1; d2[::7] = -1
    neg2 = torch.from_numpy(np.stack([s2, d2]))
    loss2, aux2 = model(x.cuda(), g, step=5, eps=eps.cuda(), neg_edge_index=neg2.cuda())
    loss2.backward()
    assert int(aux2["dec_sums"][2]) == int((s2 >= 0).sum())
    _check_step_against_fp64_oracle(ref, model, loss2, aux2, (x, ei, eps, neg2.to(torch.int64), torch.from_numpy(s2 >= 0)), {})


@pytest.mark.parametrize("negatives", ["uniform", "sigma"])
def test_full_step_vs_oracle_on_a_trained_state(negatives):
    """The step check of test_full_step_vs_oracle / _uniform_negatives_vs_oracle, but on the state 40 optimizer steps leave behind
    (lr 2e-2) instead of the random initialisation: codebook and encoder have moved towards each other, code use is uneven,
    the decoder's logits have grown.  Loss, its parts and EVERY parameter gradient against the fp64 oracle loaded with the
    trained weights."""
    from causal_eeg_vae_amd import VGAEVQ, EventGraph, sigma_params
    from causal_eeg_vae_amd.trainer import Trainer
    from oracle import c_oracle, ref_torch
    N, E, Fi, D, n_e = 3000, 48000, 32, 64, 128
    ei_np = _graph(N, E, 11)
    torch.manual_seed(1)
    ref = ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = VGAEVQ(Fi, D, n_e, 0.25, seed=3, negatives=negatives).cuda()
    model.load_state_dict(ref.state_dict())
    x = torch.randn(N, Fi)
    ei = torch.from_numpy(ei_np)
    g = EventGraph(ei.cuda(), N)
    tr = Trainer(model, lr=2e-2)
    first = None
    for _ in range(40):
        l, aux_t = tr.step(x.cuda(), g)
        first = float(l) if first is None else first
    assert float(l) < first                                   # it did train
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    model.zero_grad(set_to_none=True)
    eps = torch.randn(N, D)
    step = 77
    if negatives == "uniform":
        loss, aux = model(x.cuda(), g, step=step, eps=eps.cuda())
        loss.backward()
        rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
        s, d = c_oracle.neg_sample_uniform(rowptr, col, N, E, model.negative_seed(step))
        np.testing.assert_array_equal(aux["neg_edge_index"].cpu().numpy(), np.stack([s, d]))
        neg_ei, valid = torch.from_numpy(np.stack([s, d]).astype(np.int64)), None
    else:
        sig = sigma_params(N, 0, step)
        loss, aux = model(x.cuda(), g, step=step, eps=eps.cuda(), sigma=sig)
        loss.backward()
        rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
        s, sg, v = ref_torch.sigma_negatives_np(rowptr, col, N, sig[0], sig[1])
        neg_ei, valid = torch.from_numpy(np.stack([s, sg]).astype(np.int64)), torch.from_numpy(v)
    _check_step_against_fp64_oracle(ref, model, loss, aux, (x, ei, eps, neg_ei, valid), {})


@pytest.mark.parametrize("N,E,D,n_e,negative", [(1000, 20000, 64, 100, True), (777, 5000, 128, 512, False), (64, 0, 64, 8, True),
                                                 (130, 9000, 256, 33, True), (5, 40, 192, 576, False)])
def test_pair_list_code_form_vs_fp64(N, E, D, n_e, negative):
    """ceeg_pair_bce_codes_fwd / _bwd on an arbitrary pair list (duplicates, i == j pairs, holes, ids >= N, one hub node with
    thousands of incidences, N not a multiple of 64, the empty list): sum, count and dz against fp64 torch on the same list;
    twice in a row bit-identical (the sort is stable: fixed summation order)."""
    from causal_eeg_vae_amd.models import _PairBCECodes
    rng = np.random.default_rng(N + E)
    W = torch.from_numpy(rng.normal(0, 0.4, (n_e, D)).astype(np.float32))
    code = torch.from_numpy(rng.integers(0, n_e, N).astype(np.int32))
    i, j = rng.integers(0, N, E), rng.integers(0, N, E)
    if E:
        i[: E // 16] = 5                     # hub: a 16th of the pairs share one endpoint (a fp32 chain of ~1000 terms; the sampler
        j[E // 4: E // 4 + 10] = i[E // 4: E // 4 + 10]     # itself never makes one: rows of a uniform draw are Poisson(2E/N));  i == j
        i[-7:] = i[-14:-7]; j[-7:] = j[-14:-7]              # duplicates
        i[3::11] = -1; j[3::11] = -1          # holes
        j[4::97] = N + 3                      # out of range: skipped as well
    ei = torch.from_numpy(np.stack([i, j]).astype(np.int64))
    ok = torch.from_numpy((i >= 0) & (j >= 0) & (i < N) & (j < N))
    outs = []
    for _ in range(2):
        z = W.cuda()[code.cuda().long()].requires_grad_(True)
        s, c = _PairBCECodes.apply(z, ei.cuda(), negative, W.cuda(), code.cuda())
        (s * 0.37).backward()
        outs.append((s.detach().cpu(), c.cpu(), z.grad.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    z64 = W.double()[code.long()].requires_grad_(True)
    ii, jj = ei[0][ok], ei[1][ok]
    from oracle import ref_torch
    p = ref_torch.decode(z64, torch.stack([ii, jj]))          # float64 dots, the fp32 sigmoid's roundings (oracle/ref_torch.py)
    terms = -torch.log(1 - p + 1e-15) if negative else -torch.log(p + 1e-15)
    ref = terms.sum()
    (ref * 0.37).backward()
    assert int(outs[0][1]) == int(ok.sum())
    _close(outs[0][0], ref, what="sum")
    _close(outs[0][2], z64.grad, what="dz")


@pytest.mark.parametrize("per_graph,D,n_e", [(True, 12, 32), (False, 12, 32), (True, 64, 100), (False, 64, 100)])
def test_conditional_batched_step_vs_oracle(per_graph, D, n_e):
    """Config 4's form: 8 collated graphs, per-graph condition c[batch] concatenated to the encoder
    input and to the latent before decoding; negatives inside each graph (block permutation) or over the whole batch."""
    from causal_eeg_vae_amd import ConditionalVGAEVQ, EventGraph, sigma_params
    from causal_eeg_vae_amd.synth import collate_graphs
    from oracle import ref_torch
    G, n, e, Fi, C = 8, 500, 6000, 28, 4   # D = 64 with per-graph negatives: the code-form decoder with the condition as a row addend
    eis = [torch.from_numpy(_graph(n, e, 10 + g)) for g in range(G)]
    ei, batch, N = collate_graphs(eis, [n] * G)
    assert N == G * n and int(batch[-1]) == G - 1
    torch.manual_seed(1)
    ref = ref_torch.RefVGAEVQ(Fi + C, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = ConditionalVGAEVQ(Fi, C, D, n_e, 0.25).cuda()
    model.load_state_dict(ref.state_dict())
    x, cond, eps = torch.randn(N, Fi), torch.randn(G, C), torch.randn(N, D)
    mod = n if per_graph else N
    sig = sigma_params(mod, 0, 3)
    ei_np = ei.numpy()
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, sg, valid = ref_torch.sigma_negatives_np(rowptr, col, N, sig[0], sig[1], mod)
    if per_graph:
        assert np.array_equal(s // n, sg // n)  # every negative pair stays inside its graph
    neg_ei = torch.from_numpy(np.stack([s, sg]).astype(np.int64))
    g = EventGraph(ei.cuda(), N)
    loss, aux = model(x.cuda(), g, batch.cuda(), cond.cuda(), step=3, eps=eps.cuda(), sigma=sig, nodes_per_graph=n if per_graph else None)
    loss.backward()
    assert int(aux["dec_sums"][2].item()) == int(valid.sum())
    _check_step_against_fp64_oracle(ref, model, loss, aux, (x, ei, eps, neg_ei, torch.from_numpy(valid)), dict(batch=batch, condition=cond))


@pytest.mark.parametrize("per_graph", [True, False])
def test_conditional_step_without_materialised_condition_rows(per_graph):
    """A condition that carries no gradient never becomes an [N, C] tensor: cat((x, c[batch])) is one pass (ceeg_concat_condition) and
    the decoder's per-row addend is |c_graph|^2 gathered through batch.  Same loss and same parameter gradients, bit for bit, as the
    form that materialises c[batch] (taken when the condition requires a gradient); the kernel alone equals torch.cat."""
    from causal_eeg_vae_amd import ConditionalVGAEVQ, EventGraph, _lib
    from causal_eeg_vae_amd.synth import collate_graphs
    G, n, e, Fi, C, D, n_e = 6, 400, 5000, 28, 4, 64, 100
    eis = [torch.from_numpy(_graph(n, e, 30 + g)) for g in range(G)]
    ei, batch, N = collate_graphs(eis, [n] * G)
    torch.manual_seed(2)
    model = ConditionalVGAEVQ(Fi, C, D, n_e, 0.25).cuda()
    x, cond = torch.randn(N, Fi, device="cuda"), torch.randn(G, C, device="cuda")
    g = EventGraph(ei.cuda(), N)
    b = batch.cuda()
    outs = []
    for with_grad in (False, True):
        model.zero_grad(set_to_none=True)
        c = cond.clone().requires_grad_(with_grad)
        loss, aux = model(x, g, b, c, step=3, nodes_per_graph=n if per_graph else None)
        loss.backward()
        outs.append((loss.detach().clone(), aux["idx"].clone(), [p.grad.clone() for p in model.parameters()]))
        if with_grad:
            assert c.grad is not None and bool(torch.isfinite(c.grad).all())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for p0, p1 in zip(outs[0][2], outs[1][2]):
        assert torch.equal(p0, p1)
    out = torch.empty(N, Fi + C, device="cuda")
    L = _lib.lib()
    _lib.check(L.ceeg_concat_condition(x.data_ptr(), x.stride(0), N, Fi, cond.data_ptr(), C, G, b.data_ptr(), out.data_ptr(), _lib.stream()),
               "ceeg_concat_condition")
    assert torch.equal(out, torch.cat((x, cond[b]), dim=1))
    assert L.ceeg_concat_condition(x.data_ptr(), x.stride(0), N, Fi, cond.data_ptr(), 3, G, b.data_ptr(), out.data_ptr(), _lib.stream()) == -1


@pytest.mark.parametrize("N,E,D,n_e", [(3000, 50000, 128, 512), (1000, 30000, 32, 16)])
def test_fused_decoder_codebook_indexed_equals_expanded(N, E, D, n_e, monkeypatch):
    """Quantised latent passed as (codebook, int32 code per node): same loss, counts and gradient - bit for bit - as
    the expanded [N, D] matrix (include/ceeg.h, zidx of ceeg_decoder_fused_fwd / _bwd)."""
    from causal_eeg_vae_amd import EventGraph, sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    monkeypatch.setenv("CEEG_DEC_CODES", "0")  # this test is about the zidx form of the row-gather kernels (decoder.hip)
    g = EventGraph(torch.from_numpy(_graph(N, E, 5)).cuda(), N)
    a_, b_, ainv = sigma_params(N, 1, 9)
    torch.manual_seed(3)
    W = (torch.randn(n_e, D) * (2.0 / D ** 0.5)).cuda()
    idx = torch.randint(0, n_e, (N,), dtype=torch.int32).cuda()
    zq = W[idx.long()]
    for row_sums in (True, False):
        x1 = zq.clone().requires_grad_(True)
        r1, s1 = _FusedDecoder.apply(x1, g, a_, b_, ainv, row_sums)
        (r1 * 0.7).backward()
        x2 = zq.clone().requires_grad_(True)
        r2, s2 = _FusedDecoder.apply(x2, g, a_, b_, ainv, row_sums, None, W, idx)
        (r2 * 0.7).backward()
        assert torch.equal(r1, r2) and torch.equal(s1, s2)
        assert torch.equal(x1.grad, x2.grad)
    # the fused two-sum backward kernel == the same sums run as two SpMM passes, bit for bit (hub rows included)
    import os
    grads = []
    try:
        os.environ["CEEG_DEC_BWD_GROUPED"] = "0"   # the one-row-per-wave form shares the SpMM's summation order
        for env in ("0", "1"):
            os.environ["CEEG_DEC_BWD_TWO_PASS"] = env
            x3 = zq.clone().requires_grad_(True)
            r3, _ = _FusedDecoder.apply(x3, g, a_, b_, ainv, True, None, W, idx)
            (r3 * 0.7).backward()
            grads.append(x3.grad.clone())
    finally:
        os.environ.pop("CEEG_DEC_BWD_TWO_PASS", None)
        os.environ.pop("CEEG_DEC_BWD_GROUPED", None)
    assert torch.equal(grads[0], grads[1])
    # default (row per lane group) form: same sums, other order - both within tolerance of the exact (fp64) gradient
    from oracle import ref_torch
    ei_np = g.edge_index.cpu().numpy()
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s_, sg_, valid = ref_torch.sigma_negatives_np(rowptr, col, N, a_, b_)
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    a64 = zq.cpu().double().requires_grad_(True)
    ref = ref_torch.recon_loss(a64, torch.from_numpy(np.stack([col, rows]).astype(np.int64)),
                               torch.from_numpy(np.stack([s_, sg_]).astype(np.int64))[:, torch.from_numpy(valid)])
    (ref * 0.7).backward()
    _close(r2, ref)
    _close(x2.grad, a64.grad)
    _close(grads[0], a64.grad)


def test_config1_plumbing_csv_to_training(tmp_path):
    """BASELINE config 1 as plumbing: an event-graph CSV pair of the reference's schema and size (97 node rows, 234
    edge rows; synthetic content) -> ingestion -> `entities`-weighted GCNConv vs the oracle -> a few VGAE+VQ
    optimizer steps on the tiny graph (N not a multiple of anything, isolated nodes, DAG)."""
    pd = pytest.importorskip("pandas")
    from causal_eeg_vae_amd import EventGraph, GCNConv, VGAEVQ, ingest
    from causal_eeg_vae_amd.trainer import Trainer
    from oracle import ref_torch
    rng = np.random.default_rng(0)
    n, rows = 97, 234
    ids = rng.permutation(200)[:n]
    nodes = pd.DataFrame({"title": [f"t{i}" for i in range(n)], "text": "x", "publish_date": "d", "url": "u", "id": ids})
    nodes.loc[3, "text"] = np.nan
    dst = rng.integers(1, n, rows)
    src = (dst * rng.random(rows) ** 2).astype(np.int64)
    rep = rng.integers(0, 40, rows)
    src, dst = np.where(rep < 8, src[rep], src), np.where(rep < 8, dst[rep], dst)
    edges = pd.DataFrame({"old_id": ids[src], "new_id": ids[dst], "entity": "e", "entity_type": "LOC", "date_diff": 1})
    nodes.to_csv(tmp_path / "p_nodes.csv", index=False)
    edges.to_csv(tmp_path / "p_edges.csv", index=False)
    g = ingest.load_event_graph_csv(str(tmp_path / "p_nodes.csv"), str(tmp_path / "p_edges.csv"))
    N = g["num_nodes"]
    assert N == 96 and g["entities"].max() >= 3
    keep = g["edge_index"][0] != g["edge_index"][1]
    ei = torch.from_numpy(g["edge_index"][:, keep])
    ew = torch.from_numpy(g["entities"][keep]).float()
    graph = EventGraph(ei.cuda(), N, ew.cuda())
    torch.manual_seed(0)
    x = torch.randn(N, 32)
    ref = ref_torch.RefGCNConv(32, 16)
    conv = GCNConv(32, 16).cuda()
    conv.load_state_dict(ref.state_dict())
    _close(conv(x.cuda(), graph), ref(x, ei, ew))
    model = VGAEVQ(32, 8, 16, 0.25).cuda()
    tr = Trainer(model, lr=1e-2)
    losses = [float(tr.step(x.cuda(), graph)[0]) for _ in range(30)]
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])


@pytest.mark.parametrize("N,E,Fi,C,H,concat", [(3000, 40000, 48, 64, 1, True), (1500, 20000, 32, 16, 4, True), (800, 12000, 20, 8, 2, False),
                                               (2000, 300000, 16, 128, 1, True), (50, 60, 8, 4, 1, True), (1200, 90000, 24, 256, 1, True),
                                               (900, 9000, 12, 192, 2, True), (300, 4000, 40, 2048, 1, True), (400, 30000, 16, 640, 2, True),
                                               (250, 2000, 24, 1100, 1, False)])
def test_gatconv_fwd_bwd_vs_oracle(N, E, Fi, C, H, concat):
    """GATConv: output and every gradient (input, transform, both attention vectors, bias) against the
    written-out PyG formulas; rows longer than 64 entries (multi-batch softmax) in the dense case; pre-existing
    self-loops are replaced by one per node."""
    from causal_eeg_vae_amd import GATConv
    from oracle import ref_torch
    ei = torch.from_numpy(_graph(N, E, N + C))
    ei = torch.cat([ei, torch.tensor([[5, 7], [5, 7]])], dim=1)
    torch.manual_seed(C)
    ref = ref_torch.RefGATConv(Fi, C, H, concat)
    with torch.no_grad():
        ref.bias.normal_()
    conv = GATConv(Fi, C, heads=H, concat=concat).cuda()
    conv.load_state_dict(ref.state_dict())
    ref = to64(ref)
    x = torch.randn(N, Fi)
    r = torch.randn(N, H * C if concat else C)
    a = x.double().requires_grad_(True)
    out_ref = ref(a, ei)
    (out_ref * r.double()).sum().backward()
    b = x.clone().cuda().requires_grad_(True)
    out = conv(b, ei.cuda())
    (out * r.cuda()).sum().backward()
    _close(out, out_ref)
    _close(b.grad, a.grad)
    grads_close(conv, ref)
    grads = {name: p.grad.clone() for name, p in conv.named_parameters()}
    bg = b.grad.clone()
    conv.zero_grad(set_to_none=True)
    b.grad = None
    out2 = conv(b, ei.cuda())
    (out2 * r.cuda()).sum().backward()
    assert torch.equal(out, out2) and torch.equal(bg, b.grad)  # no atomics anywhere: bitwise run-to-run
    for name, p in conv.named_parameters():
        assert torch.equal(p.grad, grads[name]), name


def test_global_attention_and_graph_layernorm_vs_oracle():
    """attention pooling and graph-mode LayerNorm over a collated batch (graphs of 1 .. 5000 nodes,
    2048-wide features as the reference's conditioner uses: two 1024-column SpMM calls)."""
    from causal_eeg_vae_amd.pool import GlobalAttention, GraphLayerNorm
    from oracle import ref_torch
    sizes = [1, 5000, 3, 70, 64, 65, 900]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    N, G = int(batch.shape[0]), len(sizes)
    for F in (2048, 12):
        torch.manual_seed(F)
        gate_nn = torch.nn.Linear(F, 1)
        gate64 = to64(gate_nn)
        x = torch.randn(N, F)
        r = torch.randn(G, F)
        a = x.double().requires_grad_(True)
        ref = ref_torch.ref_global_attention(a, gate64(a).view(-1), batch, G)
        (ref * r.double()).sum().backward()
        gw = gate64.weight.grad.clone()
        pool = GlobalAttention(gate_nn).cuda()
        b = x.clone().cuda().requires_grad_(True)
        out = pool(b, batch.cuda(), G)
        (out * r.cuda()).sum().backward()
        _close(out, ref)
        _close(b.grad, a.grad)
        _close(pool.gate_nn.weight.grad, gw)
        gate_nn.cpu()
        ln = GraphLayerNorm(F).cuda()
        with torch.no_grad():
            ln.weight.normal_(); ln.bias.normal_()
        a2 = x.double().requires_grad_(True)
        ref2 = ref_torch.ref_graph_layer_norm(a2, batch, G, ln.weight.detach().cpu().double(), ln.bias.detach().cpu().double())
        (ref2 * x.double()).sum().backward()
        b2 = x.clone().cuda().requires_grad_(True)
        out2 = ln(b2, batch.cuda(), G)
        (out2 * x.cuda()).sum().backward()
        _close(out2, ref2)
        _close(b2.grad, a2.grad)


@pytest.mark.parametrize("F,relu,affine,offset", [(2048, True, True, 0.0), (64, False, True, 0.0), (12, True, False, 0.0), (256, True, True, 300.0)])
def test_graph_layernorm_kernels(F, relu, affine, offset):
    """ceeg_graph_layernorm_fwd / _bwd (scripts/models.py:107-108: LayerNorm(mode='graph') + ReLU) vs the
    fp64 restatement: output, input gradient, weight / bias gradients; graphs of 1 .. 5000 nodes and an EMPTY graph in the
    middle; a batch whose mean is 300 standard deviations away from zero (fp64 statistics); bitwise reproducible."""
    from causal_eeg_vae_amd.pool import GraphLayerNorm
    from oracle import ref_torch
    sizes = [1, 5000, 3, 0, 70, 64, 65, 900]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    N, G = int(batch.shape[0]), len(sizes)
    torch.manual_seed(F)
    x = torch.randn(N, F) + offset
    r = torch.randn(N, F)
    ln = GraphLayerNorm(F, affine=affine).cuda()
    w64 = b64 = None
    if affine:
        with torch.no_grad():
            ln.weight.normal_(); ln.bias.normal_()
        w64 = ln.weight.detach().cpu().double().requires_grad_(True)
        b64 = ln.bias.detach().cpu().double().requires_grad_(True)
    a = x.double().requires_grad_(True)
    ref = ref_torch.ref_graph_layer_norm(a, batch, G, w64, b64)
    if relu:
        ref = ref.relu()
    (ref * r.double()).sum().backward()
    outs = []
    for _ in range(2):
        ln.zero_grad(set_to_none=True)
        b = x.clone().cuda().requires_grad_(True)
        out = ln(b, batch.cuda(), G, relu=relu)
        (out * r.cuda()).sum().backward()
        outs.append((out.detach(), b.grad, None if not affine else ln.weight.grad.clone(), None if not affine else ln.bias.grad.clone()))
    for u, v in zip(outs[0], outs[1]):
        assert u is None or torch.equal(u, v)
    out, dx, dw, db = outs[0]
    _close(out, ref)
    _close(dx, a.grad)
    if affine:
        _close(dw, w64.grad)
        _close(db, b64.grad)


@pytest.mark.parametrize("D", [128, 32])
def test_fused_decoder_indegree_hubs(D):
    """Destination hubs (rows of 700, 300 and 5000 entries: split at global 256-entry positions and computed piecewise)
    next to ordinary rows: loss, valid-negative count and gradient vs the oracle; both backward forms; reproducible."""
    from causal_eeg_vae_amd import EventGraph, sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    from oracle import ref_torch
    N = 6000
    rng = np.random.default_rng(D)
    base = _graph(N, 20000, 77)
    hubs = [np.stack([rng.choice(N - 1, k, replace=False) + (1 if h == 0 else 0), np.full(k, h)]) for h, k in ((0, 700), (4000, 300), (5999, 5000))]
    hubs = [hb[:, hb[0] != hb[1]] for hb in hubs]
    ei_np = np.concatenate([base] + hubs, axis=1)
    ei_np = np.unique(ei_np, axis=1)
    g = EventGraph(torch.from_numpy(ei_np).cuda(), N)
    assert g.dec.max_row > 256
    a_, b_, ainv = sigma_params(N, 2, 4)
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, sig, valid = ref_torch.sigma_negatives_np(rowptr, col, N, a_, b_)
    z0 = torch.randn(N, D) * (1.5 / D ** 0.5)
    a = z0.double().requires_grad_(True)
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    pos_ei = torch.from_numpy(np.stack([col, rows]).astype(np.int64))
    neg_ei = torch.from_numpy(np.stack([s, sig]).astype(np.int64))[:, torch.from_numpy(valid)]
    ref = ref_torch.recon_loss(a, pos_ei, neg_ei)
    ref.backward()
    outs = []
    for row_sums in (True, True, False):
        b = z0.clone().cuda().requires_grad_(True)
        rec, sums = _FusedDecoder.apply(b, g, a_, b_, ainv, row_sums)
        rec.backward()
        outs.append((rec.detach(), b.grad))
        assert int(sums[2].item()) == int(valid.sum())
        _close(rec, ref)
        _close(b.grad, a.grad)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_edge_cases_empty_and_single():
    """Degenerate inputs the reference-style callers can produce: a graph without edges, a single node, one latent row."""
    from causal_eeg_vae_amd import EventGraph, GATConv, GCNConv, VectorQuantizer, sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    empty = torch.zeros(2, 0, dtype=torch.int64, device="cuda")
    for N in (1, 5):
        g = EventGraph(empty, N)
        x = torch.randn(N, 8, device="cuda", requires_grad=True)
        conv = GCNConv(8, 4).cuda()
        out = conv(x, g)                                   # A_hat = I: out = x W^T + b
        torch.testing.assert_close(out, x @ conv.lin.weight.t() + conv.bias, rtol=1e-5, atol=1e-6)
        out.sum().backward()
        gat = GATConv(8, 4).cuda()
        o2 = gat(x, empty)                                 # only the self-loop: softmax weight 1
        torch.testing.assert_close(o2, x @ gat.lin_src.weight.t() + gat.bias, rtol=1e-5, atol=1e-6)
        z = torch.randn(N, 8, device="cuda", requires_grad=True)
        a, b, ainv = sigma_params(N, 0, 0)
        rec, sums = _FusedDecoder.apply(z, g, a, b, ainv)
        rec.backward()
        assert float(rec.detach()) == 0.0 and float(sums.abs().sum()) == 0.0 and float(z.grad.abs().sum()) == 0.0
    vq = VectorQuantizer(16, 8, 0.25).cuda()
    for B in (1, 3):
        z = torch.randn(B, 8, device="cuda", requires_grad=True)
        loss, zq, ppl, enc, idx = vq(z)
        (loss + zq.sum()).backward()
        assert tuple(idx.shape) == (B, 1) and tuple(enc.shape) == (B, 16) and torch.isfinite(loss) and torch.isfinite(z.grad).all()
        assert torch.equal(zq.detach(), vq.embedding.weight[idx[:, 0]])


def _codes_case(ei_np, N, D, n_e, sig, mod=None, seed=3):
    """Quantised latent through (a) the code-form decoder (Gram-matrix dots, LDS-resident codebook slab), (b) the table
    form of the row-gather kernels (CEEG_DEC_CODES=0) and (c) the oracle on the expanded matrix with explicit negatives."""
    import os
    from causal_eeg_vae_amd import EventGraph, _lib
    from causal_eeg_vae_amd.models import _FusedDecoder
    from oracle import ref_torch
    a_, b_, ainv = sig
    g = EventGraph(torch.from_numpy(ei_np).cuda(), N)
    torch.manual_seed(seed)
    W = (torch.randn(n_e, D) * (2.0 / D ** 0.5))
    W[:3] *= 6  # a few saturated sigmoids: the literal +EPS form must match
    idx = torch.randint(0, n_e, (N,), dtype=torch.int32)
    zq = W[idx.long()]
    assert _lib.lib().ceeg_decoder_codes_supported(n_e, D) == 1
    outs = []
    for env in ("1", "1", "0"):
        os.environ["CEEG_DEC_CODES"] = env
        try:
            x = zq.clone().cuda().requires_grad_(True)
            r, s = _FusedDecoder.apply(x, g, a_, b_, ainv, True, mod, W.cuda(), idx.cuda())
            (r * 0.7).backward()
            outs.append((r.detach(), s.clone(), x.grad.clone()))
        finally:
            os.environ.pop("CEEG_DEC_CODES", None)
    (r1, s1, g1), (r2, s2, g2), (r0, s0, g0) = outs
    assert torch.equal(r1, r2) and torch.equal(s1, s2) and torch.equal(g1, g2)     # bitwise reproducible
    assert float(s1[2]) == float(s0[2])                                            # valid-negative count is exact
    m = N if mod is None else mod
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    sig_rows = (rows - rows % m) + (a_ * (rows % m) + b_) % m
    has = set(zip(col.tolist(), rows.tolist()))
    valid = np.array([c != sr and (c, sr) not in has for c, sr in zip(col.tolist(), sig_rows.tolist())], dtype=bool)
    assert int(s1[2].item()) == int(valid.sum())
    a = zq.double().requires_grad_(True)
    pos_ei = torch.from_numpy(np.stack([col, rows]).astype(np.int64))
    neg_ei = torch.from_numpy(np.stack([col, sig_rows]).astype(np.int64))[:, torch.from_numpy(valid)]
    ref = ref_torch.recon_loss(a, pos_ei, neg_ei)
    (ref * 0.7).backward()
    for r_, g_ in ((r1, g1), (r0, g0)):   # code form and table form of the row-gather kernels, each against the exact value
        _close(r_, ref)
        _close(g_, a.grad)


@pytest.mark.parametrize("N,E,D,n_e", [(3000, 50000, 128, 512), (2000, 30000, 64, 100), (1500, 20000, 256, 512), (900, 9000, 192, 7),
                                       (64, 200, 128, 512), (65, 300, 64, 33)])
def test_decoder_code_form(N, E, D, n_e):
    from causal_eeg_vae_amd import sigma_params
    _codes_case(_graph(N, E, N + D), N, D, n_e, sigma_params(N, 1, 9))


def test_decoder_code_form_per_graph_blocks_and_hubs():
    """sigma permuting inside equal-sized graphs of a collated batch; in-degree hubs (rows of 700 / 300 / 5000 entries) and
    the out-degree hubs of the generator (transposed rows > 256 entries): every long list goes through the piece kernels."""
    from causal_eeg_vae_amd import EventGraph, sigma_params
    n, G = 500, 6
    eis = [_graph(n, 6000, 100 + g) + g * n for g in range(G)]
    _codes_case(np.concatenate(eis, axis=1), n * G, 128, 512, sigma_params(n, 4, 2), mod=n)
    N = 6000
    rng = np.random.default_rng(5)
    base = _graph(N, 20000, 77)
    hubs = [np.stack([rng.choice(N - 1, k, replace=False) + (1 if h == 0 else 0), np.full(k, h)]) for h, k in ((0, 700), (4000, 300), (5999, 5000))]
    hubs = [hb[:, hb[0] != hb[1]] for hb in hubs]
    ei_np = np.unique(np.concatenate([base] + hubs, axis=1), axis=1)
    g = EventGraph(torch.from_numpy(ei_np).cuda(), N)
    assert g.dec.max_row > 256 and g.dec.max_trow > 256
    assert g.dec.long_items(False) is not None and g.dec.long_items(True) is not None
    _codes_case(ei_np, N, 128, 512, sigma_params(N, 2, 4))


def test_decoder_code_form_degenerate():
    from causal_eeg_vae_amd import EventGraph, sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    empty = torch.zeros(2, 0, dtype=torch.int64, device="cuda")
    for N in (1, 5):
        g = EventGraph(empty, N)
        W = torch.randn(16, 64, device="cuda")
        idx = torch.zeros(N, dtype=torch.int32, device="cuda")
        z = W[idx.long()].clone().requires_grad_(True)
        a, b, ainv = sigma_params(N, 0, 0)
        rec, sums = _FusedDecoder.apply(z, g, a, b, ainv, True, None, W, idx)
        rec.backward()
        assert float(rec.detach()) == 0.0 and float(sums.abs().sum()) == 0.0 and float(z.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("N,D,n_e,with_eps", [(5000, 128, 512, False), (777, 32, 40, True), (300, 256, 512, False), (1, 8, 4, True)])
def test_fused_reparam_vq_backward_equals_the_two_steps(N, D, n_e, with_eps, monkeypatch):
    """ceeg_reparam_vq_bwd (dz of the quantiser never written, z recomputed, bias column sums in the same pass) gives the
    gradient of ceeg_vq_backward followed by ceeg_reparam_kl_bwd bit for bit; its column sums equal the gradient's."""
    from causal_eeg_vae_amd import VectorQuantizer
    from causal_eeg_vae_amd.models import _reparam_vq
    torch.manual_seed(N + D)
    ml0 = torch.randn(N, 2 * D, device="cuda")
    ml0[::5, D:] = 10.7   # clamped rows: zero logstd gradient
    eps = torch.randn(N, D, device="cuda") if with_eps else None
    vq = VectorQuantizer(n_e, D, 0.25, return_one_hot=False).cuda()
    with torch.no_grad():
        vq.embedding.weight.normal_()
    r = torch.randn(N, D, device="cuda")
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("CEEG_FUSED_RVQ", fused)
        vq.embedding.weight.grad = None
        ml = ml0.clone().requires_grad_(True)
        z, kl, vq_loss, z_q, ppl, idx = _reparam_vq(ml, eps, 1234, True, vq)
        loss = 1.3 * vq_loss + (z_q * r).sum() + 0.7 * kl
        loss.backward()
        outs.append((z.detach(), kl.detach(), vq_loss.detach(), z_q.detach(), idx, ml.grad.clone(), vq.embedding.weight.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # the head function: its bias gradient (the fused backward's column sums, a real output) equals the column sums of the
    # gradient that reaches mu | logstd, and its other gradients those of the dense product in front of _reparam_vq
    from causal_eeg_vae_amd.models import _HeadReparamVQ
    ah = ml0.clone()                                            # ml = ah (I + small)^T + small: the clamped rows stay clamped
    wcat = (torch.eye(2 * D, device="cuda") + 0.01 * torch.randn(2 * D, 2 * D, device="cuda")).requires_grad_(True)
    bcat = (0.05 * torch.randn(2 * D, device="cuda")).requires_grad_(True)
    ah_a = ah.clone().requires_grad_(True)
    vq.embedding.weight.grad = None
    ml, z, kl, vq_loss, z_q, ppl, idx = _HeadReparamVQ.apply(ah_a, wcat, bcat, eps, 1234, True, False, vq.embedding.weight, vq.beta)
    (1.3 * vq_loss + (z_q * r).sum() + 0.7 * kl).backward()
    got = (ah_a.grad.clone(), wcat.grad.clone(), bcat.grad.clone(), vq.embedding.weight.grad.clone())
    ah_b, w_b, b_b = ah.clone().requires_grad_(True), wcat.detach().clone().requires_grad_(True), bcat.detach().clone().requires_grad_(True)
    vq.embedding.weight.grad = None
    ml_b = torch.addmm(b_b, ah_b, w_b.t().contiguous())
    ml_b.retain_grad()
    z2, kl2, vq_loss2, z_q2, _, idx2 = _reparam_vq(ml_b, eps, 1234, True, vq)
    (1.3 * vq_loss2 + (z_q2 * r).sum() + 0.7 * kl2).backward()
    assert torch.equal(ml.detach(), ml_b.detach()) and torch.equal(idx, idx2)
    ref = ml_b.grad.double().sum(0)
    np.testing.assert_allclose(got[2].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(ref.abs().max()) + 1e-30)
    assert torch.equal(got[3], vq.embedding.weight.grad)
    np.testing.assert_allclose(got[0].cpu().numpy(), ah_b.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(ah_b.grad.abs().max()))
    np.testing.assert_allclose(got[1].cpu().numpy(), w_b.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(w_b.grad.abs().max()))


def test_head_bias_gradient_when_the_encoder_output_feeds_a_second_loss():
    """aux['ml'] (the fused mu | logstd encoder output) also feeds the loss: a second gradient then arrives for that output of
    the head function, is added to the fused reparam/VQ backward's result, and the column sums that kernel left no longer
    describe the total - they are recomputed.  The bias gradients must still be exact."""
    from causal_eeg_vae_amd import VGAEVQ, EventGraph, sigma_params
    from oracle import ref_torch
    N, E, Fi, D, n_e = 1500, 20000, 16, 16, 32
    ei_np = _graph(N, E, 4)
    torch.manual_seed(2)
    ref = ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = VGAEVQ(Fi, D, n_e, 0.25).cuda()
    model.load_state_dict(ref.state_dict())
    x, eps, r = torch.randn(N, Fi), torch.randn(N, D), torch.randn(N, 2 * D)
    sig = sigma_params(N, 0, 1)
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s, sg, valid = ref_torch.sigma_negatives_np(rowptr, col, N, sig[0], sig[1])
    neg_ei = torch.from_numpy(np.stack([s, sg]).astype(np.int64))
    ei = torch.from_numpy(ei_np)
    loss, aux = model(x.cuda(), EventGraph(ei.cuda(), N), step=1, eps=eps.cuda(), sigma=sig)
    (loss + (aux["ml"] * r.cuda()).sum()).backward()
    ref = to64(ref)
    loss_ref, aux_ref = ref(x.double(), ei, eps.double(), neg_ei, torch.from_numpy(valid), force_idx=aux["idx"].cpu())
    (loss_ref + (torch.cat((aux_ref["mu"], aux_ref["logstd"]), dim=1) * r.double()).sum()).backward()
    grads_close(model, ref)


@pytest.mark.parametrize("flip", [False, True])
def test_decoder_code_form_with_condition_columns(flip):
    """cat((z_q, c[batch])) decoded (a) in code form with |c_graph|^2 as the row addend and coef_sum * c as the condition's
    gradient, (b) as the concatenated matrix through the row-gather kernels: same loss, count, gradients.  Node 0 of every
    graph is a hub with more than 256 out-edges (flip: in-edges), so the coefficient sums take their long-list phase for the
    transposed list (flip: for the row and sigma-row lists)."""
    from causal_eeg_vae_amd import EventGraph, sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    import os
    n, G, D, C, n_e = 800, 4, 128, 8, 512
    eis = [_graph(n, 10000, 200 + g)[::-1 if flip else 1] + g * n for g in range(G)]
    ei_np = np.ascontiguousarray(np.concatenate(eis, axis=1))
    N = n * G
    g = EventGraph(torch.from_numpy(ei_np).cuda(), N)
    assert (g.dec.max_row if flip else g.dec.max_trow) > 256
    a_, b_, ainv = sigma_params(n, 7, 1)
    torch.manual_seed(0)
    W = (torch.randn(n_e, D) * (1.5 / D ** 0.5)).cuda()
    idx = torch.randint(0, n_e, (N,), dtype=torch.int32).cuda()
    cond = (torch.randn(G, C) * 0.5).cuda()
    batch = torch.arange(N, device="cuda") // n
    outs = []
    for env in ("1", "0"):
        os.environ["CEEG_DEC_CODES"] = env
        try:
            zq = W[idx.long()].clone().requires_grad_(True)
            c = cond.clone().requires_grad_(True)
            cn = c.index_select(0, batch)
            rec, sums = _FusedDecoder.apply(zq, g, a_, b_, ainv, True, n, W, idx, cn)
            (rec * 1.3).backward()
            outs.append((rec.detach(), sums.clone(), zq.grad.clone(), c.grad.clone()))
        finally:
            os.environ.pop("CEEG_DEC_CODES", None)
    (r1, s1, gz1, gc1), (r0, s0, gz0, gc0) = outs
    assert float(s1[2]) == float(s0[2])
    # the plain concatenation (no table at all)
    zc = torch.cat((W[idx.long()], cond.index_select(0, batch)), dim=1).clone().requires_grad_(True)
    r2, s2 = _FusedDecoder.apply(zc, g, a_, b_, ainv, True, n)
    (r2 * 1.3).backward()
    # all three against the fp64 oracle on the concatenated matrix with the same explicit negatives
    from oracle import ref_torch
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    s_, sg_, valid = ref_torch.sigma_negatives_np(rowptr, col, N, a_, b_, n)
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    z64 = W[idx.long()].cpu().double().requires_grad_(True)
    c64 = cond.cpu().double().requires_grad_(True)
    ref = ref_torch.recon_loss(torch.cat((z64, c64[batch.cpu()]), dim=1), torch.from_numpy(np.stack([col, rows]).astype(np.int64)),
                               torch.from_numpy(np.stack([s_, sg_]).astype(np.int64))[:, torch.from_numpy(valid)])
    (ref * 1.3).backward()
    assert int(s1[2]) == int(valid.sum())
    for r_, gz_, gc_ in ((r1, gz1, gc1), (r0, gz0, gc0)):
        _close(r_, ref)
        _close(gz_, z64.grad)
        _close(gc_, c64.grad)
    _close(r2, ref)
    _close(zc.grad[:, :D], z64.grad)


@pytest.mark.parametrize("N,T,H", [(37, 60, 768), (5, 1, 4), (130, 7, 48), (3, 300, 1100)])
def test_token_attention_pool_kernels(N, T, H):
    """ceeg_token_pool_fwd / _bwd vs the reference's two lines (scripts/models.py:118-119) in fp64: output and the gradients of
    hidden, attn.weight, attn.bias; the reference's own shape (60 tokens x 768) included; bitwise reproducible."""
    from causal_eeg_vae_amd.pool import token_attention_pool
    torch.manual_seed(N + T)
    attn = torch.nn.Linear(H, 1)
    hidden, r = torch.randn(N, T, H), torch.randn(N, H)
    a64 = to64(attn)
    h64 = hidden.double().requires_grad_(True)
    ref = torch.bmm(h64.permute(0, 2, 1), a64(h64)).squeeze(2)
    (ref * r.double()).sum().backward()
    attn = attn.cuda()
    outs = []
    for _ in range(2):
        attn.zero_grad(set_to_none=True)
        hc = hidden.clone().cuda().requires_grad_(True)
        out = token_attention_pool(hc, attn)
        (out * r.cuda()).sum().backward()
        outs.append((out.detach(), hc.grad, attn.weight.grad.clone(), attn.bias.grad.clone()))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    _close(outs[0][0], ref)
    _close(outs[0][1], h64.grad)
    _close(outs[0][2], a64.weight.grad)
    _close(outs[0][3], a64.bias.grad)


def test_graph_conditioner_vs_oracle():
    """The reference's GraphConditioner (scripts/models.py:95-122; SURVEY.md section 8(f) N1 + N4) end to end on a collated
    batch: token attention pooling -> 3 x [GATConv -> graph LayerNorm -> ReLU] -> GlobalAttention, with a stand-in token
    encoder (fixed hidden states).  The reference's state-dict layout loads; output and every gradient against the
    restatement."""
    from types import SimpleNamespace
    from causal_eeg_vae_amd.conditioner import GraphConditioner
    from causal_eeg_vae_amd.synth import collate_graphs
    from oracle import ref_torch
    torch.manual_seed(5)
    sizes = [40, 1, 130, 77]
    eis = [torch.from_numpy(_graph(n, max(1, 6 * n), 900 + n)) if n > 1 else torch.zeros(2, 0, dtype=torch.int64) for n in sizes]
    ei, batch, N = collate_graphs(eis, sizes)
    T, H = 7, 48
    hidden = torch.randn(N, T, H)

    class Enc(torch.nn.Module):  # stand-in for the token encoder: returns the hidden states it was given
        def __init__(self, h):
            super().__init__()
            self.h = torch.nn.Parameter(h.clone())

        def forward(self, input_ids=None, attention_mask=None):
            return SimpleNamespace(last_hidden_state=self.h)

    ref = ref_torch.RefGraphConditioner(Enc(hidden), H, [64, 320, 32])
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1 and p is not ref.embedding.h:
                p.normal_(0.5, 0.3)
    mine = GraphConditioner(Enc(hidden), H, [64, 320, 32])
    missing = mine.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    mine = mine.cuda()
    ref = to64(ref)
    inp = SimpleNamespace(input_ids=None, attention_mask=None, edge_index=ei, input_ids_batch=batch, num_graphs=len(sizes))
    inp_gpu = SimpleNamespace(input_ids=None, attention_mask=None, edge_index=ei.cuda(), input_ids_batch=batch.cuda(), num_graphs=len(sizes))
    r = torch.randn(len(sizes), 32)
    out_ref = ref(inp)
    (out_ref * r.double()).sum().backward()
    out = mine(inp_gpu)
    (out * r.cuda()).sum().backward()
    assert out.shape == (len(sizes), 32)
    _close(out, out_ref)
    want = dict(ref.named_parameters())
    scale = max(float(w.grad.abs().max()) for w in want.values())
    for name, p in mine.named_parameters():
        # a gradient that is analytically ~0 (the attention vectors in front of a graph LayerNorm, which removes their
        # common shift: 1e-9 of the model's largest gradient) is a difference of O(1) terms: its absolute floor is tied to
        # the largest gradient of the model, the relative part stays the north-star 1e-5
        g, w = p.grad.cpu().double(), want[name].grad.double()
        tol = 1e-5 * w.abs() + 1e-6 * float(w.abs().max()) + 1e-6 * scale
        assert bool(((g - w).abs() <= tol).all()), (name, float((g - w).abs().max()), float(w.abs().max()), scale)


def test_pool_kernels_reduced_precision_and_odd_widths():
    """The fp32-only kernels behind token_attention_pool / GraphLayerNorm never reinterpret other dtypes: a bf16 hidden state
    (an autocast token encoder) or a bf16 / half activation is widened, the result comes back in the input's dtype and equals
    the fp32 path on the same (rounded) values; integer inputs raise; widths the fused LayerNorm does not take (F % 4 != 0)
    run the torch-op form on the device and match the fp64 restatement."""
    from causal_eeg_vae_amd.pool import GraphLayerNorm, token_attention_pool
    from oracle import ref_torch
    torch.manual_seed(3)
    attn = torch.nn.Linear(48, 1).cuda()
    hidden = torch.randn(9, 7, 48, device="cuda")
    for dt in (torch.bfloat16, torch.float16):
        hb = hidden.to(dt)
        out = token_attention_pool(hb, attn)
        assert out.dtype == dt
        want = token_attention_pool(hb.float(), attn)
        assert torch.equal(out, want.to(dt))
    with pytest.raises(TypeError):
        token_attention_pool(torch.zeros(2, 3, 8, dtype=torch.int32, device="cuda"), attn)
    hb = hidden.to(torch.bfloat16).requires_grad_(True)
    token_attention_pool(hb, attn).float().sum().backward()
    assert hb.grad is not None and hb.grad.dtype == torch.bfloat16

    sizes = [5, 1, 40]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ln = GraphLayerNorm(64).cuda()
    x = torch.randn(sum(sizes), 64, device="cuda")
    xb = x.to(torch.bfloat16)
    out = ln(xb, batch.cuda(), len(sizes), relu=True)
    assert out.dtype == torch.bfloat16 and torch.equal(out, ln(xb.float(), batch.cuda(), len(sizes), relu=True).to(torch.bfloat16))
    for F in (10, 7):
        ln = GraphLayerNorm(F).cuda()
        with torch.no_grad():
            ln.weight.normal_(); ln.bias.normal_()
        x = torch.randn(sum(sizes), F)
        a = x.double().requires_grad_(True)
        ref = ref_torch.ref_graph_layer_norm(a, batch, len(sizes), ln.weight.detach().cpu().double(), ln.bias.detach().cpu().double())
        (ref * x.double()).sum().backward()
        b = x.clone().cuda().requires_grad_(True)
        out = ln(b, batch.cuda(), len(sizes))
        (out * x.cuda()).sum().backward()
        _close(out, ref)
        _close(b.grad, a.grad)


def test_minibatch_stream_construction_equals_static_graph():
    """CollatedBatch.event_graph() - collation, both CSR pairs, piece lists and the "no edge leaves its graph" check in ONE host
    read (what bench.py --fresh-batches times per step) - feeds the conditional step exactly like the separately built
    EventGraph: loss and every gradient bit for bit; a batch with an edge crossing graphs is flagged on the device and takes the
    concatenating decoder (same result as the static path on that batch)."""
    from causal_eeg_vae_amd import ConditionalVGAEVQ, EventGraph
    from causal_eeg_vae_amd.collate import CollatedBatch
    G, n, e, Fi, C, D, n_e = 6, 400, 5000, 28, 4, 64, 64
    eis = [torch.from_numpy(_graph(n, e, 40 + g)).cuda() for g in range(G)]
    torch.manual_seed(2)
    model = ConditionalVGAEVQ(Fi, C, D, n_e, 0.25).cuda()
    with torch.no_grad():
        model.vector_quantization.embedding.weight.normal_(0, 0.3)
    x, cond = torch.randn(G * n, Fi, device="cuda"), torch.randn(G, C, device="cuda")

    def run(graph, batch):
        model.zero_grad(set_to_none=True)
        loss, aux = model(x, graph, batch, cond, step=1, nodes_per_graph=n)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in model.parameters()], aux

    cb = CollatedBatch(eis, [n] * G)
    g_stream = cb.event_graph()
    assert g_stream.checks == {("intra", n): True} and g_stream._dec is not None
    assert ConditionalVGAEVQ._intra_graph(g_stream, cb.batch, n) is True
    l1, g1, _ = run(g_stream, cb.batch)
    l2, g2, _ = run(EventGraph(cb.edge_index, cb.num_nodes), cb.batch)
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    # one edge of graph 0 re-pointed into graph 1 (ids are local: n + 3 lies outside graph 0's block)
    bad = [e_.clone() for e_ in eis]
    bad[0][1, 0] = n + 3
    cb2 = CollatedBatch(bad, [n] * G)
    g_bad = cb2.event_graph()
    assert g_bad.checks == {("intra", n): False}
    l3, g3, _ = run(g_bad, cb2.batch)
    l4, g4, _ = run(EventGraph(cb2.edge_index, cb2.num_nodes), cb2.batch)
    assert torch.equal(l3, l4) and all(torch.equal(a, b) for a, b in zip(g3, g4))


def test_minibatch_stream_prefetch_equals_serial_loop():
    """collate.MinibatchStream: minibatches collated and built ahead on a side stream while the steps train on the current one.  Seven
    optimizer steps over minibatches of different sizes (so that the allocator hands the structures of minibatch k to the build
    of minibatch k+2) give the same graphs, losses and final parameters, bit for bit, as the plain loop on one stream."""
    from causal_eeg_vae_amd import ConditionalVGAEVQ, MinibatchStream
    from causal_eeg_vae_amd.trainer import Trainer
    n, Fi, C, D, n_e = 300, 28, 4, 64, 64
    sizes = [(5, 3000), (6, 2500), (4, 4100), (6, 2500), (5, 3000), (3, 5200), (6, 1800)]
    mbs = [([torch.from_numpy(_graph(n, e, 100 * k + g)).cuda() for g in range(G)], [n] * G) for k, (G, e) in enumerate(sizes)]
    gen = torch.Generator(device="cuda").manual_seed(5)
    feats = [(torch.randn(G * n, Fi, device="cuda", generator=gen), torch.randn(G, C, device="cuda", generator=gen)) for G, _ in sizes]

    def train(prefetch):
        torch.manual_seed(2)
        model = ConditionalVGAEVQ(Fi, C, D, n_e, 0.25).cuda()
        with torch.no_grad():
            model.vector_quantization.embedding.weight.normal_(0, 0.3)
        trainer = Trainer(model, lr=1e-2)
        losses, structs = [], []
        ms = MinibatchStream(mbs, prefetch=prefetch, timing=True)
        for (cb, graph), (x, cond) in zip(ms, feats):
            assert graph.checks == {("intra", n): True}
            structs.append([t.clone() for t in (graph.gcn.rowptr, graph.gcn.col, graph.gcn.t_perm, graph.dec.t_col, graph.gcn_val, cb.batch)])
            loss, _ = trainer.step(x, graph, cb.batch, cond, nodes_per_graph=n)
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        assert len(ms.build_events) == 2 * len(mbs) and all(a.elapsed_time(b) > 0 for a, b in ms.build_events)   # begin + finish
        return losses, structs, [p.detach().clone() for p in model.parameters()]

    l_a, s_a, p_a = train(False)
    l_b, s_b, p_b = train(True)
    assert all(torch.equal(a, b) for a, b in zip(l_a, l_b))
    assert all(torch.equal(a, b) for sa, sb in zip(s_a, s_b) for a, b in zip(sa, sb))
    assert all(torch.equal(a, b) for a, b in zip(p_a, p_b))
    assert list(MinibatchStream([], prefetch=True)) == []


@pytest.mark.parametrize("prefetch,depth", [(False, 1), (True, 1), (True, 3)])
def test_minibatch_stream_follow_keys_and_edge_weights(prefetch, depth):
    """MinibatchStream items with followed keys and collated edge weights give what CollatedBatch(...).event_graph(edge_weight)
    gives directly (any window depth, also deeper than the sequence is long)."""
    from causal_eeg_vae_amd import CollatedBatch, MinibatchStream
    n = 120
    items, direct = [], []
    for k, (G, e) in enumerate([(3, 900), (2, 1500), (4, 700), (1, 50)]):
        eis = [torch.from_numpy(_graph(n, e, 300 + 10 * k + g)).cuda() for g in range(G)]
        follow = {"tok": [torch.full((g + 2, 3), float(10 * k + g), device="cuda") for g in range(G)]}
        w = torch.rand(G * e, device="cuda") + 0.5
        items.append((eis, [n] * G, follow, w))
        cb = CollatedBatch(eis, [n] * G, follow)
        direct.append((cb, cb.event_graph(edge_weight=w)))
    if not prefetch:   # an unfinished graph says so instead of failing somewhere inside a step
        from causal_eeg_vae_amd import _lib
        unfinished = direct[0][0].event_graph(defer=True)
        with pytest.raises(_lib.CeegError, match="finish"):
            unfinished.gcn_val
        assert CollatedBatch.finish_graph(unfinished).gcn_val.shape == direct[0][1].gcn_val.shape
    got = list(MinibatchStream(items, prefetch=prefetch, depth=depth))
    torch.cuda.synchronize()
    assert len(got) == len(direct)
    for (cb, g), (cb0, g0) in zip(got, direct):
        assert torch.equal(cb.edge_index, cb0.edge_index) and torch.equal(cb.batch, cb0.batch)
        assert torch.equal(cb.tok, cb0.tok) and torch.equal(cb.tok_batch, cb0.tok_batch)
        assert g.checks == g0.checks and g.reuse is False
        for a, b in ((g.gcn.rowptr, g0.gcn.rowptr), (g.gcn.col, g0.gcn.col), (g.gcn.t_perm, g0.gcn.t_perm), (g.gcn_val, g0.gcn_val),
                     (g.gcn_tval, g0.gcn_tval), (g.dinv, g0.dinv), (g.dec.col, g0.dec.col)):
            assert torch.equal(a, b)


@pytest.mark.parametrize("negatives", ["sigma", "uniform"])
def test_bf16_step_vs_oracle_at_bf16_tolerance(negatives):
    """The opt-in reduced-precision step (VGAEVQ(compute_dtype=torch.bfloat16): bf16 storage of the encoder's activations, bf16
    MFMA products, fp32 sums, fp32 quantiser / decoder) against the fp64 oracle on the same fp32 inputs, at the north star's
    bf16 tolerance: rtol 1e-2 with the absolute floor 1e-2 max|ref| for gradients (a gradient entry is a long sum of rounded
    terms) - loss terms, latent, every parameter gradient.  The bf16 noise on z (~0.4 % per rounding) moves rows across
    near-by code boundaries: every flipped row's exact distance to the GPU's code must be within 3 % of the row's scale of the
    minimum, and losses / gradients are then compared with the GPU's codes forced.  Two runs are bitwise identical."""
    from causal_eeg_vae_amd import VGAEVQ, EventGraph, sigma_params
    from oracle import c_oracle, ref_torch
    N, E, Fi, D, n_e = 4000, 64000, 16, 16, 64
    ei_np = _graph(N, E, 2)
    torch.manual_seed(0)
    ref = ref_torch.RefVGAEVQ(Fi, D, n_e, 0.25)
    with torch.no_grad():
        ref.vector_quantization.embedding.weight.normal_(0, 0.3)
    model = VGAEVQ(Fi, D, n_e, 0.25, seed=3, negatives=negatives, compute_dtype=torch.bfloat16).cuda()
    model.load_state_dict(ref.state_dict())
    x, eps = torch.randn(N, Fi), torch.randn(N, D)
    ei = torch.from_numpy(ei_np)
    g = EventGraph(ei.cuda(), N)
    rowptr, col, _ = ref_torch.csr_build_np(ei_np[0], ei_np[1], N, False)
    if negatives == "sigma":
        sig = sigma_params(N, 0, 5)
        s, sg, valid = ref_torch.sigma_negatives_np(rowptr, col, N, sig[0], sig[1])
        neg_ei, valid_t, kw = torch.from_numpy(np.stack([s, sg]).astype(np.int64)), torch.from_numpy(valid), dict(sigma=sig)
    else:
        s, d = c_oracle.neg_sample_uniform(rowptr, col, N, E, model.negative_seed(5))
        neg_ei, valid_t, kw = torch.from_numpy(np.stack([s, d]).astype(np.int64)), None, {}
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        loss, aux = model(x.cuda().to(torch.bfloat16), g, step=5, eps=eps.cuda(), **kw)
        loss.backward()
        runs.append((loss.detach().clone(), aux["idx"].clone(), [p.grad.clone() for p in model.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][2], runs[1][2]))
    ref64 = to64(ref)
    with torch.no_grad():
        _, aux0 = ref64(x.double(), ei, eps.double(), neg_ei, valid_t)
    _close(aux["z"], aux0["z"], rtol=1e-2, atol_rel=1e-2)
    n_flip = flipped_rows_are_near_ties(aux["idx"], ref64.vector_quantization, bound_rel=3e-2)
    assert n_flip < N // 4
    loss_ref, aux_ref = ref64(x.double(), ei, eps.double(), neg_ei, valid_t, force_idx=aux["idx"].cpu())
    loss_ref.backward()
    for k in ("kl", "rec", "vq"):
        _close(aux[k], aux_ref[k], rtol=1e-2, atol_rel=1e-2, what=k)
    _close(loss, loss_ref, rtol=1e-2, atol_rel=1e-2)
    grads_close(model, ref64, rtol=1e-2, atol_rel=1e-2)
