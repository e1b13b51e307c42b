"""BASELINE.json's full sizes through size-independent properties (the oracle cannot run these in seconds):
config 2 (1 M nodes / 16 M edges): CSR order and checksums, SpMM adjointness <A x, y> = <x, A^T y>, linearity,
bitwise reproducibility, fused decoder determinism + gradient consistency with the four-pass form, the sampled
negatives and their pair-list loss / gradient against float64;
config 3 (4 Mi x 512 x 256): z_q rows are codebook rows, histogram sums to B, idempotence (fp32), reproducibility."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def config2():
    from causal_eeg_vae_amd import EventGraph
    from causal_eeg_vae_amd.synth import synth_event_graph
    N, E = 1_000_000, 16_000_000
    ei = synth_event_graph(N, E, seed=3, device="cuda")
    g = EventGraph(ei, N)
    g.dec
    return N, E, ei, g


def test_config2_csr_sorted_and_checksums(config2):
    N, E, ei, g = config2
    for pair, loops in ((g.gcn, N), (g.dec, 0)):
        assert pair.nnz == E + loops
        rp = pair.rowptr.long()
        assert int(rp[0]) == 0 and int(rp[-1]) == pair.nnz and bool((rp[1:] >= rp[:-1]).all())
        rows = torch.repeat_interleave(torch.arange(N, device="cuda"), rp[1:] - rp[:-1])
        key = rows * N + pair.col.long()
        assert bool((key[1:] > key[:-1]).all())            # strictly ascending (dst, src): sorted, no duplicates
        # checksum of checksums: the multiset of (src, dst) pairs is the input's (+ one loop per node)
        want = (ei[1] * N + ei[0]).sum() + (loops and (torch.arange(N, device="cuda") * (N + 1)).sum())
        assert int(key.sum()) == int(want)
        # transpose holds the same entries grouped by source, and t_perm points back at them
        trp = pair.t_rowptr.long()
        trow = torch.repeat_interleave(torch.arange(N, device="cuda"), trp[1:] - trp[:-1])
        assert torch.equal(rows[pair.t_perm.long()], pair.t_col.long()) and torch.equal(pair.col.long()[pair.t_perm.long()], trow)


@pytest.mark.parametrize("F", [128, 256])
def test_config2_spmm_adjoint_linear_reproducible(config2, F):
    from causal_eeg_vae_amd.models import _spmm
    N, E, ei, g = config2
    gen = torch.Generator(device="cuda").manual_seed(F)
    x = torch.randn(N, F, device="cuda", generator=gen)
    y = torch.randn(N, F, device="cuda", generator=gen)
    A = lambda v: _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, v, N, max_row=g.gcn.max_row)
    AT = lambda v: _spmm(g.gcn.t_rowptr, g.gcn.t_col, g.gcn_tval, None, v, N, max_row=g.gcn.max_trow)
    ax, aty = A(x), AT(y)
    assert torch.equal(ax, A(x)) and torch.equal(aty, AT(y))
    lhs, rhs = (ax.double() * y.double()).sum(), (x.double() * aty.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * float((ax.double().abs() * y.double().abs()).sum())
    # the permuted-value form of the transposed pass gives the same numbers as the pre-permuted stream
    assert torch.equal(aty, _spmm(g.gcn.t_rowptr, g.gcn.t_col, g.gcn_val, g.gcn.t_perm, y, N))
    # linearity, both sides fp32: each within rounding of the exact sum, so they differ by at most a few ulp of sum |terms|
    mag = A((2 * x + y).abs()) + 1e-30
    assert float(((A(2 * x + y) - (2 * ax + A(y))).abs() / mag).max()) < 1e-5
    # exactness on a sample of rows (hub rows of the transpose included) against an fp64 sum
    rp, trp = g.gcn.rowptr.long(), g.gcn.t_rowptr.long()
    pick = torch.cat([torch.arange(0, N, 997, device="cuda"), torch.topk(trp[1:] - trp[:-1], 8).indices])
    for out, ptr, col, val, src in ((ax, rp.cpu(), g.gcn.col, g.gcn_val, x), (aty, trp.cpu(), g.gcn.t_col, g.gcn_tval, y)):
        for r in pick.tolist():
            lo, hi = int(ptr[r]), int(ptr[r + 1])
            ref = (val[lo:hi].double()[:, None] * src[col[lo:hi].long()].double()).sum(0)
            tol = 1e-5 * ref.abs() + 1e-6 * float(ref.abs().max())
            assert bool(((out[r].double() - ref).abs() <= tol).all()), r
    # rows of A_hat sum to <= 1-ish positive numbers: A 1 = D^-1/2 (A+I) D^-1/2 1 is finite and positive
    ones = A(torch.ones(N, F, device="cuda"))
    assert bool(torch.isfinite(ones).all()) and float(ones.min()) > 0


def _decoder_fp64_on_gpu(zq, pair, N, a, b):
    """recon_loss (literal +EPS form) and its gradient for the CSR-ordered positives (col[p], r) and the permutation-
    structured negatives (col[p], sigma(r)), in float64 with torch on the GPU as the checker (chunks of 2 M entries;
    index_add_ in fp64: its atomics' order moves the 16th digit, not the 6th)."""
    EPS = 1e-15
    rp = pair.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(N, device="cuda"), rp[1:] - rp[:-1])
    cols = pair.col.long()
    sig = (a * rows + b) % N
    key = rows * N + cols                                   # ascending: the CSR is sorted by (dst, src)
    nkey = sig * N + cols
    pos = torch.searchsorted(key, nkey).clamp(max=key.numel() - 1)
    valid = (key[pos] != nkey) & (cols != sig)
    z = zq.double()
    grad = torch.zeros_like(z)
    n_valid = int(valid.sum())
    E = cols.numel()
    loss = torch.zeros((), dtype=torch.float64, device="cuda")
    for lo in range(0, E, 2_000_000):
        sl = slice(lo, lo + 2_000_000)
        zi = z[cols[sl]]
        for other, neg, w in ((rows[sl], False, None), (sig[sl], True, valid[sl])):
            zj = z[other]
            p = torch.sigmoid((zi * zj).sum(1))
            if neg:
                term, c = -torch.log(1 - p + EPS), p * (1 - p) / (1 - p + EPS) / n_valid
                term, c = term * w, c * w
                loss += term.sum() / n_valid
            else:
                term, c = -torch.log(p + EPS), -p * (1 - p) / (p + EPS) / E
                loss += term.sum() / E
            grad.index_add_(0, cols[sl], c[:, None] * zj)
            grad.index_add_(0, other, c[:, None] * zi)
    return loss, grad, n_valid


def test_config2_fused_decoder(config2):
    from causal_eeg_vae_amd import sigma_params
    from causal_eeg_vae_amd.models import _FusedDecoder
    from _parity import close
    N, E, ei, g = config2
    gen = torch.Generator(device="cuda").manual_seed(1)
    W = torch.randn(512, 128, device="cuda", generator=gen) * 0.2
    idx = torch.randint(0, 512, (N,), device="cuda", generator=gen, dtype=torch.int32)
    zq = W[idx.long()]
    a, b, ainv = sigma_params(N, 0, 11)
    outs = []
    for kw in (dict(row_sums=True, table=True), dict(row_sums=True, table=True), dict(row_sums=True, table=False), dict(row_sums=False, table=False)):
        x = zq.clone().requires_grad_(True)
        args = (W, idx) if kw["table"] else (None, None)
        rec, sums = _FusedDecoder.apply(x, g, a, b, ainv, kw["row_sums"], None, *args)
        rec.backward()
        outs.append((rec.detach(), sums.detach(), x.grad))
    o = outs[1]                                             # rerun of the code form (Gram dots, LDS codebook slab): bitwise
    assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
    assert torch.equal(outs[3][0], outs[2][0])              # four-pass backward: same forward kernel, same loss
    # every form (code form; expanded-matrix row-gather kernels with the fused and with the four-pass backward) against the
    # exact value at full size: north-star tolerance, valid-negative count exact
    loss64, grad64, n_valid = _decoder_fp64_on_gpu(zq, g.dec, N, a, b)
    for rec, sums, grad in (outs[0], outs[2], outs[3]):
        assert int(sums[2]) == n_valid
        close(rec, loss64)
        close(grad, grad64)
    assert 0.99 * E < n_valid <= E                          # few permuted pairs collide with edges on a sparse graph
    assert float(outs[0][0]) > 0


def test_config2_sampled_negatives(config2, monkeypatch):
    """The literal sampler path at full size: E pairs from ceeg_neg_sample_uniform_filtered - identical with and without the
    bitmap prefilter, none of them an edge or a self pair, (nearly) no holes, reproducible - and the pair-list code form
    (ceeg_pair_bce_codes_*: incidence records, one radix sort, LDS codebook slab) against float64 torch on the same pairs:
    count exact, loss and gradient at the north-star tolerance, two runs bitwise equal."""
    from causal_eeg_vae_amd.models import _PairBCECodes, negative_sampling_raw
    from _parity import close
    N, E, ei, g = config2
    neg = negative_sampling_raw(g, N, E, 1234)
    monkeypatch.setenv("CEEG_NEG_FILTER", "0")
    assert torch.equal(negative_sampling_raw(g, N, E, 1234), neg)
    monkeypatch.delenv("CEEG_NEG_FILTER")
    assert torch.equal(negative_sampling_raw(g, N, E, 1234), neg)
    ok = neg[0] >= 0
    assert int(ok.sum()) >= E - 16 and bool((neg[1][ok] >= 0).all()) and bool((neg[0][~ok] == -1).all())
    s, d = neg[0][ok].long(), neg[1][ok].long()
    assert bool((s != d).all()) and int(s.max()) < N and int(d.max()) < N
    rp = g.dec.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(N, device="cuda"), rp[1:] - rp[:-1])
    key = rows * N + g.dec.col.long()                       # ascending (dst, src)
    nkey = d * N + s
    pos = torch.searchsorted(key, nkey).clamp(max=key.numel() - 1)
    assert not bool((key[pos] == nkey).any())               # no sampled pair is an edge
    assert 0.4 < float((s < N // 2).double().mean()) < 0.6  # uniform, not degree-weighted
    gen = torch.Generator(device="cuda").manual_seed(2)
    W = torch.randn(512, 128, device="cuda", generator=gen) * 0.2
    code = torch.randint(0, 512, (N,), device="cuda", generator=gen, dtype=torch.int32)
    zq = W[code.long()]
    outs = []
    for _ in range(2):
        x = zq.clone().requires_grad_(True)
        ns, nc = _PairBCECodes.apply(x, neg, True, W, code)
        (ns / nc).backward()
        outs.append((ns.detach(), nc, x.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    assert int(outs[0][1]) == int(ok.sum())
    z = zq.double()
    grad = torch.zeros_like(z)
    loss = torch.zeros((), dtype=torch.float64, device="cuda")
    M = s.numel()
    for lo in range(0, M, 2_000_000):
        si, di = s[lo:lo + 2_000_000], d[lo:lo + 2_000_000]
        zi, zj = z[si], z[di]
        p = torch.sigmoid((zi * zj).sum(1))
        loss += (-torch.log(1 - p + 1e-15)).sum() / M
        c = p * (1 - p) / (1 - p + 1e-15) / M
        grad.index_add_(0, si, c[:, None] * zj)
        grad.index_add_(0, di, c[:, None] * zi)
    close(outs[0][0] / outs[0][1], loss)
    close(outs[0][2], grad)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_config3_vq_properties(dt):
    """Config 3 at full size (4 Mi x 512 x 256).  Besides the structural properties, EVERY row's code is checked against exact
    (fp64) scores computed by torch on the GPU in 256 Ki-row chunks:
      fp32: d64 = |z|^2 + |e|^2 - 2 <z, e>; the returned code is the lowest-index exact argmin, or its exact distance
            exceeds the minimum by less than the fp32 rounding bound of the distance (2 (D+3) 2^-24 (|z| + max|e|)^2);
      bf16: s64 = <z, bf16(W)> - |bf16(W)|^2 / 2 on the bf16-rounded operands; top - got <= the fp32-accumulation bound
            4 D 2^-24 max_c sum_k |z_k e_ck|.
    The number of rows that differ from the exact argmin is printed.  Loss and perplexity are compared with fp64 sums."""
    from causal_eeg_vae_amd import VectorQuantizer
    B, n_e, D = 4 * 1024 * 1024, 512, 256
    torch.manual_seed(0)
    vq = VectorQuantizer(n_e, D, 0.25, return_one_hot=False).cuda()
    with torch.no_grad():
        vq.embedding.weight.normal_()
        z = torch.randn(B, D, device="cuda").to(dt)
        l1, q1, p1, _, i1 = vq(z)
        l2, q2, p2, _, i2 = vq(z)
        assert torch.equal(i1, i2) and torch.equal(l1, l2) and torch.equal(p1, p2)
        assert i1.dtype == torch.int64 and tuple(i1.shape) == (B, 1)
        assert int(i1.min()) >= 0 and int(i1.max()) < n_e
        hist = torch.bincount(i1[:, 0], minlength=n_e)
        assert int(hist.sum()) == B
        W = vq.embedding.weight.detach()
        Wd = (W.to(torch.bfloat16) if dt == torch.bfloat16 else W).double()
        ee = (Wd * Wd).sum(1)
        e_norm_max = float(ee.max().sqrt())
        n_diff, worst, sq = 0, 0.0, 0.0
        CH = 256 * 1024
        for lo in range(0, B, CH):
            sl = slice(lo, lo + CH)
            assert torch.equal(q1[sl], W[i1[sl, 0]].to(dt))  # z_q rows ARE (rounded) codebook rows
            zd = z[sl].double()
            got_i = i1[sl]
            dot = zd @ Wd.t()
            if dt == torch.float32:
                zz = (zd * zd).sum(1, keepdim=True)
                d = zz + ee - 2.0 * dot
                best = d.min(1, keepdim=True)
                excess = (d.gather(1, got_i) - best.values)[:, 0]
                bound = 2.0 * (D + 3) * 2.0 ** -24 * (zz[:, 0].sqrt() + e_norm_max) ** 2
                differs = best.indices != got_i
            else:
                sc = dot - 0.5 * ee
                best = sc.max(1, keepdim=True)
                excess = (best.values - sc.gather(1, got_i))[:, 0]
                bound = 4.0 * D * 2.0 ** -24 * (zd.abs() @ Wd.abs().t()).max(1).values
                differs = best.indices != got_i
            assert bool((excess <= bound).all()), f"rows {lo}..: excess {float(excess.max()):.3e} over bound"
            assert bool((excess[~differs[:, 0]] == 0).all())
            n_diff += int(differs.sum())
            worst = max(worst, float((excess / bound).max()))
            sq += float(((q1[sl].double() - zd) ** 2).sum())
            del zd, dot
        print(f"config 3 {dt}: {n_diff} of {B} rows differ from the exact argmin (worst excess / bound = {worst:.3e})")
        assert n_diff <= B // 1000
        want_loss = 1.25 * sq / (B * D)
        np.testing.assert_allclose(float(l1), want_loss, rtol=1e-5 if dt == torch.float32 else 1e-2)
        p = hist.double() / B
        np.testing.assert_allclose(float(p1), float(torch.exp(-(p * torch.log(p + 1e-10)).sum())), rtol=1e-5)
        del q2, l2, p2, i2
        if dt == torch.float32:
            _, _, _, _, i3 = vq(q1)                         # idempotence: a codebook row is its own nearest code
            assert torch.equal(i3, i1)


def test_config2_gatconv_properties(config2):
    """GATConv (next row N1) at the full graph size, through properties: softmax weights of every destination sum to 1
    (read off by aggregating a column of ones), the hub sources' attention gradients summed by pieces equal an fp64
    scatter-add of the per-entry values, forward and every gradient bitwise reproducible."""
    from causal_eeg_vae_amd import GATConv, _lib
    N, E, ei, g = config2
    L = _lib.lib()
    pair = GATConv.attention_graph(ei, N)
    assert pair.max_trow > 10_000 and pair.long_items(True) is not None  # a hub source: thousands of outgoing edges
    torch.manual_seed(0)
    conv = GATConv(32, 64, heads=1).cuda()
    with torch.no_grad():
        conv.lin_src.weight.zero_()       # projected features = 0 -> out = bias + sum_j w_ij * 0; use the bias-free identity below
    x = torch.randn(N, 32, device="cuda")
    out0 = conv(x, pair)
    assert torch.equal(out0, conv.bias.detach().expand(N, -1))           # zero features: only the epilogue bias is left
    torch.manual_seed(1)
    conv = GATConv(32, 64, heads=1).cuda()
    xr = x.clone().requires_grad_(True)
    r = torch.randn(N, 64, device="cuda")
    res = []
    for _ in range(2):
        conv.zero_grad(set_to_none=True)
        xr.grad = None
        out = conv(xr, pair)
        (out * r).sum().backward()
        res.append((out.detach().clone(), xr.grad.clone(), [p.grad.clone() for p in conv.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    assert all(bool(torch.isfinite(t).all()) for t in res[0][2])
    # segment sums by pieces at full size against an fp64 scatter-add
    v = torch.randn(pair.nnz, device="cuda")
    items = pair.long_items(True)
    chunk = int(L.ceeg_decoder_codes_chunk())
    ws = _lib.workspace(L.ceeg_segment_sum_items_ws_bytes(items.shape[0]), v.device)
    got = torch.empty(N, device="cuda")
    _lib.check(L.ceeg_segment_sum_perm_items(pair.t_rowptr.data_ptr(), pair.t_perm.data_ptr(), v.data_ptr(), N, items.data_ptr(), items.shape[0],
                                             chunk, got.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "items")
    src = v[pair.t_perm.long()].double()
    trp = pair.t_rowptr.long()

    def seg(t):  # fp64 prefix sums, differenced at the row boundaries (no atomics on the hub rows)
        cs = torch.cat([torch.zeros(1, dtype=torch.float64, device="cuda"), t.cumsum(0)])
        return cs[trp[1:]] - cs[trp[:-1]]

    ref, mag = seg(src), seg(src.abs())
    assert float(((got.double() - ref).abs() / (mag + 1e-30)).max()) < 1e-5


def test_uniform_step_above_pair_key_limit():
    """N >= 2^22 - 1 does not fit the 32-bit incidence keys (node << 10 | code): the literal sampler step must fall back to the
    CSR-building form and still produce a finite loss, the full pair count and gradients for every parameter."""
    from causal_eeg_vae_amd import VGAEVQ, EventGraph, _lib
    from causal_eeg_vae_amd.synth import synth_event_graph
    N, E, Fi, D, n_e = (1 << 22) + 1000, 8_000_000, 16, 64, 64
    assert _lib.lib().ceeg_pair_bce_codes_supported(n_e, D, N, E) == 0
    assert _lib.lib().ceeg_pair_bce_codes_supported(n_e, D, (1 << 22) - 2, E) == 1
    ei = synth_event_graph(N, E, seed=5, device="cuda")
    g = EventGraph(ei, N)
    torch.manual_seed(0)
    model = VGAEVQ(Fi, D, n_e, 0.25, seed=1, negatives="uniform").cuda()
    x = torch.randn(N, Fi, device="cuda")
    loss, aux = model(x, g, step=0)
    loss.backward()
    assert bool(torch.isfinite(loss)) and int(aux["dec_sums"][2]) >= E - 16
    neg = aux["neg_edge_index"]
    assert int(neg.max()) < N and bool((neg[0] != neg[1]).all() | (neg[0] < 0).any())
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name


@pytest.mark.parametrize("n_in,n_out,relu", [(128, 256, True), (256, 256, False)])
def test_config2_dense_products_with_the_tuned_table(n_in, n_out, relu):
    """The step's dense products at config-2 size with the tuned GEMM table active (library kernels chosen per shape,
    causal_eeg_vae_amd/tuned_gemm_gfx950.csv): y = x W^T (+ b, ReLU), dx = g W and the split-K weight gradient against float64
    on sampled rows / in full, at the north-star tolerance; two runs bitwise equal."""
    from causal_eeg_vae_amd import _lib
    from causal_eeg_vae_amd.models import _Linear
    _lib.enable_tuned_gemms()
    N = 1_000_000
    gen = torch.Generator(device="cuda").manual_seed(n_in)
    x = torch.randn(N, n_in, device="cuda", generator=gen).requires_grad_(True)
    w = (torch.randn(n_out, n_in, device="cuda", generator=gen) / n_in ** 0.5).requires_grad_(True)
    b = torch.randn(n_out, device="cuda", generator=gen).requires_grad_(True) if relu else None
    g = torch.randn(N, n_out, device="cuda", generator=gen)
    outs = []
    for _ in range(2):
        for t in (x, w, b):
            if t is not None:
                t.grad = None
        y = _Linear.apply(x, w, b, relu)
        y.backward(g)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    assert all(torch.equal(a, c) for a, c in zip(outs[0], outs[1]))
    y, dx, dw = outs[0]
    rows = torch.randint(0, N, (4096,), device="cuda", generator=gen)
    y64 = x.detach()[rows].double() @ w.detach().double().t()
    if relu:
        y64 = (y64 + b.detach().double()).relu()
    tol = lambda ref: 1e-5 * ref.abs() + 1e-6 * float(ref.abs().max())
    assert bool(((y[rows].double() - y64).abs() <= tol(y64)).all())
    gm = g * (y > 0) if relu else g
    dx64 = gm[rows].double() @ w.detach().double()
    assert bool(((dx[rows].double() - dx64).abs() <= tol(dx64)).all())
    # the weight gradient is a 1M-deep sum of random-sign terms (|sum| ~ 10^3, sum of |terms| ~ 6 x 10^5): an fp32 accumulation
    # cannot hold rtol 1e-5 of the (cancelled) result; the bound is stated against the sum of absolute terms instead
    dw64 = torch.zeros(n_out, n_in, dtype=torch.float64, device="cuda")
    dwabs = torch.zeros_like(dw64)
    for lo in range(0, N, 125_000):
        gs, xs = gm[lo:lo + 125_000].double(), x.detach()[lo:lo + 125_000].double()
        dw64 += gs.t() @ xs
        dwabs += gs.abs().t() @ xs.abs()
    ratio = float(((dw.double() - dw64).abs() / dwabs).max())
    print(f"dW: max |err| / sum|terms| = {ratio:.3e}")
    assert ratio <= 1e-7


def test_config4_minibatch_structures_derived_equal_sorted(monkeypatch):
    """BASELINE configs[3] at full size (64 graphs x 100 k edges x 6250 nodes, collated on the device): the A + I pair derived
    from the raw-edge pair (ceeg_csr_add_self_loops) equals the one built by its own sort and transpose, array for array, and
    the structural properties hold: one loop entry per node, sorted rows, t_perm a permutation that maps every transposed entry
    to the forward entry with swapped endpoints."""
    from causal_eeg_vae_amd import CollatedBatch
    from causal_eeg_vae_amd.synth import synth_event_graph
    G, n, e = 64, 6250, 100000
    eis = [synth_event_graph(n, e, seed=4096 * 7 + g, device="cuda") for g in range(G)]
    eis[3][:, :50] = eis[3][0, :50]                           # 50 input self-loops (several per node) in one graph
    cb = CollatedBatch(eis, [n] * G)
    monkeypatch.setenv("CEEG_GCN_FROM_RAW", "1")
    a = cb.event_graph()
    monkeypatch.setenv("CEEG_GCN_FROM_RAW", "0")
    b = cb.event_graph()
    N = G * n
    assert a.gcn.nnz == b.gcn.nnz and a.gcn.max_row == b.gcn.max_row and a.gcn.max_trow == b.gcn.max_trow
    for name in ("rowptr", "col", "perm", "t_rowptr", "t_col", "t_perm"):
        assert torch.equal(getattr(a.gcn, name), getattr(b.gcn, name)), name
    for t in (False, True):
        ia, ib = a.gcn.long_items(t), b.gcn.long_items(t)
        assert (ia is None and ib is None) or torch.equal(ia, ib)
    assert torch.equal(a.gcn_val, b.gcn_val) and torch.equal(a.gcn_tval, b.gcn_tval)
    p = a.gcn
    rows = torch.repeat_interleave(torch.arange(N, device="cuda"), (p.rowptr[1:] - p.rowptr[:-1]).long())
    assert int((p.col.long() == rows).sum()) == N                                  # exactly one loop entry per node ...
    assert p.nnz == a.dec.nnz - 50 + N                                            # ... and the 50 input loops are gone
    key = rows * N + p.col.long()
    assert bool((key[1:] >= key[:-1]).all())                                       # (row, col) ascending
    assert torch.equal(torch.sort(p.t_perm.long()).values, torch.arange(p.nnz, device="cuda"))
    t_rows = torch.repeat_interleave(torch.arange(N, device="cuda"), (p.t_rowptr[1:] - p.t_rowptr[:-1]).long())
    assert torch.equal(p.col.long()[p.t_perm.long()], t_rows) and torch.equal(rows[p.t_perm.long()], p.t_col.long())
