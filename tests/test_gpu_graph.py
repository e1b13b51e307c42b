"""GPU parity: CSR build / transpose / gcn_norm / SpMM / samplers through the C-ABI vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand_graph(N, E, seed, loops=True, hubs=0):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, N, E)
    dst = rng.integers(0, N, E)
    if hubs:
        src[: hubs] = 3          # a source hub (long row in the transpose)
        dst[hubs: 2 * hubs] = 5  # a destination hub (long row in the CSR)
    if loops:
        src[-7:] = dst[-7:]      # a few pre-existing self-loops, one duplicated
        src[-1] = src[-2]; dst[-1] = dst[-2]
    return src.astype(np.int64), dst.astype(np.int64)


@pytest.mark.parametrize("N,E,self_loops,hubs", [(50, 300, True, 0), (50, 300, False, 0), (1000, 20000, True, 700),
                                                 (4000, 3000, True, 0), (7, 0, True, 0), (100000, 1600000, True, 5000)])
def test_csr_build_transpose_bit_exact(N, E, self_loops, hubs):
    from causal_eeg_vae_amd import CsrPair
    from oracle import ref_torch
    src, dst = _rand_graph(N, E, N + E, loops=E > 10, hubs=hubs)
    ei = torch.from_numpy(np.stack([src, dst])).cuda()
    p = CsrPair(ei, N, self_loops)
    rowptr, col, perm = ref_torch.csr_build_np(src, dst, N, self_loops)
    assert p.nnz == rowptr[-1]
    np.testing.assert_array_equal(p.rowptr.cpu().numpy(), rowptr)
    np.testing.assert_array_equal(p.col.cpu().numpy(), col)
    np.testing.assert_array_equal(p.perm.cpu().numpy(), perm)
    t_rowptr, t_col, t_perm = ref_torch.csr_transpose_np(rowptr, col, N)
    np.testing.assert_array_equal(p.t_rowptr.cpu().numpy(), t_rowptr)
    np.testing.assert_array_equal(p.t_col.cpu().numpy(), t_col)
    np.testing.assert_array_equal(p.t_perm.cpu().numpy(), t_perm)
    # sortedness property (the canonical (dst, src) order)
    key = np.repeat(np.arange(N), np.diff(rowptr)) * (N + 1) + p.col.cpu().numpy()
    assert (np.diff(key) >= 0).all()
    # the record the device filled (one host read): longest rows and the piece lists of the long rows, against a plain walk
    from causal_eeg_vae_amd.models import build_long_items
    chunk = 256
    for rp, mr, items in ((rowptr, p.max_row, p.long_items(False)), (t_rowptr, p.max_trow, p.long_items(True))):
        lens = np.diff(rp)
        assert mr == (int(lens.max()) if N else 0)
        want = []
        for r in np.nonzero(lens > chunk)[0]:
            lo, hi = int(rp[r]), int(rp[r + 1])
            for c in range(lo // chunk, (hi - 1) // chunk + 1):
                want.append((max(lo, c * chunk), min(hi, (c + 1) * chunk), 2 * c + int(c == lo // chunk and lo != c * chunk), int(r)))
        if not want:
            assert items is None
        else:
            assert items.cpu().numpy().tolist() == [list(w) for w in want]
            assert torch.equal(items, build_long_items(torch.from_numpy(rp.astype(np.int32)).cuda(), chunk))


@pytest.mark.parametrize("from_raw", ["1", "0"])
@pytest.mark.parametrize("N,E", [(3000, 50000), (1, 3), (2, 5), (70, 40)])
def test_event_graph_single_read_construction(N, E, from_raw, monkeypatch):
    """EventGraph(with_dec=True, checks=...) builds both pairs and reads every host-side quantity in one record; ids outside
    [0, N) and input self-loops make the entry counts data-dependent (nnz < capacity) and the structures still equal the
    oracle's.  from_raw = 1: A + I derived from the raw-edge pair (ceeg_csr_add_self_loops), 0: built by its own sort - the input
    has repeated self-loops on one node, repeated edges, nodes without any edge and (N = 1) nothing but loops."""
    from causal_eeg_vae_amd import EventGraph
    from oracle import ref_torch
    monkeypatch.setenv("CEEG_GCN_FROM_RAW", from_raw)
    src, dst = _rand_graph(N, E, 5, hubs=max(N // 5, 1))
    if N >= 70:
        src[:5] = -1; dst[5:9] = N + 3                    # skipped by the builder
        src[9:12] = dst[9:12] = 17                        # three loops on one node: the last one's edge id survives
        src[12:14] = dst[12:14] = N - 1
        src[14:17] = 3; dst[14:17] = 40                   # a repeated edge
        keep = (src != 33) & (dst != 33)                  # node 33: no edge at all
        src, dst = np.where(keep, src, 1), np.where(keep, dst, 2)
        E = src.shape[0]
    ei = torch.from_numpy(np.stack([src, dst])).cuda()
    g = EventGraph(ei, N, with_dec=True, checks={"yes": torch.ones((), dtype=torch.bool, device="cuda"), "no": (ei[0] > N + 5).any()})
    assert g.checks == {"yes": True, "no": False} and g._dec is not None
    ok = (src >= 0) & (src < N) & (dst >= 0) & (dst < N)
    eid = np.nonzero(ok)[0]
    for pair, loops in ((g.gcn, True), (g.dec, False)):
        rowptr, col, perm = ref_torch.csr_build_np(src[ok], dst[ok], N, loops)
        assert pair.nnz == rowptr[-1] <= E + (N if loops else 0) and (N < 70 or pair.nnz < E + (N if loops else 0))
        np.testing.assert_array_equal(pair.rowptr.cpu().numpy(), rowptr)
        np.testing.assert_array_equal(pair.col.cpu().numpy(), col)
        np.testing.assert_array_equal(pair.perm.cpu().numpy(), np.where(perm >= 0, eid[np.maximum(perm, 0)], -1))
        t_rowptr, t_col, t_perm = ref_torch.csr_transpose_np(rowptr, col, N)
        np.testing.assert_array_equal(pair.t_rowptr.cpu().numpy(), t_rowptr)
        np.testing.assert_array_equal(pair.t_col.cpu().numpy(), t_col)
        np.testing.assert_array_equal(pair.t_perm.cpu().numpy(), t_perm)
        assert pair.max_row == int(np.diff(rowptr).max()) and pair.max_trow == int(np.diff(t_rowptr).max())


@pytest.mark.parametrize("weighted", [False, True])
def test_gcn_norm(weighted):
    from causal_eeg_vae_amd import EventGraph
    from oracle import ref_torch
    N, E = 3000, 40000
    src, dst = _rand_graph(N, E, 11, hubs=900)
    ew = np.random.default_rng(1).uniform(0.5, 5.0, E).astype(np.float32) if weighted else None
    g = EventGraph(torch.from_numpy(np.stack([src, dst])).cuda(), N, None if ew is None else torch.from_numpy(ew).cuda())
    rowptr, col, perm = ref_torch.csr_build_np(src, dst, N, True)
    dinv, val = ref_torch.gcn_norm_np(rowptr, col, perm, ew, N)
    np.testing.assert_allclose(g.dinv.cpu().numpy(), dinv, rtol=1e-5)
    np.testing.assert_allclose(g.gcn_val.cpu().numpy(), val, rtol=1e-5)


@pytest.mark.parametrize("F", [4, 8, 40, 128, 256, 512])
def test_spmm_vs_oracle_and_determinism(F):
    from causal_eeg_vae_amd import EventGraph
    from causal_eeg_vae_amd.models import _spmm
    from oracle import ref_torch
    N, E = 5000, 80000
    src, dst = _rand_graph(N, E, F, hubs=3000)  # rows of 3000 entries: exercises the long-row split (chunk 256)
    g = EventGraph(torch.from_numpy(np.stack([src, dst])).cuda(), N)
    x = torch.randn(N, F, device="cuda")
    bias = torch.randn(F, device="cuda")
    rowptr, col, perm = ref_torch.csr_build_np(src, dst, N, True)
    _, val = ref_torch.gcn_norm_np(rowptr, col, perm, None, N)
    out = _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x, N, bias=bias)
    ref = ref_torch.spmm_np(rowptr, col, val, x.cpu().numpy()) + bias.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6 * scale)
    out2 = _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x, N, bias=bias)
    assert torch.equal(out, out2)  # bitwise reproducible
    # transposed pass with permuted values == A^T x
    t_rowptr, t_col, t_perm = ref_torch.csr_transpose_np(rowptr, col, N)
    outT = _spmm(g.gcn.t_rowptr, g.gcn.t_col, g.gcn_val, g.gcn.t_perm, x, N, relu=True)
    refT = np.maximum(ref_torch.spmm_np(t_rowptr, t_col, val[t_perm], x.cpu().numpy()), 0)
    np.testing.assert_allclose(outT.cpu().numpy(), refT, rtol=1e-5, atol=1e-6 * np.abs(refT).max())
    # linearity: A(2x + y) == 2 A x + A y
    y = torch.randn(N, F, device="cuda")
    lhs = _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, 2 * x + y, N)
    rhs = 2 * _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x, N) + _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, y, N)
    # both sides fp32, each within rounding of the exact sum: they differ by at most a few ulp of sum |terms|
    mag = _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, (2 * x + y).abs(), N) + 1e-30
    assert float(((lhs - rhs).abs() / mag).max()) < 1e-5


@pytest.mark.parametrize("N,E", [(1, 0), (2, 1), (7, 12), (33, 100), (96, 145), (130, 900)])
def test_spmm_tiny_graphs(N, E):
    """Graphs smaller than one queue draw per workgroup / XCD (every row must still be drained), empty edge lists."""
    from causal_eeg_vae_amd import EventGraph
    from causal_eeg_vae_amd.models import _spmm
    from oracle import ref_torch
    rng = np.random.default_rng(N)
    src = rng.integers(0, N, E)
    dst = rng.integers(0, N, E)
    g = EventGraph(torch.from_numpy(np.stack([src, dst])).cuda(), N)
    rowptr, col, perm = ref_torch.csr_build_np(src, dst, N, True)
    _, val = ref_torch.gcn_norm_np(rowptr, col, perm, None, N)
    t_rowptr, t_col, t_perm = ref_torch.csr_transpose_np(rowptr, col, N)
    for F in (4, 16, 128):
        x = torch.randn(N, F, device="cuda")
        out = _spmm(g.gcn.rowptr, g.gcn.col, g.gcn_val, None, x, N)
        ref = ref_torch.spmm_np(rowptr, col, val, x.cpu().numpy())
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6 * max(np.abs(ref).max(), 1e-30))
        outT = _spmm(g.gcn.t_rowptr, g.gcn.t_col, g.gcn_tval, None, x, N)
        refT = ref_torch.spmm_np(t_rowptr, t_col, val[t_perm], x.cpu().numpy())
        np.testing.assert_allclose(outT.cpu().numpy(), refT, rtol=1e-5, atol=1e-6 * max(np.abs(refT).max(), 1e-30))


@pytest.mark.parametrize("prefilter", ["1", "0"])
@pytest.mark.parametrize("N,E", [(200, 6000), (5000, 40000)])
def test_negative_sampling_bit_exact(N, E, prefilter, monkeypatch):
    """Samples bit-exact with the C oracle, with and without the edge-set bitmap prefilter (a clear bit proves "not an edge";
    set bits still take the exact search) - N = 200 / E = 6000 is dense enough that rejections happen."""
    from causal_eeg_vae_amd import negative_sampling, CsrPair
    from oracle import c_oracle, ref_torch
    monkeypatch.setenv("CEEG_NEG_FILTER", prefilter)
    src, dst = _rand_graph(N, E, 5, loops=False)
    ei = torch.from_numpy(np.stack([src, dst])).cuda()
    neg = negative_sampling(ei, N, 5000, seed=77)
    rowptr, col, _ = ref_torch.csr_build_np(src, dst, N, False)
    s, d = c_oracle.neg_sample_uniform(rowptr, col, N, 5000, 77)
    keep = s >= 0
    np.testing.assert_array_equal(neg.cpu().numpy(), np.stack([s[keep], d[keep]]))
    edges = set(zip(src.tolist(), dst.tolist()))
    assert all((a, b) not in edges and a != b for a, b in neg.t().tolist())


def test_philox_eps_matches_oracle():
    from causal_eeg_vae_amd.models import _ReparamKL
    from oracle import c_oracle
    N, D = 1000, 64
    mu = torch.zeros(N, D, device="cuda"); ls = torch.zeros(N, D, device="cuda")
    z, kl = _ReparamKL.apply(mu, ls, None, 4242, True, False)  # z = eps exactly
    eps = c_oracle.philox_normal(4242, N * D).reshape(N, D)
    # the Philox integers are bit-exact; the floats go through logf / sincosf of two different libms (device, glibc)
    np.testing.assert_allclose(z.cpu().numpy(), eps, rtol=1e-5, atol=1e-6 * float(np.abs(eps).max()))
    assert abs(kl.item()) < 1e-6


@pytest.mark.parametrize("N,F,ld", [(1, 4, 4), (777, 32, 32), (5000, 64, 128), (100_003, 256, 256), (300_001, 128, 128)])
def test_bias_relu_and_column_sums(N, F, ld):
    """Streaming helpers of a GCN layer (include/ceeg.h: ceeg_bias_relu / ceeg_relu_bwd_colsum / ceeg_colsum) against
    torch: the elementwise parts bit-exact, the column sums within fp32 summation error of an fp64 sum, and
    identical between two runs (fixed-order reduction)."""
    from causal_eeg_vae_amd import _lib
    from causal_eeg_vae_amd.models import _colsum, _relu_bwd_colsum
    L = _lib.lib()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(N + F)
    ybuf = torch.randn(N, ld, generator=gen).to(dev)
    y = ybuf[:, :F]
    bias = torch.randn(F, generator=gen).to(dev)
    want = torch.relu(y + bias)
    got = ybuf.clone()
    _lib.check(L.ceeg_bias_relu(got.data_ptr(), ld, N, F, bias.data_ptr(), 1, _lib.stream()), "ceeg_bias_relu")
    torch.cuda.synchronize()
    assert torch.equal(got[:, :F], want)
    if ld > F:
        assert torch.equal(got[:, F:], ybuf[:, F:])  # padding columns untouched

    g = torch.randn(N, F, generator=gen).to(dev)
    gm, cs = _relu_bwd_colsum(g, want)
    gm2, cs2 = _relu_bwd_colsum(g, want)
    ref = g * (want > 0)
    assert torch.equal(gm, ref)
    assert torch.equal(cs, cs2)
    ref64 = ref.double().sum(0)
    scale = ref.abs().double().sum(0) + 1e-30
    assert float(((cs.double() - ref64).abs() / scale).max()) < 1e-6
    cs3 = _colsum(g)
    assert torch.equal(cs3, _colsum(g))
    assert float(((cs3.double() - g.double().sum(0)).abs() / (g.abs().double().sum(0) + 1e-30)).max()) < 1e-6


@pytest.mark.parametrize("N,F,ld", [(1, 4, 4), (1000, 32, 32), (777, 36, 40), (5000, 256, 256)])
def test_row_squared_norms(N, F, ld):
    """ceeg_row_sqnorm against float64 torch; the k-ascending fmaf chain restated on the CPU gives the same bits."""
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    torch.manual_seed(N + F)
    x = torch.randn(N, ld, device="cuda")
    out = torch.empty(N, device="cuda")
    _lib.check(L.ceeg_row_sqnorm(x.data_ptr(), ld, N, F, out.data_ptr(), _lib.stream()), "ceeg_row_sqnorm")
    ref = x[:, :F].double().square().sum(1)
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5)
    xc = x[: min(N, 50), :F].cpu().numpy()
    chain = np.zeros(xc.shape[0], dtype=np.float32)
    for k in range(F):   # fmaf(a, a, acc) = the float64 product-sum rounded once (exact in float64 for float32 operands)
        chain = (xc[:, k].astype(np.float64) * xc[:, k].astype(np.float64) + chain.astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(out[: xc.shape[0]].cpu().numpy(), chain)
    assert L.ceeg_row_sqnorm(x.data_ptr(), ld, N, 6, out.data_ptr(), _lib.stream()) == -1   # F % 4 != 0


@pytest.mark.parametrize("sizes", [[500] * 6, [1, 7, 300, 64, 1000], [8192, 10]])
def test_collated_batch_block_keyed_build_equals_general(sizes, monkeypatch):
    """CollatedBatch.event_graph sorts its raw-edge CSR on 32-bit keys that hold the source relative to its graph
    (ceeg_csr_build_blocks); CEEG_CSR_BLOCKS=0 keeps the general 64-bit keys: every array of both CSR pairs equal, for equal and
    ragged graph sizes, duplicate edges and self-loops included.  A list with a local id outside its graph (an edge that leaves its
    block) fails the device-side check and the construction falls back: again the general build's arrays."""
    from causal_eeg_vae_amd import _lib
    from causal_eeg_vae_amd.collate import CollatedBatch
    rng = np.random.default_rng(len(sizes))
    eis = []
    for n in sizes:
        e = 0 if n == 1 else int(rng.integers(1, 12 * n))
        ei = rng.integers(0, n, size=(2, e))
        if e > 4:
            ei[:, 1] = ei[:, 0]          # a duplicate edge
            ei[1, 2] = ei[0, 2]          # a self-loop
        eis.append(torch.from_numpy(ei).cuda())

    def arrays(g):
        out = []
        for p in (g.dec, g.gcn):
            out += [p.rowptr, p.col, p.perm, p.t_rowptr, p.t_col, p.t_perm]
        return out + [g.gcn_val, g.gcn_tval, g.dinv]

    for bad in (False, True):
        lists = [e.clone() for e in eis]
        if bad:
            k = max(range(len(sizes)), key=lambda i: lists[i].shape[1])
            lists[k][0, 0] = sizes[k] + 3          # not a local id of graph k: lands in the next graph's block (or past the batch)
        got = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("CEEG_CSR_BLOCKS", flag)
            cb = CollatedBatch(lists, sizes)
            g = cb.event_graph()
            got[flag] = arrays(g)
            assert bool(cb.outside[0] != 0) == bad
        for a, b in zip(got["1"], got["0"]):
            assert torch.equal(a, b)
    L = _lib.lib()
    assert L.ceeg_csr_build_blocks_supported(400_000, 6250) == 1 and L.ceeg_csr_build_blocks_supported(1 << 20, 1 << 13) == 0


def test_spmm_piece_list_equals_row_scan():
    """ceeg_spmm_csr_items (hub-row pieces from the graph's static list) == ceeg_spmm_csr (per-chunk row search), bit for
    bit, forward and transposed, with and without hub rows."""
    from causal_eeg_vae_amd import EventGraph
    from causal_eeg_vae_amd.models import _spmm
    from causal_eeg_vae_amd.synth import synth_event_graph
    for N, E in ((50_000, 800_000), (3_000, 20_000)):
        g = EventGraph(synth_event_graph(N, E, seed=4, device="cuda"), N)
        p = g.gcn
        if N == 50_000:
            assert p.max_trow > 256 and p.long_items(True) is not None
        for F in (256, 128, 20):
            x = torch.randn(N, F, device="cuda")
            for rp, col, val, mr, tr in ((p.rowptr, p.col, g.gcn_val, p.max_row, False), (p.t_rowptr, p.t_col, g.gcn_tval, p.max_trow, True)):
                a = _spmm(rp, col, val, None, x, N, max_row=mr)
                b = _spmm(rp, col, val, None, x, N, max_row=mr, items=p.long_items(tr))
                c = _spmm(rp, col, val, None, x, N)  # no hint at all
                assert torch.equal(a, b) and torch.equal(a, c)


def test_segment_sum_piece_list_equals_row_walk():
    """ceeg_segment_sum_perm_items (hub rows by pieces of the graph's static list) against ceeg_segment_sum_perm (one wave
    per row) and an fp64 scatter-add: with a permutation, without one, with and without hub rows; deterministic."""
    from causal_eeg_vae_amd import EventGraph, _lib
    from causal_eeg_vae_amd.synth import synth_event_graph
    L = _lib.lib()
    chunk = int(L.ceeg_decoder_codes_chunk())
    for N, E in ((50_000, 800_000), (3_000, 20_000)):
        p = EventGraph(synth_event_graph(N, E, seed=9, device="cuda"), N).gcn
        items = p.long_items(True)
        if N == 50_000:
            assert items is not None and p.max_trow > chunk
        n_items = 0 if items is None else items.shape[0]
        v = torch.randn(p.nnz, device="cuda")
        ws = _lib.workspace(L.ceeg_segment_sum_items_ws_bytes(n_items), v.device)
        for perm in (p.t_perm, None):
            a = torch.empty(N, device="cuda")
            b = torch.full((N,), float("nan"), device="cuda")
            b2 = torch.full((N,), float("nan"), device="cuda")
            _lib.check(L.ceeg_segment_sum_perm(p.t_rowptr.data_ptr(), _lib.ptr(perm), v.data_ptr(), N, a.data_ptr(), _lib.stream()), "plain")
            for out in (b, b2):
                _lib.check(L.ceeg_segment_sum_perm_items(p.t_rowptr.data_ptr(), _lib.ptr(perm), v.data_ptr(), N, _lib.ptr(items), n_items, chunk,
                                                         out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "items")
            assert torch.equal(b, b2)
            rows = torch.repeat_interleave(torch.arange(N, device="cuda"), (p.t_rowptr[1:] - p.t_rowptr[:-1]).long())
            src = v if perm is None else v[perm.long()]
            ref = torch.zeros(N, dtype=torch.float64, device="cuda").index_add_(0, rows, src.double())
            mag = torch.zeros(N, dtype=torch.float64, device="cuda").index_add_(0, rows, src.double().abs())
            assert float(((b.double() - ref).abs() / (mag + 1e-30)).max()) < 1e-5
            assert float(((a.double() - ref).abs() / (mag + 1e-30)).max()) < 1e-5


def test_gat_entry_points_reject_bad_arguments():
    """The GAT score / segment-sum entry points return error codes (no launch) for shapes they do not implement, a short
    workspace, or missing pointers - the same loud-failure rule as the rest of the C-ABI."""
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    N, H, C = 100, 2, 8
    xp = torch.randn(N, H * C, device="cuda")
    att = torch.randn(H * C, device="cuda")
    a = torch.empty(H, N, device="cuda")
    st = _lib.stream()
    assert L.ceeg_gat_scores_fwd(xp.data_ptr(), xp.stride(0), att.data_ptr(), att.data_ptr(), N, H, C, a.data_ptr(), a.data_ptr(), st) == 0
    assert L.ceeg_gat_scores_fwd(xp.data_ptr(), xp.stride(0), att.data_ptr(), att.data_ptr(), N, H, 6, a.data_ptr(), a.data_ptr(), st) < 0   # C % 4
    assert L.ceeg_gat_scores_fwd(xp.data_ptr(), 8, att.data_ptr(), att.data_ptr(), N, H, C, a.data_ptr(), a.data_ptr(), st) < 0            # ld < H*C
    assert L.ceeg_gat_scores_fwd(None, xp.stride(0), att.data_ptr(), att.data_ptr(), N, H, C, a.data_ptr(), a.data_ptr(), st) < 0
    d = torch.empty(2, H * C, device="cuda")
    ws = _lib.workspace(L.ceeg_gat_scores_bwd_ws_bytes(N, H, C), xp.device)
    dxp = torch.zeros_like(xp)
    args = (xp.data_ptr(), xp.stride(0), att.data_ptr(), att.data_ptr(), a.data_ptr(), a.data_ptr(), N, H, C, dxp.data_ptr(), dxp.stride(0),
            d[0].data_ptr(), d[1].data_ptr())
    assert L.ceeg_gat_scores_bwd(*args, ws.data_ptr(), 16, st) < 0                       # workspace too small
    assert L.ceeg_gat_scores_bwd(*args, None, ws.numel(), st) < 0
    assert L.ceeg_gat_scores_bwd(*args[:8], 2052, *args[9:], ws.data_ptr(), ws.numel(), st) < 0  # C > 2048
    rp = torch.zeros(N + 1, dtype=torch.int32, device="cuda")
    out = torch.empty(N, device="cuda")
    v = torch.zeros(1, device="cuda")
    items = torch.zeros(1, 4, dtype=torch.int32, device="cuda")
    assert L.ceeg_segment_sum_perm_items(rp.data_ptr(), None, v.data_ptr(), N, items.data_ptr(), 1, 256, out.data_ptr(), None, 0, st) < 0  # no workspace
    assert L.ceeg_segment_sum_perm_items(rp.data_ptr(), None, v.data_ptr(), N, None, 0, 256, out.data_ptr(), None, 0, st) == 0
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0


def test_device_collation_follow_batch_and_separate_round_trip():
    """CollatedBatch (ceeg_collate_edges + ceeg_segment_ids) against a per-graph restatement of
    what PyG's DataLoader(..., follow_batch=[...]) builds (scripts/run.py:310-326) - offset edge_index, `batch`, `ptr`, one
    `<key>_batch` per followed key - and `separate` (scripts/dataset.py:66-92) returning every sample bit for bit; graphs
    without edges, without nodes, and a followed key with zero rows in some samples."""
    from causal_eeg_vae_amd import CollatedBatch
    rng = np.random.default_rng(0)
    sizes = [5, 1, 0, 300, 17, 2000, 1]
    n_edges = [9, 0, 0, 4000, 40, 30000, 0]
    tok_rows = [5, 1, 0, 7, 0, 33, 2]            # a followed key with its own row count per sample (e.g. token rows)
    eis, toks = [], []
    for n, e, t in zip(sizes, n_edges, tok_rows):
        eis.append(torch.from_numpy(rng.integers(0, max(n, 1), (2, e))).to(torch.int64))
        toks.append(torch.from_numpy(rng.standard_normal((t, 6)).astype(np.float32)))
    c = CollatedBatch([e.cuda() for e in eis], sizes, follow={"input_ids": [t.cuda() for t in toks]})
    # the restatement: one graph at a time
    offs, want_ei, want_batch, want_tb = 0, [], [], []
    for g, (ei, n, t) in enumerate(zip(eis, sizes, toks)):
        want_ei.append(ei + offs)
        want_batch.append(torch.full((n,), g, dtype=torch.int64))
        want_tb.append(torch.full((t.shape[0],), g, dtype=torch.int64))
        offs += n
    assert c.num_nodes == offs and c.num_graphs == len(sizes)
    assert torch.equal(c.edge_index.cpu(), torch.cat(want_ei, dim=1))
    assert torch.equal(c.batch.cpu(), torch.cat(want_batch))
    assert torch.equal(c.edge_batch.cpu(), torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(n_edges)))
    assert c.ptr.cpu().tolist() == np.concatenate([[0], np.cumsum(sizes)]).tolist()
    assert torch.equal(c.input_ids.cpu(), torch.cat(toks)) and torch.equal(c.input_ids_batch.cpu(), torch.cat(want_tb))
    for g in range(len(sizes)):
        ei, n, parts = c.separate(g)
        assert n == sizes[g] and torch.equal(ei.cpu(), eis[g]) and torch.equal(parts["input_ids"].cpu(), toks[g])
    # the function the models use returns the same tensors
    from causal_eeg_vae_amd.synth import collate_graphs
    e2, b2, n2 = collate_graphs([e.cuda() for e in eis], sizes)
    assert n2 == offs and torch.equal(e2, c.edge_index) and torch.equal(b2, c.batch)
    L = __import__("causal_eeg_vae_amd")._lib.lib()
    assert L.ceeg_collate_edges(None, 0, c.ptr.data_ptr(), c.ptr.data_ptr(), 0, 5, 1, None, 0, None, None) < 0   # G = 0: refused


@pytest.mark.parametrize("F", [128, 256, 40])
def test_interleaved_value_stream_and_masked_aggregation_bitwise(F, monkeypatch):
    """EventGraph.aggregate (ceeg_spmm_csr_cv: (column, value) pairs in one 8-byte stream) equals ceeg_spmm_csr_items bit for bit -
    with bias + ReLU, hub rows included - and ceeg_spmm_csr_masked / the masked cv form equal the plain transposed pass followed by
    the ReLU mask (g * (y > 0))."""
    from causal_eeg_vae_amd import EventGraph, _lib
    from causal_eeg_vae_amd.models import _spmm, _spmm_cv
    N, E = 5000, 80000
    src, dst = _rand_graph(N, E, F + 1, hubs=3000)
    g = EventGraph(torch.from_numpy(np.stack([src, dst])).cuda(), N)
    p = g.gcn
    x = torch.randn(N, F, device="cuda")
    bias = torch.randn(F, device="cuda")
    want = _spmm(p.rowptr, p.col, g.gcn_val, None, x, N, bias=bias, relu=True, max_row=p.max_row, items=p.long_items(False))
    got = g.aggregate(x, bias=bias, relu=True)
    assert g._gcn_cv is not None and torch.equal(got, want)
    monkeypatch.setenv("CEEG_SPMM_CV", "0")
    assert torch.equal(g.aggregate(x, bias=bias, relu=True), want)
    y = torch.randn(N, F, device="cuda")
    plain = _spmm(p.t_rowptr, p.t_col, g.gcn_tval, None, x, N, max_row=p.max_trow, items=p.long_items(True))
    masked = _spmm(p.t_rowptr, p.t_col, g.gcn_tval, None, x, N, max_row=p.max_trow, items=p.long_items(True), mask=y)
    assert torch.equal(masked, plain * (y > 0))
    tcv = torch.stack([p.t_col, g.gcn_tval.view(torch.int32)], dim=1).contiguous()
    assert torch.equal(_spmm_cv(p.t_rowptr, tcv, x, N, p.nnz, mask=y, max_row=p.max_trow, items=p.long_items(True)), masked)
    L = _lib.lib()
    ptr = x.data_ptr()
    assert L.ceeg_spmm_csr_cv(p.rowptr.data_ptr(), None, ptr, F, N, p.nnz, F, None, 0, None, 0, ptr, F, 0, None, 0, ptr, 1 << 20, _lib.stream()) == -1
