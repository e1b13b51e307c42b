"""CPU: pin the oracle against the golden vectors written from the LIVE reference
(/root/reference/scripts/quantizer.py via tests/golden/make_vq_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, ref_torch

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vq_*.npz")))


def _ulp_bound(g):
    """Rounding bound on a distance: a few ulp of |z|^2+|e|^2 (SURVEY.md §7 hard part 1)."""
    z, W = g["z"].astype(np.float64), g["W"].astype(np.float64)
    mag = (z ** 2).sum(1) + (W ** 2).sum(1).max() + 2 * np.abs(z @ W.T).max(1)
    return 8 * np.spacing(mag.astype(np.float32)).astype(np.float64) + 1e-30


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_c_oracle_matches_reference_golden(path):
    g = np.load(path)
    o = c_oracle.vq_forward_f32(g["z"], g["W"], float(g["beta"]))
    ref_idx = g["idx"][:, 0]
    mism = np.nonzero(o["idx"] != ref_idx)[0]
    # any index mismatch vs the literal reference must be a rounding-sensitive row (top-2 gap below the bound)
    bound = _ulp_bound(g)
    z, W = g["z"].astype(np.float64), g["W"].astype(np.float64)
    for b in mism:
        d = ((z[b] - W) ** 2).sum(1)
        assert abs(d[o["idx"][b]] - d[ref_idx[b]]) <= bound[b], (b, d[o["idx"][b]], d[ref_idx[b]], bound[b])
    assert len(mism) <= max(1, len(ref_idx) // 50)
    same = o["idx"] == ref_idx
    np.testing.assert_allclose(o["z_q"][same], g["z_q"][same], rtol=1e-5, atol=1e-6)
    if len(mism) == 0:
        np.testing.assert_allclose(o["loss"], g["loss"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(o["perplexity"], g["perplexity"], rtol=1e-5)


@pytest.mark.parametrize("name", ["tie_dup_rows", "tie_all_equal", "tie_exact_hit"])
def test_tie_break_lowest_index(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"vq_{name}.npz"))
    o = c_oracle.vq_forward_f32(g["z"], g["W"], float(g["beta"]))
    if name == "tie_all_equal":
        assert (o["idx"] == 0).all() and (g["idx"] == 0).all()
    if name == "tie_dup_rows":
        assert (o["idx"] < g["W"].shape[0] // 2).all()
    np.testing.assert_array_equal(o["idx"], g["idx"][:, 0])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_torch_restatement_matches_reference_golden(path):
    g = np.load(path)
    n_e, D = g["W"].shape
    vq = ref_torch.RefVectorQuantizer(n_e, D, float(g["beta"]))
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.from_numpy(g["W"]))
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    loss, zq, ppl, onehot, idx = vq(z)
    (loss + (zq * torch.from_numpy(g["r"])).sum()).backward()
    np.testing.assert_array_equal(idx.numpy(), g["idx"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(ppl.item(), g["perplexity"], rtol=1e-6)
    np.testing.assert_allclose(zq.detach().numpy(), g["z_q"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(z.grad.numpy(), g["dz"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(vq.embedding.weight.grad.numpy(), g["dW"], rtol=1e-5, atol=1e-7)
    assert onehot.shape == (g["z"].shape[0], n_e)


def test_philox_known_answer():
    # Random123 known-answer test for philox4x32-10: counter = key = 0
    out = c_oracle.philox_raw(0, 1, 0)  # counter (0,0,tag=0,0), key 0
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


def test_philox_normal_moments():
    e = c_oracle.philox_normal(1234, 1 << 16)
    assert abs(e.mean()) < 0.02 and abs(e.std() - 1) < 0.02 and np.isfinite(e).all()


def test_gcn_norm_known_answers():
    # path 0->1->2 plus an isolated node 3 and a pre-existing self-loop on 1
    src = np.array([0, 1, 1]); dst = np.array([1, 2, 1])
    rowptr, col, perm = ref_torch.csr_build_np(src, dst, 4, True)
    assert rowptr.tolist() == [0, 1, 3, 5, 6]
    assert col.tolist() == [0, 0, 1, 1, 2, 3]
    assert perm.tolist() == [-1, 0, 2, 1, -1, -1]  # node 1's loop keeps edge 2's identity
    dinv, val = ref_torch.gcn_norm_np(rowptr, col, perm, None, 4)
    np.testing.assert_allclose(dinv, [1, 2 ** -0.5, 2 ** -0.5, 1])
    np.testing.assert_allclose(val, [1, 2 ** -0.5, 0.5, 0.5, 0.5, 1])
    # same through the module restatement
    conv = ref_torch.RefGCNConv(2, 2)
    with torch.no_grad():
        conv.lin.weight.copy_(torch.eye(2))
    x = torch.tensor([[1., 0.], [0., 1.], [1., 1.], [2., 2.]])
    out = conv(x, torch.tensor([src, dst]))
    exp = ref_torch.spmm_np(rowptr, col, val, x.numpy())
    np.testing.assert_allclose(out.detach().numpy(), exp, rtol=1e-6)
    t_rowptr, t_col, t_perm = ref_torch.csr_transpose_np(rowptr, col, 4)
    assert t_rowptr.tolist() == [0, 2, 4, 5, 6] and t_col.tolist() == [0, 1, 1, 2, 2, 3]


def test_vgae_formulas_known_answers():
    mu = torch.tensor([[0.5, -1.0]]); ls = torch.tensor([[0.0, 11.0]])
    kl = ref_torch.kl_loss(mu, ls)
    exp = -0.5 * ((1 + 0 - 0.25 - 1) + (1 + 20 - 1 - np.exp(20.0)))
    np.testing.assert_allclose(kl.item(), exp, rtol=1e-6)
    z = torch.tensor([[1.0, 0.0], [1.0, 1.0], [-30.0, 0.0]])
    pos = torch.tensor([[0], [1]]); neg = torch.tensor([[0], [2]])
    l = ref_torch.recon_loss(z, pos, neg)
    s = 1 / (1 + np.exp(-1.0))
    np.testing.assert_allclose(l.item(), -np.log(s + 1e-15) - np.log(1 - 0.0 + 1e-15), rtol=1e-6)
    # saturation: the literal +EPS form caps at -log(1e-15)
    lsat = ref_torch.recon_loss(torch.tensor([[40.0], [1.0]]), torch.tensor([[0], [1]]), torch.tensor([[0], [1]]))
    np.testing.assert_allclose(lsat.item(), 0.0 + -np.log(np.float32(1e-15)), rtol=1e-5)


def test_mfma_bf16_step_model_reproduces_recorded_hardware_outputs():
    """oracle_mfma_bf16_step8 (the restatement of v_mfma_f32_16x16x32_bf16's arithmetic the bf16 quantiser oracle is built
    on) against outputs RECORDED on an MI355X by scripts/mfma_probe.hip (inputs: scripts/mfma_probe_gen*.py; fixture =
    28 MFMAs x 256 dot products: random operands over exponent spreads 0..30, cancellation probes, quantiser-like data).
    Every dot product must come out bit for bit (+0 == -0 allowed: the sign of an exact zero is not modelled)."""
    from oracle import c_oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mfma_bf16_probe.npz"))
    n = 0
    for fam in ("random", "cancel", "vq"):
        A, B, C, D = g[fam + "_A"], g[fam + "_B"], g[fam + "_C"], g[fam + "_D"]
        for t in range(A.shape[0]):
            for r in range(16):
                for c in range(16):
                    got = c_oracle.mfma_bf16_dot(A[t, r], B[t, c], float(C[t, r, c]))
                    want = D[t, r, c]
                    assert got.view(np.uint32) == want.view(np.uint32) or (got == 0 and want == 0), (fam, t, r, c, got, want)
                    n += 1
    assert n == 28 * 256
