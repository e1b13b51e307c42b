"""GPU parity: VectorQuantizer through the C-ABI (libceeg.so) vs the oracle and the reference golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vq_*.npz")))


def _mod(n_e, D, beta, W, one_hot=True):
    from causal_eeg_vae_amd import VectorQuantizer
    vq = VectorQuantizer(n_e, D, beta, return_one_hot=one_hot).cuda()
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.as_tensor(W))
    return vq


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[3:-4] for p in CASES])
def test_fp32_golden_and_oracle(path):
    from oracle import c_oracle
    g = np.load(path)
    n_e, D = g["W"].shape
    beta = float(g["beta"])
    vq = _mod(n_e, D, beta, g["W"])
    z = torch.from_numpy(g["z"]).cuda().requires_grad_(True)
    loss, z_q, ppl, onehot, idx = vq(z)
    (loss + (z_q * torch.from_numpy(g["r"]).cuda()).sum()).backward()
    # return-type fidelity (quantizer.py:67)
    assert idx.dtype == torch.int64 and idx.shape == (z.shape[0], 1)
    assert onehot.shape == (z.shape[0], n_e) and onehot.dtype == torch.float32
    assert loss.dim() == 0 and ppl.dim() == 0
    o = c_oracle.vq_forward_f32(g["z"], g["W"], beta)
    gi = idx[:, 0].cpu().numpy()
    np.testing.assert_array_equal(gi, o["idx"])  # bit-exact vs the CPU restatement (specified fmaf-chain order)
    assert torch.equal(onehot.argmax(1), idx[:, 0])
    np.testing.assert_allclose(loss.item(), o["loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ppl.item(), o["perplexity"], rtol=1e-5)
    # the literal reference (live scripts/quantizer.py, MKL summation order unknown): its indices are reproduced on every
    # golden case, so every float output and both gradients are compared with it unconditionally
    np.testing.assert_array_equal(gi, g["idx"][:, 0])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ppl.item(), g["perplexity"], rtol=1e-5)
    np.testing.assert_allclose(z_q.detach().cpu().numpy(), g["z_q"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(z.grad.cpu().numpy(), g["dz"], rtol=1e-5, atol=1e-6 * np.abs(g["dz"]).max())
    np.testing.assert_allclose(vq.embedding.weight.grad.cpu().numpy(), g["dW"], rtol=1e-5, atol=1e-6 * max(np.abs(g["dW"]).max(), 1e-30))


@pytest.mark.parametrize("B,n_e,D,init", [(5000, 512, 256, "normal"), (5000, 512, 256, "ref"), (3001, 512, 128, "normal"),
                                          (1000, 512, 8, "ref"), (777, 100, 24, "normal"), (130, 33, 200, "normal")])
def test_fp32_random_bit_exact_vs_oracle(B, n_e, D, init):
    from oracle import c_oracle
    rng = np.random.default_rng(B + n_e + D)
    W = rng.standard_normal((n_e, D)).astype(np.float32) if init == "normal" else \
        rng.uniform(-1.0 / n_e, 1.0 / n_e, (n_e, D)).astype(np.float32)
    zn = rng.standard_normal((B, D)).astype(np.float32)
    vq = _mod(n_e, D, 0.25, W, one_hot=False)
    loss, z_q, ppl, onehot, idx = vq(torch.from_numpy(zn).cuda())
    assert onehot is None
    o = c_oracle.vq_forward_f32(zn, W, 0.25)
    np.testing.assert_array_equal(idx[:, 0].cpu().numpy(), o["idx"])
    np.testing.assert_array_equal(z_q.detach().cpu().numpy(), o["z_q"])
    np.testing.assert_allclose(loss.item(), o["loss"], rtol=1e-5)
    np.testing.assert_allclose(ppl.item(), o["perplexity"], rtol=1e-5)


@pytest.mark.parametrize("B,n_e,D,init", [(4096, 512, 256, "normal"), (3000, 512, 256, "ref"), (1000, 512, 128, "normal"), (300, 96, 40, "normal"),
                                          (64, 512, 8, "ref"), (3001, 500, 64, "normal"), (2000, 512, 128, "ref")])
def test_bf16_bit_exact_vs_mfma_restatement(B, n_e, D, init, monkeypatch):
    """bf16 mode against oracle_vq_forward_bf16, the CPU restatement of the kernels' arithmetic on
    v_mfma_f32_16x16x32_bf16 (chained 8-product steps with the instruction's own truncation rules, pinned to recorded
    hardware outputs by tests/test_oracle_golden.py): codes and z_q BIT-EXACT, for both kernels (codebook-stationary and
    row-stationary), including the reference's near-tie-heavy init U(+-1/n_e) and duplicated codebook rows."""
    from oracle import c_oracle
    rng = np.random.default_rng(B + n_e + D)
    W = rng.standard_normal((n_e, D)).astype(np.float32) if init == "normal" else \
        rng.uniform(-1.0 / n_e, 1.0 / n_e, (n_e, D)).astype(np.float32)
    W[n_e // 2] = W[3]  # an exact tie: the lower index must win
    z = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).to(torch.bfloat16)
    z[:5] = torch.from_numpy(W[3]).to(torch.bfloat16)
    o = c_oracle.vq_forward_bf16(z.view(torch.int16).numpy().view(np.uint16), W, 0.25)
    vq = _mod(n_e, D, 0.25, W, one_hot=False)
    for bs in ("1", "0"):
        monkeypatch.setenv("CEEG_VQ_BF16_BS", bs)
        loss, z_q, ppl, _, idx = vq(z.cuda())
        np.testing.assert_array_equal(idx[:, 0].cpu().numpy(), o["idx"])
        np.testing.assert_array_equal(z_q.cpu().view(torch.int16).numpy().view(np.uint16), o["z_q"])
        np.testing.assert_allclose(loss.item(), o["loss"], rtol=1e-4)   # north star for bf16 floats: 1e-2
        np.testing.assert_allclose(ppl.item(), o["perplexity"], rtol=1e-5)
    assert (o["idx"][:5] == 3).all()


@pytest.mark.parametrize("B,n_e,D", [(4096, 512, 256), (1000, 512, 128), (300, 96, 40), (64, 512, 8)])
def test_bf16_indices_within_rounding_bound(B, n_e, D):
    from oracle import ref_torch
    torch.manual_seed(B + D)
    W = torch.randn(n_e, D)
    z = torch.randn(B, D).to(torch.bfloat16)
    vq = _mod(n_e, D, 0.25, W.numpy(), one_hot=False)
    zc = z.cuda().requires_grad_(True)
    loss, z_q, ppl, _, idx = vq(zc)
    assert z_q.dtype == torch.bfloat16
    sc = ref_torch.vq_bf16_scores64(z, W)
    best = sc.argmax(1)
    gi = idx[:, 0].cpu()
    got = sc.gather(1, gi[:, None])[:, 0]
    top = sc.max(1).values
    # fp32 accumulation of 256 exact bf16 products: |error| <= ~D * 2^-24 * sum|a b|; indices must be optimal within it
    bound = 4 * D * 2.0 ** -24 * (z.double().abs() @ W.to(torch.bfloat16).double().abs().t()).max(1).values
    assert ((top - got) <= bound).all()
    assert (gi == best).double().mean() > 0.999
    zq_ref = W[gi].to(torch.bfloat16)
    assert torch.equal(z_q.cpu(), zq_ref)
    m = ((W[gi].double() - z.double()) ** 2).mean()
    np.testing.assert_allclose(loss.item(), float(1.25 * m), rtol=1e-2)
    (loss + z_q.float().sum()).backward()
    dz_ref = 1.0 + 2.0 / (B * D) * (z.double() - W[gi].double())
    np.testing.assert_allclose(zc.grad.float().cpu().numpy(), dz_ref.numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("B,n_e,D", [(1, 512, 256), (63, 512, 256), (65, 100, 256), (20001, 512, 256), (70000, 256, 128),
                                     (4097, 128, 64), (33333, 512, 64), (129, 500, 128)])
def test_bf16_codebook_stationary_kernel_equals_row_stationary(B, n_e, D, monkeypatch):
    """The persistent codebook-in-registers bf16 kernel (D in {64,128,256}, padded codebook of 128/256/512 codes) against
    the row-stationary one it replaced (CEEG_VQ_BF16_BS=0): same MFMA products in the same k order, so codes and z_q are
    bitwise equal; the loss differs only in how the squared error is summed.  Shapes: a single row, ragged last tiles,
    padded codebooks (100 -> 128, 500 -> 512), more tiles than workgroups (70000 / 64 > 256: the persistent loop), and
    duplicate codes (ties must go to the lowest index in both)."""
    torch.manual_seed(B + n_e + D)
    W = torch.randn(n_e, D)
    W[n_e // 2] = W[3]  # an exact tie between codes 3 and n_e // 2
    z = torch.randn(B, D)
    z[: min(B, 7)] = W[3]  # rows that sit on the tied pair
    z = z.to(torch.bfloat16).cuda()
    vq = _mod(n_e, D, 0.25, W.numpy(), one_hot=False)
    monkeypatch.setenv("CEEG_VQ_BF16_BS", "1")
    l1, q1, p1, _, i1 = vq(z)
    monkeypatch.setenv("CEEG_VQ_BF16_BS", "0")
    l0, q0, p0, _, i0 = vq(z)
    assert torch.equal(i1, i0) and torch.equal(q1, q0) and torch.equal(p1, p0)
    assert (i1[: min(B, 7), 0] == 3).all()
    assert torch.equal(q1, vq.embedding.weight[i1[:, 0]].to(torch.bfloat16))
    # both kernels: loss = (1 + beta) mean((z_q - z)^2) for the bf16 z_q they return (rows sitting exactly on a code add 0)
    want = 1.25 * ((q1.double() - z.double()) ** 2).mean().item()
    np.testing.assert_allclose(l1.item(), want, rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(l0.item(), want, rtol=1e-4, atol=1e-12)


def test_large_properties_and_determinism():
    """Config-3-like size-independent properties: z_q rows ARE codebook rows, histogram sums to B,
    idempotence (quantising z_q returns the same codes), bitwise run-to-run reproducibility."""
    B, n_e, D = 1 << 18, 512, 256
    torch.manual_seed(0)
    W = torch.randn(n_e, D)
    vq = _mod(n_e, D, 0.25, W.numpy(), one_hot=False)
    for dt in (torch.float32, torch.bfloat16):
        z = torch.randn(B, D, device="cuda").to(dt).requires_grad_(True)
        l1, q1, p1, _, i1 = vq(z)
        l2, q2, p2, _, i2 = vq(z)
        assert torch.equal(i1, i2) and torch.equal(q1, q2) and torch.equal(l1, l2) and torch.equal(p1, p2)
        assert torch.equal(q1.detach(), vq.embedding.weight[i1[:, 0]].to(dt))
        assert int(torch.bincount(i1[:, 0], minlength=n_e).sum()) == B
        _, _, _, _, i3 = vq(q1.detach())
        if dt == torch.float32:
            assert torch.equal(i3, i1)
        (l1 + q1.float().sum()).backward()
        g1 = vq.embedding.weight.grad.clone(); vq.embedding.weight.grad = None
        dz1 = z.grad.clone(); z.grad = None
        l1b, q1b, _, _, _ = vq(z)
        (l1b + q1b.float().sum()).backward()
        assert torch.equal(g1, vq.embedding.weight.grad) and torch.equal(dz1, z.grad)
        vq.embedding.weight.grad = None
        # codebook gradient vs a dense fp64 torch restatement (torch on the GPU as the checker)
        onehot = torch.zeros(B, n_e, device="cuda").scatter_(1, i1, 1.0)
        cnt = onehot.sum(0)
        ref = 0.25 * 2.0 / (B * D) * (cnt.double()[:, None] * vq.embedding.weight.detach().double() - onehot.double().t() @ z.detach().double())
        np.testing.assert_allclose(g1.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("B,n_e,D,skew", [(1, 5, 8, False), (1000, 100, 32, False), (5000, 65, 16, True), (20000, 1000, 64, False),
                                          (17 * 1024 + 3, 130, 128, True), (40000, 64, 256, True), (3000, 4096, 32, False)])
def test_codebook_gradient_ragged_shapes_and_skew(B, n_e, D, skew):
    """The backward chain (slice histograms -> scan over 64-code workgroups x 16 slice groups -> stable scatter -> chunk sums ->
    codebook gradient) on shapes that leave its tiles ragged: fewer slices than slice groups, n_e not a multiple of 64, a last slice
    of 3 rows, unused codes, and (skew) nearly every row on ONE code so that a code owns far more than 16 chunks of 256 rows.
    dW and dz against the closed forms in fp64; two runs bit-identical."""
    rng = np.random.default_rng(B + n_e + D)
    W = rng.standard_normal((n_e, D)).astype(np.float32)
    zn = rng.standard_normal((B, D)).astype(np.float32)
    if skew:
        keep = rng.random(B) < 0.02
        zn = np.where(keep[:, None], zn, W[n_e // 3] + 1e-3 * zn).astype(np.float32)
    vq = _mod(n_e, D, 0.25, W, one_hot=False)
    grads = []
    for _ in range(2):
        z = torch.from_numpy(zn).cuda().requires_grad_(True)
        loss, z_q, _, _, idx = vq(z)
        loss.backward()
        grads.append((vq.embedding.weight.grad.clone(), z.grad.clone(), idx.clone()))
        vq.embedding.weight.grad = None
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    i64 = grads[0][2][:, 0].cpu()
    if skew:
        assert int((i64 == n_e // 3).sum()) > 0.9 * B
    z64, W64 = torch.from_numpy(zn).double(), torch.from_numpy(W).double()
    dW = torch.zeros_like(W64).index_add_(0, i64, W64[i64] - z64) * (0.25 * 2.0 / (B * D))
    dz = 2.0 / (B * D) * (z64 - W64[i64])
    np.testing.assert_allclose(grads[0][0].cpu().double().numpy(), dW.numpy(), rtol=1e-5, atol=2e-6 * float(dW.abs().max()) + 1e-12)
    np.testing.assert_allclose(grads[0][1].cpu().double().numpy(), dz.numpy(), rtol=1e-5, atol=1e-6 * float(dz.abs().max()) + 1e-12)


def test_largest_codebook_forward_backward_and_argument_checks():
    """n_e = 4096 (the advertised limit): forward bit-exact vs the C oracle, both gradients vs the closed forms in fp64; one
    more code is refused; a bad dW workspace is refused BEFORE anything is launched (dz stays untouched)."""
    from causal_eeg_vae_amd import _lib
    from oracle import c_oracle
    B, n_e, D = 700, 4096, 32
    rng = np.random.default_rng(1)
    W = rng.standard_normal((n_e, D)).astype(np.float32)
    zn = rng.standard_normal((B, D)).astype(np.float32)
    vq = _mod(n_e, D, 0.25, W, one_hot=False)
    z = torch.from_numpy(zn).cuda().requires_grad_(True)
    loss, z_q, ppl, _, idx = vq(z)
    o = c_oracle.vq_forward_f32(zn, W, 0.25)
    np.testing.assert_array_equal(idx[:, 0].cpu().numpy(), o["idx"])
    np.testing.assert_allclose(loss.item(), o["loss"], rtol=1e-5)
    r = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
    (loss + (z_q * r).sum()).backward()
    z64, W64, i64 = torch.from_numpy(zn).double(), torch.from_numpy(W).double(), torch.from_numpy(o["idx"])
    dz = r.cpu().double() + 2.0 / (B * D) * (z64 - W64[i64])
    dW = torch.zeros_like(W64).index_add_(0, i64, W64[i64] - z64) * (0.25 * 2.0 / (B * D))
    np.testing.assert_allclose(z.grad.cpu().double().numpy(), dz.numpy(), rtol=1e-5, atol=1e-6 * float(dz.abs().max()))
    np.testing.assert_allclose(vq.embedding.weight.grad.cpu().double().numpy(), dW.numpy(), rtol=1e-5, atol=1e-6 * float(dW.abs().max()))
    L = _lib.lib()
    p = z.data_ptr()
    assert L.ceeg_vq_forward(p, p, B, 4097, D, 0, 0.25, p, p, p, p, None, p, 1 << 30, _lib.stream()) == -2   # unsupported: no launch
    dzb = torch.full((B, D), 7.0, device="cuda")
    s = torch.ones(1, device="cuda")
    rc = L.ceeg_vq_backward(z.data_ptr(), vq.embedding.weight.data_ptr(), idx.data_ptr(), None, s.data_ptr(), B, n_e, D, 0, 0.25,
                            dzb.data_ptr(), vq.embedding.weight.grad.data_ptr(), None, 0, _lib.stream())
    torch.cuda.synchronize()
    assert rc == -3 and bool((dzb == 7.0).all())


def test_errors_are_loud():
    from causal_eeg_vae_amd import VectorQuantizer, CeegError
    vq = VectorQuantizer(16, 8, 0.25)
    with pytest.raises(CeegError):
        vq(torch.randn(4, 8))  # CPU tensor: no fallback
    vq = VectorQuantizer(16, 300, 0.25).cuda()
    with pytest.raises(CeegError):
        vq(torch.randn(4, 300, device="cuda"))  # e_dim > 256 unsupported
    with pytest.raises(ValueError):
        VectorQuantizer(16, 8, 0.25).cuda()(torch.randn(4, 9, device="cuda"))


def _fwd_f32_abi(z, W, beta=0.25):
    """ceeg_vq_forward (fp32) through the C-ABI; returns (idx, z_q, loss, screened) with screened = 1 / 0 / -1 as
    ceeg_vq_forward_screened reports it."""
    import ctypes
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    B, D = z.shape
    n_e = W.shape[0]
    zq = torch.empty_like(z)
    idx = torch.empty(B, dtype=torch.int64, device="cuda")
    sc = torch.empty(2, device="cuda")
    ws = _lib.workspace(L.ceeg_vq_forward_ws_bytes(B, n_e, D, _lib.CEEG_F32), z.device)
    _lib.check(L.ceeg_vq_forward(z.data_ptr(), W.data_ptr(), B, n_e, D, _lib.CEEG_F32, beta, zq.data_ptr(), idx.data_ptr(), sc[0:].data_ptr(),
                                 sc[1:].data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_vq_forward")
    how = ctypes.c_int(-7)
    _lib.check(L.ceeg_vq_forward_screened(ws.data_ptr(), ws.numel(), B, n_e, D, _lib.CEEG_F32, ctypes.addressof(how), _lib.stream()),
               "ceeg_vq_forward_screened")
    return idx, zq, float(sc[0]), how.value


def _screen_case(kind, B, n_e, D, rng):
    z = rng.standard_normal((B, D)).astype(np.float32)
    W = rng.standard_normal((n_e, D)).astype(np.float32)
    if kind == "small_codebook":
        W *= 0.1
    elif kind == "ref_init":                      # the reference's U(+-1/n_e) initialisation
        W = rng.uniform(-1.0 / n_e, 1.0 / n_e, (n_e, D)).astype(np.float32)
    elif kind == "all_near":                      # one code plus noise in the last bits: every code is a candidate of every row
        W = (W[:1] * (1 + 2e-7 * rng.standard_normal((n_e, D)))).astype(np.float32)
    elif kind == "duplicates":                    # exact ties between code pairs: the lower index must win
        W[n_e // 2:] = W[: n_e - n_e // 2]
    elif kind == "near_duplicates":               # ties broken in the last bits of the fp32 chain
        W[1::2] = W[0::2][: W[1::2].shape[0]] * (1 + 3e-7 * rng.standard_normal((W[1::2].shape[0], 1)).astype(np.float32))
    elif kind == "rows_on_codes":                 # z almost equal to a code: distances cancel to ~0
        z = (W[rng.integers(0, n_e, B)] * (1 + 1e-6 * rng.standard_normal((B, D)))).astype(np.float32)
    elif kind == "wide_range":                    # a few huge components per row and per code
        z[:, ::17] *= 3e3
        W[:, 5::29] *= 1e3
    elif kind == "tiny":
        z *= 1e-19
        W *= 1e-19
    elif kind == "huge":
        z *= 1e15
        W *= 1e3
    elif kind == "mixed_rows":                    # row scales from 1e-6 to 1e6 in one batch
        z *= (10.0 ** rng.uniform(-6, 6, (B, 1))).astype(np.float32)
    elif kind == "zero_rows":
        z[::3] = 0
        W[7] = 0
    return z, W


@pytest.mark.parametrize("kind", ["normal", "small_codebook", "ref_init", "all_near", "duplicates", "near_duplicates", "rows_on_codes", "wide_range", "tiny",
                                  "huge", "mixed_rows", "zero_rows"])
@pytest.mark.parametrize("B,n_e,D", [(3001, 512, 128), (4100, 100, 32), (1500, 256, 256), (33, 64, 64), (1, 512, 128)])
def test_fp32_screened_forward_bit_exact(kind, B, n_e, D, monkeypatch):
    """The screened fp32 forward (bf16 MFMA scores + error bound -> candidates -> the contract's fmaf chains; csrc/vq.hip
    vq_screen_f32) against the C oracle and against the full kernel (CEEG_VQ_SCREEN=0): codes and z_q bit for bit on spread-out
    data, exact and near ties, rows sitting on codes, extreme and mixed scales - whichever of the two forms ends up deciding."""
    from oracle import c_oracle
    rng = np.random.default_rng(sum(map(ord, kind)) * 1000003 + B * 131 + n_e * 7 + D)
    zn, Wn = _screen_case(kind, B, n_e, D, rng)
    z, W = torch.from_numpy(zn).cuda(), torch.from_numpy(Wn).cuda()
    idx, zq, loss, how = _fwd_f32_abi(z, W)
    assert how in (0, 1)
    if kind in ("normal", "small_codebook", "ref_init") and n_e * D >= 64 * 64:
        assert how == 1                            # spread-out data must not need the full kernel
    if kind == "all_near" and n_e >= 256 and B >= 33:
        assert how == 0                            # more candidates than the list holds: handed over on the device
    o = c_oracle.vq_forward_f32(zn, Wn, 0.25)
    np.testing.assert_array_equal(idx.cpu().numpy(), o["idx"])
    np.testing.assert_array_equal(zq.cpu().numpy(), o["z_q"])
    monkeypatch.setenv("CEEG_VQ_SCREEN", "0")
    idx0, zq0, loss0, how0 = _fwd_f32_abi(z, W)
    assert how0 == -1 and torch.equal(idx, idx0) and torch.equal(zq, zq0)
    np.testing.assert_allclose(loss, loss0, rtol=1e-5, atol=1e-30)


def test_fp32_screened_forward_non_finite_rows(monkeypatch):
    """NaN / inf in z: the screening kernel raises its flag and the full kernel's rules decide; a NaN code is never a candidate, as
    it never wins the full kernel's strict comparison - outputs equal the unscreened call's on every row."""
    rng = np.random.default_rng(11)
    B, n_e, D = 2000, 512, 128
    z = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
    W = torch.from_numpy(rng.standard_normal((n_e, D)).astype(np.float32)).cuda()
    for case in ("nan_row", "inf_row", "nan_code"):
        zz, WW = z.clone(), W.clone()
        if case == "nan_row":
            zz[5, 3] = float("nan"); zz[1999] = float("nan")
        elif case == "inf_row":
            zz[17, 100] = float("inf"); zz[18, 0] = -float("inf")
        else:
            WW[44, 7] = float("nan")
        monkeypatch.setenv("CEEG_VQ_SCREEN", "1")
        idx, zq, _, how = _fwd_f32_abi(zz, WW)
        assert how == (1 if case == "nan_code" else 0)   # a NaN code never wins in either form; a NaN / inf row has no finite maximum
        monkeypatch.setenv("CEEG_VQ_SCREEN", "0")
        idx0, zq0, _, _ = _fwd_f32_abi(zz, WW)
        assert torch.equal(idx, idx0) and torch.equal(zq.view(torch.int32), zq0.view(torch.int32))
