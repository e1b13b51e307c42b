"""GPU: ceeg_vq_forward_static (the entry point that allocates nothing) against ceeg_vq_forward, through the C-ABI.

The static entry point keeps the forward's scratch in a fixed device array inside libceeg.so instead of a caller
workspace.  Two calls in a row, the second with a SMALLER input (stale scratch of the first call must not leak into it),
every output bit for bit equal to the workspace entry point's, in fp32 and bf16; sizes beyond the array are refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(static, z, W, beta):
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    B, D = z.shape
    n_e = W.shape[0]
    dt = _lib.CEEG_BF16 if z.dtype == torch.bfloat16 else _lib.CEEG_F32
    z_q = torch.empty_like(z)
    idx = torch.empty(B, dtype=torch.int64, device=z.device)
    sc = torch.empty(2, dtype=torch.float32, device=z.device)
    hist = torch.empty(n_e, dtype=torch.int32, device=z.device)
    if static:
        _lib.check(L.ceeg_vq_forward_static_prepare(), "prepare")   # per-device set-up (the symbol lookup), outside the measured call
        torch.cuda.synchronize()
        free0, alloc0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
        rc = L.ceeg_vq_forward_static(z.data_ptr(), W.data_ptr(), B, n_e, D, dt, beta, z_q.data_ptr(), idx.data_ptr(),
                                      sc[0:].data_ptr(), sc[1:].data_ptr(), hist.data_ptr(), _lib.stream())
    else:
        ws = _lib.workspace(L.ceeg_vq_forward_ws_bytes(B, n_e, D, dt), z.device)
        rc = L.ceeg_vq_forward(z.data_ptr(), W.data_ptr(), B, n_e, D, dt, beta, z_q.data_ptr(), idx.data_ptr(),
                               sc[0:].data_ptr(), sc[1:].data_ptr(), hist.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream())
    _lib.check(rc, "vq forward")
    torch.cuda.synchronize()
    if static:   # the entry point allocated nothing: neither through torch nor behind its back (driver-level free memory)
        assert torch.cuda.memory_allocated() == alloc0 and torch.cuda.mem_get_info()[0] == free0
    return z_q, idx, sc, hist


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_static_entry_point_equals_workspace_entry_point_twice(dtype):
    from oracle import c_oracle
    torch.manual_seed(11)
    n_e, D = 512, 128
    W = torch.randn(n_e, D, device="cuda")
    for B in (70001, 3000):  # second call smaller: reads only what it wrote itself
        z = torch.randn(B, D, device="cuda").to(dtype)
        want = _call(False, z, W, 0.25)
        got = _call(True, z, W, 0.25)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert int(got[3].sum()) == B
        if dtype == torch.float32:  # and the oracle itself, bit-exact indices
            o = c_oracle.vq_forward_f32(z.cpu().numpy(), W.cpu().numpy(), 0.25)
            np.testing.assert_array_equal(got[1].cpu().numpy(), o["idx"])
            np.testing.assert_allclose(got[2][0].item(), o["loss"], rtol=1e-5)


def test_static_entry_point_refuses_what_does_not_fit():
    from causal_eeg_vae_amd import _lib
    L = _lib.lib()
    cap = L.ceeg_vq_forward_static_max_bytes()
    assert cap == 4 << 20
    assert L.ceeg_vq_forward_ws_bytes(1 << 22, 512, 256, _lib.CEEG_BF16) <= cap  # config 3 fits
    B, n_e, D = 1 << 22, 4096, 256
    assert L.ceeg_vq_forward_ws_bytes(B, n_e, D, _lib.CEEG_F32) > cap
    d = torch.empty(16, device="cuda")  # never dereferenced: the size check comes before any launch
    p = d.data_ptr()
    rc = L.ceeg_vq_forward_static(p, p, B, n_e, D, _lib.CEEG_F32, 0.25, p, p, p, p, None, _lib.stream())
    assert rc == -6 and b"scratch" in L.ceeg_strerror(rc)
