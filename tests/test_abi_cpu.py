"""CPU: the C-ABI library loads and exports every symbol include/ceeg.h declares; host-side helpers."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ceeg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ceeg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from causal_eeg_vae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ceeg.h but not exported by libceeg.so"
    assert set(names) == set(_lib.SIGNATURES), "ctypes SIGNATURES and include/ceeg.h disagree"
    lib.ceeg_version.restype = ctypes.c_int
    assert lib.ceeg_version() == 100
    lib.ceeg_strerror.restype = ctypes.c_char_p
    assert b"workspace" in lib.ceeg_strerror(-3)


def test_library_is_built_from_the_sources_in_the_tree():
    """The hash the Makefile bakes into libceeg.so equals the hash of csrc/ + include/ceeg.h in the tree: a stale binary
    (what the GPU box would run, since it does not build) fails here and in smoke()."""
    from causal_eeg_vae_amd import _lib, _srchash
    assert _lib.check_build_matches_sources() == _srchash.source_hash()
    assert len(_srchash.source_files()) >= 12


def test_no_cpu_fallback_in_product_path():
    from causal_eeg_vae_amd import VectorQuantizer, CeegError
    with pytest.raises(CeegError):
        VectorQuantizer(8, 4, 0.25)(torch.randn(3, 4))
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "causal_eeg_vae_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "import_module(\"oracle" not in src and "__import__(\"oracle" not in src


def test_vector_quantizer_constructor_matches_reference_contract():
    from causal_eeg_vae_amd import VectorQuantizer
    vq = VectorQuantizer(512, 8, 0.25)  # scripts/run.py:296-298
    assert list(vq.state_dict()) == ["embedding.weight"]
    w = vq.embedding.weight
    assert w.shape == (512, 8) and float(w.abs().max()) <= 1.0 / 512 + 1e-9  # quantizer.py:23
    assert (vq.n_e, vq.e_dim, vq.beta) == (512, 8, 0.25)


def test_sigma_params_are_bijections():
    from causal_eeg_vae_amd import sigma_params
    for N in (2, 3, 10, 97, 1000, 1 << 20, 999_983):
        for step in range(4):
            a, b, ainv = sigma_params(N, 7, step)
            assert 1 <= a < N + (N == 2) and 0 <= b < N and math.gcd(a, N) == 1 and (a * ainv) % N == 1 % N
    a, b, _ = sigma_params(1000, 0, 0)
    assert len({(a * x + b) % 1000 for x in range(1000)}) == 1000
    assert sigma_params(1000, 0, 0) != sigma_params(1000, 0, 1)  # fresh negatives every step


def test_synthetic_event_graph_shape_and_determinism():
    from causal_eeg_vae_amd.synth import synth_event_graph, collate_graphs
    ei = synth_event_graph(20000, 320000, seed=3)
    assert ei.shape == (2, 320000) and bool((ei[0] < ei[1]).all())            # DAG, time-ordered, no self-loops
    assert torch.unique(ei[1] * 20000 + ei[0]).numel() == 320000              # unique edges
    assert torch.equal(ei, synth_event_graph(20000, 320000, seed=3))
    deg = torch.bincount(ei[0], minlength=20000)
    assert deg.max() > 50 * deg.float().median()                              # heavy-tailed out-degree (hubs)
    e2, batch, n = collate_graphs([ei[:, :100], ei[:, 100:150]], [20000, 20000])
    assert n == 40000 and e2.shape[1] == 150 and int(e2[:, 100:].min()) >= 20000 and batch.shape[0] == 40000


def test_csr_oracle_drops_and_keeps_self_loops():
    from oracle import ref_torch
    src = np.array([2, 2, 0, 1, 1]); dst = np.array([2, 2, 1, 0, 1])
    rp, col, perm = ref_torch.csr_build_np(src, dst, 3, True)
    assert rp.tolist() == [0, 2, 4, 5] and col.tolist() == [0, 1, 0, 1, 2] and perm.tolist() == [-1, 3, 2, 4, 1]
    rp, col, perm = ref_torch.csr_build_np(src, dst, 3, False)
    assert rp.tolist() == [0, 1, 3, 5] and col.tolist() == [1, 0, 1, 2, 2] and perm.tolist() == [3, 2, 4, 0, 1]


def test_tuned_gemm_table_is_well_formed_and_inert_without_a_gpu():
    """causal_eeg_vae_amd/tuned_gemm_gfx950.csv (a TunableOp result file: validators, then one library kernel per GEMM shape
    of BASELINE's configs) parses, names gfx950, and _lib.enable_tuned_gemms() leaves PyTorch alone when there is no GPU."""
    import os
    import torch
    from causal_eeg_vae_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "tuned_gemm_gfx950.csv")
    rows = [line.strip().split(",") for line in open(path) if line.strip()]
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert validators["GCN_ARCH_NAME"].startswith("gfx950") and "PT_VERSION" in validators and "HIPBLASLT_VERSION" in validators
    entries = [r for r in rows if r[0] != "Validator"]
    assert len(entries) >= 8 and all(len(r) == 4 and r[0].startswith("Gemm") and float(r[3]) > 0 for r in entries)
    assert len({(r[0], r[1]) for r in entries}) == len(entries)          # one kernel per (operation, shape)
    if not torch.cuda.is_available():
        assert _lib.enable_tuned_gemms() is False   # (torch.cuda.tunable itself cannot even be queried without a device)
