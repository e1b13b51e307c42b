"""CPU tests of host-side logic that the GPU entry points rely on: hub-row piece lists, the per-step node permutation."""
import math

import numpy as np
import pytest
import torch


@pytest.mark.parametrize("seed,chunk", [(0, 8), (1, 8), (2, 5), (3, 256)])
def test_long_items_cover_every_long_row_once(seed, chunk):
    """build_long_items (include/ceeg.h: ceeg_decoder_codes_*, ceeg_spmm_csr_items) against a plain Python walk: pieces tile
    exactly the long rows, are cut at global multiples of the chunk, and no two pieces share a (chunk, slot) table slot."""
    from causal_eeg_vae_amd.models import build_long_items
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 3 * chunk, 200)
    lens[rng.integers(0, 200, 10)] = rng.integers(chunk + 1, 12 * chunk, 10)
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    items = build_long_items(torch.from_numpy(rowptr.astype(np.int32)), chunk)
    want = []
    for r in range(200):
        lo, hi = int(rowptr[r]), int(rowptr[r + 1])
        if hi - lo <= chunk:
            continue
        for c in range(lo // chunk, (hi - 1) // chunk + 1):
            plo, phi = max(lo, c * chunk), min(hi, (c + 1) * chunk)
            slot = 1 if (c == lo // chunk and lo != c * chunk) else 0
            want.append((plo, phi, 2 * c + slot, r))
    if not want:
        assert items is None
        return
    got = [tuple(int(v) for v in row) for row in items]
    assert got == want
    slots = [g[2] for g in got]
    assert len(set(slots)) == len(slots)
    assert all(0 < g[1] - g[0] <= chunk for g in got)


def test_long_items_none_without_long_rows():
    from causal_eeg_vae_amd.models import build_long_items
    assert build_long_items(torch.tensor([0, 3, 3, 10], dtype=torch.int32), 256) is None


@pytest.mark.parametrize("N", [1, 2, 5, 96, 1000, 6250, 1_000_000])
def test_sigma_params_is_a_permutation_with_its_inverse(N):
    from causal_eeg_vae_amd import sigma_params
    for step in range(3):
        a, b, ainv = sigma_params(N, 7, step)
        assert 0 < a and 0 <= b < max(N, 1) and math.gcd(a, N) == 1 and (a * ainv) % N == 1 % N
        if N <= 6250:
            x = np.arange(N)
            y = (a * x + b) % N
            assert len(np.unique(y)) == N
            assert np.array_equal((ainv * ((y - b) % N)) % N, x)      # the inverse the decoder's backward uses
    assert sigma_params(N, 7, 0) == sigma_params(N, 7, 0)
    if N > 5:
        assert sigma_params(N, 7, 0) != sigma_params(N, 7, 1)
