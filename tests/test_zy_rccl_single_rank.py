"""The `nccl` (= RCCL) branch of the data-parallel exchange on real hardware, as far as one GPU allows: a ONE-rank RCCL group
(two RCCL ranks cannot share a device) runs trainer.broadcast_params_, the flat gradient all-reduce queued right behind a real
backward, a MAX all-reduce and a barrier on device tensors (scripts/rccl_single_rank.py, a child process so that the process
group never leaks into the other tests).  Collected late on purpose: a box whose RCCL cannot come up skips, it does not fail."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_exchange():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_single_rank.py")], capture_output=True, text=True, timeout=150, env=env)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL did not come up within 150 s on this box")
    lines = [t for t in r.stdout.splitlines() if t.startswith("{")]
    if not lines:
        if "NCCL" in r.stderr or "RCCL" in r.stderr or "ProcessGroup" in r.stderr:
            pytest.skip("RCCL initialisation failed on this box: " + r.stderr[-300:])
        raise AssertionError(r.stderr[-2000:])
    out = json.loads(lines[-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["broadcast_unchanged"] and out["allreduce_sum_div2_exact"] and out["max_reduce"] == 0.125
    assert r.returncode == 0
