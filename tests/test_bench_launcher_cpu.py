"""bench.py's own launcher branch (`python bench.py --gpus N` with no torch.distributed.run around it), driven on the CPU
with a stub in place of the per-rank program: N fresh child ranks, rank 0's JSON line relayed as the last stdout line, the
children's failure = a non-zero exit code."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


STUB = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print(f"noise from rank {rank}")
    if "--fail" in sys.argv and rank == 1:
        sys.exit(7)
    if rank == 0 and "--silent" not in sys.argv:
        print(json.dumps({"metric": "stub", "value": float(t.item()), "n_gpus": world, "ranks": dist.get_world_size(),
                          "backend": dist.get_backend(), "argv": sys.argv[1:]}))
    dist.destroy_process_group()
""")


def _run(tmp_path, capsys, argv):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    rc = _bench().launch_ranks(2, argv, script=str(stub), timeout=240)
    out = capsys.readouterr()
    return rc, out.out, out.err


def test_launcher_relays_rank0_line_and_exit_code(tmp_path, capsys):
    rc, out, err = _run(tmp_path, capsys, ["--gpus", "2", "--steps", "3"])
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1                                   # exactly the result line on stdout
    rec = json.loads(lines[-1])
    assert rec["ranks"] == 2 and rec["n_gpus"] == 2 and rec["backend"] == "gloo" and rec["value"] == 3.0
    assert rec["argv"] == ["--gpus", "2", "--steps", "3"]     # the ranks got the same arguments
    assert "noise from rank" in err                          # everything else went to stderr


def test_launcher_failed_rank_is_nonzero(tmp_path, capsys):
    rc, out, _ = _run(tmp_path, capsys, ["--fail"])
    assert rc != 0 and out.strip() == ""


def test_launcher_no_result_line_is_nonzero(tmp_path, capsys):
    rc, out, _ = _run(tmp_path, capsys, ["--silent"])
    assert rc == 4 and out.strip() == ""


def test_gpus_gt_visible_devices_exits_2_without_launching():
    """`python bench.py --gpus 2` on a box with fewer GPUs (this container has none) refuses with exit code 2 and says how
    to rehearse; with WORLD_SIZE set to something else it refuses too - neither touches a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CEEG_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=240)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "CEEG_SHARE_GPU" in r.stderr
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
