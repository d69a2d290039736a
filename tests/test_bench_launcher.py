"""CPU: `python bench.py --gpus N` with no torchrun environment starts N ranks itself and prints ONE JSON line
(round-2 review, item 2: without this an 8-GPU node would yield eight N = 1 numbers), and the torchrun form of the
driver contract keeps working.  `--dry-run` runs everything of an N-rank run except the device work: launcher, process
group (gloo here), the nnz-balanced row cut, one collective, rank 0's line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _one_line(out):
    lines = [x for x in out.splitlines() if x.strip()]
    assert len(lines) == 1, lines                       # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_launches_its_own_ranks(n, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--steps", "3",
                        "--warmup", "1"], cwd=str(tmp_path), env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == n and line["ranks_in_process_group"] == n
    assert len(line["extra"]["nnz_per_rank"]) == n and all(x > 0 for x in line["extra"]["nnz_per_rank"])
    assert line["steps"] == 3 and line["warmup"] == 1


def test_bench_under_torchrun_form(tmp_path):
    """the driver's own launch line (torch.distributed.run sets WORLD_SIZE): no second launcher level"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--dry-run"], cwd=str(tmp_path), env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 2 and len(line["extra"]["nnz_per_rank"]) == 2
    assert "launching" not in r.stderr


def test_bench_single_gpu_dry_run_needs_no_launcher(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], cwd=str(tmp_path), env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 1 and "launching" not in r.stderr
