"""`python bench.py --gpus N` without torchrun must start its N ranks itself: the self-launch path
is exercised here on CPU with --launch-check (ranks form a gloo group and all-reduce their rank; no GPU is touched), for the
bare form the driver uses and for the torch.distributed.run form."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "1"
    return env


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", [2, 4])
def test_bench_starts_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-check"], env=_clean_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]      # (gloo itself chats on stdout; RCCL does not)
    assert len(lines) == 1, r.stdout                      # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["launch_check"] and out["n_gpus"] == n and out["rank_sum"] == n * (n + 1) / 2
    assert out["master"].startswith("127.0.0.1:")


@pytest.mark.timeout(600)
def test_bench_under_torch_distributed_run():
    from radnet_hip.launch import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), BENCH, "--gpus", "2", "--launch-check"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["rank_sum"] == 3.0


@pytest.mark.timeout(300)
def test_a_failing_rank_ends_the_job_with_its_code(tmp_path):
    from radnet_hip.launch import spawn_ranks
    script = tmp_path / "r.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    assert spawn_ranks(3, [sys.executable, str(script)]) == 7      # the sleeping ranks are terminated, not waited for


def test_world_size_mismatch_is_refused():
    env = _clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
