"""`python bench.py --gpus N` must start N ranks itself: the launcher path, the
rendezvous and the timing contract's MAX/SUM reductions, run here with 2 gloo ranks on CPU
(`--dry-run`: no engine, no GPU; the printed line is flagged dry_run and carries no measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_without_a_wrapper():
    out = _run(["--gpus", "2", "--dry-run"])
    assert out["n_gpus"] == 2 and out["world_size_observed"] == 2 and out["collective_backend"] == "gloo"
    assert out["max_check"] == 2.0      # MAX over ranks of (1 + rank)
    assert out["sum_check"] == 3.0      # SUM over ranks of (rank + 1)
    assert out["dry_run"] is True and out["value"] is None


def test_single_rank_needs_no_process_group():
    out = _run(["--gpus", "1", "--dry-run"])
    assert out["n_gpus"] == 1 and out["world_size_observed"] == 1 and out["collective_backend"] is None


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
