"""The driver's contract with bench.py, executed on the GPU: `python bench.py --gpus 1 --steps K --warmup W` with the
driver's small K must be a steady-state measurement, and `--gpus 2` must start two ranks by
itself (rehearsed on one card with the gloo backend: C4_BENCH_BACKEND=gloo C4_BENCH_DEVICE=0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line on stdout
    return json.loads(lines[0])


def test_driver_invocation_measures_steady_state():
    d = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2", "--precise-compare", "0"])
    assert d["metric"] == "mcts_node_expansions_per_sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["timed_region_s"] >= 0.1                                    # not a sub-millisecond transient
    assert abs(d["ms_per_step"] * d["steps"] / 1000.0 - d["timed_region_s"]) < 1e-6
    assert d["preroll_games"] >= 4 * 4096 and d["preroll_s"] > 0           # declared, untimed pre-roll
    assert d["games_per_sec"] > 5000 and d["moves_per_sec"] > 0            # searches complete inside the timed region
    assert 0.7 < d["eval_cache_hit_rate"] < 0.9                            # warm table, not 4096 identical openings
    assert d["value"] > 1.0e8 and d["bad_evals"] == 0
    g = d["games_exported_on_device"]
    assert g["exported"] == g["finished"] and g["dropped"] == 0 and g["positions"] > 20 * g["exported"] / 2
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["quanta_per_launch"] == 256 and r["launches"] == 20
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["avg_launch_ms"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"]   # events agree with the wall clock
    if r["traffic"] is not None:                                           # only from a PMC record of this launch shape
        assert r["traffic"] <= 1.1 * r["algorithmic_bytes_per_launch"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


def test_gpus_2_launches_two_ranks_on_the_card():
    d = run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--precise-compare", "0",
                   "--profile-steps", "0", "--preroll-games-per-slot", "1"],
                  env_extra={"C4_BENCH_BACKEND": "gloo", "C4_BENCH_DEVICE": "0"})
    assert d["n_gpus"] == 2 and d["world_size_observed"] == 2 and d["collective_backend"] == "gloo"
    assert d["scaling"] == "weak" and "8192" not in d["config"]["workload"]      # 4096 games PER GPU
    assert d["value"] > 5.0e7 and d["games_per_sec"] > 0
