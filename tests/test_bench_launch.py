"""bench.py's launch contract on the CPU: `python bench.py --gpus N` started plainly must start its own one-process-per-rank
launcher (before any GPU call), rendezvous, time, and relay exactly one JSON line from rank 0.  `--dry-run` swaps the GPU step
for the product's host-side shard planner; everything else is the path the GPU run takes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_prints_one_line():
    d = _run(["--gpus", "2"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True
    assert d["quartets_over_all_ranks"] == d["quartets_expected"] == 120          # the two shards cover H2O/STO-3G's quartets


def test_single_rank_runs_in_process():
    d = _run([])
    assert d["n_gpus"] == 1 and d["quartets_over_all_ranks"] == d["quartets_expected"]
