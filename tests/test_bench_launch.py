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


def test_dry_run_line_fits_the_drivers_tail():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    assert len(last) < 4096
    json.loads(last)


def test_compact_line_from_a_full_measurement_stays_under_4k():
    """The line the driver parses is built from everything `measure` collects; round 2's full record was 21.6 KB on one line and
    the driver's ~8 KB stdout tail cut its head off.  The canned record is that very line (profiles/r02_bench_default.json)."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))
    assert len(json.dumps(full)) > 8192                                  # the canned record is the oversized one
    full["roofline"]["peak_measured"] = 54.321; full["roofline"]["frac_of_measured"] = 0.04
    full["roofline"]["traffic_source"] = "profiles/r03_pmc_h2o_ccpvtz.json (committed rocprofv3 --pmc passes, not this run)"
    full["rccl"] = "x" * 500
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.MAX_LINE_BYTES == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "iter_breakdown_ms", "roofline", "cpu_baseline", "detail"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "kernel_alg_bytes",
              "kernel_alg_flops", "other_roof", "peak_measured", "frac_of_measured"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "one_thread_value", "scf_iter_ms"):
        assert k in line["cpu_baseline"], k
    assert line["config"]["workload"].startswith("H2O/cc-pVTZ")
    assert len(line["scaling_reference"]) <= 12
    # nothing bulky leaks through
    for k in ("units_alone", "traffic_detail", "eigensolve_counters", "accumulation", "committed_counters"):
        assert k not in text
    # a pathological record still yields a line that fits: optional blocks are shed
    full["scaling_reference"]["workload"] = "y" * 5000
    assert len(json.dumps(bench.compact_line(full, "bench_detail.json"), separators=(",", ":"))) < 4096
