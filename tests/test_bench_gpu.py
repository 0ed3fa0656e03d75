"""bench.py end to end on the GPU box at a miniature configuration: the single-rank line with its roofline / cpu_baseline
objects, and the driver's N > 1 form (the script starts its own ranks) with two ranks sharing the one GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SMALL = ["--seq", "48", "--batch", "3", "--layers", "2", "--d_model", "128", "--steps", "1", "--warmup", "1"]


def run_bench(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_single_rank_line_has_roofline_and_cpu_baseline():
    rec = run_bench([])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["unit"] == "chord-tokens/s" and rec["dtype"] == "f32"
    r = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert 0 < r["frac"] < 1 and r["bound"] == "hbm" and r["traffic"] is None        # the committed PMC pass is for config 2's shape only
    assert 0 < r["prefill"]["frac"] < 1 and r["prefill"]["bound"] == "mfma"
    assert r["whole_step"]["frac"] > 0 and r["decode_gemm"]["avg_launch_us"] > 0
    c = rec["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["value_8_threads"] > 0


def test_bench_starts_two_ranks_on_one_gpu():
    rec = run_bench(["--gpus", "2", "--no_cpu_baseline", "--no_roofline"], {"AMT_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "2"})
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["global_batch"] == 6 and rec["scaling"] == "weak"
