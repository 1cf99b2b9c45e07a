"""bench.py's N > 1 line assembly without a GPU: world 2 over gloo, one process per rank started by bench.py's own launcher
(path_tracer_amd/launch.py), a stub renderer on the CPU (tests/bench_stub.py) in place of capi.Renderer.  Everything else is
the product's: the process group set-up, distributed.pass_range / reduce_and_finish, the reduce timing, the max-over-ranks
time, and rank 0 printing ONE JSON line -- so that the first real 8-GPU run cannot die on Python."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(world, steps=3, warmup=1, extra_env=None, args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PT_BENCH_STUB="bench_stub", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""),
               GLOO_SOCKET_IFNAME="lo", PT_LAUNCH_TIMEOUT="120")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup), *args],
                       env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.parametrize("world", [2, 3])
def test_split_line_is_assembled_by_rank_0_only(world):
    r, lines = _run(world)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["vs_baseline"] is None and "STUB" in d["data"]
    cfg = d["config"]
    assert cfg["spp_per_gpu"] == 64 and cfg["spp_total"] == 64 * world and cfg["parallelism"].startswith("spp-split x%d" % world)
    # whole-job throughput: all ranks' samples over the (max-over-ranks) time of the timed steps
    assert d["value"] == pytest.approx(1920 * 1080 * 64 * world * 3 / (d["ms_per_step"] * 3e-3), rel=1e-6)
    # the reduce is timed inside the line
    assert cfg["reduce_ms"] > 0 and 0 < cfg["reduce_fraction_of_step"] <= 1.0
    # rank 0's counters cover ITS passes only
    assert d["roofline"]["work_per_launch"]["samples"] == 1920 * 1080 * 64
    assert "cpu_baseline" not in d and "extra" not in d          # single-GPU legs stay out of a split run


def test_a_rank_that_fails_ends_the_job_with_its_code():
    """the sample-count check of reduce_and_finish (warm-up step) fires when a rank contributes the wrong number of passes: the
    launcher must end every rank and return non-zero, not hang"""
    r, lines = _run(2, extra_env={"BENCH_STUB_BREAK_RANK": "1"})
    assert r.returncode != 0 and not lines
    assert "sample count after reduce" in r.stderr
