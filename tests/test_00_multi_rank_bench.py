"""The multi-GPU leg of bench.py as a driver would start it -- `python bench.py --gpus 2` with no launcher -- rehearsed on
the one GPU of the test box: two ranks share it and every exchange goes through the host (--backend gloo).  Rank 0
re-renders one pose that the other rank delivered and compares the bytes (--verify).

This file sorts first on purpose: the ranks are grandchildren of this process, and they are started before this
process has made any GPU call of its own."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1100)
def test_bench_starts_its_own_ranks_and_the_gathered_frames_are_the_rendered_ones():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--verify", "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1000)
    assert r.returncode == 0, f"bench.py --gpus 2 failed (rc {r.returncode}):\n{r.stderr[-3000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected one JSON line on stdout, got: {r.stdout[-1000:]}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["verified_frames_from_other_ranks"] == 1
    assert out["config"]["backend"] == "gloo" and "REHEARSAL" in out["config"]["parallelism"]
    # every N > 1 line carries its own one-GPU reference of the SAME workload
    assert out["one_gpu_same_workload"]["ms_per_step"] > 0 and out["speedup"] > 0
    # both passes of the 4K configuration are sharded
    b = out["secondary"]["bands_4k_with_depth_pass_2gpu"]
    assert "rm_shadowmap_directional_bands_device" in b["what"] and b["speedup"] > 0
    assert out["value"] > 0 and out["config"]["poses"] == 64
