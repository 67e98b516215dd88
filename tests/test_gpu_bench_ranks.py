"""The N > 1 path of bench.py kept alive on a one-GPU box: `RT_BENCH_REHEARSE=1 python bench.py --gpus 2 ...` starts its own two
ranks (a child `python -m torch.distributed.run`), both on GPU 0 over gloo -- RCCL does not accept two ranks on one device -- and
goes through scene broadcast, band rendering, the pipelined gather and the reductions; the frame rank 0 ends up with must be the
frame a single rank renders.  Both runs are child processes of the test (nothing here replaces a process that has touched the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "3", "--warmup", "1", "--width", "640", "--height", "360", "--max-recursion-limit", "3", "--no-cpu-baseline", "--no-north-star"]


def run_bench(extra, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("pipeline", ["auto", "wavefront"])
def test_two_rank_rehearsal_gathers_the_one_rank_frame(pipeline):
    one = run_bench(["--gpus", "1", "--pipeline", pipeline], {})
    two = run_bench(["--gpus", "2", "--pipeline", pipeline], {"RT_BENCH_REHEARSE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["ranks"]["world_size"] == 2 and "gloo" in two["ranks"]["backend"]
    assert two["scaling"] == "strong" and "REHEARSAL" in two["data"]
    assert one["frame_crc32"] is not None and one["frame_crc32"] == two["frame_crc32"]
    assert one["rays_per_frame"] == two["rays_per_frame"]
    for rep in (one, two):
        assert rep["roofline"]["frac"] > 0 and rep["value"] > 0 and len(json.dumps(rep)) < 6000
