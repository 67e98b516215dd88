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
COMMON = ["--steps", "3", "--warmup", "1", "--width", "640", "--height", "352", "--max-recursion-limit", "3", "--no-cpu-baseline", "--no-north-star"]


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


@pytest.mark.parametrize("dist_backend", ["torch", "cabi"])
def test_single_rank_through_the_rccl_code_path(dist_backend):
    """RT_BENCH_FORCE_DIST=1 under torch.distributed.run with one rank: the process group, the scene broadcast and the gather of the
    accumulated frame run over RCCL (torch.distributed's, or the library's own calls of include/rt_dist.h), and the frame that comes
    out of the gather is the frame of the plain single-process run."""
    import socket
    plain = run_bench(["--gpus", "1"], {})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update({"RT_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py")] + COMMON + ["--gpus", "1", "--dist", dist_backend]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rep = json.loads(lines[0])
    assert rep["ranks"]["world_size"] == 1 and ("RCCL" in rep["ranks"]["backend"])
    assert rep["frame_crc32"] == plain["frame_crc32"]
