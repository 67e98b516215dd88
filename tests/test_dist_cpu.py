"""CPU tests (gloo, world_size 2 and 3) of the multi-GPU band sharding: scene broadcast and band gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rdist


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, height, width, band_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        scene = None
        if rank == 0:
            rng = np.random.default_rng(0)
            scene = dict(vertices=rng.random((10, 4), dtype=np.float32), normals=rng.random((10, 4), dtype=np.float32), tri_refs=rng.integers(0, 10, (7, 4)).astype(np.uint32),
                         nodes=rng.integers(0, 99, 13).astype(np.uint32), bounds_min=np.array([-1, -2, -3], np.float32), bounds_max=np.array([1, 2, 3], np.float32),
                         sky=rng.integers(0, 256, (6, 4, 4, 4)).astype(np.uint8))
        got = rdist.broadcast_scene(scene, dev)
        digest = float(sum(np.asarray(v, np.float64).sum() for v in got.values()))
        # every rank fills its compact band buffer with (frame row, x) of the pixels it owns
        pad = rdist.padded_row_count(height, band_rows, world)
        rows = rdist.local_to_frame_rows(height, band_rows, world, rank)
        local = torch.full((pad, width, 4), -7.0)
        for lr, y in enumerate(rows):
            if y >= 0:
                local[lr, :, 0] = float(y)
                local[lr, :, 1] = torch.arange(width, dtype=torch.float32)
                local[lr, :, 2] = float(rank)
        frame = rdist.gather_frame(local, height, band_rows)
        ok = True
        if rank == 0:
            ys = torch.arange(height, dtype=torch.float32)[:, None].expand(height, width)
            xs = torch.arange(width, dtype=torch.float32)[None, :].expand(height, width)
            owner = ((torch.arange(height) // band_rows) % world).float()[:, None].expand(height, width)
            ok = bool(frame.shape == (height, width, 4) and torch.equal(frame[..., 0], ys) and torch.equal(frame[..., 1], xs) and torch.equal(frame[..., 2], owner))
        q.put((rank, ok, digest))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,band_rows", [(2, 40, 16), (2, 64, 16), (3, 50, 8), (2, 7, 16)])
def test_broadcast_and_gather(world, height, band_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, 12, band_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert len({round(d, 6) for _, _, d in res}) == 1   # every rank received the same scene


def test_row_maps_partition_the_frame():
    for height, band_rows, parts in [(2160, 16, 8), (720, 16, 4), (50, 8, 3), (7, 16, 2)]:
        seen = np.zeros(height, int)
        for p in range(parts):
            rows = rdist.local_to_frame_rows(height, band_rows, parts, p)
            assert len(rows) == rdist.local_row_count(height, band_rows, parts, p) <= rdist.padded_row_count(height, band_rows, parts)
            seen[rows[rows >= 0]] += 1
        assert (seen == 1).all()


def test_band_row_count_matches_library():
    from raytracer_amd import api
    u = np.zeros(70, np.float32)
    for height, band_rows, parts in [(2160, 16, 8), (720, 16, 4), (50, 8, 3), (7, 16, 2)]:
        for p in range(parts):
            fr = api.make_frame(u, 64, height, band=(band_rows, parts, p))
            assert api.band_row_count(fr) == rdist.local_row_count(height, band_rows, parts, p)


def _gatherer_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import dist as rdist
    h, w, rows = 100, 8, 16
    pad = rdist.padded_row_count(h, rows, world)
    g = rdist.FrameGatherer(h, rows)
    ok = True
    for k in range(3):
        local = torch.full((pad, w, 4), -1.0)
        ys = rdist.local_to_frame_rows(h, rows, world, rank)
        for lr, y in enumerate(ys):
            if y >= 0:
                local[lr] = float(1000 * k + y)
        g.submit(local)
        frame = g.last()
        if rank == 0:
            ok = ok and bool((frame[:, 0, 0] == torch.arange(h, dtype=torch.float32) + 1000 * k).all())
        else:
            ok = ok and frame is None
    q.put((rank, ok))
    dist.destroy_process_group()


def test_frame_gatherer_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    ps = [ctx.Process(target=_gatherer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == {0: True, 1: True}
