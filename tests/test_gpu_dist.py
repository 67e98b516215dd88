"""Two ranks sharing the one GPU of the box (gloo backend: RCCL does not accept two ranks on one device): every rank
renders its interleaved bands into a compact device buffer, FrameGatherer collects them on rank 0 over two accumulated
frames, and the result equals the frame a single rank renders."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, w=320, h=200, rows=16):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import api, scenes, dist as rdist
    from raytracer_amd.camera import Camera
    from raytracer_amd.params import Uniforms
    dev = torch.device("cuda", 0)
    arrays = None
    if rank == 0:
        arrays = scenes.small_test_scene().as_dict()
        arrays["sky"] = np.random.default_rng(1).integers(0, 256, size=(6, 32, 32, 4), dtype=np.uint8)
    arrays = rdist.broadcast_scene(arrays, dev)
    sc = api.Scene.from_dict(arrays, device=0)
    sc.set_skybox(arrays["sky"])
    us = [Uniforms.from_camera(Camera(eye=(2.5, 2.5, 2.5), at=(0.0, 1.5, 0.0), aspect=w / h), time=k / 60.0).pack() for k in range(2)]
    band = torch.zeros((rdist.padded_row_count(h, rows, world), w, 4), dtype=torch.float32, device=dev)
    g = rdist.FrameGatherer(h, rows)
    for k, u in enumerate(us):
        fr = api.make_frame(u, w, h, accum_index=k + 1, band=(rows, world, rank), compact=True, stream=torch.cuda.current_stream().cuda_stream)
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 0, band.data_ptr()
        sc.render_raw(fr, tg)
        g.submit(band)
    frame = g.last()
    ok = True
    if rank == 0:
        ref = None
        for k, u in enumerate(us):
            ref = sc.render(u, w, h, accum_index=k + 1, color=ref, planes=False)["rgba"]
        ok = bool(np.array_equal(frame.cpu().numpy().view(np.uint32), ref.view(np.uint32)))
    else:
        ok = frame is None
    sc.close()
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size,rows", [(2, (320, 200), 16), (3, (250, 100), 8)])   # ragged: 12.5 strips for 2 ranks, 12.5 for 3
def test_ranks_bands_gathered_equal_single_rank_frame(world, size, rows):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() + 7 * world) % 200
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, size[0], size[1], rows)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == {r: True for r in range(world)}
