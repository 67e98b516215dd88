"""One rank of the two-process test of include/rt_dist.h (started by tests/test_gpu_dist_cabi.py, one process per GPU):
    python dist_cabi_child.py <rank> <world> <id file> <output .npy on the root>
The root builds the small test scene and a sky, every rank joins the communicator (the id travels through the file), receives the
scene through rt_dist_scene_create / rt_dist_skybox_set, renders its bands of two accumulated frames and takes part in the gather;
the root stores the gathered colour and depth planes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, id_file, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    from raytracer_amd import api, dist_cabi, scenes
    from raytracer_amd.camera import Camera
    from raytracer_amd.params import Uniforms
    torch.cuda.set_device(rank)
    comm = dist_cabi.DistComm(dist_cabi.exchange_id_by_file(id_file, rank), world, rank, rank)
    arrays = sky = None
    if rank == 0:
        arrays = scenes.small_test_scene().as_dict()
        sky = np.random.default_rng(1).integers(0, 256, size=(6, 64, 64, 4), dtype=np.uint8)
    sc = comm.scene_create(arrays)
    comm.skybox_set(sc, sky)
    w, h = 200, 150
    stream = torch.cuda.current_stream().cuda_stream
    band = depth_band = None
    for k in range(2):
        u = Uniforms.from_camera(Camera(eye=(2.5, 2.5, 2.5), at=(0.0, 1.5, 0.0), aspect=w / h), time=k / 60.0).pack()
        fr = comm.frame_bands(api.make_frame(u, w, h, accum_index=k + 1, stream=stream))
        if band is None:
            rows = api.band_row_count(fr)
            band = torch.zeros((rows, w, 4), device="cuda"); depth_band = torch.zeros((rows, w), device="cuda")
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color, tg.depth = api.RT_MEM_DEVICE, 0, band.data_ptr(), depth_band.data_ptr()
        sc.render_raw(fr, tg)
    frame = torch.full((h, w, 4), float("nan"), device="cuda") if rank == 0 else None
    depth = torch.full((h, w), float("nan"), device="cuda") if rank == 0 else None
    comm.gather_frame(fr, band.data_ptr(), frame.data_ptr() if rank == 0 else None, 16, stream)
    comm.gather_frame(fr, depth_band.data_ptr(), depth.data_ptr() if rank == 0 else None, 4, stream)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out, np.concatenate([frame.cpu().numpy().reshape(h, w * 4), depth.cpu().numpy()], axis=1))
    sc.close()
    comm.close()


if __name__ == "__main__":
    main()
