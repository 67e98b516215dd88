"""Multi-GPU rendering: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

Pixels are independent in the reference's shader (/root/reference/app/assets/ray_tracer.frag has no inter-pixel
communication; blending is per pixel), so the frame shards by screen-space bands: rows are cut into strips of
`band_rows` rows that are dealt round-robin to the ranks (cost per pixel varies by more than 10x between sky and
mesh, so contiguous bands would be unbalanced).  Each rank renders its strips into a compact buffer; progressive
accumulation stays local to the band.  Two collectives exist, both outside the per-pixel data path:

  broadcast_scene   once per scene: the four scene arrays, bounds and the skybox from rank 0
  gather_frame      once per presented frame: every rank's compact band buffer to rank 0, which re-interleaves them

The functions work on any backend/device, which is how the logic is tested on CPU with gloo; with gloo and device
tensors the payload is staged through host memory (gloo has no device gather).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def n_strips(height: int, band_rows: int) -> int:
    return (height + band_rows - 1) // band_rows


def local_row_count(height: int, band_rows: int, parts: int, part: int) -> int:
    """Rows of the compact buffer of `part` (whole strips) -- same rule as rt_band_row_count()."""
    s = n_strips(height, band_rows)
    mine = (s - part + parts - 1) // parts if s > part else 0
    return mine * band_rows


def padded_row_count(height: int, band_rows: int, parts: int) -> int:
    """Row count every rank's send buffer is padded to (rank 0 owns the most strips)."""
    return local_row_count(height, band_rows, parts, 0)


def local_to_frame_rows(height: int, band_rows: int, parts: int, part: int) -> np.ndarray:
    """Frame row of every compact row of `part` (-1 for rows of a strip that lies beyond the frame)."""
    n = local_row_count(height, band_rows, parts, part)
    lr = np.arange(n)
    y = ((lr // band_rows) * parts + part) * band_rows + lr % band_rows
    return np.where(y < height, y, -1)


def broadcast_scene(arrays: dict | None, device: torch.device, src: int = 0) -> dict:
    """Broadcast a scene dict (vertices, normals, tri_refs, nodes, bounds_min, bounds_max[, sky]) from `src`.
    Non-source ranks pass None.  Returns numpy arrays on every rank."""
    rank = dist.get_rank()
    keys = ["vertices", "normals", "tri_refs", "nodes", "bounds_min", "bounds_max", "sky"]
    dtypes = {"vertices": np.float32, "normals": np.float32, "tri_refs": np.uint32, "nodes": np.uint32, "bounds_min": np.float32, "bounds_max": np.float32, "sky": np.uint8}
    meta = [None]
    if rank == src:
        meta = [{k: tuple(np.asarray(arrays[k]).shape) for k in keys if arrays.get(k) is not None}]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for k in keys:
        if k not in meta[0]:
            continue
        shape = meta[0][k]
        nbytes = int(np.prod(shape)) * np.dtype(dtypes[k]).itemsize
        if rank == src:
            buf = torch.from_numpy(np.ascontiguousarray(arrays[k], dtypes[k]).view(np.uint8).reshape(-1).copy()).to(device)
        else:
            buf = torch.empty(max(nbytes, 0), dtype=torch.uint8, device=device)
        if nbytes:
            if dist.get_backend() == "gloo" and buf.is_cuda:
                host = buf.cpu()
                dist.broadcast(host, src=src)
                buf = host
            else:
                dist.broadcast(buf, src=src)
        out[k] = buf.cpu().numpy().view(dtypes[k]).reshape(shape)
    return out


def gather_frame(local: torch.Tensor, height: int, band_rows: int, dst: int = 0):
    """Gather compact band buffers ([padded_rows, width, C] on every rank) to `dst` and re-interleave them into a
    [height, width, C] frame.  Returns the frame on `dst`, None elsewhere."""
    parts, rank = dist.get_world_size(), dist.get_rank()
    pad = padded_row_count(height, band_rows, parts)
    assert local.shape[0] == pad, (local.shape, pad)
    if dist.get_backend() == "gloo" and local.is_cuda:
        host = local.cpu()
        bufs = [torch.empty_like(host) for _ in range(parts)] if rank == dst else None
        dist.gather(host, bufs, dst=dst)
        if rank == dst:
            bufs = [b.to(local.device) for b in bufs]
    else:
        bufs = [torch.empty_like(local) for _ in range(parts)] if rank == dst else None
        dist.gather(local, bufs, dst=dst)
    if rank != dst:
        return None
    # [part, strip_local, band_rows, W, C] -> [strip_local, part, band_rows, W, C] -> rows in frame order
    stack = torch.stack(bufs, 0)
    w, c = local.shape[1], local.shape[2]
    s_local = pad // band_rows
    frame = stack.view(parts, s_local, band_rows, w, c).permute(1, 0, 2, 3, 4).reshape(s_local * parts * band_rows, w, c)
    return frame[:height].contiguous()


class FrameGatherer:
    """Gathers every frame's bands to rank `dst` while the next frame is already being rendered.

    submit(local) snapshots the rank's compact band buffer on the caller's current stream (a device-to-device copy;
    the renderer keeps accumulating into `local`) and, on a side stream, gathers the snapshots into one preallocated
    [parts, rows, width, C] block on `dst` and re-interleaves that into the frame with a single strided copy; the next
    submit first waits until the previous gather has consumed the snapshot.  last() waits for the gather in flight and
    returns the newest frame on `dst` (None elsewhere).  Nothing is allocated per frame.  On CPU tensors everything is
    synchronous."""

    def __init__(self, height: int, band_rows: int, dst: int = 0):
        self.height, self.band_rows, self.dst = height, band_rows, dst
        self.snap = None
        self.stacked = None
        self.frame = None
        self.side = None
        self.done = None
        self.submitted = False

    def _setup(self, local: torch.Tensor):
        parts, rank = dist.get_world_size(), dist.get_rank()
        pad = padded_row_count(self.height, self.band_rows, parts)
        assert local.shape[0] == pad, (local.shape, pad)
        self.snap = torch.empty_like(local)
        if rank == self.dst:
            self.stacked = torch.empty((parts,) + tuple(local.shape), dtype=local.dtype, device=local.device)
            self.frame = torch.empty((pad * parts,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if local.is_cuda:
            self.side = torch.cuda.Stream(device=local.device)
            self.done = torch.cuda.Event()

    def _gather(self):
        parts, rank = dist.get_world_size(), dist.get_rank()
        staged = dist.get_backend() == "gloo" and self.snap.is_cuda      # gloo has no device gather
        src = self.snap.cpu() if staged else self.snap
        if rank == self.dst:
            block = torch.empty_like(self.stacked, device="cpu") if staged else self.stacked
            dist.gather(src, list(block.unbind(0)), dst=self.dst)
            if staged:
                self.stacked.copy_(block)
            w_c = tuple(self.snap.shape[1:])
            s_local = self.snap.shape[0] // self.band_rows
            # [part, strip_local, band_rows, ...] -> [strip_local, part, band_rows, ...] = rows in frame order
            self.frame.view((s_local, parts, self.band_rows) + w_c).copy_(self.stacked.view((parts, s_local, self.band_rows) + w_c).transpose(0, 1))
        else:
            dist.gather(src, None, dst=self.dst)

    def submit(self, local: torch.Tensor):
        if self.snap is None:
            self._setup(local)
        self.submitted = True
        if not local.is_cuda:
            self.snap.copy_(local)
            self._gather()
            return
        main = torch.cuda.current_stream(local.device)
        main.wait_event(self.done)          # the previous gather has read the snapshot (no-op the first time)
        self.snap.copy_(local)
        ready = torch.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            self._gather()
            self.done.record(self.side)

    def last(self):
        if self.side is not None:
            self.side.synchronize()
        if not self.submitted or self.frame is None:
            return None
        return self.frame[:self.height]
