"""Multi-GPU rendering: one process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm).

The path shards by image row tiles with no exchange during compute:
  * the scene (triangles + CSR grid) is REPLICATED: built on rank 0, broadcast with RCCL, ingested by
    the other ranks through hvrt_set_mesh_device / hvrt_set_grid_csr_device;
  * row tile k of a frame belongs to rank k % world (interleaved: silhouette rows cost far more than
    background rows), each rank renders its tiles into a compact slab;
  * ONE collective per frame gathers the slabs (all_gather of equal, padded slabs over xGMI) and a
    de-interleave kernel (hvrt_assemble_parts_device) rebuilds the frame.  There is no reduction, so an
    N-GPU frame is bytewise the 1-GPU frame.
  * what crosses xGMI is, by default, the 8-bit image (rgb8=True): every rank converts its slab on the device with
    the reference's own rule (ImageSaver.h:28-33, hvrt_quantise_rgb8_device) before the collective, so 3 bytes per
    pixel move instead of 12; the float frame is exchanged when the caller asks (rgb8=False).

torch is plumbing here (device buffers, streams, the process group); every kernel is libhvrt's.
The collective helpers are device-agnostic so that the choreography is testable on CPU with gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from . import capi


@dataclass
class Partition:
    """Row-tile partition of an H-row frame over `world` parts (host arithmetic only)."""
    height: int
    tile_rows: int
    world: int

    def rows_of(self, part: int) -> int:
        n_tiles = -(-self.height // self.tile_rows)
        return sum(min(self.tile_rows, self.height - k * self.tile_rows) for k in range(part, n_tiles, self.world))

    @property
    def slab_rows(self) -> int:
        return max(self.rows_of(p) for p in range(self.world))

    def global_rows(self, part: int) -> np.ndarray:
        """Global row index of every local row of `part`, in slab order."""
        n_tiles = -(-self.height // self.tile_rows)
        out = []
        for k in range(part, n_tiles, self.world):
            out.extend(range(k * self.tile_rows, min((k + 1) * self.tile_rows, self.height)))
        return np.asarray(out, np.int64)


def gather_parts(local_slab: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of equal-shaped slabs -> [world, *slab.shape] on every rank (RCCL on GPUs, gloo on CPU)."""
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(local_slab.shape), dtype=local_slab.dtype, device=local_slab.device)
    if world == 1:
        out[0].copy_(local_slab)
        return out
    try:
        dist.all_gather_into_tensor(out, local_slab.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):      # gloo builds without the fused form
        dist.all_gather(list(out.unbind(0)), local_slab.contiguous(), group=group)
    return out


def exchange_parts(local: torch.Tensor, out: torch.Tensor | None = None, group=None, async_op: bool = False):
    """Batch of `world` frames: local[f] is this rank's slab of frame f.  One all_to_all sends it to rank f and
    brings in every rank's slab of frame `rank`: out[p] = part p of the frame this rank owns.  xGMI is a
    point-to-point mesh, so this is its native pattern, and each rank moves (world-1)/world of ONE frame per
    batch instead of world-1 frames as with an all_gather of every frame.  Returns (out, work or None)."""
    world = dist.get_world_size(group)
    assert local.shape[0] == world, "exchange_parts: one slab per rank expected"
    if out is None:
        out = torch.empty_like(local)
    if world == 1:
        out.copy_(local)
        return out, None
    work = dist.all_to_all_single(out, local.contiguous(), group=group, async_op=async_op)
    return out, work


def broadcast_arrays(tensors, src: int = 0, group=None):
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


class DistributedRenderer:
    """Replicated-scene, row-tile-partitioned renderer for one rank (one GPU)."""

    def __init__(self, ctx: capi.Context, rank: int, world: int, device: torch.device, tile_rows: int = 16):
        self.ctx, self.rank, self.world, self.device, self.tile_rows = ctx, rank, world, device, tile_rows
        # run libhvrt on a stream torch knows, so renders and collectives order through events
        self.stream = torch.cuda.ExternalStream(ctx.stream, device=device)

    # ---- scene replication -------------------------------------------------------------------
    def replicate_scene(self, tri_aos: np.ndarray | None, dims, src: int = 0):
        """Rank `src` holds tri_aos (others pass None).  Builds the grid on src, broadcasts mesh + CSR."""
        nx, ny, nz = dims
        meta = torch.zeros(16, dtype=torch.float64, device=self.device)
        if self.rank == src:
            self.ctx.set_mesh(tri_aos)
            mn, mx = self.ctx.mesh_bounds()
            vs = capi.voxel_size(mn, mx, nx, ny, nz)
            info = self.ctx.build_grid(nx, ny, nz, mn, mx, vs)
            meta[:9] = torch.tensor(list(mn) + list(mx) + [float(vs)] * 3, dtype=torch.float64)
            meta[9], meta[10] = float(info.n_triangles), float(info.n_refs)
        if self.world > 1:
            dist.broadcast(meta, src=src)
        m = meta.cpu().numpy()
        mn, mx, vs3 = m[:3].astype(np.float32), m[3:6].astype(np.float32), m[6:9].astype(np.float32)
        n_tris, n_refs = int(m[9]), int(m[10])
        if self.world > 1:
            tri = torch.empty(max(n_tris, 1) * 12, dtype=torch.float32, device=self.device)
            off = torch.empty(nx * ny * nz + 1, dtype=torch.int32, device=self.device)
            idx = torch.empty(max(n_refs, 1), dtype=torch.int32, device=self.device)
            if self.rank == src:
                self.ctx.export_scene_device(tri.data_ptr(), off.data_ptr(), idx.data_ptr())
                self.ctx.synchronize()
            torch.cuda.synchronize(self.device)
            broadcast_arrays([tri, off, idx], src=src)
            torch.cuda.synchronize(self.device)
            if self.rank != src:
                self.ctx.set_mesh_device(tri.data_ptr(), n_tris)
                self.ctx.set_grid_csr_device(nx, ny, nz, mn, mx, vs3, off.data_ptr(), idx.data_ptr(), n_refs)
        return mn, mx, vs3

    # ---- frames --------------------------------------------------------------------------------
    def alloc(self, width: int, height: int, n_frames: int = 1, rgb8: bool = False):
        """rgb8: quantise every slab on the device and exchange 3 bytes per pixel (width must be a multiple of 4)."""
        part = Partition(height, self.tile_rows, self.world)
        self.part = part
        if rgb8 and width % 4:
            raise ValueError("rgb8 exchange: the image width must be a multiple of 4 (rows move as 4-byte words)")
        self.rgb8 = rgb8
        self.local = torch.zeros((n_frames, part.slab_rows, width, 3), dtype=torch.float32, device=self.device)
        dt = torch.uint8 if rgb8 else torch.float32
        if rgb8:
            self.local8 = torch.zeros((n_frames, part.slab_rows, width, 3), dtype=torch.uint8, device=self.device)
        self.gathered = [torch.empty((self.world, part.slab_rows, width, 3), dtype=dt, device=self.device)
                         for _ in range(n_frames)]
        self.frames = torch.empty((n_frames, height, width, 3), dtype=dt, device=self.device)
        self.width, self.height = width, height

    def _quantise(self, src: torch.Tensor, dst: torch.Tensor):
        self.ctx.quantise_rgb8_device(src.data_ptr(), src.numel(), dst.data_ptr())

    def render_frames(self, cam_frames, spp: int = 1, seed: int = 0):
        """Renders this rank's tiles of every frame, gathers each frame as soon as it is rendered (the
        next frame's traversal overlaps the previous frame's all_gather), reassembles on every rank.
        Returns float frames, or uint8 frames when allocated with rgb8=True."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        rgb8 = getattr(self, "rgb8", False)
        send = self.local8 if rgb8 else self.local
        for f, cam in enumerate(cam_frames):
            self.ctx.render_part_device(cam, self.width, self.height, spp, seed, self.tile_rows, self.rank,
                                        self.world, d_rgb=self.local[f].data_ptr())
            if rgb8:
                self._quantise(self.local[f], self.local8[f])
            if self.world == 1:
                continue
            ev = self.stream.record_event()
            cur.wait_event(ev)
            dist.all_gather_into_tensor(self.gathered[f], send[f])
        if self.world == 1:
            cur.wait_stream(self.stream)
            return send[:, :self.height]                # one part: the slab is the frame
        self.stream.wait_stream(cur)
        for f in range(len(cam_frames)):
            self.ctx.assemble_parts_device(self.gathered[f].data_ptr(), self.width, self.height, self.tile_rows,
                                           self.world, self.part.slab_rows, 3 if rgb8 else 12, self.frames[f].data_ptr())
        cur.wait_stream(self.stream)
        return self.frames

    # ---- batches of `world` frames, one owner per frame (weak scaling) ---------------------------
    def alloc_batch(self, width: int, height: int, rgb8: bool = False):
        """Double-buffered: batch k renders while batch k-1 is exchanged.  rgb8: exchange the 8-bit image."""
        part = Partition(height, self.tile_rows, self.world)
        self.part = part
        if rgb8 and width % 4:
            raise ValueError("rgb8 exchange: the image width must be a multiple of 4 (rows move as 4-byte words)")
        self.rgb8 = rgb8
        shape = (self.world, part.slab_rows, width, 3)
        dt = torch.uint8 if rgb8 else torch.float32
        self.b_local = [torch.zeros(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.b_send = [torch.zeros(shape, dtype=torch.uint8, device=self.device) for _ in range(2)] if rgb8 else self.b_local
        self.b_recv = [torch.empty(shape, dtype=dt, device=self.device) for _ in range(2)]
        self.b_frame = [torch.empty((height, width, 3), dtype=dt, device=self.device) for _ in range(2)]
        self.width, self.height = width, height
        self._batch = 0
        self._pending = None       # (buffer index, work) of the batch whose exchange is in flight

    def _finish_pending(self):
        """Queues, on libhvrt's stream, the wait for the pending exchange and the reassembly of its frame."""
        if self._pending is None:
            return None
        b, work = self._pending
        self._pending = None
        if self.world == 1:
            return self.b_send[b][0, :self.height]
        with torch.cuda.stream(self.stream):
            if work is not None:
                work.wait()                                # stream-level wait, the host does not block
            self.ctx.assemble_parts_device(self.b_recv[b].data_ptr(), self.width, self.height, self.tile_rows,
                                           self.world, self.part.slab_rows, 3 if self.rgb8 else 12,
                                           self.b_frame[b].data_ptr())
        return self.b_frame[b]

    def render_batch(self, cam_frames, spp: int = 1, seed: int = 0):
        """Submits a batch of `world` frames: this rank renders its row tiles of every frame, then ONE
        all_to_all hands frame f's slabs to rank f.  The exchange runs on RCCL's stream while the next
        batch renders.  Returns the previous batch's frame owned by this rank (frame index == rank; valid
        on torch's current stream), or None on the first call; flush() returns the last one."""
        assert len(cam_frames) == self.world, "render_batch: one camera per rank"
        b = self._batch & 1
        self._batch += 1
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        # one launch for the whole batch: a part of a single frame is too small a launch (its duration is set
        # by its longest tiles), a queue over all frames' tiles keeps the GPU at throughput
        self.ctx.render_batch_part_device(list(cam_frames), self.width, self.height, spp, seed, self.tile_rows,
                                          self.rank, self.world, self.part.slab_rows,
                                          d_rgb=self.b_local[b].data_ptr())
        if self.rgb8:
            self._quantise(self.b_local[b], self.b_send[b])
        work = None
        if self.world > 1:
            cur.wait_event(self.stream.record_event())     # RCCL's stream orders itself after `cur`
            _, work = exchange_parts(self.b_send[b], self.b_recv[b], async_op=True)
        done = self._finish_pending()                      # batch k-1: exchanged while batch k rendered
        self._pending = (b, work)
        cur.wait_stream(self.stream)
        return done

    def flush(self):
        """Completes the batch in flight; returns its frame (valid on torch's current stream)."""
        done = self._finish_pending()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return done
