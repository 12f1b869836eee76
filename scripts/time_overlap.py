"""One-GPU rehearsal of the multi-GPU batch step's overlap: a rank's share of N frames is rendered on libhvrt's
stream while a stand-in for the previous batch's exchange (device copies of the slab buffer, `reps` times over, on
a second stream, ordered by the same events as DistributedRenderer.render_batch) is in flight.  If the step time
does not grow with the stand-in's length, a collective's kernels do get their share of the GPU beside the
persistent render kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

N, W, H, TR = 8, 3840, 2160, 16
ctx = ra.Context(0)
ctx.set_mesh(meshgen.cube_sphere(289))
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, 256, 256, 256)
ctx.build_grid(256, 256, 256, mn, mx, vs)
cams = [capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H)) for _ in range(N)]
dev = torch.device("cuda:0")
rows = capi.part_rows(H, TR, 0, N)
slab = max(capi.part_rows(H, TR, p, N) for p in range(N))
local = [torch.zeros((N, slab, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
recv = [torch.empty_like(local[0]) for _ in range(2)]
rs = torch.cuda.ExternalStream(ctx.stream, device=dev)
cs = torch.cuda.Stream(device=dev)


def run(reps, steps=30):
    pend = None
    torch.cuda.synchronize()
    t0 = None
    for k in range(steps + 5):
        if k == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        b = k & 1
        ctx.render_batch_part_device(cams, W, H, 1, 0, TR, 0, N, slab, d_rgb=local[b].data_ptr())
        ev = rs.record_event()
        cs.wait_event(ev)
        with torch.cuda.stream(cs):
            for _ in range(reps):
                recv[b].copy_(local[b])
            done = cs.record_event()
        if pend is not None:
            rs.wait_event(pend)          # batch k-1's exchange must be over before its buffers are reused
        pend = done
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for reps in (0, 1, 8, 32):
    # the stand-in alone
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        for _ in range(reps * 10):
            recv[0].copy_(local[0])
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 10 * 1e3
    print(f"stand-in x{reps}: alone {alone:.3f} ms per batch; step with overlap {run(reps):.3f} ms", flush=True)
ctx.close()
