"""Cost of the row-tile partition on one GPU: N part-renders of one 4K frame (what a rank does per batch at N GPUs) vs one full render."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch

tris, g, W, H = meshgen.cube_sphere(289), 256, 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
ctx.build_grid(g, g, g, mn, mx, vs)
cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
for n in (1, 2, 4, 8):
    slab = max(capi.part_rows(H, 16, p, n) for p in range(n))
    buf = torch.empty((n, slab, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(6):
        ctx.synchronize()
        t0 = time.perf_counter()
        for p in range(n):
            ctx.render_part_device(cam, W, H, 1, 0, 16, p, n, d_rgb=buf[p].data_ptr())
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"n_parts {n}: {best:.3f} ms for the whole frame in {n} launches", flush=True)
    # what a rank does per step of the multi-GPU batch mode: its part of n frames, ONE launch
    best = 1e9
    for rep in range(6):
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.render_batch_part_device([cam] * n, W, H, 1, 0, 16, 0, n, slab, d_rgb=buf.data_ptr())
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"           {best:.3f} ms for part 0 of {n} frames in one launch ({W*H/best/1e3:.0f} Mrays/s)", flush=True)
ctx.close()
