"""Launch duration against the number of identical 4K frames in one launch (config 2): the increment per frame is
the throughput cost of a frame, the rest of a one-frame launch is its start-up and tail."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
W, H = 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(meshgen.cube_sphere(289))
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, 256, 256, 256)
ctx.build_grid(256, 256, 256, mn, mx, vs)
cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
for n in (1, 2, 4, 8):
    buf = torch.empty((n, H, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ms = []
    for i in range(8):
        ctx.render_batch_part_device([cam] * n, W, H, 1, 0, H, 0, 1, H, d_rgb=buf.data_ptr())
        ctx.synchronize()
        if i >= 2: ms.append(ctx.timings().render_ms)
    print(f"{n} frames in one launch: {np.mean(ms):.3f} ms ({np.mean(ms)/n:.3f} per frame)", flush=True)
ctx.close()
