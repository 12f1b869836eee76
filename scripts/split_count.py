import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch
W,H=3840,2160
ctx = ra.Context(0); ctx.set_mesh(meshgen.cube_sphere(289)); mn, mx = ctx.mesh_bounds(); vs = capi.voxel_size(mn, mx, 256,256,256); ctx.build_grid(256,256,256, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
tx, ty = (W + 7) // 8, (H + 7) // 8
for i in range(8):
    ctx.render_part_device(cam, W, H, 1, 1, H, 0, 1, rgb.data_ptr(), 0, 0, 0); ctx.synchronize()
    ms = ctx.timings().render_ms
    c = ctx.tile_costs(tx * ty).astype(np.int64)
    fl = (c >> 31) != 0
    d = (c & 0x7FFFFFFF) * 16 / 2.4e3
    print(f"launch {i+1}: {ms:.3f} ms; split-flagged tiles {int(fl.sum())}; unsplit durations mean {d[~fl].mean():.1f} p99 {np.percentile(d[~fl],99):.0f} max {d[~fl].max():.0f} us; sum {d[~fl].sum()/1e3:.0f} ms", flush=True)
