"""Per-launch render_ms of N launches of one camera (history effects show as a pattern): time_series.py c4 12"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch
which = sys.argv[1]; reps = int(sys.argv[2])
tris, g, W, H, spp = {"c2": (meshgen.cube_sphere(289), 256, 3840, 2160, 1), "c4": (meshgen.cube_sphere(289), 256, 7680, 4320, 16)}[which]
ctx = ra.Context(0); ctx.set_mesh(tris); mn, mx = ctx.mesh_bounds(); vs = capi.voxel_size(mn, mx, g, g, g); ctx.build_grid(g, g, g, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0"); hit = torch.empty((H, W), dtype=torch.int32, device="cuda:0"); t = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
out = []
for i in range(reps):
    ctx.render_part_device(cam, W, H, spp, 1, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0); ctx.synchronize()
    tx, ty = (W + 7) // 8, (H + 7) // 8
    c = ctx.tile_costs(tx * ty)
    out.append((ctx.timings().render_ms, int((c >> 31).sum())))
print(which, " ".join(f"{m:.2f}/{n}" for m, n in out))
