"""Config 2 scene at 4K seen from 0, 10 and 45 degrees around the vertical axis: the first five launches of each view on a
grid that has just been built (no tile history), and the view's V (voxels) and T (tests) per ray.
  env: HVRT_TILE_ORDER=0 (every launch like a first one), HVRT_TILE_SPLIT=0 (order only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
W, H = 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(meshgen.cube_sphere(289))
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, 256, 256, 256)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
axis = np.asarray(up, np.float64); rel = np.asarray(pos, np.float64) - np.asarray(la, np.float64)
def cam_at(deg):
    a = np.deg2rad(deg)
    rot = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
    return capi.make_camera_frame((np.asarray(la, np.float64) + rot).astype(np.float32), la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda:0")
for deg in (0.0, 10.0, 45.0):
    ctx.build_grid(256, 256, 256, mn, mx, vs)      # forgets the history
    ms = []
    for i in range(5):
        ctx.render_part_device(cam_at(deg), W, H, 1, 0, H, 0, 1, rgb.data_ptr(), 0, 0, 0)
        ctx.synchronize(); ms.append(ctx.timings().render_ms)
    cnt.zero_(); torch.cuda.synchronize()
    ctx.render_part_device(cam_at(deg), W, H, 1, 0, H, 0, 1, rgb.data_ptr(), 0, 0, cnt.data_ptr())
    ctx.synchronize()
    V, T = (cnt.cpu().numpy() / float(W * H)).tolist()
    print(f"deg {deg}: launches " + " ".join(f"{v:.2f}" for v in ms) + f"   V {V:.1f} T {T:.1f}", flush=True)
ctx.close()
