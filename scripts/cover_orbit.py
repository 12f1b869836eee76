"""The tile cover along an orbit (config 2's scene, 4K): marked tiles and the triangles that went through the second pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch
W, H = 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(meshgen.cube_sphere(289))
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, 256, 256, 256)
ctx.build_grid(256, 256, 256, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
axis = np.asarray(up, np.float64); rel = np.asarray(pos, np.float64) - np.asarray(la, np.float64)
def cam_at(deg):
    a = np.deg2rad(deg)
    rot = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
    return capi.make_camera_frame((np.asarray(la, np.float64) + rot).astype(np.float32), la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
degs = [int(v) for v in os.environ.get("HVRT_ORBIT_DEGS", "0,1,2,3,5,8,13,21,34,45").split(",")]
for deg in degs:
    ctx.render_part_device(cam_at(float(deg)), W, H, 1, 0, H, 0, 1, rgb.data_ptr(), 0, 0, 0); ctx.synchronize()
    ms = ctx.timings().render_ms
    cov = ctx.tile_cover()
    print(f"{deg:3d} deg: render {ms:.3f} ms; marked {cov.mean()*100:.2f} % of the tiles; second pass {ctx.last_cover_general_case} triangles", flush=True)
ctx.close()
