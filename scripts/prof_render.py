"""Short profiling target: config 2 scene, a few build + render launches (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

n, g, W, H, spp, seed = 289, 256, 3840, 2160, 1, 0
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
mesh = None
if which == "small":
    n, g, W, H = 58, 50, 800, 600
elif which == "c4":
    W, H, spp, seed = 7680, 4320, 16, 1
elif which == "c3":
    mesh, g = meshgen.skewed_statue(913, 1100), 512
ctx = ra.Context(0)
ctx.set_mesh(mesh if mesh is not None else meshgen.cube_sphere(n))
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
for _ in range(2):
    ctx.build_grid(g, g, g, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
import torch
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
hit = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
t = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
for _ in range(4):
    ctx.render_part_device(cam, W, H, spp, seed, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0)
    ctx.synchronize()
print("render_ms", ctx.timings().render_ms, "build_ms", ctx.timings().build_total_ms)
ctx.close()
