"""Batched traceRay (hvrt_trace_device) on config 2: the 4K frame's primary rays handed over as a ray list."""
import os, sys
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
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
o = np.array(cam.origin, np.float32); ll = np.array(cam.lower_left, np.float32)
hor = np.array(cam.horizontal, np.float32); ver = np.array(cam.vertical, np.float32)
u = (np.arange(W, dtype=np.float32) / np.float32(W))[None, :, None]
v = (np.arange(H, dtype=np.float32) / np.float32(H))[:, None, None]
d = ((ll + hor * u) + ver * v) - o
d = (d / np.sqrt((d * d).sum(-1, keepdims=True))).astype(np.float32).reshape(-1, 3)
n = d.shape[0]
for name, perm in (("row-major ray order", None), ("8x8-tiled ray order", "tile"), ("shuffled ray order", "rand")):
    dd = d
    if perm == "tile":
        idx = np.arange(n).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
        dd = d[idx]
    elif perm == "rand":
        dd = d[np.random.default_rng(1).permutation(n)]
    d_d = torch.from_numpy(np.ascontiguousarray(dd)).to("cuda:0")
    d_o = torch.from_numpy(np.broadcast_to(o, (n, 3)).copy()).to("cuda:0")
    d_h = torch.zeros(n * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ms = []
    for i in range(6):
        ctx.trace_device(d_o.data_ptr(), d_d.data_ptr(), n, d_h.data_ptr())
        ctx.synchronize()
        ms.append(ctx.timings().render_ms)
    hits = int(d_h.cpu().numpy().view(capi.HIT_DTYPE)["hit"].sum())
    print(f"{name}: {min(ms[1:]):.3f} ms -> {n/min(ms[1:])/1e3:.0f} Mrays/s, {hits} hits", flush=True)
ctx.close()
