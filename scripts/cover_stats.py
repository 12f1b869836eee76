"""Tile cover on a bench scene: share of tiles marked, share that hold a hit, and the check hits-are-inside-the-cover."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
view = sys.argv[3] if len(sys.argv) > 3 else "front"
if which == "c2":
    tris, g, W, H, spp = meshgen.cube_sphere(289), 256, 3840, 2160, 1
elif which == "c3":
    tris, g, W, H, spp = meshgen.skewed_statue(913, 1100), 512, 3840, 2160, 1
elif which == "c4":
    tris, g, W, H, spp = meshgen.cube_sphere(289), 256, 7680, 4320, 16
elif which == "small":
    tris, g, W, H, spp = meshgen.cube_sphere(58), 50, 800, 600, 1
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
info = ctx.build_grid(g, g, g, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, view, aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
hit = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
t = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
ms = []
for i in range(reps + 2):
    ctx.render_part_device(cam, W, H, spp, 1, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0)
    ctx.synchronize()
    if i >= 2: ms.append(ctx.timings().render_ms)
print(f"{which} {view}: tris {info.n_triangles}; render {W}x{H} spp {spp}: mean {np.mean(ms):.3f} ms min {np.min(ms):.3f} ms -> {W*H*spp/np.mean(ms)/1e3:.1f} Mrays/s", flush=True)
try:
    cov = ctx.tile_cover()
except Exception as e:
    print("no cover:", e); sys.exit(0)
h = (hit.cpu().numpy() != -1)
ty, tx = cov.shape
hp = np.zeros((ty * 8, tx * 8), bool); hp[:H, :W] = h
ht = hp.reshape(ty, 8, tx, 8).any(axis=(1, 3))
print("triangles in the general case:", ctx.last_cover_general_case)
print(f"tiles {tx} x {ty}: marked {cov.mean():.4f}, with a hit {ht.mean():.4f}, hit outside the cover: {int((ht & ~cov).sum())}", flush=True)
rows = cov.sum(axis=1)
print("marked per row of tiles (every 16th):", rows[::16].tolist())

cost = ctx.tile_costs(tx * ty).reshape(ty, tx)
split = (cost >> 31) != 0
c = (cost & 0x7FFFFFFF).astype(np.float64) * 16 / 2.4e3      # microseconds at 2.4 GHz
w = cov & ~split
print(f"walked by k_render: {int(w.sum())} tiles, us: mean {c[w].mean():.1f} p50 {np.percentile(c[w],50):.1f} p90 {np.percentile(c[w],90):.1f} p99 {np.percentile(c[w],99):.1f} max {c[w].max():.1f}; sum {c[w].sum()/1e3:.1f} ms")
print(f"k_render_split: {int(split.sum())} tiles, quad-sum us: mean {c[split].mean() if split.any() else 0:.1f} max {c[split].max() if split.any() else 0:.1f}; sum {c[split].sum()/1e3:.1f} ms")
print(f"skipped: {int((~cov).sum())} tiles, us: mean {c[~cov].mean():.2f} max {c[~cov].max():.2f}; sum {c[~cov].sum()/1e3:.1f} ms")
