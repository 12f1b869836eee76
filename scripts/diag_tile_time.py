"""Diagnostic build only (libhvrt_diag.so, -DHVRT_DIAG_TILE_TIME): longest and mean 8x8 tile in shader clocks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HVRT_LIB"] = os.path.join(ROOT, "raytracer_amd", "lib", "libhvrt_diag.so")
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

tris, g, W, H = meshgen.cube_sphere(289), 256, 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
ctx.build_grid(g, g, g, mn, mx, vs)
cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
for _ in range(2):
    rgb, hit, t, cnt = ctx.render(cam, W, H)
d = t[::8, ::8].astype(np.float64)             # one value per tile: duration in clocks
st = hit[::8, ::8].astype(np.int64)
st = (st - st.min()) * 256.0
print("tile duration percentiles (clk):", {q: int(np.percentile(d, q)) for q in (50, 90, 99, 99.9, 100)})
ys, xs = np.unravel_index(np.argsort(d, axis=None)[::-1][:12], d.shape)
for y, x in zip(ys, xs):
    print(f"  tile ({x},{y}): {int(d[y, x])} clk, started at {int(st[y, x])} clk")
rows = d.mean(axis=1)
print("mean duration by tile row (every 27th):", [int(v) for v in rows[::27]])
print("kernel span in clk:", int((st + d).max()))
n_tiles = (W // 8) * (H // 8)
print(f"tiles {n_tiles}: longest {cnt[0]} clk, mean {cnt[1]/n_tiles:.0f} clk, kernel {ctx.timings().render_ms:.3f} ms")
ctx.close()
