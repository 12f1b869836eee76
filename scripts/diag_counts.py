"""Diagnostic build only (libhvrt_diag.so, -DHVRT_DIAG): tests actually executed per ray and wave-level test iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HVRT_LIB", os.path.join(ROOT, "raytracer_amd", "lib", "libhvrt_diag.so"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c3":
    tris, g, W, H = meshgen.skewed_statue(289, 2100), 512, 3840, 2160
else:
    tris, g, W, H = meshgen.cube_sphere(289), 256, 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
ctx.build_grid(g, g, g, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
for _ in range(3):
    cnt = ctx.render(cam, W, H)[3]        # the third launch: tile history in place
print('render_ms', ctx.timings().render_ms)
n = W * H
print(f"raw counters / rays: first {cnt[0]/n:.3f} second {cnt[1]/n:.3f}   (DIAG 11: cooperative | per-lane chunks per ray slot; "
      f"12: cooperative | per-lane wave test iterations per ray slot; 13: cooperative iterations | tests the lanes wanted; "
      f"21: steps in two-axis runs | general steps of runs with k >= 4; 22: general steps of runs with k < 4 | runs; "
      f"23: general steps of runs with k >= 4 where every lane has some far axis | where a lane has none)")
print(f"[DIAG=31: clocks waiting for the queue head in all {cnt[0]:.4g}, pulls {cnt[1]:.4g} -> {cnt[0]/max(cnt[1],1):.0f} clocks per pull]")
print(f"[DIAG=32: wave lifetimes summed {cnt[0]:.4g} clocks, inside tiles {cnt[1]:.4g} -> {cnt[1]/max(cnt[0],1):.3f} of a wave's life is tile work; "
      f"mean lifetime {cnt[0]/8192:.4g} clocks per wave (8192 resident waves)]")
print(f"[DIAG=33: mean end {cnt[0]/8192:.1f}, last end {cnt[1]:.1f} (10-ns ticks, low 32 bits) -> the mean wave ends {(cnt[1]-cnt[0]/8192)/100:.1f} us before the last]")
print(f"[DIAG=34: mean start {cnt[0]/8192:.1f}, first start {0xFFFFFFFF-cnt[1]:.1f} -> the mean wave starts {(cnt[0]/8192-(0xFFFFFFFF-cnt[1]))/100:.1f} us after the first]")
print(f"[DIAG=2: second/first = share of wave iterations whose lanes all test one entry: {cnt[1]/max(cnt[0],1):.3f}]")
print(f"tests executed per ray {cnt[1]/n:.2f}; wave test iterations per ray-slot {cnt[0]/n:.2f} (x64 lanes)")
ctx.close()
