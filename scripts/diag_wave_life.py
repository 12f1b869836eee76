"""Diagnostic builds 35 / 36 (make -C raytracer_amd/csrc diag DIAG=35): how much of a launch the mean wave is alive."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HVRT_LIB", os.path.join(ROOT, "raytracer_amd", "lib", "libhvrt_diag.so"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
tris, g, W, H, spp = meshgen.cube_sphere(289), 256, 3840, 2160, 1
if which == "c4": W, H, spp = 7680, 4320, 16
ctx = ra.Context(0)
ctx.set_mesh(tris); mn, mx = ctx.mesh_bounds(); vs = capi.voxel_size(mn, mx, g, g, g); ctx.build_grid(g, g, g, mn, mx, vs)
cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
for i in range(4):
    ctx.render(cam, W, H, spp=spp, seed=1, want_hit=True, want_t=True, want_counters=False)
    ms = ctx.timings().render_ms
    raw = ctx.last_hit_count()
    life = raw / 8192 / 100.0
    print(f'   raw counter {raw} (DIAG=40: low word = waves more than 0.5 ms late: {raw & 0xFFFFFFFF}, high word = more than 20 us late: {raw >> 32}; DIAG=41: tiles done by such waves)')
    print(f"launch {i}: render_ms {ms:.3f}, mean wave lifetime (DIAG=35; DIAG=36: mean start delay behind workgroup 0) {life:.1f} us -> {life/ms/10:.1f} % of the launch", flush=True)
