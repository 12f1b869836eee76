"""What the first hvrt_render of a fresh context costs (the reference constructs a tracer per image and renders once,
TraceImages.cpp:215-229): first context of the process, then a second context in the same process (kernels loaded,
scratch set up), each: first call and best of the next four.  Config 2, rgb + statistics (what IRayTracer::render asks for)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
tris, g, W, H = meshgen.cube_sphere(289), 256, 3840, 2160
lib = capi.load_library()
rgb = np.ones((H, W, 3), np.float32)
cnt = np.zeros(2, np.uint64)
p = lambda a: a.ctypes.data_as(C.c_void_p)
for k in range(2):
    t0 = time.perf_counter()
    ctx = ra.Context(0)
    t1 = time.perf_counter()
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, g, g, g)
    ctx.build_grid(g, g, g, mn, mx, vs)
    t2 = time.perf_counter()
    cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
    ts, ks = [], []
    for i in range(5):
        a = time.perf_counter()
        assert lib.hvrt_render(ctx._h, C.byref(cam), W, H, 1, 0, p(rgb), None, None, p(cnt)) == 0
        ts.append((time.perf_counter() - a) * 1e3)
        ks.append(ctx.timings().render_ms)
    print(f"context {k}: create {1e3*(t1-t0):.1f} ms, mesh + grid {1e3*(t2-t1):.1f} ms, first render {ts[0]:.2f} ms, second {ts[1]:.2f}, best {min(ts[1:]):.2f}; traversal by events: " + " ".join(f"{k:.2f}" for k in ks), flush=True)
    ctx.close()
