"""hvrt_render with host buffers (the PCIe-inclusive rate): config 2, buffers reused across calls."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
rgb = np.zeros((H, W, 3), np.float32); hit = np.zeros((H, W), np.uint32); t = np.zeros((H, W), np.float32)
rgb[:] = 1; hit[:] = 1; t[:] = 1      # touch the pages: a caller's std::vector<Vec3>(W * H) has been zero-filled too
lib = capi.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
cnt = np.zeros(2, np.uint64)
for what, args in (("rgb + counters (what IRayTracer::render needs)", (p(rgb), None, None, p(cnt))), ("rgb + hit + t", (p(rgb), p(hit), p(t), None)), ("rgb only", (p(rgb), None, None, None))):
    ts = []
    for i in range(8):
        t0 = time.perf_counter()
        rc = lib.hvrt_render(ctx._h, C.byref(cam), W, H, 1, 0, args[0], args[1], args[2], args[3])
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    best = min(ts[1:])
    print(f"hvrt_render {what}: first call {ts[0]*1e3:.2f} ms, then best {best*1e3:.2f} ms per 4K frame -> {W*H/best/1e6:.0f} Mrays/s PCIe-inclusive; "
          f"hits {ctx.last_hit_count()} == {int((hit != 0xFFFFFFFF).sum()) if what.startswith('rgb + hit') else 'n/a'}", flush=True)
ctx.close()
