"""Times the traversal kernel (no counters) on config 2 (or config 3-like skewed scene): mean of N launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if which == "c2":
    tris, g, W, H = meshgen.cube_sphere(289), 256, 3840, 2160
elif which == "c3":
    tris, g, W, H = meshgen.skewed_statue(289, 2100), 512, 3840, 2160     # 1.0M base + 8.8M patch
elif which == "c3b":
    tris, g, W, H = meshgen.skewed_statue(913, 1100), 512, 3840, 2160     # the bench's config 3
elif which == "c4":
    tris, g, W, H = meshgen.cube_sphere(289), 256, 7680, 4320                 # config 2 scene at 8K, 16 spp
elif which == "small":
    tris, g, W, H = meshgen.cube_sphere(58), 50, 800, 600
spp = 16 if which == "c4" else 1
if os.environ.get("HVRT_TIME_WH"):          # e.g. 1920x1080: same scene, other frame size
    W, H = (int(v) for v in os.environ["HVRT_TIME_WH"].split("x"))
ctx = ra.Context(0)
import os
if os.environ.get("HVRT_TIME_MODE"):          # 1: nearest-hit mode (hvrt_set_mode)
    ctx.set_mode(int(os.environ["HVRT_TIME_MODE"]))
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
info = ctx.build_grid(g, g, g, mn, mx, vs)
bt = []
for _ in range(3):
    ctx.build_grid(g, g, g, mn, mx, vs); tm = ctx.timings(); bt.append(tm.build_total_ms)
print(f"{which}: tris {info.n_triangles} refs {info.n_refs} maxlist {info.max_list} build {min(bt):.3f} ms "
      f"(count {tm.build_count_ms:.3f} scan {tm.build_scan_ms:.3f} fill {tm.build_fill_ms:.3f} sort {tm.build_sort_ms:.3f})", flush=True)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
hit = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
t = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
ms = []
for i in range(reps + 2):
    ctx.render_part_device(cam, W, H, spp, 1, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0)
    ctx.synchronize()
    if i >= 2 or os.environ.get("HVRT_TIME_LIST"): ms.append(ctx.timings().render_ms)
if os.environ.get("HVRT_TIME_QUEUED"):        # the same launches queued back to back, waited for once (hvrt_render_history)
    import time
    t0 = time.perf_counter()
    for i in range(reps):
        ctx.render_part_device(cam, W, H, spp, 1, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0)
    q = ctx.render_history(reps)
    wall = (time.perf_counter() - t0) * 1e3 / reps
    print(f"queued: wall {wall:.3f} ms per launch; per launch " + " ".join(f"{v:.3f}" for v in q), flush=True)
    t0 = time.perf_counter()
    for i in range(reps):
        ctx.render_part_device(cam, W, H, spp, 1, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0)
        ctx.synchronize()
    print(f"waited for one by one: wall {(time.perf_counter() - t0) * 1e3 / reps:.3f} ms per launch", flush=True)
if os.environ.get("HVRT_TIME_LIST"): print("launches:", " ".join(f"{v:.2f}" for v in ms), flush=True)
print(f"render {W}x{H} spp {spp}: mean {np.mean(ms):.3f} ms min {np.min(ms):.3f} ms -> {W*H*spp/np.mean(ms)/1e3:.1f} Mrays/s; checksum hit {int(hit.long().sum())}", flush=True)
ctx.close()
