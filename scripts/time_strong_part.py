"""What ONE rank does per step of `bench.py --scaling strong` at N GPUs, measured on one GPU: the same part (rank p of N,
16-row tiles interleaved) of the same frame, launched again and again so that the tile history of that shape is in use."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch

W, H, spp, seed = (7680, 4320, 16, 1) if "c4" in sys.argv[1:] else (3840, 2160, 1, 0)
tris, g = meshgen.cube_sphere(289), 256
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
ctx.build_grid(g, g, g, mn, mx, vs)
cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
for n in (1, 2, 4, 8):
    slab = max(capi.part_rows(H, 16, p, n) for p in range(n))
    buf = torch.empty((slab, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for p in sorted({0, n // 2, n - 1}):
        ms = []
        for rep in range(7):
            ctx.render_part_device(cam, W, H, spp, seed, 16, p, n, d_rgb=buf.data_ptr())
            ctx.synchronize()
            ms.append(ctx.timings().render_ms)
        print(f"N {n} part {p}: first {ms[0]:.3f} second {ms[1]:.3f} later {np.mean(ms[2:]):.3f} ms", flush=True)
ctx.close()
