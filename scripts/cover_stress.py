"""The tile cover against the plain traversal on the bench meshes at full size: N random cameras, each frame rendered by the
counting kernels (every tile walked) and by the plain ones (cover), hit ids / t / colours compared bit for bit.
  python scripts/cover_stress.py [c2|c3] [N] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
tris, g = (meshgen.cube_sphere(289), 256) if which == "c2" else (meshgen.skewed_statue(289, 700), 512)
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
ctx.build_grid(g, g, g, mn, mx, vs)
# a second context on the same scene whose launches never have history: the give-up variant (all tiles at once where they
# are few) and the split kernel's walk that looks ahead behind it
os.environ["HVRT_TILE_ORDER"] = "0"
ctx2 = ra.Context(0)
del os.environ["HVRT_TILE_ORDER"]
info = ctx.grid_info()
a_, o_, i_ = ctx.device_arrays()
ctx2.set_mesh_device(a_, info.n_triangles)
ctx2.set_grid_csr_device(g, g, g, mn, mx, np.asarray([vs] * 3, np.float32), o_, i_, info.n_refs)
c = 0.5 * (np.asarray(mn, np.float64) + np.asarray(mx, np.float64))
ext = float(np.max(np.asarray(mx) - np.asarray(mn)))
bad = 0
for k in range(n):
    W, H = [(3840, 2160), (1920, 1080), (2561, 1439), (7680, 4320)][k % 4]
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    dist = ext * float(rng.choice([0.3, 0.52, 0.8, 1.5, 2.5, 6.0]))
    pos = c + d * dist
    la = c + rng.normal(size=3) * ext * float(rng.choice([0.0, 0.1, 0.5]))
    fov = float(rng.choice([10.0, 40.0, 60.0, 90.0]))
    fr = capi.make_camera_frame(tuple(pos.astype(np.float32)), tuple(la.astype(np.float32)), (0, 1, 0), fov, W / H)
    bufs = []
    for counters in (True, False):
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
        hit = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
        t = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda:0")
        for launch in range(1 if counters else 2):
            ctx.render_part_device(fr, W, H, 1, 0, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), cnt.data_ptr() if counters else 0)
            ctx.synchronize()
        bufs.append((rgb, hit, t))
    rgb2 = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0"); hit2 = torch.empty((H, W), dtype=torch.int32, device="cuda:0"); t2 = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
    ctx2.render_part_device(fr, W, H, 1, 0, H, 0, 1, rgb2.data_ptr(), hit2.data_ptr(), t2.data_ptr(), 0)
    ctx2.synchronize()
    same2 = bool((bufs[0][1] == hit2).all()) and bool((bufs[0][2].view(torch.int32) == t2.view(torch.int32)).all()) and bool((bufs[0][0].view(torch.int32) == rgb2.view(torch.int32)).all())
    same = same2 and bool((bufs[0][1] == bufs[1][1]).all()) and bool((bufs[0][2].view(torch.int32) == bufs[1][2].view(torch.int32)).all()) \
        and bool((bufs[0][0].view(torch.int32) == bufs[1][0].view(torch.int32)).all())
    cov = ctx.tile_cover()
    hits = float((bufs[0][1] != -1).float().mean())
    print(f"{k:3d} {W}x{H} dist {dist / ext:4.2f} fov {fov:5.1f}: hit fraction {hits:.4f}, tiles marked {cov.mean():.4f} -> {'same' if same else 'DIFFERENT'}", flush=True)
    if not same:
        bad += 1
        diff = (bufs[0][1] != bufs[1][1]).nonzero()
        print("    first differing pixels (y, x):", diff[:5].tolist(), "counting kernel:", bufs[0][1][diff[0][0], diff[0][1]].item(), "plain:", bufs[1][1][diff[0][0], diff[0][1]].item())
print("frames that differ:", bad)
sys.exit(1 if bad else 0)
