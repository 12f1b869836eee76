"""A camera that moves (config 2 scene, 4K): per-frame kernel time of an orbit of `step` degrees per frame, queued launches,
against the same views standing still (each rendered until its own history has settled).
  python scripts/time_moving.py [step_degrees] [n_frames]        env: HVRT_TILE_SPLIT=0 / HVRT_TILE_ORDER=0 to compare"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch

step = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = 3840, 2160
ctx = ra.Context(0)
ctx.set_mesh(meshgen.cube_sphere(289))
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, 256, 256, 256)
ctx.build_grid(256, 256, 256, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
axis = np.asarray(up, np.float64); rel = np.asarray(pos, np.float64) - np.asarray(la, np.float64)
def cam_at(deg):
    a = np.deg2rad(deg)
    rot = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
    return capi.make_camera_frame((np.asarray(la, np.float64) + rot).astype(np.float32), la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
def render(c):
    ctx.render_part_device(c, W, H, 1, 0, H, 0, 1, rgb.data_ptr(), 0, 0, 0)
for _ in range(4): render(cam_at(0.0))
ctx.synchronize()
for k in range(1, n + 1): render(cam_at(step * k))
mov = ctx.render_history(n)
print("moving :", " ".join(f"{v:.2f}" for v in mov), f"  mean {np.mean(mov):.3f}", flush=True)
if os.environ.get("HVRT_MOVING_ONLY"): ctx.close(); sys.exit(0)
still = []
for k in range(1, n + 1, max(1, n // 5)):
    for _ in range(5): render(cam_at(step * k))
    still.append((k, float(np.mean(ctx.render_history(2)))))
print("still  :", " ".join(f"[{k}] {v:.2f}" for k, v in still), flush=True)
ctx.close()
