"""When the tiles of a launch start: after warm-up launches, one launch records start times instead of durations (HVRT_DEBUG_TILE_START)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen
import torch
which = sys.argv[1]
tris, g, W, H, spp = {"c2": lambda: (meshgen.cube_sphere(289), 256, 3840, 2160, 1), "c3": lambda: (meshgen.skewed_statue(913, 1100), 512, 3840, 2160, 1), "c4": lambda: (meshgen.cube_sphere(289), 256, 7680, 4320, 16)}[which]()
ctx = ra.Context(0); ctx.set_mesh(tris); mn, mx = ctx.mesh_bounds(); vs = capi.voxel_size(mn, mx, g, g, g); ctx.build_grid(g, g, g, mn, mx, vs)
pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
cam = capi.make_camera_frame(pos, la, up, fov, asp)
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0"); hit = torch.empty((H, W), dtype=torch.int32, device="cuda:0"); t = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
tx, ty = (W + 7) // 8, (H + 7) // 8
def launch():
    ctx.render_part_device(cam, W, H, spp, 1, H, 0, 1, rgb.data_ptr(), hit.data_ptr(), t.data_ptr(), 0); ctx.synchronize()
    return ctx.timings().render_ms
for _ in range(5): ms = launch()
if len(sys.argv) > 2 and sys.argv[2] == "moving":      # an orbit of 1 degree per frame: the recorded launch is ordered by another frame's durations
    axis = np.asarray(up, np.float64) / np.linalg.norm(up); rel = np.asarray(pos, np.float64) - np.asarray(la, np.float64)
    def orbit(k):
        a = np.deg2rad(float(k))
        rot = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
        return capi.make_camera_frame((np.asarray(la, np.float64) + rot).astype(np.float32), la, up, fov, asp)
    for k in range(1, 6):
        cam = orbit(k); ms = launch()
    cam = orbit(6)
dur = ctx.tile_costs(tx * ty).astype(np.int64)
os.environ["HVRT_DEBUG_TILE_START"] = "1"
ms2 = launch()
os.environ["HVRT_DEBUG_TILE_START"] = "0"
st = ctx.tile_costs(tx * ty).astype(np.int64)
if len(sys.argv) > 2 and sys.argv[2] == "moving":
    # the recorded launch left the order as it was: the same camera once more gives this frame's own durations under that order
    launch()
    dur = ctx.tile_costs(tx * ty).astype(np.int64)
split = (dur >> 31) != 0
d_us = (dur & 0x7FFFFFFF) * 16 / 2.4e3
ok = ~split
st31 = st & 0x7FFFFFFF
cov = ctx.tile_cover().reshape(-1)
w = ok & cov
ref = int(np.median(st31[w]))
rel = ((st31 - ref + (1 << 30)) % (1 << 31)) - (1 << 30)      # signed distance from the median, wrap-safe
s_us = (rel - rel[w].min()) / 100.0
print(f"{which}: launch {ms:.2f} ms, the recorded one {ms2:.2f} ms; {int(w.sum())} walked tiles in k_render")
end = s_us + d_us
sp = split & cov
if sp.any():
    q_us = (st[sp] & 0x7FFFFFFF) * 16 / 2.4e3      # the recorded launch skips the order build: the split tiles hold the sum of their 16 quads
    print(f"split tiles: {int(sp.sum())}; wave time per tile (16 quads) p10 {np.percentile(q_us,10):.0f} p50 {np.percentile(q_us,50):.0f} p90 {np.percentile(q_us,90):.0f} max {q_us.max():.0f} us, sum {q_us.sum()/1e3:.1f} ms")
print(f"walked tiles: durations p10 {np.percentile(d_us[w],10):.0f} p50 {np.percentile(d_us[w],50):.0f} p90 {np.percentile(d_us[w],90):.0f} p99 {np.percentile(d_us[w],99):.0f} max {d_us[w].max():.0f} us, sum {d_us[w].sum()/1e3:.1f} ms; split tiles {int((split & cov).sum())}, their durations p50 {np.percentile(d_us[split & cov],50) if (split & cov).any() else 0:.0f} max {d_us[split & cov].max() if (split & cov).any() else 0:.0f} sum {d_us[split & cov].sum()/1e3:.1f} ms")
print(f"walked tiles: starts p50 {np.percentile(s_us[w],50):.0f} p90 {np.percentile(s_us[w],90):.0f} p99 {np.percentile(s_us[w],99):.0f} max {s_us[w].max():.0f} us; ends max {end[w].max():.0f} us")
if False: print(f"skipped tiles: starts p1 {np.percentile(s_us[ok & ~cov],1):.0f} p50 {np.percentile(s_us[ok & ~cov],50):.0f} p99 {np.percentile(s_us[ok & ~cov],99):.0f} max {s_us[ok & ~cov].max():.0f} us")
# in flight over time
edges = np.linspace(0, max(end[w].max(), 1), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    print(f"  t={mid:8.0f} us: walked tiles in flight {int(((s_us[w] <= mid) & (end[w] > mid)).sum())}")
import ctypes as C
slots = np.zeros(tx * ty, dtype=np.uint32)
ctx._lib.hvrt_debug_tile_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert ctx._lib.hvrt_debug_tile_slots(ctx._h, slots.ctypes.data_as(C.c_void_p), slots.size) == 0
sl = (slots & 0x0FFFFFFF).astype(np.int64)
for lo in range(0, 8192, 512):
    m = w & (sl >= lo) & (sl < lo + 512)
    if m.any(): print(f"  slots {lo:5d}..{lo+511:5d}: {int(m.sum()):4d} walked tiles, start p10 {np.percentile(s_us[m],10):4.0f} p50 {np.percentile(s_us[m],50):4.0f} p90 {np.percentile(s_us[m],90):4.0f}, duration p50 {np.percentile(d_us[m],50):4.0f} max {d_us[m].max():4.0f}")
lend = np.argsort(-(end * w))[:16]
print("latest walked ends: (tile) start us, duration us (launch before), end")
for i in lend: print(f"   ({i % tx:4d},{i // tx:4d}) {s_us[i]:8.0f} {d_us[i]:8.0f} {end[i]:8.0f}  slot {sl[i]:6d} xcd {(slots[i] >> 28) & 7} own {slots[i] >> 31}")
late = np.argsort(-(s_us * w))[:8]
print("latest walked starts: (tile x, y) start us, duration us of the launch before, slot, xcd, own")
for i in late:
    print(f"   ({i % tx:4d},{i // tx:4d}) {s_us[i]:8.0f} {d_us[i]:8.0f}  slot {slots[i] & 0x0FFFFFFF:7d} xcd {(slots[i] >> 28) & 7} own {slots[i] >> 31}")
sl = (slots & 0x0FFFFFFF).astype(np.int64)
print("walked tiles by slot range: slots < 8192:", int((w & (sl < 8192)).sum()), " 8192..40000:", int((w & (sl >= 8192) & (sl < 40000)).sum()), " beyond:", int((w & (sl >= 40000)).sum()))
b = w & (sl >= 40000)
if b.any(): print("  walked tiles in slots beyond 40000: durations before us p50", np.percentile(d_us[b], 50), "max", d_us[b].max(), "starts p50", np.percentile(s_us[b], 50))
