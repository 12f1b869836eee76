"""Developer check on a GPU box: HIP path vs oracle on small + C2-sized scenes (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import meshgen, capi
import oracle

def run(name, tris, g, W, H, check_render=True, threads=16):
    print(f"== {name}: {tris.shape[0]} tris, grid {g}^3, {W}x{H}", flush=True)
    ctx = ra.Context(0)
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    omn, omx = oracle.mesh_bounds(tris[:, :9].reshape(-1, 3))
    assert (mn.view(np.uint32) == omn.view(np.uint32)).all() and (mx.view(np.uint32) == omx.view(np.uint32)).all(), (mn, omn, mx, omx)
    vs = capi.voxel_size(mn, mx, g, g, g)
    assert vs == oracle.voxel_size(mn, mx, g, g, g)
    for rep in range(3):
        info = ctx.build_grid(g, g, g, mn, mx, vs)
        tm = ctx.timings()
        print(f"  build: total {tm.build_total_ms:.3f} ms (count {tm.build_count_ms:.3f} scan {tm.build_scan_ms:.3f} fill {tm.build_fill_ms:.3f} sort {tm.build_sort_ms:.3f}) refs {info.n_refs} occ {info.n_occupied} maxlist {info.max_list}", flush=True)
    off, idx = ctx.get_csr()
    t0 = time.time(); ooff, oidx = oracle.voxelize(tris, g, g, g, mn, vs); print(f"  oracle voxelize {time.time()-t0:.2f}s", flush=True)
    assert np.array_equal(off, ooff), "offsets differ"
    assert np.array_equal(idx, oidx), "indices differ"
    print("  CSR bit-exact", flush=True)
    pos, look, up, fov, asp = capi.default_camera(mn, mx, "front", W / H)
    fr = capi.make_camera_frame(pos, look, up, fov, asp)
    ofr = oracle.camera_frame(pos, look, up, fov, asp)
    assert bytes(fr) == bytes(ofr)
    for rep in range(3):
        rgb, hit, t, cnt = ctx.render(fr, W, H)
        tm = ctx.timings()
        print(f"  render: {tm.render_ms:.3f} ms -> {W*H/tm.render_ms/1e3:.1f} Mrays/s; V/ray {cnt[0]/(W*H):.1f} T/ray {cnt[1]/(W*H):.1f} hits {(hit!=0xffffffff).sum()}", flush=True)
    if check_render:
        sc = oracle.OracleScene(tris, g, g, g, mn, vs, ooff, oidx)
        t0 = time.time(); orgb, ohit, ot, ocnt = sc.render(ofr, W, H, threads=threads); print(f"  oracle render {time.time()-t0:.2f}s", flush=True)
        print("  hit mismatches", (hit != ohit).sum(), "t mismatches", (t.view(np.uint32) != ot.view(np.uint32)).sum(),
              "rgb mismatches", (rgb.view(np.uint32) != orgb.view(np.uint32)).sum(), "counters", cnt, ocnt, flush=True)
        assert np.array_equal(hit, ohit) and np.array_equal(t.view(np.uint32), ot.view(np.uint32)) and np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)) and np.array_equal(cnt, ocnt)
        print("  render bit-exact", flush=True)
    ctx.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which in ("small", "all"):
        run("cube-sphere 20", meshgen.cube_sphere(20), 16, 160, 120)
        run("cube-sphere 58 (40k)", meshgen.cube_sphere(58), 50, 800, 600)
        run("slivers", meshgen.slivers(300, 3), 32, 320, 240)
        run("skewed", meshgen.skewed_statue(30, 200), 32, 320, 240)
    if which in ("c2", "all"):
        run("C2 cube-sphere 289 (1.0M) 256^3 4K", meshgen.cube_sphere(289), 256, 3840, 2160, check_render=False)
