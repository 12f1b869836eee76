"""Times hvrt_build_grid on big synthetic meshes (config 5: 50M triangles into 1024^3; config 3: skewed 10M into 512^3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
t0 = time.time()
if which == "c5":
    tris, g = meshgen.cube_sphere(2041), 1024
elif which == "c3":
    tris, g = meshgen.skewed_statue(289, 2100), 512
elif which == "c2":
    tris, g = meshgen.cube_sphere(289), 256
print(f"{which}: mesh {tris.shape[0]} triangles generated in {time.time()-t0:.1f}s", flush=True)
ctx = ra.Context(0)
defer = len(sys.argv) > 2 and sys.argv[2] == "defer"
ctx.set_deferred_accel(defer)
t0 = time.time(); ctx.set_mesh(tris); print(f"upload {time.time()-t0:.2f}s", flush=True)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
for rep in range(4):
    info = ctx.build_grid(g, g, g, mn, mx, vs)
    tm = ctx.timings()
    if defer:
        csr_total = tm.build_total_ms
        ctx.finish_accel()
        tm2 = ctx.timings()
        print(f"  deferred: CSR alone {csr_total:.3f} ms, derived data afterwards {tm2.build_expand_ms:.3f} ms")
    nv = g ** 3
    b = 72.0 * info.n_triangles + 12.0 * nv + 20.0 * info.n_refs
    print(f"  build total {tm.build_total_ms:.3f} ms (count {tm.build_count_ms:.3f} scan {tm.build_scan_ms:.3f} fill {tm.build_fill_ms:.3f} "
          f"sort {tm.build_sort_ms:.3f} expand {tm.build_expand_ms:.3f}) refs {info.n_refs} occ {info.n_occupied} maxlist {info.max_list} "
          f"-> B_build {b/1e9:.2f} GB = {b/tm.build_total_ms/1e6:.0f} GB/s", flush=True)
ctx.close()
