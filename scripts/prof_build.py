"""Short profiling target for the voxel build: N builds of one workload's grid (rocprofv3 --pmc / --kernel-trace).
usage: prof_build.py c5|c2|c3 [n_builds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tris, g = {"c5": (lambda: meshgen.cube_sphere(2041), 1024), "c3": (lambda: meshgen.skewed_statue(913, 1100), 512),
           "c2": (lambda: meshgen.cube_sphere(289), 256)}[which]
tris = tris()
ctx = ra.Context(0)
ctx.set_mesh(tris)
mn, mx = ctx.mesh_bounds()
vs = capi.voxel_size(mn, mx, g, g, g)
for _ in range(n):
    info = ctx.build_grid(g, g, g, mn, mx, vs)
tm = ctx.timings()
print("builds", n, "refs", info.n_refs, "build_ms", tm.build_total_ms, flush=True)
ctx.close()
