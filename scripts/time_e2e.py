"""End-to-end TraceImage (the drop-in entry, C++ host mirror): 1.0 M-triangle binary STL -> 4K PNG, wall clock by stage."""
import os, sys, subprocess, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import meshgen
d = tempfile.mkdtemp()
stl = os.path.join(d, "sphere.stl")
t0 = time.time()
meshgen.write_binary_stl(stl, meshgen.cube_sphere(289)[:, :9].reshape(-1, 3, 3))
print(f"wrote {os.path.getsize(stl)/1e6:.0f} MB STL in {time.time()-t0:.1f}s", flush=True)
exe = os.path.join(ROOT, "raytracer_amd", "lib", "trace_image")
env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raytracer_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
for out in ("a.png", "b.ppm"):
    t0 = time.time()
    p = subprocess.run([exe, "render", stl, out, "--outdir", d + "/", "--size", "3840x2160", "--grid", "256"], env=env,
                       capture_output=True, text=True)
    print(f"{out}: rc {p.returncode}, {time.time()-t0:.2f}s wall")
    print("\n".join(l for l in (p.stdout + p.stderr).splitlines() if any(k in l for k in ("Loaded", "loaded", " in ", "ms", "Success"))))
