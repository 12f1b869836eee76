"""End-to-end TraceImage (the drop-in entry, C++ host mirror) by phase, from trace_image's own --phases lines:
   (a) the reference's defaults: small STL fixture, 50^3, 800x600, PPM;  (b) 1.0 M-triangle binary STL -> 256^3 -> 4K PPM and PNG.
Each as one process running the call three times (--repeat 3): the first call pays the process's start-up, the later ones
take their context from the pool -- what the reference's main() (a list of models, RayTracer.cpp:20-40) would see.
With HVRT_E2E_DEVICES=0,0,0,0 the same over a device group (copy transport on one GPU)."""
import json, os, sys, subprocess, time, tempfile
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
if os.environ.get("HVRT_E2E_DEVICES"):
    env.update(HVRT_DEVICES=os.environ["HVRT_E2E_DEVICES"], HVRT_GROUP_MIN_PIXELS="1")
small = os.path.join(ROOT, "tests", "golden", "fixture_binary.stl")
cases = [("defaults 800x600 ppm", [small, "a.ppm"]), ("1M tris 4K ppm", [stl, "b.ppm", "--size", "3840x2160", "--grid", "256"]),
         ("1M tris 4K png", [stl, "c.png", "--size", "3840x2160", "--grid", "256"])]
out = {}
for name, args in cases:
    ph = os.path.join(d, "ph.jsonl")
    t0 = time.time()
    p = subprocess.run([exe, "render", *args, "--outdir", d + "/", "--repeat", "3", "--phases", ph, "--quiet"], env=env, capture_output=True, text=True)
    wall = time.time() - t0
    calls = [json.loads(l) for l in open(ph)] if os.path.exists(ph) else []
    out[name] = {"process_wall_s": wall, "calls": calls}
    print(f"{name}: rc {p.returncode}, process {wall:.2f} s")
    for c in calls:
        print("   call %d: total %8.2f ms | context %7.2f | read %7.2f | upload+build %7.2f | render+copy %7.2f | save %7.2f" %
              (c["call"], c["total_ms"], c["context_ms"], c["load_ms"], c["build_ms"], c["render_ms"], c["save_ms"]))
print(json.dumps(out))
