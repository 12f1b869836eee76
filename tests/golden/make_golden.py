"""Generates tests/golden/*.npz from the REAL reference functions (oracle/_ref/libref.so, compiled from
/root/reference/RayTracer/{GeometryUtils.cpp,Shaders.cpp,STLReader.h,VectorMath.h} by `make -C oracle ref`).

Run in the authoring container only:  python tests/golden/make_golden.py
The .npz files hold inputs and the reference's outputs (data, not code); they are what pins the
oracle on machines where the reference sources do not exist (the GPU box).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from raytracer_amd import meshgen  # noqa: E402


def p(a):
    return C.c_void_p(a.ctypes.data)


def mt_cases(rng, n_random=3000):
    tris, dirs, orgs = [], [], []
    # random triangles / rays aimed near the triangle (hits, near misses, edge grazes)
    for _ in range(n_random):
        v = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
        w = rng.dirichlet([1, 1, 1]) if rng.random() < 0.7 else rng.uniform(-0.3, 1.3, 3)
        w = w / w.sum()
        target = (w[:, None] * v).sum(0)
        o = rng.uniform(-3, 3, 3)
        d = target - o
        d = d / np.linalg.norm(d)
        if rng.random() < 0.1:
            d = -d   # behind the origin: t < 0
        tris.append(np.concatenate([v.ravel(), [0, 0, 1]]))
        dirs.append(d)
        orgs.append(o)
    # exact edge / vertex / boundary cases on a unit right triangle (u == 0, v == 0, u + v == 1, eps bands)
    base = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    eps = np.finfo(np.float32).eps
    for (x, y) in [(0, 0), (1, 0), (0, 1), (0.5, 0), (0, 0.5), (0.5, 0.5), (0.25, 0.25), (-0.5, 0.25),
                   (-eps, 0.3), (-2 * eps, 0.3), (-0.5 * eps, 0.3), (1 + eps, 0), (1 + 2 * eps, 0), (0.5 + eps, 0.5),
                   (0.5, 0.5 + 2 * eps), (0.3, -eps), (0.3, -3 * eps), (1e-30, 1e-30), (2, 2)]:
        for oz, dz in [(-1, 1), (1, -1), (-1, -1), (0, 1), (-eps, 1), (-2 * eps, 1), (-1e-3, 1)]:
            tris.append(base)
            dirs.append([0, 0, dz])
            orgs.append([x, y, oz])
    # parallel rays, degenerate triangles, tiny determinants, huge coordinates, NaN / inf
    specials = [
        (base, [1, 0, 0], [0, 0.2, 0]), (base, [1, 0, 0], [-1, 0.2, 0.0]), (base, [0, 1, 0], [0.2, -1, 1e-8]),
        ([0, 0, 0, 1, 1, 1, 2, 2, 2, 0, 0, 0], [0, 0, 1], [0.5, 0.5, -1]),       # collinear
        ([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 1], [0, 0, -1]),           # point
        ([0, 0, 0, 1e-4, 0, 0, 0, 1e-4, 0, 0, 0, 1], [0, 0, 1], [2e-5, 2e-5, -1]),  # det ~ 1e-8 < eps
        ([0, 0, 0, 1e-3, 0, 0, 0, 1e-3, 0, 0, 0, 1], [0, 0, 1], [2e-4, 2e-4, -1]),  # det ~ 1e-6 > eps
        ([0, 0, 0, 1e18, 0, 0, 0, 1e18, 0, 0, 0, 1], [0, 0, 1], [1e17, 1e17, -1]),  # overflow to inf
        (base, [0, 0, np.nan], [0.2, 0.2, -1]), (base, [0, 0, 1], [np.nan, 0.2, -1]),
        (base, [0, 0, np.inf], [0.2, 0.2, -1]), (base, [0, 0, 1], [0.2, 0.2, -np.inf]),
        ([np.nan, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], [0, 0, 1], [0.2, 0.2, -1]),
        (base, [0, 0, 0], [0.2, 0.2, -1]), (base, [0.0, -0.0, 1], [0.2, 0.2, -1]),
    ]
    for t, d, o in specials:
        tris.append(np.asarray(t, np.float32))
        dirs.append(d)
        orgs.append(o)
    return (np.asarray(tris, np.float32), np.asarray(dirs, np.float32), np.asarray(orgs, np.float32))


def ray_box_vectors():
    """rayBoxIntersection (GeometryUtils.cpp:15-37): random rays against a box, rays inside it, axis-parallel
    rays (zero components: +-inf reciprocals, 0 * inf = NaN on a face), NaN / inf inputs."""
    r = oracle.ref()
    rng = np.random.default_rng(20261111)
    n = 3000
    bmin = np.array([-1.0, -0.5, 0.25], np.float32)
    bmax = np.array([1.0, 1.5, 2.0], np.float32)
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o[:300] = rng.uniform(bmin, bmax, (300, 3)).astype(np.float32)             # origins inside
    aim = rng.random(n) < 0.5
    tgt = rng.uniform(bmin - 0.2, bmax + 0.2, (n, 3))
    dd = tgt - o
    dd = (dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-9)).astype(np.float32)
    d[aim] = dd[aim]
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0], [0.0, -0.0, 1.0],
                     [np.nan, 0, 1], [0, np.inf, 0], [1, 1, 0], [0, 1, 1]], np.float32)
    k = 300
    for a in axes:
        d[k:k + 40] = a
        k += 40
    o[k:k + 20] = bmin                                                         # on the min corner
    o[k + 20:k + 40, 0] = bmin[0]                                              # on a face, sliding along it
    d[k + 20:k + 40, 0] = 0.0
    o[k + 40:k + 44] = [np.nan, 0, 0]
    o[k + 44:k + 48] = [np.inf, 0, 0]
    hit = np.zeros(n, np.int32)
    tn = np.zeros(n, np.float32)
    tf = np.zeros(n, np.float32)
    r.ref_ray_box_batch(p(o), p(d), p(bmin), p(bmax), C.c_uint(n), p(hit), p(tn), p(tf))
    np.savez_compressed(os.path.join(HERE, "raybox_vectors.npz"), orgs=o, dirs=d, bmin=bmin, bmax=bmax, hit=hit,
                        t_near_bits=tn.view(np.uint32), t_far_bits=tf.view(np.uint32))
    print("raybox_vectors", n, "cases,", int(hit.sum()), "hits")


def uv_sphere_trig_tables():
    """uv_sphere_trig.npz: the sin / cos tables (float64, this container's libm -- the one the survey session's
    reference runs used) of the two sin/cos spheres whose REFERENCE outputs SURVEY.md records (Appendix C: 100 x 200
    -> 39 800 triangles; section 6: 708 x 708 -> 1 001 820 triangles).  With them meshgen.uv_sphere_soup rebuilds
    those meshes bit for bit on any machine, so the recorded-run tests never depend on the local libm.  Written only
    if the meshes they give reproduce the recorded triangle / vertex counts."""
    tabs = {}
    for name, (a, b) in {"100x200": (100, 200), "708x708": (708, 708)}.items():
        t = meshgen.uv_sphere_trig_tables(a, b)
        for k, v in zip(("sin_th", "cos_th", "cos_ph", "sin_ph"), t):
            tabs[f"{k}_{name}"] = v
    tris, verts = meshgen.stl_reader_view(meshgen.uv_sphere_soup(100, 200, 1.0, tuple(tabs[f"{k}_100x200"] for k in ("sin_th", "cos_th", "cos_ph", "sin_ph"))))
    assert tris.shape[0] == 39800 and verts.shape[0] == 20101, "this libm does not give the survey's Appendix C sphere"
    np.savez_compressed(os.path.join(HERE, "uv_sphere_trig.npz"), **tabs)
    print("uv_sphere_trig.npz:", {k: v.shape for k, v in tabs.items()})


def main():
    uv_sphere_trig_tables()
    if not oracle.have_ref():
        raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle ref` where /root/reference exists")
    if "--only-raybox" in sys.argv:       # added after the other files were committed; leaves them untouched
        ray_box_vectors()
        return
    r = oracle.ref()
    rng = np.random.default_rng(20261110)

    # ---- Moller-Trumbore (GeometryUtils.cpp:57-97)
    tris, dirs, orgs = mt_cases(rng)
    hit, t = oracle.ref_ray_triangle_batch(tris, dirs, orgs)
    t = np.where(hit == 1, t, np.float32(0)).astype(np.float32)   # t is unspecified on a miss
    np.savez_compressed(os.path.join(HERE, "mt_vectors.npz"), tris=tris, dirs=dirs, orgs=orgs, hit=hit,
                        t_bits=t.view(np.uint32))
    print("mt_vectors", tris.shape[0], "cases,", int(hit.sum()), "hits")

    # ---- triangle min/max (GeometryUtils.cpp:43-51), NaN / signed-zero order cases included
    tm = rng.uniform(-5, 5, (400, 12)).astype(np.float32)
    tm[:20, :9] = rng.choice(np.array([0.0, -0.0, 1.0, np.nan, np.inf, -np.inf], np.float32), (20, 9))
    mn = np.zeros((tm.shape[0], 3), np.float32)
    mx = np.zeros((tm.shape[0], 3), np.float32)
    for i in range(tm.shape[0]):
        r.ref_triangle_min_max(C.c_void_p(tm.ctypes.data + 48 * i), C.c_void_p(mn.ctypes.data + 12 * i),
                               C.c_void_p(mx.ctypes.data + 12 * i))
    np.savez_compressed(os.path.join(HERE, "minmax_vectors.npz"), tris=tm, mn_bits=mn.view(np.uint32),
                        mx_bits=mx.view(np.uint32))

    # ---- Lambert shader as computeShading calls it (Shaders.cpp:4-22, OctreeLikeRayTracer.cpp:30-34)
    ts = rng.uniform(-2, 2, (600, 12)).astype(np.float32)
    ts[:10, 9:] = 0.0                         # zero normal
    ts[10:20, 9:] *= np.float32(1e-25)        # tiny normal (length underflows)
    og = rng.uniform(-6, 6, (600, 3)).astype(np.float32)
    og[20:25] = ts[20:25, :3]                 # origin == v0: zero light vector
    rgb = np.zeros((600, 3), np.float32)
    for i in range(600):
        r.ref_lambert(C.c_void_p(ts.ctypes.data + 48 * i), C.c_void_p(og.ctypes.data + 12 * i),
                      C.c_void_p(rgb.ctypes.data + 12 * i))
    np.savez_compressed(os.path.join(HERE, "lambert_vectors.npz"), tris=ts, origins=og, rgb_bits=rgb.view(np.uint32))

    # ---- vector math (VectorMath.h:9-43)
    a = rng.uniform(-3, 3, (500, 3)).astype(np.float32)
    b = rng.uniform(-3, 3, (500, 3)).astype(np.float32)
    a[:5] = 0
    a[5:8] *= np.float32(1e-30)
    nrm = np.zeros_like(a)
    crs = np.zeros_like(a)
    dt = np.zeros(500, np.float32)
    for i in range(500):
        pa, pb = C.c_void_p(a.ctypes.data + 12 * i), C.c_void_p(b.ctypes.data + 12 * i)
        r.ref_normalize(pa, C.c_void_p(nrm.ctypes.data + 12 * i))
        r.ref_cross(pa, pb, C.c_void_p(crs.ctypes.data + 12 * i))
        dt[i] = r.ref_dot(pa, pb)
    np.savez_compressed(os.path.join(HERE, "vecmath_vectors.npz"), a=a, b=b, normalize_bits=nrm.view(np.uint32),
                        cross_bits=crs.view(np.uint32), dot_bits=dt.view(np.uint32))

    # ---- STL ingest (STLReader.h:245-289, 451-535, 666-727): duplicates, degenerates, binary + ASCII
    soup = meshgen.uv_sphere_soup(6, 8, 1.0)                      # pole degenerates + shared vertices
    extra = np.array([[[0, 0, 0], [0, 0, 0], [1, 1, 1]],          # two equal corners -> dropped
                      [[2, 2, 2], [3, 2, 2], [2, 3, 2]],
                      [[2, 2, 2], [3, 2, 2], [2, 3, 2]]], np.float32)   # exact duplicate triangle -> kept
    soup = np.concatenate([soup, extra]).astype(np.float32)
    normals = meshgen.face_normals(soup)
    for kind in ("binary", "ascii"):
        path = os.path.join(HERE, f"fixture_{kind}.stl")
        if kind == "binary":
            meshgen.write_binary_stl(path, soup, normals)
        else:
            meshgen.write_ascii_stl(path, soup, normals)
        co, ti, nr = oracle.ref_stl_load(path)
        np.savez_compressed(os.path.join(HERE, f"stl_{kind}_expected.npz"), coords_bits=co.view(np.uint32), tris=ti,
                            normals_bits=nr.view(np.uint32))
        print("stl", kind, soup.shape[0], "written ->", ti.shape[0], "kept,", co.shape[0], "vertices")

    ray_box_vectors()


if __name__ == "__main__":
    main()
