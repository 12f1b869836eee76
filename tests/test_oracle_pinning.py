"""Pins the CPU restatement (oracle/cpu_ref.c) before anything trusts it.

1. Against golden vectors produced by the REAL reference functions (tests/golden/*.npz, made by
   tests/golden/make_golden.py from oracle/_ref/libref.so): Moller-Trumbore, triangle min/max, Lambert
   shader, vector math -- bit for bit.
2. Live against oracle/_ref/libref.so where it exists (authoring container and, because the built .so
   travels, the GPU box) on fresh random inputs.
3. Against the reference run outputs recorded in SURVEY.md Appendix C / section 6 (the voxelizer and the
   DDA loop cannot be compiled here without spdlog, so these recorded numbers are their only pin).
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle
from raytracer_amd import meshgen

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(a):
    return C.c_void_p(a.ctypes.data)


def orc_mt(tris, dirs, orgs):
    l = oracle.lib()
    n = tris.shape[0]
    hit = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    for i in range(n):
        tt = C.c_float(0)
        hit[i] = l.orc_ray_triangle(C.c_void_p(tris.ctypes.data + 48 * i), C.c_void_p(dirs.ctypes.data + 12 * i),
                                    C.c_void_p(orgs.ctypes.data + 12 * i), C.byref(tt))
        t[i] = tt.value if hit[i] else 0.0
    return hit, t


def test_moller_trumbore_golden():
    z = np.load(os.path.join(G, "mt_vectors.npz"))
    tris, dirs, orgs = (np.ascontiguousarray(z[k]) for k in ("tris", "dirs", "orgs"))
    hit, t = orc_mt(tris, dirs, orgs)
    assert np.array_equal(hit, z["hit"])
    assert np.array_equal(t.view(np.uint32), z["t_bits"])
    assert 0 < hit.sum() < hit.size


def test_triangle_min_max_golden():
    z = np.load(os.path.join(G, "minmax_vectors.npz"))
    tris = np.ascontiguousarray(z["tris"])
    mn = np.zeros((tris.shape[0], 3), np.float32)
    mx = np.zeros_like(mn)
    for i in range(tris.shape[0]):
        oracle.lib().orc_triangle_min_max(C.c_void_p(tris.ctypes.data + 48 * i), C.c_void_p(mn.ctypes.data + 12 * i),
                                          C.c_void_p(mx.ctypes.data + 12 * i))
    assert np.array_equal(mn.view(np.uint32), z["mn_bits"])
    assert np.array_equal(mx.view(np.uint32), z["mx_bits"])


def test_lambert_golden():
    z = np.load(os.path.join(G, "lambert_vectors.npz"))
    tris, og = np.ascontiguousarray(z["tris"]), np.ascontiguousarray(z["origins"])
    out = np.zeros((tris.shape[0], 3), np.float32)
    h = np.zeros(1, oracle.HIT_DTYPE)
    for i in range(tris.shape[0]):
        h["hit"] = 1
        h["point"] = tris[i, :3]
        h["origin"] = og[i]
        h["normal"] = tris[i, 9:]
        oracle.lib().orc_shade(None, _p(h), C.c_float(0.5), C.c_void_p(out.ctypes.data + 12 * i))
    assert np.array_equal(out.view(np.uint32), z["rgb_bits"])


def test_vector_math_golden():
    # normalize / cross / dot are static inside cpu_ref.c; they are exercised through the camera frame:
    # here the golden vectors pin numpy restatements of the same expressions, which test_camera_frame
    # then compares with orc_camera_frame_make.
    z = np.load(os.path.join(G, "vecmath_vectors.npz"))
    a, b = z["a"], z["b"]
    f = np.float32
    dot = ((a[:, 0] * b[:, 0]).astype(f) + (a[:, 1] * b[:, 1]).astype(f)).astype(f) + (a[:, 2] * b[:, 2]).astype(f)
    assert np.array_equal(dot.astype(f).view(np.uint32), z["dot_bits"])
    crs = np.stack([(a[:, 1] * b[:, 2]).astype(f) - (a[:, 2] * b[:, 1]).astype(f),
                    (a[:, 2] * b[:, 0]).astype(f) - (a[:, 0] * b[:, 2]).astype(f),
                    (a[:, 0] * b[:, 1]).astype(f) - (a[:, 1] * b[:, 0]).astype(f)], 1).astype(f)
    assert np.array_equal(crs.view(np.uint32), z["cross_bits"])
    l2 = ((a[:, 0] * a[:, 0]).astype(f) + (a[:, 1] * a[:, 1]).astype(f)).astype(f) + (a[:, 2] * a[:, 2]).astype(f)
    ln = np.sqrt(l2.astype(f)).astype(f)
    with np.errstate(all="ignore"):
        nrm = np.where((ln > 0)[:, None], (a / ln[:, None]).astype(f), a)
    assert np.array_equal(nrm.astype(f).view(np.uint32), z["normalize_bits"])


def _np_frame(pos, la, up, fov, aspect):
    f = np.float32

    def norm(v):
        ln = np.sqrt(f(f(f(v[0] * v[0]) + f(v[1] * v[1])) + f(v[2] * v[2])))
        return (v / ln).astype(f) if ln > 0 else v

    def cross(a, b):
        return np.array([f(a[1] * b[2]) - f(a[2] * b[1]), f(a[2] * b[0]) - f(a[0] * b[2]),
                         f(a[0] * b[1]) - f(a[1] * b[0])], f)

    pos, la, up = (np.asarray(x, f) for x in (pos, la, up))
    fov_rad = f(f(f(fov) * f(3.14159265)) / f(180.0))
    hh = f(math.tan(float(f(fov_rad / f(2.0)))))   # glibc tanf is correctly rounded from double tan here
    hw = f(f(aspect) * hh)
    fwd = norm((la - pos).astype(f))
    right = norm(cross(fwd, up))
    upv = cross(right, fwd)
    hor = (right * f(f(2.0) * hw)).astype(f)
    ver = (upv * f(f(2.0) * hh)).astype(f)
    ll = (((pos - (hor * f(0.5)).astype(f)).astype(f) - (ver * f(0.5)).astype(f)).astype(f) + fwd).astype(f)
    return np.concatenate([pos, ll, hor, ver]).astype(f)


@pytest.mark.parametrize("view", ["front", "side", "top"])
def test_camera_frame_restatement(view):
    from raytracer_amd import capi
    pos, la, up, fov, asp = capi.default_camera([-1, -0.5, -2], [1, 0.7, 2.5], view)
    fr = oracle.frame_to_array(oracle.camera_frame(pos, la, up, fov, asp))
    ref = _np_frame(pos, la, up, fov, asp)
    # tanf vs double tan may differ in the last bit on other libms; everything else must be exact
    assert np.allclose(fr, ref, rtol=2e-7, atol=0)
    assert np.array_equal(fr[:3], ref[:3])


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libref.so not built (needs /root/reference)")
def test_moller_trumbore_live_vs_reference():
    rng = np.random.default_rng(7)
    n = 20000
    v = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    w = rng.dirichlet([1, 1, 1], n)
    w = np.where(rng.random((n, 1)) < 0.6, w, rng.uniform(-0.2, 1.2, (n, 3)))
    target = (w[:, :, None] * v).sum(1) / w.sum(1, keepdims=True)
    o = rng.uniform(-3, 3, (n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tris = np.concatenate([v.reshape(n, 9), np.zeros((n, 3), np.float32)], 1).astype(np.float32)
    dirs, orgs = d.astype(np.float32), o.astype(np.float32)
    rh, rt = oracle.ref_ray_triangle_batch(tris, dirs, orgs)
    oh, ot = orc_mt(tris, dirs, orgs)
    rt = np.where(rh == 1, rt, 0).astype(np.float32)
    assert np.array_equal(rh, oh)
    assert np.array_equal(rt.view(np.uint32), ot.view(np.uint32))


# ---------------------------------------------------------------- SURVEY.md Appendix C (recorded reference run)

RECORDED = {
    # outputs of the real reference (compiled in the survey session) on the 100 x 200 sin/cos
    # sphere, 50^3 grid, 800 x 600, default front camera
    "appendix_c": dict(nlat=100, nlon=200, grid=50, W=800, H=600, tris=39800, verts=20101, refs=144331, occupied=14960,
                       max_list=504, hits=35137, V=37.9, T=111.1),
    # SURVEY.md section 6, second recorded run: 708 x 708 sphere, 256^3 grid, 384 x 216, same camera rule
    "section_6": dict(nlat=708, nlon=708, grid=256, W=384, H=216, tris=1001820, refs=3617504, V=192.6, T=226.4),
}


def recorded_mesh(which):
    """The mesh of a recorded reference run, rebuilt from the survey machine's sin/cos tables (committed as
    tests/golden/uv_sphere_trig.npz by make_golden.py): no dependence on this machine's libm."""
    r = RECORDED[which]
    z = np.load(os.path.join(G, "uv_sphere_trig.npz"))
    key = f"{r['nlat']}x{r['nlon']}"
    trig = tuple(z[f"{k}_{key}"] for k in ("sin_th", "cos_th", "cos_ph", "sin_ph"))
    return meshgen.stl_reader_view(meshgen.uv_sphere_soup(r["nlat"], r["nlon"], 1.0, trig))


def test_survey_appendix_c_recorded_reference_outputs():
    """Mesh, grid and camera of SURVEY.md Appendix C (39 800-tri sin/cos sphere, 50^3, 800x600, front camera).
    Expected values are the real reference's outputs recorded there.  This is the pin of the two functions that
    cannot be compiled here (voxelizeTrianglesDirect, SceneCreator.cpp:197-281; traverseVoxelGrid and its helpers,
    OctreeLikeRayTracer.cpp:155-336): reference-held values vs the oracle.  tests/test_gpu_recorded_runs.py holds
    the same values against the HIP path, with no oracle in between."""
    r = RECORDED["appendix_c"]
    tris, verts = recorded_mesh("appendix_c")
    assert tris.shape[0] == r["tris"] and verts.shape[0] == r["verts"]      # "39 800 kept, 20 101 unique vertices"
    sc, mn, mx, vs = oracle.build_scene(tris, 50, 50, 50, points=verts)
    cnt = np.diff(sc.offsets.astype(np.int64))
    assert sc.indices.size == 144331                 # "CSR: refs 144 331"
    assert int((cnt > 0).sum()) == 14960             # "occupied 14 960 / 125 000"
    assert int(cnt.max()) == 504                     # "longest list 504 (pole voxels)"
    # strictly ascending lists, x-fastest exclusive scan (F3)
    for v in np.flatnonzero(cnt > 1)[:2000]:
        seg = sc.indices[sc.offsets[v]:sc.offsets[v + 1]]
        assert (np.diff(seg.astype(np.int64)) > 0).all()
    from raytracer_amd import capi
    pos, la, up, fov, asp = capi.default_camera(mn, mx, "front")
    fr = oracle.camera_frame(pos, la, up, fov, asp)
    W, H = 800, 600
    rgb, hit, t, c = sc.render(fr, W, H, threads=8)
    miss = 0xFFFFFFFF
    assert int((hit != miss).sum()) == 35137         # "hits 35 137 / 480 000 (7.32 %)"
    assert abs(c[0] / (W * H) - 37.9) < 0.05         # "37.9 voxels/ray"
    assert abs(c[1] / (W * H) - 111.1) < 0.05        # "111.1 tri-tests/ray"
    assert hit[320, 400] == 18100                    # px(400,320) -> hit tri 18100
    assert hit[320, 401] == 18099                    # px(401,320) -> hit tri 18099
    assert hit[300, 410] == miss                     # px(410,300) dir.y=+0 -> MISS inside the silhouette (F7)
    assert hit[300, 300] == miss                     # px(300,300) likewise
    assert hit[300, 400] != miss                     # px(400,300) hits
    assert abs(float(rgb[300, 400, 0]) - 1.09975) < 5e-6   # "probe centre pixel = 1.09975"
    # per-pixel probes: voxels visited / tests (Appendix C lines)
    o = np.array([fr.origin.x, fr.origin.y, fr.origin.z], np.float32)

    def probe(x, y):
        d = np.zeros(3, np.float32)
        oracle.lib().orc_generate_ray(C.byref(fr), C.c_float(np.float32(x) / np.float32(W)),
                                      C.c_float(np.float32(y) / np.float32(H)), _p(d))
        out, cc = sc.trace(o[None], d[None])
        return out[0], cc

    h, cc = probe(400, 300)
    assert h["hit"] == 1 and cc[0] == 1 and cc[1] == 18      # "hit, 1 voxel, 18 tests"
    h, cc = probe(400, 320)
    assert h["triangle_index"] == 18100 and cc[0] == 4       # "4 voxels"
    h, cc = probe(401, 320)
    assert h["triangle_index"] == 18099 and cc[0] == 5       # "5 voxels (spurious first X step)"
    h, cc = probe(410, 300)
    assert h["hit"] == 0 and cc[0] == 50                     # "MISS ..., 50 voxels (X never steps)"


def test_survey_section_6_second_recorded_reference_run():
    """the reference on the 708 x 708 sphere (1 001 820 triangles after the STL reader) in a
    256^3 grid: 3 617 504 references; 384 x 216 default front camera (aspect 800/600 as setupCamera hard-codes,
    TraceImages.cpp:143-144): 192.6 voxels and 226.4 triangle tests per ray."""
    r = RECORDED["section_6"]
    tris, verts = recorded_mesh("section_6")
    assert tris.shape[0] == r["tris"]
    sc, mn, mx, vs = oracle.build_scene(tris, 256, 256, 256, points=verts)
    assert sc.indices.size == r["refs"]
    from raytracer_amd import capi
    fr = oracle.camera_frame(*capi.default_camera(mn, mx, "front"))
    W, H = r["W"], r["H"]
    rgb, hit, t, c = sc.render(fr, W, H, threads=8)
    assert abs(c[0] / (W * H) - r["V"]) < 0.05
    assert abs(c[1] / (W * H) - r["T"]) < 0.05


def test_ray_box_golden():
    """rayBoxIntersection (GeometryUtils.cpp:15-37), the entry test of the nearest-hit mode: decision and both
    distances bit for bit against the real function's outputs (zero / NaN / inf direction components included)."""
    z = np.load(os.path.join(G, "raybox_vectors.npz"))
    o, d, bmin, bmax = z["orgs"], z["dirs"], z["bmin"], z["bmax"]
    l = oracle.lib()
    tn, tf = np.zeros(1, np.float32), np.zeros(1, np.float32)
    for i in range(o.shape[0]):
        hit = l.orc_ray_box(C.c_void_p(o.ctypes.data + 12 * i), C.c_void_p(d.ctypes.data + 12 * i), _p(bmin), _p(bmax),
                            _p(tn), _p(tf))
        assert hit == z["hit"][i], i
        assert tn.view(np.uint32)[0] == z["t_near_bits"][i] and tf.view(np.uint32)[0] == z["t_far_bits"][i], i


def test_nearest_hit_mode_is_the_brute_force_nearest_hit():
    """The oracle's mode 1 (extension; the reference has no such mode -- parity unpinned): for
    every ray, the triangle and t it reports are those of the smallest t over ALL triangles of the mesh
    (Moller-Trumbore as the reference computes it), on a closed mesh seen from outside and from inside."""
    tris = meshgen.cube_sphere(12)
    sc, mn, mx, vs = oracle.build_scene(tris, 20, 20, 20)
    sc.set_mode(1, mx)
    rng = np.random.default_rng(3)
    n = 1500
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    o[:200] = rng.uniform(-0.5, 0.5, (200, 3))                   # inside the sphere
    tgt = rng.normal(size=(n, 3)) * 0.6
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    got, _ = sc.trace(o, d)
    T = tris.shape[0]
    n_hit = 0
    for i in range(n):
        hit, t = orc_mt(tris, np.repeat(d[i:i + 1], T, 0), np.repeat(o[i:i + 1], T, 0))
        if hit.any():
            tt = np.where(hit == 1, t, np.float32(np.inf))
            best = float(tt.min())
            assert got["hit"][i] == 1 and got["t"][i] == np.float32(best), i
            assert tt[got["triangle_index"][i]] == np.float32(best), i
            n_hit += 1
        else:
            assert got["hit"][i] == 0, i
    assert n_hit > n // 3
    # on a grid so coarse that front and back faces share voxels the reference's rule ("first voxel with a hit,
    # last entry that hits") is visibly not nearest-hit, while mode 1 still is
    sc2, mn2, mx2, vs2 = oracle.build_scene(tris, 2, 2, 2)
    ref0, _ = sc2.trace(o, d)
    sc2.set_mode(1, mx2)
    near, _ = sc2.trace(o, d)
    assert (ref0["t"] != near["t"]).any()
    assert np.array_equal(near["t"].view(np.uint32), got["t"].view(np.uint32))      # grid-independent
