"""The separating-axis option of the voxelizer (hvrt_set_overlap_test(ctx, 1)) -- an extension: the reference's rule is the
bounding box (SceneCreator.cpp:197-281; triangleIntersectsBox is declared in GeometryUtils.h:19 and defined nowhere), so
nothing in the reference pins this: PARITY UNPINNED.  What is checked instead:
  * the CSR is the CPU restatement's, bit for bit (oracle.voxelize(..., sat=True)), incl. triangles that span many voxels;
  * every list is a subset of the list the reference's rule builds, and the voxel that holds a point of a triangle lists it;
  * the traversal on such a grid is the oracle's traversal on the same CSR, bit for bit (the "also listed in that face
    neighbour" bits can no longer be read off the triangle's range: they are searched for);
  * in nearest-hit mode the frames are those of a bounding-box grid: same t everywhere, same triangle wherever t is not shared.
"""
import numpy as np
import pytest

import oracle
import raytracer_amd as ra
from raytracer_amd import capi, meshgen

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF

SCENES = {
    "sphere_24": (lambda: meshgen.cube_sphere(24), (32, 32, 32), (240, 180)),
    "offcentre_noncubic": (lambda: meshgen.cube_sphere(20, 0.7, center=(3.0, -2.0, 11.5)), (40, 13, 27), (240, 180)),
    "slivers": (lambda: meshgen.slivers(400, 5), (32, 32, 32), (200, 150)),              # long thin triangles: ranges of hundreds of voxels
    "random_soup": (lambda: meshgen.random_soup(3000, 4), (33, 17, 29), (200, 150)),
    "skewed_long_lists": (lambda: meshgen.skewed_statue(24, 120), (32, 32, 32), (200, 150)),
    "box_on_voxel_planes": (lambda: meshgen.box(), (8, 8, 8), (120, 90)),                # faces exactly on voxel boundaries
    "random_soup_cubic": (lambda: meshgen.random_soup(3000, 4), (24, 24, 24), (200, 150)),
}


@pytest.fixture()
def sat_ctx():
    c = ra.Context(0)
    c.set_overlap_test(1)
    yield c
    c.close()


def _build(c, tris, dims):
    c.set_mesh(tris)
    mn, mx = c.mesh_bounds()
    vs = capi.voxel_size(mn, mx, *dims)
    info = c.build_grid(*dims, mn, mx, vs)
    return info, mn, mx, vs


@pytest.mark.parametrize("name", sorted(SCENES))
def test_sat_csr_is_the_restatements_and_a_subset_of_the_box_rule(sat_ctx, ctx, name):
    gen, dims, _ = SCENES[name]
    tris = np.ascontiguousarray(gen(), np.float32)
    info, mn, mx, vs = _build(sat_ctx, tris, dims)
    off, idx = sat_ctx.get_csr()
    ooff, oidx = oracle.voxelize(tris, *dims, mn, vs, sat=True)
    assert np.array_equal(off, ooff), "SAT CSR offsets differ from the restatement"
    assert np.array_equal(idx, oidx), "SAT CSR indices differ from the restatement"
    cnt = np.diff(off.astype(np.int64))
    assert info.n_refs == idx.size and info.n_occupied == int((cnt > 0).sum()) and info.max_list == int(cnt.max())
    # against the reference's rule, on the other context
    _build(ctx, tris, dims)
    boff, bidx = ctx.get_csr()
    assert idx.size <= bidx.size
    nvox = dims[0] * dims[1] * dims[2]
    vox_s = np.repeat(np.arange(nvox, dtype=np.int64), cnt)
    vox_b = np.repeat(np.arange(nvox, dtype=np.int64), np.diff(boff.astype(np.int64)))
    T = tris.shape[0]
    assert np.isin(vox_s * T + idx, vox_b * T + bidx).all(), "a SAT list holds a triangle the bounding-box list does not"
    # the voxel that holds a point of a triangle lists it (corners nudged inwards, centroid, edge midpoints)
    v = tris[:, :9].reshape(-1, 3, 3).astype(np.float64)
    cen = v.mean(axis=1, keepdims=True)
    pts = np.concatenate([cen, cen + 0.98 * (v - cen), 0.5 * (v + np.roll(v, 1, axis=1))], axis=1)      # (T, 7, 3)
    # (the cell of a point by the voxelizer's own fp32 arithmetic, SceneCreator.cpp:23-37: monotonic, so inside the triangle's range)
    vs32 = np.float32(vs[0] if np.ndim(vs) else vs)
    cell = ((pts.astype(np.float32) - np.asarray(mn, np.float32)) / vs32).astype(np.int64)
    # geometry outside the grid (a non-cubic grid under a round mesh: the reference clamps its indices into the border
    # voxels) overlaps no voxel and is listed nowhere under this rule: points inside the grid only
    inside = ((cell >= 0) & (cell < np.asarray(dims))).all(axis=2)
    cell = np.minimum(np.maximum(cell, 0), np.asarray(dims) - 1)
    flat = cell[..., 0] + dims[0] * (cell[..., 1] + dims[1] * cell[..., 2])
    tri_id = np.broadcast_to(np.arange(T)[:, None], flat.shape)
    keys = (flat * T + tri_id)[inside]
    assert np.isin(keys, vox_s * T + idx).all(), "a voxel that holds a point of a triangle does not list it"


@pytest.mark.parametrize("name", sorted(SCENES))
def test_traversal_on_a_sat_grid_is_the_oracles_on_the_same_csr(sat_ctx, name):
    gen, dims, (W, H) = SCENES[name]
    tris = np.ascontiguousarray(gen(), np.float32)
    info, mn, mx, vs = _build(sat_ctx, tris, dims)
    off, idx = sat_ctx.get_csr()
    sc = oracle.OracleScene(tris, *dims, mn, vs, off, idx, mx=mx)
    for view in ("front", "top"):
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
        for mode in (0, 1):
            sat_ctx.set_mode(mode)
            sc.set_mode(mode, mx)
            exp = sc.render(fr, W, H, threads=8)
            for launch in range(2):             # without and with tile history
                rgb, hit, t, cnt = sat_ctx.render(fr, W, H)
                assert np.array_equal(hit, exp[1]), f"{view} mode {mode}: hit ids differ"
                assert np.array_equal(t.view(np.uint32), exp[2].view(np.uint32))
                assert np.array_equal(rgb.view(np.uint32), exp[0].view(np.uint32))
                assert np.array_equal(cnt, exp[3])
    sat_ctx.set_mode(0)


@pytest.mark.parametrize("name", ["sphere_24", "random_soup_cubic", "skewed_long_lists"])
def test_nearest_hit_frames_do_not_depend_on_the_overlap_rule(sat_ctx, ctx, name):
    """(Cubic grids: the mesh lies inside the grid.  Where it overhangs -- a non-cubic grid -- the reference's rule lists the
    overhang in the border voxels and this one lists it nowhere, so rays can hit it on one grid and not on the other.)"""
    gen, dims, (W, H) = SCENES[name]
    # odd frame sizes: with an even one the middle column / row has a direction component of exactly 0, and on such rays the
    # reference's DDA (kept as it is: SURVEY F7, inf / NaN tMax) leaves the geometric voxel sequence -- what it still hits then
    # depends on which neighbours happen to list a triangle
    W, H = W + 1, H + 1
    tris = np.ascontiguousarray(gen(), np.float32)
    info_s, mn, mx, vs = _build(sat_ctx, tris, dims)
    info_b, _, _, _ = _build(ctx, tris, dims)
    assert info_s.n_refs < info_b.n_refs
    try:
        sat_ctx.set_mode(1)
        ctx.set_mode(1)
        for view in ("front", "side"):
            fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
            rs, hs, ts, cs = sat_ctx.render(fr, W, H)
            rb, hb, tb, cb = ctx.render(fr, W, H)
            assert np.array_equal(ts.view(np.uint32), tb.view(np.uint32)), "nearest-hit t differs between the two grids"
            assert np.array_equal(hs == MISS, hb == MISS)
            same = int((hs == hb).sum())
            assert same >= hs.size * 0.9999, f"hit ids equal on only {same}/{hs.size}"         # (a t shared by two triangles: an edge)
            assert cs[1] < cb[1], "fewer triangle tests on the SAT grid"
    finally:
        sat_ctx.set_mode(0)
        ctx.set_mode(0)


def test_overlap_rule_is_per_context_and_checked(sat_ctx, ctx):
    with pytest.raises(capi.HvrtError):
        sat_ctx.set_overlap_test(2)
    tris = np.ascontiguousarray(meshgen.cube_sphere(12), np.float32)
    a, mn, mx, vs = _build(sat_ctx, tris, (16, 16, 16))
    sat_ctx.set_overlap_test(0)                      # back to the reference's rule: the next build is the reference's grid
    b, _, _, _ = _build(sat_ctx, tris, (16, 16, 16))
    ooff, oidx = oracle.voxelize(tris, 16, 16, 16, mn, vs)
    off, idx = sat_ctx.get_csr()
    assert a.n_refs < b.n_refs == oidx.size and np.array_equal(off, ooff) and np.array_equal(idx, oidx)


def test_sat_grid_with_deferred_derived_data(sat_ctx):
    """hvrt_set_deferred_accel: the face bits are then made at the first render, from the lists (not from the ranges)."""
    tris = np.ascontiguousarray(meshgen.cube_sphere(24), np.float32)
    dims = (32, 32, 32)
    sat_ctx.set_deferred_accel(True)
    try:
        info, mn, mx, vs = _build(sat_ctx, tris, dims)
        off, idx = sat_ctx.get_csr()
        sc = oracle.OracleScene(tris, *dims, mn, vs, off, idx, mx=mx)
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "side"))
        exp = sc.render(fr, 240, 180, threads=8)
        rgb, hit, t, cnt = sat_ctx.render(fr, 240, 180)
        assert np.array_equal(hit, exp[1]) and np.array_equal(t.view(np.uint32), exp[2].view(np.uint32)) and np.array_equal(cnt, exp[3])
    finally:
        sat_ctx.set_deferred_accel(False)


def test_hvrt_overlap_in_the_environment_sets_the_default_rule(monkeypatch):
    """HVRT_OVERLAP=sat: a context starts with the separating-axis rule (what hvrt_set_overlap_test(ctx, 1) selects); any
    other value leaves the reference's bounding-box rule."""
    tris = np.ascontiguousarray(meshgen.slivers(200, 3), np.float32)
    dims = (24, 24, 24)
    got = {}
    for value in ("sat", "aabb"):
        monkeypatch.setenv("HVRT_OVERLAP", value)
        c = ra.Context(0)
        try:
            assert c.counter_width() == (32 if value == "sat" else 16)
            info, mn, mx, vs = _build(c, tris, dims)
            got[value] = c.get_csr()
            want = oracle.voxelize(tris, *dims, mn, vs, sat=(value == "sat"))
            assert np.array_equal(got[value][0], want[0]) and np.array_equal(got[value][1], want[1])
        finally:
            c.close()
    assert got["sat"][1].size < got["aabb"][1].size
