"""GPU parity: voxel-grid build through the C ABI vs the oracle -- bit-exact CSR (offsets + indices)."""
import numpy as np
import pytest

import oracle
from raytracer_amd import capi, meshgen

pytestmark = pytest.mark.gpu


def l1_distance(occ):
    """min(255, city-block distance to the nearest True cell) of a (nz, ny, nx) mask -- plain numpy sweeps."""
    d = np.where(occ, 0, 1 << 20).astype(np.int64)
    for ax in range(3):
        v = np.moveaxis(d, ax, 0)
        for i in range(1, v.shape[0]):
            np.minimum(v[i], v[i - 1] + 1, out=v[i])
        for i in range(v.shape[0] - 2, -1, -1):
            np.minimum(v[i], v[i + 1] + 1, out=v[i])
    return np.minimum(d, 255).astype(np.uint8)


def check_distance_field(ctx, off, dims):
    nx, ny, nz = dims
    occ = (np.diff(off.astype(np.int64)) > 0).reshape(nz, ny, nx)
    got = ctx.distance_field()
    want = l1_distance(occ)
    assert np.array_equal(got, want), f"distance field differs in {int((got != want).sum())} voxels"


def build_both(ctx, tris, dims, points=None):
    nx, ny, nz = dims
    ctx.set_mesh(tris)
    pts = tris[:, :9].reshape(-1, 3) if points is None else points
    mn, mx = ctx.compute_bounds(pts) if points is not None else ctx.mesh_bounds()
    omn, omx = oracle.mesh_bounds(pts)
    assert np.array_equal(mn.view(np.uint32), omn.view(np.uint32))
    assert np.array_equal(mx.view(np.uint32), omx.view(np.uint32))
    vs = capi.voxel_size(mn, mx, nx, ny, nz)
    info = ctx.build_grid(nx, ny, nz, mn, mx, vs)
    off, idx = ctx.get_csr()
    ooff, oidx = oracle.voxelize(tris, nx, ny, nz, mn, vs)
    assert np.array_equal(off, ooff), "CSR offsets differ"
    assert np.array_equal(idx, oidx), "CSR indices differ"
    cnt = np.diff(ooff.astype(np.int64))
    assert info.n_refs == oidx.size and info.n_occupied == int((cnt > 0).sum())
    assert info.max_list == (int(cnt.max()) if cnt.size else 0)
    check_distance_field(ctx, off, dims)
    return info, off, idx, mn, mx, vs


def _fresh(env=None):
    import os
    import raytracer_amd as ra
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        return ra.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


FIXTURES = {
    "cube_on_voxel_planes": lambda: (meshgen.box(), (8, 8, 8)),
    "sphere_16": lambda: (meshgen.cube_sphere(12), (16, 16, 16)),
    "sphere_32": lambda: (meshgen.cube_sphere(24), (32, 32, 32)),
    "sphere_50": lambda: (meshgen.cube_sphere(58), (50, 50, 50)),
    "offcentre_noncubic": lambda: (meshgen.cube_sphere(20, 0.7, center=(3.0, -2.0, 11.5)) *
                                   np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], np.float32), (40, 13, 27)),
    "offcentre_box_overhang": lambda: (meshgen.box((1.0, 2.0, 3.0), (4.0, 2.5, 3.25)), (20, 20, 20)),
    "slivers": lambda: (meshgen.slivers(400, 5), (32, 32, 32)),
    "random_soup": lambda: (meshgen.random_soup(5000, 11), (33, 17, 29)),
    "one_triangle": lambda: (meshgen.to_aos(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0.5]]], np.float32)), (5, 5, 5)),
    "single_voxel_grid": lambda: (meshgen.cube_sphere(6), (1, 1, 1)),
    "flat_grid": lambda: (meshgen.cube_sphere(10), (64, 64, 1)),
    # wide grids: the distance field's x pass takes 16 voxels per lane from nx = 1024 on (one, three, two steps per line)
    "wide_1024": lambda: (meshgen.random_soup(300, 3, extent=1.0, size=0.05) * np.array([40, 1, 1] * 3 + [1, 1, 1], np.float32), (1024, 6, 5)),
    "wide_2064": lambda: (meshgen.random_soup(200, 4, extent=1.0, size=0.04) * np.array([80, 1, 1] * 3 + [1, 1, 1], np.float32), (2064, 3, 4)),
    "wide_1040_sparse": lambda: (meshgen.random_soup(6, 5, extent=1.0, size=0.02) * np.array([60, 1, 1] * 3 + [1, 1, 1], np.float32), (1040, 2, 2)),
}


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_csr_bit_exact(ctx, name):
    tris, dims = FIXTURES[name]()
    build_both(ctx, np.ascontiguousarray(tris, np.float32), dims)


def test_long_lists_all_sort_tiers(ctx):
    """Lists of > 32 (LDS bitonic), > 4096 (128 KiB LDS) and > 32768 (global-memory bitonic) entries."""
    tris = meshgen.skewed_statue(12, 300)           # 180 000 patch triangles under a handful of voxels
    info, *_ = build_both(ctx, tris, (8, 8, 8))
    assert info.max_list > 32768
    info, *_ = build_both(ctx, tris, (24, 24, 24))
    assert 4096 < info.max_list <= 32768
    info, *_ = build_both(ctx, meshgen.skewed_statue(30, 60), (32, 32, 32))
    assert 32 < info.max_list <= 4096


def test_a_list_beyond_16_bits_rejects_the_narrow_counters():
    """Four voxels of this scene hold more than 65 535 triangles (89 060 in the longest list).  A fresh context counts
    into 16-bit counters first: the carry shows as two differing totals, the front of the build is issued again with
    32-bit counters -- seen here as the context's counter width going from 16 to 32 for good -- and the CSR is the
    oracle's bit for bit, as it is on a context that never used the narrow counters (HVRT_BUILD_NARROW=0)."""
    tris = meshgen.skewed_statue(12, 420)
    for env, first in (({}, 16), ({"HVRT_BUILD_NARROW": "0"}, 32)):
        c = _fresh(env)
        try:
            assert c.counter_width() == first
            info, *_ = build_both(c, meshgen.cube_sphere(10), (16, 16, 16))       # short lists: the width stays
            assert c.counter_width() == first
            info, *_ = build_both(c, tris, (8, 8, 8))
            assert info.max_list > 65535
            assert c.counter_width() == 32
            info, *_ = build_both(c, tris, (8, 8, 8))                             # and again, now wide from the start
            assert info.max_list > 65535 and c.counter_width() == 32
        finally:
            c.close()


def test_big_triangles_take_the_wave_cooperative_path(ctx):
    """Two triangles spanning a whole slab of the grid (range volume >> 32 voxels) among small ones."""
    big = np.array([[[-1, -1, 0.013], [1, -1, 0.013], [1, 1, 0.013]],
                    [[-1, -1, -0.4], [1, 1, 0.9], [-1, 1, -0.4]]], np.float32)
    tris = np.concatenate([meshgen.cube_sphere(10), meshgen.to_aos(big), meshgen.random_soup(700, 2, size=0.6)])
    build_both(ctx, tris, (48, 48, 48))


def test_empty_and_degenerate_inputs(ctx):
    # no triangles at all: every voxel empty
    ctx.set_mesh(np.zeros((0, 12), np.float32))
    one = np.ones(3, np.float32)
    info = ctx.build_grid(4, 4, 4, -one, one, 0.5)
    off, idx = ctx.get_csr()
    assert info.n_refs == 0 and idx.size == 0 and not off.any() and off.size == 65
    # a mesh that is a single point: voxel size 0 -> 0/0 = NaN -> index 0 (cvttss2si + clamp), as the reference
    pt = meshgen.to_aos(np.full((3, 3, 3), 2.5, np.float32))
    build_both(ctx, pt, (4, 4, 4))
    # NaN / inf corners
    t = meshgen.random_soup(200, 9)
    t[3, 0] = np.nan
    t[17, 4] = np.inf
    t[40, 8] = -np.inf
    t[77, :9] = np.nan
    ctx.set_mesh(t)
    mn, mx = ctx.mesh_bounds()
    omn, omx = oracle.mesh_bounds(t[:, :9].reshape(-1, 3))
    assert np.array_equal(mn.view(np.uint32), omn.view(np.uint32)) and np.array_equal(mx.view(np.uint32), omx.view(np.uint32))
    vs = capi.voxel_size(mn, mx, 16, 16, 16)
    ctx.build_grid(16, 16, 16, mn, mx, vs)
    off, idx = ctx.get_csr()
    ooff, oidx = oracle.voxelize(t, 16, 16, 16, mn, vs)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)


def test_bounds_signed_zero_first_occurrence_wins(ctx):
    """std::min keeps the first of +0 / -0 (SceneCreator.cpp:71-76); the GPU reduction must as well."""
    pts = np.ones((100000, 3), np.float32)
    pts[70000] = [-0.0, 0.0, -0.0]
    pts[70001] = [0.0, -0.0, 0.0]
    pts[5] = [3.0, 3.0, 3.0]
    pts[99999] = [3.0, 3.0, 3.0]
    pts[12345] = [np.nan, np.inf, -np.inf]
    mn, mx = ctx.compute_bounds(pts)
    omn, omx = oracle.mesh_bounds(pts)
    assert np.array_equal(mn.view(np.uint32), omn.view(np.uint32))
    assert np.array_equal(mx.view(np.uint32), omx.view(np.uint32))
    assert np.signbit(mn[0]) and not np.signbit(mn[1]) and np.signbit(mn[2])


def test_build_is_deterministic(ctx):
    tris = meshgen.skewed_statue(20, 120)
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, 40, 40, 40)
    first = None
    for _ in range(10):
        ctx.build_grid(40, 40, 40, mn, mx, vs)
        off, idx = ctx.get_csr()
        if first is None:
            first = (off, idx)
        assert np.array_equal(off, first[0]) and np.array_equal(idx, first[1])


def test_ingest_cpu_built_csr_round_trip(ctx):
    tris = meshgen.cube_sphere(16)
    sc, mn, mx, vs = oracle.build_scene(tris, 20, 20, 20)
    ctx.set_mesh(tris)
    ctx.set_grid_csr(20, 20, 20, mn, mx, vs, sc.offsets, sc.indices)
    off, idx = ctx.get_csr()
    assert np.array_equal(off, sc.offsets) and np.array_equal(idx, sc.indices)
    info = ctx.grid_info()
    cnt = np.diff(sc.offsets.astype(np.int64))
    assert info.n_occupied == int((cnt > 0).sum()) and info.max_list == int(cnt.max())


@pytest.mark.parametrize("dims,fill", [((520, 3, 2), 0.01), ((1028, 2, 1), 0.003), ((70, 5, 3), 0.02), ((256, 9, 7), 0.0),
                                       ((64, 64, 64), 0.0005), ((300, 1, 1), 0.01), ((4, 300, 2), 0.02), ((16388, 1, 2), 0.0002),
                                       # sweeps cut into four segments per line (lines of >= 64 voxels): uneven segment
                                       # lengths, a last segment that is empty (65 = 17 + 17 + 17 + 14; 129, 257), values
                                       # that saturate at 255 across segments, and the packed form (>= 262 144 lines)
                                       ((8, 100, 67), 0.001), ((12, 65, 130), 0.0003), ((4, 257, 129), 0.00005), ((3, 1100, 2), 0.0004),
                                       ((512, 512, 64), 0.00002), ((516, 70, 512), 0.00001), ((64, 66, 70), 0.0)])
def test_distance_field_of_ingested_grids(ctx, dims, fill):
    """Every kernel variant of the empty-space distance field (4-voxel and 1-voxel x pass, packed and plain
    sweeps, whole-line and segmented, saturation at 255, an empty grid) against plain numpy sweeps, on synthetic
    occupancy."""
    nx, ny, nz = dims
    rng = np.random.default_rng(nx * 31 + ny * 7 + nz)
    occ = rng.random(nx * ny * nz) < fill
    off = np.concatenate([[0], np.cumsum(occ)]).astype(np.uint32)
    idx = np.zeros(int(occ.sum()), np.uint32)
    ctx.set_mesh(meshgen.box())
    one = np.ones(3, np.float32)
    ctx.set_grid_csr(nx, ny, nz, np.zeros(3, np.float32), one * max(dims), one, off, idx)
    check_distance_field(ctx, off, dims)


def test_errors_are_reported_not_swallowed(ctx):
    one = np.ones(3, np.float32)
    ctx.set_mesh(meshgen.box())
    with pytest.raises(capi.HvrtError):
        ctx.build_grid(0, 4, 4, -one, one, 0.5)
    with pytest.raises(capi.HvrtError):
        ctx.build_grid(70000, 1, 1, -one, one, 0.5)
    with pytest.raises(capi.HvrtError):
        ctx.grid_info()          # the failed build left no grid behind
    ctx.build_grid(4, 4, 4, -one, one, 0.5)
    assert ctx.grid_info().n_refs > 0


def test_config2_full_size_csr(ctx):
    """BASELINE config 2: 1 002 252-triangle sphere into 256^3 -- full bit-exact compare plus the
    size-independent properties (monotone offsets, ascending lists, per-triangle reference count)."""
    tris = meshgen.cube_sphere(289)
    assert tris.shape[0] == 1002252
    info, off, idx, mn, mx, vs = build_both(ctx, tris, (256, 256, 256))
    assert (np.diff(off.astype(np.int64)) >= 0).all() and off[0] == 0 and off[-1] == idx.size
    # ascending inside every list: a descent may only happen at a list boundary
    desc = np.flatnonzero(np.diff(idx.astype(np.int64)) <= 0) + 1
    assert np.isin(desc, off).all()
    # every triangle is referenced exactly vol(range) times
    refs = np.bincount(idx, minlength=tris.shape[0])
    t = tris[:, :9].reshape(-1, 3, 3)
    f = np.float32
    lo = np.clip(((t.min(1) - mn).astype(f) / f(vs)).astype(np.int32), 0, 255)
    hi = np.clip(((t.max(1) - mn).astype(f) / f(vs)).astype(np.int32), 0, 255)
    assert np.array_equal(refs, np.prod(hi - lo + 1, axis=1))


def test_config3_like_skewed_full_size_csr(ctx):
    """BASELINE config 3 shape: 9.8 M triangles, 8.8 M of them in a small patch, 512^3 (longest list 3706):
    CSR and distance field compared in full."""
    tris = meshgen.skewed_statue(289, 2100)
    info, off, idx, mn, mx, vs = build_both(ctx, tris, (512, 512, 512))
    assert info.max_list > 1000 and info.n_refs > 15_000_000


def test_config3_bench_mesh_full_size_csr(ctx):
    """BASELINE config 3 on the very mesh bench.py --workload c3 times (cube-sphere N = 913 = 10 002 828
    triangles, plus the dense patch; 512^3): CSR and distance field compared in full."""
    tris = meshgen.skewed_statue(913, 1100)
    assert tris.shape[0] == 10_002_828 + 2 * 1100 * 1100
    info, off, idx, mn, mx, vs = build_both(ctx, tris, (512, 512, 512))
    assert info.max_list > 1000 and info.n_refs > 25_000_000


def test_config5_full_size_csr(ctx):
    """BASELINE config 5: 49 988 172 triangles into 1024^3 (build only).  Offsets (4.3 GB) and indices compared
    bit for bit with the oracle; the distance field checked through the property that defines it: 0 exactly
    on occupied voxels, elsewhere 1 + the smallest of the six neighbours (saturating at 255)."""
    g = 1024
    tris = meshgen.cube_sphere(2041)
    assert tris.shape[0] == 49_988_172
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, g, g, g)
    info = ctx.build_grid(g, g, g, mn, mx, vs)
    off, idx = ctx.get_csr()
    ooff, oidx = oracle.voxelize(tris, g, g, g, mn, vs)
    del tris
    assert info.n_refs == oidx.size
    assert np.array_equal(idx, oidx), "CSR indices differ"
    del idx, oidx
    assert np.array_equal(off, ooff), "CSR offsets differ"
    del ooff
    occ = (off[1:] != off[:-1]).reshape(g, g, g)
    del off
    assert info.n_occupied == int(occ.sum())
    d = ctx.distance_field()
    assert np.array_equal(d == 0, occ)
    del occ
    m = np.full(d.shape, 255, np.uint8)
    for ax in range(3):
        lo = [slice(None)] * 3; hi = [slice(None)] * 3
        lo[ax], hi[ax] = slice(0, g - 1), slice(1, g)
        np.minimum(m[tuple(hi)], d[tuple(lo)], out=m[tuple(hi)])
        np.minimum(m[tuple(lo)], d[tuple(hi)], out=m[tuple(lo)])
    want = np.where(m == 255, 255, m + 1).astype(np.uint8)
    free = d != 0
    assert np.array_equal(d[free], want[free])


@pytest.mark.parametrize("reuse", ["1", "0"])
def test_builds_that_go_on_without_the_host_match_sized_builds(reuse):
    """A context that has built before keeps its arrays and lets the kernels behind the scan read the reference count
    on the device (no host round trip); a build that outgrows the arrays writes nothing and is repeated with new ones.
    Small -> large (outgrows) -> small (fits) -> the same again, every CSR against the oracle, every frame against a
    context that always waits for the count (HVRT_BUILD_REUSE=0 runs the same sequence that way)."""
    c = _fresh({"HVRT_BUILD_REUSE": reuse})
    try:
        scenes = [(meshgen.cube_sphere(10), (16, 16, 16)), (meshgen.skewed_statue(30, 200), (48, 48, 48)),
                  (meshgen.cube_sphere(14), (20, 20, 20)), (meshgen.cube_sphere(14), (20, 20, 20)),
                  (meshgen.random_soup(3000, 4), (31, 9, 40))]
        for tris, dims in scenes:
            info, off, idx, mn, mx, vs = build_both(c, tris, dims)
            fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front"))
            rgb, hit, t, cnt = c.render(fr, 160, 120)
            orgb, ohit, ot, ocnt = oracle.OracleScene(tris, *dims, mn, vs, off, idx).render(fr, 160, 120)
            assert np.array_equal(hit, ohit) and np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))
            assert np.array_equal(cnt, ocnt)
    finally:
        c.close()


def test_deferred_acceleration_data(ctx):
    """hvrt_set_deferred_accel: the build returns after the CSR; distance field, face bits and records appear at first use
    and the frames are the ones an eager build gives."""
    tris, dims = meshgen.skewed_statue(24, 80), (40, 40, 40)
    c = _fresh()
    try:
        c.set_deferred_accel(True)
        for _ in range(2):                         # the second build reuses the first one's arrays
            info, off, idx, mn, mx, vs = build_both(c, tris, dims)      # ends with get_distance_field: builds the data
            assert c.timings().build_expand_ms > 0.0
        c.build_grid(*dims, mn, mx, vs)
        assert c.timings().build_expand_ms == 0.0
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front"))
        import torch
        d = torch.empty((64, 64, 3), dtype=torch.float32, device="cuda:0")
        with pytest.raises(capi.HvrtError, match="hvrt_finish_accel"):
            c.render_part_device_noalloc(fr, 64, 64, 1, 0, 16, 0, 1, d.data_ptr())
        rgb, hit, t, cnt = c.render(fr, 200, 150)               # first use builds the derived data
        assert c.timings().build_expand_ms > 0.0
        ctx.set_mesh(tris)
        ctx.build_grid(*dims, mn, mx, vs)
        rgb2, hit2, t2, cnt2 = ctx.render(fr, 200, 150)
        assert np.array_equal(hit, hit2) and np.array_equal(rgb.view(np.uint32), rgb2.view(np.uint32))
        assert np.array_equal(t.view(np.uint32), t2.view(np.uint32)) and np.array_equal(cnt, cnt2)
        # an ingested CSR, deferred, then finished explicitly
        c.set_grid_csr(*dims, mn, mx, vs, off, idx)
        c.finish_accel()
        rgb3, hit3, *_ = c.render(fr, 200, 150)
        assert np.array_equal(hit, hit3) and np.array_equal(rgb.view(np.uint32), rgb3.view(np.uint32))
    finally:
        c.close()
