"""GPU parity of the tile cover (raytracer_amd/csrc/hvrt_cover.hip).

Launches without counters -- the ones bench.py times and the host layer uses -- run a pass over the triangles first
that marks the 8x8-pixel tiles in which a ray can hit anything; the other tiles are not walked, their pixels get the
reference's result for a miss.  The launches WITH counters walk every tile (the counters are the reference's per-ray
statistics), so a frame rendered both ways compares the cover against the plain traversal bit for bit; the frames with
counters are in turn the oracle's.  The cases are the ones the cover's analysis has to hold for: cameras inside the
mesh and in the planes of its triangles, grazing views, triangles behind the eye, large / degenerate / sliver
triangles, frames that are no multiple of the tile size, row tiles of fewer than eight rows, batches, supersampling,
non-finite and degenerate camera frames, a mesh that changes under a camera that stands."""
import numpy as np
import pytest

from raytracer_amd import capi, meshgen

from test_gpu_render import MISS, SCENES, assert_frames_equal, render_both, setup_scene

pytestmark = pytest.mark.gpu


def _hits_inside_cover(ctx, hit):
    """Every tile that holds a hit is marked (the converse need not hold: marking more is allowed)."""
    cov = ctx.tile_cover()
    ty, tx = cov.shape
    H, W = hit.shape
    hp = np.zeros((ty * 8, tx * 8), bool)
    hp[:H, :W] = hit != MISS
    ht = hp.reshape(ty, 8, tx, 8).any(axis=(1, 3))
    assert not (ht & ~cov).any(), f"{int((ht & ~cov).sum())} tiles with a hit are not in the cover"
    return cov, ht


def _plain(ctx, fr, W, H, **kw):
    rgb, hit, t, none = ctx.render(fr, W, H, want_counters=False, **kw)
    assert none is None
    return rgb, hit, t


@pytest.mark.parametrize("view", ["front", "side", "top"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_cover_holds_every_hit_and_is_tight_on_the_reference_views(ctx, name, view):
    gen, dims, (W, H) = SCENES[name]
    tris = gen()
    sc, mn, mx = setup_scene(ctx, tris, dims)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
    exp = sc.render(fr, W, H, threads=8)
    for launch in range(3):              # first launch, ordered launch, ordered launch with the cover standing
        rgb, hit, t = _plain(ctx, fr, W, H)
        assert np.array_equal(hit, exp[1])
        assert np.array_equal(t.view(np.uint32), exp[2].view(np.uint32))
        assert np.array_equal(rgb.view(np.uint32), exp[0].view(np.uint32))
    cov, ht = _hits_inside_cover(ctx, hit)
    if name.startswith("sphere") and view == "front":
        # what the pass is for: the sphere's tiles, not the frame's
        assert cov.sum() <= 2 * ht.sum() + 16, (int(cov.sum()), int(ht.sum()))


CAMERAS = [
    ((0.1, 0.05, 0.2), (1, 0.3, 0.2), 60.0),          # inside the mesh: triangles all around and behind the eye
    ((0.0, 0.0, 0.0), (0.3, 0.9, -0.2), 100.0),       # the centre, wide angle
    ((0, 0, -1e6), (0, 0, 0), 60.0),                  # very far: the mesh is a fraction of a pixel
    ((0, 0, -1e3), (0, 0, 0), 0.2),                   # far with a long lens
    ((40, 30, -20), (0, 0, 0), 60.0),                 # oblique
    ((0.5, 0.5, 0.5), (0.5, 0.5, 5), 60.0),           # looking away from most of it
    ((1.0, 0.0, 0.0), (0.0, 0.0, 0.0), 90.0),         # on the surface, looking in
    ((1.2, 0.0, 0.0), (1.2, 5.0, 0.0), 120.0),        # beside it, looking along it: the mesh at the frame's edge
    ((0.0, 3.0, 0.0), (0.0, 0.0, 0.0), 170.0),        # nearly a half-space of directions
    ((2.5, 1e-3, 1e-3), (0.0, 0.0, 0.0), 1.0),        # a long lens on a few triangles
]


@pytest.mark.parametrize("cam", range(len(CAMERAS)))
def test_cover_under_hard_cameras(ctx, cam):
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    pos, la, fov = CAMERAS[cam]
    fr = capi.make_camera_frame(pos, la, (0, 1, 0) if abs(la[1] - pos[1]) < 4 else (0, 0, 1), fov, 4 / 3)
    got = render_both(ctx, fr, 203, 150)
    assert_frames_equal(got, sc.render(fr, 203, 150, threads=8))
    _plain(ctx, fr, 203, 150)
    _hits_inside_cover(ctx, got[1])


def _special_meshes():
    big = np.array([[-50, -1, -50, 50, -1, -50, 0, -1, 60, 0, 1, 0],           # one huge triangle under the scene
                    [-1, -1, 3, 1, -1, 3, 0, 2, 3, 0, 0, -1],                   # a wall facing the front camera
                    [0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0, 0, 1],     # a point
                    [0, 0, 0, 1, 1, 1, 2, 2, 2, 0, 0, 1],                       # a line (zero area)
                    [0, 0.5, -2, 1e-4, 0.5, 2, -1e-4, 0.5, 2, 0, 1, 0]], np.float32)     # a needle
    yield "large_and_degenerate", np.concatenate([big, meshgen.cube_sphere(6, 0.5)]), (12, 12, 12)
    yield "slivers", meshgen.slivers(300, 7), (20, 20, 20)
    yield "soup", meshgen.random_soup(1500, 9), (17, 9, 13)
    yield "box", meshgen.box(), (4, 4, 4)


@pytest.mark.parametrize("which", range(4))
def test_cover_with_large_degenerate_and_edge_on_triangles(ctx, which):
    """Includes cameras that lie exactly in the planes of triangles (the box's faces, the big triangle's plane): the sign
    of the hit parameter's numerator is then not known and the pass falls back to the general block test."""
    name, tris, dims = list(_special_meshes())[which]
    sc, mn, mx = setup_scene(ctx, tris, dims)
    c = 0.5 * (np.asarray(mn, np.float64) + np.asarray(mx, np.float64))
    ext = float(np.max(np.asarray(mx) - np.asarray(mn)))
    cams = [(c + [0, 0, -2 * ext], c), (c + [1.5 * ext, 0.3 * ext, 0.2 * ext], c),
            ((float(mx[0]), float(c[1]), float(c[2]) - 2 * ext), (float(mx[0]), float(c[1]), float(c[2]))),     # in the plane x = max
            ((float(c[0]), float(mn[1]), float(c[2]) - 1.5 * ext), c),                                           # in the plane y = min
            (c + [0, 0, 0.1 * ext], c + [0.3, 0.2, 1.0])]                                                        # inside
    for pos, la in cams:
        fr = capi.make_camera_frame(tuple(np.asarray(pos, np.float32)), tuple(np.asarray(la, np.float32)), (0, 1, 0), 70.0, 1.5)
        got = render_both(ctx, fr, 150, 100)
        assert_frames_equal(got, sc.render(fr, 150, 100, threads=8))
        _plain(ctx, fr, 150, 100)
        _hits_inside_cover(ctx, got[1])


def test_cover_random_cameras(ctx):
    rng = np.random.default_rng(20261111)
    for gen, dims in ((lambda: meshgen.cube_sphere(20, 0.7, center=(3.0, -2.0, 11.5)), (40, 13, 27)),
                      (lambda: meshgen.random_soup(2000, 4), (21, 17, 19))):
        tris = gen()
        sc, mn, mx = setup_scene(ctx, tris, dims)
        c = 0.5 * (np.asarray(mn, np.float64) + np.asarray(mx, np.float64))
        ext = float(np.max(np.asarray(mx) - np.asarray(mn)))
        for k in range(24):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            dist = ext * float(rng.choice([0.05, 0.4, 0.8, 1.5, 4.0, 30.0]))
            pos = c + d * dist
            la = c + rng.normal(size=3) * ext * float(rng.choice([0.0, 0.2, 1.0]))
            fov = float(rng.choice([5.0, 40.0, 60.0, 110.0]))
            W, H = (int(v) for v in rng.choice([[97, 61], [160, 120], [64, 64], [33, 130]]))
            fr = capi.make_camera_frame(tuple(pos.astype(np.float32)), tuple(la.astype(np.float32)), (0, 1, 0), fov, W / H)
            got = render_both(ctx, fr, W, H)
            assert_frames_equal(got, sc.render(fr, W, H, threads=8))


def test_cover_with_supersampling_batches_and_row_tiles(ctx):
    import torch
    tris = meshgen.cube_sphere(20)
    sc, mn, mx = setup_scene(ctx, tris, (32, 32, 32))
    W, H = 250, 141
    frames = [capi.make_camera_frame(*capi.default_camera(mn, mx, v, aspect=W / H)) for v in ("front", "side", "top")]
    # supersampled, several launches: first launch, ordered, tiles split by sample
    exp = sc.render(frames[0], W, H, threads=16, spp=5, seed=3)
    for launch in range(4):
        assert_frames_equal(render_both(ctx, frames[0], W, H, spp=5, seed=3), exp)
    # a batch: one cover per frame
    for n_parts, tile_rows in ((1, 16), (3, 8), (4, 5), (8, 1)):
        for part in range(n_parts):
            rows = capi.part_rows(H, tile_rows, part, n_parts)
            rgb = torch.zeros((3, rows, W, 3), dtype=torch.float32, device="cuda:0")
            hit = torch.zeros((3, rows, W), dtype=torch.int32, device="cuda:0")
            for launch in range(2):
                ctx.render_batch_part_device(frames, W, H, 1, 0, tile_rows, part, n_parts, rows, rgb.data_ptr(), hit.data_ptr(), 0, 0)
                ctx.synchronize()
                ys = [y for y in range(H) if (y // tile_rows) % n_parts == part]
                for f, fr in enumerate(frames):
                    e = sc.render(fr, W, H, threads=8)
                    assert np.array_equal(hit[f].cpu().numpy().view(np.uint32), e[1][ys])
                    assert np.array_equal(rgb[f].cpu().numpy().view(np.uint32), e[0][ys].view(np.uint32))


def test_cover_gives_way_on_frames_it_cannot_analyse(ctx):
    """Non-finite, degenerate and absurdly scaled camera frames mark everything: the walk decides, as without the pass."""
    tris = meshgen.cube_sphere(10)
    sc, mn, mx = setup_scene(ctx, tris, (12, 12, 12))
    good = capi.make_camera_frame(*capi.default_camera(mn, mx))
    bad = []
    for field, value in (("lower_left", float("nan")), ("horizontal", float("inf")), ("origin", 1e30), ("vertical", 0.0)):
        fr = capi.CameraFrame.from_buffer_copy(bytes(good))
        arr = getattr(fr, field)
        if field == "vertical":
            for k in range(3):
                arr[k] = 0.0
        else:
            arr[1] = value
        bad.append(fr)
    collinear = capi.CameraFrame.from_buffer_copy(bytes(good))
    for k in range(3):
        collinear.vertical[k] = collinear.horizontal[k]
    bad.append(collinear)
    for fr in bad:
        with np.errstate(all="ignore"):
            exp = sc.render(fr, 64, 48, threads=4)
        got = render_both(ctx, fr, 64, 48)
        assert np.array_equal(got[1], exp[1])
        assert np.array_equal(got[0].view(np.uint32), exp[0].view(np.uint32))
        _plain(ctx, fr, 64, 48)
        assert ctx.tile_cover().all()


def test_cover_follows_the_mesh_under_a_camera_that_stands(ctx):
    """The cover is kept while cameras, frame size and mesh stand; a new mesh under the same camera gets a new one."""
    a, b = meshgen.cube_sphere(12, 0.4, center=(-0.5, 0.0, 0.0)), meshgen.cube_sphere(12, 0.4, center=(0.5, 0.3, 0.0))
    mn, mx = np.array([-1, -1, -1], np.float32), np.array([1, 1, 1], np.float32)
    vs = capi.voxel_size(mn, mx, 16, 16, 16)
    fr = capi.make_camera_frame((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), (0, 1, 0), 50.0, 4 / 3)
    covers = []
    for tris in (a, b, a):
        ctx.set_mesh(tris)
        ctx.build_grid(16, 16, 16, mn, mx, vs)
        with_cnt = ctx.render(fr, 200, 150)
        for launch in range(2):
            rgb, hit, t = _plain(ctx, fr, 200, 150)
            assert np.array_equal(hit, with_cnt[1]) and np.array_equal(rgb.view(np.uint32), with_cnt[0].view(np.uint32))
        covers.append(_hits_inside_cover(ctx, hit)[0])
        assert (hit != MISS).any()
    assert not np.array_equal(covers[0], covers[1]) and np.array_equal(covers[0], covers[2])


@pytest.mark.parametrize("env", [{"HVRT_TILE_COVER": "0"}, {"HVRT_TILE_COVER_CACHE": "0"}, {"HVRT_SAMPLE_SPLIT": "0"},
                                 {"HVRT_TILE_SPLIT": "0"}, {"HVRT_TILE_ORDER": "0"}, {"HVRT_WARMUP": "0", "HVRT_TILE_ABORT_US": "1"},
                                 {"HVRT_FEW_TILES": "0"}, {"HVRT_FEW_TILES": "1000000"}, {"HVRT_FEW_TILES": "1000000", "HVRT_TILE_ORDER": "0"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_plain_launches_under_the_switches(monkeypatch, env):
    """The launches without counters (the ones the cover, the queue of marked tiles and the split by sample belong to) under
    the switches that steer them: four launches per view and sample count, frames bit for bit the counting kernel's, which
    are the oracle's."""
    import raytracer_amd as ra
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    own = ra.Context(0)
    try:
        for name, (W, H) in (("skewed_long_lists", (640, 360)), ("sphere_50_default_res", (800, 600))):
            gen, dims, _ = SCENES[name]
            sc, mn, mx = setup_scene(own, gen(), dims)
            for view, spp in (("front", 1), ("side", 4)):
                fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view, aspect=W / H))
                exp = sc.render(fr, W, H, threads=16, spp=spp, seed=2)
                for launch in range(4):
                    rgb, hit, t = _plain(own, fr, W, H, spp=spp, seed=2)
                    assert np.array_equal(hit, exp[1])
                    assert np.array_equal(t.view(np.uint32), exp[2].view(np.uint32))
                    assert np.array_equal(rgb.view(np.uint32), exp[0].view(np.uint32))
        if env.get("HVRT_TILE_COVER") == "0":
            with pytest.raises(Exception):
                own.tile_cover()
    finally:
        own.close()
