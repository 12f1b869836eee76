"""GPU parity: primary-ray traversal / shading through the C ABI vs the oracle.

The bar (north_star): hit ids equal on >= 99.99 % of pixels, t / normals within 1e-5 relative.  What is
asserted here is stricter -- hit ids, t, colours and the voxel / triangle-test counters are all
BIT-EXACT -- because the kernels keep the reference's fp32 operation order (no FMA contraction) and its
NaN / inf control flow."""
import numpy as np
import pytest

import oracle
from raytracer_amd import capi, meshgen

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


def setup_scene(ctx, tris, dims):
    sc, mn, mx, vs = oracle.build_scene(tris, *dims)
    ctx.set_mesh(tris)
    ctx.build_grid(*dims, mn, mx, vs)
    return sc, mn, mx


def render_both(ctx, *args, **kw):
    """hvrt_render with and without the counters: two different kernel instantiations (the one without is the one
    bench.py times; it checks "still in the grid" once per run of empty-space steps instead of per step).  Their
    frames must be the same bits; the one with counters is returned for the comparison with the oracle."""
    with_cnt = ctx.render(*args, **kw)
    rgb, hit, t, none = ctx.render(*args, want_counters=False, **kw)
    assert none is None
    assert np.array_equal(hit, with_cnt[1]), "hit ids differ between the counting and the plain kernel"
    assert np.array_equal(t.view(np.uint32), with_cnt[2].view(np.uint32))
    assert np.array_equal(rgb.view(np.uint32), with_cnt[0].view(np.uint32))
    return with_cnt


def assert_frames_equal(got, exp):
    rgb, hit, t, cnt = got
    orgb, ohit, ot, ocnt = exp
    n = hit.size
    same = int((hit == ohit).sum())
    assert same >= n * 0.9999, f"hit ids equal on only {same}/{n}"
    assert np.array_equal(hit, ohit)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))
    if cnt is not None:
        assert np.array_equal(cnt, ocnt)
    # the tolerance the metric states, spelled out (trivially true once bits agree)
    m = ohit != MISS
    assert np.all(np.abs(t[m] - ot[m]) <= 1e-5 * np.abs(ot[m]))


SCENES = {
    "cube_on_voxel_planes": (lambda: meshgen.box(), (8, 8, 8), (160, 120)),
    "sphere_16": (lambda: meshgen.cube_sphere(12), (16, 16, 16), (200, 150)),
    "sphere_50_default_res": (lambda: meshgen.cube_sphere(58), (50, 50, 50), (800, 600)),   # config 1 shape
    "offcentre_noncubic": (lambda: meshgen.cube_sphere(20, 0.7, center=(3.0, -2.0, 11.5)), (40, 13, 27), (240, 180)),
    "slivers": (lambda: meshgen.slivers(400, 5), (32, 32, 32), (320, 240)),
    "skewed_long_lists": (lambda: meshgen.skewed_statue(30, 120), (32, 32, 32), (320, 240)),
    "random_soup": (lambda: meshgen.random_soup(3000, 4), (33, 17, 29), (256, 192)),
}


@pytest.mark.parametrize("view", ["front", "side", "top"])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_render_bit_exact(ctx, name, view):
    """FRONT / SIDE / TOP cameras of TraceImageMultiView (TraceImages.cpp:148-197).  Centred meshes make
    dir components exactly zero on the middle row / column: the inf / NaN tMax paths."""
    gen, dims, (W, H) = SCENES[name]
    tris = gen()
    sc, mn, mx = setup_scene(ctx, tris, dims)
    pos, la, up, fov, asp = capi.default_camera(mn, mx, view)
    fr = capi.make_camera_frame(pos, la, up, fov, asp)
    got = render_both(ctx, fr, W, H)
    exp = sc.render(fr, W, H, threads=8)
    assert_frames_equal(got, exp)
    if name.startswith("sphere") and view == "front":
        hit = got[1]
        assert (hit != MISS).any()
        # F7: on the middle row (dir.y == 0 exactly) rays miss inside the silhouette
        mid = hit[H // 2]
        above = hit[H // 2 + 1]
        assert (mid == MISS).sum() > (above == MISS).sum()


def test_odd_sizes_and_tiny_images(ctx):
    tris = meshgen.cube_sphere(12)
    sc, mn, mx = setup_scene(ctx, tris, (16, 16, 16))
    pos, la, up, fov, asp = capi.default_camera(mn, mx)
    fr = capi.make_camera_frame(pos, la, up, fov, asp)
    for (W, H) in [(1, 1), (3, 5), (9, 7), (63, 65), (130, 17)]:
        assert_frames_equal(render_both(ctx, fr, W, H), sc.render(fr, W, H, threads=1))


def test_camera_inside_and_far_away(ctx):
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    for pos, la in [((0.1, 0.05, 0.2), (1, 0.3, 0.2)),        # origin inside the grid
                    ((0, 0, -1e6), (0, 0, 0)),                # very far: huge tMax values
                    ((40, 30, -20), (0, 0, 0)),               # oblique, start voxel clamped on 3 axes
                    ((0.5, 0.5, 0.5), (0.5, 0.5, 5))]:
        fr = capi.make_camera_frame(pos, la, (0, 1, 0), 60.0, 4 / 3)
        assert_frames_equal(render_both(ctx, fr, 160, 120), sc.render(fr, 160, 120, threads=4))


def test_trace_ray_batch(ctx):
    """IRayTracer::traceRay semantics incl. axis-parallel rays, zero directions, NaN, origins inside."""
    tris = meshgen.cube_sphere(20)
    sc, mn, mx = setup_scene(ctx, tris, (32, 32, 32))
    rng = np.random.default_rng(5)
    n = 4000
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    aim = rng.random(n) < 0.6                                   # aim most rays at the sphere
    tgt = rng.normal(size=(n, 3)) * 0.4
    dd = tgt - o
    dd = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    d[aim] = dd[aim]
    o[:200] = rng.uniform(-0.9, 0.9, (200, 3))                  # origins inside the grid
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0],
                     [-0.0, 0.0, 1.0], [np.nan, 0, 1], [0, np.inf, 0], [1, 1, 0]], np.float32)
    k = 200
    for a in axes:
        d[k:k + 20] = a
        k += 20
    o[k:k + 20] = np.float32(-1.0)                               # exactly on the grid's min corner
    got, gcnt = ctx.trace(o, d)
    exp, ecnt = sc.trace(o, d)
    assert np.array_equal(gcnt, ecnt)
    for f in ("hit", "triangle_index", "voxel_index"):
        assert np.array_equal(got[f], exp[f]), f
    for f in ("t", "point", "origin", "normal"):
        assert np.array_equal(got[f].view(np.uint32), exp[f].view(np.uint32)), f
    assert 0 < got["hit"].sum() < n
    miss = got["hit"] == 0
    assert np.all(got["t"][miss] == np.finfo(np.float32).max) and np.all(got["triangle_index"][miss] == 0)
    # device-resident form: same records, same counters
    import torch
    d_o, d_d = torch.from_numpy(o).to("cuda:0"), torch.from_numpy(d).to("cuda:0")
    d_h = torch.zeros(n * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
    d_c = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    ctx.trace_device(d_o.data_ptr(), d_d.data_ptr(), n, d_h.data_ptr(), d_c.data_ptr())
    ctx.synchronize()
    dev = d_h.cpu().numpy().view(capi.HIT_DTYPE)
    assert dev.tobytes() == got.tobytes()
    assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), gcnt)


def test_jittered_supersampling_extension(ctx):
    """spp > 1 (config 4's mode): no reference counterpart -- pinned by the oracle only."""
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    pos, la, up, fov, asp = capi.default_camera(mn, mx, aspect=16 / 9)
    fr = capi.make_camera_frame(pos, la, up, fov, asp)
    for spp, seed in [(4, 1), (16, 1234)]:
        got = render_both(ctx, fr, 192, 108, spp=spp, seed=seed)
        exp = sc.render(fr, 192, 108, spp=spp, seed=seed, threads=8)
        assert_frames_equal(got, exp)
    a = render_both(ctx, fr, 192, 108, spp=4, seed=1)[0]
    b = render_both(ctx, fr, 192, 108, spp=4, seed=2)[0]
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("n_parts,tile_rows", [(2, 16), (3, 8), (8, 16), (4, 37), (8, 1)])
def test_row_tile_partition_reassembles_to_the_single_gpu_frame(ctx, n_parts, tile_rows):
    """The multi-GPU partition on one GPU: render every part into its slab, reassemble, compare bytewise
    with the one-part frame (no reduction anywhere => exact equality is the right assertion)."""
    import torch
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    W, H = 200, 150
    pos, la, up, fov, asp = capi.default_camera(mn, mx)
    fr = capi.make_camera_frame(pos, la, up, fov, asp)
    full_rgb, full_hit, full_t, _ = render_both(ctx, fr, W, H)
    slab = max(capi.part_rows(H, tile_rows, p, n_parts) for p in range(n_parts))
    dev = torch.device("cuda:0")
    parts_rgb = torch.zeros((n_parts, slab, W, 3), dtype=torch.float32, device=dev)
    parts_hit = torch.zeros((n_parts, slab, W), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for p in range(n_parts):
        ctx.render_part_device(fr, W, H, 1, 0, tile_rows, p, n_parts,
                               d_rgb=parts_rgb[p].data_ptr(), d_hit=parts_hit[p].data_ptr())
    out_rgb = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    out_hit = torch.empty((H, W), dtype=torch.int32, device=dev)
    ctx.assemble_parts_device(parts_rgb.data_ptr(), W, H, tile_rows, n_parts, slab, 12, out_rgb.data_ptr())
    ctx.assemble_parts_device(parts_hit.data_ptr(), W, H, tile_rows, n_parts, slab, 4, out_hit.data_ptr())
    ctx.synchronize()
    assert np.array_equal(out_rgb.cpu().numpy().view(np.uint32), full_rgb.view(np.uint32))
    assert np.array_equal(out_hit.cpu().numpy().view(np.uint32), full_hit)


@pytest.mark.parametrize("n_parts,part,tile_rows", [(1, 0, 16), (3, 1, 8), (8, 7, 16)])
def test_batch_launch_equals_single_launches_and_order_changes_nothing(ctx, n_parts, part, tile_rows):
    """hvrt_render_batch_part_device: one launch for several cameras == the launches one by one, bytewise;
    repeated launches (the 2nd and 3rd pull their tiles longest-first from the previous launch's per-tile
    clocks) give the same bytes again."""
    import torch
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    W, H = 203, 157
    cams = [capi.make_camera_frame(*capi.default_camera(mn, mx, v)) for v in ("front", "side", "top", "front")]
    rows = capi.part_rows(H, tile_rows, part, n_parts)
    slab = rows + 3                                   # slabs may be padded
    dev = torch.device("cuda:0")
    single = torch.zeros((len(cams), slab, W, 3), dtype=torch.float32, device=dev)
    single_hit = torch.zeros((len(cams), slab, W), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for f, cam in enumerate(cams):
        ctx.render_part_device(cam, W, H, 1, 0, tile_rows, part, n_parts,
                               d_rgb=single[f].data_ptr(), d_hit=single_hit[f].data_ptr())
    ctx.synchronize()
    want, want_hit = single.cpu().numpy(), single_hit.cpu().numpy()
    for rep in range(3):
        batch = torch.zeros_like(single)
        batch_hit = torch.zeros_like(single_hit)
        torch.cuda.synchronize()
        ctx.render_batch_part_device(cams, W, H, 1, 0, tile_rows, part, n_parts, slab,
                                     d_rgb=batch.data_ptr(), d_hit=batch_hit.data_ptr())
        ctx.synchronize()
        assert np.array_equal(batch.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"rgb differs, launch {rep}"
        assert np.array_equal(batch_hit.cpu().numpy(), want_hit), f"hit ids differ, launch {rep}"
    with pytest.raises(capi.HvrtError):
        ctx.render_batch_part_device(cams, W, H, 1, 0, tile_rows, part, n_parts, rows - 1 if rows else -1,
                                     d_rgb=batch.data_ptr())
    with pytest.raises(capi.HvrtError):
        ctx.render_batch_part_device(cams * 9, W, H, 1, 0, tile_rows, part, n_parts, slab, d_rgb=batch.data_ptr())


def test_batch_of_many_tiles_takes_the_scanned_queue_compaction(ctx):
    """32 frames of 1024x520 are 266 240 tiles = 1040 workgroups for the queue's compaction: from 1024 on k_compact_scan turns
    the per-workgroup counts into prefixes in one launch (instead of every workgroup adding up all the counts before its
    own).  Same bytes as the launches one by one, on the launch without history and on two with it."""
    import torch
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    W, H, n = 1024, 520, 32
    views = ("front", "side", "top")
    cams = [capi.make_camera_frame(*capi.default_camera(mn, mx, views[f % 3])) for f in range(n)]
    dev = torch.device("cuda:0")
    single = torch.zeros((3, H, W, 3), dtype=torch.float32, device=dev)
    single_hit = torch.zeros((3, H, W), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for f in range(3):
        ctx.render_part_device(cams[f], W, H, 1, 0, H, 0, 1, d_rgb=single[f].data_ptr(), d_hit=single_hit[f].data_ptr())
    ctx.synchronize()
    want, want_hit = single.cpu().numpy().view(np.uint32), single_hit.cpu().numpy()
    assert int((want_hit != -1).sum()) > 1000
    for rep in range(3):
        batch = torch.zeros((n, H, W, 3), dtype=torch.float32, device=dev)
        batch_hit = torch.zeros((n, H, W), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.render_batch_part_device(cams, W, H, 1, 0, H, 0, 1, H, d_rgb=batch.data_ptr(), d_hit=batch_hit.data_ptr())
        ctx.synchronize()
        got, got_hit = batch.cpu().numpy().view(np.uint32), batch_hit.cpu().numpy()
        for f in range(n):
            assert np.array_equal(got[f], want[f % 3]), f"rgb differs, frame {f}, launch {rep}"
            assert np.array_equal(got_hit[f], want_hit[f % 3]), f"hit ids differ, frame {f}, launch {rep}"


def test_launches_queued_without_waiting_keep_their_durations_and_their_bytes(ctx):
    """hvrt_render_history: launches queued back to back (what bench.py's timed region does) are each bracketed by their
    own pair of events -- the last 32 are kept -- and every one of them writes the frame the first one wrote."""
    import torch
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    W, H = 320, 200
    cam = capi.make_camera_frame(*capi.default_camera(mn, mx, "front"))
    dev = torch.device("cuda:0")
    first = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.render_part_device(cam, W, H, 1, 0, H, 0, 1, d_rgb=first.data_ptr())
    ctx.synchronize()
    want = first.cpu().numpy().view(np.uint32)
    outs = [torch.zeros_like(first) for _ in range(5)]
    torch.cuda.synchronize()
    for o in outs:
        ctx.render_part_device(cam, W, H, 1, 0, H, 0, 1, d_rgb=o.data_ptr())        # no wait in between
    ms = ctx.render_history(5)
    assert len(ms) == 5 and all(0.0 < m < 100.0 for m in ms), ms
    assert ms[-1] == pytest.approx(ctx.timings().render_ms, abs=1e-6)                 # the last launch is hvrt_timings' one
    for k, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want), f"queued launch {k} differs"
    assert len(ctx.render_history(3)) == 3
    for _ in range(40):
        ctx.render_part_device(cam, W, H, 1, 0, H, 0, 1, d_rgb=outs[0].data_ptr())
    assert len(ctx.render_history(64)) == 32                                           # the ring holds 32
    assert np.array_equal(outs[0].cpu().numpy().view(np.uint32), want)
    with pytest.raises(capi.HvrtError):
        ctx._ck(0 if ctx._lib.hvrt_render_history(ctx._h, None, 4) >= 0 else -1)
    # the same with tiles that go to k_render_split (a scene with long lists: launches 2.. of a shape hand the long tiles to
    # the second kernel, which k_render's stream waits for -- k_wait_split_started -- before its persistent waves fill the GPU)
    tris = meshgen.skewed_statue(24, 120)
    sc, mn, mx = setup_scene(ctx, tris, (32, 32, 32))
    W, H = 640, 360
    cam = capi.make_camera_frame(*capi.default_camera(mn, mx, aspect=W / H))
    exp = sc.render(cam, W, H, threads=16)
    want_rgb = np.ascontiguousarray(exp[0], np.float32).reshape(H, W, 3).view(np.uint32)
    outs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(6)]
    hits = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(6)]
    torch.cuda.synchronize()
    for _ in range(2):                                   # history: from the second launch on some tiles are split
        ctx.render_part_device(cam, W, H, 1, 0, H, 0, 1, d_rgb=outs[0].data_ptr())
        ctx.synchronize()
    outs[0].zero_()
    torch.cuda.synchronize()
    for o, h in zip(outs, hits):
        ctx.render_part_device(cam, W, H, 1, 0, H, 0, 1, d_rgb=o.data_ptr(), d_hit=h.data_ptr())
    assert len(ctx.render_history(6)) == 6
    want_hit = np.asarray(exp[1]).reshape(H, W).astype(np.uint32)
    for k, (o, h) in enumerate(zip(outs, hits)):
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want_rgb), f"queued launch {k} with split tiles: rgb differs from the oracle"
        assert np.array_equal(h.cpu().numpy().view(np.uint32), want_hit), f"queued launch {k} with split tiles: hit ids differ"


def _orbit_camera(mn, mx, degrees, aspect):
    pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=aspect)
    a = np.deg2rad(degrees)
    rel = np.asarray(pos, np.float64) - np.asarray(la, np.float64)
    axis = np.asarray(up, np.float64)
    rot = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
    return capi.make_camera_frame((np.asarray(la, np.float64) + rot).astype(np.float32), la, up, fov, asp)


@pytest.mark.parametrize("max_deg", [None, "0"])
def test_a_camera_that_moves_changes_only_the_order(monkeypatch, max_deg):
    """Launches of one shape whose camera moves: steps of 1 degree are ordered by the previous frame's tile durations, a jump of
    20 degrees is past HVRT_HISTORY_MAX_DEG (2.5) and runs like a first launch -- or, with the limit removed, is ordered by
    the far view's durations.  Queued without waiting; every frame must be the oracle's, bit for bit (rgb and hit ids)."""
    import torch
    import raytracer_amd as ra
    if max_deg is not None:
        monkeypatch.setenv("HVRT_HISTORY_MAX_DEG", max_deg)
    c = ra.Context(0)
    try:
        tris = meshgen.skewed_statue(24, 120)
        sc, mn, mx = setup_scene(c, tris, (32, 32, 32))
        W, H = 640, 360
        angles = [0.0, 0.0, 1.0, 2.0, 22.0, 23.0, 23.0, 3.0, -40.0]
        cams = [_orbit_camera(mn, mx, a, W / H) for a in angles]
        dev = torch.device("cuda:0")
        outs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in angles]
        hits = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in angles]
        torch.cuda.synchronize()
        for cam, o, h in zip(cams, outs, hits):
            c.render_part_device(cam, W, H, 1, 0, H, 0, 1, d_rgb=o.data_ptr(), d_hit=h.data_ptr())     # no wait in between
        c.synchronize()
        for a, cam, o, h in zip(angles, cams, outs, hits):
            exp = sc.render(cam, W, H, threads=16)
            assert np.array_equal(o.cpu().numpy().view(np.uint32), np.ascontiguousarray(exp[0], np.float32).reshape(H, W, 3).view(np.uint32)), a
            assert np.array_equal(h.cpu().numpy().view(np.uint32), np.asarray(exp[1]).reshape(H, W).view(np.uint32)), a
    finally:
        c.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_split_tiles_change_nothing(ctx, mode):
    """From the second launch of a shape on, tiles that took longer than a balanced launch would are rendered by
    k_render_split (2x2 pixel quads, every ray on 16 lanes that share out each list).  A 640x360 frame of a scene
    with long lists has both kinds of tiles; a 96x64 frame has only split ones.  Frames, hit ids, t and the
    counters of launches 1..4 are the oracle's, in both traversal modes, with 1 and 3 samples per pixel."""
    tris = meshgen.skewed_statue(24, 120)
    sc, mn, mx = setup_scene(ctx, tris, (32, 32, 32))
    try:
        ctx.set_mode(mode)
        sc.set_mode(mode, mx)
        for (W, H, spp) in ((640, 360, 1), (96, 64, 3)):
            fr = capi.make_camera_frame(*capi.default_camera(mn, mx, aspect=W / H))
            exp = sc.render(fr, W, H, threads=16, spp=spp, seed=5)
            for launch in range(4):
                assert_frames_equal(render_both(ctx, fr, W, H, spp=spp, seed=5), exp)
    finally:
        ctx.set_mode(0)
        sc.set_mode(0)


def test_tiles_given_up_on_are_rendered_by_the_split_kernel(monkeypatch):
    """A launch without tile history runs the k_render variant that gives up on a tile whose rays are still inside
    long lists after HVRT_TILE_ABORT_US (0.3 ms by default); k_render_split renders those tiles behind it.  With
    the limit at 1 us every tile that meets a list of more than 32 entries takes that route: the frames (first
    launch of each shape), hit ids, t and counters are still the oracle's, in both modes, with 1 and 3 samples."""
    import raytracer_amd as ra
    monkeypatch.setenv("HVRT_TILE_ABORT_US", "1")
    own = ra.Context(0)
    try:
        tris = meshgen.skewed_statue(24, 120)
        sc, mn, mx = setup_scene(own, tris, (32, 32, 32))
        assert own.grid_info().max_list > 64
        for mode in (0, 1):
            own.set_mode(mode)
            sc.set_mode(mode, mx)
            for (W, H, spp) in ((328, 200, 1), (96, 64, 3)):
                fr = capi.make_camera_frame(*capi.default_camera(mn, mx, aspect=W / H))
                exp = sc.render(fr, W, H, threads=16, spp=spp, seed=5)
                for launch in range(2):          # the second one has history and splits from the start
                    assert_frames_equal(render_both(own, fr, W, H, spp=spp, seed=5), exp)
    finally:
        own.close()


def _quantise_ref(rgb):
    """ImageSaver.h:28-33: (uint8_t)clamp(c * 255.0f, 0.0f, 255.0f), truncating; NaN -> 0 (documented)."""
    v = rgb.astype(np.float32) * np.float32(255.0)
    k = np.where(v < 0, np.float32(0), np.where(v > 255, np.float32(255), v))
    return np.where(np.isnan(k), 0, k).astype(np.int32).astype(np.uint8)


def test_views_rgb8_is_the_float_frames_quantised_like_the_reference(ctx):
    import torch
    tris = meshgen.cube_sphere(16)
    sc, mn, mx = setup_scene(ctx, tris, (24, 24, 24))
    W, H = 203, 157
    cams = [capi.make_camera_frame(*capi.default_camera(mn, mx, v)) for v in ("front", "side", "top")]
    rgb8, hit, cnt = ctx.render_views_rgb8(cams, W, H)
    tot = np.zeros(2, np.uint64)
    for f, cam in enumerate(cams):
        rgb, h, t, c = render_both(ctx, cam, W, H)
        assert np.array_equal(rgb8[f], _quantise_ref(rgb)), f"view {f}"
        assert np.array_equal(hit[f], h)
        tot += c
    assert np.array_equal(cnt, tot)
    # the device conversion on awkward values and a length that is not a multiple of 4
    vals = np.array([-1.0, -0.0, 0.0, 0.0039, 0.00393, 0.5, 0.999, 1.0, 1.1, 300.0, np.nan, np.inf, -np.inf, 1e-40,
                     0.1 + 0.9, 254.9999 / 255.0, 255.0 / 255.0], np.float32)
    vals = np.concatenate([vals, np.random.default_rng(5).random(1000).astype(np.float32) * 1.2 - 0.1])[:1015]
    d_in = torch.from_numpy(vals).to("cuda:0")
    d_out = torch.zeros(vals.size + 1, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ctx.quantise_rgb8_device(d_in.data_ptr(), vals.size, d_out.data_ptr())
    ctx.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:-1], _quantise_ref(vals)) and got[-1] == 0


NEAREST_SCENES = {
    "sphere_24": lambda: (meshgen.cube_sphere(16), (24, 24, 24)),
    "coarse_grid_front_and_back_share_voxels": lambda: (meshgen.cube_sphere(16), (3, 3, 3)),
    "skewed_long_lists": lambda: (meshgen.skewed_statue(12, 60), (16, 16, 16)),
    "slivers": lambda: (meshgen.slivers(300, 7), (20, 20, 20)),
    "offcentre_noncubic": lambda: (meshgen.cube_sphere(14, 0.7, center=(3.0, -2.0, 11.5)), (30, 11, 19)),
}


@pytest.mark.parametrize("name", sorted(NEAREST_SCENES))
def test_nearest_hit_mode_matches_the_oracle(ctx, name):
    """hvrt_set_mode(1) (extension, no reference counterpart -- parity unpinned by the reference; the oracle's
    mode 1 is itself checked against brute force over all triangles in test_oracle_pinning.py): frames from
    three cameras, a camera inside the mesh, and a ray batch, bit for bit incl. the counters; then back to
    mode 0, which must be the reference's frame again."""
    tris, dims = NEAREST_SCENES[name]()
    sc, mn, mx = setup_scene(ctx, tris, dims)
    front = capi.make_camera_frame(*capi.default_camera(mn, mx))
    ref_frame = render_both(ctx, front, 160, 120)
    try:
        ctx.set_mode(1)
        sc.set_mode(1, mx)
        for view in ("front", "side", "top"):
            fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
            assert_frames_equal(render_both(ctx, fr, 160, 120), sc.render(fr, 160, 120, threads=4))
            assert_frames_equal(render_both(ctx, fr, 160, 120), sc.render(fr, 160, 120, threads=4))   # re-ordered launch
        c = (np.asarray(mn) + np.asarray(mx)) / 2
        inside = capi.make_camera_frame(tuple(c), tuple(c + np.array([0.3, 0.2, 1.0], np.float32)), (0, 1, 0), 80.0, 4 / 3)
        assert_frames_equal(render_both(ctx, inside, 120, 90, spp=2, seed=3), sc.render(inside, 120, 90, spp=2, seed=3, threads=4))
        rng = np.random.default_rng(11)
        n = 3000
        o = (c + rng.uniform(-2.5, 2.5, (n, 3)) * (np.asarray(mx) - np.asarray(mn))).astype(np.float32)
        tgt = c + rng.normal(size=(n, 3)) * 0.3 * (np.asarray(mx) - np.asarray(mn))
        d = tgt - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        d[:12] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0], [np.nan, 0, 1],
                           [0, np.inf, 0], [1, 1, 0], [0.0, -0.0, 1.0], [0, 1, 1]], np.float32)
        o[12:24] = np.asarray(mn, np.float32)                      # on the grid's min corner
        got, gcnt = ctx.trace(o, d)
        exp, ecnt = sc.trace(o, d)
        assert got.tobytes() == exp.tobytes() and np.array_equal(gcnt, ecnt)
        assert 0 < got["hit"].sum() < n
    finally:
        ctx.set_mode(0)
        sc.set_mode(0)
    assert_frames_equal(render_both(ctx, front, 160, 120), ref_frame)
    assert_frames_equal(ref_frame, sc.render(front, 160, 120, threads=4))
    with pytest.raises(capi.HvrtError):
        ctx.set_mode(2)


def test_render_requires_a_scene(ctx):
    ctx.set_mesh(meshgen.box())      # drops any grid
    fr = capi.make_camera_frame((0, 0, -5), (0, 0, 0), (0, 1, 0), 60.0, 4 / 3)
    with pytest.raises(capi.HvrtError):
        render_both(ctx, fr, 16, 16)


def test_config2_full_size_render_rows(ctx):
    """BASELINE config 2 at full size (1 002 252 triangles, 256^3, 3840x2160): EVERY pixel of the frame compared
    bit-for-bit with the oracle (hit id, t, colour; the oracle traces the 8.3 M rays in about a second on the
    box's 16 host threads), then frame-level properties."""
    tris = meshgen.cube_sphere(289)
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, 256, 256, 256)
    ctx.build_grid(256, 256, 256, mn, mx, vs)
    off, idx = ctx.get_csr()
    sc = oracle.OracleScene(tris, 256, 256, 256, mn, vs, off, idx)   # CSR itself is checked in test_gpu_build
    W, H = 3840, 2160
    pos, la, up, fov, asp = capi.default_camera(mn, mx, aspect=W / H)
    fr = capi.make_camera_frame(pos, la, up, fov, asp)
    rgb, hit, t, cnt = render_both(ctx, fr, W, H)
    rows = list(range(H))
    ohit = np.zeros((len(rows), W), np.uint32)
    ot = np.zeros((len(rows), W), np.float32)
    orgb = np.zeros((len(rows), W, 3), np.float32)
    from concurrent.futures import ThreadPoolExecutor

    def band(k):
        r = sc.render(fr, W, H, threads=1, rows=(rows[k], rows[k] + 1))
        orgb[k], ohit[k], ot[k] = r[0][rows[k]], r[1][rows[k]], r[2][rows[k]]

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(band, range(len(rows))))
    assert int((hit == ohit).sum()) >= 0.9999 * W * H          # the bar north_star states ...
    assert np.array_equal(hit, ohit)                            # ... and what is actually met: every pixel
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))
    # properties over the whole frame
    m = hit != MISS
    assert 0.03 < m.mean() < 0.10
    assert hit[m].max() < tris.shape[0]
    assert np.all(t[~m] == np.finfo(np.float32).max) and np.all(t[m] > 0) and np.all(np.isfinite(t[m]))
    assert np.all((rgb[m] >= 0.1) & (rgb[m] <= 1.1001))
    v = (np.arange(H, dtype=np.float32) / np.float32(H))[:, None]
    bg = np.float32(0.5) + np.float32(0.3) * v
    assert np.array_equal(np.broadcast_to(bg, (H, W))[~m], rgb[..., 0][~m])


def _rows_against_oracle(ctx, tris, g, W, H, rows, spp=1, seed=0, launches=1, mode=0):
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, g, g, g)
    ctx.build_grid(g, g, g, mn, mx, vs)
    off, idx = ctx.get_csr()
    sc = oracle.OracleScene(tris, g, g, g, mn, vs, off, idx)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, aspect=W / H))
    try:
        ctx.set_mode(mode)
        sc.set_mode(mode, mx)
        for _ in range(launches):        # later launches pull their tiles in the order the first one measured
            rgb, hit, t, cnt = render_both(ctx, fr, W, H, spp=spp, seed=seed)
    finally:
        ctx.set_mode(0)
    ohit = np.zeros((len(rows), W), np.uint32)
    ot = np.zeros((len(rows), W), np.float32)
    orgb = np.zeros((len(rows), W, 3), np.float32)
    from concurrent.futures import ThreadPoolExecutor

    def band(k):
        r = sc.render(fr, W, H, threads=1, rows=(rows[k], rows[k] + 1), spp=spp, seed=seed)
        orgb[k], ohit[k], ot[k] = r[0][rows[k]], r[1][rows[k]], r[2][rows[k]]

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(band, range(len(rows))))
    assert np.array_equal(hit[rows], ohit)
    assert np.array_equal(t[rows].view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(rgb[rows].view(np.uint32), orgb.view(np.uint32))
    return rgb, hit, t


def test_config3_like_full_size_render_rows(ctx):
    """BASELINE config 3 shape (9.8 M triangles, dense patch with 3706-entry lists, 512^3, 3840x2160): every
    24th row, and every 2nd row across the patch, bit for bit against the oracle; third launch (the second is
    ordered and hands the tiles that cross the patch to k_render_split, the third builds on the second's clocks)."""
    W, H = 3840, 2160
    rows = sorted(set(range(0, H, 24)) | set(range(1000, 1160, 2)))
    _rows_against_oracle(ctx, meshgen.skewed_statue(289, 2100), 512, W, H, rows, launches=3)


def test_config3_like_full_size_render_rows_nearest_hit_mode(ctx):
    """The same scene and frame in nearest-hit mode (hvrt_set_mode(1); parity unpinned by the reference, the
    oracle's mode 1 is the checker): every 48th row and every 4th across the patch, third launch."""
    W, H = 3840, 2160
    rows = sorted(set(range(5, H, 48)) | set(range(1000, 1160, 4)))
    _rows_against_oracle(ctx, meshgen.skewed_statue(289, 2100), 512, W, H, rows, launches=3, mode=1)


def test_config3_bench_mesh_full_size_render_rows(ctx):
    """BASELINE config 3 on the very mesh bench.py --workload c3 times (cube-sphere N = 913 plus the dense
    patch, 12.4 M triangles, 512^3, 3840x2160): every 24th row and every 2nd row across the patch, third launch."""
    W, H = 3840, 2160
    rows = sorted(set(range(3, H, 24)) | set(range(1000, 1160, 2)))
    _rows_against_oracle(ctx, meshgen.skewed_statue(913, 1100), 512, W, H, rows, launches=3)


def test_config4_8k_16spp_rows(ctx):
    """BASELINE config 4: the config 2 scene at 7680x4320 with 16 jittered samples per pixel; every 16th row (270 rows,
    33 M rays) against the oracle (colour is the mean of 16 samples summed in sample order; hit id / t are sample 0's)."""
    W, H = 7680, 4320
    rows = list(range(7, H, 16))
    _rows_against_oracle(ctx, meshgen.cube_sphere(289), 256, W, H, rows, spp=16, seed=1)


SWITCHES = [{"HVRT_WARMUP": "0"}, {"HVRT_TILE_SPLIT": "0"}, {"HVRT_TILE_ORDER": "0"}, {"HVRT_SPLIT_WAIT": "0"},
            {"HVRT_TILE_ABORT_US": "0"}, {"HVRT_HISTORY_MAX_DEG": "0"}, {"HVRT_WARMUP": "0", "HVRT_TILE_ABORT_US": "1"},
            {"HVRT_DF_LOW_PRIORITY": "1"}, {"HVRT_INLINE_RECORDS": "0"}, {"HVRT_TRACE_BUILD": "1", "HVRT_TRACE_TIMES": "1"},
            {"HVRT_BUILD_NARROW": "0", "HVRT_BUILD_REUSE": "0"}, {"HVRT_FEW_TILES": "0"},
            {"HVRT_FEW_TILES": "1000000", "HVRT_INLINE_RECORDS": "0"}]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_render_under_every_scheduling_switch(monkeypatch, env):
    """The environment switches that change how a frame is scheduled (never what it is): a fresh context per setting,
    three launches per view so that the first-launch, ordered and split paths all run, frames bit for bit the oracle's.
    HVRT_WARMUP=0 is the case round 2's last fix was for (the split kernel's report word on a context whose queue scratch
    no warm-up launch had touched)."""
    import raytracer_amd as ra
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    own = ra.Context(0)
    try:
        for name, size in (("skewed_long_lists", (1280, 720)), ("sphere_50_default_res", (800, 600)), ("slivers", (320, 240))):
            gen, dims, _ = SCENES[name]
            W, H = size
            tris = gen()
            sc, mn, mx = setup_scene(own, tris, dims)
            for view in ("front", "side"):
                fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view, aspect=W / H))
                exp = sc.render(fr, W, H, threads=16)
                for _ in range(3):
                    assert_frames_equal(own.render(fr, W, H), exp)
                # rgb alone: the two-launch path of hvrt_render (frames of 2^21 pixels and more take it)
            if name == "skewed_long_lists":
                W, H = 2048, 1100
                fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
                exp = sc.render(fr, W, H, threads=16)
                for _ in range(3):
                    rgb, none_h, none_t, cnt = own.render(fr, W, H, want_hit=False, want_t=False)
                    assert np.array_equal(rgb.view(np.uint32), exp[0].view(np.uint32)) and np.array_equal(cnt, exp[3])
    finally:
        own.close()


@pytest.mark.parametrize("inline", ["1", "0"])
def test_the_walk_that_looks_ahead_on_hard_cameras(monkeypatch, inline):
    """trace_ray_ahead (k_render_split's walk: the ray's next voxels from its steps alone, their list bounds fetched
    together, looked at in the ray's order) on every tile that meets an occupied voxel: with HVRT_TILE_ORDER=0 every plain
    launch is one without history, with HVRT_FEW_TILES at a million it gives all its tiles up at once.  Cameras inside the
    grid, a million units away, oblique (start voxel clamped on three axes), degenerate directions on the middle row and
    column, rays that leave the grid in the middle of a batch, grids one voxel wide (the step's increments coincide), odd
    frame sizes, supersampling; per-reference records and indexed ones.  Frames bit for bit the oracle's."""
    import raytracer_amd as ra
    monkeypatch.setenv("HVRT_TILE_ORDER", "0")
    monkeypatch.setenv("HVRT_FEW_TILES", "1000000")
    monkeypatch.setenv("HVRT_INLINE_RECORDS", inline)
    own = ra.Context(0)

    def plain(fr, W, H, **kw):
        rgb, hit, t, none = own.render(fr, W, H, want_counters=False, **kw)
        assert none is None
        return rgb, hit, t, None
    try:
        tris = meshgen.cube_sphere(16)
        sc, mn, mx = setup_scene(own, tris, (24, 24, 24))
        hits = 0
        for pos, la in [((0.1, 0.05, 0.2), (1, 0.3, 0.2)), ((0, 0, -1e6), (0, 0, 0)), ((40, 30, -20), (0, 0, 0)),
                        ((0.5, 0.5, 0.5), (0.5, 0.5, 5)), ((0, 0, -2.5), (0, 0, 0)), ((3, 0, 0), (0, 0, 0))]:
            fr = capi.make_camera_frame(pos, la, (0, 1, 0), 60.0, 4 / 3)
            for (W, H) in ((160, 120), (63, 65), (9, 7)):
                got = plain(fr, W, H)
                assert_frames_equal(got, sc.render(fr, W, H, threads=8)[:3] + (None,))
                hits += int((got[1] != MISS).sum())
        assert hits > 10000
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx, aspect=16 / 9))
        assert_frames_equal(plain(fr, 192, 108, spp=4, seed=7), sc.render(fr, 192, 108, spp=4, seed=7, threads=8)[:3] + (None,))
        for dims in ((1, 24, 24), (24, 1, 24), (24, 24, 1), (1, 1, 24), (40, 13, 27)):
            sc, mn, mx = setup_scene(own, tris, dims)
            for view in ("front", "side", "top"):
                fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
                assert_frames_equal(plain(fr, 200, 150), sc.render(fr, 200, 150, threads=8)[:3] + (None,))
        for name in ("slivers", "random_soup", "skewed_long_lists", "offcentre_noncubic"):
            gen, dims, (W, H) = SCENES[name]
            sc, mn, mx = setup_scene(own, gen(), dims)
            for view in ("front", "side", "top"):
                fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
                assert_frames_equal(plain(fr, W, H), sc.render(fr, W, H, threads=8)[:3] + (None,))
        # the nearest-hit mode takes the same route (its split kernel keeps the walk that does not look ahead)
        sc, mn, mx = setup_scene(own, tris, (24, 24, 24))
        own.set_mode(1)
        sc.set_mode(1, mx)
        for view in ("front", "side", "top"):
            fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
            assert_frames_equal(plain(fr, 200, 150), sc.render(fr, 200, 150, threads=8)[:3] + (None,))
    finally:
        own.close()


@pytest.mark.parametrize("name", ["sphere_16", "slivers", "skewed_long_lists"])
def test_render_from_an_ingested_cpu_grid(ctx, name):
    """IRayTracer's constructor path: the CSR comes from the CPU (hvrt_set_grid_csr); the skip bits are
    then derived by searching the neighbour lists instead of from the builder's ranges."""
    gen, dims, (W, H) = SCENES[name]
    tris = gen()
    sc, mn, mx, vs = oracle.build_scene(tris, *dims)
    ctx.set_mesh(tris)
    ctx.set_grid_csr(*dims, mn, mx, vs, sc.offsets, sc.indices)
    for view in ("front", "top"):
        pos, la, up, fov, asp = capi.default_camera(mn, mx, view)
        fr = capi.make_camera_frame(pos, la, up, fov, asp)
        assert_frames_equal(render_both(ctx, fr, W, H), sc.render(fr, W, H, threads=8))


def test_ingested_grid_with_unsorted_or_foreign_lists(ctx):
    """A CSR that is not the AABB-range CSR (lists shuffled, some entries dropped): results must still be the
    reference algorithm's on THAT grid (last list entry that hits wins), skip bits or not."""
    tris = meshgen.cube_sphere(14)
    sc, mn, mx, vs = oracle.build_scene(tris, 20, 20, 20)
    rng = np.random.default_rng(8)
    off, idx = sc.offsets.astype(np.int64), sc.indices.copy()
    new_idx, new_off = [], [0]
    for v in range(off.size - 1):
        seg = idx[off[v]:off[v + 1]]
        seg = seg[rng.random(seg.size) < 0.8]
        rng.shuffle(seg)
        new_idx.append(seg)
        new_off.append(new_off[-1] + seg.size)
    nidx = np.concatenate(new_idx).astype(np.uint32)
    noff = np.asarray(new_off, np.uint32)
    sc2 = oracle.OracleScene(tris, 20, 20, 20, mn, vs, noff, nidx)
    ctx.set_mesh(tris)
    ctx.set_grid_csr(20, 20, 20, mn, mx, vs, noff, nidx)
    pos, la, up, fov, asp = capi.default_camera(mn, mx)
    fr = capi.make_camera_frame(pos, la, up, fov, asp)
    assert_frames_equal(render_both(ctx, fr, 200, 150), sc2.render(fr, 200, 150, threads=8))
    bad = nidx.copy()
    bad[5] = tris.shape[0] + 7
    with pytest.raises(capi.HvrtError):
        ctx.set_grid_csr(20, 20, 20, mn, mx, vs, noff, bad)


@pytest.mark.parametrize("inline_records", ["1", "0"])
def test_ingested_csr_is_validated_in_both_record_modes(monkeypatch, inline_records):
    """hvrt_set_grid_csr(_device) checks the caller's arrays before anything is derived from them -- offsets[0] == 0,
    offsets never decrease, offsets[n_voxels] == n_refs, every index < n_tris -- with per-reference records
    (HVRT_INLINE_RECORDS=1, the default) and in the indexed mode (=0), whose traversal reads tri_rec[index] directly.
    The indexed mode's frames are the oracle's too (it runs the per-lane walk, not the wave-synchronous one)."""
    import raytracer_amd as ra
    import torch
    monkeypatch.setenv("HVRT_INLINE_RECORDS", inline_records)
    own = ra.Context(0)
    try:
        tris = meshgen.skewed_statue(12, 60)
        sc, mn, mx, vs = oracle.build_scene(tris, 16, 16, 16)
        own.set_mesh(tris)
        own.set_grid_csr(16, 16, 16, mn, mx, vs, sc.offsets, sc.indices)
        for view in ("front", "side"):
            fr = capi.make_camera_frame(*capi.default_camera(mn, mx, view))
            assert_frames_equal(render_both(own, fr, 200, 150), sc.render(fr, 200, 150, threads=8))
        o = np.array([[0.1, 0.2, -4.0], [3.0, 0.1, 0.2]], np.float32)
        d = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0]], np.float32)
        got, gcnt = own.trace(o, d)
        exp, ecnt = sc.trace(o, d)
        assert got.tobytes() == exp.tobytes() and np.array_equal(gcnt, ecnt)
        # a grid built on the GPU in this mode renders the same frame
        own.build_grid(16, 16, 16, mn, mx, vs)
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx))
        assert_frames_equal(render_both(own, fr, 200, 150), sc.render(fr, 200, 150, threads=8))

        def rejected(off, idx, match):
            with pytest.raises(capi.HvrtError, match=match):
                own.set_grid_csr(16, 16, 16, mn, mx, vs, off, idx)
            with pytest.raises(capi.HvrtError):          # the failed call leaves the context without a grid
                own.grid_info()

        bad = sc.indices.copy(); bad[len(bad) // 2] = tris.shape[0]
        rejected(sc.offsets, bad, "triangle index")
        bad = sc.indices.copy(); bad[-1] = 0xFFFFFFFF
        rejected(sc.offsets, bad, "triangle index")
        occupied = np.flatnonzero(np.diff(sc.offsets.astype(np.int64)) > 0)
        off = sc.offsets.copy(); off[occupied[3] + 1] = off[occupied[3]] - 1 if off[occupied[3]] else 0xFFFFFFF0
        off[-1] = sc.offsets[-1]
        rejected(off, sc.indices, "offsets are malformed")
        off = sc.offsets.copy(); off[0] = 1
        rejected(off, sc.indices, "offsets are malformed")
        # device form: n_refs comes from the caller and must be offsets[n_voxels]
        d_off = torch.from_numpy(sc.offsets.view(np.int32)).to("cuda:0")
        d_idx = torch.from_numpy(sc.indices.view(np.int32)).to("cuda:0")
        torch.cuda.synchronize()
        with pytest.raises(capi.HvrtError, match="offsets are malformed"):
            own.set_grid_csr_device(16, 16, 16, mn, mx, vs, d_off.data_ptr(), d_idx.data_ptr(), int(sc.offsets[-1]) - 1)
        own.set_grid_csr_device(16, 16, 16, mn, mx, vs, d_off.data_ptr(), d_idx.data_ptr(), int(sc.offsets[-1]))
        assert_frames_equal(render_both(own, fr, 200, 150), sc.render(fr, 200, 150, threads=8))
    finally:
        own.close()
