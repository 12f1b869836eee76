"""The C-ABI group layer (include/hvrt.h: hvrt_group_*): one process, one context per device, the frame's row tiles
interleaved over the members; host frames are copied out by the members themselves (each its own tiles, straight into
the caller's arrays), device-resident frames are gathered on member 0.

This box has one GPU, so the members of the multi-member cases are several contexts on that one device (the copy
transport: device-to-device copies ordered with events; a communicator cannot hold a device twice).  What that pins is
the choreography -- partition, per-member launch, slab placement, reassembly, summed statistics -- bit for bit against
one context.  The RCCL transport's own calls are exercised where one GPU allows it: a one-rank communicator
(ncclCommInitAll over {0}) and the in-place scene broadcast on it."""
import os

import numpy as np
import pytest

from raytracer_amd import capi, meshgen

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


def _scene(c, tris, g):
    c.set_mesh(tris)
    mn, mx = c.mesh_bounds()
    vs = capi.voxel_size(mn, mx, g, g, g)
    c.build_grid(g, g, g, mn, mx, vs)
    return mn, mx


@pytest.mark.parametrize("members,size", [(1, (333, 211)), (2, (640, 360)), (3, (333, 211)), (5, (200, 70)), (8, (512, 100))])
def test_group_render_equals_one_context(ctx, members, size):
    W, H = size
    tris = meshgen.skewed_statue(24, 60)
    mn, mx = _scene(ctx, tris, 40)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
    rgb, hit, t, cnt = ctx.render(fr, W, H)
    grp = capi.Group([0] * members)
    try:
        assert grp.size == members and grp.transport == ("none" if members == 1 else "copy")
        _scene(grp.root, tris, 40)
        grp.replicate()
        for spp, seed in ((1, 0), (3, 7)):
            if spp > 1:
                rgb, hit, t, cnt = ctx.render(fr, W, H, spp=spp, seed=seed)
            grgb, gcnt, ghits = grp.render(fr, W, H, spp=spp, seed=seed)
            assert np.array_equal(grgb.view(np.uint32), rgb.view(np.uint32))
            assert np.array_equal(gcnt, cnt)
            assert ghits == int((hit != MISS).sum())
    finally:
        grp.close()


@pytest.mark.parametrize("members", [2, 3, 8])
def test_group_mode_and_hit_ids_cover_every_member(ctx, members):
    """Nearest-hit mode set on member 0 alone (what a caller holding only that context does), or through
    hvrt_group_set_mode, reaches every member: the frame, the hit ids and t are one context's in that mode -- not a frame
    striped by member (round 2's bug).  The two modes differ on this scene, so a stripe would show."""
    W, H = 333, 211
    tris = meshgen.skewed_statue(24, 60)
    mn, mx = _scene(ctx, tris, 40)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
    want = {}
    try:
        for mode in (0, 1):
            ctx.set_mode(mode)
            want[mode] = ctx.render(fr, W, H)
    finally:
        ctx.set_mode(0)
    assert not np.array_equal(want[0][1], want[1][1])
    grp = capi.Group([0] * members)
    try:
        _scene(grp.root, tris, 40)
        grp.replicate()
        for how, mode in (("root", 1), ("root", 0), ("group", 1), ("group", 0)):
            if how == "root":
                grp.root.set_mode(mode)
            else:
                grp.set_mode(mode)
                assert all(grp.member(i).get_mode() == mode for i in range(members))
            rgb, hit, t, cnt, hits = grp.render(fr, W, H, want_hit=True, want_t=True)
            assert all(grp.member(i).get_mode() == mode for i in range(members))
            wrgb, whit, wt, wcnt = want[mode]
            assert np.array_equal(hit, whit), (how, mode)
            assert np.array_equal(rgb.view(np.uint32), wrgb.view(np.uint32)) and np.array_equal(t.view(np.uint32), wt.view(np.uint32))
            assert np.array_equal(cnt, wcnt) and hits == int((whit != MISS).sum())
            assert grp.last_exchange_bytes == 0.0          # host frames: nothing travels between the devices
    finally:
        grp.close()


@pytest.mark.parametrize("members,size", [(2, (1920, 1083)), (3, (2048, 1030)), (8, (3840, 2160))])
def test_group_host_frames_through_the_pinned_ring(ctx, members, size):
    """Frames large enough that every member's copy-out takes the ring of pinned chunks, with a last tile of fewer than
    16 rows: each member writes its own row tiles into the caller's arrays, and the arrays are one context's."""
    W, H = size
    tris = meshgen.cube_sphere(40)
    mn, mx = _scene(ctx, tris, 48)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
    wrgb, whit, wt, wcnt = ctx.render(fr, W, H)
    grp = capi.Group([0] * members)
    try:
        _scene(grp.root, tris, 48)
        grp.replicate()
        for _ in range(2):             # the second launch of the shape runs with tile history
            rgb, hit, t, cnt, hits = grp.render(fr, W, H, want_hit=True, want_t=True)
            assert np.array_equal(rgb.view(np.uint32), wrgb.view(np.uint32))
            assert np.array_equal(hit, whit) and np.array_equal(t.view(np.uint32), wt.view(np.uint32))
            assert np.array_equal(cnt, wcnt) and grp.last_exchange_bytes == 0.0
        tm = grp.timings()
        assert tm.create_ms > 0 and tm.replicate_ms > 0 and tm.render_ms > 0 and tm.total_ms >= tm.download_ms > 0
    finally:
        grp.close()


@pytest.mark.parametrize("members", [1, 3])
def test_group_device_resident_frame_is_gathered_on_member_0(ctx, members):
    W, H = 640, 360
    tris = meshgen.skewed_statue(24, 60)
    mn, mx = _scene(ctx, tris, 40)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front", aspect=W / H))
    wrgb, whit, wt, wcnt = ctx.render(fr, W, H)
    grp = capi.Group([0] * members)
    try:
        _scene(grp.root, tris, 40)
        grp.replicate()
        d_rgb, cnt, hits = grp.render_device(fr, W, H)
        rgb = grp.root.download(d_rgb, np.empty((H, W, 3), np.float32))
        assert np.array_equal(rgb.view(np.uint32), wrgb.view(np.uint32)) and np.array_equal(cnt, wcnt)
        assert hits == int((whit != MISS).sum())
        moved = sum(capi.part_rows(H, 16, i, members) for i in range(1, members)) * W * 12
        assert grp.last_exchange_bytes == float(moved)
    finally:
        grp.close()


def test_download_blocks_scatters_a_compact_run(ctx):
    """hvrt_download_blocks: block j of the device run lands j strides into the destination, the last block may be short;
    small runs are copied directly, large ones through the pinned ring (chunks that straddle block boundaries)."""
    import torch
    rng = np.random.default_rng(5)
    for n_bytes, block, stride in ((1000, 96, 200), (5 << 20, 100_000, 250_000), (40 << 20, 3 << 20, (3 << 20) + 4096), (333, 4096, 0)):
        src = rng.integers(0, 256, n_bytes, dtype=np.uint8)
        d = torch.from_numpy(src).cuda()
        torch.cuda.synchronize()
        n_blocks = -(-n_bytes // block)
        out = np.full(max(stride, block) * n_blocks + 64, 0xEE, np.uint8)
        ctx.download_blocks(d.data_ptr(), n_bytes, out, block, stride if block < n_bytes else block, dst_offset=32)
        want = np.full_like(out, 0xEE)
        for j in range(n_blocks):
            piece = src[j * block:(j + 1) * block]
            want[32 + j * stride:32 + j * stride + piece.size] = piece
        assert np.array_equal(out, want), (n_bytes, block, stride)
    with pytest.raises(capi.HvrtError, match="overlap"):
        ctx.download_blocks(d.data_ptr(), 1000, out, 100, 50)


def test_clear_scene_returns_a_context_to_its_defaults(ctx):
    c = capi.Context(0)
    try:
        tris = meshgen.cube_sphere(10)
        mn, mx = _scene(c, tris, 16)
        c.set_mode(1)
        c.clear_scene()
        assert c.get_mode() == 0
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx))
        with pytest.raises(capi.HvrtError, match="no grid|no mesh"):
            c.render(fr, 64, 48)
        _scene(c, tris, 16)            # and it takes a scene again
        _scene(ctx, tris, 16)
        a, b = c.render(fr, 64, 48), ctx.render(fr, 64, 48)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    finally:
        c.close()


def test_group_errors_are_reported(ctx):
    with pytest.raises(capi.HvrtError):
        capi.Group([])
    grp = capi.Group([0, 0])
    try:
        with pytest.raises(capi.HvrtError, match="no mesh|no grid|member 0"):
            grp.replicate()
    finally:
        grp.close()
    old = os.environ.get("HVRT_GROUP_TRANSPORT")
    os.environ["HVRT_GROUP_TRANSPORT"] = "rccl"
    try:
        with pytest.raises(capi.HvrtError, match="twice"):
            capi.Group([0, 0])
    finally:
        if old is None:
            del os.environ["HVRT_GROUP_TRANSPORT"]
        else:
            os.environ["HVRT_GROUP_TRANSPORT"] = old


def test_rccl_transport_on_a_one_rank_communicator(ctx):
    """librccl is found and opened at run time, ncclCommInitAll gives a communicator for device 0, the scene broadcast
    (in place on the only rank) goes through it, and the frame is the one a plain context renders."""
    tris = meshgen.cube_sphere(20)
    mn, mx = _scene(ctx, tris, 24)
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front"))
    rgb, hit, t, cnt = ctx.render(fr, 320, 240)
    old = os.environ.get("HVRT_GROUP_TRANSPORT")
    os.environ["HVRT_GROUP_TRANSPORT"] = "rccl"
    try:
        grp = capi.Group([0])
    finally:
        if old is None:
            del os.environ["HVRT_GROUP_TRANSPORT"]
        else:
            os.environ["HVRT_GROUP_TRANSPORT"] = old
    try:
        assert grp.transport == "rccl"
        _scene(grp.root, tris, 24)
        grp.replicate()
        grgb, gcnt, ghits = grp.render(fr, 320, 240)
        assert np.array_equal(grgb.view(np.uint32), rgb.view(np.uint32)) and np.array_equal(gcnt, cnt)
    finally:
        grp.close()
