"""The C-ABI group layer (include/hvrt.h: hvrt_group_*): one process, one context per device, the frame's row tiles
interleaved over the members, slabs gathered on member 0.

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
