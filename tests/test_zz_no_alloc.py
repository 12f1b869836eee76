"""hvrt_render_part_device_noalloc: the one entry point that allocates nothing (fixed scratch inside libhvrt).

It has no counterpart in the reference; it exists for callers that cannot call hipMalloc.  The bar is the one of
every other GPU test: its outputs are bit for bit those of hvrt_render_part_device, the entry point it is a
variant of (which the other tests compare with the oracle) -- on two calls in a row, the second with a smaller
image, so that what the first call left in the scratch array (queue head, tile durations, camera frame) is
seen to be cleared or overwritten.  Collected last on purpose (file name)."""
import numpy as np
import pytest

from raytracer_amd import capi, meshgen

pytestmark = pytest.mark.gpu


def _render(ctx, entry, fr, W, H, spp, seed, tile_rows, part, n_parts):
    import torch
    rows = capi.part_rows(H, tile_rows, part, n_parts)
    dev = torch.device("cuda:0")
    rgb = torch.full((rows, W, 3), -7.0, dtype=torch.float32, device=dev)
    hit = torch.full((rows, W), 12345, dtype=torch.int32, device=dev)
    t = torch.full((rows, W), -7.0, dtype=torch.float32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    entry(fr, W, H, spp, seed, tile_rows, part, n_parts, d_rgb=rgb.data_ptr(), d_hit=hit.data_ptr(),
          d_t=t.data_ptr(), d_counters=cnt.data_ptr())
    ctx.synchronize()
    return rgb.cpu().numpy(), hit.cpu().numpy(), t.cpu().numpy(), cnt.cpu().numpy()


def _same(a, b):
    for x, y, what in zip(a, b, ("rgb", "hit id", "t", "counters")):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                              y.view(np.uint32) if y.dtype == np.float32 else y), what


def test_noalloc_render_equals_render_part_device_twice_in_a_row(ctx):
    tris = meshgen.skewed_statue(24, 120)          # long lists: the existing entry point splits tiles, this one not
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, 32, 32, 32)
    ctx.build_grid(32, 32, 32, mn, mx, vs)
    # first a large image, then smaller ones (whole frame, a partition's part, several samples per pixel)
    calls = [(640, 360, 1, 0, 360, 0, 1), (203, 157, 1, 0, 8, 1, 3), (96, 64, 3, 5, 64, 0, 1), (1, 1, 1, 0, 1, 0, 1)]
    for (W, H, spp, seed, tile_rows, part, n_parts) in calls:
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx, aspect=W / H))
        want = _render(ctx, ctx.render_part_device, fr, W, H, spp, seed, tile_rows, part, n_parts)
        got = _render(ctx, ctx.render_part_device_noalloc, fr, W, H, spp, seed, tile_rows, part, n_parts)
        _same(got, want)
        assert (want[1] != -1).any() or W == 1
    # nearest-hit mode goes through the same entry point
    try:
        ctx.set_mode(1)
        fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "side", aspect=4 / 3))
        _same(_render(ctx, ctx.render_part_device_noalloc, fr, 160, 120, 1, 0, 120, 0, 1),
              _render(ctx, ctx.render_part_device, fr, 160, 120, 1, 0, 120, 0, 1))
    finally:
        ctx.set_mode(0)


def test_noalloc_render_rejects_what_does_not_fit_its_scratch(ctx):
    tris = meshgen.cube_sphere(8)
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    ctx.build_grid(8, 8, 8, mn, mx, capi.voxel_size(mn, mx, 8, 8, 8))
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx))
    # 16384 x 16384 pixels = 4 Mi tiles > 2^20: refused before anything is touched (the pointers are never used)
    with pytest.raises(capi.HvrtError, match="exceed the fixed scratch"):
        ctx.render_part_device_noalloc(fr, 16384, 16384, 1, 0, 16384, 0, 1, d_rgb=16)
    # ... but one eighth of it fits (8 parts) -- only checked for acceptance of the size, not rendered: a null
    # output set renders nothing visible, so use a small frame for the positive case instead
    got = _render(ctx, ctx.render_part_device_noalloc, fr, 64, 48, 1, 0, 48, 0, 1)
    _same(got, _render(ctx, ctx.render_part_device, fr, 64, 48, 1, 0, 48, 0, 1))
