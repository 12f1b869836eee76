"""The HIP path against values the REFERENCE itself produced -- no oracle in between.

The two central functions of the path, voxelizeTrianglesDirect (SceneCreator.cpp:197-281) and traverseVoxelGrid with
initializeTraversal / advanceToNextVoxel / testVoxelTriangles (OctreeLikeRayTracer.cpp:155-336), include Logger.h ->
spdlog and cannot be compiled in this image, so no golden vector can be generated from them here.  What exists are
the outputs of the real reference recorded by the survey session on two sin/cos
spheres; the meshes are rebuilt bit for bit from that machine's sin/cos tables (tests/golden/uv_sphere_trig.npz).
tests/test_oracle_pinning.py checks the oracle against the same numbers; here libhvrt is, through the C ABI."""
import numpy as np
import pytest

from raytracer_amd import capi
from test_oracle_pinning import RECORDED, recorded_mesh

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


def _scene(ctx, which):
    r = RECORDED[which]
    tris, verts = recorded_mesh(which)
    assert tris.shape[0] == r["tris"]
    ctx.set_mesh(tris)
    # the reference takes its bounds from the de-duplicated vertex list (computeMeshBounds, SceneCreator.cpp:49-84)
    mn, mx = ctx.compute_bounds(verts)
    g = r["grid"]
    vs = capi.voxel_size(mn, mx, g, g, g)
    info = ctx.build_grid(g, g, g, mn, mx, vs)
    return r, info, mn, mx


def test_appendix_c_recorded_reference_outputs_on_the_gpu(ctx):
    r, info, mn, mx = _scene(ctx, "appendix_c")
    assert info.n_refs == 144331 and info.n_occupied == 14960 and info.max_list == 504     # "CSR: refs 144 331, occupied 14 960 / 125 000, longest list 504"
    off, idx = ctx.get_csr()
    cnt = np.diff(off.astype(np.int64))
    assert off[0] == 0 and off[-1] == 144331 and int((cnt > 0).sum()) == 14960 and int(cnt.max()) == 504
    for v in np.flatnonzero(cnt > 1)[:3000]:                                     # F3: strictly ascending lists
        assert (np.diff(idx[off[v]:off[v + 1]].astype(np.int64)) > 0).all()
    W, H = r["W"], r["H"]
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front"))          # aspect 800/600, TraceImages.cpp:143-144
    rgb, hit, t, c = ctx.render(fr, W, H)
    assert int((hit != MISS).sum()) == 35137                 # "hits 35 137 / 480 000 (7.32 %)"
    assert abs(c[0] / (W * H) - 37.9) < 0.05                 # "37.9 voxels/ray"
    assert abs(c[1] / (W * H) - 111.1) < 0.05                # "111.1 tri-tests/ray"
    assert hit[320, 400] == 18100 and hit[320, 401] == 18099  # px(400,320), px(401,320)
    assert hit[300, 410] == MISS and hit[300, 300] == MISS    # F7: dir.y == +0 misses inside the silhouette
    assert hit[300, 400] != MISS
    assert abs(float(rgb[300, 400, 0]) - 1.09975) < 5e-6     # "probe centre pixel = 1.09975"
    # the probe pixels as single rays through hvrt_trace: voxels visited and tests as the reference counted them
    o = np.array(fr.origin, np.float32)
    ll, hor, ver = (np.array(v, np.float32) for v in (fr.lower_left, fr.horizontal, fr.vertical))

    def probe(x, y):
        u, v = np.float32(x) / np.float32(W), np.float32(y) / np.float32(H)
        p = ((ll + hor * u).astype(np.float32) + (ver * v).astype(np.float32)).astype(np.float32) - o
        n = np.sqrt((p[0] * p[0] + p[1] * p[1]).astype(np.float32) + p[2] * p[2]).astype(np.float32)
        out, cc = ctx.trace(o[None], (p / n).astype(np.float32)[None])
        return out[0], cc

    h, cc = probe(400, 300)
    assert h["hit"] == 1 and cc[0] == 1 and cc[1] == 18       # "hit, 1 voxel, 18 tests"
    h, cc = probe(400, 320)
    assert h["triangle_index"] == 18100 and cc[0] == 4        # "4 voxels"
    h, cc = probe(401, 320)
    assert h["triangle_index"] == 18099 and cc[0] == 5        # "5 voxels (spurious first X step)"
    h, cc = probe(410, 300)
    assert h["hit"] == 0 and cc[0] == 50                      # "MISS inside silhouette, 50 voxels (X never steps)"


def test_section_6_second_recorded_reference_run_on_the_gpu(ctx):
    r, info, mn, mx = _scene(ctx, "section_6")
    assert info.n_refs == 3617504                             # "1.0 M tris -> 256^3: 3 617 504 refs"
    W, H = r["W"], r["H"]
    fr = capi.make_camera_frame(*capi.default_camera(mn, mx, "front"))
    rgb, hit, t, c = ctx.render(fr, W, H)
    assert abs(c[0] / (W * H) - 192.6) < 0.05                 # "192.6 voxels/ray"
    assert abs(c[1] / (W * H) - 226.4) < 0.05                 # "226.4 tri-tests/ray"
    assert (hit != MISS).any()
