"""The oracle's separating-axis voxelizer (oracle/cpu_ref.c: orc_voxelize_sat, the checker of hvrt_set_overlap_test(ctx, 1)).
PARITY UNPINNED: the reference declares triangleIntersectsBox (GeometryUtils.h:19) and defines it nowhere, and its
voxelizer uses the bounding box alone -- there is no reference output to compare with.  What can be checked without one:
the rule never drops a voxel that holds a point of the triangle (dense sampling), and never adds one to the box rule's list."""
import numpy as np

import oracle
from raytracer_amd import meshgen


def _lists(off, idx, T):
    nvox = off.size - 1
    vox = np.repeat(np.arange(nvox, dtype=np.int64), np.diff(off.astype(np.int64)))
    return vox * T + idx


def test_sat_rule_is_conservative_and_inside_the_box_rule():
    tris = np.ascontiguousarray(meshgen.random_soup(1500, 9), np.float32)        # triangles of all sizes and directions
    g = 20
    mn, mx = oracle.mesh_bounds(tris[:, :9].reshape(-1, 3))
    vs = oracle.voxel_size(mn, mx, g, g, g)
    so, si = oracle.voxelize(tris, g, g, g, mn, vs, sat=True)
    bo, bi = oracle.voxelize(tris, g, g, g, mn, vs)
    T = tris.shape[0]
    sat, box = _lists(so, si, T), _lists(bo, bi, T)
    assert sat.size < box.size and np.isin(sat, box).all()
    listed = set(sat.tolist())
    rng = np.random.default_rng(3)
    checked = 0
    for t in range(0, T, 5):
        v = tris[t, :9].reshape(3, 3).astype(np.float64)
        r = rng.random((1500, 2))
        m = r.sum(1) > 1
        r[m] = 1 - r[m]
        p = v[0] + r[:, :1] * (v[1] - v[0]) + r[:, 1:] * (v[2] - v[0])
        cell = np.floor((p - mn.astype(np.float64)) / float(vs)).astype(np.int64)
        ins = ((cell >= 0) & (cell < g)).all(1)
        for f in np.unique((cell[:, 0] + g * (cell[:, 1] + g * cell[:, 2]))[ins]):
            checked += 1
            assert int(f) * T + t in listed, f"voxel {int(f)} holds a point of triangle {t} and does not list it"
    assert checked > 2000


def test_sat_rule_on_degenerate_and_axis_aligned_triangles():
    """Zero-area triangles (every axis test is 0 > 0: not separated) keep their bounding-box voxels; a box whose faces lie
    exactly on voxel planes keeps both voxels either side of a face (the voxel is grown by 2^-10 of its half size)."""
    pts = np.array([[[0.1, 0.1, 0.1], [0.9, 0.9, 0.9], [0.5, 0.5, 0.5]],          # collinear
                    [[0.3, 0.3, 0.3], [0.3, 0.3, 0.3], [0.3, 0.3, 0.3]],          # a point
                    [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], np.float32)
    tris = meshgen.to_aos(pts)
    mn = np.zeros(3, np.float32)
    so, si = oracle.voxelize(tris, 4, 4, 4, mn, 0.25, sat=True)
    T = 3
    keys = set(_lists(so, si, T).tolist())
    for k in range(4):                                 # the diagonal's voxels list the collinear triangle
        assert (k + 4 * (k + 4 * k)) * T + 0 in keys
    assert (1 + 4 * (1 + 4 * 1)) * T + 1 in keys       # the point's voxel
    assert (0 + 4 * (0 + 4 * 0)) * T + 2 in keys and (3 + 4 * (0 + 4 * 0)) * T + 2 in keys
    assert (3 + 4 * (3 + 4 * 0)) * T + 2 not in keys   # the far corner of the right triangle's bounding box
    box = meshgen.box()
    bmn, bmx = oracle.mesh_bounds(box[:, :9].reshape(-1, 3))
    vs = oracle.voxel_size(bmn, bmx, 8, 8, 8)
    so, si = oracle.voxelize(box, 8, 8, 8, bmn, vs, sat=True)
    bo, bi = oracle.voxelize(box, 8, 8, 8, bmn, vs)
    assert 0 < si.size <= bi.size
    occ = np.diff(so.astype(np.int64)).reshape(8, 8, 8) > 0
    assert occ[0].all() and occ[:, 0].all() and occ[:, :, 0].all()       # the faces on the grid's lower planes
