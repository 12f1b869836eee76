"""Synthetic meshes for BASELINE.json's configs and the parity fixtures.

All generators that feed parity tests are libm-free (only IEEE + - * / sqrt in numpy), so the same
bits come out on every machine; the one exception, ``uv_sphere``, reproduces the sin/cos sphere of
SURVEY.md Appendix C and is used only where those recorded reference numbers are checked.

A mesh is returned as ``tri_aos``: float32 array [T, 12] = v0, v1, v2, normal -- the layout of the
reference's ``Triangle`` (Voxels.h:10-13) and of ``hvrt_set_mesh``.
"""
from __future__ import annotations

import math
import struct

import numpy as np


def face_normals(tris: np.ndarray) -> np.ndarray:
    """Unit normals of [T,3,3] triangles in fp32 (zero for degenerate ones)."""
    e1 = (tris[:, 1] - tris[:, 0]).astype(np.float32)
    e2 = (tris[:, 2] - tris[:, 0]).astype(np.float32)
    n = np.empty_like(e1)
    n[:, 0] = e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1]
    n[:, 1] = e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2]
    n[:, 2] = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    ln = np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2]).astype(np.float32))
    ln = np.where(ln > 0, ln, np.float32(1)).astype(np.float32)
    return (n / ln[:, None]).astype(np.float32)


def to_aos(tris: np.ndarray, normals: np.ndarray | None = None) -> np.ndarray:
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 3, 3)
    if normals is None:
        normals = face_normals(tris)
    out = np.empty((tris.shape[0], 12), np.float32)
    out[:, :9] = tris.reshape(-1, 9)
    out[:, 9:] = normals
    return out


def _normalize_rows(p: np.ndarray, radius: float) -> np.ndarray:
    l2 = (p.astype(np.float64) ** 2).sum(-1).astype(np.float32)
    ln = np.sqrt(l2).astype(np.float32)
    return (p.astype(np.float32) / ln[..., None] * np.float32(radius)).astype(np.float32)


def _grid_patch(p: np.ndarray) -> np.ndarray:
    """[n+1, m+1, 3] point grid -> [2*n*m, 3, 3] triangles (two per quad, row-major)."""
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.stack([a, b, c], axis=2)
    t2 = np.stack([a, c, d], axis=2)
    return np.stack([t1, t2], axis=2).reshape(-1, 3, 3)


def cube_sphere(n: int, radius: float = 1.0, center=(0.0, 0.0, 0.0)) -> np.ndarray:
    """Cube-sphere with n x n quads per face: 12*n^2 triangles, no poles, libm-free.
    n = 289 -> 1 002 252 triangles (config 2), 913 -> 10 002 828, 2041 -> 49 988 172 (config 5)."""
    s = (np.arange(n + 1, dtype=np.float64) * 2.0 / n - 1.0).astype(np.float32)
    u, v = np.meshgrid(s, s, indexing="ij")
    one = np.ones_like(u)
    faces = []
    # (axis carrying +-1, sign); orientation chosen so that normals point outwards
    for axis, sign in ((0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1)):
        comps = [None, None, None]
        comps[axis] = one * np.float32(sign)
        o1, o2 = [(1, 2), (2, 0), (0, 1)][axis]
        if sign > 0:
            comps[o1], comps[o2] = u, v
        else:
            comps[o1], comps[o2] = v, u
        p = np.stack(comps, axis=-1)
        faces.append(_grid_patch(_normalize_rows(p, radius)))
    tris = np.concatenate(faces, axis=0)
    tris = (tris + np.asarray(center, np.float32)).astype(np.float32)
    return to_aos(tris)


def skewed_statue(n_base: int, n_patch: int, radius: float = 1.0, patch_half: float = 0.1) -> np.ndarray:
    """'Scanned-statue-class' mesh with skewed triangle density (config 3): a cube-sphere of 12*n_base^2
    triangles plus a 2*n_patch^2-triangle high-resolution shell over ~1 % of the surface (a
    [-patch_half, patch_half]^2 window of the +z face, lifted by 0.05 % so the surfaces do not
    coincide).  Voxels under the patch hold lists of 10^2..10^4 references."""
    base = cube_sphere(n_base, radius)
    s = (np.arange(n_patch + 1, dtype=np.float64) * (2.0 * patch_half) / n_patch - patch_half).astype(np.float32)
    u, v = np.meshgrid(s, s, indexing="ij")
    p = np.stack([u, v, np.ones_like(u)], axis=-1)
    patch = _grid_patch(_normalize_rows(p, radius * 1.0005))
    return np.concatenate([base, to_aos(patch)], axis=0)


def uv_sphere_soup(nlat: int, nlon: int, radius: float = 1.0, trig=None) -> np.ndarray:
    """The sin/cos sphere of SURVEY.md Appendix C as a triangle soup [2*nlat*nlon, 3, 3] INCLUDING the
    degenerate pole triangles (the STL reader drops those).  libm-dependent unless `trig` is given:
    (sin_th[nlat+1], cos_th[nlat+1], cos_ph[nlon+1], sin_ph[nlon+1]) as float64, e.g. the tables recorded from the
    survey machine's libm in tests/golden/uv_sphere_trig.npz -- every further operation is an IEEE double product
    and a rounding to float32, the same everywhere."""
    if trig is None:
        th = [math.pi * i / nlat for i in range(nlat + 1)]
        ph = [2 * math.pi * j / nlon for j in range(nlon + 1)]
        trig = (np.array([math.sin(t) for t in th]), np.array([math.cos(t) for t in th]),
                np.array([math.cos(p) for p in ph]), np.array([math.sin(p) for p in ph]))
    sin_th, cos_th, cos_ph, sin_ph = (np.asarray(a, np.float64) for a in trig)
    assert sin_th.shape == (nlat + 1,) and cos_ph.shape == (nlon + 1,)
    pts = np.empty((nlat + 1, nlon + 1, 3), np.float32)
    rs = float(radius) * sin_th                                  # (radius * sin(th)) * cos(ph), left to right
    pts[:, :, 0] = (rs[:, None] * cos_ph[None, :]).astype(np.float32)
    pts[:, :, 1] = (float(radius) * cos_th)[:, None].astype(np.float32)
    pts[:, :, 2] = (rs[:, None] * sin_ph[None, :]).astype(np.float32)
    a, b, c, d = pts[:-1, :-1], pts[1:, :-1], pts[1:, 1:], pts[:-1, 1:]
    t1 = np.stack([a, c, b], axis=2)
    t2 = np.stack([a, d, c], axis=2)
    return np.stack([t1, t2], axis=2).reshape(-1, 3, 3)


def uv_sphere_trig_tables(nlat: int, nlon: int):
    """The four libm tables uv_sphere_soup needs (this machine's libm)."""
    th = [math.pi * i / nlat for i in range(nlat + 1)]
    ph = [2 * math.pi * j / nlon for j in range(nlon + 1)]
    return (np.array([math.sin(t) for t in th]), np.array([math.cos(t) for t in th]),
            np.array([math.cos(p) for p in ph]), np.array([math.sin(p) for p in ph]))


def stl_reader_view(soup: np.ndarray):
    """What the reference's STL reader makes of a triangle soup (STLReader.h:451-535): triangles whose corners are
    not three distinct vertices (exact float equality) are dropped, vertices are de-duplicated.  Returns
    (kept triangles as the 12-float AoS with computed normals, unique vertices)."""
    a, b, c = soup[:, 0], soup[:, 1], soup[:, 2]
    keep = ~((a == b).all(1) | (a == c).all(1) | (b == c).all(1))
    return to_aos(soup[keep]), np.unique(soup.reshape(-1, 3), axis=0)


def box(lo=(-1.0, -1.0, -1.0), hi=(1.0, 1.0, 1.0)) -> np.ndarray:
    """Axis-aligned box, 12 triangles, vertices exactly on the bounds."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0],
                  [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]
    tris = []
    for a, b, c, d in quads:
        tris.append([v[a], v[b], v[c]])
        tris.append([v[a], v[c], v[d]])
    return to_aos(np.array(tris, np.float32))


def random_soup(n: int, seed: int, extent: float = 1.0, size: float = 0.2, offset=(0.0, 0.0, 0.0)) -> np.ndarray:
    """n random triangles (uniform centres in [-extent, extent]^3, edge scale `size`)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    t = (c + rng.uniform(-size, size, (n, 3, 3)) + np.asarray(offset)).astype(np.float32)
    return to_aos(t)


def slivers(n: int, seed: int) -> np.ndarray:
    """Long thin triangles spanning many voxels (hits outside the tested voxel)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, 3))
    b = a + rng.uniform(-1.5, 1.5, (n, 3))
    c = a + rng.uniform(-0.02, 0.02, (n, 3))
    return to_aos(np.stack([a, b, c], axis=1).astype(np.float32))


def write_binary_stl(path: str, tris: np.ndarray, normals: np.ndarray | None = None, header: bytes = b"hvrt") -> None:
    """Binary STL: 80-byte header, u32 count, then (normal, v0, v1, v2, u16 attr) per triangle."""
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 3, 3)
    if normals is None:
        normals = face_normals(tris)
    rec = np.zeros(tris.shape[0], dtype=[("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")])
    rec["n"] = normals
    rec["v"] = tris.reshape(-1, 9)
    with open(path, "wb") as f:
        f.write(header[:80].ljust(80, b"\0"))
        f.write(struct.pack("<I", tris.shape[0]))
        f.write(rec.tobytes())


def write_ascii_stl(path: str, tris: np.ndarray, normals: np.ndarray | None = None, name: str = "hvrt") -> None:
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 3, 3)
    if normals is None:
        normals = face_normals(tris)
    with open(path, "w") as f:
        f.write(f"solid {name}\n")
        for t, n in zip(tris, normals):
            f.write(f"  facet normal {n[0]:.9g} {n[1]:.9g} {n[2]:.9g}\n    outer loop\n")
            for v in t:
                f.write(f"      vertex {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
            f.write("    endloop\n  endfacet\n")
        f.write(f"endsolid {name}\n")
