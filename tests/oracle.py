"""ctypes access to the checker: oracle/liboracle.so (our CPU restatement) and, where it has been built,
oracle/_ref/libref.so (the real reference functions).  Test infrastructure only -- nothing under
raytracer_amd/ imports this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libref.so")


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("origin", V3), ("lower_left", V3), ("horizontal", V3), ("vertical", V3)]


class Scene(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("min", V3), ("vs", V3),
                ("tris", C.c_void_p), ("n_tris", C.c_uint32), ("offsets", C.c_void_p), ("indices", C.c_void_p),
                ("max", V3), ("mode", C.c_int)]


HIT_DTYPE = np.dtype([("hit", "<u4"), ("t", "<f4"), ("point", "<f4", 3), ("origin", "<f4", 3),
                      ("normal", "<f4", 3), ("triangle_index", "<u4"), ("voxel_index", "<u4")])

_lib = None
_ref = None


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        l = C.CDLL(LIB)
        l.orc_voxel_size.restype = C.c_float
        l.orc_voxelize.restype = C.c_uint64
        l.orc_voxelize_sat.restype = C.c_uint64
        l.orc_triangle_box_overlap.restype = C.c_int
        l.orc_ray_triangle.restype = C.c_int
        l.orc_ray_box.restype = C.c_int
        _lib = l
    return _lib


def have_ref() -> bool:
    return os.path.exists(REF_LIB)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        r = C.CDLL(REF_LIB)
        r.ref_stl_open.restype = C.c_void_p
        r.ref_stl_open.argtypes = [C.c_char_p]
        r.ref_stl_error.restype = C.c_char_p
        r.ref_stl_error.argtypes = [C.c_void_p]
        r.ref_stl_num_vrts.restype = C.c_ulonglong
        r.ref_stl_num_vrts.argtypes = [C.c_void_p]
        r.ref_stl_num_tris.restype = C.c_ulonglong
        r.ref_stl_num_tris.argtypes = [C.c_void_p]
        r.ref_stl_copy.argtypes = [C.c_void_p] * 4
        r.ref_stl_close.argtypes = [C.c_void_p]
        r.ref_dot.restype = C.c_float
        _ref = r
    return _ref


# ---------------------------------------------------------------- restatement wrappers

def mesh_bounds(points: np.ndarray):
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_mesh_bounds(_p(p), C.c_uint64(p.shape[0]), _p(mn), _p(mx))
    return mn, mx


def voxel_size(mn, mx, nx, ny, nz) -> np.float32:
    mn = np.ascontiguousarray(mn, np.float32)
    mx = np.ascontiguousarray(mx, np.float32)
    return np.float32(lib().orc_voxel_size(_p(mn), _p(mx), nx, ny, nz))


def voxelize(tri_aos: np.ndarray, nx, ny, nz, mn, vs3, sat=False):
    """sat: the separating-axis option (not the reference's rule; orc_voxelize_sat)."""
    t = np.ascontiguousarray(tri_aos, np.float32).reshape(-1, 12)
    mn = np.ascontiguousarray(mn, np.float32)
    vs3 = np.ascontiguousarray(np.broadcast_to(np.asarray(vs3, np.float32), (3,)))
    nvox = nx * ny * nz
    off = np.zeros(nvox + 1, np.uint32)
    fn = lib().orc_voxelize_sat if sat else lib().orc_voxelize
    n = fn(_p(t), C.c_uint32(t.shape[0]), nx, ny, nz, _p(mn), _p(vs3), _p(off), None)
    idx = np.zeros(n, np.uint32)
    n2 = fn(_p(t), C.c_uint32(t.shape[0]), nx, ny, nz, _p(mn), _p(vs3), _p(off), _p(idx))
    assert n == n2 == int(off[-1])
    return off, idx


def camera_frame(pos, look_at, up, fov, aspect) -> Frame:
    f = Frame()
    a = [np.ascontiguousarray(x, np.float32) for x in (pos, look_at, up)]
    lib().orc_camera_frame_make(_p(a[0]), _p(a[1]), _p(a[2]), C.c_float(fov), C.c_float(aspect), C.byref(f))
    return f


def frame_to_array(f) -> np.ndarray:
    """Either an oracle Frame or a capi.CameraFrame -> 12 floats."""
    return np.frombuffer(bytes(f), np.float32).copy()


class OracleScene:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, tri_aos, nx, ny, nz, mn, vs3, offsets, indices, mx=None, mode=0):
        self.tris = np.ascontiguousarray(tri_aos, np.float32).reshape(-1, 12)
        self.offsets = np.ascontiguousarray(offsets, np.uint32)
        self.indices = np.ascontiguousarray(indices, np.uint32)
        vs3 = np.broadcast_to(np.asarray(vs3, np.float32), (3,))
        self.c = Scene(nx, ny, nz, V3(*[float(x) for x in mn]), V3(*[float(x) for x in vs3]),
                       self.tris.ctypes.data, self.tris.shape[0], self.offsets.ctypes.data,
                       self.indices.ctypes.data if self.indices.size else None,
                       V3(*[float(x) for x in (mx if mx is not None else (0, 0, 0))]), mode)

    def set_mode(self, mode: int, mx=None):
        """0 = the reference's traversal; 1 = nearest-hit mode (needs the grid's maxBound)."""
        if mx is not None:
            self.c.max = V3(*[float(x) for x in mx])
        self.c.mode = mode

    def render(self, frame, W, H, spp=1, seed=0, threads=8, rows=None):
        """Full-frame arrays; rows=(y0,y1) limits the work (other rows stay zero)."""
        f = Frame.from_buffer_copy(bytes(frame))
        rgb = np.zeros((H, W, 3), np.float32)
        hit = np.zeros((H, W), np.uint32)
        t = np.zeros((H, W), np.float32)
        y0, y1 = rows if rows else (0, H)
        threads = max(1, min(threads, y1 - y0))
        bands = np.linspace(y0, y1, threads + 1).astype(int)
        cnts = [np.zeros(2, np.uint64) for _ in range(threads)]

        def work(k):
            lib().orc_render_rows(C.byref(self.c), C.byref(f), W, H, int(bands[k]), int(bands[k + 1]), spp,
                                  C.c_uint32(seed), _p(rgb), _p(hit), _p(t), _p(cnts[k]))

        if threads == 1:
            work(0)
        else:
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(work, range(threads)))
        return rgb, hit, t, np.sum(cnts, axis=0).astype(np.uint64)

    def trace(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros(o.shape[0], HIT_DTYPE)
        cnt = np.zeros(2, np.uint64)
        l = lib()
        for i in range(o.shape[0]):
            l.orc_trace(C.byref(self.c), C.c_void_p(o.ctypes.data + 12 * i), C.c_void_p(d.ctypes.data + 12 * i),
                        C.c_void_p(out.ctypes.data + 52 * i), _p(cnt))
        return out, cnt


def build_scene(tri_aos, nx, ny, nz, points=None):
    """bounds -> voxel size -> CSR, as BuildVoxelGridFromStlMesh does (SceneCreator.cpp:308-329).
    `points` = the vertex list the bounds are taken over (default: the triangle corners)."""
    t = np.ascontiguousarray(tri_aos, np.float32).reshape(-1, 12)
    pts = t[:, :9].reshape(-1, 3) if points is None else points
    mn, mx = mesh_bounds(pts)
    vs = voxel_size(mn, mx, nx, ny, nz)
    off, idx = voxelize(t, nx, ny, nz, mn, vs)
    return OracleScene(t, nx, ny, nz, mn, vs, off, idx), mn, mx, vs


def fnv1a64(a: np.ndarray) -> int:
    """FNV-1a over 32-bit words (same fold as SURVEY's probe used for CSR / hit-id checksums)."""
    h = 1469598103934665603
    for v in np.ascontiguousarray(a).view(np.uint32).ravel().tolist():
        h ^= v
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


# ---------------------------------------------------------------- real reference (oracle/_ref)

def ref_stl_load(path: str):
    r = ref()
    h = C.c_void_p(r.ref_stl_open(path.encode()))
    try:
        err = r.ref_stl_error(h).decode()
        if err:
            raise RuntimeError(err)
        nv, nt = r.ref_stl_num_vrts(h), r.ref_stl_num_tris(h)
        co = np.zeros((nv, 3), np.float32)
        ti = np.zeros((nt, 3), np.uint32)
        nr = np.zeros((nt, 3), np.float32)
        r.ref_stl_copy(h, _p(co), _p(ti), _p(nr))
    finally:
        r.ref_stl_close(h)
    return co, ti, nr


def ref_ray_triangle_batch(tris, dirs, orgs):
    n = tris.shape[0]
    tris = np.ascontiguousarray(tris, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    orgs = np.ascontiguousarray(orgs, np.float32)
    hit = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    ref().ref_ray_triangle_batch(_p(tris), _p(dirs), _p(orgs), C.c_uint(n), _p(hit), _p(t))
    return hit, t
