"""ctypes binding of libhvrt (include/hvrt.h) -- the Python-side mirror of the C ABI.

Nothing here computes anything on the CPU: every method forwards to the HIP library and raises
``HvrtError`` on a non-zero status.  If the shared library is missing the import of
``raytracer_amd`` still works (so that CPU-only tooling can load), but creating a ``Context`` or
calling any entry point raises loudly -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HVRT_LIB") or os.path.join(_HERE, "lib", "libhvrt.so")

HVRT_MISS = 0xFFFFFFFF

# every symbol include/hvrt.h declares (tests check the library exports each of them)
SYMBOLS = [
    "hvrt_create", "hvrt_destroy", "hvrt_last_error", "hvrt_stream", "hvrt_set_stream", "hvrt_synchronize", "hvrt_set_mode", "hvrt_get_timings",
    "hvrt_set_mesh", "hvrt_set_mesh_device", "hvrt_compute_bounds", "hvrt_mesh_bounds", "hvrt_voxel_size",
    "hvrt_build_grid", "hvrt_set_grid_csr", "hvrt_set_grid_csr_device", "hvrt_get_grid_info", "hvrt_get_csr",
    "hvrt_get_distance_field", "hvrt_get_device_arrays", "hvrt_export_scene_device", "hvrt_make_camera_frame", "hvrt_render", "hvrt_render_part_device", "hvrt_render_batch_part_device", "hvrt_quantise_rgb8_device", "hvrt_render_views_rgb8", "hvrt_trace_device",
    "hvrt_part_rows", "hvrt_assemble_parts_device", "hvrt_trace", "hvrt_render_part_device_noalloc", "hvrt_last_hit_count",
    "hvrt_set_deferred_accel", "hvrt_finish_accel", "hvrt_download", "hvrt_device_count", "hvrt_render_history", "hvrt_get_tile_cover", "hvrt_get_tile_costs", "hvrt_set_overlap_test",
    "hvrt_group_create", "hvrt_group_destroy", "hvrt_group_last_error", "hvrt_group_size", "hvrt_group_transport",
    "hvrt_group_member", "hvrt_group_replicate_scene", "hvrt_group_render", "hvrt_group_render_device", "hvrt_group_set_mode",
    "hvrt_group_get_timings", "hvrt_group_last_exchange_bytes", "hvrt_get_mode", "hvrt_clear_scene", "hvrt_download_blocks", "hvrt_get_counter_width",
]


class HvrtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hvrt error {code}: {msg}")
        self.code = code


class CameraFrame(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]


class GridInfo(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("min_bound", C.c_float * 3), ("max_bound", C.c_float * 3), ("voxel_size", C.c_float * 3),
                ("n_voxels", C.c_uint64), ("n_refs", C.c_uint64), ("n_occupied", C.c_uint64),
                ("max_list", C.c_uint32), ("n_triangles", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("build_total_ms", C.c_float), ("build_count_ms", C.c_float), ("build_scan_ms", C.c_float),
                ("build_fill_ms", C.c_float), ("build_sort_ms", C.c_float), ("build_expand_ms", C.c_float),
                ("render_ms", C.c_float),
                ("trace_ms", C.c_float)]


class GroupTimings(C.Structure):
    _fields_ = [("create_ms", C.c_float), ("replicate_ms", C.c_float), ("render_ms", C.c_float),
                ("download_ms", C.c_float), ("total_ms", C.c_float)]


HIT_DTYPE = np.dtype([("hit", "<u4"), ("t", "<f4"), ("point", "<f4", 3), ("origin", "<f4", 3),
                      ("normal", "<f4", 3), ("triangle_index", "<u4"), ("voxel_index", "<u4")])
assert HIT_DTYPE.itemsize == 52

_lib = None


def load_library() -> C.CDLL:
    """Load libhvrt.so (built in-tree by raytracer_amd/csrc/Makefile).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HvrtError(-100, f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If torch is going to be
    # used in this process (tests, bench, RCCL driver) it must be the first to load the runtime, or
    # torch later finds "No HIP GPUs"; libhvrt then binds to the copy already mapped.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, f3 = C.c_void_p, C.POINTER(C.c_float)
    lib.hvrt_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.hvrt_destroy.argtypes = [vp]
    lib.hvrt_destroy.restype = None
    lib.hvrt_last_error.argtypes = [vp]
    lib.hvrt_last_error.restype = C.c_char_p
    lib.hvrt_stream.argtypes = [vp]
    lib.hvrt_stream.restype = vp
    lib.hvrt_set_stream.argtypes = [vp, vp]
    lib.hvrt_synchronize.argtypes = [vp]
    lib.hvrt_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.hvrt_set_mesh.argtypes = [vp, vp, C.c_uint32]
    lib.hvrt_set_mesh_device.argtypes = [vp, vp, C.c_uint32]
    lib.hvrt_compute_bounds.argtypes = [vp, vp, C.c_uint64, f3, f3]
    lib.hvrt_mesh_bounds.argtypes = [vp, f3, f3]
    lib.hvrt_voxel_size.argtypes = [f3, f3, C.c_int, C.c_int, C.c_int]
    lib.hvrt_voxel_size.restype = C.c_float
    lib.hvrt_build_grid.argtypes = [vp, C.c_int, C.c_int, C.c_int, f3, f3, f3]
    lib.hvrt_set_grid_csr.argtypes = [vp, C.c_int, C.c_int, C.c_int, f3, f3, f3, vp, vp]
    lib.hvrt_set_grid_csr_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, f3, f3, f3, vp, vp, C.c_uint64]
    lib.hvrt_get_grid_info.argtypes = [vp, C.POINTER(GridInfo)]
    lib.hvrt_get_csr.argtypes = [vp, vp, vp]
    lib.hvrt_get_distance_field.argtypes = [vp, vp]
    lib.hvrt_get_device_arrays.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.hvrt_export_scene_device.argtypes = [vp, vp, vp, vp]
    lib.hvrt_make_camera_frame.argtypes = [f3, f3, f3, C.c_float, C.c_float, C.POINTER(CameraFrame)]
    lib.hvrt_make_camera_frame.restype = None
    lib.hvrt_render.argtypes = [vp, C.POINTER(CameraFrame), C.c_int, C.c_int, C.c_int, C.c_uint32, vp, vp, vp, vp]
    lib.hvrt_render_part_device.argtypes = [vp, C.POINTER(CameraFrame), C.c_int, C.c_int, C.c_int, C.c_uint32,
                                            C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.hvrt_render_part_device_noalloc.argtypes = lib.hvrt_render_part_device.argtypes
    lib.hvrt_last_hit_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.hvrt_download.argtypes = [vp, vp, vp, C.c_uint64]
    lib.hvrt_render_history.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    lib.hvrt_get_tile_cover.argtypes = [vp, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint32)]
    lib.hvrt_get_tile_costs.argtypes = [vp, vp, C.c_uint64]
    lib.hvrt_group_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    lib.hvrt_group_destroy.argtypes = [vp]
    lib.hvrt_group_destroy.restype = None
    lib.hvrt_group_last_error.argtypes = [vp]
    lib.hvrt_group_last_error.restype = C.c_char_p
    lib.hvrt_group_size.argtypes = [vp]
    lib.hvrt_group_transport.argtypes = [vp]
    lib.hvrt_group_transport.restype = C.c_char_p
    lib.hvrt_group_member.argtypes = [vp, C.c_int]
    lib.hvrt_group_member.restype = vp
    lib.hvrt_group_replicate_scene.argtypes = [vp]
    lib.hvrt_group_render.argtypes = [vp, C.POINTER(CameraFrame), C.c_int, C.c_int, C.c_int, C.c_uint32, vp, vp, vp, vp, vp]
    lib.hvrt_group_render_device.argtypes = [vp, C.POINTER(CameraFrame), C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(vp), vp, vp]
    lib.hvrt_group_set_mode.argtypes = [vp, C.c_int]
    lib.hvrt_group_get_timings.argtypes = [vp, C.POINTER(GroupTimings)]
    lib.hvrt_group_last_exchange_bytes.argtypes = [vp]
    lib.hvrt_group_last_exchange_bytes.restype = C.c_double
    lib.hvrt_get_mode.argtypes = [vp]
    lib.hvrt_get_counter_width.argtypes = [vp]
    lib.hvrt_clear_scene.argtypes = [vp]
    lib.hvrt_download_blocks.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.hvrt_set_deferred_accel.argtypes = [vp, C.c_int]
    lib.hvrt_set_overlap_test.argtypes = [vp, C.c_int]
    lib.hvrt_finish_accel.argtypes = [vp]
    lib.hvrt_render_batch_part_device.argtypes = [vp, C.POINTER(CameraFrame), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.hvrt_quantise_rgb8_device.argtypes = [vp, vp, C.c_uint64, vp]
    lib.hvrt_render_views_rgb8.argtypes = [vp, C.POINTER(CameraFrame), C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.hvrt_part_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.hvrt_assemble_parts_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.hvrt_trace.argtypes = [vp, vp, vp, C.c_uint32, vp, vp]
    lib.hvrt_trace_device.argtypes = [vp, vp, vp, C.c_uint32, vp, vp]
    _lib = lib
    return lib


def _f3(a) -> "C.Array":
    a = np.asarray(a, dtype=np.float32).reshape(3)
    return (C.c_float * 3)(*[float(x) for x in a])


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def make_camera_frame(position, look_at, up, fov_degrees: float, aspect: float) -> CameraFrame:
    """Loop-invariant head of generateRay (OctreeLikeRayTracer.cpp:342-359), computed by libhvrt."""
    lib = load_library()
    fr = CameraFrame()
    lib.hvrt_make_camera_frame(_f3(position), _f3(look_at), _f3(up), C.c_float(fov_degrees), C.c_float(aspect),
                               C.byref(fr))
    return fr


def voxel_size(min_bound, max_bound, nx: int, ny: int, nz: int) -> np.float32:
    lib = load_library()
    return np.float32(lib.hvrt_voxel_size(_f3(min_bound), _f3(max_bound), nx, ny, nz))


def part_rows(height: int, tile_rows: int, part: int, n_parts: int) -> int:
    return load_library().hvrt_part_rows(height, tile_rows, part, n_parts)


def default_camera(min_bound, max_bound, view: str = "front", aspect: float = 800.0 / 600.0):
    """TraceImages::setupCamera / setupCameraWithView (TraceImages.cpp:125-197): position, lookAt, up,
    fov 60, aspect (the reference hard-codes 800/600).  fp32 arithmetic as in the reference."""
    mn = np.asarray(min_bound, np.float32)
    mx = np.asarray(max_bound, np.float32)
    center = ((mn + mx) * np.float32(0.5)).astype(np.float32)      # VoxelGrid.h:30-32
    size = (mx - mn).astype(np.float32)
    dist = np.float32(np.max(size)) * np.float32(2.5)
    pos = center.copy()
    up = np.array([0, 1, 0], np.float32)
    if view == "front":
        pos[2] = center[2] - dist
    elif view == "side":
        pos[0] = center[0] + dist
    elif view == "top":
        pos[1] = center[1] + dist
        up = np.array([0, 0, 1], np.float32)
    else:
        raise ValueError(view)
    return pos, center, up, 60.0, float(np.float32(aspect))


class Group:
    """Several GPUs of one node (include/hvrt.h, group section): member 0 holds the scene the caller sets, replicate()
    hands it to the others, render() traces a frame over all of them."""

    def __init__(self, device_ids):
        self._lib = load_library()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self._lib.hvrt_group_create(C.byref(h), ids, len(device_ids))
        if rc != 0:
            raise HvrtError(rc, self._lib.hvrt_group_last_error(None).decode())
        self._g = h
        self.root = self.member(0)                     # member 0, owned by the group

    def _ck(self, rc):
        if rc != 0:
            raise HvrtError(rc, self._lib.hvrt_group_last_error(self._g).decode())

    def member(self, i: int) -> "Context":
        c = Context.__new__(Context)
        c._lib, c._h, c._borrowed = self._lib, C.c_void_p(self._lib.hvrt_group_member(self._g, i)), True
        return c

    @property
    def size(self) -> int:
        return self._lib.hvrt_group_size(self._g)

    @property
    def transport(self) -> str:
        return self._lib.hvrt_group_transport(self._g).decode()

    def replicate(self):
        self._ck(self._lib.hvrt_group_replicate_scene(self._g))

    def set_mode(self, mode: int):
        self._ck(self._lib.hvrt_group_set_mode(self._g, mode))

    def timings(self) -> GroupTimings:
        t = GroupTimings()
        self._ck(self._lib.hvrt_group_get_timings(self._g, C.byref(t)))
        return t

    @property
    def last_exchange_bytes(self) -> float:
        return float(self._lib.hvrt_group_last_exchange_bytes(self._g))

    def render(self, frame: CameraFrame, width: int, height: int, spp: int = 1, seed: int = 0, want_hit=False, want_t=False):
        """Host frame: every member copies its own row tiles into the arrays returned here."""
        rgb = np.empty((height, width, 3), np.float32)
        hit = np.empty((height, width), np.uint32) if want_hit else None
        t = np.empty((height, width), np.float32) if want_t else None
        cnt = np.zeros(2, np.uint64)
        hits = C.c_uint64()
        self._ck(self._lib.hvrt_group_render(self._g, C.byref(frame), width, height, spp, seed, _ptr(rgb), _ptr(hit), _ptr(t),
                                             _ptr(cnt), C.cast(C.byref(hits), C.c_void_p)))
        if want_hit or want_t:
            return rgb, hit, t, cnt, int(hits.value)
        return rgb, cnt, int(hits.value)

    def render_device(self, frame: CameraFrame, width: int, height: int, spp: int = 1, seed: int = 0):
        """Frame gathered and reassembled in member 0's memory; returns (device address, counters, hits)."""
        d = C.c_void_p()
        cnt = np.zeros(2, np.uint64)
        hits = C.c_uint64()
        self._ck(self._lib.hvrt_group_render_device(self._g, C.byref(frame), width, height, spp, seed, C.byref(d), _ptr(cnt),
                                                    C.cast(C.byref(hits), C.c_void_p)))
        return d.value, cnt, int(hits.value)

    def close(self):
        if getattr(self, "_g", None):
            self.root._h = None
            self._lib.hvrt_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class BuildResult:
    info: GridInfo
    timings: Timings


class Context:
    """One GPU, one scene (mirrors the life cycle of the reference's tracer object)."""

    def __init__(self, device_id: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.hvrt_create(C.byref(h), device_id)
        if rc != 0:
            raise HvrtError(rc, self._lib.hvrt_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):
            self._lib.hvrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc: int):
        if rc != 0:
            raise HvrtError(rc, self._lib.hvrt_last_error(self._h).decode())

    # -- mesh
    def set_mesh(self, tri_aos: np.ndarray):
        t = np.ascontiguousarray(tri_aos, dtype=np.float32).reshape(-1, 12)
        self._ck(self._lib.hvrt_set_mesh(self._h, _ptr(t), t.shape[0]))

    def set_mesh_device(self, d_ptr: int, n_tris: int):
        self._ck(self._lib.hvrt_set_mesh_device(self._h, C.c_void_p(d_ptr), n_tris))

    def compute_bounds(self, points: np.ndarray):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        self._ck(self._lib.hvrt_compute_bounds(self._h, _ptr(p), p.shape[0], mn, mx))
        return np.array(mn, np.float32), np.array(mx, np.float32)

    def mesh_bounds(self):
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        self._ck(self._lib.hvrt_mesh_bounds(self._h, mn, mx))
        return np.array(mn, np.float32), np.array(mx, np.float32)

    # -- grid
    def build_grid(self, nx, ny, nz, min_bound, max_bound, voxel_size3) -> GridInfo:
        vs = np.broadcast_to(np.asarray(voxel_size3, np.float32), (3,))
        self._ck(self._lib.hvrt_build_grid(self._h, nx, ny, nz, _f3(min_bound), _f3(max_bound), _f3(vs)))
        return self.grid_info()

    def set_grid_csr(self, nx, ny, nz, min_bound, max_bound, voxel_size3, offsets, indices):
        vs = np.broadcast_to(np.asarray(voxel_size3, np.float32), (3,))
        o = np.ascontiguousarray(offsets, dtype=np.uint32)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        if o.shape[0] != nx * ny * nz + 1:
            raise ValueError("offsets must have nx*ny*nz+1 entries")
        if i.shape[0] != int(o[-1]):
            raise ValueError("indices must have offsets[-1] entries")
        self._ck(self._lib.hvrt_set_grid_csr(self._h, nx, ny, nz, _f3(min_bound), _f3(max_bound), _f3(vs),
                                             _ptr(o), _ptr(i) if i.size else None))

    def set_grid_csr_device(self, nx, ny, nz, min_bound, max_bound, voxel_size3, d_offsets: int, d_indices: int,
                            n_refs: int):
        vs = np.broadcast_to(np.asarray(voxel_size3, np.float32), (3,))
        self._ck(self._lib.hvrt_set_grid_csr_device(self._h, nx, ny, nz, _f3(min_bound), _f3(max_bound), _f3(vs),
                                                    C.c_void_p(d_offsets), C.c_void_p(d_indices), n_refs))

    def grid_info(self) -> GridInfo:
        gi = GridInfo()
        self._ck(self._lib.hvrt_get_grid_info(self._h, C.byref(gi)))
        return gi

    def get_csr(self):
        gi = self.grid_info()
        off = np.empty(gi.n_voxels + 1, np.uint32)
        idx = np.empty(gi.n_refs, np.uint32)
        self._ck(self._lib.hvrt_get_csr(self._h, _ptr(off), _ptr(idx) if idx.size else None))
        return off, idx

    def distance_field(self) -> np.ndarray:
        """(nz, ny, nx) uint8: min(255, city-block distance to the nearest occupied voxel)."""
        gi = self.grid_info()
        out = np.empty(gi.n_voxels, np.uint8)
        self._ck(self._lib.hvrt_get_distance_field(self._h, _ptr(out)))
        return out.reshape(gi.nz, gi.ny, gi.nx)

    def device_arrays(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self._lib.hvrt_get_device_arrays(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def export_scene_device(self, d_tri_aos: int = 0, d_offsets: int = 0, d_indices: int = 0):
        vp = lambda p: C.c_void_p(p) if p else None
        self._ck(self._lib.hvrt_export_scene_device(self._h, vp(d_tri_aos), vp(d_offsets), vp(d_indices)))

    def timings(self) -> Timings:
        t = Timings()
        self._ck(self._lib.hvrt_get_timings(self._h, C.byref(t)))
        return t

    def render_history(self, n: int = 32) -> list:
        """Waits for the stream; durations (ms) of the last (up to n, at most 32) asynchronous traversals, oldest first."""
        buf = (C.c_float * max(1, n))()
        got = self._lib.hvrt_render_history(self._h, buf, max(1, n))
        if got < 0:
            self._ck(-1)
        return [float(buf[i]) for i in range(got)]

    def tile_cover(self, frame: int = 0) -> np.ndarray:
        """The tile cover of the last launch without counters: a bool array (rows of tiles, tiles per row), row 0 = the
        frame's row 0; True where the tile was walked."""
        dims = (C.c_uint32 * 4)()
        self._ck(self._lib.hvrt_get_tile_cover(self._h, frame, None, 0, dims))
        tx, ty, wx = int(dims[0]), int(dims[1]), int(dims[2])
        self.last_cover_general_case = int(dims[3])      # triangles the pass's second kernel handled
        words = np.zeros(ty * wx, dtype=np.uint32)
        self._ck(self._lib.hvrt_get_tile_cover(self._h, frame, words.ctypes.data_as(C.c_void_p), words.size, dims))
        bits = np.unpackbits(words.view(np.uint8).reshape(ty, wx * 4), axis=1, bitorder="little")
        return bits[:, :tx].astype(bool)

    def tile_costs(self, n_tiles: int) -> np.ndarray:
        """Durations (16 shader clocks; top bit: rendered by the 16-wave kernel) of the last launch's tiles (a diagnostic)."""
        out = np.zeros(n_tiles, dtype=np.uint32)
        got = self._lib.hvrt_get_tile_costs(self._h, out.ctypes.data_as(C.c_void_p), out.size)
        if got < 0:
            self._ck(got)
        return out[:got]

    def set_overlap_test(self, mode: int):
        """0 = every voxel of a triangle's bounding-box range lists it (the reference's rule, default); 1 = those it overlaps
        (separating-axis test).  For later build_grid calls."""
        self._ck(self._lib.hvrt_set_overlap_test(self._h, mode))

    def set_mode(self, mode: int):
        """0 = the reference's traversal (default), 1 = nearest-hit mode."""
        self._ck(self._lib.hvrt_set_mode(self._h, mode))

    def get_mode(self) -> int:
        return self._lib.hvrt_get_mode(self._h)

    def counter_width(self) -> int:
        """16 or 32: the per-voxel counters the next build_grid counts into first (hvrt_get_counter_width)."""
        return self._lib.hvrt_get_counter_width(self._h)

    def clear_scene(self):
        """Forget mesh and grid, settings back to their defaults; buffers and warmed-up queues are kept."""
        self._ck(self._lib.hvrt_clear_scene(self._h))

    def download(self, d_src: int, out: np.ndarray):
        self._ck(self._lib.hvrt_download(self._h, _ptr(out), C.c_void_p(d_src), out.nbytes))
        return out

    def download_blocks(self, d_src: int, n_bytes: int, out: np.ndarray, block_bytes: int, dst_stride: int, dst_offset: int = 0):
        """Compact device run -> blocks `dst_stride` bytes apart in `out` (hvrt_download_blocks)."""
        self._ck(self._lib.hvrt_download_blocks(self._h, C.c_void_p(out.ctypes.data + dst_offset), C.c_void_p(d_src), n_bytes,
                                                block_bytes, dst_stride))
        return out

    def set_deferred_accel(self, on: bool):
        """Keep libhvrt's derived acceleration data out of build_grid / set_grid_csr (built at first use instead)."""
        self._ck(self._lib.hvrt_set_deferred_accel(self._h, 1 if on else 0))

    def finish_accel(self):
        self._ck(self._lib.hvrt_finish_accel(self._h))

    def synchronize(self):
        self._ck(self._lib.hvrt_synchronize(self._h))

    def set_stream(self, hip_stream: int | None):
        self._ck(self._lib.hvrt_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    @property
    def stream(self) -> int:
        return self._lib.hvrt_stream(self._h) or 0

    # -- rendering
    def render(self, frame: CameraFrame, width: int, height: int, spp: int = 1, seed: int = 0,
               want_hit=True, want_t=True, want_counters=True):
        rgb = np.empty((height, width, 3), np.float32)
        hit = np.empty((height, width), np.uint32) if want_hit else None
        t = np.empty((height, width), np.float32) if want_t else None
        cnt = np.zeros(2, np.uint64) if want_counters else None
        self._ck(self._lib.hvrt_render(self._h, C.byref(frame), width, height, spp, seed,
                                       _ptr(rgb), _ptr(hit), _ptr(t), _ptr(cnt)))
        return rgb, hit, t, cnt

    def last_hit_count(self) -> int:
        """Pixels of the last render() / render_views_rgb8() whose ray hit a triangle (counted on the device)."""
        n = C.c_uint64()
        self._ck(self._lib.hvrt_last_hit_count(self._h, C.byref(n)))
        return int(n.value)

    def render_part_device(self, frame: CameraFrame, width: int, height: int, spp: int, seed: int,
                           tile_rows: int, part: int, n_parts: int,
                           d_rgb: int = 0, d_hit: int = 0, d_t: int = 0, d_counters: int = 0):
        vp = lambda p: C.c_void_p(p) if p else None
        self._ck(self._lib.hvrt_render_part_device(self._h, C.byref(frame), width, height, spp, seed, tile_rows,
                                                   part, n_parts, vp(d_rgb), vp(d_hit), vp(d_t), vp(d_counters)))

    def render_part_device_noalloc(self, frame: CameraFrame, width: int, height: int, spp: int, seed: int,
                                   tile_rows: int, part: int, n_parts: int,
                                   d_rgb: int = 0, d_hit: int = 0, d_t: int = 0, d_counters: int = 0):
        """render_part_device through the entry point that allocates nothing (fixed scratch inside libhvrt)."""
        vp = lambda p: C.c_void_p(p) if p else None
        self._ck(self._lib.hvrt_render_part_device_noalloc(self._h, C.byref(frame), width, height, spp, seed, tile_rows,
                                                           part, n_parts, vp(d_rgb), vp(d_hit), vp(d_t), vp(d_counters)))

    def render_batch_part_device(self, frames, width: int, height: int, spp: int, seed: int,
                                 tile_rows: int, part: int, n_parts: int, slab_rows: int,
                                 d_rgb: int = 0, d_hit: int = 0, d_t: int = 0, d_counters: int = 0):
        """One launch for `len(frames)` camera frames; frame f's rows go to slab f of the output buffers."""
        vp = lambda p: C.c_void_p(p) if p else None
        arr = (CameraFrame * len(frames))(*frames)
        self._ck(self._lib.hvrt_render_batch_part_device(self._h, arr, len(frames), width, height, spp, seed, tile_rows,
                                                         part, n_parts, slab_rows, vp(d_rgb), vp(d_hit), vp(d_t),
                                                         vp(d_counters)))

    def quantise_rgb8_device(self, d_rgb: int, n_values: int, d_rgb8: int):
        self._ck(self._lib.hvrt_quantise_rgb8_device(self._h, C.c_void_p(d_rgb), n_values, C.c_void_p(d_rgb8)))

    def render_views_rgb8(self, frames, width: int, height: int, want_hit=True):
        """len(frames) views in one launch -> (n, H, W, 3) uint8 (+ (n, H, W) uint32 hit ids, counters)."""
        n = len(frames)
        arr = (CameraFrame * n)(*frames)
        rgb8 = np.empty((n, height, width, 3), np.uint8)
        hit = np.empty((n, height, width), np.uint32) if want_hit else None
        cnt = np.zeros(2, np.uint64)
        self._ck(self._lib.hvrt_render_views_rgb8(self._h, arr, n, width, height, _ptr(rgb8), _ptr(hit), _ptr(cnt)))
        return rgb8, hit, cnt

    def assemble_parts_device(self, d_parts: int, width: int, height: int, tile_rows: int, n_parts: int,
                              slab_rows: int, bytes_per_pixel: int, d_frame: int):
        self._ck(self._lib.hvrt_assemble_parts_device(self._h, C.c_void_p(d_parts), width, height, tile_rows,
                                                      n_parts, slab_rows, bytes_per_pixel, C.c_void_p(d_frame)))

    def trace_device(self, d_origins: int, d_dirs: int, n: int, d_hits: int, d_counters: int = 0):
        self._ck(self._lib.hvrt_trace_device(self._h, C.c_void_p(d_origins), C.c_void_p(d_dirs), n, C.c_void_p(d_hits),
                                             C.c_void_p(d_counters) if d_counters else None))

    def trace(self, origins: np.ndarray, dirs: np.ndarray, want_counters=True):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("origins and dirs must have the same shape")
        out = np.zeros(o.shape[0], HIT_DTYPE)
        cnt = np.zeros(2, np.uint64) if want_counters else None
        self._ck(self._lib.hvrt_trace(self._h, _ptr(o), _ptr(d), o.shape[0], _ptr(out), _ptr(cnt)))
        return out, cnt
