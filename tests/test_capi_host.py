"""CPU-side checks of the boundary: libhvrt loads, exports every symbol include/hvrt.h declares, its
host-only entry points agree with the oracle, and it fails loudly (no fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hvrt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hvrt_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libhvrt.so does not export {s}"
    assert sorted(capi.SYMBOLS) == syms, "capi.SYMBOLS out of date with include/hvrt.h"


def test_struct_layouts_match_header():
    assert C.sizeof(capi.CameraFrame) == 48
    assert capi.HIT_DTYPE.itemsize == 52        # RayHit is 52 bytes in the reference
    assert C.sizeof(capi.GridInfo) == 3 * 4 + 9 * 4 + 3 * 8 + 2 * 4
    assert C.sizeof(capi.Timings) == 8 * 4


def test_create_without_gpu_fails_loudly_or_works():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.HvrtError) as e:
        capi.Context(0)
    assert e.value.code == -2 and "no HIP device" in str(e.value)


@pytest.mark.parametrize("dims", [(50, 50, 50), (16, 32, 8), (1, 1, 1), (256, 3, 77)])
def test_voxel_size_matches_oracle(dims):
    rng = np.random.default_rng(sum(dims))
    for _ in range(50):
        mn = rng.uniform(-10, 10, 3).astype(np.float32)
        mx = (mn + rng.uniform(0, 20, 3)).astype(np.float32)
        a = capi.voxel_size(mn, mx, *dims)
        b = oracle.voxel_size(mn, mx, *dims)
        assert a.view(np.uint32) == b.view(np.uint32)
    z = np.zeros(3, np.float32)
    assert capi.voxel_size(z, z, *dims) == 0.0       # a one-point mesh: the reference divides 0 by n too


@pytest.mark.parametrize("view", ["front", "side", "top"])
def test_camera_frame_matches_oracle(view):
    rng = np.random.default_rng(3)
    for _ in range(20):
        mn = rng.uniform(-5, 5, 3).astype(np.float32)
        mx = (mn + rng.uniform(0.1, 9, 3)).astype(np.float32)
        pos, la, up, fov, asp = capi.default_camera(mn, mx, view, aspect=rng.choice([800 / 600, 16 / 9]))
        a = capi.make_camera_frame(pos, la, up, fov, asp)
        b = oracle.camera_frame(pos, la, up, fov, asp)
        assert bytes(a) == bytes(b)
    # degenerate: camera at the look-at point (zero forward vector stays zero, VectorMath.h:11-14)
    a = capi.make_camera_frame([1, 2, 3], [1, 2, 3], [0, 1, 0], 60.0, 1.5)
    b = oracle.camera_frame([1, 2, 3], [1, 2, 3], [0, 1, 0], 60.0, 1.5)
    assert bytes(a) == bytes(b)


def test_part_rows_partition_covers_every_row_once():
    for H in (1, 7, 16, 600, 2160, 4320):
        for tile in (1, 8, 16, 37):
            for n in (1, 2, 3, 4, 8):
                rows = [capi.part_rows(H, tile, p, n) for p in range(n)]
                assert sum(rows) == H
                n_tiles = -(-H // tile)
                for p in range(n):
                    exp = sum(min(tile, H - k * tile) for k in range(p, n_tiles, n))
                    assert rows[p] == exp
    assert capi.part_rows(10, 4, 3, 2) == 0 and capi.part_rows(10, 0, 0, 1) == 0


def test_default_camera_is_the_reference_rule():
    # TraceImages.cpp:125-146: position = centre - (0, 0, 2.5 * maxSize), fov 60, aspect 800/600
    pos, la, up, fov, asp = capi.default_camera([-1, -1, -1], [1, 1, 1])
    assert np.array_equal(pos, np.array([0, 0, -5], np.float32)) and np.array_equal(la, np.zeros(3, np.float32))
    assert fov == 60.0 and np.float32(asp) == np.float32(800.0) / np.float32(600.0)
    pos, la, up, fov, asp = capi.default_camera([-1, -1, -1], [1, 1, 1], "top")
    assert np.array_equal(pos, np.array([0, 5, 0], np.float32)) and np.array_equal(up, np.array([0, 0, 1], np.float32))


def test_product_does_not_touch_the_oracle():
    """The product tree must not import / link / load anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle|cpu_ref|liboracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
