"""The C++ host mirror (raytracer_amd/host): STL ingest numbering vs the real STLReader.h (golden), and --
on the GPU -- TraceImage(stl, out, HIP_VOXEL) end to end vs the oracle, multi-view, .voxgrid round trip,
image encoders, and the reference's error behaviour (log + swallow -> CLI exit code)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle
from raytracer_amd import capi, meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "raytracer_amd", "lib", "trace_image")
MISS = 0xFFFFFFFF


def run(*args, check=True):
    p = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=300)
    if check and p.returncode != 0:
        raise AssertionError(f"trace_image {' '.join(args)} -> {p.returncode}\n{p.stderr[-2000:]}")
    return p


def stl_info(path, tmp):
    out = os.path.join(tmp, "info.bin")
    run("stl-info", path, out)
    b = open(out, "rb").read()
    nv, nt = struct.unpack("<QQ", b[:16])
    o = 16
    co = np.frombuffer(b[o:o + 12 * nv], np.float32).reshape(-1, 3); o += 12 * nv
    ti = np.frombuffer(b[o:o + 12 * nt], np.uint32).reshape(-1, 3); o += 12 * nt
    nr = np.frombuffer(b[o:o + 12 * nt], np.float32).reshape(-1, 3)
    return co, ti, nr


@pytest.mark.parametrize("kind", ["binary", "ascii"])
def test_stl_loader_reproduces_reference_numbering(kind, tmp_path):
    """Golden = output of the reference's own stl_reader::StlMesh on the same file."""
    co, ti, nr = stl_info(os.path.join(G, f"fixture_{kind}.stl"), str(tmp_path))
    z = np.load(os.path.join(G, f"stl_{kind}_expected.npz"))
    assert np.array_equal(co.view(np.uint32), z["coords_bits"])
    assert np.array_equal(ti, z["tris"])
    assert np.array_equal(nr.view(np.uint32), z["normals_bits"])


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libref.so not built")
def test_stl_loader_live_vs_reference_reader(tmp_path):
    rng = np.random.default_rng(3)
    soup = np.concatenate([meshgen.cube_sphere(7)[:, :9].reshape(-1, 3, 3),
                           rng.integers(-2, 3, (300, 3, 3)).astype(np.float32)])   # many duplicates / degenerates
    for kind in ("binary", "ascii"):
        path = str(tmp_path / f"m_{kind}.stl")
        (meshgen.write_binary_stl if kind == "binary" else meshgen.write_ascii_stl)(path, soup)
        co, ti, nr = stl_info(path, str(tmp_path))
        rco, rti, rnr = oracle.ref_stl_load(path)
        assert np.array_equal(co, rco) and np.array_equal(ti, rti)      # values (+-0 may differ in sign bit only)
        assert np.array_equal(nr.view(np.uint32), rnr.view(np.uint32))


def test_stl_loader_errors(tmp_path):
    assert run("stl-info", str(tmp_path / "missing.stl"), str(tmp_path / "o.bin"), check=False).returncode == 1
    bad = tmp_path / "trunc.stl"
    bad.write_bytes(b"\0" * 80 + struct.pack("<I", 5) + b"\0" * 60)
    assert run("stl-info", str(bad), str(tmp_path / "o.bin"), check=False).returncode == 1


# ---------------------------------------------------------------------------------------------- GPU

def oracle_frame_for_stl(path, g, W, H, view="front"):
    """What the reference pipeline computes for this file: reader numbering -> bounds over the deduped
    vertices -> 50^3 grid -> default camera (aspect 800/600) -> frame."""
    soup_co, ti, nr = None, None, None
    if oracle.have_ref():
        soup_co, ti, nr = oracle.ref_stl_load(path)
    else:                                     # GPU box without libref: the golden numbering of the fixture
        z = np.load(os.path.join(G, "stl_binary_expected.npz"))
        soup_co, ti, nr = z["coords_bits"].view(np.float32), z["tris"], z["normals_bits"].view(np.float32)
    tris = np.concatenate([soup_co[ti].reshape(-1, 9), nr], 1).astype(np.float32)
    sc, mn, mx, vs = oracle.build_scene(tris, g, g, g, points=soup_co)
    pos, la, up, fov, asp = capi.default_camera(mn, mx, view)
    fr = oracle.camera_frame(pos, la, up, fov, asp)
    return sc.render(fr, W, H, threads=8), sc, mn, mx


def read_dump(path, W, H):
    b = open(path, "rb").read()
    rgb = np.frombuffer(b[:W * H * 12], np.float32).reshape(H, W, 3)
    hit = np.frombuffer(b[W * H * 12:], np.uint32).reshape(H, W)
    return rgb, hit


def quantise(rgb):
    return np.clip(rgb * np.float32(255.0), np.float32(0), np.float32(255)).astype(np.uint8)   # ImageSaver.h:30-32


def decode_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    o, idat, w, h = 8, b"", 0, 0
    while o < len(b):
        n, typ = struct.unpack(">I4s", b[o:o + 8])
        data = b[o + 8:o + 8 + n]
        assert zlib.crc32(typ + data) == struct.unpack(">I", b[o + 8 + n:o + 12 + n])[0]
        if typ == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
            assert data[8:] == bytes([8, 2, 0, 0, 0])
        elif typ == b"IDAT":
            idat += data
        o += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.gpu
def test_trace_image_drop_in_end_to_end(tmp_path):
    """Config 1's plumbing: TraceImage(stl, png, HIP_VOXEL) at the reference's defaults (50^3, 800x600, front
    camera) on a committed STL fixture (the reference's bundled stl2.bin is absent from its checkout)."""
    stl = os.path.join(G, "fixture_binary.stl")
    dump = str(tmp_path / "raw.bin")
    run("render", stl, "out.png", "--outdir", str(tmp_path) + "/", "--dump-raw", dump)
    rgb, hit = read_dump(dump, 800, 600)
    (orgb, ohit, ot, oc), *_ = oracle_frame_for_stl(stl, 50, 800, 600)
    assert np.array_equal(hit, ohit) and (hit != MISS).any()
    assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))
    assert np.array_equal(decode_png(str(tmp_path / "out.png")), quantise(orgb))
    # PPM path
    run("render", stl, "out.ppm", "--outdir", str(tmp_path) + "/", "--quiet")
    b = open(tmp_path / "out.ppm", "rb").read()
    head = b"P6\n800 600\n255\n"
    assert b.startswith(head) and np.array_equal(np.frombuffer(b[len(head):], np.uint8).reshape(600, 800, 3), quantise(orgb))


@pytest.mark.gpu
def test_trace_image_multiview_and_voxgrid_round_trip(tmp_path):
    stl = str(tmp_path / "sphere.stl")
    meshgen.write_binary_stl(stl, meshgen.cube_sphere(14)[:, :9].reshape(-1, 3, 3))
    out = str(tmp_path) + "/"
    run("render", stl, "mv.ppm", "--multiview", "--outdir", out, "--size", "200x150", "--grid", "24", "--quiet")
    frames = {}
    for view in ("front", "side", "top"):
        b = open(tmp_path / f"mv_{view}.ppm", "rb").read()
        head = b"P6\n200 150\n255\n"
        (orgb, ohit, _, _), *_ = oracle_frame_for_stl(stl, 24, 200, 150, view)
        assert np.array_equal(np.frombuffer(b[len(head):], np.uint8).reshape(150, 200, 3), quantise(orgb)), view
        frames[view] = orgb
    # .voxgrid: written with the reference's byte layout, read back (no index array in the file -> the CSR
    # is rebuilt on the GPU), rendered: same frame as from the STL
    vg = str(tmp_path / "sphere.voxgrid")
    run("save-voxgrid", stl, vg, "--grid", "24")
    size = os.path.getsize(vg)
    n_tris = 12 * 14 * 14
    assert size == 12 + 48 + 24 ** 3 * 28 + 4 + n_tris * 48
    dump = str(tmp_path / "vg.bin")
    run("render", vg, "vg.ppm", "--outdir", out, "--size", "200x150", "--dump-raw", dump, "--quiet")
    rgb, hit = read_dump(dump, 200, 150)
    assert np.array_equal(rgb.view(np.uint32), frames["front"].view(np.uint32))


@pytest.mark.gpu
def test_trace_image_nearest_hit_knob(tmp_path):
    """TraceImages::nearestHit / trace_image --nearest: the frame is the oracle's mode-1 frame (an extension; the
    default stays the reference's traversal, checked above)."""
    stl = os.path.join(G, "fixture_binary.stl")
    dump = str(tmp_path / "raw.bin")
    run("render", stl, "n.ppm", "--outdir", str(tmp_path) + "/", "--size", "320x240", "--grid", "20", "--nearest",
        "--dump-raw", dump, "--quiet")
    rgb, hit = read_dump(dump, 320, 240)
    (orgb0, ohit0, _, _), sc, mn, mx = oracle_frame_for_stl(stl, 20, 320, 240)
    sc.set_mode(1, mx)
    pos, la, up, fov, asp = capi.default_camera(mn, mx, "front")
    orgb, ohit, _, _ = sc.render(oracle.camera_frame(pos, la, up, fov, asp), 320, 240, threads=8)
    assert np.array_equal(hit, ohit) and np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))
    assert not np.array_equal(ohit, ohit0)          # the two traversals do differ on this scene


@pytest.mark.gpu
def test_trace_image_error_behaviour(tmp_path):
    stl = os.path.join(G, "fixture_binary.stl")
    # unsupported extension: the reference throws "Unsupported file format", TraceImage logs and returns
    p = run("render", os.path.join(G, "mt_vectors.npz"), "x.ppm", "--outdir", str(tmp_path) + "/", check=False)
    assert p.returncode == 1 and "Unsupported file format: .npz" in p.stderr
    # the CPU enumerators are not served by a hidden fallback
    p = run("render", stl, "x.ppm", "--method", "octree", "--outdir", str(tmp_path) + "/", check=False)
    assert p.returncode == 1 and "reference's CPU tracer" in p.stderr
    assert not os.path.exists(tmp_path / "x.ppm")


@pytest.mark.gpu
def test_jpg_output_names_are_written_as_ppm_under_a_ppm_name(tmp_path):
    """The reference's driver passes .jpg names (RayTracer.cpp:24-27); this library has no JPEG encoder, so the
    file is written as PPM with the extension replaced -- never PPM bytes under a JPEG name."""
    stl = os.path.join(G, "fixture_binary.stl")
    out = str(tmp_path) + "/"
    p = run("render", stl, "view.jpg", "--outdir", out, "--size", "160x120", "--grid", "16")
    assert not os.path.exists(tmp_path / "view.jpg")
    assert open(tmp_path / "view.ppm", "rb").read(2) == b"P6"
    assert "view.ppm" in (p.stdout + p.stderr)
    run("render", stl, "mv.jpg", "--multiview", "--outdir", out, "--size", "160x120", "--grid", "16", "--quiet")
    for view in ("front", "side", "top"):
        assert open(tmp_path / f"mv_{view}.ppm", "rb").read(2) == b"P6"
        assert not os.path.exists(tmp_path / f"mv_{view}.jpg")


@pytest.mark.gpu
def test_trace_image_over_a_device_group(tmp_path):
    """TraceImage(stl, out, HIP_VOXEL) with the scene spanning several devices (HVRT_DEVICES; here three contexts on
    the one GPU of this box, the copy transport): HipScene replicates the scene, HipVoxelRayTracer::render goes through
    hvrt_group_render, and the file is byte for byte the one a single device writes."""
    stl = os.path.join(G, "fixture_binary.stl")
    out = str(tmp_path) + "/"
    run("render", stl, "one.ppm", "--outdir", out, "--size", "640x360", "--grid", "32", "--quiet")
    env = dict(os.environ, HVRT_DEVICES="0,0,0", HVRT_GROUP_MIN_PIXELS="1")
    p = subprocess.run([EXE, "render", stl, "three.ppm", "--outdir", out, "--size", "640x360", "--grid", "32"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    log = p.stdout + p.stderr
    assert "spans 3 devices (copy transport" in log and "on 3 devices" in log and "scene replicated to 3 devices" in log
    assert open(tmp_path / "one.ppm", "rb").read() == open(tmp_path / "three.ppm", "rb").read()


@pytest.mark.gpu
def test_trace_image_group_keeps_mode_and_hit_ids(tmp_path):
    """--nearest with the hit-id image kept, over three members: byte for byte one device's frame and ids (round 2:
    the mode reached member 0 only, and keepHitIds fell back to one device)."""
    stl = os.path.join(G, "fixture_binary.stl")
    out = str(tmp_path) + "/"
    one, three = str(tmp_path / "one.bin"), str(tmp_path / "three.bin")
    run("render", stl, "one.ppm", "--outdir", out, "--size", "640x360", "--grid", "32", "--nearest", "--dump-raw", one, "--quiet")
    run("render", stl, "ref.ppm", "--outdir", out, "--size", "640x360", "--grid", "32", "--quiet")
    env = dict(os.environ, HVRT_DEVICES="0,0,0", HVRT_GROUP_MIN_PIXELS="1")
    p = subprocess.run([EXE, "render", stl, "three.ppm", "--outdir", out, "--size", "640x360", "--grid", "32", "--nearest",
                        "--dump-raw", three], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "on 3 devices" in p.stdout + p.stderr
    assert open(one, "rb").read() == open(three, "rb").read()
    assert open(tmp_path / "one.ppm", "rb").read() == open(tmp_path / "three.ppm", "rb").read()
    assert open(tmp_path / "one.ppm", "rb").read() != open(tmp_path / "ref.ppm", "rb").read()      # the modes differ here


@pytest.mark.gpu
def test_a_tracer_per_image_reuses_pooled_contexts(tmp_path):
    """The reference's flow makes a tracer (here: a scene, a context, with HVRT_DEVICES a whole group) per image.  The
    second and third image of a process take theirs from the pool: no second hvrt_create / hvrt_group_create, the scene
    is replicated again (it is a new scene), and every image is the same file."""
    import json
    stl = os.path.join(G, "fixture_binary.stl")
    out = str(tmp_path) + "/"
    for env_extra, tag in (({}, "one"), ({"HVRT_DEVICES": "0,0,0,0", "HVRT_GROUP_MIN_PIXELS": "1"}, "four")):
        ph = str(tmp_path / f"{tag}.jsonl")
        p = subprocess.run([EXE, "render", stl, f"{tag}.ppm", "--outdir", out, "--size", "640x360", "--grid", "32", "--repeat", "3",
                            "--phases", ph], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env_extra))
        assert p.returncode == 0, p.stderr[-2000:]
        log = p.stdout + p.stderr
        calls = [json.loads(l) for l in open(ph)]
        assert len(calls) == 3
        # the first call pays the runtime's start-up; the later ones take a warm context in well under that
        assert calls[1]["context_ms"] < 0.25 * calls[0]["context_ms"] and calls[2]["context_ms"] < 0.25 * calls[0]["context_ms"], calls
        if tag == "four":
            assert log.count("group created in") == 1 and log.count("group taken from the pool") == 2
            assert log.count("scene replicated to 4 devices") == 3
    assert open(tmp_path / "one.ppm", "rb").read() == open(tmp_path / "four.ppm", "rb").read()
    # HVRT_CONTEXT_CACHE=0: no pool -- every image creates and destroys its own context; the same file
    ph = str(tmp_path / "nocache.jsonl")
    p = subprocess.run([EXE, "render", stl, "nocache.ppm", "--outdir", out, "--size", "640x360", "--grid", "32", "--repeat", "3", "--phases", ph,
                        "--quiet"], capture_output=True, text=True, timeout=300, env=dict(os.environ, HVRT_CONTEXT_CACHE="0"))
    assert p.returncode == 0, p.stderr[-2000:]
    calls = [json.loads(l) for l in open(ph)]
    assert len(calls) == 3 and all(c["context_ms"] > 1.0 for c in calls), calls          # a context per image costs milliseconds each time
    assert open(tmp_path / "one.ppm", "rb").read() == open(tmp_path / "nocache.ppm", "rb").read()
