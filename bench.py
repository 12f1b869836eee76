#!/usr/bin/env python3
"""bench.py -- primary-ray throughput at 4K on BASELINE.json config 2 (default), the other configs on request.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|small] [--scaling weak|strong] [--moving-camera]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Workloads (config.workload; synthetic, libm-free generators of raytracer_amd/meshgen.py):
  c2  cube-sphere, 1 002 252 triangles, 256^3, 3840x2160, 1 spp            -- the configuration the metric is quoted on
  c3  cube-sphere N=913 (10 002 828 triangles) + a 2.4 M-triangle patch over 1 % of the surface, 512^3, 3840x2160
  c4  the c2 scene at 7680x4320, 16 jittered samples per pixel (seed 1)
  c5  voxel-build stress: cube-sphere N=2041 (49 988 172 triangles) into 1024^3; build only (metric: build ms)
  small  config-1-shaped plumbing case (40 368 triangles, 50^3, 800x600)
Camera: the reference's front rule (TraceImages.cpp:125-146) with aspect = W/H, pixel-corner sampling.

A step = one pass of the hot path over one batch of rays:
  N = 1          the traversal kernel renders one frame (rgb + hit id + t written) from the HBM-resident scene.
  N > 1, weak    (default) a batch of N frames; every frame's 16-row tiles are interleaved over the N ranks, each
                 rank traces one frame's worth of rays per step, ONE RCCL all_to_all of 8-bit slabs hands frame f to
                 rank f (overlapped with the next batch's traversal), which reassembles it.
  N > 1, strong  one frame per step, its row tiles interleaved over the N ranks, RCCL all_gather of the 8-bit slabs,
                 reassembled on every rank (config 4's "row tiles partitioned across 8 MI355X").
value = all rays of the K steps / max-over-ranks wall time.
Output: ONE JSON line on rank 0 (metric / roofline / cpu_baseline as DESIGN.md section 6 defines them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (mesh, grid, W, H, spp, seed)
    "c2": (("cube_sphere", 289), 256, 3840, 2160, 1, 0),
    "c3": (("skewed_statue", 913, 1100), 512, 3840, 2160, 1, 0),
    "c4": (("cube_sphere", 289), 256, 7680, 4320, 16, 1),
    "c5": (("cube_sphere", 2041), 1024, 0, 0, 0, 0),
    "small": (("cube_sphere", 58), 50, 800, 600, 1, 0),          # config-1-shaped plumbing case (not a bench line)
}


def make_mesh(spec):
    from raytracer_amd import meshgen
    return getattr(meshgen, spec[0])(*spec[1:])


def cpu_baseline(tris, g, mn, vs, off, idx, cam, W, H, spp, seed, n_rays_target=8.3e6):
    """Times the oracle (our CPU port of the reference path, kind 'port') on a bounded sample of the SAME
    frame: evenly spaced rows, all host threads of this box's CPU share."""
    import oracle
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    sc = oracle.OracleScene(tris, g, g, g, mn, vs, off, idx)
    n_rows = min(H, max(threads, int(n_rays_target // (W * spp))))
    rows = np.linspace(0, H - 1, n_rows).astype(int)
    from concurrent.futures import ThreadPoolExecutor
    import ctypes as C
    f = oracle.Frame.from_buffer_copy(bytes(cam))
    cnts = [np.zeros(2, np.uint64) for _ in range(threads)]

    def work(k):
        for y in rows[k::threads]:
            oracle.lib().orc_render_rows(C.byref(sc.c), C.byref(f), W, H, int(y), int(y) + 1, spp, C.c_uint32(seed),
                                         None, None, None, C.c_void_p(cnts[k].ctypes.data))

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    rays = len(rows) * W * spp
    # one core, every 16th of those rows (SURVEY 8d asks for both)
    rows1 = rows[::16]
    t2 = time.perf_counter()
    for y in rows1:
        oracle.lib().orc_render_rows(C.byref(sc.c), C.byref(f), W, H, int(y), int(y) + 1, spp, C.c_uint32(seed),
                                     None, None, None, C.c_void_p(cnts[0].ctypes.data))
    dt1 = time.perf_counter() - t2
    # voxel build on one core (flat CSR port of SceneCreator.cpp:197-281)
    t1 = time.perf_counter()
    oracle.voxelize(tris, g, g, g, mn, vs)
    bt = time.perf_counter() - t1
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{len(rows)} evenly spaced rows of the same {W}x{H} frame at {spp} spp ({rays} rays, {dt:.2f} s); "
                      f"oracle/cpu_ref.c, one trace per sample",
            "value_1core": len(rows1) * W * spp / dt1 / 1e6,   # Mrays/s, one thread, every 16th row of the sample
            "build_ms_1core": bt * 1e3 / 2.0,   # voxelize() runs the count pass twice (sizing + fill)
            "reference_itself": "the reference's own OctreeLikeRayTracer does not compile in this image (spdlog); measured by "
                                "the survey session on one core of the authoring container: 0.07-0.09 Mrays/s on a 1.0 M-triangle "
                                "sphere at 256^3, build 1.9-2.1 s (SURVEY.md section 6)"}


def load_profile(workload):
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tp)).get(workload, {})
    except Exception:
        return {}


def render_roofline(prof, k_ms, rays, bytes_per_ray):
    """What binds k_render, as a fraction <= 1 of that resource, plus the HBM yardstick north_star asks for.

    Live in this run: kernel_ms (HIP events on libhvrt's stream) and everything computed from it alone (requested_GBps;
    hbm_frac and reuse_factor use the profile's byte count with this run's duration).  Not live: the busy fractions --
    they come from the counter passes installed under profiles/ (traffic.json names the file per workload) and are
    labelled so in `from_profiles`.  `frac` is the FLOOR of VALU issue: every vector instruction priced at the guide's
    2 clocks per wave64 instruction (MI355X_MICROARCH.md; nothing issues faster), so it cannot exceed 1 and cannot
    overstate; the same count at the 2.2 clocks this repository's micro-benchmark measures for plain VOP1 / VOP2
    instructions is given beside it.  (Round 2's estimate priced with a static instruction mix exceeded 1 on two
    workloads and is no longer part of the line.)"""
    requested = bytes_per_ray * rays                       # bytes the REFERENCE algorithm asks for, per launch
    out = {"kernel": "k_render", "kernel_ms": k_ms,
           "kernel_ms_is": "HIP events around the whole launch on libhvrt's stream: k_render and, beside it, k_render_split and "
                           "k_fill_background, behind k_compact_queue (and the cover pass when the cover does not stand)",
           "algorithmic_bytes_per_ray": bytes_per_ray,
           "requested_GBps": requested / (k_ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0,
           "live": ["kernel_ms", "requested_GBps", "algorithmic_bytes_per_ray", "hbm_frac (this run's kernel_ms, the profile's bytes)", "reuse_factor"]}
    traffic = prof.get("k_render_hbm_bytes_per_launch")
    out["traffic"] = traffic
    if traffic:
        out["hbm_frac"] = traffic / (k_ms * 1e-3) / 8e12    # counter-based: HBM is nowhere near binding
        out["reuse_factor"] = requested / traffic           # requested bytes served from cache or never asked for
    cands = {k: prof[k] for k in ("valu_issue_frac_guide_clk", "vmem_return_frac", "salu_issue_frac", "lds_frac") if k in prof}
    if cands:
        bound = max(cands, key=cands.get)
        name = {"valu_issue_frac_guide_clk": "valu_issue"}.get(bound, bound.replace("_frac", ""))
        out.update({"bound": name, "achieved": cands[bound], "peak": 1.0,
                    "unit": "busy clocks of the binding unit per kernel clock", "frac": cands[bound],
                    "frac_is": "a floor (2 clocks per VALU wave-instruction, the guide's figure); static: from the counter passes under "
                               "profiles/, not measured in this run",
                    "bound_note": "with the tile cover a frame leaves k_render about as many tiles to walk as the GPU has wave slots: the "
                                  "launch lasts as long as its longest tiles and no unit is busy for most of it; where every tile is walked "
                                  "(HVRT_TILE_COVER=0, roofline_no_tile_cover) the same kernel is bound by VALU issue"})
        out["candidates"] = cands
        fp = {k: prof[k] for k in ("valu_issue_frac", "valu_insts_per_launch", "valu_clocks_per_inst_floor", "vector_loads_per_launch",
                                   "vector_loads_per_tile", "kernel_cycles", "l2_hit_rate", "source", "busy_source") if k in prof}
        if "valu_issue_frac" in fp:
            fp["valu_issue_frac_at_2p2_clk"] = fp.pop("valu_issue_frac")
        fp["source_kind"] = "profiles (static), not this run"
        out["from_profiles"] = fp
    else:
        # no profile installed for this workload: the HBM yardstick alone (requested bytes against the 8 TB/s peak)
        out.update({"bound": "hbm", "achieved": out["requested_GBps"], "peak": 8000.0, "unit": "GB/s",
                    "frac": None, "note": "no counter profile for this workload under profiles/; requested (algorithmic) "
                                          "bytes are mostly served from cache, so their rate is not a fraction of the HBM peak"})
    return out


def rehearse_on_one_gpu(dist, D, torch):
    """HVRT_BENCH_REHEARSE=1: every rank uses GPU 0 and the collectives travel over gloo, staged through host memory
    (RCCL cannot put two ranks on one device).  For checking, on a one-GPU machine, that the N > 1 code of this file
    runs end to end (tests/test_gpu_distributed.py); the numbers of such a run mean nothing."""
    real_broadcast, real_a2a, real_gather = dist.broadcast, dist.all_to_all_single, dist.all_gather_into_tensor

    class Done:
        def wait(self):
            return True

    def broadcast(t, src=0, group=None, async_op=False):
        if not t.is_cuda:
            return real_broadcast(t, src=src, group=group)
        torch.cuda.synchronize()
        h = t.cpu()
        real_broadcast(h, src=src, group=group)
        t.copy_(h)
        torch.cuda.synchronize()

    def all_to_all_single(out, inp, group=None, async_op=False):
        torch.cuda.synchronize()
        hi, ho = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        real_a2a(ho, hi, group=group)
        out.copy_(ho)
        torch.cuda.synchronize()
        return Done()

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        torch.cuda.synchronize()
        hi = inp.cpu().contiguous()
        parts = [torch.empty_like(hi) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, hi, group=group)
        out.copy_(torch.stack(parts).reshape(out.shape))
        torch.cuda.synchronize()

    D.dist.broadcast, D.dist.all_to_all_single, D.dist.all_gather_into_tensor = broadcast, all_to_all_single, all_gather_into_tensor


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--moving-camera", action="store_true",
                    help="N = 1: after the timed region, an orbit of 1 degree per frame (adds `moving_camera` to the line; off by "
                         "default so that the default command's kernel trace holds the timed region's launches only)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c4": 5, "c5": 5}.get(args.workload, 20)

    if os.environ.get("HVRT_BENCH_WATCHDOG"):      # seconds: dump every thread's stack and exit instead of hanging (diagnostics)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["HVRT_BENCH_WATCHDOG"]), exit=True)
    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import capi
    from raytracer_amd import distributed as D
    from raytracer_amd.distributed import DistributedRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    rehearse = os.environ.get("HVRT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rccl = {"world_size": 1, "backend": None}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            rehearse_on_one_gpu(dist, D, torch)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        rccl = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                "exchange": "rgb8 slabs (3 B per pixel), " + ("all_to_all, frame f to rank f" if args.scaling == "weak" else "all_gather, every rank gets the frame")}

    mesh_spec, g, W, H, spp, seed = WORKLOADS[args.workload]
    ctx = ra.Context(local_rank)
    dr = DistributedRenderer(ctx, rank, world, device, tile_rows=16)
    tris = make_mesh(mesh_spec) if rank == 0 else None
    mn, mx, vs3 = dr.replicate_scene(tris, (g, g, g))
    info = ctx.grid_info()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        ctx.synchronize()

    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rccl": rccl}
    if rehearse:
        base["rehearsal"] = "HVRT_BENCH_REHEARSE=1: all ranks on GPU 0, collectives over gloo through host memory -- the numbers mean nothing"
    scene = {"refs": int(info.n_refs), "occupied_voxels": int(info.n_occupied), "max_list": int(info.max_list)}
    nvox = g ** 3
    b_build = 72.0 * info.n_triangles + 12.0 * nvox + 20.0 * info.n_refs

    # ---- voxel-grid build (rank 0's GPU; mesh resident in HBM) --------------------------------------------------------
    def timed_builds(n):
        bt, last = [], None
        for _ in range(n):
            ctx.build_grid(g, g, g, mn, mx, vs3)
            last = ctx.timings()
            bt.append(last.build_total_ms)
        return bt, last

    def split_builds(n):
        """The CSR alone (what B_build prices and the reference's grid is) and libhvrt's derived acceleration data alone:
        builds with hvrt_set_deferred_accel, so that the two do not overlap on the GPU as they do in a normal build."""
        ctx.set_deferred_accel(True)
        csr, accel, phases = [], [], None
        try:
            for _ in range(n):
                ctx.build_grid(g, g, g, mn, mx, vs3)
                phases = ctx.timings()
                csr.append(phases.build_total_ms)
                ctx.finish_accel()
                accel.append(ctx.timings().build_expand_ms)
        finally:
            ctx.set_deferred_accel(False)
        return float(np.median(csr)), float(np.median(accel)), phases

    def build_report(bt, tmb, n_split=4):
        ms = float(np.median(bt))
        csr_ms, accel_ms, ph = split_builds(n_split)
        prof = load_profile(args.workload)
        r = {"bound": "hbm", "achieved": b_build / (csr_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
             "frac": b_build / (csr_ms * 1e-3) / 8e12, "algorithmic_bytes": b_build,
             "csr_ms": csr_ms, "accel_ms": accel_ms, "build_total_ms": ms, "frac_of_total": b_build / (ms * 1e-3) / 8e12,
             "csr_phases_ms": {"count": ph.build_count_ms, "scan": ph.build_scan_ms, "fill": ph.build_fill_ms, "order": ph.build_sort_ms},
             "traffic": prof.get("build_hbm_bytes"), "traffic_source": prof.get("build_source"),
             "note": "B_build = 72 T + 12 Nvox + 20 Nrefs prices the CSR (count, scan, fill, order) -- the reference's whole grid; "
                     "csr_ms is that alone (hvrt_set_deferred_accel), accel_ms libhvrt's own derived data (distance field, face "
                     "bits, per-reference records) alone, build_total_ms a normal build, where the two overlap on the GPU"}
        brk = {"count": tmb.build_count_ms, "scan": tmb.build_scan_ms, "fill": tmb.build_fill_ms, "sort": tmb.build_sort_ms,
               "distance_field+face_bits+records": tmb.build_expand_ms}
        return ms, brk, r

    if args.workload == "c5":
        # build only: a step is one build of the resident 50 M-triangle mesh into 1024^3
        if world > 1:
            raise SystemExit("c5 is a one-GPU workload (the voxel build does not shard: replicas only)")
        timed_builds(args.warmup)
        barrier()
        t0 = time.perf_counter()
        bt, tmb = timed_builds(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        ms, brk, rb = build_report(bt, tmb)
        out = dict(base, metric="voxel-grid build ms", value=dt / args.steps * 1e3, unit="ms", ms_per_step=dt / args.steps * 1e3,
                   higher_is_better=False, scaling="weak",
                   config={"workload": "c5", "mesh": "cube-sphere n=2041", "triangles": int(info.n_triangles), "grid": [g, g, g],
                           "step": "one hvrt_build_grid of the HBM-resident mesh (CSR + derived acceleration data)"},
                   voxel_build_ms=ms, voxel_build_breakdown_ms=brk, scene=scene, roofline=rb)
        if not args.no_cpu_baseline:
            import oracle
            sub = tris[: tris.shape[0] // 16]                        # a bounded sample: 1/16 of the triangles, same grid
            t1 = time.perf_counter()
            oracle.voxelize(sub, g, g, g, mn, float(vs3[0]))
            bt1 = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": bt1 * 1e3 / 2.0 * 16, "unit": "ms (extrapolated x16)", "cores": 1, "kind": "port",
                                   "sample": f"oracle/cpu_ref.c count+fill of the first 1/16 of the triangles ({sub.shape[0]}) into the same "
                                             f"1024^3 grid took {bt1 / 2.0:.2f} s; the reference's own nested-vector build cannot hold 1024^3 in 64 GB"}
        print(json.dumps(out), flush=True)
        ctx.close()
        return

    pos, la, up, fov, asp = capi.default_camera(mn, mx, "front", aspect=W / H)
    cam = capi.make_camera_frame(pos, la, up, fov, asp)

    build_ms = None
    if rank == 0:
        bt, tmb = timed_builds(5)
        build_ms, build_brk, roof_build = build_report(bt, tmb)

    # ---- per-ray work of this frame (V voxels, T triangle tests) from the kernel's own counters; this is also the first
    # launch of the shape on this context (no tile history): its duration is cold_ms (counting kernel, hence an upper bound)
    rgb_d = torch.empty((H, W, 3), dtype=torch.float32, device=device)
    hit_d = torch.empty((H, W), dtype=torch.int32, device=device)
    t_d = torch.empty((H, W), dtype=torch.float32, device=device)
    cnt_d = torch.zeros(2, dtype=torch.int64, device=device)
    torch.cuda.synchronize(device)
    ctx.render_part_device(cam, W, H, spp, seed, H, 0, 1, rgb_d.data_ptr(), hit_d.data_ptr(), t_d.data_ptr(), cnt_d.data_ptr())
    ctx.synchronize()
    V, T = (cnt_d.cpu().numpy() / float(W * H * spp)).tolist()
    hit_frac = float((hit_d != -1).float().mean().item())
    bytes_per_ray = 8.0 * V + 40.0 * T + (12.0 + 8.0) / spp          # +8 because hit id and t are written
    scene.update({"voxels_per_ray": V, "tri_tests_per_ray": T, "hit_fraction": hit_frac})

    cold_ms = unordered_ms = no_cover_ms = None
    if world == 1:
        # first launch of a shape (no tile history: the give-up variant + the split kernel behind it), plain kernel
        ctx.build_grid(g, g, g, mn, mx, vs3)                           # a new grid forgets the history
        ctx.render_part_device(cam, W, H, spp, seed, H, 0, 1, rgb_d.data_ptr(), hit_d.data_ptr(), t_d.data_ptr(), 0)
        ctx.synchronize()
        cold_ms = float(ctx.timings().render_ms)
        # the same frame with nothing carried over from one launch to the next: no tile history (HVRT_TILE_ORDER=0) and the
        # tile cover worked out again by every launch (HVRT_TILE_COVER_CACHE=0) -- and, separately, without the tile cover at
        # all (every tile walked, with history: what rounds 1 and 2 measured).  The switches are read when a context is created.
        def other_context(env, n=4):
            for k, v in env.items():
                os.environ[k] = v
            try:
                cx = ra.Context(local_rank)
            finally:
                for k in env:
                    del os.environ[k]
            a, o, i = ctx.device_arrays()
            cx.set_mesh_device(a, info.n_triangles)
            cx.set_grid_csr_device(g, g, g, mn, mx, vs3, o, i, info.n_refs)
            ms = []
            for _ in range(n):
                cx.render_part_device(cam, W, H, spp, seed, H, 0, 1, rgb_d.data_ptr(), hit_d.data_ptr(), t_d.data_ptr(), 0)
                cx.synchronize()
                ms.append(cx.timings().render_ms)
            cx.close()
            return float(np.mean(ms[1:]))
        unordered_ms = other_context({"HVRT_TILE_ORDER": "0", "HVRT_TILE_COVER_CACHE": "0"})
        no_cover_ms = other_context({"HVRT_TILE_COVER": "0"}, n=6)

    kernel_ms = []
    if world == 1:
        def step():
            ctx.render_part_device(cam, W, H, spp, seed, H, 0, 1, rgb_d.data_ptr(), hit_d.data_ptr(), t_d.data_ptr(), 0)
        rays_per_step = W * H * spp
        scaling = "weak"
    elif args.scaling == "strong":
        dr.alloc(W, H, n_frames=1, rgb8=True)

        def step():
            dr.render_frames([cam], spp=spp, seed=seed)
        rays_per_step = W * H * spp
        scaling = "strong"
    else:
        # weak scaling: a batch of `world` frames per step; every rank traces its row tiles of every frame
        # (one frame's worth of rays per rank), one all_to_all hands frame f to rank f, which reassembles it.
        # The exchange of batch k overlaps the traversal of batch k+1.
        dr.alloc_batch(W, H, rgb8=True)
        cams = [cam] * world

        def step():
            dr.render_batch(cams, spp=spp, seed=seed)
        rays_per_step = W * H * spp * world
        scaling = "weak"

    for _ in range(args.warmup):
        step()
    if world > 1 and scaling == "weak":
        dr.flush()
    barrier()
    t0 = time.perf_counter()
    if world == 1:
        # the K launches are queued back to back (each followed, on the GPU, by the tile order for the next one) and
        # waited for once; the per-launch kernel durations come from the pairs of HIP events libhvrt records around
        # every traversal on its own stream (hvrt_render_history keeps the last 32), read after the stream drained.
        for k in range(args.steps):
            step()
            if (k + 1) % 32 == 0 or k + 1 == args.steps:
                kernel_ms.extend(ctx.render_history((k % 32) + 1))
    else:
        for _ in range(args.steps):
            step()
        if scaling == "weak":
            dr.flush()             # the last batch's exchange + reassembly belong to the timed region
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- a camera that moves: every frame differs from the one whose tile durations ordered its queue ------------------
    # (the timed region above renders ONE camera K times; the history is keyed by the launch shape, not by the camera, so
    # an orbit of 1 degree per frame about the look-at point shows what the previous frame's durations are worth for the next)
    moving = None
    if world == 1 and args.moving_camera:
        n_mov = min(args.steps, 32)
        axis = np.asarray(up, np.float64) / np.linalg.norm(up)
        rel = np.asarray(pos, np.float64) - np.asarray(la, np.float64)
        cams_mov = []
        for k in range(1, n_mov + 1):
            a = np.deg2rad(float(k))
            rot = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
            cams_mov.append(capi.make_camera_frame((np.asarray(la, np.float64) + rot).astype(np.float32), la, up, fov, asp))
        ctx.synchronize()
        t1 = time.perf_counter()
        for cm in cams_mov:
            ctx.render_part_device(cm, W, H, spp, seed, H, 0, 1, rgb_d.data_ptr(), hit_d.data_ptr(), t_d.data_ptr(), 0)
        ctx.synchronize()
        mov_ms = (time.perf_counter() - t1) * 1e3 / n_mov
        mov_k = ctx.render_history(n_mov)
        for _ in range(6):                  # the orbit's last camera, standing still: what that view costs with its own history
            ctx.render_part_device(cams_mov[-1], W, H, spp, seed, H, 0, 1, rgb_d.data_ptr(), hit_d.data_ptr(), t_d.data_ptr(), 0)
        ctx.synchronize()
        still_k = ctx.render_history(4)
        moving = {"ms_per_frame": mov_ms, "kernel_ms": float(np.mean(mov_k)), "kernel_ms_last_frame": float(mov_k[-1]) if len(mov_k) else None,
                  "kernel_ms_last_view_standing_still": float(np.mean(still_k)), "frames": n_mov,
                  "Mrays_s": W * H * spp / (mov_ms * 1e-3) / 1e6,
                  "what": "the camera orbits the look-at point by 1 degree per frame (queued launches); each frame's queue is "
                          "ordered by the previous, different frame's tile durations"}

    # ---- the interface the reference calls (IRayTracer::render -> hvrt_render: host buffers, PCIe included) -----------
    interface_ms = None
    if world == 1 and rank == 0 and args.workload in ("c2", "small"):
        host_rgb = np.ones((H, W, 3), np.float32)           # touched pages, like a caller's zero-filled std::vector<Vec3>
        import ctypes as C
        lib = capi.load_library()
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            rc = lib.hvrt_render(ctx._h, C.byref(cam), W, H, spp, seed, host_rgb.ctypes.data_as(C.c_void_p), None, None, None)
            ts.append((time.perf_counter() - t1) * 1e3)
            assert rc == 0
        interface_ms = {"rgb_first_call": ts[0], "rgb_warm": float(min(ts[1:])),
                        "what": "hvrt_render(rgb only) into the caller's pageable buffer, what HipVoxelRayTracer::render does; "
                                "PCIe-inclusive, never `value`"}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = rays_per_step / (ms_per_step * 1e-3) / 1e6
        out = dict(base, metric="primary Mrays/s (node) at 4K" if args.workload == "c2" else "primary Mrays/s (node)",
                   value=value, unit="Mrays/s", ms_per_step=ms_per_step, higher_is_better=True, scaling=scaling,
                   config={"workload": args.workload, "mesh": " ".join(str(v) for v in mesh_spec), "triangles": int(info.n_triangles),
                           "grid": [g, g, g], "image": [W, H], "spp": spp, "camera": "reference front rule, aspect=W/H",
                           "frames_per_step": world if scaling == "weak" else 1, "tile_rows": 16,
                           "outputs": "rgb+hit_id+t" if world == 1 else "rgb, exchanged as 8-bit slabs"},
                   voxel_build_ms=build_ms, voxel_build_breakdown_ms=build_brk, scene=scene)
        if world > 1:
            # rank 0's last traversal launch (HIP events on libhvrt's stream): its row tiles of the step's frame(s).
            # The busy fractions are the one-GPU profile's (same kernel, same scene); only kernel_ms and the rates are live.
            k_ms = float(ctx.timings().render_ms)
            rays0 = dr.part.rows_of(rank) * W * spp * (world if scaling == "weak" else 1)
            prof = dict(load_profile(args.workload))
            if prof.get("k_render_hbm_bytes_per_launch"):       # counted on a whole frame: this launch's share of it
                prof["k_render_hbm_bytes_per_launch"] *= rays0 / float(W * H * spp)
            out["roofline"] = dict(render_roofline(prof, k_ms, rays0, bytes_per_ray),
                                   launch="rank 0's last launch of the timed region: %d rays (its row tiles of %s)"
                                          % (rays0, "the batch's %d frames" % world if scaling == "weak" else "the frame"))
        if kernel_ms:
            k_ms = float(np.mean(kernel_ms))
            out["roofline"] = render_roofline(load_profile(args.workload), k_ms, W * H * spp, bytes_per_ray)
            out["cold_ms"] = cold_ms                     # first launch of the shape, no tile history
            out["unordered_ms"] = unordered_ms           # HVRT_TILE_ORDER=0, HVRT_TILE_COVER_CACHE=0: every launch like a first one
            out["unordered_Mrays_s"] = W * H * spp / (unordered_ms * 1e-3) / 1e6 if unordered_ms else None
            # what holds whatever the previous frame was: no tile history, the tile cover worked out by the launch itself
            # (`value` is a replay of one camera: its tile durations order the queue, its cover stands)
            out["value_unconditional"] = out["unordered_Mrays_s"]
            out["no_tile_cover_ms"] = no_cover_ms        # HVRT_TILE_COVER=0: every tile walked (with history), as in rounds 1 and 2
            if no_cover_ms and load_profile(args.workload + "_no_cover"):
                # the same kernel where it is bound by throughput: every tile walked (its own counter passes, this run's duration)
                out["roofline_no_tile_cover"] = render_roofline(load_profile(args.workload + "_no_cover"), no_cover_ms, W * H * spp, bytes_per_ray)
            out["no_tile_cover_Mrays_s"] = W * H * spp / (no_cover_ms * 1e-3) / 1e6 if no_cover_ms else None
            if moving:
                out["moving_camera"] = moving
        out["roofline_build"] = roof_build
        if interface_ms:
            out["interface_ms"] = interface_ms
        if world == 1 and not args.no_cpu_baseline:
            off, idx = ctx.get_csr()
            if tris is None:
                tris = make_mesh(mesh_spec)
            out["cpu_baseline"] = cpu_baseline(tris, g, mn, float(vs3[0]), off, idx, cam, W, H, spp, seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
