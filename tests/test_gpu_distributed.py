"""GPU rehearsal of the multi-GPU path on ONE card (the N > 1 run itself belongs to the driver).

RCCL refuses two ranks on one device, so the exchange is rehearsed in three pieces, each through the
product code that runs at N > 1:
  * scene replication: export on one context, ingest through the *_device entry points on another, render;
  * DistributedRenderer.render_frames with world = 2 and 3, with torch.distributed's all_gather replaced by
    a stand-in that supplies the other ranks' slabs (rendered beforehand through the C ABI) -- the stream
    choreography, the slab layout and the reassembly are the real ones;
  * a real one-rank "nccl" (RCCL) process group on the card: broadcast / all_gather_into_tensor / barrier
    with the argument forms distributed.py and bench.py use.
"""
import os
import socket

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import capi, meshgen

pytestmark = pytest.mark.gpu


def _scene(ctx, n=16, g=24):
    tris = meshgen.cube_sphere(n)
    ctx.set_mesh(tris)
    mn, mx = ctx.mesh_bounds()
    vs = capi.voxel_size(mn, mx, g, g, g)
    ctx.build_grid(g, g, g, mn, mx, vs)
    pos, la, up, fov, asp = capi.default_camera(mn, mx)
    return tris, mn, mx, vs, capi.make_camera_frame(pos, la, up, fov, asp)


def test_scene_replication_through_device_entry_points(ctx):
    import torch
    g = 24
    tris, mn, mx, vs, cam = _scene(ctx, 16, g)
    info = ctx.grid_info()
    dev = torch.device("cuda:0")
    tri = torch.empty(info.n_triangles * 12, dtype=torch.float32, device=dev)
    off = torch.empty(g ** 3 + 1, dtype=torch.int32, device=dev)
    idx = torch.empty(info.n_refs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.export_scene_device(tri.data_ptr(), off.data_ptr(), idx.data_ptr())
    ctx.synchronize()
    want = ctx.render(cam, 160, 120)
    other = ra.Context(0)
    try:
        other.set_mesh_device(tri.data_ptr(), info.n_triangles)
        vs3 = np.full(3, vs, np.float32)
        other.set_grid_csr_device(g, g, g, mn, mx, vs3, off.data_ptr(), idx.data_ptr(), info.n_refs)
        got = other.render(cam, 160, 120)
        a, b = ctx.get_csr(), other.get_csr()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(ctx.distance_field(), other.distance_field())
    finally:
        other.close()
    for w, gt in zip(want, got):
        assert np.array_equal(w.view(np.uint32) if w.dtype == np.float32 else w,
                              gt.view(np.uint32) if gt.dtype == np.float32 else gt)


@pytest.mark.parametrize("world,tile_rows,H", [(2, 16, 150), (3, 8, 97), (8, 16, 130)])
def test_render_frames_with_stand_in_all_gather(ctx, monkeypatch, world, tile_rows, H):
    import torch
    from raytracer_amd import distributed as D
    tris, mn, mx, vs, cam = _scene(ctx, 16, 24)
    W = 200
    dev = torch.device("cuda:0")
    full = ctx.render(cam, W, H)[0]
    cam2 = capi.make_camera_frame(*capi.default_camera(mn, mx, "side"))
    full2 = ctx.render(cam2, W, H)[0]
    part = D.Partition(H, tile_rows, world)
    # the other ranks' slabs, rendered beforehand
    slabs = {}
    for f, c in enumerate((cam, cam2)):
        for p in range(world):
            s = torch.zeros((part.slab_rows, W, 3), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.render_part_device(c, W, H, 1, 0, tile_rows, p, world, d_rgb=s.data_ptr())
            ctx.synchronize()
            slabs[(f, p)] = s
    for rank in range(world):
        calls = []

        def fake_all_gather(out, inp, group=None, async_op=False, _rank=rank, _calls=calls):
            f = len(_calls)
            _calls.append(f)
            for p in range(world):
                out[p].copy_(inp if p == _rank else slabs[(f, p)])

        monkeypatch.setattr(D.dist, "all_gather_into_tensor", fake_all_gather)
        dr = D.DistributedRenderer(ctx, rank, world, dev, tile_rows=tile_rows)
        dr.alloc(W, H, n_frames=2)
        frames = dr.render_frames([cam, cam2])
        torch.cuda.synchronize()
        got = frames.cpu().numpy()
        assert len(calls) == 2
        assert np.array_equal(got[0].view(np.uint32), full.view(np.uint32)), f"frame 0 differs on rank {rank}"
        assert np.array_equal(got[1].view(np.uint32), full2.view(np.uint32)), f"frame 1 differs on rank {rank}"


class _StreamWork:
    """What an asynchronous RCCL work handle does on wait(): the CURRENT stream waits for the collective."""
    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)
        return True


@pytest.mark.parametrize("world,tile_rows,H", [(2, 16, 150), (4, 8, 97), (8, 16, 2160 // 8)])
def test_render_batch_with_stand_in_all_to_all(ctx, monkeypatch, world, tile_rows, H):
    """render_batch's double-buffered pipeline over four batches (cameras change per batch), with the
    all_to_all replaced by a stand-in that delivers the other ranks' slabs of the frame this rank owns."""
    import torch
    from raytracer_amd import distributed as D
    tris, mn, mx, vs, cam = _scene(ctx, 16, 24)
    W = 200
    dev = torch.device("cuda:0")
    views = ["front", "side", "top"]
    cams = [capi.make_camera_frame(*capi.default_camera(mn, mx, v)) for v in views]
    fulls = [ctx.render(c, W, H)[0] for c in cams]
    part = D.Partition(H, tile_rows, world)
    slabs = {}
    for v in range(3):
        for p in range(world):
            s = torch.zeros((part.slab_rows, W, 3), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.render_part_device(cams[v], W, H, 1, 0, tile_rows, p, world, d_rgb=s.data_ptr())
            ctx.synchronize()
            slabs[(v, p)] = s
    for rank in (0, world - 1):
        batches = [[(k + f) % 3 for f in range(world)] for k in range(4)]     # view index per frame of batch k
        state = {"k": 0}

        def fake_a2a(out, inp, group=None, async_op=False, _rank=rank, _state=state):
            k = _state["k"]
            _state["k"] += 1
            v = batches[k][_rank]                      # the frame this rank owns in batch k
            for p in range(world):
                out[p].copy_(inp[_rank] if p == _rank else slabs[(v, p)])
            return _StreamWork(torch.cuda.current_stream().record_event())

        monkeypatch.setattr(D.dist, "all_to_all_single", fake_a2a)
        monkeypatch.setattr(D.dist, "get_world_size", lambda group=None: world)
        dr = D.DistributedRenderer(ctx, rank, world, dev, tile_rows=tile_rows)
        dr.alloc_batch(W, H)
        got = []
        for k in range(4):
            done = dr.render_batch([cams[v] for v in batches[k]])
            if done is not None:
                torch.cuda.synchronize()
                got.append(done.cpu().numpy().copy())
        done = dr.flush()
        torch.cuda.synchronize()
        got.append(done.cpu().numpy().copy())
        assert len(got) == 4 and state["k"] == 4
        for k in range(4):
            want = fulls[batches[k][rank]]
            assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), f"batch {k} differs on rank {rank}"


def test_one_rank_rccl_group_collectives_and_renderer(ctx):
    """backend "nccl" is RCCL here: the calls distributed.py / bench.py make, on a real one-rank group."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import distributed as D
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = torch.arange(1024, dtype=torch.float32, device=dev)
        dist.broadcast(x, src=0)
        out = torch.empty((1, 1024), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out, x)
        tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(out[0], x) and float(tt.item()) == 1.5
        assert torch.equal(D.gather_parts(x)[0], x)
        dr = D.DistributedRenderer(ctx, 0, 1, dev, tile_rows=16)
        tris = meshgen.cube_sphere(16)
        mn, mx, vs3 = dr.replicate_scene(tris, (24, 24, 24))
        cam = capi.make_camera_frame(*capi.default_camera(mn, mx))
        dr.alloc(200, 150, n_frames=2)
        frames = dr.render_frames([cam, cam])
        torch.cuda.synchronize()
        want = ctx.render(cam, 200, 150)[0]
        assert np.array_equal(frames[0].cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert np.array_equal(frames[1].cpu().numpy().view(np.uint32), want.view(np.uint32))
        # batch mode on the same one-rank group, and the all_to_all call form itself
        y = torch.empty_like(out)
        h = dist.all_to_all_single(y, out, async_op=True)
        h.wait()
        torch.cuda.synchronize()
        assert torch.equal(y, out)
        dr.alloc_batch(200, 150)
        assert dr.render_batch([cam]) is None
        first = dr.render_batch([cam])
        torch.cuda.synchronize()
        assert np.array_equal(first.cpu().numpy().view(np.uint32), want.view(np.uint32))
        last = dr.flush()
        torch.cuda.synchronize()
        assert np.array_equal(last.cpu().numpy().view(np.uint32), want.view(np.uint32))
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, out_dir):
    """One of two real processes sharing the one GPU.  RCCL cannot put two ranks on one device, so the wire is
    gloo with the tensors staged through host memory; everything else -- replicate_scene on a rank that does
    NOT hold the mesh, the batch pipeline, the reassembly -- is the code bench.py runs at N > 1."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from raytracer_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real_broadcast, real_a2a = dist.broadcast, dist.all_to_all_single

    class Done:
        def wait(self):
            return True

    def staged_broadcast(t, src=0, group=None, async_op=False):
        if t.is_cuda:
            torch.cuda.synchronize()
            h = t.cpu()
            real_broadcast(h, src=src, group=group)
            t.copy_(h)
            torch.cuda.synchronize()
        else:
            real_broadcast(t, src=src, group=group)

    def staged_a2a(out, inp, group=None, async_op=False):
        torch.cuda.synchronize()           # the caller ordered the collective after the render on `cur`
        hi, ho = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        real_a2a(ho, hi, group=group)
        out.copy_(ho)
        torch.cuda.synchronize()
        return Done()

    D.dist.broadcast = staged_broadcast
    D.dist.all_to_all_single = staged_a2a
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        ctx = ra.Context(0)
        dr = D.DistributedRenderer(ctx, rank, world, dev, tile_rows=16)
        tris = meshgen.cube_sphere(20) if rank == 0 else None
        mn, mx, vs3 = dr.replicate_scene(tris, (32, 32, 32))
        info = ctx.grid_info()
        off, idx = ctx.get_csr()
        chk = torch.tensor([int(off.astype(np.int64).sum()), int(idx.astype(np.int64).sum()), int(info.n_refs)])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, dist.ReduceOp.MIN)
        dist.all_reduce(hi, dist.ReduceOp.MAX)
        assert torch.equal(lo, hi) and int(info.n_refs) > 0, "scene differs between ranks"
        W, H = 203, 150
        views = ["front", "side", "top"]
        cams = [capi.make_camera_frame(*capi.default_camera(mn, mx, v)) for v in views]
        dr.alloc_batch(W, H)
        batches = [[(k + f) % 3 for f in range(world)] for k in range(3)]
        got = []
        for k in range(3):
            done = dr.render_batch([cams[v] for v in batches[k]])
            if done is not None:
                torch.cuda.synchronize()
                got.append(done.cpu().numpy().copy())
        done = dr.flush()
        torch.cuda.synchronize()
        got.append(done.cpu().numpy().copy())
        ok = len(got) == 3
        for k in range(3):
            want = ctx.render(cams[batches[k][rank]], W, H)[0]        # this rank owns frame index `rank`
            ok = ok and np.array_equal(got[k].view(np.uint32), want.view(np.uint32))
        flag = torch.tensor([int(ok)])
        dist.all_reduce(flag, dist.ReduceOp.MIN)
        if rank == 0:
            np.save(os.path.join(out_dir, "ok.npy"), flag.numpy())
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_share_the_gpu_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "ok.npy")[0] == 1


@pytest.mark.parametrize("world,tile_rows,H", [(2, 16, 150), (8, 16, 130)])
def test_render_frames_exchanging_8_bit_pixels(ctx, monkeypatch, world, tile_rows, H):
    """rgb8=True: every rank quantises its slab on the device (ImageSaver.h's rule) BEFORE the collective, 3 bytes
    per pixel cross the wire instead of 12, and the reassembled 8-bit frame is the 1-GPU float frame quantised
    (SURVEY.md 8f row 3).  Stand-in all_gather as above; also the batch pipeline's buffers in this mode."""
    import torch
    from raytracer_amd import distributed as D
    from test_gpu_render import _quantise_ref
    tris, mn, mx, vs, cam = _scene(ctx, 16, 24)
    W = 200
    dev = torch.device("cuda:0")
    want = _quantise_ref(ctx.render(cam, W, H)[0])
    part = D.Partition(H, tile_rows, world)
    slabs8 = {}
    for p in range(world):
        s = torch.zeros((part.slab_rows, W, 3), dtype=torch.float32, device=dev)
        s8 = torch.zeros((part.slab_rows, W, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.render_part_device(cam, W, H, 1, 0, tile_rows, p, world, d_rgb=s.data_ptr())
        ctx.quantise_rgb8_device(s.data_ptr(), s.numel(), s8.data_ptr())
        ctx.synchronize()
        slabs8[p] = s8
    for rank in (0, world - 1):
        seen = []

        def fake_all_gather(out, inp, group=None, async_op=False, _rank=rank):
            seen.append((inp.dtype, inp.numel()))
            for p in range(world):
                out[p].copy_(inp if p == _rank else slabs8[p])

        monkeypatch.setattr(D.dist, "all_gather_into_tensor", fake_all_gather)
        dr = D.DistributedRenderer(ctx, rank, world, dev, tile_rows=tile_rows)
        dr.alloc(W, H, n_frames=1, rgb8=True)
        frames = dr.render_frames([cam])
        torch.cuda.synchronize()
        assert seen == [(torch.uint8, part.slab_rows * W * 3)]          # 3 bytes per pixel on the wire
        assert frames.dtype == torch.uint8 and np.array_equal(frames[0].cpu().numpy(), want), f"rank {rank}"
    with pytest.raises(ValueError):
        D.DistributedRenderer(ctx, 0, world, dev, tile_rows=tile_rows).alloc(201, H, rgb8=True)
    with pytest.raises(capi.HvrtError):          # a 3-byte row that is not a whole number of words
        ctx.assemble_parts_device(slabs8[0].data_ptr(), 201, H, tile_rows, world, part.slab_rows, 3, slabs8[0].data_ptr())
