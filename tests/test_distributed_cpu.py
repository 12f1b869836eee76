"""N > 1 choreography on CPU: world_size-2 (and 3) gloo processes run the same partition -> gather -> reassemble
sequence as the GPU path (raytracer_amd/distributed.py).  The per-rank "render" here is the oracle restricted
to the rank's rows (there is no GPU in this container); what is under test is the row-tile partition, the
collective and the reassembly: the N-rank frame must equal the 1-rank frame bytewise."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tile_rows, W, H, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from raytracer_amd import capi, meshgen
    from raytracer_amd.distributed import Partition, gather_parts

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # scene replication: rank 0 owns the mesh, broadcasts it (the GPU path broadcasts mesh + CSR with RCCL)
        if rank == 0:
            tris = torch.from_numpy(meshgen.cube_sphere(10))
            n = torch.tensor([tris.shape[0]])
        else:
            n = torch.zeros(1, dtype=torch.int64)
        dist.broadcast(n, 0)
        if rank != 0:
            tris = torch.empty((int(n), 12), dtype=torch.float32)
        dist.broadcast(tris, 0)
        sc, mn, mx, vs = oracle.build_scene(tris.numpy(), 16, 16, 16)
        pos, la, up, fov, asp = capi.default_camera(mn, mx)
        fr = oracle.camera_frame(pos, la, up, fov, asp)

        part = Partition(H, tile_rows, world)
        assert part.rows_of(rank) == capi.part_rows(H, tile_rows, rank, world)
        rows = part.global_rows(rank)
        slab = torch.zeros((part.slab_rows, W, 3), dtype=torch.float32)
        for lr, y in enumerate(rows):                      # "render my tiles into my compact slab"
            rgb, _, _, _ = sc.render(fr, W, H, threads=1, rows=(int(y), int(y) + 1))
            slab[lr] = torch.from_numpy(rgb[y])
        gathered = gather_parts(slab)                      # the one collective of the path
        frame = np.zeros((H, W, 3), np.float32)
        for p in range(world):                             # reassembly (hvrt_assemble_parts_device on the GPU)
            pr = part.global_rows(p)
            frame[pr] = gathered[p, :len(pr)].numpy()
        # the same exchange moving 8-bit pixels (what the GPU path sends by default: quantised before the collective
        # with ImageSaver.h's rule, 3 bytes per pixel instead of 12)
        q = lambda a: np.clip(a * np.float32(255.0), 0, 255).astype(np.int32).astype(np.uint8)
        gathered8 = gather_parts(torch.from_numpy(q(slab.numpy())))
        frame8 = np.zeros((H, W, 3), np.uint8)
        for p in range(world):
            pr = part.global_rows(p)
            frame8[pr] = gathered8[p, :len(pr)].numpy()
        assert gathered8.dtype == torch.uint8 and np.array_equal(frame8, q(frame))
        if rank == 0:
            full, _, _, _ = sc.render(fr, W, H, threads=2)
            np.save(os.path.join(out_dir, "ok.npy"), np.array([np.array_equal(frame.view(np.uint32), full.view(np.uint32))]))
        # every rank holds the same frame after the all_gather
        chk = torch.from_numpy(frame.view(np.int32).astype(np.int64).sum(keepdims=True).reshape(1))
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, dist.ReduceOp.MIN)
        dist.all_reduce(hi, dist.ReduceOp.MAX)
        assert int(lo) == int(hi)
    finally:
        dist.destroy_process_group()


def _worker_batch(rank, world, port, tile_rows, W, H, out_dir):
    """Batch mode: `world` frames (different cameras), every rank renders its tiles of every frame, ONE
    all_to_all (exchange_parts) hands frame f's slabs to rank f, which reassembles and checks it."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from raytracer_amd import capi, meshgen
    from raytracer_amd.distributed import Partition, exchange_parts

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, mn, mx, vs = oracle.build_scene(meshgen.cube_sphere(10), 16, 16, 16)
        views = ["front", "side", "top"]
        frs = [oracle.camera_frame(*capi.default_camera(mn, mx, views[f % 3])) for f in range(world)]
        part = Partition(H, tile_rows, world)
        rows = part.global_rows(rank)
        local = torch.zeros((world, part.slab_rows, W, 3), dtype=torch.float32)
        for f in range(world):
            for lr, y in enumerate(rows):
                rgb, _, _, _ = sc.render(frs[f], W, H, threads=1, rows=(int(y), int(y) + 1))
                local[f, lr] = torch.from_numpy(rgb[y])
        recv, work = exchange_parts(local, async_op=True)
        if work is not None:
            work.wait()
        frame = np.zeros((H, W, 3), np.float32)
        for p in range(world):
            pr = part.global_rows(p)
            frame[pr] = recv[p, :len(pr)].numpy()
        full, _, _, _ = sc.render(frs[rank], W, H, threads=1)     # the frame this rank owns
        ok = torch.tensor([int(np.array_equal(frame.view(np.uint32), full.view(np.uint32)))])
        dist.all_reduce(ok, dist.ReduceOp.MIN)
        if rank == 0:
            np.save(os.path.join(out_dir, "ok_batch.npy"), ok.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows,H", [(2, 16, 90), (3, 8, 50)])
def test_batch_all_to_all_gives_every_rank_its_frame(tmp_path, world, tile_rows, H):
    port = _free_port()
    mp.spawn(_worker_batch, args=(world, port, tile_rows, 48, H, str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "ok_batch.npy")[0] == 1


@pytest.mark.parametrize("world,tile_rows,H", [(2, 16, 90), (2, 7, 45), (3, 8, 50)])
def test_partition_gather_reassemble_equals_single_rank(tmp_path, world, tile_rows, H):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, tile_rows, 64, H, str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "ok.npy")[0]


def test_partition_host_arithmetic():
    sys.path.insert(0, ROOT)
    from raytracer_amd.distributed import Partition
    for H, tile, world in [(2160, 16, 8), (2160, 16, 3), (50, 8, 3), (7, 16, 2), (4320, 16, 8)]:
        p = Partition(H, tile, world)
        allrows = np.concatenate([p.global_rows(r) for r in range(world)])
        assert sorted(allrows.tolist()) == list(range(H))
        assert all(len(p.global_rows(r)) == p.rows_of(r) <= p.slab_rows for r in range(world))
