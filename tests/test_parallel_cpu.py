"""N>1 path on the CPU: two ranks over gloo shard the frame's tiles, render their own tiles (the oracle
stands in for the device renderer -- test infrastructure only), gather to rank 0 and assemble.  The
assembled frame must equal the single-rank frame byte for byte."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import parallel
    import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prims = crt.scene_teapot()
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    params = crt.default_params(width=w, height=h, bounces=2, rank=rank, nranks=world)
    sc = orc.Scene(prims)
    sc.build()
    ids, _ = crt.tiles_owned(params)
    tiles_x = (w + 31) // 32
    frame = np.zeros((h, w, 3), np.uint8)
    for t in ids:                                   # render only this rank's tiles
        x0, y0 = (t % tiles_x) * 32, (t // tiles_x) * 32
        rect = [x0, y0, min(x0 + 32, w), min(y0 + 32, h)]
        part = sc.render(lights, cam, tgt, w, h, 2, rect=rect, nthreads=1)["bgr"]
        frame[rect[1]:rect[3], rect[0]:rect[2]] = part[rect[1]:rect[3], rect[0]:rect[2]]
    packed = parallel.pack_tiles_host(params, rank, frame)
    assert len(packed) == parallel.packed_nbytes(params, rank)
    full = parallel.gather_frame_host(dist, params, packed, rank, world)
    if rank == 0:
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharding_gather_equals_single_rank(tmp_path, world):
    import torch.multiprocessing as mp
    import cudaraytracer_amd as crt
    import orc
    w, h = 200, 110                                  # clipped tiles on both edges
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, w, h, out), nprocs=world, join=True)
    got = np.load(out)
    sc = orc.Scene(crt.scene_teapot())
    sc.build()
    cam, tgt = crt.default_camera()
    want = sc.render(crt.default_lights(), cam, tgt, w, h, 2, nthreads=4)["bgr"]
    assert np.array_equal(got, want)


def test_pack_unpack_roundtrip():
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import parallel
    rng = np.random.default_rng(0)
    w, h, world = 333, 205, 4
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = np.zeros_like(frame)
    for r in range(world):
        p = crt.default_params(width=w, height=h, rank=r, nranks=world)
        parallel.unpack_tiles_host(p, r, parallel.pack_tiles_host(p, r, frame), out)
    assert np.array_equal(out, frame)
