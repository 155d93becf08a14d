"""The multi-GPU host path itself (parallel.ShardedRenderer: blob broadcast, import, per-rank tiles, pack,
gather, unpack) rehearsed as 3 processes on ONE GPU over gloo; RCCL replaces gloo on a real node."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,mode", [(3, 0), (2, 2)])
def test_sharded_renderer_gathers_the_single_gpu_frame(tmp_path, world, mode):
    import cudaraytracer_amd as crt
    w, h, bounces = 450, 270, 3
    out = str(tmp_path / "frame.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29530 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), out, str(w), str(h), str(bounces), str(mode)],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_soup(5000, seed=6)])
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(mode)
    cam, tgt = crt.default_camera()
    try:
        full = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
    finally:
        ctx.close()
    assert np.array_equal(np.load(out), full)


@pytest.mark.parametrize("pipelined", [0, 1])
def test_sharded_renderer_animates_on_every_rank(tmp_path, pipelined):
    """SURVEY 8(e) collective (2): the ranks replicate the deterministic animation step (crt_animate_device) on the
    scene they were handed once, refit locally, rebuild on the device every 4th step; the gathered frame after 6
    steps equals one GPU doing the same.  pipelined: the tile gather of a frame runs under the render of the next
    (bench.py's throughput mode, double-buffered frames) -- the last frame, completed by finish(), is the same."""
    import cudaraytracer_amd as crt
    world, mode, w, h, bounces, frames = 2, 0, 300, 200, 2, 6
    out = str(tmp_path / "anim.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + pipelined), WORLD_SIZE=str(world), CRT_TEST_PIPELINED=str(pipelined))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), out, str(w), str(h), str(bounces), str(mode), str(frames)],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_soup(5000, seed=6), crt.scene_spheres(200, seed=8)])
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(mode)
    cam, tgt = crt.default_camera()
    try:
        for f in range(1, frames + 1):
            ctx.animate_device()
            if f % 4 == 0:
                ctx.build(2)
            else:
                ctx.refit(False)
        full = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
        still = crt.Context(0)
        still.set_scene(prims); still.set_lights(crt.default_lights()); still.build(mode)
        assert not np.array_equal(full, still.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces)))
        still.close()
    finally:
        ctx.close()
    assert np.array_equal(np.load(out), full)


def test_resolution_change_between_pipelined_renders(tmp_path):
    """frames of the old size are still in flight when the size changes: they are completed against their own buffers before
    frames, packed tiles and gather buffers are replaced (a smaller frame would otherwise be unpacked out of bounds); then a
    non-pipelined render after pipelined ones finishes what is pending first."""
    import cudaraytracer_amd as crt
    world, mode, w, h, bounces = 2, 0, 450, 270, 2
    w2, h2 = 200, 110
    for pipelined in (1, 0):
        out = str(tmp_path / f"resize{pipelined}.npy")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29551 + pipelined), WORLD_SIZE=str(world),
                   CRT_TEST_PIPELINED=str(pipelined), CRT_TEST_RESIZE=f"{w2}x{h2}")
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), out, str(w), str(h), str(bounces), str(mode)],
                                  env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(logs)
        prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_soup(5000, seed=6)])
        ctx = crt.Context(0)
        ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(mode)
        cam, tgt = crt.default_camera()
        try:
            full = ctx.render(cam, tgt, crt.default_params(width=w2, height=h2, bounces=bounces))
        finally:
            ctx.close()
        got = np.load(out)
        assert got.shape == (h2, w2, 3) and np.array_equal(got, full)


@pytest.mark.parametrize("pipelined", [0, 1])
def test_rccl_device_gather_on_a_one_rank_group(tmp_path, pipelined):
    """backend "nccl" (RCCL) with one rank: the blob broadcast, the DEVICE-tensor gather (gloo stages through the host), its
    asynchronous work handle and the wait on the collective's stream are the calls N ranks make; the frame must still be the
    single-context frame, also after animation steps."""
    import cudaraytracer_amd as crt
    world, mode, w, h, bounces, frames = 1, 0, 300, 200, 2, 5
    out = str(tmp_path / "rccl1.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29561 + pipelined), WORLD_SIZE="1", RANK="0", CRT_TEST_BACKEND="nccl",
               CRT_TEST_PIPELINED=str(pipelined))
    p = subprocess.run([sys.executable, os.path.join(HERE, "sharded_worker.py"), out, str(w), str(h), str(bounces), str(mode), str(frames)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_soup(5000, seed=6), crt.scene_spheres(200, seed=8)])
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(mode)
    cam, tgt = crt.default_camera()
    try:
        for f in range(1, frames + 1):
            ctx.animate_device()
            if f % 4 == 0:
                ctx.build(2)
            else:
                ctx.refit(False)
        full = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
    finally:
        ctx.close()
    assert np.array_equal(np.load(out), full)
