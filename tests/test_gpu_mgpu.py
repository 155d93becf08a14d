"""The C++ multi-GPU layer (crt_mgpu_*, mgpu.cpp) on the one-GPU box:
  * a 1-rank communicator through the real RCCL calls (dlopen of librccl, ncclCommInitAll, the in-place ncclBroadcast of
    the scene blob): the frame must equal the single-context frame;
  * 2, 3 and 8 LOGICAL ranks on device 0 (RCCL refuses duplicate devices, so the exchanges are device-to-device copies;
    contexts, per-rank host threads, tile lists, pack, gather buffers and unpack are the code real ranks run): the
    assembled frame must equal the single-context frame byte for byte, also after an animation step + refit + re-sync.
N > 1 over RCCL itself needs N GPUs: unmeasured here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def single(crt, prims, lights, cam, tgt, params, steps=0):
    ctx = crt.Context(0)
    try:
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
        for _ in range(steps):
            ctx.animate_device(); ctx.refit(False)
        return ctx.render(cam, tgt, params).copy(), ctx.stats()
    finally:
        ctx.close()


def test_one_rank_communicator_uses_rccl():
    import cudaraytracer_amd as crt
    prims = crt.scene_teapot(with_sphere=True)
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    p = crt.default_params(width=400, height=200, bounces=3)
    want, _ = single(crt, prims, lights, cam, tgt, p)
    m = crt.MultiGpu([0], rccl_always=True)
    try:
        assert m.size == 1 and m.uses_rccl
        m.root.set_scene(prims); m.root.build(0)
        m.set_lights(lights)
        m.sync_scene()                                  # ncclBroadcast on the 1-rank communicator
        m.render_device(cam, tgt, p)                    # the device path (no transfers to group with one rank)
        got = m.render(cam, tgt, p)
        assert np.array_equal(got, want)
    finally:
        m.close()


@pytest.mark.parametrize("world,w,h", [(2, 720, 360), (3, 333, 205), (8, 1280, 720)])
def test_logical_ranks_in_one_process(world, w, h):
    import cudaraytracer_amd as crt
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_spheres(200, seed=2)])
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    p = crt.default_params(width=w, height=h, bounces=3)
    want0, st0 = single(crt, prims, lights, cam, tgt, p)
    want2, _ = single(crt, prims, lights, cam, tgt, p, steps=2)
    m = crt.MultiGpu([0] * world)
    try:
        assert m.size == world and not m.uses_rccl
        m.root.set_scene(prims); m.root.build(0)
        m.set_lights(lights)
        m.sync_scene()
        got = m.render(cam, tgt, p)
        assert np.array_equal(got, want0)
        rays = sum(m.rank_stats(r)["rays_primary"] + m.rank_stats(r)["rays_shadow"] + m.rank_stats(r)["rays_reflect"] for r in range(world))
        assert rays == st0["rays_primary"] + st0["rays_shadow"] + st0["rays_reflect"]
        t = m.times()
        assert t["ms_render"] > 0 and t["ms_exchange"] >= 0
        for _ in range(2):                              # the scene moves on rank 0; one broadcast replicates it
            m.root.animate_device(); m.root.refit(False)
        m.sync_scene()
        assert np.array_equal(m.render(cam, tgt, p), want2)
    finally:
        m.close()


def test_mgpu_errors():
    import cudaraytracer_amd as crt
    with pytest.raises(crt.CrtError):
        crt.MultiGpu([0, 99])
    m = crt.MultiGpu([0, 0])
    try:
        with pytest.raises(crt.CrtError):               # nothing built on rank 0 yet
            m.sync_scene()
    finally:
        m.close()
