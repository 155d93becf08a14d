"""The C++ multi-GPU layer (crt_mgpu_*, mgpu.cpp) on the one-GPU box:
  * a 1-rank communicator through the real RCCL calls (dlopen of librccl, ncclCommInitAll, the in-place ncclBroadcast of
    the scene blob): the frame must equal the single-context frame;
  * 2, 3 and 8 LOGICAL ranks on device 0 (RCCL refuses duplicate devices, so the exchanges are device-to-device copies;
    contexts, per-rank host threads, tile lists, pack, gather buffers and unpack are the code real ranks run): the
    assembled frame must equal the single-context frame byte for byte, also after an animation step + refit + re-sync.
  * crt_mgpu_animate (the per-frame update replicated on every rank, no blob traffic) and crt_mgpu_submit / crt_mgpu_finish
    (three frames in flight per rank, gathers under later renders) against one context running the reference's loop;
  * grouped ncclSend / ncclRecv: on the 1-rank communicator rank 0 sends its own tiles to itself (the calls have no other
    way to run on one GPU).
N > 1 over RCCL itself needs N GPUs: unmeasured here."""
import ctypes as C

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


def _frame_at(ptr, w, h):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    a = np.zeros((h, w, 3), np.uint8)
    assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), ptr, a.nbytes, 2) == 0
    return a


@pytest.mark.parametrize("world,rccl", [(3, False), (1, True)])
def test_animate_on_every_rank_and_frames_in_flight(world, rccl):
    """25 steps of the reference's loop (animation + refit, a device rebuild at step 20: main.cu:534-541) with every frame
    submitted through crt_mgpu_submit: each frame equals the one a single context renders at that step."""
    import cudaraytracer_amd as crt
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_spheres(300, seed=4), crt.scene_soup(3000, seed=9)])
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    w, h, steps = 333, 205, 25
    p = crt.default_params(width=w, height=h, bounces=2)
    ref = crt.Context(0)
    want = []
    try:
        ref.set_scene(prims); ref.set_lights(lights); ref.build(0)
        for f in range(1, steps + 1):
            ref.animate_device()
            if f % 20 == 0:
                ref.build(2)
            else:
                ref.refit(False)
            want.append(ref.render(cam, tgt, p).copy())
    finally:
        ref.close()
    m = crt.MultiGpu([0] * world, rccl_always=rccl)
    try:
        assert m.uses_rccl == rccl
        m.set_scene(prims); m.root.build(0); m.set_lights(lights); m.sync_scene()
        got = []
        for f in range(1, steps + 1):
            m.animate(rebuild=(f % 20 == 0))
            ptr = m.submit(cam, tgt, p)
            if ptr:
                got.append(_frame_at(ptr, w, h))
            if f == 10:                                    # the synchronous entry point in between drains the pipeline first
                last = m.finish()
                if last:
                    got.append(_frame_at(last, w, h))
                assert np.array_equal(m.render(cam, tgt, p), want[9])
        last = m.finish()
        assert last
        if len(got) < steps:
            got.append(_frame_at(last, w, h))
        # frames complete in submission order; submit() reports the newest frame it completed, so some are skipped over:
        # every reported frame must be one of the wanted ones, in order, and the last must be the last
        idx = 0
        for g in got:
            while idx < steps and not np.array_equal(g, want[idx]):
                idx += 1
            assert idx < steps, "a frame returned by the pipeline matches no step of the single-context loop"
        assert np.array_equal(got[-1], want[-1])
        assert not np.array_equal(want[0], want[-1])
    finally:
        m.close()


def test_animate_needs_the_scene_on_every_rank():
    import cudaraytracer_amd as crt
    m = crt.MultiGpu([0, 0])
    try:
        m.root.set_scene(crt.scene_teapot()); m.root.build(0); m.sync_scene()
        with pytest.raises(crt.CrtError):
            m.animate()
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
