"""crt_render_submit / crt_render_wait (include/crt.h): frames in flight on several frame slots equal the frames the
synchronous entry point renders, byte for byte -- whole frames on different slots, one frame split over slots by tile
(crt_params.tile_sub / tile_nsub), with the persistent grid whole or divided, and with the scene animated between
submissions (the device orders the refit behind frames already submitted)."""
import ctypes as C

import numpy as np
import pytest

import cudaraytracer_amd as crt

pytestmark = pytest.mark.gpu


def _dev_frame(ctx, slot_ptrs, nbytes):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = []
    for ptr in slot_ptrs:
        a = np.zeros(nbytes, np.uint8)
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), ptr, nbytes, 2) == 0
        out.append(a)
    return out


def _alloc(nbytes, n):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    ptrs = []
    for _ in range(n):
        p = C.c_void_p(0)
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        assert hip.hipMemset(p, 0, nbytes) == 0
        ptrs.append(p.value)
    return hip, ptrs


@pytest.fixture(scope="module")
def soup_ctx():
    prims = crt.scene_soup(30000, seed=5)
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
    yield ctx
    ctx.close()


def test_frames_on_four_slots_equal_the_synchronous_frame(soup_ctx):
    ctx = soup_ctx
    cam, tgt = crt.default_camera()
    W, H = 640, 360
    p = crt.default_params(width=W, height=H, bounces=3)
    want = ctx.render(cam, tgt, p)
    hip, ptrs = _alloc(W * H * 3, 4)
    try:
        for div in (0, 4):
            q = crt.default_params(width=W, height=H, bounces=3, grid_div=div, timing=1)
            for k in range(4):
                ctx.render_submit(k, cam, tgt, q, ptrs[k])
            for k in range(4):
                ctx.render_wait(k)
                st = ctx.stats()
                assert st["rays_primary"] == W * H and st["ms_total"] > 0
            for got in _dev_frame(ctx, ptrs, W * H * 3):
                assert np.array_equal(got.reshape(H, W, 3), want)
    finally:
        for q in ptrs:
            hip.hipFree(C.c_void_p(q))


def test_one_frame_split_over_slots_by_tile(soup_ctx):
    ctx = soup_ctx
    cam, tgt = crt.default_camera()
    W, H = 650, 366          # clipped tiles on both edges
    p = crt.default_params(width=W, height=H, bounces=2)
    want = ctx.render(cam, tgt, p)
    hip, ptrs = _alloc(W * H * 3, 1)
    try:
        for K in (2, 3):
            hip.hipMemset(ptrs[0], 0, W * H * 3)
            rays = 0
            for k in range(K):
                ctx.render_submit(k, cam, tgt, crt.default_params(width=W, height=H, bounces=2, tile_sub=k, tile_nsub=K), ptrs[0])
            for k in range(K):
                ctx.render_wait(k)
                rays += ctx.stats()["rays_primary"]
            assert rays == W * H
            assert np.array_equal(_dev_frame(ctx, ptrs, W * H * 3)[0].reshape(H, W, 3), want)
    finally:
        hip.hipFree(C.c_void_p(ptrs[0]))


def test_slot_errors(soup_ctx):
    ctx = soup_ctx
    cam, tgt = crt.default_camera()
    p = crt.default_params(width=64, height=64, bounces=1)
    with pytest.raises(crt.CrtError):
        ctx.render_wait(1)                       # nothing submitted
    with pytest.raises(crt.CrtError):
        ctx.render_submit(9, cam, tgt, p)         # slot out of range
    ctx.render_submit(2, cam, tgt, p)
    with pytest.raises(crt.CrtError):
        ctx.render_submit(2, cam, tgt, p)        # slot busy
    ctx.render_wait(2)
    with pytest.raises(crt.CrtError):
        ctx.render_submit(1, cam, tgt, crt.default_params(width=64, height=64, bounces=1, use_graph=1))
    with pytest.raises(crt.CrtError):
        ctx.render_submit(0, cam, tgt, crt.default_params(width=64, height=64, bounces=1, tile_sub=2, tile_nsub=2))


def test_animation_is_ordered_between_frames_in_flight():
    """frame k on slot k % 3 while the scene moves between submissions: every frame equals the synchronous loop's"""
    prims = crt.scene_spheres(1500, seed=3)
    cam, tgt = crt.default_camera()
    W, H, N = 320, 180, 7
    p = crt.default_params(width=W, height=H, bounces=2)
    ref = crt.Context(0); ref.set_scene(prims); ref.set_lights(crt.default_lights()); ref.build(0)
    want = []
    for i in range(N):
        want.append(ref.render(cam, tgt, p).copy())
        ref.animate_device(); ref.refit(False)
    ref.close()
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
    hip, ptrs = _alloc(W * H * 3, N)
    try:
        for i in range(N):
            k = i % 3
            if i >= 3:
                ctx.render_wait(k)
            ctx.render_submit(k, cam, tgt, p, ptrs[i])
            ctx.animate_device(); ctx.refit(False)
        for k in range(3):
            ctx.render_wait((N - 3 + k) % 3)
        got = _dev_frame(ctx, ptrs, W * H * 3)
        for i in range(N):
            assert np.array_equal(got[i].reshape(H, W, 3), want[i]), f"frame {i}"
    finally:
        for q in ptrs:
            hip.hipFree(C.c_void_p(q))
        ctx.close()


def test_asynchronous_rebuild_in_the_animation_loop():
    """main.cu:534-541 with crt_bvh_build_async / crt_bvh_swap: the rebuild of every 20th frame runs under the frames that
    follow; every frame equals the loop that stops for a synchronous device rebuild (the image does not depend on the tree)."""
    prims = np.concatenate([crt.scene_spheres(3000, seed=11), crt.scene_soup(20000, seed=12)])
    cam, tgt = crt.default_camera()
    W, H, N = 400, 225, 50
    p = crt.default_params(width=W, height=H, bounces=2)
    ref = crt.Context(0); ref.set_scene(prims); ref.set_lights(crt.default_lights()); ref.build(0)
    want = []
    for f in range(1, N + 1):
        ref.animate_device()
        if f % 20 == 0:
            ref.build(2)
        else:
            ref.refit(False)
        want.append(ref.render(cam, tgt, p).copy())
    ref.close()
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
    try:
        with pytest.raises(crt.CrtError):
            ctx.swap()                                   # nothing under way
        building, swaps, nodes0 = False, [], ctx.stats()["nodes"]
        for f in range(1, N + 1):
            ctx.animate_device()
            if f % 20 == 0:
                ctx.build_async(); building = True
                with pytest.raises(crt.CrtError):
                    ctx.build_async()                    # one at a time
            if building and ctx.swap(wait=(f % 20 == 7)):      # by the 7th frame after the snapshot at the latest
                building = False; swaps.append(f)
            else:
                ctx.refit(False)
            got = ctx.render(cam, tgt, p)
            assert np.array_equal(got, want[f - 1]), f"frame {f}"
        assert len(swaps) == 2 and all(20 <= s_ <= 27 or 40 <= s_ <= 47 for s_ in swaps), swaps
        assert ctx.stats()["nodes"] != nodes0            # the device-built tree is the current one
        ids = ctx.reference_order()                      # ranks of the snapshot, served from the device
        assert sorted(ids.tolist()) == list(range(len(prims)))
        ctx.build_async()
        ctx.set_scene(prims)                             # abandons the build under way
        ctx.build(0)
        assert np.array_equal(ctx.render(cam, tgt, p).shape, (H, W, 3))
    finally:
        ctx.close()
