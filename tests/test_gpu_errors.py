"""Error behaviour of the C ABI on a device: every misuse returns a negative crt_status with a message, nothing
crashes, and the context stays usable afterwards (the reference ignores every CUDA error, Ray.cu:670-740)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crt():
    import cudaraytracer_amd as m
    return m


def code_of(crt, fn, *a, **k):
    with pytest.raises(crt.CrtError) as e:
        fn(*a, **k)
    assert str(e.value)                      # a message came with it
    return e.value.code


def test_call_order_and_argument_errors(crt):
    cam, tgt = crt.default_camera()
    prims = crt.scene_teapot(with_sphere=True)
    ctx = crt.Context(0)
    try:
        p = crt.default_params(width=64, height=64)
        assert code_of(crt, ctx.render, cam, tgt, p) == -2                  # CRT_ERR_STATE: no BVH yet
        assert code_of(crt, ctx.trace, np.zeros((4, 6), np.float32)) == -2
        assert code_of(crt, ctx.refit, False) == -2
        ctx.set_scene(prims)
        assert code_of(crt, ctx.render, cam, tgt, p) == -2                  # scene without a build
        assert code_of(crt, ctx.build, 7) == -1                             # CRT_ERR_ARG: unknown mode
        ctx.build(0)
        for bad in (dict(bounces=0), dict(bounces=17), dict(spp=3), dict(spp=81), dict(tile_w=12), dict(rank=2, nranks=2),
                    dict(width=0)):
            assert code_of(crt, ctx.render, cam, tgt, crt.default_params(**dict(dict(width=64, height=64), **bad))) == -1, bad
        assert code_of(crt, ctx.update_scene, prims[:-1]) == -1             # count differs
        swapped = prims.copy(); swapped["type"][0] = 0
        assert code_of(crt, ctx.update_scene, swapped) == -1                # a primitive changed its type
        assert code_of(crt, ctx.capture, 0, 64 * 64) == -2                  # capture was not enabled
        lights9 = np.zeros(9, crt.LIGHT_DTYPE)
        assert code_of(crt, ctx.set_lights, lights9) == -1                  # more than CRT_MAX_LIGHTS
        assert code_of(crt, ctx.import_blob, ctx.blob()[0], 100) == -1      # not a blob of that size
        # ... and the context still renders
        ctx.set_lights(crt.default_lights())
        img = ctx.render(cam, tgt, p)
        assert img.shape == (64, 64, 3) and img.any()
        # zero rays is not an error
        ids, ts = ctx.trace(np.zeros((0, 6), np.float32))
        assert len(ids) == 0
    finally:
        ctx.close()


def test_no_such_device(crt):
    err = C.c_int(0)
    h = crt.lib.crt_create(64, C.byref(err))
    assert not h and err.value < 0
