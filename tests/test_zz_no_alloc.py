"""crt_trace_noalloc (include/crt.h): the nearest-hit hook for callers that cannot allocate.  Its working buffers
live in a fixed per-GPU device array instead of per-call hipMalloc; results must equal crt_trace's bit for bit
(ids exact, t bit-identical: same kernels), on two calls in a row, the second with a smaller input, on both tree
formats.  Collected last (file name): the round's end runs `pytest -x`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.bufs = []

    def malloc(self, n):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), n) == 0
        self.bufs.append(p)
        return p

    def upload(self, p, a):
        a = np.ascontiguousarray(a)
        assert self.lib.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0

    def download(self, p, dtype, n):
        out = np.zeros(n, dtype)
        assert self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0
        return out

    def fill(self, p, byte, n):
        assert self.lib.hipMemset(p, byte, n) == 0

    def free_all(self):
        for p in self.bufs:
            self.lib.hipFree(p)
        self.bufs = []


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-150, 150, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d.astype(np.float32)], axis=1).astype(np.float32)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_noalloc_equals_crt_trace_twice_second_smaller(mode):
    import cudaraytracer_amd as crt
    hip = Hip()
    ctx = crt.Context(0)
    try:
        prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_soup(20000, seed=3)])
        ctx.set_scene(prims)
        ctx.build(mode)
        nmax = 300000
        d_rays = hip.malloc(nmax * 24)
        d_ids = hip.malloc(nmax * 4)
        d_ts = hip.malloc(nmax * 4)
        for n, seed in ((300000, 11), (70001, 12)):          # the second call is smaller and finds the first one's scratch
            rays = random_rays(n, seed)
            want_ids, want_ts = ctx.trace(rays)
            hip.upload(d_rays, rays)
            hip.fill(d_ids, 0x5a, nmax * 4)
            hip.fill(d_ts, 0x5a, nmax * 4)
            ctx.trace_noalloc(d_rays, n, d_ids, d_ts)
            got_ids = hip.download(d_ids, np.int32, nmax)
            got_ts = hip.download(d_ts, np.float32, nmax)
            assert (want_ids >= 0).sum() > 1000                  # the comparison sees real hits
            assert np.array_equal(got_ids[:n], want_ids)
            assert np.array_equal(got_ts[:n].view(np.uint32), want_ts.view(np.uint32))
            assert np.all(got_ids[n:] == 0x5a5a5a5a)             # nothing written past n
            assert np.all(got_ts[n:].view(np.uint32) == 0x5a5a5a5a)
    finally:
        hip.free_all()
        ctx.close()


def test_noalloc_rejects_oversize_and_bad_state():
    import cudaraytracer_amd as crt
    hip = Hip()
    ctx = crt.Context(0)
    try:
        d = hip.malloc(1024)
        with pytest.raises(crt.CrtError) as e:                   # no BVH yet
            ctx.trace_noalloc(d, 4, d, d)
        assert e.value.code == -2
        ctx.set_scene(crt.scene_teapot())
        ctx.build(0)
        with pytest.raises(crt.CrtError) as e:                   # beyond the fixed scratch array: rejected before any launch
            ctx.trace_noalloc(d, (1 << 20) + 1, d, d)
        assert e.value.code == -1 and "CRT_NOALLOC_MAX_RAYS" in str(e.value)
        with pytest.raises(crt.CrtError):
            ctx.trace_noalloc(None, 4, d, d)
        ctx.trace_noalloc(None, 0, None, None)                   # empty batch: nothing to do
    finally:
        hip.free_all()
        ctx.close()


def test_noalloc_at_the_maximum_batch():
    import cudaraytracer_amd as crt
    hip = Hip()
    ctx = crt.Context(0)
    try:
        ctx.set_scene(crt.scene_teapot(with_sphere=True))
        ctx.build(0)
        n = 1 << 20
        rays = random_rays(n, 5)
        want_ids, want_ts = ctx.trace(rays)
        d_rays, d_ids, d_ts = hip.malloc(n * 24), hip.malloc(n * 4), hip.malloc(n * 4)
        hip.upload(d_rays, rays)
        ctx.trace_noalloc(d_rays, n, d_ids, d_ts)
        assert np.array_equal(hip.download(d_ids, np.int32, n), want_ids)
        assert np.array_equal(hip.download(d_ts, np.float32, n).view(np.uint32), want_ts.view(np.uint32))
    finally:
        hip.free_all()
        ctx.close()
