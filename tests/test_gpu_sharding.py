"""Tile sharding and scene replication on the device (one physical GPU): R logical ranks, each a crt
context of its own, render their tiles; tile-major pack -> unpack into rank 0's frame must rebuild
the single-rank frame byte for byte.  The scene reaches the other contexts only as the packed blob
(crt_blob_device_ptr -> crt_blob_import_device), as it would after an RCCL broadcast."""
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
        assert self.lib.hipMemset(p, 0, n) == 0
        self.bufs.append(p)
        return p

    def to_host(self, p, n):
        out = np.zeros(n, np.uint8)
        assert self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, n, 2) == 0
        return out

    def free_all(self):
        for p in self.bufs:
            self.lib.hipFree(p)
        self.bufs = []


@pytest.mark.parametrize("world,w,h", [(2, 720, 360), (3, 333, 205), (8, 1920, 1080)])
def test_logical_ranks_rebuild_the_frame(world, w, h):
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import parallel
    hip = Hip()
    prims = crt.scene_teapot()
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    root = crt.Context(0)
    root.set_scene(prims); root.set_lights(lights); root.build(0)
    full = root.render(cam, tgt, crt.default_params(width=w, height=h, bounces=2))
    blob_ptr, blob_bytes = root.blob()
    frame0 = hip.malloc(w * h * 3)
    ctxs = []
    try:
        for r in range(world):
            p = crt.default_params(width=w, height=h, bounces=2, rank=r, nranks=world)
            if r == 0:
                ctx = root
            else:
                ctx = crt.Context(0)
                ctx.import_blob(blob_ptr, blob_bytes)       # scene arrives as the packed blob only
                ctx.set_lights(lights)
                ctxs.append(ctx)
            frame = frame0 if r == 0 else hip.malloc(w * h * 3)
            ctx.render_device(cam, tgt, p, frame)
            if r > 0:
                nb = parallel.packed_nbytes(p, r)
                packed = hip.malloc(nb)
                ctx.tiles_pack(p, frame, packed)
                # the device packing equals the host restatement used by the gloo path
                own = hip.to_host(frame, w * h * 3).reshape(h, w, 3)
                assert np.array_equal(hip.to_host(packed, nb), parallel.pack_tiles_host(p, r, own))
                root.tiles_unpack(p, r, packed, frame0)
        root.sync()                                         # unpacks queue on the root context's stream
        got = hip.to_host(frame0, w * h * 3).reshape(h, w, 3)
        assert np.array_equal(got, full)
        # rays issued by the ranks add up to the single-rank count
    finally:
        for c in ctxs:
            c.close()
        root.close()
        hip.free_all()


def test_sorted_rays_change_nothing_but_the_order():
    """crt_params.sort_rays only permutes the order in which the traversal kernel visits
    the secondary rays: frame and ray counts are identical."""
    import cudaraytracer_amd as crt
    prims = np.concatenate([crt.scene_soup(40000, seed=2), crt.scene_spheres(200, seed=2)])
    lights = np.zeros(3, crt.LIGHT_DTYPE)
    lights[:2] = crt.default_lights()
    lights[2]["point"] = 1; lights[2]["position"] = (4, 3, -4); lights[2]["color"] = (.5, .5, .5)
    lights[2]["constantTerm"] = 0.25; lights[2]["linearTerm"] = 0.003372407; lights[2]["quadraticTerm"] = 0.000045492
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
    cam, tgt = crt.default_camera()
    try:
        a = ctx.render(cam, tgt, crt.default_params(width=500, height=300, bounces=4, sort_rays=0))
        sa = ctx.stats()
        b = ctx.render(cam, tgt, crt.default_params(width=500, height=300, bounces=4, sort_rays=1))
        sb = ctx.stats()
        assert np.array_equal(a, b)
        assert all(sa[k] == sb[k] for k in ("rays_primary", "rays_shadow", "rays_reflect"))
    finally:
        ctx.close()


def test_spp_extension_is_inert_at_1_and_smooth_at_4():
    import cudaraytracer_amd as crt
    prims = crt.scene_teapot()
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
    cam, tgt = crt.default_camera()
    try:
        a = ctx.render(cam, tgt, crt.default_params(width=360, height=180, bounces=2, spp=1))
        b = ctx.render(cam, tgt, crt.default_params(width=360, height=180, bounces=2, spp=4))
        st = ctx.stats()
        assert st["rays_primary"] == 4 * 360 * 180
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        assert d.mean() < 3.0 and (d.max(axis=2) > 0).mean() < 0.25      # only edges move
        with pytest.raises(crt.CrtError):
            ctx.render(cam, tgt, crt.default_params(width=64, height=64, spp=3))
    finally:
        ctx.close()
