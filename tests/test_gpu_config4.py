"""BASELINE.json configs[4]: the 1M-triangle soup at 7680x4320 x 16 spp, image tiles sharded 8 ways.

One physical GPU, so the 8 ranks are logical (crt_params.rank / nranks): the per-rank render, the tile-major pack and
unpack are exactly what 8 processes run; the RCCL gather between pack and unpack moves bytes and is rehearsed
elsewhere (tests/test_gpu_sharded_renderer.py).  Three levels:
  * spp > 1 against the oracle's restatement of the 4x4 stratified offsets: byte-exact;
  * the whole configuration at a size the oracle finishes in seconds (30k soup, 480x270, 8 ranks, spp = 16): the frame
    assembled from the 8 ranks' packed tiles equals the oracle's frame byte for byte;
  * the full size on the device: two of the 8 ranks render their share, every owned pixel is written and nothing else,
    the ray accounting adds up, the tiles of the 8 ranks are disjoint and cover the frame, and the two ranks' tiles equal
    the same pixels of the frame one context renders alone.
"""
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

    def malloc(self, n, fill=0):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), n) == 0
        assert self.lib.hipMemset(p, fill, n) == 0
        self.bufs.append(p)
        return p

    def fill(self, p, byte, n):
        assert self.lib.hipMemset(p, byte, n) == 0

    def to_host(self, p, n):
        out = np.zeros(n, np.uint8)
        assert self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, n, 2) == 0
        return out

    def free_all(self):
        for p in self.bufs:
            self.lib.hipFree(p)
        self.bufs = []


@pytest.mark.parametrize("spp,bounces", [(4, 2), (16, 3)])
def test_spp_frames_equal_the_oracle(spp, bounces):
    import cudaraytracer_amd as crt
    import orc
    prims = crt.scene_teapot(with_sphere=True)
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    w, h = 240, 120
    sc = orc.Scene(prims); sc.build()
    want = sc.render(lights, cam, tgt, w, h, bounces, nthreads=8, spp=spp)["bgr"]
    sc.close()
    for mode in (0, 1):
        ctx = crt.Context(0)
        try:
            ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(mode)
            got = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, spp=spp))
            st = ctx.stats()
        finally:
            ctx.close()
        assert st["rays_primary"] == spp * w * h
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert diff.max() == 0, f"mode {mode}: {int((diff.max(axis=2) > 0).sum())} pixels differ, max {diff.max()}"


def test_config4_at_oracle_size_8_ranks_16_spp():
    import cudaraytracer_amd as crt
    import orc
    hip = Hip()
    prims = crt.scene_soup(30000, seed=1)
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    w, h, spp, bounces, world = 480, 270, 16, 3, 8
    sc = orc.Scene(prims); sc.build()
    want = sc.render(lights, cam, tgt, w, h, bounces, nthreads=8, spp=spp)["bgr"]
    sc.close()
    root = crt.Context(0)
    ranks = []
    try:
        root.set_scene(prims); root.set_lights(lights); root.build(0)
        blob_ptr, blob_bytes = root.blob()
        frame0 = hip.malloc(w * h * 3, 0x5a)
        total_primary = 0
        for r in range(world):
            ctx = root if r == 0 else crt.Context(0)
            ranks.append(ctx)
            if r:
                ctx.import_blob(blob_ptr, blob_bytes)          # the scene arrives as the broadcast blob
                ctx.set_lights(lights)
            p = crt.default_params(width=w, height=h, bounces=bounces, spp=spp, rank=r, nranks=world)
            ids, npx = crt.tiles_owned(p)
            mine = hip.malloc(w * h * 3, 0)
            packed = hip.malloc(len(ids) * 32 * 32 * 3, 0)
            ctx.render_device(cam, tgt, p, mine)
            total_primary += ctx.stats()["rays_primary"]
            assert ctx.stats()["rays_primary"] == npx * spp
            ctx.tiles_pack(p, mine, packed)
            root.tiles_unpack(p, r, packed, frame0)            # what rank 0 does with every gathered buffer
        root.sync()
        got = hip.to_host(frame0, w * h * 3).reshape(h, w, 3)
        assert total_primary == w * h * spp
        assert np.array_equal(got, want), f"{int((got != want).any(axis=2).sum())} pixels differ from the oracle's spp=16 frame"
    finally:
        for c in ranks:
            c.close()
        hip.free_all()


def test_config4_full_size_two_of_eight_ranks():
    import cudaraytracer_amd as crt
    hip = Hip()
    w, h, spp, bounces, world = 7680, 4320, 16, 3, 8
    lights = crt.default_lights()
    cam, tgt = crt.default_camera()
    # the tile map first (host only): the 8 ranks' tiles are disjoint and cover the frame
    seen = np.zeros(((h + 31) // 32) * ((w + 31) // 32), np.int32)
    pixels = 0
    for r in range(world):
        ids, npx = crt.tiles_owned(crt.default_params(width=w, height=h, spp=spp, rank=r, nranks=world))
        seen[ids] += 1
        pixels += npx
    assert np.all(seen == 1) and pixels == w * h
    ctx = crt.Context(0)
    try:
        ctx.set_scene(crt.scene_soup(1000000, seed=1)); ctx.set_lights(lights); ctx.build(2)     # device build: seconds matter here
        full = hip.malloc(w * h * 3, 0)
        ctx.render_device(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, spp=spp), full)
        st_full = ctx.stats()
        assert st_full["rays_primary"] == w * h * spp
        ref = hip.to_host(full, w * h * 3).reshape(h, w, 3)
        part = hip.malloc(w * h * 3, 0)
        rays = {}
        for r in (0, 5):
            p = crt.default_params(width=w, height=h, bounces=bounces, spp=spp, rank=r, nranks=world)
            ids, npx = crt.tiles_owned(p)
            hip.fill(part, 0x5a, w * h * 3)
            ctx.render_device(cam, tgt, p, part)
            st = ctx.stats()
            rays[r] = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
            assert st["rays_primary"] == npx * spp
            got = hip.to_host(part, w * h * 3).reshape(h, w, 3)
            owned = np.zeros((h, w), bool)
            tx = (w + 31) // 32
            for t in ids:
                y0, x0 = (t // tx) * 32, (t % tx) * 32
                owned[y0:y0 + 32, x0:x0 + 32] = True
            assert owned.sum() == npx
            assert np.all(got[~owned] == 0x5a)                                 # nothing outside this rank's tiles is touched
            assert np.array_equal(got[owned], ref[owned])                      # and its tiles are the single-context frame's
        total = st_full["rays_primary"] + st_full["rays_shadow"] + st_full["rays_reflect"]
        for r in rays:                                                         # a spatially uniform share: within 5 % of one eighth
            assert abs(rays[r] * world / total - 1.0) < 0.05, (r, rays[r], total)
    finally:
        ctx.close()
        hip.free_all()
