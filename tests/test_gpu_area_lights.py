"""Area lights / soft shadows (Ray.cu:563-603 is commented out in the reference).

There is no reference behaviour to pin: the oracle restates the commented block with fixed sample draws
(oracle/crt_oracle.cpp shade()), and the HIP path must agree with it to the same bar as the live light types:
8-bit BGR within 1 LSB, shadow traversals counted equal.  "parity unpinned" in the sense of the task: both
sides implement the same stated definition, the reference itself never ran it.
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
LSB = 1


@pytest.fixture(scope="module")
def crt():
    import cudaraytracer_amd as m
    return m


def area_light(crt, pos, radius, color=(.9, .9, .8)):
    l = np.zeros(1, crt.LIGHT_DTYPE)[0]
    l["area"] = 1; l["position"] = pos; l["radius"] = radius; l["color"] = color
    return l


def check(g, o, what, max_bad):
    diff = np.abs(g.astype(np.int32) - o["bgr"].astype(np.int32)).max(axis=2)
    print(f"{what}: exact={int((diff == 0).sum())} off_by_1={int((diff == 1).sum())} beyond={int((diff > LSB).sum())}")
    assert int((diff > LSB).sum()) <= max_bad, what


@pytest.mark.parametrize("bvh_mode", [0, 1])
def test_area_light_matches_oracle(crt, bvh_mode):
    prims = crt.scene_teapot(with_sphere=True)
    cam, tgt = crt.default_camera()
    lights = np.zeros(3, crt.LIGHT_DTYPE)
    lights[0] = area_light(crt, (30, 160, 40), 45.0)
    lights[1] = crt.default_lights()[0]
    lights[1]["color"] = (.3, .3, .3)
    lights[2] = area_light(crt, (-120, 90, 120), 20.0, color=(.3, .4, .6))
    ctx = crt.Context(0)
    sc = orc.Scene(prims)
    try:
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(bvh_mode)
        sc.build()
        for samples, w, h, bounces in ((30, 240, 120, 2), (7, 333, 205, 3), (1, 160, 90, 1)):
            sc.stats(reset=True)
            g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, area_samples=samples))
            o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, area_samples=samples)
            check(g, o, f"area lights bvh={bvh_mode} samples={samples} {w}x{h} bounces={bounces}", max_bad=0)
            st = ctx.stats()
            live = sc.stats(reset=True)["live_traversals"]
            issued = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
            print("rays issued", issued, "oracle live traversals", live, "shadow", st["rays_shadow"])
            assert abs(issued - live) <= 8 * (2 * samples + 1)
        # the penumbra exists: some floor pixel is partly lit (a sample count strictly between none and all blocked)
        soft = ctx.render(cam, tgt, crt.default_params(width=240, height=120, bounces=1, area_samples=30))
        hard = ctx.render(cam, tgt, crt.default_params(width=240, height=120, bounces=1, area_samples=1))
        assert len(np.unique(soft[:60, :, 2])) > len(np.unique(hard[:60, :, 2]))
    finally:
        ctx.close(); sc.close()


def test_area_light_sharded_and_supersampled(crt):
    """tiles of two ranks put together equal the whole frame (the sample choice hashes the pixel, not the queue
    position), and spp > 1 runs (each sub-sample draws its own light samples)."""
    prims = crt.scene_teapot(with_sphere=True)
    cam, tgt = crt.default_camera()
    lights = np.zeros(1, crt.LIGHT_DTYPE)
    lights[0] = area_light(crt, (30, 160, 40), 45.0)
    ctx = crt.Context(0)
    try:
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
        whole = ctx.render(cam, tgt, crt.default_params(width=200, height=130, bounces=2, area_samples=9)).copy()
        parts = np.zeros_like(whole)
        for r in range(2):
            p = crt.default_params(width=200, height=130, bounces=2, area_samples=9, rank=r, nranks=2)
            img = ctx.render(cam, tgt, p)
            tiles, _ = crt.tiles_owned(p)
            for t in tiles:
                tx, ty = t % 7, t // 7
                parts[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32] = img[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32]
        assert (parts == whole).all()
        ss = ctx.render(cam, tgt, crt.default_params(width=200, height=130, bounces=2, area_samples=9, spp=4))
        assert np.abs(ss.astype(int) - whole.astype(int)).mean() < 6
    finally:
        ctx.close()


def test_area_samples_out_of_range(crt):
    prims = crt.scene_teapot(with_sphere=True)
    cam, tgt = crt.default_camera()
    lights = np.zeros(1, crt.LIGHT_DTYPE)
    lights[0] = area_light(crt, (0, 150, 0), 10.0)
    ctx = crt.Context(0)
    try:
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
        with pytest.raises(crt.CrtError):
            ctx.render(cam, tgt, crt.default_params(width=64, height=64, area_samples=65))
        ctx.render(cam, tgt, crt.default_params(width=64, height=64, area_samples=0))    # 0 = the reference's 30
        assert ctx.stats()["rays_shadow"] > 30 * 64 * 64 // 2
    finally:
        ctx.close()
