"""Randomised whole-frame parity: scenes, lights, cameras, frame sizes, bounce counts and tree builders drawn from a
seed, the HIP path (through the C ABI) against the oracle's frame on the same inputs.

The fixed scenes of test_gpu_parity.py pin the cases BASELINE names; this file looks for what none of them
happens to contain: spheres and triangles sharing leaves, materials from black to mirror, shininess from 0 up, point
lights inside the geometry, cameras inside the room looking anywhere, frames that are not multiples of the tile.
Bar as in test_gpu_parity.py: every pixel within 1 LSB (in practice equal), per-bounce hit ids equal.
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crt():
    import cudaraytracer_amd as m
    return m


def random_scene(crt, rng):
    ntri = int(rng.integers(0, 2500))
    nsph = int(rng.integers(0 if ntri else 1, 80))
    tri = crt.scene_soup(ntri, seed=int(rng.integers(1, 1 << 30))) if ntri else np.zeros(0, crt.PRIM_DTYPE)
    if ntri:
        # the soup fills 360 x 360 x 480: pull it into a random box inside the room, stretch the triangles
        centre = rng.uniform(-60, 60, 3).astype(np.float32)
        shrink = np.float32(rng.uniform(0.1, 0.5))
        grow = np.float32(rng.uniform(1.0, 12.0))
        c = ((tri["v1"] + tri["v2"] + tri["v3"]) / np.float32(3)).astype(np.float32)
        for f in ("v1", "v2", "v3"):
            tri[f] = (centre + c * shrink + (tri[f] - c) * grow).astype(np.float32)
    sph = crt.scene_spheres(nsph, seed=int(rng.integers(1, 1 << 30))) if nsph else np.zeros(0, crt.PRIM_DTYPE)
    if nsph:
        sph["position"] = rng.uniform(-120, 120, (nsph, 3)).astype(np.float32)
        sph["radius"] = rng.uniform(1.0, 25.0, nsph).astype(np.float32)
    prims = np.concatenate([tri, sph])
    n = len(prims)
    # materials: a handful of random ones, dealt out at random (k_shade gathers them by index)
    nm = int(rng.integers(1, 7))
    amb = rng.uniform(0, 1, (nm, 3)); dif = rng.uniform(0, 1, (nm, 3)); spe = rng.uniform(0, 1, (nm, 3))
    refl = rng.uniform(0, 1, (nm, 3)) * (rng.uniform(size=(nm, 1)) < 0.7)          # some materials reflect nothing
    shin = rng.choice([0.0, 1.0, 2.0, 7.5, 20.0, 128.0], nm)
    pick = rng.integers(0, nm, n)
    prims["ambient"] = amb[pick].astype(np.float32); prims["diffuse"] = dif[pick].astype(np.float32)
    prims["specular"] = spe[pick].astype(np.float32); prims["reflective"] = refl[pick].astype(np.float32)
    prims["shininess"] = shin[pick].astype(np.float32)
    prims["objNum"] = np.arange(n, dtype=np.int32)
    order = rng.permutation(n)                # spheres and triangles interleaved in scene order (the tie-break order)
    return np.ascontiguousarray(prims[order])


def random_lights(crt, rng):
    nl = int(rng.integers(0, 5))
    lights = np.zeros(nl, crt.LIGHT_DTYPE)
    for k in range(nl):
        lights[k]["color"] = rng.uniform(.05, .6, 3)
        lights[k]["direction"] = rng.normal(size=3)
        u = rng.uniform()
        if u < 0.15:          # an area light (the reference's commented-out branch, Ray.cu:563-603, as SURVEY 8(f) N3 defines it)
            lights[k]["area"] = 1
            lights[k]["position"] = rng.uniform(-150, 150, 3)
            lights[k]["radius"] = rng.uniform(1, 40)
        elif u < 0.5:
            lights[k]["point"] = 1
            lights[k]["position"] = rng.uniform(-150, 150, 3)
            lights[k]["constantTerm"] = rng.uniform(0.1, 1.0)
            lights[k]["linearTerm"] = rng.uniform(0, 0.01)
            lights[k]["quadraticTerm"] = rng.uniform(0, 0.0001)
    return lights


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_scene_frame_equals_the_oracle(crt, seed):
    rng = np.random.default_rng(1000 + seed)
    prims = random_scene(crt, rng)
    lights = random_lights(crt, rng)
    w, h = int(rng.integers(33, 200)), int(rng.integers(17, 130))
    bounces = int(rng.integers(1, 5))
    spp = int(rng.choice([1, 1, 1, 4, 9]))
    area_samples = int(rng.integers(1, 9))
    cam = rng.uniform(-170, 170, 3).astype(np.float32)
    cam[2] = np.float32(rng.uniform(-250, 280))
    tgt = rng.uniform(-60, 60, 3).astype(np.float32)
    sc = orc.Scene(prims)
    sc.build()
    try:
        sc.stats(reset=True)
        o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, want_color=True, want_ids=True, area_samples=area_samples, spp=spp)
        ost = sc.stats(reset=True)
    finally:
        sc.close()
    npix = w * h
    # a fractional shininess makes pow(negative, s) = NaN possible: such scenes must trace zero-coefficient paths on, as the
    # reference does (0 * NaN turns the pixel NaN, Ray.cu:660); the others may end them (crt_params.zero_paths)
    fractional = bool((prims["shininess"] != np.floor(prims["shininess"])).any())
    for mode in (0, 1, 2):
        ctx = crt.Context(0)
        try:
            ctx.set_scene(prims); ctx.set_lights(lights)
            try:
                ctx.build(mode)
            except crt.CrtError:
                assert mode == 1          # the reference's builder has scenes it cannot split (tests/test_host_layer.py)
                continue
            ctx.set_capture(True)
            g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, spp=spp, area_samples=area_samples))
            what = f"seed {seed} mode {mode}: {len(prims)} prims, {len(lights)} lights, {w}x{h}, bounces={bounces}, spp={spp}"
            for b in range(bounces if spp == 1 else 0):
                ids_g, ts_g = ctx.capture(b, npix)
                traced = ids_g != -9            # the GPU ends paths whose coefficient is zero
                bad = int((ids_g[traced] != o["ids"][b][traced]).sum())
                assert bad == 0, f"{what}: bounce {b}: {bad} hit ids differ"
                hit = traced & (o["ids"][b] >= 0)
                rel = np.abs(ts_g[hit] - o["ts"][b][hit]) / np.maximum(np.abs(o["ts"][b][hit]), 1e-30)
                assert not (rel > 1e-4).any(), f"{what}: bounce {b}: {int((rel > 1e-4).sum())} hit distances beyond 1e-4 relative"
            diff = np.abs(g.astype(np.int32) - o["bgr"].astype(np.int32)).max(axis=2)
            st = ctx.stats()
            issued = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
            print(f"{what}: exact={int((diff == 0).sum())}/{npix} off_by_1={int((diff == 1).sum())} rays={issued} "
                  f"oracle traversals={ost['traversals']} live={ost['live_traversals']} fractional_shininess={fractional}")
            assert int((diff > 1).sum()) == 0, f"{what}: {int((diff > 1).sum())} pixels beyond 1 LSB"
            # every traversal the reference's logic issues, one for one: all of them when zero-coefficient paths are traced on
            assert issued == (ost["traversals"] if fractional else ost["live_traversals"]), what
        finally:
            ctx.close()


def hostile_scene(crt, rng):
    """inputs the reference accepts and shades to NaN or infinity: zero-radius spheres, zero and opposed vertex normals,
    negative and fractional shininess, infinite material values, a light without a direction"""
    prims = random_scene(crt, rng)
    n = len(prims)
    tri = np.nonzero(prims["type"] == 1)[0]
    sph = np.nonzero(prims["type"] == 0)[0]
    kind = int(rng.integers(0, 6))
    if kind == 0 and len(sph):
        prims["radius"][sph[: max(1, len(sph) // 3)]] = 0.0
    elif kind == 1 and len(tri):
        k = tri[: max(1, len(tri) // 4)]
        prims["v1Norm"][k] = 0; prims["v2Norm"][k] = 0; prims["v3Norm"][k] = 0
    elif kind == 2 and len(tri):
        k = tri[: max(1, len(tri) // 4)]
        prims["v2Norm"][k] = -prims["v1Norm"][k]; prims["v3Norm"][k] = 0          # the mix passes through the zero vector
    elif kind == 3:
        prims["shininess"][: max(1, n // 3)] = np.float32(-2.0)
    elif kind == 4:
        prims["specular"][: max(1, n // 5)] = np.float32(np.inf)
    else:
        prims["shininess"][: max(1, n // 2)] = np.float32(0.5)
    return prims, kind


@pytest.mark.parametrize("seed", list(range(12)))
def test_scenes_that_shade_to_nan_equal_the_oracle(crt, seed):
    """NaN and infinity are reference behaviour too (Ray.cu:655-660 adds them like any colour; the 8-bit conversion of Ray.cu:724-731
    then decides the pixel): same frame, same rays."""
    rng = np.random.default_rng(5000 + seed)
    prims, kind = hostile_scene(crt, rng)
    lights = random_lights(crt, rng)
    if seed % 4 == 3 and len(lights):
        lights[0]["direction"] = 0; lights[0]["point"] = 0; lights[0]["area"] = 0          # normalize(0), Ray.cu:635
    w, h, bounces = int(rng.integers(40, 120)), int(rng.integers(30, 90)), int(rng.integers(2, 5))
    cam = rng.uniform(-150, 150, 3).astype(np.float32)
    tgt = rng.uniform(-40, 40, 3).astype(np.float32)
    sc = orc.Scene(prims)
    sc.build()
    try:
        sc.stats(reset=True)
        o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, want_color=True, want_ids=True)
        ost = sc.stats(reset=True)
    finally:
        sc.close()
    for mode in (0, 2):
        ctx = crt.Context(0)
        try:
            ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(mode)
            g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
            st = ctx.stats()
        finally:
            ctx.close()
        diff = np.abs(g.astype(np.int32) - o["bgr"].astype(np.int32)).max(axis=2)
        nonfinite = int((~np.isfinite(o["color"])).any(axis=1).sum())
        issued = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
        print(f"seed {seed} kind {kind} mode {mode}: {w}x{h} bounces={bounces} non-finite pixels (oracle)={nonfinite} "
              f"exact={int((diff == 0).sum())}/{w * h} rays={issued} oracle={ost['traversals']}")
        assert int((diff > 1).sum()) == 0, f"seed {seed} kind {kind} mode {mode}: {int((diff > 1).sum())} pixels beyond 1 LSB"
        assert issued == ost["traversals"]


def test_zero_paths_parameter(crt):
    """crt_params.zero_paths on the reference's own scene (integral shininess: ending zero-coefficient paths is exact): the image
    does not depend on it, the ray count does -- 1 issues every traversal the reference's loop issues (Ray.cu:715-723), 0 and 2
    only those on paths that can still add light."""
    prims = crt.scene_teapot(with_sphere=True)
    lights = crt.default_lights(); cam, tgt = crt.default_camera()
    w, h, bounces = 240, 120, 4
    sc = orc.Scene(prims); sc.build()
    try:
        sc.stats(reset=True)
        o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16)
        ost = sc.stats(reset=True)
    finally:
        sc.close()
    assert ost["live_traversals"] < ost["traversals"]
    ctx = crt.Context(0)
    try:
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
        frames, rays = {}, {}
        for z in (0, 1, 2):
            frames[z] = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, zero_paths=z))
            st = ctx.stats()
            rays[z] = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
        with pytest.raises(crt.CrtError):
            ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, zero_paths=3))
    finally:
        ctx.close()
    for z in (0, 1, 2):
        assert np.array_equal(frames[z], o["bgr"]), f"zero_paths={z}"
    assert rays[1] == ost["traversals"] and rays[0] == rays[2] == ost["live_traversals"], (rays, ost)
