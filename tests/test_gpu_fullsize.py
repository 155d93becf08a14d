"""BASELINE.json's full sizes (1M-triangle soup, 3840x2160, bounces=3) through size-independent
properties plus an oracle sample:
  - the kernel on the wide SAH tree and on the reference's own topology (two unrelated trees, exact
    vs quantised nodes) must return identical ids and t for the same rays;
  - a sample of those rays is checked against the CPU oracle (its traversal tests ~1e5 triangles per
    ray on this scene, so the sample is small);
  - the frame renders, every pixel is written, and the ray accounting is consistent.
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def soup():
    import cudaraytracer_amd as crt
    prims = crt.scene_soup(1000000, seed=1)
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
    yield crt, prims, ctx
    ctx.close()


def test_two_topologies_agree_on_4k_rays_and_oracle_sample(soup):
    crt, prims, ctx = soup
    cam, tgt = crt.default_camera()
    rays = orc.primary_rays(cam, tgt, 3840, 2160)
    rng = np.random.default_rng(3)
    pick = rng.choice(len(rays), 400000, replace=False)
    sub = rays[pick]
    # secondary-like rays: random origins inside the soup, random directions
    o = rng.uniform([-170, -170, -270], [170, 170, 190], (200000, 3)).astype(np.float32)
    d = rng.normal(size=(200000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sub = np.concatenate([sub, np.concatenate([o, d], axis=1)])
    ids_w, ts_w = ctx.trace(sub)
    ref = crt.Context(0)
    ref.set_scene(prims); ref.build(1)
    try:
        ids_r, ts_r = ref.trace(sub)
    finally:
        ref.close()
    assert (ids_w >= 0).mean() > 0.9
    assert np.array_equal(ids_w, ids_r) and np.array_equal(ts_w, ts_r)
    sc = orc.Scene(prims)
    sc.build()
    sample = np.concatenate([np.arange(0, 400000, 4000), 400000 + np.arange(0, 200000, 2000)])
    ids_o, ts_o = sc.trace(sub[sample], nthreads=16)
    sc.close()
    assert np.array_equal(ids_w[sample], ids_o)
    rel = np.abs(ts_w[sample] - ts_o) / np.maximum(np.abs(ts_o), 1e-30)
    assert np.all(rel[ids_o >= 0] <= 1e-4)


def test_full_frame_accounting(soup):
    crt, prims, ctx = soup
    cam, tgt = crt.default_camera()
    p = crt.default_params(width=3840, height=2160, bounces=3)
    img = ctx.render(cam, tgt, p)
    st = ctx.stats()
    assert st["rays_primary"] == 3840 * 2160
    assert 0 < st["rays_reflect"] <= 2 * st["rays_primary"]
    assert 0 < st["rays_shadow"] <= 2 * (st["rays_primary"] + st["rays_reflect"])
    assert img.reshape(-1, 3).max(axis=1).min() > 0        # every pixel has ambient + something
    # rendering the same frame twice is bit-identical (queue order differs run to run, results must not)
    assert np.array_equal(img, ctx.render(cam, tgt, p))
