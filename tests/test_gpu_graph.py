"""crt_params.use_graph: the frame's launch sequence replayed as a HIP graph.  Frames must equal the plain path's, also
while the scene moves (animation + refit change the scene's contents, not the launches), and a change of camera,
parameters or lights must be picked up."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crt():
    import cudaraytracer_amd as m
    return m


def test_graph_replay_equals_plain_launches(crt):
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_spheres(64, seed=5)])
    cam, tgt = crt.default_camera()
    a = crt.Context(0); b = crt.Context(0)
    try:
        for c in (a, b):
            c.set_scene(prims); c.set_lights(crt.default_lights()); c.build(0)
        plain = crt.default_params(width=360, height=200, bounces=3)
        graph = crt.default_params(width=360, height=200, bounces=3, use_graph=1)
        for frame in range(8):
            if frame:
                for c in (a, b):
                    c.animate_device(); c.refit(False)
            want = a.render(cam, tgt, plain).copy()
            got = b.render(cam, tgt, graph)
            assert np.array_equal(want, got), f"frame {frame}"
            assert b.stats()["ms_total"] > 0 and b.stats()["rays_shadow"] == a.stats()["rays_shadow"]
        # a moved camera, another bounce count, another light set: the graph is captured again
        cam2 = cam + np.array([10, 5, -20], np.float32)
        assert np.array_equal(a.render(cam2, tgt, plain), b.render(cam2, tgt, graph))
        plain.bounces = graph.bounces = 2
        assert np.array_equal(a.render(cam2, tgt, plain), b.render(cam2, tgt, graph))
        one = crt.default_lights()[:1]
        a.set_lights(one); b.set_lights(one)
        assert np.array_equal(a.render(cam2, tgt, plain), b.render(cam2, tgt, graph))
        # a rebuild (new blob contents, possibly a new blob) and spp > 1
        for c in (a, b):
            c.build(2)
        plain.spp = graph.spp = 4
        assert np.array_equal(a.render(cam2, tgt, plain), b.render(cam2, tgt, graph))
    finally:
        a.close(); b.close()


def test_graph_survives_rebuilds_with_unchanged_parameters(crt):
    """Camera, parameters and lights stay the same while the tree under them is rebuilt (another scene of the same size,
    another builder, a deeper tree, back again): every buffer and header field a captured launch bakes in is part of
    the graph's key (d_ovf / ovf_depth were not: ADVICE r01), so each frame must equal the plain path's."""
    cam, tgt = crt.default_camera()
    n = len(crt.scene_teapot(with_sphere=True))
    deep = crt.scene_soup(n, seed=9)
    # geometric spacing along x makes every builder peel a few primitives off per level: a much deeper tree, same count
    for i in range(n):
        s = np.float32(1.09 ** (i % 512)) * np.float32(1e-3)
        for k in ("v1", "v2", "v3"):
            deep[k][i] = deep[k][i] * np.float32(0.01) + np.array([s, 0.25 * (i // 512), 0], np.float32)
    scenes = [crt.scene_teapot(with_sphere=True), deep, crt.scene_soup(n, seed=4), crt.scene_teapot(with_sphere=True)]
    a = crt.Context(0); b = crt.Context(0)
    try:
        plain = crt.default_params(width=320, height=160, bounces=2)
        graph = crt.default_params(width=320, height=160, bounces=2, use_graph=1)
        depths = []
        for mode in (0, 1, 2):
            for prims in scenes:
                for c in (a, b):
                    c.set_scene(prims); c.set_lights(crt.default_lights()); c.build(mode)
                want = a.render(cam, tgt, plain).copy()
                for _ in range(2):                      # the capture, then a replay
                    assert np.array_equal(want, b.render(cam, tgt, graph)), (mode, len(depths))
                depths.append(b.stats()["depth"])
        assert max(depths) > min(depths) + 3            # the rebuilds really changed the tree's depth
    finally:
        a.close(); b.close()
