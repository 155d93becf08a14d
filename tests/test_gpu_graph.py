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
