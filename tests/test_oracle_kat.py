"""Known-answer tests of the oracle's restated intersect functions (Ray.cu:54-457) on hand-made and
degenerate inputs (SURVEY.md 8(c) fixtures (i)), and BVH-vs-brute-force agreement."""
import numpy as np
import pytest

import orc


def tri(v1, v2, v3, n=(0, 0, 1)):
    p = np.zeros(1, orc.PRIM_DTYPE)[0]
    p["type"] = 1
    p["v1"], p["v2"], p["v3"] = v1, v2, v3
    p["v1Norm"] = p["v2Norm"] = p["v3Norm"] = n
    return p


def sph(c, r):
    p = np.zeros(1, orc.PRIM_DTYPE)[0]
    p["type"] = 0
    p["position"], p["radius"] = c, r
    return p


def test_bbox_slab_semantics():
    box = [-1, 1, -1, 1, -1, 1]
    assert orc.bbox(box, [0, 0, 5], [0, 0, -1]) == 1
    assert orc.bbox(box, [0, 0, 5], [0, 0, 1]) == 1          # box entirely behind the origin still "hits" (Ray.cu:103)
    assert orc.bbox(box, [3, 0, 5], [0, 0, -1]) == 0          # d.x == 0 and origin outside the slab: +-inf ordering rejects
    assert orc.bbox(box, [0.5, 0.5, 5], [0, 0, -1]) == 1      # d.x == d.y == 0 inside the slabs
    assert orc.bbox(box, [5, 5, 5], [-1, -1, -1]) == 1
    assert orc.bbox(box, [5, 5, 5], [-1, -1, 0.5]) == 0
    # origin exactly on a slab plane with a zero direction component: 0/0 = NaN compares false everywhere
    assert orc.bbox(box, [1, 0, 5], [0, 0, -1]) == 1


def test_triangle_known_answers():
    t = tri([-2, 0, 0], [0, 2, 0], [2, 0, 0])                # the commented-out test triangle of main.cu:271-275
    hit, tt, n, x = orc.triangle(t, [0, 1, 5], [0, 0, -1])
    assert hit and tt == 5.0                                 # either winding hits: the edge tests use the same Ng
    t2 = tri([-2, 0, 0], [2, 0, 0], [0, 2, 0])
    hit, tt, n, x = orc.triangle(t2, [0, 1, 5], [0, 0, -1])
    assert hit and tt == 5.0 and np.allclose(x, [0, 1, 0]) and np.allclose(n, [0, 0, 1])
    hit, tt, _, _ = orc.triangle(t2, [0, 1, -5], [0, 0, 1])  # from behind: denominators change sign together
    assert hit and tt == 5.0
    assert not orc.triangle(t2, [0, 1, 5], [1, 0, 0])[0]     # parallel: |denom| <= 1e-4
    assert orc.triangle(t2, [0, 1, np.float32(-5e-5)], [0, 0, -1])[0]      # t = -5e-5 >= -1e-4 accepted
    assert not orc.triangle(t2, [0, 1, np.float32(-2e-4)], [0, 0, -1])[0]  # t = -2e-4 rejected
    assert orc.triangle(t2, [0, 0, 5], [0, 0, -1])[0]        # on an edge: inclusive tests
    assert orc.triangle(t2, [2, 0, 5], [0, 0, -1])[0]        # on a vertex
    assert not orc.triangle(t2, [2.001, 0, 5], [0, 0, -1])[0]
    deg = tri([0, 0, 0], [1, 1, 1], [2, 2, 2])               # zero area: Ng = NaN, never hit
    assert not orc.triangle(deg, [0, 0, 5], [0, 0, -1])[0]


def test_triangle_winding_two_sided():
    # Ray.cu:155-158: with Ng from cross(v2-v1, v3-v1) the three edge tests are >= 0 for interior
    # points whatever side the ray comes from
    for t in (tri([-2, 0, 0], [0, 2, 0], [2, 0, 0]), tri([-2, 0, 0], [2, 0, 0], [0, 2, 0])):
        for o, d in (([0, 1, 5], [0, 0, -1]), ([0, 1, -5], [0, 0, 1])):
            assert orc.triangle(t, o, d)[0]


def test_sphere_known_answers():
    s = sph([0, 0, 0], 2)
    hit, t, n, x = orc.sphere(s, [0, 0, 5], [0, 0, -1])
    assert hit and t == 3.0 and np.allclose(x, [0, 0, 2]) and np.allclose(n, [0, 0, 1])
    hit, t, n, x = orc.sphere(s, [0, 0, 0], [0, 0, 1])       # origin inside: returns t2, but the point uses t (Ray.cu:135)
    assert hit and t == 2.0 and np.allclose(x, [0, 0, -2])
    assert not orc.sphere(s, [0, 0, 5], [0, 0, 1])[0]        # behind
    assert not orc.sphere(s, [2, 0, 5], [0, 0, -1])[0]       # tangent: discriminant == 0 is not > 1e-9
    assert not orc.sphere(s, [3, 0, 5], [0, 0, -1])[0]


def test_wall_order_and_fields():
    w, isect, rc = orc.wall([0, 50, 250], [0, -1, 0])
    assert w == 0 and np.allclose(isect["isectPoint"], [0, -200, 250]) and np.allclose(isect["normal"], [0, 1, 0])
    assert np.allclose(isect["color"], [0.2, 0.2 * np.float32(0.2), 0.2 * np.float32(0.2)]) and isect["shininess"] == 2
    assert np.all(rc == 0) and np.all(isect["reflectionCoef"] == 1)
    assert orc.wall([0, 0, 0], [-1, 0, 0])[0] == 1
    assert orc.wall([0, 0, 0], [0, 0, -1])[0] == 2
    assert orc.wall([0, 0, 0], [0, 0, 1])[0] == 3
    assert orc.wall([0, 0, 0], [1, 0, 0])[0] == 4
    assert orc.wall([0, 0, 0], [0, 1, 0])[0] == 5
    # first wall in the fixed order wins, not the nearest: from outside the room looking down, the
    # floor plane is reached beyond the ceiling plane but the floor is tested first
    w, isect, _ = orc.wall([0, 500, 0], [0, -1, 0])
    assert w == 0 and isect["isectPoint"][1] == -200
    # raytype 1 keeps the incoming coefficient and the incoming colour
    w, isect, rc = orc.wall([0, 0, 0], [0, -1, 0], raytype=1, coef_in=(.5, .25, .125), color_in=(.1, .2, .3))
    assert np.allclose(isect["reflectionCoef"], [.5, .25, .125]) and np.allclose(isect["color"], [.1, .2, .3]) and np.all(rc == 0)
    # a ray along a room edge misses every wall (strict inequalities)
    assert orc.wall([200, 0, 0], [0, -1, 0])[0] in (-1, 0, 1, 4)


@pytest.mark.parametrize("scene", ["teapot", "spheres"])
def test_bvh_equals_brute_force(scene):
    import cudaraytracer_amd as crt
    prims = crt.scene_teapot() if scene == "teapot" else crt.scene_spheres(600, seed=2)
    sc = orc.Scene(prims)
    sc.build()
    cam, tgt = crt.default_camera()
    rays = orc.primary_rays(cam, tgt, 240, 120)
    ids, ts = sc.trace(rays)
    bid, bts = sc.trace_brute(rays)
    # brute force visits scene order, the BVH its own leaf order: they can only differ on exact t ties
    diff = ids != bid
    assert np.array_equal(ts, bts)
    assert diff.sum() == 0 or np.all(ts[diff] == bts[diff])


def test_refit_grows_only_and_animation_runs():
    import cudaraytracer_amd as crt
    prims = crt.scene_teapot()
    sc = orc.Scene(prims)
    sc.build()
    before = sc.dump()["boxes"].copy()
    for _ in range(5):
        sc.animate()
        sc.refit()
    after = sc.dump()["boxes"]
    assert np.all(after[:, 0::2] <= before[:, 0::2]) and np.all(after[:, 1::2] >= before[:, 1::2])
    assert np.any(after != before)


def test_mismatch_classifier_tells_the_classes_apart():
    """tests/parity_classes.py words the failure of a parity test: fed fabricated mismatches it must name G2 for a tie,
    G3 when the reference's own tree and its own brute-force loop disagree, and G1 with the grazing predicate otherwise."""
    import cudaraytracer_amd as crt
    from parity_classes import classify, margins
    prims = crt.scene_teapot(with_sphere=True)
    sc = orc.Scene(prims); sc.build()
    cam, tgt = crt.default_camera()
    rays = orc.primary_rays(cam, tgt, 120, 60)
    ids_o, ts_o = sc.trace(rays)
    hit = np.flatnonzero(ids_o >= 0)
    assert len(hit) > 100
    ids_g, ts_g = ids_o.copy(), ts_o.copy()
    a, b, c = hit[0], hit[1], hit[2]
    ids_g[a] = (ids_o[a] + 1) % len(prims)                      # same t, other primitive: a tie resolved differently
    ids_g[b] = -1; ts_g[b] = 1e9                                # the "gpu" misses what the oracle hits
    msg = classify(ids_g, ts_g, ids_o, ts_o, ids_g != ids_o, sc, rays, prims)
    assert f"ray {a}:" in msg and "G2 equal-t tie" in msg.split(f"ray {b}:")[0]
    assert "G1 threshold grazing" in msg.split(f"ray {b}:")[1] and "nearest predicate" in msg
    # G3: pretend the reference's tree had lost hit c (what a box false-negative looks like from outside)
    ids_f, ts_f = ids_o.copy(), ts_o.copy()
    ids_f[c] = -1; ts_f[c] = 1e9
    msg3 = classify(ids_o, ts_o, ids_f, ts_f, ids_o != ids_f, sc, rays, prims)
    assert "G3 box false-negative" in msg3 and "agrees with the brute-force result" in msg3
    m = margins(prims[ids_o[c]], rays[c])
    assert all(v > 0 for v in m.values())                      # a solid hit passes every predicate with room
    sc.close()
