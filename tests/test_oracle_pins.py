"""Pins of the CPU oracle against the reference's OWN compiled sources (no GPU).

What can be pinned: the vendored GLM arithmetic the reference instantiates and its OBJ loader --
compiled from /root/reference by oracle/Makefile into oracle/_ref/ref_driver, and frozen as golden
vectors by tests/golden/make_golden.py.  What cannot: Ray.cu / bvh.cu (CUDA sources, unbuildable in
this image without stand-ins; the reference ships no tests or golden vectors) -- for those the only
anchors are the ray/test counts that SURVEY.md/BASELINE.md record from the surveyor's host build of the
reference (test_survey_probe_counts).
"""
import os

import numpy as np
import pytest

import orc

G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_floats(a, b):
    """bitwise equal, NaNs compare equal to NaNs regardless of payload/sign"""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((bits(a) == bits(b)) | both_nan))


def test_vec3_restatement_matches_vendored_glm_golden():
    z = np.load(os.path.join(G, "glm_probe.npz"))
    got = orc.math_probe(z["inputs"])
    want = z["outputs"]
    bad = ~((bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want)))
    assert not bad.any(), f"columns differing: {sorted(set(np.nonzero(bad)[1]))}"


@pytest.mark.skipif(not orc.ref_driver_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_vec3_restatement_matches_vendored_glm_live(tmp_path):
    rng = np.random.default_rng(5)
    x = (rng.normal(size=(20000, 9)) * 10 ** rng.uniform(-3, 3, size=(20000, 1))).astype(np.float32)
    assert same_floats(orc.math_probe(x), orc.ref_math(x, str(tmp_path)))


def test_obj_loader_matches_reference_golden():
    z = np.load(os.path.join(G, "teapot_obj.npz"))
    v, n = orc.load_obj(os.path.join(os.path.dirname(os.path.dirname(G)), "cudaraytracer_amd", "data", "teapot3.obj"))
    assert v.shape == (3072, 3)            # main.cu:236 hard-codes 3072 corners
    assert np.array_equal(bits(v), bits(z["verts"])) and np.array_equal(bits(n), bits(z["normals"]))


@pytest.mark.skipif(not orc.ref_driver_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_obj_loader_matches_reference_live(tmp_path):
    # a small file with comments, a vt line, an unknown token and negative exponents
    p = tmp_path / "m.obj"
    p.write_text("# c\nv 1 2 3\nv -1e-3 0.5 4\nv 0 0 1\nvt 0.1 0.2\nvn 0 0 1\nvn 0 1 0\nvn 1 0 0\ng grp\nf 1 2 3\nf 3 2 1\n")
    v, n = orc.load_obj(str(p))
    rv, rn = orc.ref_obj(str(p), str(tmp_path))
    assert np.array_equal(bits(v), bits(rv)) and np.array_equal(bits(n), bits(rn))


def test_survey_probe_counts():
    """SURVEY.md 3.1/5 and BASELINE.md 2 record, from a host build of the reference itself: 47 nodes,
    24 leaves, largest leaf 59; 549,612 / 1,132,664 / 2,358,648 traversals per 720x360 frame at
    1 / 2 / 4 bounces; 31,439 primary object hits.  Every hit/miss decision, facing test and
    reflection feeds these counts, so they anchor the unpinned render logic."""
    import cudaraytracer_amd as crt
    prims = crt.scene_teapot(with_sphere=True)
    sc = orc.Scene(prims)
    sc.build()
    d = sc.dump()
    assert len(d["boxes"]) == 47 and int(d["isleaf"].sum()) == 24 and int(d["count"].max()) == 59
    cam, tgt = crt.default_camera()
    lights = crt.default_lights()
    for bounces, want in ((1, 549612), (2, 1132664), (4, 2358648)):
        r = sc.render(lights, cam, tgt, 720, 360, bounces, nthreads=8, want_ids=(bounces == 1))
        st = sc.stats(reset=True)
        assert st["traversals"] == want
        if bounces == 1:
            assert st["primary"] == 259200 and st["shadow"] == 290412
            assert int((r["ids"][0] >= 0).sum()) == 31439
            # 6.5 box, 35.2 triangle, 0.08 sphere tests per traversal at bounce 0
            assert abs(st["box_tests"] / want - 6.5) < 0.05 and abs(st["tri_tests"] / want - 35.2) < 0.05
            assert abs(st["sph_tests"] / want - 0.08) < 0.005
