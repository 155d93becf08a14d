"""How far apart are the two readings of the reference's `pow(float, 2)` / `sqrt(pow + pow + pow)` expressions
(Ray.cu:110, 592-593, 606-607, 626-627)?  The oracle evaluates them in float, as the CUDA device build of the reference
does (the kernels' parity target); a HOST build of the same source promotes them to double (SURVEY.md 8(c) names the host
build).  liboracle_hostpow.so is the oracle compiled with -DORC_HOST_POW; this test renders two scenes with both and
bounds the difference -- the stated exception class of DESIGN.md 2 ("G7 pow promotion").  No GPU involved."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SNIPPET = r'''
import sys, json, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import cudaraytracer_amd as crt, orc
def scenes():
    cam, tgt = crt.default_camera()
    # (a) configs[2]'s kind of scene: radii uniform in [2, 8], so r*r is not exact in float
    yield "spheres", crt.scene_spheres(768, seed=1), crt.default_lights(), cam, tgt
    # (b) a point light (attenuation + distance tests, Ray.cu:604-632) and a directional one over the teapot and spheres
    lights = np.zeros(2, crt.LIGHT_DTYPE)
    lights[0] = crt.default_lights()[0]
    lights[1]["point"] = 1; lights[1]["position"] = (30.5, 120.25, 90.125); lights[1]["color"] = (.9, .8, .7)
    lights[1]["constantTerm"] = .9; lights[1]["linearTerm"] = .0007; lights[1]["quadraticTerm"] = .00003
    yield "point_light", np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_spheres(200, seed=3)]), lights, cam, tgt
out = {{}}
for name, prims, lights, cam, tgt in scenes():
    sc = orc.Scene(prims); sc.build()
    r = sc.render(lights, cam, tgt, 480, 270, 3, nthreads=8, want_color=True, want_ids=True)
    np.savez({out!r} + name + ".npz", bgr=r["bgr"], ids=r["ids"], ts=r["ts"], color=r["color"])
    sc.close()
'''


def _render(tmp_path, variant):
    out = str(tmp_path / (variant or "float")) + "_"
    env = dict(os.environ)
    if variant:
        env["ORC_VARIANT"] = variant
    else:
        env.pop("ORC_VARIANT", None)
    subprocess.check_call([sys.executable, "-c", SNIPPET.format(root=os.path.dirname(HERE), here=HERE, out=out)], env=env)
    return {k: np.load(out + k + ".npz") for k in ("spheres", "point_light")}


def test_float_and_double_pow_readings_differ_by_a_bounded_handful(tmp_path):
    a, b = _render(tmp_path, ""), _render(tmp_path, "hostpow")
    report = {}
    for name in a:
        ida, idb = a[name]["ids"], b[name]["ids"]
        ta, tb = a[name]["ts"], b[name]["ts"]
        same_hit = ida == idb
        px = np.abs(a[name]["bgr"].astype(np.int32) - b[name]["bgr"].astype(np.int32)).max(axis=2)
        report[name] = {
            "rays": int(ida.size), "ids_differ": int((~same_hit).sum()),
            "t_bits_differ_on_same_hit": int((same_hit & (ta.view(np.uint32) != tb.view(np.uint32))).sum()),
            "t_max_rel": float(np.max(np.abs(ta - tb)[same_hit] / np.maximum(np.abs(ta[same_hit]), 1e-30))) if same_hit.any() else 0.0,
            "pixels": int(px.size), "pixels_differ": int((px > 0).sum()), "pixels_beyond_1_lsb": int((px > 1).sum()),
        }
    print("hostpow delta:", json.dumps(report))
    for name, r in report.items():
        # the classes are the same grazing classes as SURVEY 9.1 G1 (a predicate a few ulp from its threshold) and G5
        # (8-bit truncation): a handful of rays, never a visible difference
        assert r["ids_differ"] <= 2e-4 * r["rays"], (name, r)
        assert r["t_max_rel"] <= 1e-3, (name, r)
        assert r["pixels_differ"] <= 2e-3 * r["pixels"], (name, r)
        assert r["pixels_beyond_1_lsb"] <= 1e-4 * r["pixels"], (name, r)
    # and the two readings are NOT identical on these scenes (for r = 20, the reference's own sphere, they are): the switch works
    assert sum(r["t_bits_differ_on_same_hit"] + r["ids_differ"] + r["pixels_differ"] for r in report.values()) > 0
