"""debug helper: the frame of tests/test_gpu_fuzz.py::test_lattice_scenes_rays_through_edges_and_vertices for one seed: per-bounce ids of
the pixels that differ, and for a differing nearest hit the oracle's brute-force answer for the same ray"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cudaraytracer_amd as crt
import orc
from test_gpu_fuzz import lattice_scene, random_lights

seed = int(sys.argv[1])
rng = np.random.default_rng(17000 + seed)
prims = None
for _ in range(20):
    prims = lattice_scene(crt, rng)
    sc = orc.Scene(prims)
    try:
        sc.build()
        break
    except RuntimeError:
        prims = None
    finally:
        sc.close()
lights = random_lights(crt, rng)
for l in lights:
    l["position"] = np.round(l["position"] / 8) * 8
    l["direction"] = np.round(l["direction"] * 2)
    if not l["direction"].any():
        l["direction"] = (0, -1, 0)
cam = (rng.integers(-14, 15, 3) * 8).astype(np.float32)
tgt = (rng.integers(-3, 4, 3) * 8).astype(np.float32)
if np.array_equal(cam, tgt) or (cam[0] == tgt[0] and cam[2] == tgt[2]):
    cam = cam + np.float32(8) * np.array([1, 0, 1], np.float32)
w, h = int(rng.choice([64, 128])), int(rng.choice([32, 64]))
print("prims", len(prims), "lights", lights, "cam", cam, "tgt", tgt, w, h)
sc = orc.Scene(prims); sc.build()
o = sc.render(lights, cam, tgt, w, h, 3, nthreads=16, want_color=True, want_ids=True)
for mode in (0, 1, 2):
    ctx = crt.Context(0)
    ctx.set_scene(prims); ctx.set_lights(lights)
    try:
        ctx.build(mode)
    except crt.CrtError:
        continue
    ctx.set_capture(True)
    g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=3))
    caps = [ctx.capture(b, w * h) for b in range(3)]
    ys, xs = np.nonzero((g != o["bgr"]).any(axis=2))
    print("mode", mode, "bad pixels", len(ys))
    for y, x in list(zip(ys, xs))[:4]:
        pix = y * w + x
        print(f"  frame[{y},{x}] gpu={g[y, x]} oracle={o['bgr'][y, x]} colour={o['color'][pix]}")
        for b in range(3):
            print(f"    b{b}: gpu id {caps[b][0][pix]} t {caps[b][1][pix]!r}  oracle id {o['ids'][b][pix]} t {o['ts'][b][pix]!r}" if "ts" in o else f"    b{b}: gpu id {caps[b][0][pix]} t {caps[b][1][pix]!r}  oracle id {o['ids'][b][pix]}")
    ctx.close()
sc.close()
if len(sys.argv) > 3:
    y, x = int(sys.argv[2]), int(sys.argv[3])
    sc = orc.Scene(prims); sc.build()
    for li in range(len(lights)):
        one = lights[li:li + 1]
        oo = sc.render(one, cam, tgt, w, h, 3, nthreads=16, want_color=True, want_ids=True)
        ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(one); ctx.build(0)
        gg = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=3))
        ctx.close()
        print("light", li, "alone: differing pixels", int((gg != oo["bgr"]).any(axis=2).sum()), "at the pixel gpu", gg[y, x], "oracle", oo["bgr"][y, x])
    # the chain of the pixel's rays with the oracle's own geometry, and every shadow ray of light `li` against tree and brute force
    li = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    L = lights[li]
    pr = orc.primary_rays(cam, tgt, w, h).reshape(h * w, 6)
    ray = pr[y * w + x].copy()
    f = np.float32
    for b in range(3):
        ids, ts, nrm, pts = sc.trace(ray.reshape(1, 6), with_geometry=True)
        ib, tb = sc.trace_brute(ray.reshape(1, 6), nthreads=1)
        print("bounce", b, "ray", ray, "tree hit", ids[0], repr(ts[0]), "brute", ib[0], repr(tb[0]), "N", nrm[0], "P", pts[0])
        if ids[0] < 0:
            break
        N = nrm[0].astype(f); P = (pts[0].astype(f) + N * f(0.7)).astype(f)          # Ray.cu: the shading point is lifted off the surface
        dl = np.asarray(L["direction"], f); dl = (dl * (f(1) / np.sqrt((dl * dl).sum(dtype=f), dtype=f))).astype(f)
        print("   dot(reflectFromLight, N) =", repr(f((dl * N).sum(dtype=f))))
        sray = np.concatenate([P, -dl]).astype(f)
        i2, t2 = sc.trace(sray.reshape(1, 6)); i3, t3 = sc.trace_brute(sray.reshape(1, 6), nthreads=1)
        ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(0)
        i4, t4 = ctx.trace(sray.reshape(1, 6)); ctx.close()
        print("   shadow ray", sray, "oracle tree", i2[0], repr(t2[0]), "oracle brute", i3[0], repr(t3[0]), "gpu", i4[0], repr(t4[0]))
        d = ray[3:].astype(f)
        r = d - f(2) * f((d * N).sum(dtype=f)) * N
        r = (r * (f(1) / np.sqrt((r * r).sum(dtype=f), dtype=f))).astype(f)
        ray = np.concatenate([P, r]).astype(f)
    sc.close()
