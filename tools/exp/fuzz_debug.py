"""debug helper for tests/test_gpu_fuzz.py: print the pixels where the HIP frame and the oracle's differ for one seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cudaraytracer_amd as crt
import orc
from test_gpu_fuzz import random_scene, random_lights

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 13
rng = np.random.default_rng(1000 + seed)
prims = random_scene(crt, rng)
lights = random_lights(crt, rng)
w, h = int(rng.integers(33, 200)), int(rng.integers(17, 130))
bounces = int(rng.integers(1, 5))
cam = rng.uniform(-170, 170, 3).astype(np.float32)
cam[2] = np.float32(rng.uniform(-250, 280))
tgt = rng.uniform(-60, 60, 3).astype(np.float32)
print("seed", seed, "prims", len(prims), "lights", lights, "wh", w, h, "bounces", bounces, "cam", cam, tgt)
sc = orc.Scene(prims); sc.build()
o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, want_color=True, want_ids=True)
ctx = crt.Context(0)
ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0); ctx.set_capture(True)
g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
diff = np.abs(g.astype(np.int32) - o["bgr"].astype(np.int32)).max(axis=2)
ys, xs = np.nonzero(diff > 1)
print("bad pixels", len(ys))
caps = [ctx.capture(b, w * h) for b in range(bounces)]
# the frame is bottom-up BGR: row r of the array is image row r; pixel index as the capture planes use it
for y, x in list(zip(ys, xs))[:12]:
    print(f"frame[{y},{x}] gpu={g[y, x]} oracle={o['bgr'][y, x]}")
    for guess, pix in (("row-major", y * w + x), ("flipped", (h - 1 - y) * w + x)):
        line = []
        for b in range(bounces):
            ig, tg = caps[b][0][pix], caps[b][1][pix]
            io, to = o["ids"][b][pix], o["ts"][b][pix]
            m = ""
            if io >= 0:
                p = prims[io]
                m = f" type={p['type']} shin={p['shininess']} refl={p['reflective']}"
            line.append(f"b{b}: g=({ig},{tg:.6g}) o=({io},{to:.6g}){m}")
        print("   ", guess, " | ".join(line))
# which lights matter: re-render with each light alone
for k in range(len(lights)):
    one = lights[k:k + 1].copy()
    ctx.set_lights(one)
    g1 = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
    o1 = sc.render(one, cam, tgt, w, h, bounces, nthreads=16, want_color=True, want_ids=False)
    d1 = np.abs(g1.astype(np.int32) - o1["bgr"].astype(np.int32)).max(axis=2)
    print("light", k, "point" if one[0]["point"] else "directional", "alone: bad pixels", int((d1 > 1).sum()))
for b in range(1, bounces + 1):
    ctx.set_lights(lights)
    gb = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=b))
    ob = sc.render(lights, cam, tgt, w, h, b, nthreads=16, want_color=True, want_ids=False)
    db = np.abs(gb.astype(np.int32) - ob["bgr"].astype(np.int32)).max(axis=2)
    print("bounces", b, ": bad pixels", int((db > 1).sum()))
ctx.close(); sc.close()
