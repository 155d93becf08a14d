"""debug helper: hostile-scene seed of tests/test_gpu_fuzz.py, per-bounce traced sets of both sides"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cudaraytracer_amd as crt
import orc
from test_gpu_fuzz import hostile_scene, random_lights

seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
prims, kind = hostile_scene(crt, rng)
lights = random_lights(crt, rng)
if seed % 4 == 3 and len(lights):
    lights[0]["direction"] = 0; lights[0]["point"] = 0; lights[0]["area"] = 0
w, h, bounces = int(rng.integers(40, 120)), int(rng.integers(30, 90)), int(rng.integers(2, 5))
cam = rng.uniform(-150, 150, 3).astype(np.float32)
tgt = rng.uniform(-40, 40, 3).astype(np.float32)
print("seed", seed, "kind", kind, "lights", lights, w, h, bounces, "shininess values", np.unique(prims["shininess"]))
sc = orc.Scene(prims); sc.build()
sc.stats(reset=True)
o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, want_color=True, want_ids=True)
print("oracle stats", sc.stats(reset=True))
ctx = crt.Context(0)
ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0); ctx.set_capture(True)
g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces))
print("gpu stats", {k: v for k, v in ctx.stats().items() if k.startswith("rays")})
for b in range(bounces):
    ids_g, ts_g = ctx.capture(b, w * h)
    print("bounce", b, "gpu traced", int((ids_g != -9).sum()), "gpu object hits", int((ids_g >= 0).sum()), "walls", int(((ids_g <= -2) & (ids_g != -9)).sum()),
          "oracle object hits", int((o["ids"][b] >= 0).sum()), "walls", int((o["ids"][b] <= -2).sum()), "dead", int((o["ids"][b] == -1).sum()))
diff = np.abs(g.astype(np.int32) - o["bgr"].astype(np.int32)).max(axis=2)
ys, xs = np.nonzero(diff > 1)
print("bad pixels", len(ys))
for y, x in list(zip(ys, xs))[:8]:
    pix = y * w + x
    print(f"frame[{y},{x}] gpu={g[y, x]} oracle={o['bgr'][y, x]} colour={o['color'][pix]}",
          " | ".join(f"b{b}: g={ctx.capture(b, w * h)[0][pix]} o={o['ids'][b][pix]}" for b in range(bounces)))
ctx.close(); sc.close()
