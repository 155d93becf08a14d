"""debug helper: the hostile scene of fill_the_chip seed N: nearest-hit ids of random rays and of rays aimed at the point light,
host-built wide tree (mode 0) against device-built (mode 2) against the oracle's tree and its brute-force loop"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cudaraytracer_amd as crt
import orc
from test_gpu_fuzz import hostile_scene, random_lights, random_scene

seed = int(sys.argv[1])
rng = np.random.default_rng(23000 + seed)
random_scene(crt, rng)
prims, kind = hostile_scene(crt, rng)
print("kind", kind, len(prims))
r2 = np.random.default_rng(1)
n = 2_000_000
o = r2.uniform(-160, 160, (n, 3)).astype(np.float32)
tgt = np.where(r2.random((n, 1)) < 0.5, np.array([[-75.784775, -99.67875, -120.4102]], np.float32), r2.uniform(-100, 100, (n, 3)).astype(np.float32))
d = (tgt - o).astype(np.float64); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d.astype(np.float32)], axis=1)
res = {}
for mode in (0, 2):
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(mode)
    res[mode] = ctx.trace(rays)
    sh = ctx.trace_shadow(rays)
    res[("sh", mode)] = sh
    ctx.close()
bad = np.flatnonzero((res[0][0] != res[2][0]) | (res[0][1].view(np.uint32) != res[2][1].view(np.uint32)))
print("nearest-hit rays differing between mode 0 and mode 2:", len(bad))
print("shadow flags differing:", int((res[("sh", 0)] != res[("sh", 2)]).sum()))
sc = orc.Scene(prims); sc.build()
io, to = sc.trace(rays, nthreads=16)
print("mode 0 vs oracle:", int((res[0][0] != io).sum()), "mode 2 vs oracle:", int((res[2][0] != io).sum()))
for i in bad[:6]:
    ib, tb = sc.trace_brute(rays[i].reshape(1, 6), nthreads=1)
    print("ray", i, rays[i], "mode0", res[0][0][i], repr(res[0][1][i]), "mode2", res[2][0][i], repr(res[2][1][i]), "oracle", io[i], repr(to[i]), "brute", ib[0], repr(tb[0]))
    for j in {int(res[0][0][i]), int(res[2][0][i])}:
        if j >= 0:
            print("   prim", j, {k: prims[j][k] for k in ("type", "position", "radius", "v1", "v2", "v3")})
sc.close()
