"""tools/exp/lpt_probe.py -- does visiting the (predictably) long rays first shorten a persistent launch?  Random rays in
the soup, ordered by the length of their chord through the scene box (longest first / shortest first / as generated)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(0)
rng = np.random.default_rng(1)
lo = np.array([-182, -182, -282], np.float32); hi = np.array([182, 182, 202], np.float32)
for n in (1048576, 4194304):
    o = np.stack([rng.uniform(-180, 180, n), rng.uniform(-180, 180, n), rng.uniform(-280, 200, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    with np.errstate(divide="ignore"):
        t1 = (lo - o) / d; t2 = (hi - o) / d
    texit = np.minimum.reduce(np.maximum(t1, t2), axis=1)          # origin is inside the box: chord = exit distance
    rays = np.ascontiguousarray(np.concatenate([o, d], 1))
    for name, order in (("as generated", np.arange(n)), ("longest chord first", np.argsort(-texit)), ("shortest chord first", np.argsort(texit)),
                        ("two buckets, long half first", np.argsort(texit < np.median(texit), kind="stable"))):
        r = np.ascontiguousarray(rays[order])
        ms = []
        for _ in range(4):
            ctx.trace(r); ms.append(ctx.stats()["ms_trace"])
        print(f"n={n} {name:30s}: {min(ms):.3f} ms", flush=True)
