"""dev experiment: k_trace8 launch time against ray count (fixed cost / tail of a persistent launch)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(0)
rng = np.random.default_rng(1)
for n in (1000, 65536, 262144, 524288, 1048576, 2097152, 4194304, 8388608):
    o = np.stack([rng.uniform(-180, 180, n), rng.uniform(-180, 180, n), rng.uniform(-280, 200, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.ascontiguousarray(np.concatenate([o, d], 1))
    ms = []
    for _ in range(4):
        ctx.trace(rays); ms.append(ctx.stats()["ms_trace"])
    print(f"n={n}: {min(ms):.3f} ms  ({n / min(ms) / 1e3:.0f} Mrays/s)", flush=True)
