"""dev experiment: duration of ONE k_trace8 launch against the number of rays it serves (1M-triangle soup, rays = a strided
subset of the 4K frame's primary rays re-traced through the parity hook).  Run under rocprofv3 --kernel-trace; the launch
floor (the dependent node-visit chain of the longest ray) is the intercept."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(0)
rng = np.random.default_rng(1)
n = 1 << 22
o = rng.uniform(-150, 150, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
for k in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
    for rep in range(3):
        ctx.trace(rays[:k])
    print("traced", k, flush=True)
