"""tools/exp/build_quality.py -- trace time of the host SAH tree (mode 0) against the device LBVH (mode 2) on scenes
with structure (teapot mesh, sphere cloud, clustered soup), 1920x1080, bounces=3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracer_amd as crt

def clustered(n, seed):
    rng = np.random.default_rng(seed)
    p = crt.scene_soup(n, seed=seed)
    centres = rng.uniform(-150, 150, (40, 3)) * np.array([1, 1, 1.3])
    which = rng.integers(0, 40, n)
    c = centres[which] + rng.normal(size=(n, 3)) * 12
    old = p["v1"].copy()
    for f in ("v1", "v2", "v3"):
        p[f] = (p[f] - old + c * 0 + c).astype(np.float32) if f == "v1" else (p[f] - old + c).astype(np.float32)
    return p

scenes = {
    "teapot+sphere": crt.scene_teapot(with_sphere=True),
    "teapot x64 grid": None,
    "4096 spheres": crt.scene_spheres(4096, seed=1),
    "200k clustered soup": clustered(200000, 3),
    "1M uniform soup": crt.scene_soup(1000000, seed=1),
}
tp = crt.scene_teapot(with_sphere=False)
grid = []
for i in range(8):
    for j in range(8):
        q = tp.copy()
        for f in ("v1", "v2", "v3"):
            q[f] = (q[f] * np.float32(0.4) + np.array([(i - 3.5) * 45, -100 + (j % 3) * 60, (j - 3.5) * 60], np.float32)).astype(np.float32)
        grid.append(q)
scenes["teapot x64 grid"] = np.concatenate(grid)
cam, tgt = crt.default_camera()
for name, prims in scenes.items():
    row = []
    for mode in (0, 2):
        ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(mode)
        p = crt.default_params(width=1920, height=1080, bounces=3)
        ctx.render_device(cam, tgt, p)
        ms = []
        for _ in range(5):
            ctx.render_device(cam, tgt, p); ms.append(ctx.stats()["ms_trace"])
        st = ctx.stats()
        row.append((np.mean(ms), st["nodes"], st["depth"]))
        ctx.close()
    print(f"{name:22s} n={len(prims):8d}  host SAH: trace {row[0][0]:7.3f} ms nodes {row[0][1]} depth {row[0][2]}   device LBVH: trace {row[1][0]:7.3f} ms nodes {row[1][1]} depth {row[1][2]}   ratio {row[1][0] / row[0][0]:.2f}", flush=True)
