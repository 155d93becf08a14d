"""dev experiment: device time of one rank's share of the 4K bench frame for 1/2/4/8 ranks (strong-scaling efficiency of the render alone)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
cam, tgt = crt.default_camera()
base = None
for nr in (1, 2, 4, 8):
    worst = 0; walls = []
    for r in range(nr):
        p = crt.default_params(width=3840, height=2160, bounces=3, rank=r, nranks=nr)
        ctx.render_device(cam, tgt, p)
        ms = []; 
        for _ in range(5):
            t0 = time.perf_counter(); ctx.render_device(cam, tgt, p); w = (time.perf_counter() - t0) * 1e3
            ms.append(ctx.stats()["ms_total"]); walls.append(w)
        worst = max(worst, float(np.mean(ms)))
    if base is None: base = worst
    print(f"nranks={nr}: slowest rank device {worst:.3f} ms (ideal {base / nr:.3f}), render efficiency {base / nr / worst:.3f}, wall per call {np.mean(walls):.3f} ms", flush=True)
