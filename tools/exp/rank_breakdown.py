"""dev experiment: where one of 8 ranks' share of the 4K bench frame spends its time (plain launches vs HIP graph replay)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
cam, tgt = crt.default_camera()
for nr in (8, 4):
    for graph in (0, 1):
        p = crt.default_params(width=3840, height=2160, bounces=3, rank=0, nranks=nr)
        p.use_graph = graph
        for _ in range(3):
            ctx.render_device(cam, tgt, p)
        ms, wall = [], []
        for _ in range(10):
            t0 = time.perf_counter(); ctx.render_device(cam, tgt, p); wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(ctx.stats()["ms_total"])
        st = ctx.stats()
        print(f"nranks={nr} graph={graph}: device {np.mean(ms):.3f} ms, wall {np.mean(wall):.3f} ms; trace {st['ms_trace']:.3f} shade {st['ms_shade']:.3f} other {st['ms_other']:.3f}", flush=True)
