"""dev experiment: do persistent launches of independent tile sets overlap their tails when issued from different streams?
K logical ranks on ONE GPU (crt_mgpu with the device repeated: a context, a stream and a host thread each) render the 4K bench
frame concurrently; the wall time of the render phase is compared with one context rendering the whole frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(int(os.environ.get("TRIS", 1_000_000)), seed=1)
cam, tgt = crt.default_camera()
for k in (1, 2, 3, 4):
    mg = crt.MultiGpu([0] * k)
    mg.root.set_scene(prims); mg.set_lights(crt.default_lights()); mg.root.build(0); mg.sync_scene()
    p = crt.default_params(width=3840, height=2160, bounces=3)
    mg.render_device(cam, tgt, p)
    rs = []; ws = []
    for _ in range(6):
        t0 = time.perf_counter(); mg.render_device(cam, tgt, p); ws.append((time.perf_counter() - t0) * 1e3)
        rs.append(mg.times()["ms_render"])
    per = [mg.rank_stats(r)["ms_total"] for r in range(k)]
    print(f"logical ranks {k}: render phase {np.mean(rs):.3f} ms (min {np.min(rs):.3f}), whole call {np.mean(ws):.3f} ms; per-rank device ms {[round(x, 2) for x in per]}", flush=True)
    mg.close()
