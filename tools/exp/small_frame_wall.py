"""tools/exp/small_frame_wall.py -- wall time per frame against device time for the reference's own case (teapot 720x360):
how much of a small frame is host-side launch work?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_teapot(with_sphere=True)
cam, tgt = crt.default_camera()
ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
for w, h, b in ((720, 360, 1), (720, 360, 2), (1920, 1080, 4)):
    p = crt.default_params(width=w, height=h, bounces=b)
    for _ in range(20): ctx.render_device(cam, tgt, p)
    n = 300
    t0 = time.perf_counter()
    dev = 0.0
    for _ in range(n):
        ctx.render_device(cam, tgt, p); dev += ctx.stats()["ms_total"]
    wall = (time.perf_counter() - t0) / n * 1e3
    st = ctx.stats()
    print(f"{w}x{h} bounces={b}: wall {wall:.3f} ms/frame, device {dev / n:.3f} ms (trace {st['ms_trace']:.3f}, shade {st['ms_shade']:.3f}, other {st['ms_other']:.3f})")
    t0 = time.perf_counter()
    for _ in range(n):
        img = ctx.render(cam, tgt, p)
    print(f"    with the copy to the host: wall {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame")
    g = crt.default_params(width=w, height=h, bounces=b, use_graph=1)
    for _ in range(20): ctx.render_device(cam, tgt, g)
    t0 = time.perf_counter(); dev = 0.0
    for _ in range(n):
        ctx.render_device(cam, tgt, g); dev += ctx.stats()["ms_total"]
    print(f"    replayed as a HIP graph: wall {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame, device {dev / n:.3f} ms")
