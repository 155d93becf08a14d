"""tools/exp/soak.py [frames] -- the reference's frame loop (animate, refit, rebuild every 20th, render) for thousands of
frames on one context: device memory must not grow, no call may fail, and the last frame must equal the one a fresh
context renders after the same number of animation steps."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracer_amd as crt

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value

prims = np.concatenate([crt.scene_soup(300000, seed=1), crt.scene_spheres(2000, seed=1), crt.scene_teapot(with_sphere=True)])
cam, tgt = crt.default_camera()
p = crt.default_params(width=1920, height=1080, bounces=3)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
ctx.render_device(cam, tgt, p)
base = None
t0 = time.time()
for f in range(1, frames + 1):
    ctx.animate_device()
    if f % 20 == 0:
        ctx.build(2)
    else:
        ctx.refit(False)
    ctx.render_device(cam, tgt, p)
    if f == 40:
        base = free_bytes()
    if f % 500 == 0:
        print(f"frame {f}: {time.time() - t0:.1f} s, free device memory {free_bytes() / 2**20:.0f} MiB (at frame 40: {base / 2**20:.0f})", flush=True)
last = ctx.render(cam, tgt, p).copy()
grew = base - free_bytes()
ctx.close()
fresh = crt.Context(0); fresh.set_scene(prims); fresh.set_lights(crt.default_lights()); fresh.build(0)
for f in range(1, frames + 1):
    fresh.animate_device()
fresh.build(2)
want = fresh.render(cam, tgt, p)
fresh.close()
diff = int((last != want).any(axis=2).sum())
print(f"{frames} frames in {time.time() - t0:.1f} s; device memory grew by {grew / 2**20:.1f} MiB since frame 40; pixels differing from a fresh context: {diff}")
assert grew <= 64 << 20 and diff == 0
