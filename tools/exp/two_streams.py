"""tools/exp/two_streams.py -- do the tails of the persistent launches overlap when two tile groups of ONE frame are
rendered by two contexts (two streams, two host threads) on one GPU?  Both write their tiles into the same device
frame.  Compared with one context rendering the whole frame."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracer_amd as crt

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prims = crt.scene_soup(1000000, seed=1)
cam, tgt = crt.default_camera()
lights = crt.default_lights()
W, H, B = 3840, 2160, 3
one = crt.Context(0); one.set_scene(prims); one.set_lights(lights); one.build(0)
p1 = crt.default_params(width=W, height=H, bounces=B)
one.render_device(cam, tgt, p1)
t0 = time.perf_counter()
for _ in range(10):
    one.render_device(cam, tgt, p1)
t_one = (time.perf_counter() - t0) / 10 * 1e3
want = one.render(cam, tgt, p1).copy()
ptr, nbytes = one.blob()
ctxs = []
for r in range(groups):
    c = crt.Context(0); c.import_blob(ptr, nbytes); c.set_lights(lights); ctxs.append(c)
frame_ptr = one.frame_ptr() if hasattr(one, "frame_ptr") else None
params = [crt.default_params(width=W, height=H, bounces=B, rank=r, nranks=groups) for r in range(groups)]
bar = threading.Barrier(groups + 1)
def worker(r):
    for it in range(12):
        bar.wait()
        ctxs[r].render_device(cam, tgt, params[r])
        bar.wait()
ths = [threading.Thread(target=worker, args=(r,)) for r in range(groups)]
for t in ths: t.start()
times = []
for it in range(12):
    bar.wait(); t0 = time.perf_counter(); bar.wait(); times.append((time.perf_counter() - t0) * 1e3)
for t in ths: t.join()
print(f"one context, whole frame: {t_one:.2f} ms wall;  {groups} contexts, {groups} tile groups concurrently: {np.mean(times[2:]):.2f} ms wall (min {min(times[2:]):.2f})")
# correctness: the groups' tiles put together equal the whole frame
got = np.zeros_like(want)
for r in range(groups):
    img = ctxs[r].render(cam, tgt, params[r])
    tiles, _ = crt.tiles_owned(params[r])
    tx_n = (W + 31) // 32
    for t in tiles:
        tx, ty = t % tx_n, t // tx_n
        got[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32] = img[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32]
print("frames equal:", bool((got == want).all()))
