"""debug helper: the animation-loop test of tests/test_gpu_fuzz.py replayed for one seed, with a diagnosis of a differing synchronous frame"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cudaraytracer_amd as crt
import orc
from test_gpu_fuzz import random_scene, random_lights, _hip

def run(crt, seed):
    """the reference's loop (main.cu:534-573: move, refit or rebuild, render) with every choice drawn from the seed: tight or
    grow-only refit, host / device / asynchronous rebuild, host-side or device-side animation step, the frame rendered
    synchronously or submitted on one of the frame slots and collected up to three steps later -- after the scene has moved on.
    Every frame must be the oracle's frame of the scene as it was when the frame was submitted."""
    rng = np.random.default_rng(9000 + seed)
    prims = random_scene(crt, rng)
    n = len(prims)
    prims["direction"] = rng.normal(size=(n, 3)).astype(np.float32)
    prims["speed"] = rng.uniform(0, 1.5, n).astype(np.float32)
    prims["shininess"] = np.floor(prims["shininess"])          # frames of both kinds (zero_paths) are covered above; keep this one fast
    lights = random_lights(crt, rng)
    lights = lights[[not l["area"] for l in lights]] if len(lights) else lights
    w, h, bounces = int(rng.integers(60, 160)), int(rng.integers(40, 100)), int(rng.integers(1, 4))
    cam = rng.uniform(-150, 150, 3).astype(np.float32)
    tgt = rng.uniform(-40, 40, 3).astype(np.float32)
    p = crt.default_params(width=w, height=h, bounces=bounces)
    hip, C = _hip()
    nbytes = w * h * 3
    bufs = []
    for _ in range(4):
        q = C.c_void_p(0)
        assert hip.hipMalloc(C.byref(q), nbytes) == 0
        bufs.append(q.value)
    sc = orc.Scene(prims); sc.build()
    ctx = crt.Context(0)
    pending = {}          # slot -> (step it was submitted at, the oracle's frame of that step)
    host = prims.copy()
    log = []

    def collect(slot):
        step0, want = pending.pop(slot)
        ctx.render_wait(slot)
        got = np.zeros(nbytes, np.uint8)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), bufs[slot], nbytes, 2) == 0
        assert np.array_equal(got.reshape(h, w, 3), want), f"seed {seed}: frame submitted at step {step0} on slot {slot}; log: {log}"

    try:
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(int(rng.choice([0, 2])))
        building = False
        for step in range(int(os.environ.get("CRT_FUZZ_STEPS", "14"))):          # a soak: CRT_FUZZ_STEPS=400
            # ---- move the scene
            on_device = bool(rng.integers(0, 2)) and not building       # crt_scene_update during an asynchronous build is fine too, but keep the host copy in step
            crt.animate_host(host); sc.animate()
            if on_device:
                ctx.animate_device()
            else:
                if rng.integers(0, 2):
                    ctx.animate_device()          # the device moves it ...
                    ctx.update_scene(host)        # ... and the host overwrites it with the same bytes
                else:
                    ctx.update_scene(host)
            # ---- refit or rebuild
            kind = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
            if building:
                kind = 5 if rng.integers(0, 2) else 0
            if kind == 0: ctx.refit(False)
            elif kind == 1: ctx.refit(True)
            elif kind == 2: ctx.build(2)
            elif kind == 3:
                for s in list(pending): collect(s)          # crt_scene_set waits for the frames in flight itself; collect them first so their buffers are compared
                ctx.set_scene(host); ctx.build(0)
            elif kind == 4:
                ctx.build_async(); building = True; ctx.refit(False)
            elif kind == 5:
                if ctx.swap(wait=True): building = False
                else: ctx.refit(False)
            log.append((step, "dev" if on_device else "host", kind))
            sc.refit()
            # ---- render
            want = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16)["bgr"]
            slot = int(rng.integers(0, 4))
            if slot == 0 and rng.integers(0, 2):
                if 0 in pending: collect(0)          # crt_render is submit + wait on slot 0
                ctx.set_capture(True)
                got = ctx.render(cam, tgt, p)
                if not np.array_equal(got, want):
                    bad = (got != want).any(axis=2)
                    sc.all_boxes_hit(True)
                    ob = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16)["bgr"]
                    sc.all_boxes_hit(False)
                    print("step", step, "pixels differ", int(bad.sum()), "not explained by G3", int((got[bad] != ob[bad]).any(axis=1).sum()), "log", log)
                    wi = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, want_ids=True)
                    ys, xs = np.nonzero(bad)
                    for y, x in list(zip(ys, xs))[:4]:
                        pix = y * w + x
                        for b in range(bounces):
                            ig, tg = ctx.capture(b, w * h)
                            print(f"   [{y},{x}] gpu={got[y,x]} oracle={want[y,x]} b{b}: gpu {ig[pix]} {tg[pix]!r} oracle {wi['ids'][b][pix]} {wi['ts'][b][pix]!r}")
                    print("  prim types / radii of the gpu hits:", [(int(host[i]["type"]), float(host[i]["radius"])) for i in set(int(ctx.capture(b, w * h)[0][ys[0] * w + xs[0]]) for b in range(bounces)) if i >= 0])
                    return
                ctx.set_capture(False)
            else:
                if slot in pending: collect(slot)
                ctx.render_submit(slot, cam, tgt, p, bufs[slot])
                pending[slot] = (step, want)
            for s in [s for s, (st0, _) in pending.items() if step - st0 >= 3]:
                collect(s)
        for s in list(pending):
            collect(s)
        assert ctx.get_scene().tobytes() == host.tobytes()
    finally:
        ctx.close(); sc.close()
        for q in bufs:
            hip.hipFree(C.c_void_p(q))



run(crt, int(sys.argv[1]))
print("done")
