"""debug helper: one scene of tests/test_gpu_fuzz.py::test_random_scenes_at_frame_sizes_that_fill_the_chip (seed, family),
every build mode against the oracle, per-bounce ids of the differing pixels"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cudaraytracer_amd as crt
import orc
from test_gpu_fuzz import hostile_scene, random_lights, random_scene, lattice_scene

seed, want = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(23000 + seed)
w, h, bounces = 960, 540, 3
scenes = [("random", random_scene(crt, rng)), ("hostile", hostile_scene(crt, rng)[0])]
for _ in range(20):
    p = lattice_scene(crt, rng)
    sc = orc.Scene(p)
    try:
        sc.build(); scenes.append(("lattice", p)); break
    except RuntimeError:
        pass
    finally:
        sc.close()
for name, prims in scenes:
    lights = random_lights(crt, rng)
    cam = (rng.integers(-14, 15, 3) * 8).astype(np.float32) if name == "lattice" else rng.uniform(-150, 150, 3).astype(np.float32)
    tgt = rng.uniform(-40, 40, 3).astype(np.float32)
    if cam[0] == tgt[0] and cam[2] == tgt[2]:
        cam[0] += 8
    if name != want:
        continue
    print(name, len(prims), "prims", len(lights), "lights", lights)
    sc = orc.Scene(prims); sc.build()
    o = sc.render(lights, cam, tgt, w, h, bounces, nthreads=16, area_samples=4, want_color=True, want_ids=True)
    for mode in (0, 2):
        ctx = crt.Context(0)
        ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(mode); ctx.set_capture(True)
        g = ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, area_samples=4))
        caps = [ctx.capture(b, w * h) for b in range(bounces)]
        ys, xs = np.nonzero((g != o["bgr"]).any(axis=2))
        print("mode", mode, "bad pixels", len(ys))
        for y, x in list(zip(ys, xs))[:8]:
            pix = y * w + x
            print(f"  frame[{y},{x}] gpu={g[y, x]} oracle={o['bgr'][y, x]} colour={o['color'][pix]}")
            for b in range(bounces):
                ig, tg = caps[b][0][pix], caps[b][1][pix]
                print(f"    b{b}: gpu id {ig} t {tg!r}  oracle id {o['ids'][b][pix]}")
                for i in (ig, o['ids'][b][pix]):
                    if i >= 0:
                        print("      prim", i, {k: prims[i][k] for k in ("type", "position", "radius", "v1", "v2", "v3", "reflective")})
        if len(ys):
            y, x = ys[0], xs[0]
            pr = orc.primary_rays(cam, tgt, w, h).reshape(h * w, 6)[y * w + x].reshape(1, 6)
            print("  primary ray", pr[0], "oracle tree", sc.trace(pr), "oracle brute", sc.trace_brute(pr, nthreads=1), "gpu", ctx.trace(pr))
        ctx.close()
    sc.close()
    # which light: the same frames with one light at a time, mode 0 against mode 2
    if len(sys.argv) > 3:
        y, x = int(sys.argv[3]), int(sys.argv[4])
        for li in range(len(lights)):
            one = lights[li:li + 1]
            fr = []
            for mode in (0, 2):
                ctx = crt.Context(0)
                ctx.set_scene(prims); ctx.set_lights(one); ctx.build(mode)
                fr.append(ctx.render_hdr(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, area_samples=4)) if hasattr(ctx, "render_hdr")
                          else ctx.render(cam, tgt, crt.default_params(width=w, height=h, bounces=bounces, area_samples=4)))
                ctx.close()
            nb = int((fr[0] != fr[1]).any(axis=2).sum())
            print("light", li, "pixels differing between mode 0 and mode 2:", nb, "at the pixel:", fr[0][y, x], fr[1][y, x])
