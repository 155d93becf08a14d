#!/usr/bin/env python3
"""tools/bench_configs.py -- BASELINE.json's other configurations (bench.py measures configs[3]).

One JSON line per configuration: frame ms (device), Mrays/s, and for the scenes the oracle renders in
seconds its time on the host cores beside it.  Run on the GPU box: python tools/bench_configs.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402
import orc  # noqa: E402


def rays_of(st):
    return st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]


def timed_frames(ctx, cam, tgt, p, n):
    ctx.render_device(cam, tgt, p)
    ms, rays = [], 0
    for _ in range(n):
        ctx.render_device(cam, tgt, p)
        st = ctx.stats()
        ms.append(st["ms_total"]); rays = rays_of(st)
    return float(np.mean(ms)), rays


def teapot(width, height, bounces, name, threads):
    prims = crt.scene_teapot(with_sphere=True)
    lights = crt.default_lights(); cam, tgt = crt.default_camera()
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
    p = crt.default_params(width=width, height=height, bounces=bounces)
    ms, rays = timed_frames(ctx, cam, tgt, p, 20)
    g = ctx.render(cam, tgt, p)
    sc = orc.Scene(prims); sc.build()
    out = {"config": name, "frame_ms_device": round(ms, 4), "rays_per_frame": rays, "mrays_per_s": round(rays / ms / 1e3, 1)}
    for nt in threads:
        sc.stats(reset=True)
        t0 = time.perf_counter()
        o = sc.render(lights, cam, tgt, width, height, bounces, nthreads=nt)
        dt = time.perf_counter() - t0
        st = sc.stats(reset=True)
        out[f"oracle_ms_{nt}_threads"] = round(dt * 1e3, 1)
        out[f"oracle_mrays_per_s_{nt}_threads"] = round(st["traversals"] / dt / 1e6, 3)
    diff = np.abs(g.astype(np.int32) - o["bgr"].astype(np.int32)).max(axis=2)
    out["pixels_beyond_1_lsb"] = int((diff > 1).sum()); out["pixels_off_by_1"] = int((diff == 1).sum())
    ctx.close(); sc.close()
    print(json.dumps(out), flush=True)


def spheres(n=4096, frames=60, width=1920, height=1080, bounces=2, rebuild_mode=0, prims=None, label=None):
    if prims is None:
        prims = crt.scene_spheres(n, seed=1)
    lights = crt.default_lights(); cam, tgt = crt.default_camera()
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
    p = crt.default_params(width=width, height=height, bounces=bounces)
    ctx.render_device(cam, tgt, p)
    if rebuild_mode in (2, 3):                  # the first device build allocates its scratch (and, asynchronous, the spare blob): not part of the loop
        if rebuild_mode == 3:
            ctx.build_async(); ctx.swap(True)
        else:
            ctx.build(2)
        ctx.render_device(cam, tgt, p)
    upd, ren, kinds, rays = [], [], [], 0
    building = False
    for f in range(1, frames + 1):
        t0 = time.perf_counter()
        ctx.animate_device()
        if rebuild_mode == 3:                   # device rebuild under the following frames (crt_bvh_build_async / crt_bvh_swap)
            if f % 20 == 0:
                ctx.build_async(); building = True
            if building and ctx.swap(False):
                building = False; kinds.append("swap")
            else:
                ctx.refit(False); kinds.append("rebuild" if f % 20 == 0 else "refit")
        elif f % 20 == 0:
            ctx.build(rebuild_mode); kinds.append("rebuild")   # 0: host SAH build + upload, 2: LBVH on the device
        else:
            ctx.refit(False); kinds.append("refit")
        ctx.sync()
        upd.append((time.perf_counter() - t0) * 1e3)
        ctx.render_device(cam, tgt, p)
        st = ctx.stats(); ren.append(st["ms_total"]); rays = rays_of(st)
    if building:
        ctx.swap(True)
    upd = np.array(upd); ren = np.array(ren); is_rb = np.array([k == "rebuild" for k in kinds])
    swaps = [i + 1 for i, k in enumerate(kinds) if k == "swap"]
    print(json.dumps({"config": (label or f"configs[2]: {n} animated spheres") + f", {width}x{height}, bounces={bounces}, {frames} frames, "
                                "device animation + refit every frame, " + {0: "host SAH", 2: "device LBVH", 3: "asynchronous device LBVH"}[rebuild_mode] + " rebuild every 20th",
                      "swap_frames": swaps, "frame_ms_around_rebuilds": {str(f): [round(float(upd[f - 1]), 3), round(float(ren[f - 1]), 3)] for f in range(1, frames + 1) if (f % 20) in (0, 1, 2, 3) and f >= 20},
                      "frame_ms_worst_over_mean": round(float((upd + ren).max() / (upd + ren).mean()), 3),
                      "render_ms_mean": round(float(ren.mean()), 4), "render_ms_worst": round(float(ren.max()), 4),
                      "update_ms_refit_mean": round(float(upd[~is_rb].mean()), 4), "update_ms_refit_worst": round(float(upd[~is_rb].max()), 4),
                      "update_ms_rebuild_mean": round(float(upd[is_rb].mean()), 4),
                      "frame_ms_mean": round(float((upd + ren).mean()), 4), "frame_ms_worst": round(float((upd + ren).max()), 4),
                      "rays_per_frame_last": rays, "mrays_per_s_render": round(rays / float(ren.mean()) / 1e3, 1)}), flush=True)
    ctx.close()


def soup_8k_one_rank(nranks=8, spp=16):
    prims = crt.scene_soup(1000000, seed=1)
    lights = crt.default_lights(); cam, tgt = crt.default_camera()
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
    p = crt.default_params(width=7680, height=4320, bounces=3, spp=spp, rank=0, nranks=nranks)
    ctx.render_device(cam, tgt, p)
    t0 = time.perf_counter()
    ctx.render_device(cam, tgt, p); ctx.sync()
    wall = (time.perf_counter() - t0) * 1e3
    st = ctx.stats()
    print(json.dumps({"config": f"configs[4] on ONE GPU as logical rank 0 of {nranks}: 1M-triangle soup, 7680x4320, {spp} spp, bounces=3 "
                                "(this rank's tiles only; the 8-GPU frame is the max over ranks plus the tile gather)",
                      "rank_frame_ms_wall": round(wall, 2), "rank_frame_ms_device": round(st["ms_total"], 2),
                      "rank_rays": rays_of(st), "mrays_per_s": round(rays_of(st) / st["ms_total"] / 1e3, 1)}), flush=True)
    ctx.close()


def area_lights(samples=30):
    """soft shadows multiply the shadow rays: one area light x `samples` on the bench scene."""
    prims = crt.scene_soup(1000000, seed=1)
    cam, tgt = crt.default_camera()
    lights = np.zeros(2, crt.LIGHT_DTYPE)
    lights[0] = crt.default_lights()[0]
    lights[1]["area"] = 1; lights[1]["position"] = (0, 190, 100); lights[1]["radius"] = 60; lights[1]["color"] = (.8, .8, .8)
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(lights); ctx.build(0)
    for w, h, b in ((1920, 1080, 2), (3840, 2160, 2)):
        p = crt.default_params(width=w, height=h, bounces=b, area_samples=samples)
        ms, rays = timed_frames(ctx, cam, tgt, p, 3)
        st = ctx.stats()
        print(json.dumps({"config": f"area light x {samples} samples + 1 directional light, 1M-triangle soup, {w}x{h}, bounces={b}",
                          "frame_ms_device": round(ms, 3), "rays_per_frame": rays, "rays_shadow": st["rays_shadow"],
                          "mrays_per_s": round(rays / ms / 1e3, 1), "ms_trace": round(st["ms_trace"], 3),
                          "ms_shade_resolve": round(st["ms_shade"], 3)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["0", "1", "2", "2big", "4", "area"]
    if "area" in which:
        area_lights()
    if "0" in which:
        teapot(720, 360, 1, "configs[0]: teapot3.obj + sphere, 720x360, bounces=1 (no reflection)", (1, 16))
    if "1" in which:
        teapot(1920, 1080, 4, "configs[1]: teapot3.obj + sphere, 1920x1080, reflection depth 3 (bounces=4)", (16,))
    if "2" in which:
        spheres()
        spheres(rebuild_mode=2)
        spheres(rebuild_mode=3)
    if "2big" in which:       # the animation loop on the 1M-triangle soup: only feasible with the device rebuild
        soup = crt.scene_soup(1000000, seed=1)
        spheres(frames=40, width=3840, height=2160, bounces=3, rebuild_mode=2, prims=soup, label="animation loop on the 1M-triangle soup")
        spheres(frames=60, width=3840, height=2160, bounces=3, rebuild_mode=3, prims=soup, label="animation loop on the 1M-triangle soup")
        spheres(frames=21, width=3840, height=2160, bounces=3, rebuild_mode=0, prims=soup, label="animation loop on the 1M-triangle soup")
    if "4" in which:
        soup_8k_one_rank()
