"""dev experiment: one rank's share of the 4K bench frame on one GPU (strong-scaling efficiency of the render alone).

  python tools/exp/rank_slice.py            the synchronous path: device ms of the slowest rank's share for 1/2/4/8 ranks
  python tools/exp/rank_slice.py async      the 1/8 (and 1/4) share with K frame slots in flight (crt_render_submit/wait):
                                            "frames"    K whole frames of the share in flight, wall ms per frame (throughput)
                                            "subframes" ONE frame of the share split over K slots by tile, wall ms per frame (latency)
                                            each with the persistent grid whole (launches take turns) or divided by K (side by side)
CRT_TRACE_BLOCK=256|128|64 picks the block size of k_trace8 for the process."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt

prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(0)
cam, tgt = crt.default_camera()
W, H, B = 3840, 2160, 3
block = os.environ.get("CRT_TRACE_BLOCK", "256")


def sync_table():
    base = None
    for nr in (1, 2, 4, 8):
        worst = 0; walls = []
        for r in range(nr if nr < 8 else 2):
            p = crt.default_params(width=W, height=H, bounces=B, rank=r, nranks=nr)
            ctx.render_device(cam, tgt, p)
            ms = []
            for _ in range(5):
                t0 = time.perf_counter(); ctx.render_device(cam, tgt, p); w = (time.perf_counter() - t0) * 1e3
                ms.append(ctx.stats()["ms_total"]); walls.append(w)
            worst = max(worst, float(np.mean(ms)))
        if base is None: base = worst
        print(json.dumps({"mode": "sync", "block": block, "nranks": nr, "device_ms": round(worst, 3), "ideal_ms": round(base / nr, 3),
                          "efficiency": round(base / nr / worst, 3), "wall_ms_per_call": round(float(np.mean(walls)), 3)}), flush=True)
    return base


def run_async(nr, K, mode, div, full_ms, frames=24):
    ps = []
    for k in range(K):
        kw = dict(width=W, height=H, bounces=B, rank=0, nranks=nr, timing=1, grid_div=div)
        if mode == "subframes":
            kw.update(tile_sub=k, tile_nsub=K)
        ps.append(crt.default_params(**kw))
    def one_pass(n):
        if mode == "frames":
            busy = [False] * K
            for i in range(n):
                k = i % K
                if busy[k]: ctx.render_wait(k)
                ctx.render_submit(k, cam, tgt, ps[k]); busy[k] = True
            for k in range(K):
                if busy[k]: ctx.render_wait(k)
        else:
            for i in range(n):
                for k in range(K): ctx.render_submit(k, cam, tgt, ps[k])
                for k in range(K): ctx.render_wait(k)
    one_pass(K * 2)
    t0 = time.perf_counter(); one_pass(frames); wall = (time.perf_counter() - t0) * 1e3 / frames
    print(json.dumps({"mode": mode, "block": block, "nranks": nr, "slots": K, "grid_div": div, "wall_ms_per_frame": round(wall, 3),
                      "ideal_ms": round(full_ms / nr, 3), "efficiency": round(full_ms / nr / wall, 3)}), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "async":
    p1 = crt.default_params(width=W, height=H, bounces=B)
    ctx.render_device(cam, tgt, p1)
    t0 = time.perf_counter()
    for _ in range(5): ctx.render_device(cam, tgt, p1)
    full = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"mode": "full_frame_sync", "block": block, "wall_ms": round(full, 3)}), flush=True)
    which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 4]
    brief = len(sys.argv) > 3
    for nr in which:
        run_async(nr, 1, "frames", 1, full)
        for K in (2, 3, 4):
            for mode in ("frames",) if brief else ("frames", "subframes"):
                for div in (1,) if brief else (1, K):
                    run_async(nr, K, mode, div, full)
else:
    sync_table()
