#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its own workload: Mrays/s (primary + shadow + reflection
traversals actually issued) and frame ms on the synthetic 1M-triangle soup at 3840x2160, bounces=3
(configs[3]); on N GPUs the frame grows with N at a fixed pixel count per GPU and its tiles are sharded over the
ranks (weak scaling, configs[4]'s mechanism; --strong shards the same 4K frame instead).

A "step" is one frame of the hot path: gen_primary -> [trace -> shade -> shadow trace -> resolve] x
bounces -> convert, with the scene + BVH already resident in HBM and the frame left in HBM
(N>1: plus the tile gather to rank 0 over RCCL).  One JSON line on stdout from rank 0.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_ISSUE_PEAK_G = 256 * 4 * 2.4 / 4    # 256 CUs x 4 SIMDs, one 64-wide VALU instruction per SIMD per 4 clocks, 2.4 GHz
_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (the GPU harness kills a command that stays silent for minutes)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(nrays, nbox, ntri, nsph):
    """32 B per AABB test (24 B bounds + 8 B link), 36 B per triangle test (9 floats),
    16 B per sphere test, 48 B per ray (32 B ray record + 16 B hit record)."""
    return 32 * nbox + 36 * ntri + 16 * nsph + 48 * nrays


def cpu_baseline(prims, lights, cam, tgt, width, height, bounces, max_threads):
    """The oracle (CPU restatement of the reference's intersect/shade logic, 64-primitive leaves,
    recursive unordered traversal) timed on this box's host cores on a bounded sample of the same
    workload: the centred 1/16 sub-rectangle of the frame.  A reported baseline, not the target."""
    import orc
    sc = orc.Scene(prims)
    sc.build()
    cores = max(1, min(max_threads, len(os.sched_getaffinity(0))))
    # probe a 16x16 block at the frame centre, then size the sample for roughly 15 s of CPU work
    # (on the 1M-triangle soup the reference's unordered, unpruned 64-leaf traversal tests ~1e5
    # triangles per ray, so the sample is a few thousand rays)
    cx, cy = width // 2, height // 2
    sc.stats(reset=True)
    t0 = time.perf_counter()
    sc.render(lights, cam, tgt, width, height, bounces, rect=[cx - 8, cy - 8, cx + 8, cy + 8], nthreads=cores)
    per_px = (time.perf_counter() - t0) / 256
    log(f"cpu baseline probe: {per_px * 1e6:.1f} us per pixel on {cores} threads")
    side_w, side_h = 16, 16
    while side_w * 2 <= width // 4 and side_h * 2 <= height // 4 and (side_w * 2) * (side_h * 2) * per_px < 15.0:
        side_w *= 2
        side_h *= 2
    rect = [cx - side_w // 2, cy - side_h // 2, cx - side_w // 2 + side_w, cy - side_h // 2 + side_h]
    sc.stats(reset=True)
    t0 = time.perf_counter()
    sc.render(lights, cam, tgt, width, height, bounces, rect=rect, nthreads=cores)
    dt = time.perf_counter() - t0
    st = sc.stats(reset=True)
    sc.close()
    return {
        "value": round(st["traversals"] / dt / 1e6, 6), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"centred {rect[2] - rect[0]}x{rect[3] - rect[1]} sub-rectangle of the {width}x{height} frame, "
                  f"bounces={bounces}: {st['traversals']} traversals issued by the reference's logic "
                  f"({st['live_traversals']} on live paths) in {dt:.2f} s, "
                  f"{st['box_tests'] / max(1, st['traversals']):.1f} box + {st['tri_tests'] / max(1, st['traversals']):.1f} "
                  f"triangle tests per traversal",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--bvh", type=int, default=0, help="0 wide SAH built on the host (default), 1 reference topology, 2 wide LBVH built on the device")
    ap.add_argument("--sort-rays", type=int, default=0, help="1: visit secondary rays in Morton order of their origin cell")
    ap.add_argument("--strong", action="store_true", help="N > 1: shard the SAME --width x --height frame (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--out-bmp", default=None)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    dist = torch = None
    if world > 1:
        # torch BEFORE the library: PyTorch bundles its own HIP runtime, and a process that loads libcrt_amd.so (system
        # runtime) first and torch second ends up with two runtimes, the second of which finds no device
        import torch
        import torch.distributed as dist
    import cudaraytracer_amd as crt

    cam, tgt = crt.default_camera()
    lights = crt.default_lights()
    # N > 1: WEAK scaling -- the pixels per GPU stay those of the --width x --height frame and the frame grows with N
    # (3840x2160, 7680x2160, 7680x4320, 15360x4320 for N = 1, 2, 4, 8; configs[4] is the N = 4 frame size); its tiles
    # are sharded over the ranks and gathered to rank 0 every frame
    fw, fh = a.width, a.height
    if world > 1 and not a.strong:
        k = world.bit_length() - 1 if world & (world - 1) == 0 else None
        if k is None:
            fw = a.width * world
        else:
            fw, fh = a.width << ((k + 1) // 2), a.height << (k // 2)
    params = crt.default_params(width=fw, height=fh, bounces=a.bounces, spp=a.spp, rank=rank, nranks=world,
                                sort_rays=a.sort_rays)
    prims = crt.scene_soup(a.tris, seed=1) if (rank == 0 or world == 1) else None
    log(f"scene generated: {a.tris} triangles")

    if world > 1:
        # rehearsal on a one-GPU box: CRT_DIST_BACKEND=gloo CRT_FORCE_DEVICE=0 puts every rank on one device
        backend = os.environ.get("CRT_DIST_BACKEND", "nccl")
        if "CRT_FORCE_DEVICE" in os.environ:
            local_rank = int(os.environ["CRT_FORCE_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        from cudaraytracer_amd.parallel import ShardedRenderer
        sr = ShardedRenderer(dist, torch, rank, world, local_rank)
        sr.broadcast_scene(prims, lights, a.bvh)
        ctx = sr.ctx

        def step():
            sr.render(cam, tgt, params)

        def fence():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        ctx = crt.Context(local_rank)
        ctx.set_scene(prims)
        ctx.set_lights(lights)
        ctx.build(a.bvh)

        def step():
            ctx.render_device(cam, tgt, params)

        def fence():
            ctx.sync()

    # one untimed frame with the counting kernels: AABB / triangle / sphere test counts of exactly
    # the traversal the timed kernels perform (same packed BVH, same order) -> algorithmic bytes
    log("scene + BVH resident; counting frame")
    ctx.set_counting(True)
    step()
    cst = ctx.stats()
    ctx.set_counting(False)
    log(f"counting frame done: device {cst['ms_total']:.2f} ms; warmup")

    for _ in range(a.warmup):
        step()
    fence()
    log("timed steps")
    t0 = time.perf_counter()
    acc = {"ms_trace": 0.0, "ms_trace_nearest": 0.0, "ms_trace_shadow": 0.0, "ms_trace_mixed": 0.0, "ms_total": 0.0, "ms_shade": 0.0,
           "ms_other": 0.0, "launches_trace_nearest": 0, "launches_trace_shadow": 0, "launches_trace_mixed": 0}
    for _ in range(a.steps):
        step()
        st = ctx.stats()
        for k in acc:
            acc[k] += st[k]
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed steps done: {elapsed / a.steps * 1e3:.3f} ms/step")

    rays_local = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
    rays = np.array([st["rays_primary"], st["rays_shadow"], st["rays_reflect"]], np.float64)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor(rays, dtype=torch.float64, device="cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = r.cpu().numpy()
    total_rays = float(rays.sum())

    if rank == 0:
        ms_step = elapsed / a.steps * 1e3
        # roofline of the dominant kernel on rank 0's launches.  Wide BVH: k_trace8 serves the shadow rays of a bounce
        # and the nearest-hit rays of the next in one launch, so the kernel's figure covers all its launches of a
        # frame (bounces + 1) and all their rays; reference topology: k_trace<NEAREST>'s own launches.
        n_near = cst["rays_primary"] + cst["rays_reflect"]
        bytes_near = algorithmic_bytes(n_near, cst["box_tests"], cst["tri_tests"], cst["sph_tests"])
        bytes_shadow = algorithmic_bytes(cst["rays_shadow"], cst["box_tests_shadow"], cst["tri_tests_shadow"],
                                         cst["sph_tests_shadow"])
        if a.bvh != 1:
            launches = max(1, acc["launches_trace_nearest"] + acc["launches_trace_shadow"] + acc["launches_trace_mixed"])
            avg_ms = acc["ms_trace"] / launches
            per_launch_bytes = (bytes_near + bytes_shadow) / max(1, cst["launches_trace_nearest"] + cst["launches_trace_shadow"]
                                                                  + cst["launches_trace_mixed"])
        else:
            launches = max(1, acc["launches_trace_nearest"])
            avg_ms = acc["ms_trace_nearest"] / launches
            per_launch_bytes = bytes_near / max(1, cst["launches_trace_nearest"])
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch of the same kernel from the committed PMC passes (rocprofv3 cannot be
        # nested inside this process): only quoted for the default workload those passes measured
        traffic = None
        valu_instr = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        default_workload = (a.tris, a.width, a.height, a.bounces, a.spp, a.bvh, world) == (1000000, 3840, 2160, 3, 1, 0, 1)
        if default_workload and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = float(tj["hbm_bytes_per_launch"])
                valu_instr = tj.get("valu_wave_instructions_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec (primary+shadow+reflect), 1M-tri 4K scene",
            "value": round(total_rays * a.steps / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if a.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[3]: synthetic {a.tris}-triangle random soup (seed 1), {fw}x{fh} ({a.width * a.height} pixels per GPU), "
                            f"bounces={a.bounces} (primary + {a.bounces - 1} reflections), 2 directional lights, "
                            f"spp={a.spp}; frame tiles 32x32 sharded over {world} GPU(s)" + ("; sorted secondary rays" if a.sort_rays else ""),
                "bvh": {0: "wide8-sah-compressed", 1: "reference-topology", 2: "wide8-lbvh-device-built"}[a.bvh],
                "rays_per_frame": {"primary": int(rays[0]), "shadow": int(rays[1]), "reflect": int(rays[2])},
                "frame_ms_device_rank0": round(acc["ms_total"] / a.steps, 4),
                "frame_ms_split_rank0": {"trace": round(acc["ms_trace"] / a.steps, 4),
                                         "trace_primary_only": round(acc["ms_trace_nearest"] / a.steps, 4),
                                         "trace_shadow_plus_next_bounce": round(acc["ms_trace_mixed"] / a.steps, 4),
                                         "trace_last_shadow_only": round(acc["ms_trace_shadow"] / a.steps, 4),
                                         "shade_resolve": round(acc["ms_shade"] / a.steps, 4),
                                         "gen_convert": round(acc["ms_other"] / a.steps, 4)},
                "bvh_nodes": cst["nodes"], "bvh_depth": cst["depth"], "scene_bytes": cst["scene_bytes"],
                "bvh_build_host_ms": round(cst["ms_build_host"], 1), "bvh_build_device_ms": round(cst["ms_build_device"], 2),
            },
            "roofline": {
                "kernel": "k_trace8" if a.bvh != 1 else "k_trace<NEAREST>", "bound": "hbm",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                # what actually binds this kernel: VALU wave-instructions per launch from the committed
                # SQ_INSTS_VALU pass over this launch duration, against one issue per SIMD per 4 clocks at 2.4 GHz
                "valu_issue": ({"achieved": round(valu_instr / (avg_ms * 1e-3) / 1e9, 1), "peak": VALU_ISSUE_PEAK_G,
                                "unit": "G wave-instructions/s", "frac": round(valu_instr / (avg_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_G, 4)}
                               if valu_instr and avg_ms > 0 else None),
                "algorithmic_bytes_per_launch": int(per_launch_bytes), "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": launches,
                "per_ray": {"node_visits": round(cst["node_visits"] / max(1, n_near), 2),
                            "box_tests": round(cst["box_tests"] / max(1, n_near), 2),
                            "tri_tests": round(cst["tri_tests"] / max(1, n_near), 2),
                            "bytes": round(bytes_near / max(1, n_near), 1)},
                "per_shadow_ray": {"node_visits": round(cst["node_visits_shadow"] / max(1, cst["rays_shadow"]), 2),
                                   "box_tests": round(cst["box_tests_shadow"] / max(1, cst["rays_shadow"]), 2),
                                   "tri_tests": round(cst["tri_tests_shadow"] / max(1, cst["rays_shadow"]), 2),
                                   "bytes": round(bytes_shadow / max(1, cst["rays_shadow"]), 1)},
            },
        }
        if world == 1:
            # the PCIe-inclusive frame (crt_render: same kernels + device->host copy of the BGR frame);
            # reported beside, never as, `value`
            ctx.render(cam, tgt, params)
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.render(cam, tgt, params)
            out["config"]["frame_ms_with_d2h_copy"] = round((time.perf_counter() - t1) / 3 * 1e3, 4)
        if a.out_bmp and world == 1:
            crt.write_bmp(a.out_bmp, ctx.render(cam, tgt, params), a.width, a.height)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prims, lights, cam, tgt, a.width, a.height, a.bounces, a.cpu_threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    _ = rays_local


if __name__ == "__main__":
    main()
