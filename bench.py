#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its own workload: Mrays/s (primary + shadow + reflection
traversals actually issued) and frame ms on the synthetic 1M-triangle soup at 3840x2160, bounces=3
(configs[3]).  On N GPUs the SAME frame is sharded by 32x32 tiles over the ranks (strong scaling: BASELINE's metric is
"1M-tri 4K scene, 1/2/4/8 GPU"); --config 4 selects configs[4] (7680x4320 x 16 spp), --weak grows the frame with N at a
fixed pixel count per GPU instead.

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
N_SIMD = 256 * 4          # 256 CUs x 4 SIMDs
SPEC_CLOCK_GHZ = 2.4      # MI355X peak engine clock
UBENCH_VALU = os.path.join(ROOT, "profiles", "r02_ubench_valu_issue.jsonl")    # tools/ubench/valu_issue.hip on MI355X
UBENCH_L1 = os.path.join(ROOT, "profiles", "r02_ubench_l1_gather.jsonl")       # tools/ubench/l1_gather.hip on MI355X
ISA_MIX = os.path.join(ROOT, "cudaraytracer_amd", "isa_mix_k_trace8.json")     # tools/isa_mix.py, written by the build
_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (the GPU harness kills a command that stays silent for minutes)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def survey_bytes(nrays, nbox, ntri, nsph):
    """SURVEY.md 8(d)'s formula, kept as a labelled secondary: 32 B per AABB test (24 B bounds + 8 B link), 36 B per
    triangle test (9 floats), 16 B per sphere test, 48 B per ray (32 B ray record + 16 B hit record).  It prices a
    layout this build does not have (a QNode8 holds a child in 10 B), so it is NOT the roofline's numerator."""
    return 32 * nbox + 36 * ntri + 16 * nsph + 48 * nrays


def layout_bytes(nrays, nvisits, nprim_tests):
    """Bytes the traversal actually requests with the layout it reads (layout.h): 80 B per node visit (QNode8), 48 B per
    primitive test (Geom), 48 B per ray (two float4 in, hit slot + t or a shadow flag out)."""
    return 80 * nvisits + 48 * nprim_tests + 48 * nrays


def valu_issue_model(cst):
    """VALU wave-instructions and VALU issue cycles of the frame's k_trace8 launches: the wave-level iteration counts of
    the counting frame (crt_stats.sched_*) times the per-region instruction counts of the compiled kernel, every opcode
    priced with its measured issue cost (tools/isa_mix.py <- tools/ubench/valu_issue.hip).  Sub-branches of a region
    that no lane takes are skipped by the hardware but counted here, so both figures are upper estimates; the
    instruction count is checked against SQ_INSTS_VALU of the committed PMC pass (roofline.model_check)."""
    mix = json.load(open(ISA_MIX))
    R = mix["regions"]

    def reg(name):
        r = R.get(name, {"valu": 0, "valu_issue_cycles": 0.0})
        return np.array([r["valu"], r["valu_issue_cycles"], r.get("valu_issue_cycles_nominal", r["valu_issue_cycles"])], np.float64)
    it, node, prim = cst["sched_iterations"], cst["sched_node_iterations"], cst["sched_prim_iterations"]
    sph = prim if cst["sph_tests"] + cst["sph_tests_shadow"] > 0 else 0
    times = {"prologue": cst["trace_waves"], "refill1": cst["sched_refill_iterations"], "between_refill1": it, "vote": it, "between_vote": it,
             "node": node, "pop": cst["sched_pop_iterations"], "between_pop": node, "between_node": node,
             "prim": prim, "decode": cst["sched_decode_iterations"], "between_decode": prim, "sphere": sph, "between_sphere": prim,
             "hit": cst["sched_hit_iterations"], "between_hit": prim, "between_prim": prim,
             "finish": cst["sched_finish_iterations"], "between_finish": it, "refill2": cst["sched_setup_iterations"], "between_refill2": it,
             # the tail instantiation of the loop (queue dry, work sharing): its iterations are in the node / primitive / finish
             # counts above and are priced with the main loop's regions; the region itself is the second copy of the code
             "tail": 0, "between_tail": 0}
    tot = np.zeros(3)
    mix["breakdown"] = {}
    for name in R:
        if name not in times:
            raise KeyError(f"isa_mix region {name} has no iteration count in bench.py")
        tot = tot + times[name] * reg(name)
        if times[name] and R[name]["valu"]:
            mix["breakdown"][name] = {"executions": int(times[name]), "valu_per_execution": R[name]["valu"],
                                      "issue_cycles_per_execution": R[name]["valu_issue_cycles"]}
    return float(tot[0]), float(tot[1]), float(tot[2]), mix


def ubench_line(path, key, value, k):
    for line in open(path):
        r = json.loads(line)
        if r.get(key, "").startswith(value) and r.get("waves_per_simd") == k:
            return r
    return None


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
    # the reference's own case beside it (configs[0]: teapot + sphere, 720x360, bounces = 1, the whole frame), best of 3
    import cudaraytracer_amd as crt
    tp = orc.Scene(crt.scene_teapot(with_sphere=True))
    tp.build()
    dt0 = dt1 = 1e30
    for _ in range(3):
        tp.stats(reset=True)
        t0 = time.perf_counter()
        tp.render(lights, cam, tgt, 720, 360, 1, nthreads=cores)
        dt0 = min(dt0, time.perf_counter() - t0)
        st0 = tp.stats(reset=True)
    for _ in range(3):
        t1 = time.perf_counter()
        tp.render(lights, cam, tgt, 720, 360, 1, nthreads=1)          # once single-threaded, once on all cores
        dt1 = min(dt1, time.perf_counter() - t1)
    tp.close()
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "cpu_model": model, "host_cpus_visible": os.cpu_count(), "timing": "best of 3 for reference_case; one pass for the bounded soup sample",
        "reference_case": {"workload": "configs[0]: teapot3.obj + sphere, 720x360, bounces=1, whole frame", "value": round(st0["traversals"] / dt0 / 1e6, 3),
                           "unit": "Mrays/s", "frame_ms": round(dt0 * 1e3, 2), "traversals": st0["traversals"], "cores": cores,
                           "single_thread": {"value": round(st0["traversals"] / dt1 / 1e6, 3), "frame_ms": round(dt1 * 1e3, 1)}},
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
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json configs index: 3 = 1M soup 3840x2160 bounces=3 (the metric's config), "
                    "4 = the same soup at 7680x4320 x 16 spp")
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: finish every frame's tile gather before the next frame starts")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="N=1: after the timed region, also measure the throughput with this many (1..3) frames in "
                    "flight on the context's frame slots (config.frames_in_flight; off by default so that a kernel trace of the default command holds "
                    "only launches that ran alone)")
    ap.add_argument("--slots", type=int, default=3, help="N>1: frames in flight per rank (crt_render_submit on frame slots 1..slots)")
    ap.add_argument("--bvh", type=int, default=0, help="0 wide SAH built on the host (default), 1 reference topology, 2 wide LBVH built on the device")
    ap.add_argument("--sort-rays", type=int, default=0, help="1: visit secondary rays in Morton order of their origin cell")
    ap.add_argument("--weak", action="store_true", help="N > 1: grow the frame with N at a fixed pixel count per GPU (weak scaling) "
                    "instead of sharding the same frame")
    ap.add_argument("--strong", action="store_true", help="(default since round 2; kept for old command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--out-bmp", default=None)
    a = ap.parse_args()
    if a.config not in (3, 4):
        sys.exit("bench.py --config: 3 or 4 (the other BASELINE configs are parity-test cases, tools/bench_configs.py times them)")
    cw, ch, cspp = (3840, 2160, 1) if a.config == 3 else (7680, 4320, 16)
    a.width = a.width or cw
    a.height = a.height or ch
    a.spp = a.spp or cspp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    dist = torch = None
    # CRT_BENCH_FORCE_DIST=1: one rank through the N > 1 code (process group, ShardedRenderer, the collectives) -- the rehearsal of
    # the RCCL calls a one-GPU box allows; the line it prints is not a bench result
    multi = world > 1 or bool(os.environ.get("CRT_BENCH_FORCE_DIST"))
    if multi:
        # torch BEFORE the library: PyTorch bundles its own HIP runtime, and a process that loads libcrt_amd.so (system
        # runtime) first and torch second ends up with two runtimes, the second of which finds no device
        import torch
        import torch.distributed as dist
    import cudaraytracer_amd as crt

    cam, tgt = crt.default_camera()
    lights = crt.default_lights()
    # N > 1 shards the SAME frame (BASELINE's metric: "1M-tri 4K scene, 1/2/4/8 GPU").  --weak: the pixels per GPU stay
    # those of the --width x --height frame and the frame grows with N (7680x2160, 7680x4320, 15360x4320 for N = 2, 4, 8)
    fw, fh = a.width, a.height
    if world > 1 and a.weak:
        k = world.bit_length() - 1 if world & (world - 1) == 0 else None
        if k is None:
            fw = a.width * world
        else:
            fw, fh = a.width << ((k + 1) // 2), a.height << (k // 2)
    params = crt.default_params(width=fw, height=fh, bounces=a.bounces, spp=a.spp, rank=rank, nranks=world,
                                sort_rays=a.sort_rays)
    prims = crt.scene_soup(a.tris, seed=1) if (rank == 0 or world == 1) else None
    log(f"scene generated: {a.tris} triangles")

    phase = {"render": 0.0, "pack": 0.0, "gather": 0.0, "unpack": 0.0}
    if multi:
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
        sr = ShardedRenderer(dist, torch, rank, world, local_rank, slots=a.slots, always_gather=(world == 1))
        sr.broadcast_scene(prims, lights, a.bvh)
        ctx = sr.ctx

        # throughput mode: the tile gather of frame k runs under the render of frame k+1 (double-buffered frames); the fence
        # completes the frame still in flight, so the timed region holds exactly K finished frames
        def step():
            sr.render(cam, tgt, params, phase, pipelined=not a.no_pipeline)

        def fence():
            sr.finish()
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

    # one untimed frame with the counting kernels: test counts and wave-level iteration counts of exactly the traversal
    # the timed kernels perform (same packed BVH, same order)
    log("scene + BVH resident; counting frame")
    ctx.set_counting(True)
    ctx.render_device(cam, tgt, params)          # this rank's tiles, alone on the context's own stream (the counters are one set per context)
    cst = ctx.stats()
    ctx.set_counting(False)
    log(f"counting frame done: device {cst['ms_total']:.2f} ms; warmup")

    for _ in range(a.warmup):
        step()
    fence()
    for k in phase:
        phase[k] = 0.0
    log("timed steps")
    t0 = time.perf_counter()
    acc = {"ms_trace": 0.0, "ms_trace_nearest": 0.0, "ms_trace_shadow": 0.0, "ms_trace_mixed": 0.0, "ms_total": 0.0, "ms_shade": 0.0,
           "ms_other": 0.0, "launches_trace_nearest": 0, "launches_trace_shadow": 0, "launches_trace_mixed": 0}
    for _ in range(a.steps):
        step()
        if not multi:
            st = ctx.stats()
            for k in acc:
                acc[k] += st[k]
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed steps done: {elapsed / a.steps * 1e3:.3f} ms/step")
    if multi:
        # N > 1 keeps several frames in flight per rank (frame slots): the intervals between per-kernel events then hold other
        # frames' kernels too.  The per-kernel figures (frame_ms_split_rank0, the roofline's launch duration) come from this
        # rank's frames rendered one at a time, after the timed region, with HIP events around every kernel
        for _ in range(a.steps):
            ctx.render_device(cam, tgt, params)
            st = ctx.stats()
            for k in acc:
                acc[k] += st[k]
        fence()

    rays = np.array([st["rays_primary"], st["rays_shadow"], st["rays_reflect"]], np.float64)
    per_rank = None
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor(rays, dtype=torch.float64, device="cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = r.cpu().numpy()
        # per-rank phase times (ms per frame): device render, tile pack, gather, unpack
        mine = torch.tensor([acc["ms_total"] / a.steps, phase["render"] / a.steps * 1e3, phase["pack"] / a.steps * 1e3,
                             phase["gather"] / a.steps * 1e3, phase["unpack"] / a.steps * 1e3], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(x), 4) for x in v.cpu().tolist()] for v in allr]
    total_rays = float(rays.sum())

    if rank == 0:
        ms_step = elapsed / a.steps * 1e3
        wide = a.bvh != 1
        # ---- roofline of the dominant kernel on rank 0's launches.  k_trace8 serves the shadow rays of a bounce and the
        # nearest-hit rays of the next in one launch, so the kernel's figure covers all its launches of a frame
        # (bounces + 1 per sample pass) and all their rays.
        n_near = cst["rays_primary"] + cst["rays_reflect"]
        n_launch_c = max(1, cst["launches_trace_nearest"] + cst["launches_trace_shadow"] + cst["launches_trace_mixed"])
        if wide:
            launches = max(1, acc["launches_trace_nearest"] + acc["launches_trace_shadow"] + acc["launches_trace_mixed"])
            avg_ms = acc["ms_trace"] / launches
        else:
            launches = max(1, acc["launches_trace_nearest"])
            avg_ms = acc["ms_trace_nearest"] / launches
            n_launch_c = max(1, cst["launches_trace_nearest"])
        avg_s = avg_ms * 1e-3
        visits = cst["node_visits"] + cst["node_visits_shadow"]
        prim_tests = cst["tri_tests"] + cst["sph_tests"] + cst["tri_tests_shadow"] + cst["sph_tests_shadow"]
        nrays_k = n_near + cst["rays_shadow"]
        lay_b = layout_bytes(nrays_k, visits, prim_tests) / n_launch_c
        sur_b = (survey_bytes(n_near, cst["box_tests"], cst["tri_tests"], cst["sph_tests"]) +
                 survey_bytes(cst["rays_shadow"], cst["box_tests_shadow"], cst["tri_tests_shadow"], cst["sph_tests_shadow"])) / n_launch_c
        # HBM-side bytes and SQ_INSTS_VALU per launch of the same kernel from the committed PMC passes (rocprofv3 cannot be
        # nested inside this process): quoted, with their source, only for the default workload those passes measured
        traffic = pmc_valu = pmc_lane_util = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        default_workload = (a.tris, fw, fh, a.bounces, a.spp, a.bvh, world) == (1000000, 3840, 2160, 3, 1, 0, 1)
        tsrc = None
        pmc_matches_build = None          # None: no committed PMC pass applies to this workload
        build_hash = None
        try:
            build_hash = json.load(open(ISA_MIX)).get("kernel_hash")
        except Exception:
            pass
        if default_workload and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the committed PMC pass counts for THIS binary only if it profiled the same k_trace8 instruction stream: the
                # build writes a hash of it (tools/isa_mix.py), tools/profile.sh stores the hash with the pass
                pmc_matches_build = bool(build_hash) and tj.get("kernel_hash") == build_hash
                tsrc = "profiles/latest_traffic.json <- " + str(tj.get("source", ""))
                if pmc_matches_build:
                    traffic = float(tj["hbm_bytes_per_launch"])
                    pmc_valu = tj.get("valu_wave_instructions_per_launch")
                    pmc_lane_util = tj.get("valu_lane_utilisation")
                else:
                    tsrc += " -- NOT USED: that pass profiled k_trace8 %s, this build is %s (re-run tools/profile.sh)" % (tj.get("kernel_hash"), build_hash)
            except Exception:
                traffic = None
        roof = {"kernel": "k_trace8" if wide else "k_trace<NEAREST>", "avg_launch_ms": round(avg_ms, 4), "launches_timed": launches,
                "pmc_matches_build": pmc_matches_build, "k_trace8_kernel_hash": build_hash}
        if wide and os.path.exists(ISA_MIX) and os.path.exists(UBENCH_VALU) and cst["sched_iterations"] > 0 and avg_s > 0:
            n_instr, n_cycles_meas, n_cycles, mix = valu_issue_model(cst)
            n_instr /= n_launch_c
            n_cycles /= n_launch_c
            n_cycles_meas /= n_launch_c
            cpi = n_cycles / max(1.0, n_instr)           # issue cycles per instruction of this kernel's mix at the NOMINAL class rates (2 / 4 / 8)
            cpi_meas = n_cycles_meas / max(1.0, n_instr)  # ... with every opcode at the cost the microbenchmark measured (loop overhead included)
            # the model counts every instruction of a region, also sub-branches no lane takes (the hardware skips those): where the
            # committed PMC pass of this binary and workload gives SQ_INSTS_VALU, the count is taken from it and the model gives the mix
            cal_src = None
            if pmc_valu:
                cal = pmc_valu
            else:
                # no PMC pass for this workload: the model count divided by the largest model/PMC ratio measured so far (the count errs low)
                cal = n_instr
                try:
                    cj = json.load(open(os.path.join(ROOT, "profiles", "valu_model_calibration.json")))
                    ratio = max(c["ratio"] for c in cj["calibrations"])
                    cal = n_instr / ratio
                    cal_src = "model count / %.4f, the largest model-to-SQ_INSTS_VALU ratio in profiles/valu_model_calibration.json (no PMC pass for this workload)" % ratio
                except Exception:
                    cal_src = "model (upper estimate: counts sub-branches no lane takes)"
            ub = ubench_line(UBENCH_VALU, "class", "k_trace8 node step", 5)
            clock = ub["clock_ghz"]
            # peak: every SIMD issuing every clock at the part's highest engine clock.  The clock a kernel sustains depends on
            # what it does (2.319 GHz under the microbenchmark's pure node-step stream, 2.33-2.35 under k_trace8 itself, whose
            # waves also wait for memory), so the spec maximum is the one figure no run can exceed
            peak = N_SIMD * SPEC_CLOCK_GHZ
            achieved = cal * cpi / avg_s / 1e9
            roof.update({
                "bound": "valu_issue", "unit": "G VALU issue-cycles/s",
                "achieved": round(achieved, 1), "peak": round(peak, 1), "frac": round(achieved / peak, 4),
                "frac_at_ubench_sustained_clock": round(achieved / (N_SIMD * clock), 4),
                "frac_with_measured_opcode_costs": round(cal * cpi_meas / avg_s / 1e9 / peak, 4),
                "definition": "achieved = VALU wave-instructions of one launch x issue cycles per instruction of the kernel's mix / launch duration "
                              "measured in this run (HIP events on the kernel's stream).  Mix: wave-level iteration counts of this run's counting frame x "
                              "per-region instruction counts of the compiled kernel, each opcode at the nominal issue cost of its rate class (2 / 4 / 8 cycles per "
                              "wave64 instruction per SIMD; the class is the one the microbenchmark measured: full-rate %.2f, half-rate %.2f, v_rcp/v_sqrt %.2f "
                              "cycles including its loop overhead -- frac_with_measured_opcode_costs prices with those), so the sum is a lower bound of the "
                              "cycles the stream needs.  peak = %d SIMDs x %.1f GHz (MI355X peak engine clock, MI355X_MICROARCH.md); frac_at_ubench_sustained_clock uses the "
                              "%.3f GHz the chip holds under the node step's instruction stream alone at 5 waves/SIMD"
                              % (mix["cycles_full_rate"], mix["cycles_half_rate"], mix["cycles_quarter_rate"], N_SIMD, SPEC_CLOCK_GHZ, clock),
                "peak_source": "profiles/r02_ubench_valu_issue.jsonl (tools/ubench/valu_issue.hip on MI355X); mix: cudaraytracer_amd/isa_mix_k_trace8.json (tools/isa_mix.py)",
                "valu_wave_instructions_per_launch": int(cal),
                "valu_wave_instructions_source": (tsrc + " (SQ_INSTS_VALU)") if pmc_valu else cal_src,
                "valu_wave_instructions_per_launch_model": int(n_instr),
                "model_over_pmc": round(n_instr / pmc_valu, 4) if pmc_valu else None,
                "valu_g_wave_instructions_per_s": round(cal / avg_s / 1e9, 1),
                "node_step_stream_ceiling_g_wave_instructions_per_s": ub["g_wave_instructions_per_s"],
                "avg_issue_cycles_per_instruction": round(cpi, 3), "avg_issue_cycles_per_instruction_measured_costs": round(cpi_meas, 3),
                "regions_per_frame": mix["breakdown"],
            })
        else:
            roof.update({"bound": "hbm", "unit": "GB/s", "achieved": round(lay_b / avg_s / 1e9, 2) if avg_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                         "frac": round(lay_b / avg_s / 1e9 / HBM_PEAK_GBS, 5) if avg_s > 0 else 0.0})
        # secondary figures, labelled: the bytes this layout requests (served by L1/L2/Infinity Cache: the 24 MB of qnodes and
        # 48 MB of geom fit the 256 MB Infinity Cache, so HBM is not this kernel's roof), SURVEY 8(d)'s formula, the vector L1
        l1 = ubench_line(UBENCH_L1, "pattern", "lane80", 5) if os.path.exists(UBENCH_L1) else None
        lane_req = (5 * (cst["sched_node_lanes"]) + 3 * cst["sched_prim_lanes"] + 2 * nrays_k) / n_launch_c
        roof["hbm_nominal"] = {"bytes_per_launch_layout": int(lay_b), "achieved_gbs": round(lay_b / avg_s / 1e9, 1) if avg_s > 0 else 0.0,
                               "peak_gbs": HBM_PEAK_GBS, "ratio": round(lay_b / avg_s / 1e9 / HBM_PEAK_GBS, 4) if avg_s > 0 else 0.0,
                               "bytes": "80 B per node visit (QNode8) + 48 B per primitive test (Geom) + 48 B per ray",
                               "note": "requested bytes, cache-served (scene working set 72 MB < 256 MB Infinity Cache): not a bound, may exceed 1",
                               "bytes_per_launch_survey_8d_formula": int(sur_b)}
        roof["traffic"] = traffic
        roof["traffic_source"] = tsrc
        # SURVEY 8(d) / north_star: the fraction of the HBM-bandwidth roofline, from the counters (not from requested bytes)
        roof["hbm_counter_frac"] = round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_s > 0 else None
        # of the issue cycles in `frac`, the share that did work on a lane holding a ray (VALU lane utilisation of the same PMC pass)
        roof["useful_lane_frac"] = round(roof["frac"] * pmc_lane_util, 4) if pmc_lane_util and roof.get("bound") == "valu_issue" else None
        roof["valu_lane_utilisation_pmc"] = round(pmc_lane_util, 4) if pmc_lane_util else None
        if l1 and avg_s > 0 and lane_req > 0:
            roof["vector_l1"] = {"lane_requests_per_launch": int(lane_req), "achieved_g_lane_requests_per_s": round(lane_req / avg_s / 1e9, 1),
                                 "ceiling_all_l1_hits": l1["g_lane_requests_per_s"],
                                 "frac_of_l1_hit_ceiling": round(lane_req / avg_s / 1e9 / l1["g_lane_requests_per_s"], 4),
                                 "source": "profiles/r02_ubench_l1_gather.jsonl: lane-divergent 16-byte loads of 80-byte records, L1-resident table, 5 waves/SIMD"}
        roof["per_ray"] = {"node_visits": round(cst["node_visits"] / max(1, n_near), 2), "box_tests": round(cst["box_tests"] / max(1, n_near), 2),
                           "tri_tests": round(cst["tri_tests"] / max(1, n_near), 2)}
        roof["per_shadow_ray"] = {"node_visits": round(cst["node_visits_shadow"] / max(1, cst["rays_shadow"]), 2),
                                  "box_tests": round(cst["box_tests_shadow"] / max(1, cst["rays_shadow"]), 2),
                                  "tri_tests": round(cst["tri_tests_shadow"] / max(1, cst["rays_shadow"]), 2)}
        if cst["sched_iterations"] > 0:
            roof["schedule"] = {"node_iterations": cst["sched_node_iterations"], "lanes_per_node_iteration": round(cst["sched_node_lanes"] / max(1, cst["sched_node_iterations"]), 1),
                                "prim_iterations": cst["sched_prim_iterations"], "lanes_per_prim_iteration": round(cst["sched_prim_lanes"] / max(1, cst["sched_prim_iterations"]), 1),
                                "visits_no_child_hit_frac": round(cst["visits_no_child_hit"] / max(1, visits), 3),
                                "visits_no_child_hit_because_of_t_limit_frac": round(cst["visits_no_child_hit_tlimit"] / max(1, visits), 3),
                                "child_boxes_rejected_by_t_limit_frac": round(cst["boxes_rejected_by_tlimit"] / max(1, cst["box_tests"] + cst["box_tests_shadow"]), 3)}
        scaling = "weak" if a.weak else "strong"      # N > 1 shards the same frame unless --weak
        out = {
            "metric": "Mrays/sec (primary+shadow+reflect), 1M-tri 4K scene" if a.config == 3 and not a.weak else
                      f"Mrays/sec (primary+shadow+reflect), 1M-tri {fw}x{fh} x{a.spp}spp scene",
            "value": round(total_rays * a.steps / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[{a.config}]: synthetic {a.tris}-triangle random soup (seed 1), {fw}x{fh}, "
                            f"bounces={a.bounces} (primary + {a.bounces - 1} reflections), 2 directional lights, "
                            f"spp={a.spp}; frame tiles 32x32 sharded over {world} GPU(s)" + (" (weak scaling: the frame grows with N)" if a.weak and world > 1 else "")
                            + ("; sorted secondary rays" if a.sort_rays else ""),
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
            "roofline": roof,
        }
        if multi and world == 1:
            out["config"]["rehearsal"] = "CRT_BENCH_FORCE_DIST: one rank through the N > 1 code path (backend %s); not a bench result" % dist.get_backend()
        if per_rank is not None:
            out["config"]["per_rank_ms"] = {"columns": ["device_render_one_frame_alone", "host_wait_for_render", "host_submit_and_pack_launch", "gather_issue", "gather_wait_and_unpack"],
                                            "ranks": per_rank}
            out["config"]["frames_in_flight_per_rank"] = 1 if a.no_pipeline else sr.K
            out["config"]["per_kernel_times"] = "this rank's frames rendered one at a time after the timed region (frames in flight overlap each other's kernels)"
        if not multi:
            # the PCIe-inclusive frame (crt_render: same kernels + device->host copy of the BGR frame);
            # reported beside, never as, `value`
            ctx.render(cam, tgt, params)
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.render(cam, tgt, params)
            out["config"]["frame_ms_with_d2h_copy"] = round((time.perf_counter() - t1) / 3 * 1e3, 4)
        if not multi and a.frames_in_flight:
            # throughput with several frames in flight on the context's frame slots (crt_render_submit / crt_render_wait): the
            # launches of one frame run in the tails of the other frames' persistent launches.  Beside `value`, which keeps
            # one frame complete before the next starts (its per-kernel times are what the roofline prices)
            K = max(1, min(3, a.frames_in_flight))
            q = crt.api.Params.from_buffer_copy(params)
            q.timing = 1
            busy = [False] * K

            def run(n):
                for i in range(n):
                    k = 1 + i % K
                    if busy[k - 1]:
                        ctx.render_wait(k)
                    ctx.render_submit(k, cam, tgt, q)
                    busy[k - 1] = True
                for k in range(1, K + 1):
                    if busy[k - 1]:
                        ctx.render_wait(k)
                        busy[k - 1] = False
            run(2 * K)
            t1 = time.perf_counter()
            run(4 * K)
            ms_pipe = (time.perf_counter() - t1) / (4 * K) * 1e3
            out["config"]["frames_in_flight"] = {"slots": K, "ms_per_frame": round(ms_pipe, 4), "mrays_per_s": round(total_rays / ms_pipe / 1e3, 1),
                                                 "note": "same frames, same kernels, frames in flight on frame slots 1..K; not `value`"}
        if a.out_bmp and not multi:
            crt.write_bmp(a.out_bmp, ctx.render(cam, tgt, params), fw, fh)
        if not multi and a.bvh != 2:
            # the same scene rebuilt ON THE DEVICE (lbvh.hip + refrank.hip: wide LBVH with the reference's
            # tie-break ranks), after the timed region: its build time, and the frame time on that tree
            c2 = crt.Context(local_rank)
            try:
                c2.set_scene(prims); c2.set_lights(lights)
                c2.build(2)                               # first build allocates
                bms = []
                for _ in range(3):
                    c2.build(2); bms.append(c2.stats()["ms_build_device"])
                c2.render_device(cam, tgt, params)
                t1 = time.perf_counter()
                for _ in range(3):
                    c2.render_device(cam, tgt, params)
                c2.sync()
                fms = (time.perf_counter() - t1) / 3 * 1e3
                out["config"]["bvh_build_device_ms"] = round(min(bms), 3)
                out["config"]["device_built_tree"] = {"build_ms": round(min(bms), 3), "build_ms_runs": [round(b, 3) for b in bms],
                                                      "frame_ms": round(fms, 4), "mrays_per_s": round(total_rays / fms / 1e3, 1),
                                                      "nodes": c2.stats()["nodes"], "depth": c2.stats()["depth"],
                                                      "tie_break": "reference leaf order, replayed on the device"}
            finally:
                c2.close()
        if not multi and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prims, lights, cam, tgt, fw, fh, a.bounces, a.cpu_threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
