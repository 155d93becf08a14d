"""dev experiment: the C++ multi-GPU layer (crt_mgpu_*) as 8 logical ranks on ONE GPU: frames completed one at a time
(crt_mgpu_render_device) against three frames in flight per rank (crt_mgpu_submit / crt_mgpu_finish).  On one GPU the ranks
share the chip, so the figure of interest is the ratio, and that frames keep flowing with one host thread."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prims = crt.scene_soup(1_000_000, seed=1)
cam, tgt = crt.default_camera()
m = crt.MultiGpu([0] * R)
m.set_scene(prims); m.root.build(0); m.set_lights(crt.default_lights()); m.sync_scene()
p = crt.default_params(width=3840, height=2160, bounces=3)
m.render_device(cam, tgt, p)
t0 = time.perf_counter()
for _ in range(8): m.render_device(cam, tgt, p)
sync_ms = (time.perf_counter() - t0) / 8 * 1e3
for _ in range(6): m.submit(cam, tgt, p)
m.finish()
t0 = time.perf_counter()
for _ in range(24): m.submit(cam, tgt, p)
m.finish()
pipe_ms = (time.perf_counter() - t0) / 24 * 1e3
t0 = time.perf_counter()
for f in range(1, 25):
    m.animate(rebuild=False); m.submit(cam, tgt, p)
m.finish()
anim_ms = (time.perf_counter() - t0) / 24 * 1e3
print(json.dumps({"logical_ranks_on_one_gpu": R, "frame": "1M-triangle soup 3840x2160 bounces=3", "ms_per_frame_one_at_a_time": round(sync_ms, 3),
                  "ms_per_frame_three_in_flight": round(pipe_ms, 3), "ms_per_frame_three_in_flight_with_animation_and_refit_on_every_rank": round(anim_ms, 3),
                  "times_last_sync_frame": m.times()}), flush=True)
m.close()
