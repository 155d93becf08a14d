"""dev experiment: wave-scheduling statistics of k_trace8 on the 4K bench frame (counting build)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(int(os.environ.get("TRIS", 1_000_000)), seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.set_lights(crt.default_lights()); ctx.build(int(os.environ.get("BVH", 0)))
cam, tgt = crt.default_camera()
p = crt.default_params(width=3840, height=2160, bounces=3)
ctx.render_device(cam, tgt, p)
ms = []
for _ in range(5):
    ctx.render_device(cam, tgt, p); ms.append(ctx.stats()["ms_total"])
st0 = ctx.stats()
ctx.set_counting(True); ctx.render_device(cam, tgt, p); st = ctx.stats(); ctx.set_counting(False)
rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
print(f"frame {np.mean(ms):.3f} ms (min {np.min(ms):.3f}) = {rays / np.min(ms) / 1e3:.1f} Mrays/s; trace {st0['ms_trace']:.3f} shade {st0['ms_shade']:.3f} other {st0['ms_other']:.3f}")
it = st["sched_iterations"]
print(f"iterations {it/1e6:.2f}M: node {st['sched_node_iterations']/1e6:.2f}M x {st['sched_node_lanes']/max(1,st['sched_node_iterations']):.1f} lanes, prim {st['sched_prim_iterations']/1e6:.2f}M x {st['sched_prim_lanes']/max(1,st['sched_prim_iterations']):.1f} lanes, "
      f"active lanes/iteration {st['sched_active_lanes']/max(1,st['sched_node_iterations']+st['sched_prim_iterations']):.1f}; refill {st['sched_refill_iterations']/1e6:.2f}M finish {st['sched_finish_iterations']/1e6:.2f}M setup {st['sched_setup_iterations']/1e6:.2f}M "
      f"pop {st['sched_pop_iterations']/1e6:.2f}M decode {st['sched_decode_iterations']/1e6:.2f}M hit {st['sched_hit_iterations']/1e6:.2f}M")
print(f"per ray: visits {(st['node_visits']+st['node_visits_shadow'])/rays:.2f}, tri tests {(st['tri_tests']+st['tri_tests_shadow'])/rays:.2f}; no-child-hit {st['visits_no_child_hit']/(st['node_visits']+st['node_visits_shadow']):.3f}")
print({k: st[k] for k in ("node_visits", "node_visits_shadow", "tri_tests", "tri_tests_shadow", "rays_primary", "rays_shadow", "rays_reflect")})
