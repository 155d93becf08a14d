"""tools/exp/g3_probe.py -- classify GPU/oracle id mismatches on shifted scenes: is the GPU equal to the brute-force
loop over all primitives (then the oracle's BVH lost the hit to a box false-negative, SURVEY 9.1 class G3)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracer_amd as crt
import orc
from test_gpu_robustness import rays_at, transform

for scale, shift in [(1.0, (5e4, -3e4, 2e4)), (1e-2, (900.0, 900.0, 900.0)), (1.0, (0, 0, 0))]:
    prims = transform(crt.scene_teapot(with_sphere=("sphere" in sys.argv)), scale, shift)
    rng = np.random.default_rng(5)
    rays = rays_at(rng, 60000, np.asarray(shift, np.float64) + np.array([0, 20, 0]) * scale, 60 * scale)
    sc = orc.Scene(prims); sc.build()
    ids_o, ts_o = sc.trace(rays, nthreads=16)
    ids_b, ts_b = sc.trace_brute(rays, nthreads=16)
    for mode in (0, 1):
        ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(mode)
        ids_g, ts_g = ctx.trace(rays); ctx.close()
        bad = ids_g != ids_o
        print(f"scale {scale} shift {shift} mode {mode}: gpu!=oracle {bad.sum()}  gpu!=brute {(ids_g != ids_b).sum()}  oracle!=brute {(ids_o != ids_b).sum()}"
              f"  of gpu!=oracle: oracle miss {(bad & (ids_o < 0)).sum()}, gpu miss {(bad & (ids_g < 0)).sum()}, both hit {(bad & (ids_g >= 0) & (ids_o >= 0)).sum()}")
        gb = ids_g != ids_b
        if gb.any():
            k = np.nonzero(gb)[0][:5]
            print("   gpu vs brute examples: ids", ids_g[k], ids_b[k], "t", ts_g[k], ts_b[k])
