"""Device build of the 1M-triangle soup a few times; prints ms per build.  For rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import cudaraytracer_amd as crt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
prims = crt.scene_soup(n, seed=1)
ctx = crt.Context(0)
ctx.set_scene(prims)
for it in range(4):
    t0 = time.perf_counter()
    ctx.build(2)
    print(f"build {it}: {1e3 * (time.perf_counter() - t0):.2f} ms wall, {ctx.stats()['ms_build_device']:.2f} ms reported", flush=True)
ctx.close()
