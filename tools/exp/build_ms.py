import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import cudaraytracer_amd as crt
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(2)
ms = []
for _ in range(6):
    ctx.build(2); ms.append(ctx.stats()["ms_build_device"])
print("device build ms:", [round(x, 3) for x in ms])
small = crt.scene_spheres(4096, seed=1)
c2 = crt.Context(0); c2.set_scene(small); c2.build(2)
ms = []
for _ in range(6):
    c2.build(2); ms.append(c2.stats()["ms_build_device"])
print("4096 spheres build ms:", [round(x, 3) for x in ms])
