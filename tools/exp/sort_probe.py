"""dev experiment: how much does origin-sorted ray order help k_trace8 on the 1M soup?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudaraytracer_amd as crt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
prims = crt.scene_soup(1_000_000, seed=1)
ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(0)
rng = np.random.default_rng(1)
o = np.stack([rng.uniform(-180, 180, n), rng.uniform(-180, 180, n), rng.uniform(-280, 200, n)], 1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.ascontiguousarray(np.concatenate([o, d], 1))
def morton(o, bits):
    lo = np.array([-180, -180, -280], np.float32); ext = np.array([360, 360, 480], np.float32)
    q = np.clip(((o - lo) / ext * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(len(o), np.int64)
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return key
def run(r, tag):
    ctx.trace(r); ids, ts = ctx.trace(r)
    print(f"{tag}: {ctx.stats()['ms_trace']:.3f} ms for {len(r)} rays = {len(r)/ctx.stats()['ms_trace']/1e3:.0f} Mrays/s hits={int((ids>=0).sum())}", flush=True)
    return ids, ts
i0, t0 = run(rays, "unsorted")
for bits in (3, 4, 5, 6):
    key = morton(o, bits)
    order = np.argsort(key, kind="stable")
    i1, t1 = run(np.ascontiguousarray(rays[order]), f"sorted by {bits}-bit/axis morton cell")
    assert np.array_equal(i1, i0[order]) and np.array_equal(t1, t0[order])
# + direction octant as the low key bits
key = morton(o, 5) * 8 + ((d[:, 0] < 0) * 4 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0))
order = np.argsort(key, kind="stable")
run(np.ascontiguousarray(rays[order]), "sorted by 5-bit cell + octant")
