"""k_trace8's tail: once the ray queue is dry, idle lanes take stack entries (and a copy of the ray) from busy lanes of
their wave and the lanes of a ray fold their results in LDS (kernels.hip, "steal").  Every GPU test passes through it;
this one makes it the whole launch -- far fewer rays than resident lanes, so the queue is dry at once -- and runs the
most eager setting (a steal step in every iteration, from the first idle lane) in a process of its own, because the
thresholds are read once per process (CRT_STEAL)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import cudaraytracer_amd as crt
import orc

def rays(n, seed, span):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-span, span, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.ascontiguousarray(np.concatenate([o, d.astype(np.float32)], axis=1))

bad = 0
for name, prims, mode in (("soup", crt.scene_soup(60000, seed=7), 0), ("soup-device", crt.scene_soup(60000, seed=7), 2),
                          ("ties", np.concatenate([crt.scene_teapot(with_sphere=True)] * 3), 0),
                          ("spheres", crt.scene_spheres(3000, seed=2), 0)):
    ctx = crt.Context(0); ctx.set_scene(prims); ctx.build(mode)
    sc = orc.Scene(prims); sc.build()
    for n in (37, 700, 9000, 150000):
        r = rays(n, n, 150)
        ids_o, ts_o = sc.trace(r)
        ids_g, ts_g = ctx.trace(r)
        sh_g = ctx.trace_shadow(r)
        hit = ids_o >= 0
        nb = int((ids_g != ids_o).sum()) + int((ts_g[hit] != ts_o[hit]).sum()) + int((sh_g.astype(bool) != (hit & (ts_o > 1e-9))).sum())
        print(name, n, "rays:", int(hit.sum()), "hits,", nb, "mismatches", flush=True)
        bad += nb
    ctx.close(); sc.close()
sys.exit(1 if bad else 0)
"""


@pytest.mark.parametrize("setting", ["1,1", "16,4", "65,1"])
def test_shared_rays_give_the_oracle_s_hits(setting, tmp_path):
    """nearest-hit ids and t, and the directional shadow predicate (hit with the nearest t > 1e-9, Ray.cu:650-656), equal
    the oracle's for launches that are all tail, with work sharing at its most eager, at its default and off"""
    script = tmp_path / "tail.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, CRT_STEAL=setting)
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "mismatches" in p.stdout
