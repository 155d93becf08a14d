"""The reference-named C++ API (compat wrappers: loadOBJ, newBuild/constructTree, refitTree, freeTree,
startRayTracing) driven by tools/crt_demo.cpp -- the headless replacement of main.cu's loop -- must
produce the oracle's frames: refit frames and rebuild frames alike."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "cudaraytracer_amd", "crt_demo")


def test_headless_demo_frames_match_oracle(tmp_path):
    import cudaraytracer_amd as crt
    assert os.path.exists(DEMO), "cudaraytracer_amd/crt_demo is not built (run __graft_entry__.build())"
    frames, rebuild_every = 6, 4
    prefix = str(tmp_path / "f")
    out = subprocess.run([DEMO, crt.TEAPOT_OBJ, str(frames), prefix, str(rebuild_every), "2"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ms/frame" in out.stdout                      # main.cu:529's report line
    prims = crt.scene_teapot()
    sc = orc.Scene(prims)
    sc.build()
    cam, tgt = crt.default_camera()
    lights = crt.default_lights()
    for f in range(frames):
        if f > 0:
            if f % rebuild_every == 0:
                sc.build()
            else:
                sc.refit()
        want = sc.render(lights, cam, tgt, 720, 360, 2, nthreads=16)["bgr"]
        got = np.asarray(Image.open(f"{prefix}_{f:03d}.bmp").convert("RGB"))[::-1, :, ::-1]
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32)).max(axis=2)
        assert int((diff > 1).sum()) == 0, f"frame {f}: {int((diff > 1).sum())} pixels beyond 1 LSB"
        sc.animate()
