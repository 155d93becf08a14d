#!/usr/bin/env python3
"""Generates the committed golden vectors from the REFERENCE'S OWN compiled sources.

Run in the build container (where /root/reference exists) after `make -C oracle`:
    python tests/golden/make_golden.py
Uses oracle/_ref/ref_driver = our driver + the reference's objloader.cpp + its vendored GLM 0.9.7.6,
compiled where they lie (oracle/Makefile).  Writes
    tests/golden/glm_probe.npz   inputs (n x 9) and the vendored GLM's outputs (n x 32) for dot, cross,
                                 normalize, reflect, vec/scalar, clamp, min/max, rotate*vec4, pow
    tests/golden/teapot_obj.npz  loadOBJ(teapot3.obj): un-indexed corner positions and normals
The reference's Ray.cu / bvh.cu cannot be compiled here (CUDA), so no vectors exist for them.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402


def probe_inputs():
    rng = np.random.default_rng(20260101)
    a = rng.normal(size=(4000, 9)).astype(np.float32) * np.float32(50)
    b = rng.uniform(-1, 1, size=(2000, 9)).astype(np.float32)
    special = np.array([
        [1, 0, 0, 0, 1, 0, 0, 0, 1], [0, 0, 0, 1, 2, 3, 4, 5, 6], [1e-20, 1e-20, 1e-20, 1, 1, 1, 2, 2, 2],
        [3e18, 3e18, 3e18, 2, 2, 2, 1, 1, 1], [-0.0, 0.0, -0.0, 0.5, -0.5, 0.25, 64, 2, 0],
        [0, 50, 250, 0, 0, 0, 0, 1, 0], [2, -1, -1, 0, 1, 0, 64, 0, 0], [-2, -1, 1, 0, 0, 1, 64, 0, 0],
        [1.5, 2.5, 300.0, 255, 1, 1, 0.5, 1, 1], [-3, 1e9, 0.999, -1, 1, 1, 2, 1, 1],
    ], np.float32)
    return np.concatenate([a, b, special])


def main():
    if not orc.ref_driver_available():
        sys.exit("oracle/_ref/ref_driver is missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        x = probe_inputs()
        y = orc.ref_math(x, td)
        np.savez_compressed(os.path.join(HERE, "glm_probe.npz"), inputs=x, outputs=y)
        v, n = orc.ref_obj("/root/reference/teapot3.obj", td)
        np.savez_compressed(os.path.join(HERE, "teapot_obj.npz"), verts=v, normals=n)
    print("wrote glm_probe.npz", x.shape, y.shape, "teapot_obj.npz", v.shape)


if __name__ == "__main__":
    main()
