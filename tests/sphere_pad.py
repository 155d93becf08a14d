"""The box of a sphere in the wide trees, restated for the tests that check tree dumps (test infrastructure)."""
import numpy as np


def sphere_box_half(prims):
    """layout.h sphere_box_half + bvh_build.cpp origin_bound_of restated in float32, operation by operation: the half side of
    a sphere's box in the wide trees -- the ball the float evaluation of intersectSphere's discriminant can reach"""
    f = np.float32
    tri = prims["type"] == 1
    m = f(300.0)
    if (~tri).any():
        m = max(m, (np.abs(prims["position"][~tri]) + prims["radius"][~tri][:, None]).astype(f).max())
    if tri.any():
        m = max(m, max(np.abs(prims[k][tri]).max() for k in ("v1", "v2", "v3")))
    ob = f(f(1.7320509) * f(1.001)) * f(m)
    c, r = prims["position"].astype(f), prims["radius"].astype(f)
    norm = np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]) * f(1.000001)
    e = f(1.1181e-3) * (norm + ob)
    M = np.maximum(r, e)
    rr, ee = r / M, e / M
    return (M * np.sqrt((rr * rr + ee * ee) + (f(4e-7) * rr) * rr) * f(1.000001)).astype(f)
