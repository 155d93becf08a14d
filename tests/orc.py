"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by cudaraytracer_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# ORC_VARIANT=hostpow: the restatement with pow(float,int)/sqrt in double (a host build's rounding), tests/test_oracle_hostpow.py
VARIANT = os.environ.get("ORC_VARIANT", "")
LIB = os.path.join(ORACLE_DIR, "liboracle_hostpow.so" if VARIANT == "hostpow" else "liboracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")

PRIM_DTYPE = np.dtype([
    ("type", "<i4"), ("objNum", "<i4"), ("position", "<f4", 3), ("radius", "<f4"),
    ("v1", "<f4", 3), ("v2", "<f4", 3), ("v3", "<f4", 3),
    ("v1Norm", "<f4", 3), ("v2Norm", "<f4", 3), ("v3Norm", "<f4", 3),
    ("ambient", "<f4", 3), ("diffuse", "<f4", 3), ("specular", "<f4", 3), ("reflective", "<f4", 3),
    ("shininess", "<f4"), ("direction", "<f4", 3), ("speed", "<f4"),
])
LIGHT_DTYPE = np.dtype([
    ("color", "<f4", 3), ("direction", "<f4", 3), ("point", "<i4"),
    ("constantTerm", "<f4"), ("linearTerm", "<f4"), ("quadraticTerm", "<f4"),
    ("position", "<f4", 3), ("area", "<i4"), ("radius", "<f4"),
])
ISECT_DTYPE = np.dtype([
    ("isectPoint", "<f4", 3), ("incidentDirection", "<f4", 3), ("reflectionCoef", "<f4", 3),
    ("ambient", "<f4", 3), ("diffuse", "<f4", 3), ("specular", "<f4", 3), ("reflective", "<f4", 3),
    ("shininess", "<f4"), ("normal", "<f4", 3), ("color", "<f4", 3),
])


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("traversals", "primary", "shadow", "reflect", "box_tests", "tri_tests",
                                          "sph_tests", "live_traversals")]


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "crt_oracle.cpp")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, os.path.basename(LIB)], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        vp, i = C.c_void_p, C.c_int
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, i]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_destroy.restype = None
        for name, args in {
            "orc_bbox": [vp, vp, vp], "orc_sphere": [vp, vp, vp, vp, vp, vp], "orc_triangle": [vp, vp, vp, vp, vp, vp],
            "orc_wall": [vp, vp, i, vp, vp, vp, vp], "orc_scene_size": [vp], "orc_build": [vp],
            "orc_num_nodes": [vp], "orc_num_leaves": [vp],
            "orc_tree_dump": [vp] * 9,
            "orc_render": [vp, vp, i, vp, vp, i, i, i, vp, vp, vp, vp, vp, i],
            "orc_render_area": [vp, vp, i, vp, vp, i, i, i, vp, vp, vp, vp, vp, i, i],
            "orc_render_spp": [vp, vp, i, vp, vp, i, i, i, vp, vp, vp, vp, vp, i, i, i],
            "orc_load_obj": [C.c_char_p, vp, vp, i],
        }.items():
            getattr(L, name).restype = i
            getattr(L, name).argtypes = args
        for name, args in {
            "orc_scene_get": [vp, vp], "orc_scene_set": [vp, vp], "orc_refit": [vp], "orc_animate": [vp],
            "orc_trace": [vp, vp, i, vp, vp, vp, vp, i], "orc_trace_brute": [vp, vp, i, vp, vp, i],
            "orc_primary_rays": [vp, vp, i, i, vp], "orc_stats_get": [vp, vp], "orc_stats_reset": [vp], "orc_set_all_boxes_hit": [vp, i],
            "orc_math_probe": [vp, i, vp], "orc_area_table": [vp, i], "orc_powf": [vp, vp, i, vp],
        }.items():
            getattr(L, name).restype = None
            getattr(L, name).argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f3(v):
    return np.ascontiguousarray(v, np.float32)


def bbox(box6, o, d):
    return lib().orc_bbox(_p(f3(box6)), _p(f3(o)), _p(f3(d)))


def sphere(prim, o, d):
    t = np.zeros(1, np.float32); n = np.zeros(3, np.float32); x = np.zeros(3, np.float32)
    p = np.ascontiguousarray(prim, PRIM_DTYPE).reshape(1)
    h = lib().orc_sphere(_p(p), _p(f3(o)), _p(f3(d)), _p(t), _p(n), _p(x))
    return bool(h), float(t[0]), n, x


def triangle(prim, o, d):
    t = np.zeros(1, np.float32); n = np.zeros(3, np.float32); x = np.zeros(3, np.float32)
    p = np.ascontiguousarray(prim, PRIM_DTYPE).reshape(1)
    h = lib().orc_triangle(_p(p), _p(f3(o)), _p(f3(d)), _p(t), _p(n), _p(x))
    return bool(h), float(t[0]), n, x


def wall(o, d, raytype=0, coef_in=(0, 0, 0), color_in=(0, 0, 0)):
    out = np.zeros(1, ISECT_DTYPE); rc = np.zeros(3, np.float32)
    w = lib().orc_wall(_p(f3(o)), _p(f3(d)), raytype, _p(f3(coef_in)), _p(f3(color_in)), _p(out), _p(rc))
    return w, out[0], rc


def math_probe(in9):
    in9 = np.ascontiguousarray(in9, np.float32).reshape(-1, 9)
    out = np.zeros((len(in9), 32), np.float32)
    lib().orc_math_probe(_p(in9), len(in9), _p(out))
    return out


def powf(x, y):
    """libm powf of the host, value by value (the call of Shade's specular term, Ray.cu:655)"""
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    out = np.zeros(x.shape, np.float32)
    lib().orc_powf(_p(x), _p(y), x.size, _p(out))
    return out


def load_obj(path):
    n = lib().orc_load_obj(path.encode(), None, None, 0)
    if n < 0:
        raise IOError(path)
    v = np.zeros((n, 3), np.float32); nm = np.zeros((n, 3), np.float32)
    assert lib().orc_load_obj(path.encode(), _p(v), _p(nm), n) == n
    return v, nm


def area_table(n=1024):
    out = np.zeros((n, 3), np.float32)
    lib().orc_area_table(_p(out), n)
    return out


def primary_rays(cam, target, w, h):
    out = np.zeros((w * h, 6), np.float32)
    lib().orc_primary_rays(_p(f3(cam)), _p(f3(target)), w, h, _p(out))
    return out


class Scene:
    def __init__(self, prims):
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        self.n = len(prims)
        self.h = lib().orc_scene_create(_p(prims), self.n)

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self):
        rc = lib().orc_build(self.h)
        if rc < 0:
            raise RuntimeError("oracle build: runaway recursion")

    def refit(self):
        lib().orc_refit(self.h)

    def animate(self):
        lib().orc_animate(self.h)

    def prims(self):
        out = np.zeros(self.n, PRIM_DTYPE)
        lib().orc_scene_get(self.h, _p(out))
        return out

    def set_prims(self, prims):
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        assert len(prims) == self.n
        lib().orc_scene_set(self.h, _p(prims))

    def dump(self):
        nn = lib().orc_num_nodes(self.h)
        boxes = np.zeros((nn, 6), np.float32)
        arrs = [np.zeros(nn, np.int32) for _ in range(6)]
        leaf_prims = np.zeros(self.n, np.int32)
        got = lib().orc_tree_dump(self.h, _p(boxes), *[_p(a) for a in arrs], _p(leaf_prims))
        assert got == nn
        isleaf, parent, left, right, first, count = arrs
        return dict(boxes=boxes, isleaf=isleaf, parent=parent, left=left, right=right, first=first, count=count,
                    leaf_prims=leaf_prims)

    def trace(self, rays6, nthreads=8, with_geometry=False):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = len(rays6)
        ids = np.zeros(n, np.int32); ts = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32) if with_geometry else None
        pts = np.zeros((n, 3), np.float32) if with_geometry else None
        lib().orc_trace(self.h, _p(rays6), n, _p(ids), _p(ts), _p(nrm), _p(pts), nthreads)
        return (ids, ts, nrm, pts) if with_geometry else (ids, ts)

    def all_boxes_hit(self, on=True):
        """G3 probe: the reference's traversal with every box test answered "hit" (its leaves, its order, none of the false
        negatives of its slab test)"""
        lib().orc_set_all_boxes_hit(self.h, 1 if on else 0)

    def trace_brute(self, rays6, nthreads=8):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = len(rays6)
        ids = np.zeros(n, np.int32); ts = np.zeros(n, np.float32)
        lib().orc_trace_brute(self.h, _p(rays6), n, _p(ids), _p(ts), nthreads)
        return ids, ts

    def render(self, lights, cam, target, w, h, bounces, rect=None, nthreads=8, want_color=False, want_ids=False,
               area_samples=30, spp=1):
        spp_n = int(round(spp ** 0.5))
        assert spp_n * spp_n == spp
        lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
        bgr = np.zeros((h, w, 3), np.uint8)
        col = np.zeros((h * w, 3), np.float32) if want_color else None
        ids = np.zeros((bounces, h * w), np.int32) if want_ids else None
        ts = np.zeros((bounces, h * w), np.float32) if want_ids else None
        r = np.ascontiguousarray(rect, np.int32) if rect is not None else None
        rc = lib().orc_render_spp(self.h, _p(lights), len(lights), _p(f3(cam)), _p(f3(target)), w, h, bounces, _p(r),
                                  _p(bgr), _p(col), _p(ids), _p(ts), nthreads, area_samples, spp_n)
        assert rc == 0
        return dict(bgr=bgr, color=col, ids=ids, ts=ts)

    def stats(self, reset=False):
        s = Stats()
        lib().orc_stats_get(self.h, C.byref(s))
        if reset:
            lib().orc_stats_reset(self.h)
        return {n: getattr(s, n) for n, _ in s._fields_}


def ref_driver_available():
    return os.path.exists(REF_DRIVER)


def ref_math(in9, tmpdir):
    in9 = np.ascontiguousarray(in9, np.float32).reshape(-1, 9)
    a, b = os.path.join(tmpdir, "in.bin"), os.path.join(tmpdir, "out.bin")
    in9.tofile(a)
    subprocess.check_call([REF_DRIVER, "math", a, b])
    return np.fromfile(b, np.float32).reshape(-1, 32)


def ref_obj(path, tmpdir):
    b = os.path.join(tmpdir, "obj.bin")
    subprocess.check_call([REF_DRIVER, "obj", path, b], stdout=subprocess.DEVNULL)
    raw = np.fromfile(b, np.uint8)
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    f = np.frombuffer(raw[4:], np.float32)
    return f[:3 * n].reshape(n, 3).copy(), f[3 * n:6 * n].reshape(n, 3).copy()
