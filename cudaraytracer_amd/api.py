"""ctypes binding of include/crt.h.  Mirrors the C ABI one to one; numpy arrays in, numpy arrays out.

No compute happens here and there is no fallback path: a missing ``libcrt_amd.so`` is an error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
lib_path = os.environ.get("CRT_LIB") or os.path.join(_HERE, "libcrt_amd.so")   # CRT_LIB: A/B builds in experiments
# the mesh BASELINE's configs name; kept once in the tree, beside the other golden data (CRT_TEAPOT_OBJ points elsewhere)
TEAPOT_OBJ = os.environ.get("CRT_TEAPOT_OBJ") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "teapot3.obj")

if not os.path.exists(lib_path):
    raise ImportError(
        f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
lib = C.CDLL(lib_path)

# POD mirrors of crt_prim / crt_light (include/crt.h), i.e. of SceneObject / Light
# (reference SceneObjects.hpp:16-52).
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
assert PRIM_DTYPE.itemsize == 164 and LIGHT_DTYPE.itemsize == 60


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("target", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("spp", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
                ("sort_rays", C.c_int32), ("area_samples", C.c_int32), ("use_graph", C.c_int32),
                ("tile_sub", C.c_int32), ("tile_nsub", C.c_int32), ("grid_div", C.c_int32), ("timing", C.c_int32),
                ("zero_paths", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("sph_tests", C.c_uint64),
                ("box_tests_shadow", C.c_uint64), ("tri_tests_shadow", C.c_uint64), ("sph_tests_shadow", C.c_uint64),
                ("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double), ("ms_other", C.c_double),
                ("ms_trace_nearest", C.c_double), ("ms_trace_shadow", C.c_double),
                ("launches_trace_nearest", C.c_int32), ("launches_trace_shadow", C.c_int32),
                ("nodes", C.c_int32), ("prims", C.c_int32), ("depth", C.c_int32), ("leaves", C.c_int32),
                ("ms_build_host", C.c_double), ("ms_refit", C.c_double), ("scene_bytes", C.c_uint64),
                ("node_visits", C.c_uint64), ("node_visits_shadow", C.c_uint64), ("ms_build_device", C.c_double),
                ("ms_trace_mixed", C.c_double), ("launches_trace_mixed", C.c_int32), ("pad_", C.c_int32),
                ("sched_iterations", C.c_uint64), ("sched_node_iterations", C.c_uint64), ("sched_node_lanes", C.c_uint64),
                ("sched_prim_iterations", C.c_uint64), ("sched_prim_lanes", C.c_uint64), ("sched_active_lanes", C.c_uint64),
                ("sched_refill_iterations", C.c_uint64), ("sched_finish_iterations", C.c_uint64),
                ("sched_setup_iterations", C.c_uint64),
                ("sched_pop_iterations", C.c_uint64), ("sched_decode_iterations", C.c_uint64), ("sched_hit_iterations", C.c_uint64),
                ("visits_no_child_hit", C.c_uint64), ("visits_no_child_hit_tlimit", C.c_uint64), ("boxes_rejected_by_tlimit", C.c_uint64),
                ("trace_waves", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


MAX_FRAME_SLOTS = 4      # CRT_MAX_FRAME_SLOTS

# every symbol include/crt.h declares; tests/test_abi.py checks the library exports all of them
ABI_SYMBOLS = [
    "crt_create", "crt_destroy", "crt_last_error", "crt_params_default", "crt_set_counting",
    "crt_scene_set", "crt_scene_update", "crt_lights_set", "crt_animate_device", "crt_scene_get",
    "crt_bvh_build", "crt_bvh_build_async", "crt_bvh_swap", "crt_bvh_refit", "crt_bvh_free", "crt_bvh_reference_order", "crt_bvh_dump",
    "crt_render", "crt_render_device", "crt_render_submit", "crt_render_wait", "crt_set_capture", "crt_get_capture", "crt_get_color",
    "crt_trace", "crt_trace_shadow", "crt_trace_noalloc", "crt_powf_probe",
    "crt_blob_size", "crt_blob_device_ptr", "crt_blob_import_device",
    "crt_tile_owner", "crt_tiles_owned", "crt_tiles_pack", "crt_tiles_unpack", "crt_tiles_pack_slot", "crt_slot_stream",
    "crt_get_stats", "crt_sync",
    "crt_mgpu_create", "crt_mgpu_destroy", "crt_mgpu_size", "crt_mgpu_uses_rccl", "crt_mgpu_ctx", "crt_mgpu_last_error",
    "crt_mgpu_lights_set", "crt_mgpu_sync_scene", "crt_mgpu_render", "crt_mgpu_render_device", "crt_mgpu_get_times",
    "crt_mgpu_scene_set", "crt_mgpu_animate", "crt_mgpu_submit", "crt_mgpu_finish",
    "crt_load_obj", "crt_scene_teapot", "crt_scene_spheres", "crt_scene_soup", "crt_lights_default", "crt_area_table",
    "crt_camera_default", "crt_animate_host", "crt_write_bmp", "crt_host_bvh",
]

_vp, _i, _fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
lib.crt_create.restype = _vp
lib.crt_create.argtypes = [_i, C.POINTER(_i)]
lib.crt_destroy.restype = None
lib.crt_destroy.argtypes = [_vp]
lib.crt_last_error.restype = C.c_char_p
lib.crt_last_error.argtypes = [_vp]
lib.crt_params_default.restype = None
lib.crt_params_default.argtypes = [C.POINTER(Params)]
lib.crt_camera_default.restype = None
lib.crt_camera_default.argtypes = [C.POINTER(Camera)]
for _name, _args in {
    "crt_set_counting": [_vp, _i], "crt_set_capture": [_vp, _i],
    "crt_scene_set": [_vp, _vp, _i], "crt_scene_update": [_vp, _vp, _i], "crt_lights_set": [_vp, _vp, _i],
    "crt_animate_device": [_vp], "crt_scene_get": [_vp, _vp, _i],
    "crt_bvh_build": [_vp, _i], "crt_bvh_refit": [_vp, _i], "crt_bvh_free": [_vp],
    "crt_bvh_build_async": [_vp], "crt_bvh_swap": [_vp, _i, C.POINTER(_i)],
    "crt_bvh_reference_order": [_vp, _vp, _i], "crt_bvh_dump": [_vp, _vp, _vp, _vp, _vp, _i],
    "crt_render": [_vp, C.POINTER(Camera), C.POINTER(Params), _vp],
    "crt_render_device": [_vp, C.POINTER(Camera), C.POINTER(Params), _vp],
    "crt_render_submit": [_vp, _i, C.POINTER(Camera), C.POINTER(Params), _vp], "crt_render_wait": [_vp, _i],
    "crt_tiles_pack_slot": [_vp, _i, C.POINTER(Params), _vp, _vp], "crt_slot_stream": [_vp, _i, C.POINTER(_vp)],
    "crt_get_capture": [_vp, _i, _vp, _vp, _i], "crt_get_color": [_vp, _vp, _i],
    "crt_trace": [_vp, _vp, _i, _vp, _vp], "crt_trace_shadow": [_vp, _vp, _i, _vp], "crt_powf_probe": [_vp, _vp, _vp, _i, _vp],
    "crt_trace_noalloc": [_vp, _vp, _i, _vp, _vp],
    "crt_blob_size": [_vp, C.POINTER(C.c_size_t)], "crt_blob_device_ptr": [_vp, C.POINTER(_vp)],
    "crt_blob_import_device": [_vp, _vp, C.c_size_t],
    "crt_tile_owner": [_i, _i], "crt_tiles_owned": [C.POINTER(Params), _vp, _vp, _vp, _i],
    "crt_tiles_pack": [_vp, C.POINTER(Params), _vp, _vp], "crt_tiles_unpack": [_vp, C.POINTER(Params), _i, _vp, _vp],
    "crt_get_stats": [_vp, C.POINTER(Stats)], "crt_sync": [_vp],
    "crt_load_obj": [C.c_char_p, _vp, _vp, _i], "crt_scene_teapot": [C.c_char_p, _i, _vp, _i],
    "crt_scene_spheres": [_i, C.c_uint32, _vp], "crt_scene_soup": [_i, C.c_uint32, _vp],
    "crt_host_bvh": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "crt_lights_default": [_vp], "crt_area_table": [_vp, _i], "crt_animate_host": [_vp, _i], "crt_write_bmp": [C.c_char_p, _vp, _i, _i],
}.items():
    getattr(lib, _name).restype = _i
    getattr(lib, _name).argtypes = _args


lib.crt_mgpu_create.restype = _vp
lib.crt_mgpu_create.argtypes = [_vp, _i, _i, C.POINTER(_i)]
lib.crt_mgpu_destroy.restype = None
lib.crt_mgpu_destroy.argtypes = [_vp]
lib.crt_mgpu_ctx.restype = _vp
lib.crt_mgpu_ctx.argtypes = [_vp, _i]
lib.crt_mgpu_last_error.restype = C.c_char_p
lib.crt_mgpu_last_error.argtypes = [_vp]
for _name, _args in {
    "crt_mgpu_size": [_vp], "crt_mgpu_uses_rccl": [_vp], "crt_mgpu_lights_set": [_vp, _vp, _i], "crt_mgpu_sync_scene": [_vp],
    "crt_mgpu_render": [_vp, C.POINTER(Camera), C.POINTER(Params), _vp],
    "crt_mgpu_render_device": [_vp, C.POINTER(Camera), C.POINTER(Params), C.POINTER(_vp)],
    "crt_mgpu_get_times": [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "crt_mgpu_scene_set": [_vp, _vp, _i], "crt_mgpu_animate": [_vp, _i, _i],
    "crt_mgpu_submit": [_vp, C.POINTER(Camera), C.POINTER(Params), C.POINTER(_vp)], "crt_mgpu_finish": [_vp, C.POINTER(_vp)],
}.items():
    getattr(lib, _name).restype = _i
    getattr(lib, _name).argtypes = _args


class CrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"crt error {code}: {msg}")
        self.code = code


def _ptr(a):
    return a.ctypes.data_as(_vp)


# ---------------------------------------------------------------- host-side scene layer
def load_obj(path):
    n = lib.crt_load_obj(path.encode(), None, None, 0)
    if n < 0:
        raise CrtError(n, f"cannot load {path}")
    v = np.zeros((n, 3), np.float32)
    nm = np.zeros((n, 3), np.float32)
    rc = lib.crt_load_obj(path.encode(), _ptr(v), _ptr(nm), n)
    if rc < 0:
        raise CrtError(rc, f"cannot load {path}")
    return v, nm


def scene_teapot(obj_path=None, with_sphere=True):
    obj_path = obj_path or TEAPOT_OBJ
    n = lib.crt_scene_teapot(obj_path.encode(), int(with_sphere), None, 0)
    if n < 0:
        raise CrtError(n, f"cannot load {obj_path}")
    out = np.zeros(n, PRIM_DTYPE)
    rc = lib.crt_scene_teapot(obj_path.encode(), int(with_sphere), _ptr(out), n)
    if rc < 0:
        raise CrtError(rc, "crt_scene_teapot")
    return out


def scene_spheres(n, seed=1):
    out = np.zeros(n, PRIM_DTYPE)
    rc = lib.crt_scene_spheres(n, seed, _ptr(out))
    if rc < 0:
        raise CrtError(rc, "crt_scene_spheres")
    return out


def scene_soup(ntris, seed=1):
    out = np.zeros(ntris, PRIM_DTYPE)
    rc = lib.crt_scene_soup(ntris, seed, _ptr(out))
    if rc < 0:
        raise CrtError(rc, "crt_scene_soup")
    return out


def default_lights():
    out = np.zeros(2, LIGHT_DTYPE)
    lib.crt_lights_default(_ptr(out))
    return out


def area_table(n=1024):
    """The unit-disk sample table of the area lights (include/crt.h: crt_area_table): n x (u, cos, sin)."""
    out = np.zeros((n, 3), np.float32)
    rc = lib.crt_area_table(_ptr(out), n)
    if rc < 0:
        raise CrtError(rc, "crt_area_table")
    return out


def default_camera():
    c = Camera()
    lib.crt_camera_default(C.byref(c))
    return np.array(c.position, np.float32), np.array(c.target, np.float32)


def default_params(**kw):
    p = Params()
    lib.crt_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"crt_params has no field {k}")
        setattr(p, k, v)
    return p


def animate_host(prims):
    assert prims.dtype == PRIM_DTYPE and prims.flags.c_contiguous
    rc = lib.crt_animate_host(_ptr(prims), len(prims))
    if rc < 0:
        raise CrtError(rc, "crt_animate_host")
    return prims


def host_bvh(prims, mode=0):
    """Host-only build (no GPU): dict with root box, per-node child boxes/links, parents, slot and rank maps."""
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    n = len(prims)
    nn, depth = C.c_int32(0), C.c_int32(0)
    rc = lib.crt_host_bvh(_ptr(prims), n, mode, C.byref(nn), C.byref(depth), None, None, None, None, None, None, 0)
    if rc < 0:
        raise CrtError(rc, "crt_host_bvh")
    root = np.zeros(6, np.float32)
    boxes = np.zeros((nn.value, 8, 6), np.float32)
    links = np.zeros((nn.value, 8), np.uint32)
    parents = np.zeros((nn.value, 2), np.uint32)
    s2s = np.zeros(n, np.uint32)
    rank = np.zeros(n, np.uint32)
    rc = lib.crt_host_bvh(_ptr(prims), n, mode, C.byref(nn), C.byref(depth), _ptr(root), _ptr(boxes), _ptr(links),
                          _ptr(parents), _ptr(s2s), _ptr(rank), nn.value)
    if rc < 0:
        raise CrtError(rc, "crt_host_bvh")
    return dict(root=root, boxes=boxes, links=links, parents=parents, slot_to_scene=s2s, rank_of_scene=rank,
                depth=depth.value)


def tile_owner(t, nranks):
    return lib.crt_tile_owner(t, nranks)


def tiles_owned(params):
    nt = C.c_int32(0)
    npx = C.c_int64(0)
    rc = lib.crt_tiles_owned(C.byref(params), C.byref(nt), C.byref(npx), None, 0)
    if rc < 0:
        raise CrtError(rc, "crt_tiles_owned: bad parameters")
    ids = np.zeros(nt.value, np.int32)
    lib.crt_tiles_owned(C.byref(params), C.byref(nt), C.byref(npx), _ptr(ids), nt.value)
    return ids, npx.value


def write_bmp(path, bgr, width, height):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    rc = lib.crt_write_bmp(path.encode(), _ptr(bgr), width, height)
    if rc < 0:
        raise CrtError(rc, f"cannot write {path}")


# ---------------------------------------------------------------- device context
class Context:
    """One crt_ctx: scene + BVH resident in HBM on one GPU, and the render entry points."""

    def __init__(self, device=0):
        err = _i(0)
        self.h = lib.crt_create(device, C.byref(err))
        if not self.h:
            hint = ""
            try:
                libs = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
                if len(libs) > 1:
                    hint = (" -- two HIP runtimes are loaded (" + ", ".join(sorted(libs)) + "): import torch BEFORE "
                            "cudaraytracer_amd so that both use the one PyTorch bundles")
            except OSError:
                pass
            raise CrtError(err.value, "crt_create failed (no HIP device?)" + hint)
        self.nprims = 0

    def close(self):
        if getattr(self, "h", None):
            lib.crt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise CrtError(rc, lib.crt_last_error(self.h).decode())

    def set_scene(self, prims):
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        self._ck(lib.crt_scene_set(self.h, _ptr(prims), len(prims)))
        self.nprims = len(prims)

    def update_scene(self, prims):
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        self._ck(lib.crt_scene_update(self.h, _ptr(prims), len(prims)))

    def animate_device(self):
        self._ck(lib.crt_animate_device(self.h))

    def get_scene(self):
        out = np.zeros(self.nprims, PRIM_DTYPE)
        self._ck(lib.crt_scene_get(self.h, _ptr(out), self.nprims))
        return out

    def set_lights(self, lights):
        lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
        self._ck(lib.crt_lights_set(self.h, _ptr(lights), len(lights)))

    def build(self, mode=0):
        self._ck(lib.crt_bvh_build(self.h, mode))

    def build_async(self):
        """crt_bvh_build_async: device rebuild from a snapshot of the scene, under the frames that follow"""
        self._ck(lib.crt_bvh_build_async(self.h))

    def swap(self, wait=False):
        """crt_bvh_swap: True when the asynchronously built tree became current (it is refitted: skip this frame's refit)"""
        sw = _i(0)
        self._ck(lib.crt_bvh_swap(self.h, int(wait), C.byref(sw)))
        return bool(sw.value)

    def refit(self, grow_only=False):
        self._ck(lib.crt_bvh_refit(self.h, int(grow_only)))

    def free_bvh(self):
        self._ck(lib.crt_bvh_free(self.h))

    def reference_order(self):
        out = np.zeros(self.nprims, np.int32)
        self._ck(lib.crt_bvh_reference_order(self.h, _ptr(out), self.nprims))
        return out

    def dump(self):
        n = C.c_int32(0)
        self._ck(lib.crt_bvh_dump(self.h, C.byref(n), None, None, None, 0))
        root = np.zeros(6, np.float32)
        boxes = np.zeros((n.value, 8, 6), np.float32)
        links = np.zeros((n.value, 8), np.uint32)
        self._ck(lib.crt_bvh_dump(self.h, C.byref(n), _ptr(root), _ptr(boxes), _ptr(links), n.value))
        return root, boxes, links

    def set_counting(self, on):
        self._ck(lib.crt_set_counting(self.h, int(on)))

    def set_capture(self, on):
        self._ck(lib.crt_set_capture(self.h, int(on)))

    @staticmethod
    def _cam(cam, target):
        c = Camera()
        c.position[:] = [float(x) for x in cam]
        c.target[:] = [float(x) for x in target]
        return c

    def render(self, cam, target, params):
        out = np.zeros((params.height, params.width, 3), np.uint8)
        c = self._cam(cam, target)
        self._ck(lib.crt_render(self.h, C.byref(c), C.byref(params), _ptr(out)))
        return out

    def render_device(self, cam, target, params, dev_ptr=None):
        c = self._cam(cam, target)
        self._ck(lib.crt_render_device(self.h, C.byref(c), C.byref(params), dev_ptr))

    def render_submit(self, slot, cam, target, params, dev_ptr=None):
        """crt_render_submit: queue a frame on frame slot `slot` without waiting; render_wait(slot) completes it."""
        c = self._cam(cam, target)
        self._ck(lib.crt_render_submit(self.h, slot, C.byref(c), C.byref(params), dev_ptr))

    def render_wait(self, slot):
        self._ck(lib.crt_render_wait(self.h, slot))

    def tiles_pack_slot(self, slot, params, dev_bgr, dev_packed):
        self._ck(lib.crt_tiles_pack_slot(self.h, slot, C.byref(params), dev_bgr, dev_packed))

    def capture(self, bounce, npixels):
        ids = np.zeros(npixels, np.int32)
        ts = np.zeros(npixels, np.float32)
        self._ck(lib.crt_get_capture(self.h, bounce, _ptr(ids), _ptr(ts), npixels))
        return ids, ts

    def color(self, npixels):
        out = np.zeros((npixels, 3), np.float32)
        self._ck(lib.crt_get_color(self.h, _ptr(out), npixels))
        return out

    def trace(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = len(rays6)
        ids = np.full(n, -1, np.int32)
        ts = np.zeros(n, np.float32)
        self._ck(lib.crt_trace(self.h, _ptr(rays6), n, _ptr(ids), _ptr(ts)))
        return ids, ts

    def trace_noalloc(self, dev_rays6, n, dev_ids, dev_ts):
        """crt_trace_noalloc: device pointers in and out, no allocation inside the call."""
        self._ck(lib.crt_trace_noalloc(self.h, dev_rays6, n, dev_ids, dev_ts))

    def trace_shadow(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        out = np.zeros(len(rays6), np.uint8)
        self._ck(lib.crt_trace_shadow(self.h, _ptr(rays6), len(rays6), _ptr(out)))
        return out

    def powf_probe(self, x, y):
        """pow(x, y) as k_shade's specular term evaluates it on the device (parity hook)"""
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        assert x.shape == y.shape
        out = np.zeros(x.shape, np.float32)
        self._ck(lib.crt_powf_probe(self.h, _ptr(x), _ptr(y), x.size, _ptr(out)))
        return out

    def blob(self):
        sz = C.c_size_t(0)
        p = _vp(0)
        self._ck(lib.crt_blob_size(self.h, C.byref(sz)))
        self._ck(lib.crt_blob_device_ptr(self.h, C.byref(p)))
        return p.value, sz.value

    def import_blob(self, dev_ptr, nbytes):
        self._ck(lib.crt_blob_import_device(self.h, dev_ptr, nbytes))

    def tiles_pack(self, params, dev_bgr, dev_packed):
        self._ck(lib.crt_tiles_pack(self.h, C.byref(params), dev_bgr, dev_packed))

    def tiles_unpack(self, params, src_rank, dev_packed, dev_bgr):
        self._ck(lib.crt_tiles_unpack(self.h, C.byref(params), src_rank, dev_packed, dev_bgr))

    def stats(self):
        s = Stats()
        self._ck(lib.crt_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def sync(self):
        self._ck(lib.crt_sync(self.h))


class MultiGpu:
    """crt_mgpu: one process, one context per GPU, RCCL broadcast of the scene and gather of the tiles (mgpu.cpp)."""

    def __init__(self, devices=None, rccl_always=False):
        err = _i(0)
        arr = (C.c_int * len(devices))(*devices) if devices else None
        self.h = lib.crt_mgpu_create(arr, len(devices) if devices else 0, 1 if rccl_always else 0, C.byref(err))
        if not self.h:
            raise CrtError(err.value, "crt_mgpu_create failed")
        self.size = lib.crt_mgpu_size(self.h)
        self.uses_rccl = bool(lib.crt_mgpu_uses_rccl(self.h))
        self.root = Context.__new__(Context)           # a view of rank 0's context: owned by the crt_mgpu
        self.root.h = lib.crt_mgpu_ctx(self.h, 0)
        self.root.nprims = 0
        self.root.close = lambda: None

    def _ck(self, rc):
        if rc != 0:
            raise CrtError(rc, lib.crt_mgpu_last_error(self.h).decode())

    def rank_stats(self, rank):
        s = Stats()
        rc = lib.crt_get_stats(lib.crt_mgpu_ctx(self.h, rank), C.byref(s))
        if rc:
            raise CrtError(rc, "crt_get_stats")
        return s.as_dict()

    def set_lights(self, lights):
        lights = np.ascontiguousarray(lights, LIGHT_DTYPE)
        self._ck(lib.crt_mgpu_lights_set(self.h, _ptr(lights), len(lights)))

    def sync_scene(self):
        self._ck(lib.crt_mgpu_sync_scene(self.h))

    def render(self, cam, target, params):
        out = np.zeros((params.height, params.width, 3), np.uint8)
        c = Context._cam(cam, target)
        self._ck(lib.crt_mgpu_render(self.h, C.byref(c), C.byref(params), _ptr(out)))
        return out

    def render_device(self, cam, target, params):
        c = Context._cam(cam, target)
        p = _vp(0)
        self._ck(lib.crt_mgpu_render_device(self.h, C.byref(c), C.byref(params), C.byref(p)))
        return p.value

    def set_scene(self, prims):
        """crt_mgpu_scene_set: the scene on every rank (animate() then needs no exchange)"""
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        self._ck(lib.crt_mgpu_scene_set(self.h, _ptr(prims), len(prims)))
        self.root.nprims = len(prims)

    def animate(self, rebuild=False, grow_only=False):
        self._ck(lib.crt_mgpu_animate(self.h, int(rebuild), int(grow_only)))

    def submit(self, cam, target, params):
        """crt_mgpu_submit: returns the device address of the most recent frame completed by the call, or None"""
        c = Context._cam(cam, target)
        p = _vp(0)
        self._ck(lib.crt_mgpu_submit(self.h, C.byref(c), C.byref(params), C.byref(p)))
        return p.value

    def finish(self):
        p = _vp(0)
        self._ck(lib.crt_mgpu_finish(self.h, C.byref(p)))
        return p.value

    def times(self):
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        self._ck(lib.crt_mgpu_get_times(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"ms_render": a.value, "ms_exchange": b.value, "ms_unpack": c.value}

    def close(self):
        if getattr(self, "h", None):
            lib.crt_mgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
