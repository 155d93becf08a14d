"""cudaraytracer_amd -- MI355X-native render path with CudaRayTracer's scene/BVH/render API.

The product is the C-ABI shared library ``libcrt_amd.so`` (declared in ``include/crt.h``,
built in-tree from ``cudaraytracer_amd/csrc`` by ``__graft_entry__.build()``).  This package
is the thin ctypes view of it used by the tests, ``bench.py`` and the multi-GPU launcher; it
has no compute of its own and no CPU fallback: if the library is missing, importing
``cudaraytracer_amd.api`` raises.
"""
from .api import (  # noqa: F401
    Context, MultiGpu, CrtError, PRIM_DTYPE, LIGHT_DTYPE, lib, lib_path, default_params, default_camera,
    default_lights, scene_teapot, scene_spheres, scene_soup, animate_host, load_obj, tiles_owned,
    tile_owner, write_bmp, host_bvh, area_table, TEAPOT_OBJ,
)
