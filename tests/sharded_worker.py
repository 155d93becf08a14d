"""One rank of the two-process rehearsal of ShardedRenderer on a single GPU (gloo backend, both ranks on
device 0).  Launched by tests/test_gpu_sharded_renderer.py; rank 0 saves the gathered frame."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402
from cudaraytracer_amd.parallel import ShardedRenderer  # noqa: E402


def main():
    out, w, h, bounces, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    frames = int(sys.argv[6]) if len(sys.argv) > 6 else 0          # animation steps before the frame that is saved
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    backend = os.environ.get("CRT_TEST_BACKEND", "gloo")      # "nccl" with WORLD_SIZE=1: RCCL's device gather on the one GPU there is
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    sr = ShardedRenderer(dist, torch, rank, world, 0, always_gather=(world == 1))
    prims = np.concatenate([crt.scene_teapot(with_sphere=True), crt.scene_soup(5000, seed=6)]) if rank == 0 else None
    if frames and rank == 0:
        prims = np.concatenate([prims, crt.scene_spheres(200, seed=8)])
    sr.broadcast_scene(prims, crt.default_lights() if rank == 0 else None, mode, animated=frames > 0)
    cam, tgt = crt.default_camera()
    p = crt.default_params(width=w, height=h, bounces=bounces)
    pipelined = os.environ.get("CRT_TEST_PIPELINED") == "1"      # the gather of frame k under the render of frame k+1
    for _ in range(2):                                   # twice: buffers and tile lists are reused
        frame = sr.render(cam, tgt, p, pipelined=pipelined)
    for f in range(1, frames + 1):                       # replicated animation: refit, every 4th step a device rebuild
        sr.animate(rebuild=(f % 4 == 0))
        frame = sr.render(cam, tgt, p, pipelined=pipelined)
    resize = os.environ.get("CRT_TEST_RESIZE")          # "WxH": change the resolution while pipelined frames are in flight
    if resize:
        w, h = (int(v) for v in resize.split("x"))
        p = crt.default_params(width=w, height=h, bounces=bounces)
        for _ in range(3):                               # the first call finds frames of the old size in flight: they are completed first
            frame = sr.render(cam, tgt, p, pipelined=pipelined)
        if not pipelined:
            frame = sr.render(cam, tgt, p, pipelined=False)
    if pipelined:
        frame = sr.finish()                              # render() returned the frame before; this is the last one
    if rank == 0:
        np.save(out, frame.cpu().numpy().reshape(h, w, 3))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
