"""Tile sharding of one frame over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" for the CPU
rehearsal in tests).  Pixels are independent, so the only exchanges are
  1. broadcast of the packed scene + BVH blob from rank 0 after a build, and
  2. gather of each rank's tile pixels (BGR8, tile-major) to rank 0, which scatters them into
     the bottom-up frame.
There is no reduction anywhere.  torch is used for the device buffers and the collectives only.
Import torch BEFORE this package: PyTorch bundles its own HIP runtime, and loading libcrt_amd.so
(system runtime) first leaves the process with two runtimes, the second without a device.

The tile -> rank map and the tile-major packing are defined by the C ABI (crt_tile_owner,
crt_tiles_owned, crt_tiles_pack/unpack); `pack_tiles_host` / `unpack_tiles_host` restate the same
byte layout on numpy arrays for ranks that hold their pixels on the host (the gloo tests).
"""
import ctypes as C

import numpy as np

from . import api


def packed_nbytes(params, rank=None):
    """Bytes of the tile-major buffer of `rank` (tiles are stored whole, clipped pixels left unset)."""
    p = api.Params.from_buffer_copy(params)
    if rank is not None:
        p.rank = rank
    ids, _ = api.tiles_owned(p)
    return len(ids) * p.tile_w * p.tile_h * 3


def _tile_pixel_index(params, rank):
    p = api.Params.from_buffer_copy(params)
    p.rank = rank
    ids, _ = api.tiles_owned(p)
    tw, th = p.tile_w, p.tile_h
    tiles_x = (p.width + tw - 1) // tw
    r = np.arange(tw * th)
    px = (ids[:, None] % tiles_x) * tw + (r % tw)[None, :]
    py = (ids[:, None] // tiles_x) * th + (r // tw)[None, :]
    valid = (px < p.width) & (py < p.height)
    return (py * p.width + px).ravel(), valid.ravel()


def pack_tiles_host(params, rank, frame_bgr):
    """frame (h, w, 3) uint8 -> tile-major bytes of `rank`, same layout as crt_tiles_pack."""
    pix, valid = _tile_pixel_index(params, rank)
    flat = np.ascontiguousarray(frame_bgr, np.uint8).reshape(-1, 3)
    out = np.zeros((len(pix), 3), np.uint8)
    out[valid] = flat[pix[valid]]
    return out.ravel()


def unpack_tiles_host(params, src_rank, packed, frame_bgr):
    """inverse of pack_tiles_host: scatter `src_rank`'s tiles into frame (h, w, 3) in place."""
    pix, valid = _tile_pixel_index(params, src_rank)
    flat = frame_bgr.reshape(-1, 3)
    flat[pix[valid]] = np.asarray(packed, np.uint8).reshape(-1, 3)[valid]
    return frame_bgr


def gather_frame_host(dist, params, local_packed, rank, world):
    """Gather every rank's tile bytes to rank 0 (any backend, host tensors) and assemble the frame."""
    import torch
    sizes = [packed_nbytes(params, r) for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros(mx, dtype=torch.uint8)
    buf[: len(local_packed)] = torch.from_numpy(np.ascontiguousarray(local_packed))
    out = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    frame = np.zeros((params.height, params.width, 3), np.uint8)
    for r in range(world):
        unpack_tiles_host(params, r, out[r][: sizes[r]].numpy(), frame)
    return frame


class ShardedRenderer:
    """Device path: one crt Context per rank, scene broadcast once, per-frame tile gather over RCCL.

    render(..., pipelined=True) is the throughput mode bench.py uses: `slots` frames are in flight on the context's
    frame slots 1..slots (crt_render_submit / crt_render_wait, each with the tile pack queued behind it), so that the
    launches of one frame run in the tails of the others' persistent launches, and the gathers of finished frames run
    under the renders of later ones.  Frames complete in order, `2 * slots` calls late at most; finish() drains."""

    def __init__(self, dist, torch, rank, world, local_rank, slots=3, always_gather=False):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, world
        self.dev = torch.device("cuda", local_rank)
        self.ctx = api.Context(local_rank)
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipMemcpy.restype = C.c_int
        # RCCL gathers device tensors directly; gloo (the one-GPU rehearsal) has no device gather, so stage through the host
        self.host_gather = dist.get_backend() != "nccl"
        # always_gather: run the collective for a single rank too (the one-GPU box exercises RCCL's device gather that way)
        self.always_gather = bool(always_gather)
        self.K = max(1, min(int(slots), api.MAX_FRAME_SLOTS - 1))      # slot 0 (the context's own stream) stays free for the unpacks
        self.frame = None
        self.frames = None          # 2K + 1 device frames, 2K packed-tile and gather buffers: see render()
        self.packs = None
        self.gathers = None
        self.sizes = None
        self.npix = None
        self.seq = 0                # frames submitted in pipelined mode
        self.inflight = []          # (params, frame buffer, pack buffer, slot) of frames being rendered, oldest first
        self.gathering = []         # (params, frame buffer, pack buffer, work, keep-alive) of frames being gathered, oldest first
        self.last_stats = None      # crt_stats of the frame whose render completed last

    def _bcast_bytes(self, payload):
        """broadcast a bytes-like object from rank 0 (device tensors: RCCL has no host path)"""
        torch, dist = self.torch, self.dist
        n = torch.tensor([len(payload) if self.rank == 0 else 0], dtype=torch.int64, device=self.dev)
        dist.broadcast(n, 0)
        if self.rank == 0:
            buf = torch.from_numpy(np.frombuffer(payload, np.uint8).copy()).to(self.dev)
        else:
            buf = torch.empty(int(n.item()), dtype=torch.uint8, device=self.dev)
        dist.broadcast(buf, 0)
        return buf.cpu().numpy().tobytes()

    def broadcast_scene(self, prims=None, lights=None, mode=0, animated=False):
        """Rank 0 builds; everyone ends with the same packed scene + BVH resident in its HBM.

        animated=True also hands every rank the crt_prim array (SURVEY 8(e) collective (2)), so that animate()
        can advance the scene on every rank without further communication."""
        torch, dist = self.torch, self.dist
        self.animated = animated
        if animated:
            raw = self._bcast_bytes(np.ascontiguousarray(prims, api.PRIM_DTYPE).tobytes() if self.rank == 0 else b"")
            if self.rank != 0:
                self.ctx.set_scene(np.frombuffer(raw, api.PRIM_DTYPE).copy())
        if self.rank == 0:
            self.ctx.set_scene(prims)
            self.ctx.build(mode)
            ptr, nbytes = self.ctx.blob()
        else:
            ptr, nbytes = 0, 0
        n = torch.tensor([nbytes], dtype=torch.int64, device=self.dev)
        dist.broadcast(n, 0)
        nbytes = int(n.item())
        blob = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        if self.rank == 0:
            rc = self._hip.hipMemcpy(blob.data_ptr(), ptr, nbytes, 3)   # hipMemcpyDeviceToDevice
            if rc != 0:
                raise RuntimeError(f"hipMemcpy of the scene blob failed: {rc}")
        dist.broadcast(blob, 0)
        torch.cuda.synchronize(self.dev)
        if self.rank != 0:
            self.ctx.import_blob(blob.data_ptr(), nbytes)
        lt = torch.from_numpy(np.frombuffer(np.ascontiguousarray(lights, api.LIGHT_DTYPE).tobytes(), np.uint8).copy()) \
            if self.rank == 0 else None
        ln = torch.tensor([0 if lt is None else lt.numel()], dtype=torch.int64, device=self.dev)
        dist.broadcast(ln, 0)
        lb = lt.to(self.dev) if self.rank == 0 else torch.empty(int(ln.item()), dtype=torch.uint8, device=self.dev)
        dist.broadcast(lb, 0)
        self.ctx.set_lights(np.frombuffer(lb.cpu().numpy().tobytes(), api.LIGHT_DTYPE))
        return nbytes

    def animate(self, rebuild=False, grow_only=False):
        """One step of the reference's animation (main.cu:545-573) on every rank: the step is deterministic, so the
        ranks replicate it on their own copy of the scene instead of exchanging moved primitives; then a refit, or
        (rebuild=True, the "every 20th frame" of main.cu:534-541) a rebuild on the device.  No collective."""
        if not getattr(self, "animated", False):
            raise RuntimeError("broadcast_scene(..., animated=True) first: the other ranks hold no crt_prim array")
        self.ctx.animate_device()
        if rebuild:
            self.ctx.build(2)          # CRT_BVH_WIDE_DEVICE
        else:
            self.ctx.refit(grow_only)

    def _ensure(self, params):
        torch = self.torch
        npix = params.width * params.height
        sizes = [packed_nbytes(params, r) for r in range(self.world)]
        if self.npix == npix and self.sizes == sizes:
            return
        self.finish()               # frames in flight belong to the old buffers: complete them before anything is replaced
        self.npix, self.sizes = npix, sizes
        nb = 2 * self.K
        self.frames = [torch.zeros(npix * 3, dtype=torch.uint8, device=self.dev) for _ in range(nb + 1)]
        self.frame = self.frames[0]
        mx = max(sizes)
        self.packs = [torch.zeros(mx, dtype=torch.uint8, device=self.dev) for _ in range(nb)]
        self.gathers = [[torch.zeros(mx, dtype=torch.uint8, device=self.dev) for _ in range(self.world)] for _ in range(nb)] \
            if self.rank == 0 else [None] * nb

    def _gather(self, packed, gathered, async_op):
        """gather of the packed tiles to rank 0; returns (work or None, keepalive)"""
        if self.host_gather:      # gloo rehearsal: no device gather, stage through the host
            mine = packed.cpu()
            got = [self.torch.zeros_like(mine) for _ in range(self.world)] if self.rank == 0 else None
            w = self.dist.gather(mine, got, dst=0, async_op=async_op)
            return w, (mine, got)
        w = self.dist.gather(packed, gathered, dst=0, async_op=async_op)
        return w, None

    def _complete(self, params, fb, pb, work, keep):
        """wait for a gather and scatter the other ranks' tiles into the frame (rank 0)"""
        if work is not None:
            work.wait()
        if self.host_gather and self.rank == 0:
            for r in range(1, self.world):
                self.gathers[pb][r].copy_(keep[1][r])
        # every rank waits for its part of the gather (a later frame packs into the same buffer on another stream).  The
        # collective's own stream only: a device-wide synchronise would also wait for the frames in flight
        self.torch.cuda.current_stream(self.dev).synchronize()
        if self.rank == 0:
            for r in range(1, self.world):
                self.ctx.tiles_unpack(params, r, self.gathers[pb][r].data_ptr(), self.frames[fb].data_ptr())
            self.ctx.sync()

    def _retire(self, phase=None, keep_gathers=None):
        """oldest frame in flight: wait for its render + pack, start its gather; complete gathers beyond `keep_gathers`.
        Returns the frame buffer index of the last frame completed here, or None."""
        import time
        done = None
        if self.inflight:
            params, fb, pb, slot = self.inflight.pop(0)
            t0 = time.perf_counter()
            self.ctx.render_wait(slot)
            self.last_stats = self.ctx.stats()
            t1 = time.perf_counter()
            if self.world == 1 and not self.always_gather:
                done = fb
            else:
                work, keep = self._gather(self.packs[pb], self.gathers[pb], async_op=True)
                self.gathering.append((params, fb, pb, work, keep))
            if phase is not None:
                phase["render"] = phase.get("render", 0.0) + (t1 - t0)
                phase["gather"] = phase.get("gather", 0.0) + (time.perf_counter() - t1)
        limit = self.K if keep_gathers is None else keep_gathers
        while len(self.gathering) > limit:
            params, fb, pb, work, keep = self.gathering.pop(0)
            t0 = time.perf_counter()
            self._complete(params, fb, pb, work, keep)
            if phase is not None:
                phase["unpack"] = phase.get("unpack", 0.0) + (time.perf_counter() - t0)
            done = fb
        return done

    def finish(self):
        """pipelined mode: complete every frame still in flight; returns the last one on rank 0"""
        done = None
        while self.inflight or self.gathering:
            d = self._retire(keep_gathers=0 if not self.inflight else None)
            done = d if d is not None else done
        if done is None:
            return None
        self.frame = self.frames[done]
        return self.frame if self.rank == 0 else None

    def render(self, cam, target, params, phase=None, pipelined=False):
        """Render this rank's tiles, gather to rank 0.  Returns the device frame tensor on rank 0.
        `phase` (optional dict) accumulates this rank's wall seconds per phase: render, pack, gather, unpack.

        pipelined=True: the frame is SUBMITTED on the next of K frame slots (render and tile pack queued on the slot's
        stream, no wait).  When all K slots are busy the oldest frame is waited for first and its gather started
        (asynchronously, on the collective's own stream); gathers more than K frames old are completed and unpacked.
        Frame i renders into frame buffer i mod (2K + 1) and packs into pack buffer i mod 2K: its gather is complete
        before frame i + 2K packs.  The call returns the most recent frame completed by it (the frame submitted 2K calls
        earlier, in steady state), or None; finish() completes the rest.  Throughput mode for bench.py: a step then costs
        max(render, exchange) with the launch tails of one frame filled by the next, instead of their sum."""
        import time
        params.rank, params.nranks = self.rank, self.world
        self._ensure(params)
        if pipelined:
            t0 = time.perf_counter()
            done = self._retire(phase) if len(self.inflight) >= self.K else None
            t1 = time.perf_counter()
            nb = 2 * self.K
            fb, pb, slot = self.seq % (nb + 1), self.seq % nb, 1 + self.seq % self.K
            q = api.Params.from_buffer_copy(params)
            q.timing = 1                      # frame start / end events only: per-kernel intervals mean nothing between overlapping frames
            self.ctx.render_submit(slot, cam, target, q, self.frames[fb].data_ptr())
            if self.world > 1 or self.always_gather:
                self.ctx.tiles_pack_slot(slot, q, self.frames[fb].data_ptr(), self.packs[pb].data_ptr())
            self.inflight.append((q, fb, pb, slot))
            self.seq += 1
            if phase is not None:
                phase["pack"] = phase.get("pack", 0.0) + (time.perf_counter() - t1)      # host time of submit + pack launch
            if done is None:
                return None
            self.frame = self.frames[done]
            return self.frame if self.rank == 0 else None
        self.finish()                          # a frame still in flight from pipelined calls uses the same buffers
        self.frame = self.frames[0]
        t0 = time.perf_counter()
        self.ctx.render_device(cam, target, params, self.frames[0].data_ptr())       # returns with the frame complete
        self.last_stats = self.ctx.stats()
        t1 = time.perf_counter()
        if phase is not None:
            phase["render"] = phase.get("render", 0.0) + (t1 - t0)
        if self.world == 1 and not self.always_gather:
            return self.frames[0]
        self.ctx.tiles_pack(params, self.frames[0].data_ptr(), self.packs[0].data_ptr())     # returns with the copy finished
        t2 = time.perf_counter()
        work, keep = self._gather(self.packs[0], self.gathers[0], async_op=False)
        t3 = time.perf_counter()
        self._complete(params, 0, 0, None, keep)
        if phase is not None:
            phase["pack"] = phase.get("pack", 0.0) + (t2 - t1)
            phase["gather"] = phase.get("gather", 0.0) + (t3 - t2)
            phase["unpack"] = phase.get("unpack", 0.0) + (time.perf_counter() - t3)
        return self.frames[0] if self.rank == 0 else None
