"""Scene-level data parallelism: one process per GPU, scenes dealt round-robin, NO data-path collective
(the reference does the same with nn.DataParallel, exp_runner_generic_blender_val.py:63,151).
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to synchronise the clock.

Optional second mode ("intra-scene split" for single-scene latency): the rays of ONE image are split into contiguous blocks, one per
rank.  The path has exactly one exchange step each way: the scene's source images go out in ONE broadcast (6.3 MB at 8 views; every rank then builds
the latent volume itself -- 2 ms, deterministic kernels, the same bits on every GPU -- instead of receiving 143 MB of volume + 537 MB of colour maps),
the per-ray results come back in ONE all-gather (ray_block / broadcast_tensor / gather_ray_blocks below, pipeline.render_scene_split)."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend=None, force=False):
    """Join the job's process group (a single process needs none; ``force`` = initialise it anyway, e.g. to exercise the RCCL path on one device)."""
    rank, world, local = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def scenes_for_rank(n_scenes, rank, world):
    """Scene k runs on rank k mod world (weak scaling: n_scenes = scenes_per_gpu * world)."""
    return list(range(rank, n_scenes, world))


def barrier(device=None):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def _reduce_device(device):
    """RCCL reduces device tensors, gloo (CPU tests, shared-GPU functional runs) host tensors."""
    return device if (device is not None and dist.get_backend() == "nccl") else "cpu"


def max_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_objects(obj):
    """-> [obj of rank 0, ..., obj of rank world-1] on every rank (tiny python objects: per-rank timings, device ids)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def broadcast_state_dicts(state, device=None, src=0):
    """The north star's "optional shared-backbone broadcast": rank ``src`` holds ``state`` = {network name: {key: tensor}} (e.g. read from ONE
    checkpoint file), every other rank passes None and receives an identical copy -- ONE collective on one flat fp32 buffer (< 4 MB for the lod-0
    model: a single RCCL broadcast over xGMI; gloo in the CPU tests) + one small object broadcast for the layout.  The default is for every rank
    to read the checkpoint itself; this exists for deployments where only one rank can reach the file.  Rank ``src`` may pass an exception instead."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        if isinstance(state, BaseException):
            raise state
        return state
    rank = dist.get_rank()
    layout = None
    if rank == src:                                # an exception instead of the state: rank src could not produce it -- EVERY rank raises (none is left waiting)
        layout = ("error", f"{type(state).__name__}: {state}") if isinstance(state, BaseException) else \
            [(n, k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for n, sd in state.items() for k, v in sd.items()]
    box = [layout]
    dist.broadcast_object_list(box, src=src)
    layout = box[0]
    if isinstance(layout, tuple) and layout and layout[0] == "error":
        if rank == src and isinstance(state, BaseException):
            raise state
        raise RuntimeError(f"o2345 broadcast_state_dicts: rank {src} failed to produce the weights ({layout[1]})")
    total = sum(int(torch.Size(shp).numel()) for _, _, shp, _ in layout)
    dev = _reduce_device(device)
    if rank == src:
        flat = torch.cat([state[n][k].detach().reshape(-1).to(torch.float64 if dt == "float64" else torch.float32).float() for n, k, _, dt in layout]).to(dev)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=dev)
    dist.broadcast(flat, src=src)
    flat = flat.cpu()
    out, o = {}, 0
    for n, k, shp, dt in layout:
        cnt = int(torch.Size(shp).numel())
        out.setdefault(n, {})[k] = flat[o:o + cnt].reshape(shp).to(getattr(torch, dt)).clone()
        o += cnt
    return out


def ray_block(n_rays, rank, world, align=64):
    """-> (lo, hi, per): rank ``rank`` renders rays [lo, hi); every block is ``per`` rays long (a multiple of ``align`` = one wavefront of rays) except
    that the last non-empty one may be short and trailing ones empty.  Blocks are contiguous: neighbouring rays share map pixels and voxels."""
    per = -(-int(n_rays) // int(world))
    per = max(-(-per // align) * align, align)
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays), per


def broadcast_tensor(t, src=0, device=None):
    """Rank ``src`` passes a tensor, every other rank None; all ranks return the same tensor (on ``device``): one small object broadcast for shape /
    dtype and ONE collective for the data."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    box = [None if t is None else (tuple(t.shape), str(t.dtype).replace("torch.", ""))]
    dist.broadcast_object_list(box, src=src)
    shape, dt = box[0]
    dev = _reduce_device(device)
    buf = t.detach().contiguous().to(dev) if dist.get_rank() == src else torch.empty(shape, dtype=getattr(torch, dt), device=dev)
    dist.broadcast(buf, src=src)
    return buf if device is None else buf.to(device)


def gather_ray_blocks(block, n_rays, per, device=None):
    """``block``: {key: tensor [n_block, ...]} of this rank's rays (float32 / uint8 / bool; n_block <= per, possibly 0).  -> {key: tensor [n_rays, ...]} on
    every rank, rank order = ray order: the values are packed into one float32 buffer [per, C] (exact for these types) and exchanged in ONE all-gather."""
    keys = sorted(block)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return {k: block[k] for k in keys}
    world = dist.get_world_size()
    dev = _reduce_device(device)
    n = int(block[keys[0]].shape[0])
    cols = [int(torch.Size(block[k].shape[1:]).numel()) for k in keys]
    buf = torch.zeros(per, sum(cols), dtype=torch.float32, device=dev)
    if n:
        buf[:n] = torch.cat([block[k].reshape(n, -1).to(torch.float32) for k in keys], dim=1).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    full = torch.cat(parts, dim=0)[:n_rays]
    out, o = {}, 0
    for k, c in zip(keys, cols):
        v = full[:, o:o + c].reshape((n_rays,) + tuple(block[k].shape[1:])).to(block[k].dtype)
        out[k] = v if device is None else v.to(device)
        o += c
    return out


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
