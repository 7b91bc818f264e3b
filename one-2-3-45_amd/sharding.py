"""Scene-level data parallelism: one process per GPU, scenes dealt round-robin, NO data-path collective
(the reference does the same with nn.DataParallel, exp_runner_generic_blender_val.py:63,151).
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to synchronise the clock."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend=None):
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
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
    to read the checkpoint itself; this exists for deployments where only one rank can reach the file."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return state
    rank = dist.get_rank()
    layout = None
    if rank == src:
        layout = [(n, k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for n, sd in state.items() for k, v in sd.items()]
    box = [layout]
    dist.broadcast_object_list(box, src=src)
    layout = box[0]
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


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
