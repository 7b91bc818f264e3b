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


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
