"""PyMCubes surface used by the reference (models/sparse_neus_renderer.py:12,932): marching_cubes(u, isovalue) ->
(vertices float64 [Nv,3] in index coordinates, triangles [Nt,3]).  Runs on the MI355X; numpy in, numpy out."""
import importlib

import numpy as np
import torch


def marching_cubes(volume, isovalue):
    ops = importlib.import_module("one-2-3-45_amd.ops")
    if not torch.cuda.is_available():
        raise RuntimeError("o2345 mcubes shim: HIP-only (no CPU fallback)")
    u = volume if torch.is_tensor(volume) else torch.from_numpy(np.ascontiguousarray(volume, dtype=np.float32))
    u = u.to("cuda", torch.float32).contiguous()
    v, t = ops.marching_cubes(u, float(isovalue))
    vh, th = ops.to_host_numpy(v, t)
    return vh, th
