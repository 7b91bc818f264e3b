"""spnn.Conv3d / BatchNorm / ReLU as used by BasicSparse{Conv,Deconv}olutionBlock (tsparse/modules.py:94-124)."""
import importlib
import math

import torch
import torch.nn as nn

from ..tensor import SparseTensor
from . import functional, utils  # noqa: F401

_ops = None


def _o():
    global _ops
    if _ops is None:
        _ops = importlib.import_module("one-2-3-45_amd.ops")
    return _ops


def _require_device(t):
    """HIP-only: there is no CPU fallback (ops would refuse a CPU tensor anyway; this gives the clearer message)."""
    if not t.is_cuda:
        raise RuntimeError("o2345 torchsparse shim: HIP-only (no CPU fallback)")


def _level(x):
    """Level descriptor of x's stride: (coords int32 [N,4], index grid, lattice cells/axis)."""
    lv = x.cmaps.get(x.s)
    if lv is None:
        if x.s != 1:
            raise RuntimeError("o2345 torchsparse shim: a SparseTensor must enter the network at stride 1")
        c = x.C.to(torch.int32).contiguous()
        cells = tuple(int(v) + 1 for v in c[:, :3].max(0).values.tolist())
        lin = (c[:, 0].long() * cells[1] + c[:, 1].long()) * cells[2] + c[:, 2].long()
        grid = torch.full((cells[0] * cells[1] * cells[2],), -1, dtype=torch.int32, device=c.device)
        grid[lin] = torch.arange(c.shape[0], dtype=torch.int32, device=c.device)
        lv = x.cmaps[x.s] = (c, grid, cells)
    return lv


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False, transposed=False):
        super().__init__()
        if kernel_size != 3 or dilation != 1 or bias or stride not in (1, 2):
            raise NotImplementedError("o2345 torchsparse shim: only the kernel-3 / stride-1|2 / no-bias convs of SparseCostRegNet")
        self.in_channels, self.out_channels, self.stride, self.transposed = in_channels, out_channels, stride, transposed
        self.kernel = nn.Parameter(torch.zeros(27, in_channels, out_channels))
        std = 1.0 / math.sqrt((out_channels if transposed else in_channels) * 27)
        self.kernel.data.uniform_(-std, std)

    def forward(self, x):
        ops = _o()
        _require_device(x.F)
        c_in, g_in, cells_in = _level(x)
        k = self.kernel.detach().contiguous()
        f = x.F.detach().contiguous()
        if not self.transposed and self.stride == 1:
            return x._like(ops.sparse_conv3d(0, f, g_in, cells_in, c_in, x.s, k))
        if not self.transposed:
            so = x.s * 2
            if so not in x.cmaps:
                grid, cc, n, cells = ops.sparse_downsample(c_in, x.s, cells_in)
                x.cmaps[so] = (cc, grid, cells)
            c_out, _, _ = x.cmaps[so]
            return x._like(ops.sparse_conv3d(1, f, g_in, cells_in, c_out, so, k), c_out, so)
        so = x.s // 2
        c_out, _, _ = x.cmaps[so]          # cached by the matching strided conv (torchsparse kmap reuse)
        return x._like(ops.sparse_conv3d(2, f, g_in, cells_in, c_out, so, k), c_out, so)


class BatchNorm(nn.BatchNorm1d):
    """BatchNorm over the rows of a SparseTensor.  Training mode (what the reference always runs) uses batch statistics on
    the HIP path and updates the running buffers like nn.BatchNorm1d; eval mode uses the running buffers."""
    fuse_relu = False      # set by Sequential pattern matching below

    def forward(self, x):
        ops = _o()
        f = x.F.detach().contiguous()
        if self.training:
            y, mv = ops.bn_act_rows(f, self.weight.detach(), self.bias.detach(), self.eps, slope=1.0, want_stats=True)
            with torch.no_grad():
                n = f.shape[0]
                m = self.momentum if self.momentum is not None else 0.1
                self.running_mean.mul_(1 - m).add_(mv[0], alpha=m)
                self.running_var.mul_(1 - m).add_(mv[1] * (n / max(n - 1, 1)), alpha=m)
                self.num_batches_tracked += 1
            return x._like(y)
        scale = self.weight / torch.sqrt(self.running_var + self.eps)
        return x._like(f * scale + (self.bias - self.running_mean * scale))


class ReLU(nn.ReLU):
    def forward(self, x):
        return x._like(torch.relu_(x.F) if self.inplace else torch.relu(x.F))
