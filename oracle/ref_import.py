"""Import the REAL reference modules from /root/reference on CPU (test infrastructure only).

Used (a) to pin the oracle restatement (tests/test_oracle_vs_reference.py) and (b) to generate the golden
vectors under tests/golden/ (tests/golden/make_golden.py).  /root/reference does not exist on the GPU box, so
nothing that runs there imports this module.

The reference needs torchsparse / inplace_abn / mcubes / cv2 / icecream / trimesh, none of which is installed;
they are replaced by stub modules in ``sys.modules`` (recipe: SURVEY.md Appendix E).  The torchsparse and
inplace_abn stubs are *functional*: they implement the published semantics through ``oracle.recon`` so that the
reference's own ``SparseCostRegNet`` / ``ConvBnReLU`` topology runs end to end.
"""
import os
import sys
import types

import torch
import torch.nn as nn

from . import recon as O

# the reference tree; O2345_REFERENCE_DIR points at an untracked working copy when the tree is shipped to the GPU box for CPU timing (tools/reference_cpu_on_gpu_box.sh)
REF = os.environ.get("O2345_REFERENCE_DIR", "/root/reference/reconstruction")


def available():
    return os.path.isdir(REF)


# ------------------------------------------------------------------ functional stubs
class SparseTensor:
    def __init__(self, feats, coords, stride=1):
        self.F, self.C, self.s = feats, coords, stride
        self.cmaps, self.kmaps = {}, {}

    def __add__(self, other):
        out = SparseTensor(self.F + other.F, self.C, self.s)
        out.cmaps, out.kmaps = self.cmaps, self.kmaps
        return out


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F, self.C = feats, coords


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False, transposed=False):
        super().__init__()
        assert kernel_size == 3 and dilation == 1 and not bias
        self.stride, self.transposed = stride, transposed
        kv = 27
        self.kernel = nn.Parameter(torch.zeros(kv, in_channels, out_channels))
        std = 1.0 / (((out_channels if transposed else in_channels) * kv) ** 0.5)
        self.kernel.data.uniform_(-std, std)

    def forward(self, x):
        lv = x.cmaps.setdefault(x.s, O.SparseLevel(x.C[:, :3], x.s))
        if not self.transposed:
            if self.stride == 1:
                lo = lv
            else:
                lo = x.cmaps.get(x.s * 2) or O.downsample_coords(lv)
                x.cmaps[x.s * 2] = lo
            key = (x.s, self.stride)
            if key not in x.kmaps:
                x.kmaps[key] = O.build_kmap(lv, lo)
            feats = O.sparse_conv(x.F, x.kmaps[key], self.kernel)
        else:
            lo = x.cmaps[x.s // 2]
            feats = O.sparse_conv(x.F, x.kmaps[(x.s // 2, 2)], self.kernel, True, len(lo.xyz))
        coords = torch.cat([lo.xyz.to(torch.int32), torch.zeros(len(lo.xyz), 1, dtype=torch.int32)], 1)
        out = SparseTensor(feats, coords, lo.ts)
        out.cmaps, out.kmaps = x.cmaps, x.kmaps
        return out


class BatchNorm(nn.BatchNorm1d):
    def forward(self, x):
        out = SparseTensor(super().forward(x.F), x.C, x.s)
        out.cmaps, out.kmaps = x.cmaps, x.kmaps
        return out


class ReLU(nn.ReLU):
    def forward(self, x):
        out = SparseTensor(torch.relu(x.F), x.C, x.s)
        out.cmaps, out.kmaps = x.cmaps, x.kmaps
        return out


class InPlaceABN(nn.Module):
    """Stand-in for inplace_abn.InPlaceABN (leaky_relu 0.01, training-mode statistics)."""
    abs_gamma = True

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        return O.abn_train(x, self.weight, self.bias, self.eps, 0.01, self.abs_gamma)


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ts = mod("torchsparse", SparseTensor=SparseTensor, PointTensor=PointTensor, cat=None)
    ts.tensor = mod("torchsparse.tensor", SparseTensor=SparseTensor, PointTensor=PointTensor)
    ts.nn = mod("torchsparse.nn", Conv3d=Conv3d, BatchNorm=BatchNorm, ReLU=ReLU)
    ts.nn.functional = mod("torchsparse.nn.functional")
    ts.nn.utils = mod("torchsparse.nn.utils", get_kernel_offsets=None)
    mod("inplace_abn", InPlaceABN=InPlaceABN)
    for n in ("cv2", "mcubes", "trimesh"):
        mod(n)
    mod("icecream", ic=lambda *a, **k: None)


class Conf(dict):
    def get_int(self, k, default=None):
        return int(self.get(k, default))

    get_float = get_bool = get_int


_CACHE = None


def load():
    """Return a namespace of the reference classes/functions used by tests and golden generation."""
    global _CACHE
    if _CACHE is not None:
        return _CACHE
    assert available(), "reference tree not present"
    sys.dont_write_bytecode = True
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from models.sparse_sdf_network import SparseSdfNetwork
    from models.sparse_neus_renderer import SparseNeuSRenderer
    from models.rendering_network import GeneralRenderingNetwork
    from models.fields import SingleVarianceNetwork
    from models.projector import Projector
    from ops.grid_sampler import grid_sample_3d
    from ops.back_project import back_project_sparse_type
    from models.rays import gen_rays_from_single_image
    _CACHE = types.SimpleNamespace(**locals())
    return _CACHE


def build_networks(D, seed=0, voxel_size=None):
    """Seeded reference networks as the runner builds them (exp_runner_generic_blender_val.py:93-129,
    confs/one2345_lod0_val_demo.conf:65-128) but with a DxDxD volume."""
    R = load()
    torch.manual_seed(seed)
    vs = voxel_size if voxel_size is not None else 2.0 / (D - 1)
    sdfnet = R.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=vs, vol_dims=[D, D, D], hidden_dim=128,
                                cost_type="variance_mean", d_pyramid_feature_compress=16, regnet_d_out=16,
                                num_sdf_layers=4, multires=6)
    rnet = R.GeneralRenderingNetwork(in_geometry_feat_ch=16, in_rendering_feat_ch=56, anti_alias_pooling=True)
    var = R.SingleVarianceNetwork(0.2)
    conf = Conf({"general.base_exp_dir": "/tmp", "model.h_patch_size": 3})
    renderer = R.SparseNeuSRenderer(None, sdfnet, var, rnet, 64, 64, 0, 1.0, alpha_type="div", conf=conf)
    return sdfnet, rnet, var, renderer
