"""FeatureNet + fused pyramid maps on stock PyTorch-ROCm (MIOpen convolutions), with InPlaceABN replaced by the HIP
batch-statistics op.  Mirrors models/featurenet.py:12-91 and trainer_generic.py:1104-1125; state-dict keys are
identical to the reference's (``conv0.0.conv.weight``, ``conv0.0.bn.{weight,bias,running_mean,running_var}``, ...)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class InPlaceABN(nn.Module):
    """Drop-in for inplace_abn.InPlaceABN (leaky_relu 0.01): training-mode batch statistics, as the reference runs it
    (the runner never calls .eval()).  CUDA tensors go through libo2345_hip; there is no CPU path."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu", activation_param=0.01):
        super().__init__()
        assert affine and activation == "leaky_relu"
        self.num_features, self.eps, self.momentum, self.slope = num_features, eps, momentum, activation_param
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.abs_gamma = True          # inplace_abn uses |gamma| + eps

    def forward(self, x, want_nhwc=False):
        if not x.is_cuda:
            raise RuntimeError("o2345 InPlaceABN: HIP-only op (no CPU fallback)")
        x = x.contiguous()
        if x.shape[1] not in (8, 16, 32):
            raise NotImplementedError(f"o2345 InPlaceABN: 8, 16 or 32 channels (FeatureNet / compress layer), got {x.shape[1]}")
        y, y_nhwc = ops.abn_nchw(x, self.weight.detach(), self.bias.detach(), self.eps, self.slope, self.abs_gamma,
                                 want_nchw=True, want_nhwc=want_nhwc)
        return (y, y_nhwc) if want_nhwc else y


class ConvBnReLU(nn.Module):
    def __init__(self, cin, cout, k=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = InPlaceABN(cout)

    def forward(self, x):
        return self.bn(self.conv(x))


class FeatureNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = nn.Sequential(ConvBnReLU(3, 8), ConvBnReLU(8, 8))
        self.conv1 = nn.Sequential(ConvBnReLU(8, 16, 5, 2, 2), ConvBnReLU(16, 16), ConvBnReLU(16, 16))
        self.conv2 = nn.Sequential(ConvBnReLU(16, 32, 5, 2, 2), ConvBnReLU(32, 32), ConvBnReLU(32, 32))
        self.toplayer = nn.Conv2d(32, 32, 1)
        self.lat1 = nn.Conv2d(16, 32, 1)
        self.lat0 = nn.Conv2d(8, 32, 1)
        self.smooth1 = nn.Conv2d(32, 16, 3, padding=1)
        self.smooth0 = nn.Conv2d(32, 8, 3, padding=1)

    @staticmethod
    def _up_add(x, y):
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) + y

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        f2 = self.toplayer(c2).contiguous()
        # top-down path: lateral 1x1 convolution + x2 bilinear up-sampling + add, one HIP kernel per level (csrc/featmaps.hip)
        f1 = ops.fpn_level(c1.contiguous(), f2, self.lat1.weight.detach(), self.lat1.bias.detach())
        f0 = ops.fpn_level(c0.contiguous(), f1, self.lat0.weight.detach(), self.lat0.bias.detach())
        return [f2, self.smooth1(f1), self.smooth0(f0)]


def fused_pyramid(extractor, imgs, want_cmaps=False):
    """trainer_generic.py:1104-1125: [V,3,H,W] -> fused pyramid [V,56,H,W] (+ the channel-last colour map [V,H,W,64] = rgb | features | pad when
    want_cmaps): both written by ONE kernel (csrc/featmaps.hip) instead of two F.interpolate, a cat and a re-layout pass."""
    f2, s1, s0 = extractor(imgs)
    fm, cm = ops.pyramid_pack(f2.contiguous(), s1.contiguous(), s0.contiguous(), imgs.contiguous().float())
    return (fm, cm) if want_cmaps else fm
