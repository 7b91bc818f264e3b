"""FeatureNet + fused pyramid maps, all HIP (csrc/convnet.hip direct convolutions with InPlaceABN folded into producer / consumer,
csrc/featmaps.hip top-down path and pyramid).  Mirrors models/featurenet.py:12-91 and trainer_generic.py:1104-1125; the nn.Conv2d /
InPlaceABN modules only hold the parameters, so state-dict keys are identical to the reference's (``conv0.0.conv.weight``,
``conv0.0.bn.{weight,bias,running_mean,running_var}``, ...)."""
import torch
import torch.nn as nn

from . import ops


def _prepack_after_load(module, incompatible_keys):
    module.prepack()                      # (a load_state_dict post hook must return None)


def packed_weight(conv, precision=None):
    """The kernel packing of an nn.Conv2d's weight for the numerical mode ``precision`` (None = the global mode), cached ON the module that owns
    the parameter.  Re-packed when the parameter object, its version counter, its storage or the mode changes -- load_state_dict, .to(device), an
    optimiser step, any in-place op on the Parameter.  Writes through ``.data`` (``w.data.copy_(...)``, EMA / weight-surgery code) do NOT bump the
    version counter: call ``invalidate_packed(module)`` after such an update."""
    w = conv.weight
    key = (id(w), w._version, w.data_ptr(), str(w.device), ops.conv_x3(precision))
    if getattr(conv, "_o2345_packed_key", None) != key:
        conv._o2345_packed, conv._o2345_packed_key = ops.conv2d_pack(w.detach(), precision), key
    return conv._o2345_packed


def invalidate_packed(module):
    """Drop every cached weight packing below ``module`` (after parameter updates through ``.data``, which no version counter sees)."""
    for m in module.modules():
        for attr in ("_o2345_packed_key", "_blob_key", "_key", "_costreg_key"):      # conv packings, SDF blob, colour blobs, packed sparse CNN
            if hasattr(m, attr):
                setattr(m, attr, None)


def set_precision(module, precision):
    """Numerical mode of every convolution below ``module`` ("f16x3" | "fp32" | None = follow the global O2345_PRECISION)."""
    for m in module.modules():
        if isinstance(m, (ConvBnReLU, FeatureNet)):
            m.precision = precision
    return module


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
    """nn.Conv2d (no bias) + InPlaceABN as ONE direct HIP convolution (csrc/convnet.hip): the kernel writes the raw convolution output and
    reduces its batch statistics; the normalisation + leaky ReLU is applied by whoever reads that output next (``raw``), or by one extra
    pass when a caller wants the activated tensor itself (``forward`` / ``forward_nhwc``)."""

    def __init__(self, cin, cout, k=3, stride=1, pad=1):
        super().__init__()
        if pad != k // 2:
            raise NotImplementedError("o2345 ConvBnReLU: padding = kernel // 2 (as everywhere in the reference)")
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = InPlaceABN(cout)
        self.precision = None                             # None = the global mode; SceneWeights sets its own (set_precision)

    def raw(self, x, in_scale_shift=None, nhwc_offset=None):
        """-> (raw conv output, this layer's ABN (scale | shift)); ``in_scale_shift``: x is itself a raw output to be activated on load;
        ``nhwc_offset``: x is a channel-last map whose channels nhwc_offset .. + cin are the input."""
        bn = self.bn                                      # ops.conv2d raises for CPU tensors: there is no CPU fallback
        return ops.conv2d(x.contiguous().float(), self.conv.weight.detach(), None, self.conv.stride[0], in_scale_shift, bn.slope,
                          bn=(bn.weight.detach(), bn.bias.detach(), bn.eps, bn.abs_gamma), packed=packed_weight(self.conv, self.precision),
                          precision=self.precision, nhwc_offset=nhwc_offset)

    def forward(self, x):
        r, ss = self.raw(x)
        return ops.scale_shift_act(r, ss, self.bn.slope)[0]

    def forward_nhwc(self, x, nhwc_offset=None):
        """The activated output as a channel-last map [V,H,W,C] (what the cost-volume gather reads).  ``nhwc_offset``: the INPUT is channel-last too
        (the [V,H,W,64] colour map, features at channel 3)."""
        r, ss = self.raw(x, nhwc_offset=nhwc_offset)
        return ops.scale_shift_act(r, ss, self.bn.slope, want_nchw=False, want_nhwc=True)[1]


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
        self.precision = None
        # convolution weights are packed for the kernels when they are LOADED, not inside the first forward
        self.register_load_state_dict_post_hook(_prepack_after_load)

    def prepack(self):
        if self.toplayer.weight.is_cuda:
            with torch.cuda.device(self.toplayer.weight.device):
                for m in self.modules():
                    if isinstance(m, nn.Conv2d) and m is not self.lat1 and m is not self.lat0:      # the 1x1 laterals run inside fpn_level, unpacked
                        packed_weight(m, self.precision)
        return self

    def forward(self, x):
        """[V,3,H,W] -> [f2 (32 @ H/4), smooth1 (16 @ H/2), smooth0 (8 @ H)] (featurenet.py:68-91).  25 HIP launches, no library call: every
        convolution hands its raw output + ABN (scale | shift) to the next kernel, which normalises and activates on load."""
        r, ss = x, None
        raws = []
        for seq in (self.conv0, self.conv1, self.conv2):
            for blk in seq:
                r, ss = blk.raw(r, ss)
            raws.append((r, ss))
        (c0, ss0), (c1, ss1), (c2, ss2) = raws
        slope = self.conv0[0].bn.slope
        pr = self.precision
        f2, _ = ops.conv2d(c2, self.toplayer.weight.detach(), self.toplayer.bias.detach(), 1, ss2, slope, packed=packed_weight(self.toplayer, pr), precision=pr)
        # top-down path: lateral 1x1 convolution + x2 bilinear up-sampling + add, one kernel per level (csrc/featmaps.hip)
        f1 = ops.fpn_level(c1, f2, self.lat1.weight.detach(), self.lat1.bias.detach(), ss1, slope)
        f0 = ops.fpn_level(c0, f1, self.lat0.weight.detach(), self.lat0.bias.detach(), ss0, slope)
        s1, _ = ops.conv2d(f1, self.smooth1.weight.detach(), self.smooth1.bias.detach(), packed=packed_weight(self.smooth1, pr), precision=pr)
        s0, _ = ops.conv2d(f0, self.smooth0.weight.detach(), self.smooth0.bias.detach(), packed=packed_weight(self.smooth0, pr), precision=pr)
        return [f2, s1, s0]


def fused_pyramid(extractor, imgs, want_cmaps=False, want_nchw=True):
    """trainer_generic.py:1104-1125: [V,3,H,W] -> fused pyramid [V,56,H,W] (+ the channel-last colour map [V,H,W,64] = rgb | features | pad when
    want_cmaps): both written by ONE kernel (csrc/featmaps.hip) instead of two F.interpolate, a cat and a re-layout pass.  want_nchw=False skips the
    channel-first tensor (the pipeline's compress layer reads the features out of the colour map)."""
    f2, s1, s0 = extractor(imgs)
    fm, cm = ops.pyramid_pack(f2.contiguous(), s1.contiguous(), s0.contiguous(), imgs.contiguous().float(), want_nchw=want_nchw)
    return (fm, cm) if want_cmaps else fm
