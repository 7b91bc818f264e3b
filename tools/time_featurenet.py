"""Steady-state timing of FeatureNet + pyramid + compress layer on the HIP convolution engine (diagnostic): per layer and whole."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
fn = __import__("importlib").import_module("one-2-3-45_amd.featurenet")
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
x = inp["imgs"].contiguous().float()
net = wt.featurenet


def timed(f, reps=7):
    f(); f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); r = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts) * 1e3, r


with torch.no_grad():
    r, ss = x, None
    raws = []
    for name, seq in (("conv0", net.conv0), ("conv1", net.conv1), ("conv2", net.conv2)):
        for i, blk in enumerate(seq):
            t, (r2, ss2) = timed(lambda: blk.raw(r, ss))
            w = blk.conv.weight
            flop = 2 * r2.numel() * w.shape[1] * w.shape[2] * w.shape[3]
            print(f"{name}.{i} {tuple(w.shape)} s{blk.conv.stride[0]} -> {tuple(r2.shape)}: {t:7.1f} us  ({flop / t / 1e6:6.1f} TFLOP/s)")
            r, ss = r2, ss2
        raws.append((r, ss))
    (c0, ss0), (c1, ss1), (c2, ss2_) = raws
    t, (f2, _) = timed(lambda: ops.conv2d(c2, net.toplayer.weight.detach(), net.toplayer.bias.detach(), 1, ss2_, 0.01, packed=fn.packed_weight(net.toplayer))); print(f"toplayer: {t:7.1f} us")
    t, f1 = timed(lambda: ops.fpn_level(c1, f2, net.lat1.weight.detach(), net.lat1.bias.detach(), ss1, 0.01)); print(f"fpn level 1: {t:7.1f} us")
    t, f0 = timed(lambda: ops.fpn_level(c0, f1, net.lat0.weight.detach(), net.lat0.bias.detach(), ss0, 0.01)); print(f"fpn level 0: {t:7.1f} us")
    t, (s1, _) = timed(lambda: ops.conv2d(f1, net.smooth1.weight.detach(), net.smooth1.bias.detach(), packed=fn.packed_weight(net.smooth1))); print(f"smooth1: {t:7.1f} us")
    t, (s0, _) = timed(lambda: ops.conv2d(f0, net.smooth0.weight.detach(), net.smooth0.bias.detach(), packed=fn.packed_weight(net.smooth0))); print(f"smooth0: {t:7.1f} us")
    t, (fm, cm) = timed(lambda: ops.pyramid_pack(f2, s1, s0, x)); print(f"pyramid_pack: {t:7.1f} us")
    t, (pre, ssc) = timed(lambda: wt.compress.raw(fm)); print(f"compress conv 56->16: {t:7.1f} us  ({2 * pre.numel() * 56 * 9 / t / 1e6:6.1f} TFLOP/s)")
    t, _ = timed(lambda: ops.scale_shift_act(pre, ssc, 0.01, want_nchw=False, want_nhwc=True)); print(f"compress ABN apply + NHWC: {t:7.1f} us")
    t, _ = timed(lambda: fn.fused_pyramid(net, x, want_cmaps=True)); print(f"featurenet + pyramid total: {t:7.1f} us")
    t, _ = timed(lambda: wt.compress.forward_nhwc(fm)); print(f"compress layer total: {t:7.1f} us")
    t, _ = timed(lambda: pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], 128, 2.0 / 127)); print(f"build_volume total: {t:7.1f} us")
