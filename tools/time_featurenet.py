"""Layer-by-layer steady-state timing of FeatureNet + pyramid + compress layer (diagnostic)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline = bench.pipeline
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
x = inp["imgs"]
net = wt.featurenet
def timed(fn_, reps=5):
    fn_(); fn_(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); r = fn_(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), r
tot = 0.0
with torch.no_grad():
    cur = x
    outs = {}
    for name, seq in (("conv0", net.conv0), ("conv1", net.conv1), ("conv2", net.conv2)):
        for i, blk in enumerate(seq):
            tc, y = timed(lambda: blk.conv(cur))
            tb, y2 = timed(lambda: blk.bn(y))
            print(f"{name}.{i}  conv {tuple(blk.conv.weight.shape)} s{blk.conv.stride[0]}: {tc*1e3:7.1f} us   abn: {tb*1e3:6.1f} us   out {tuple(y2.shape)}")
            tot += tc + tb
            cur = y2
        outs[name] = cur
    for nm, m, src in (("toplayer", net.toplayer, outs["conv2"]), ("lat1", net.lat1, outs["conv1"]), ("lat0", net.lat0, outs["conv0"])):
        t, _ = timed(lambda: m(src)); print(f"{nm}: {t*1e3:7.1f} us"); tot += t
    f2 = net.toplayer(outs["conv2"])
    t, f1 = timed(lambda: net._up_add(f2, net.lat1(outs["conv1"]))); print(f"up_add1 (incl lat1): {t*1e3:7.1f} us")
    t, f0 = timed(lambda: net._up_add(f1, net.lat0(outs["conv0"]))); print(f"up_add0 (incl lat0): {t*1e3:7.1f} us")
    t, s1 = timed(lambda: net.smooth1(f1)); print(f"smooth1: {t*1e3:7.1f} us"); tot += t
    t, s0 = timed(lambda: net.smooth0(f0)); print(f"smooth0: {t*1e3:7.1f} us"); tot += t
    t, fm = timed(lambda: torch.cat([F.interpolate(f2, scale_factor=4, mode="bilinear", align_corners=True), F.interpolate(s1, scale_factor=2, mode="bilinear", align_corners=True), s0], dim=1))
    print(f"pyramid fuse (2 interpolate + cat): {t*1e3:7.1f} us")
    t, _ = timed(lambda: pipeline.ops.pack_color_maps(fm.contiguous(), x.contiguous())); print(f"pack_color_maps: {t*1e3:7.1f} us")
    t, pre = timed(lambda: wt.compress.conv(fm)); print(f"compress conv: {t*1e3:7.1f} us")
    t, _ = timed(lambda: wt.compress.bn(pre.contiguous(), want_nhwc=True)); print(f"compress abn: {t*1e3:7.1f} us")
    t, _ = timed(lambda: pipeline.featurenet_forward(wt, x) if hasattr(pipeline, "featurenet_forward") else __import__("importlib").import_module("one-2-3-45_amd.featurenet").fused_pyramid(net, x)); print(f"featurenet + pyramid total: {t*1e3:7.1f} us")
