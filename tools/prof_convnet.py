"""One FeatureNet + compress-layer pass at BASELINE config 2 (8 views 256^2) for rocprofv3 --pmc runs of the convolution kernels."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline = bench.pipeline
fn = __import__("importlib").import_module("one-2-3-45_amd.featurenet")
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, int(os.environ.get("VIEWS", "8")), 0, 2)
x = inp["imgs"].contiguous().float()
with torch.no_grad():
    for _ in range(3):
        _, cm = fn.fused_pyramid(wt.featurenet, x, want_cmaps=True, want_nchw=False)      # the pipeline's path: channel-last map only
        f = wt.compress.forward_nhwc(cm, nhwc_offset=3)
torch.cuda.synchronize()
print("ok", tuple(f.shape))
