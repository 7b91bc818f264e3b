"""Runs the colour / SDF kernels of one BASELINE-config-2 render a few times (for rocprofv3 --pmc passes)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
D = 128
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
pm = out["pm"].reshape(-1)
idx = bench.render_order_index(out["pm"])
pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
if b"list_sort=1" in ops._lib.lib().o2345_knobs():
    idx = ops.list_sort_by_visibility(pts, idx, inp["proj"], 256, 256)        # the order o2345_render_rays hands to the network kernels (csrc/list_sort.hip)
o2 = {"sdf": torch.empty(pts.shape[0], device=dev), "grad": torch.empty(pts.shape[0], 3, device=dev)}
V = inp["imgs"].shape[0]
import os
for _ in range(2):        # the kernels of the default (f16x3) mode, full-size launches
    ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, index=idx, out=o2, precision="f16x3")
    ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=0, out={"sdf": o2["sdf"]}, precision="f16x3")
    ops.color_points(wt.color_xblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=False, mfma="x3")
    ops.costvol_gather(vol["feats_nhwc"], inp["aff"], (D, D, D), 2.0 / (D - 1), inp["origin"], vol["cnt"], vol["coords"])
torch.cuda.synchronize()
print("done", idx.numel())
