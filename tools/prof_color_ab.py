"""One launch of each colour kernel on the config-2 occupied points (for rocprofv3 --pmc)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], 128, 2.0 / 127)
os.environ["O2345_COLOR_KERNEL"] = "tiles"
out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
idx = bench.render_order_index(out["pm"])
pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
for env in ({"O2345_COLOR_KERNEL": "tiles"}, {"O2345_COLOR_KERNEL": "pts"}):
    os.environ.update(env)
    ops.color_points(wt.color_xblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=False, mfma="x3")
    torch.cuda.synchronize()
print("done")
