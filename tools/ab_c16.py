"""A/B of k_color_pts (32 points per wave, two gather passes) and k_color_c16 (16 points per wave, per-view inputs cached between the passes) on the
occupied points of one BASELINE-config-2 render; the two must agree to rounding."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops, pkg = bench.pipeline, bench.ops, bench.pkg
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
c16 = torch.from_numpy(pkg.weights.pack_color_c16_blob(wt.color_sd)).to(dev)
inp = bench.make_inputs(dev, 8, 0, 2)
D = 128
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
idx = bench.render_order_index(out["pm"])
pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
ev = lambda: torch.cuda.Event(enable_timing=True)
res = {}
outs = {}
for name, blob, mode in (("pts", wt.color_xblob, "x3"), ("c16", c16, "c16"), ("pts", wt.color_xblob, "x3"), ("c16", c16, "c16")):
    f = lambda: ops.color_points(blob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=True, mfma=mode)
    rgb, nv = f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = ev(), ev(); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    res.setdefault(name + "_ms", []).append(round(float(np.mean(ts)), 2))
    outs[name] = (rgb, nv)
    ops.color_stats(True); f(); res[name + "_work"] = ops.color_stats_read(); ops.color_stats(False)
res["maxdiff"] = float((outs["pts"][0] - outs["c16"][0]).abs().max())
res["nviews_equal"] = bool(torch.equal(outs["pts"][1], outs["c16"][1]))
print(json.dumps(res))
