"""Stage timing of one render call with HIP events (diagnostic)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0, sdf_precision="fp32", color_precision="fp32")
inp = bench.make_inputs(dev, 8, 0, 2)
D = 128
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = ev(), ev(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), float(np.mean(ts))
full = lambda: pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
print("render", timed(full))
out = full()
pm = out["pm"].reshape(-1)
idx = bench.render_order_index(out["pm"])
print("valid points", idx.numel(), "of", pm.numel())
R = inp["rays_o"].shape[0]
pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
o2 = {"sdf": torch.empty(pts.shape[0], device=dev), "grad": torch.empty(pts.shape[0], 3, device=dev)}
print("sdf grad indexed", timed(lambda: ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, index=idx, out=o2, precision="fp32")))
print("sdf fwd  indexed", timed(lambda: ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=0, index=idx, out={"sdf": o2["sdf"]}, precision="fp32")))
print("color VALU indexed", timed(lambda: ops.color_points(wt.color_blob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=False)))
print("color MFMA indexed", timed(lambda: ops.color_points(wt.color_mblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=False, mfma=True)))
print("view_count", timed(lambda: ops.view_count(pts, vol["maskvol"], D, inp["proj"], 8, 256, 256)))
