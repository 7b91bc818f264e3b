"""A/B of the two colour kernels on the occupied points of one BASELINE-config-2 render (same process, same data):
O2345_COLOR_KERNEL=tiles -> k_color_mfma (columns = (point, view)), default -> k_color_pts (columns = points)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
res = {}
for V in (8, 32):
    wt = pipeline.SceneWeights(dev, seed=0)
    inp = bench.make_inputs(dev, V, 0, 2 if V == 8 else 1)
    D = 128 if V == 8 else 96
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    idx = bench.render_order_index(out["pm"])
    pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    rgbs = {}
    for prec, blob, mode in (("f16x3", wt.color_xblob, "x3"), ("fp32", wt.color_mblob, True)):
        for kern in ("tiles", "pts"):
            os.environ["O2345_COLOR_KERNEL"] = kern
            f = lambda: ops.color_points(blob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=True, mfma=mode)
            rgb, nv = f(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                a, b = ev(), ev(); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            rgbs[(prec, kern)] = (rgb, nv)
            res[f"V{V}_{prec}_{kern}_ms"] = float(np.mean(ts))
        d = (rgbs[(prec, "tiles")][0] - rgbs[(prec, "pts")][0]).abs()
        res[f"V{V}_{prec}_maxdiff"] = float(d.max()); res[f"V{V}_{prec}_meandiff"] = float(d.mean())
        res[f"V{V}_{prec}_nv_mismatch"] = int((rgbs[(prec, "tiles")][1] != rgbs[(prec, "pts")][1]).sum())
    res[f"V{V}_points"] = int(idx.numel())
print(json.dumps(res))
