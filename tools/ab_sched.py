"""A/B of the colour kernels and their scheduling knobs on the occupied points of one render (BASELINE config 2: 8 views; the reference's
configuration: 32 views).  O2345_COLOR_SCHED bits: 1 = static wave priority by SIMD slot, 2 = priority 3 while gathering, 4 = evaluate EVERY view
(no wave-uniform skipping of views that see none of a tile's points: the round-2 kernel), 8 = block-interleaved tile schedule.  Every variant of a
kernel must return bit-identical colours; the two kernels agree to rounding."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
ev = lambda: torch.cuda.Event(enable_timing=True)
res = {}
for V, D, scale in ((8, 128, 2), (32, 96, 1)):
    wt = pipeline.SceneWeights(dev, seed=0)
    inp = bench.make_inputs(dev, V, 0, scale)
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    idx = bench.render_order_index(out["pm"])
    pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
    base = {}
    for kern, scheds in (("tiles", (0, 2)), ("pts", (4, 6, 0, 2, 8, 10, 0))):
        os.environ["O2345_COLOR_KERNEL"] = kern
        for sched in scheds:
            os.environ["O2345_COLOR_SCHED"] = str(sched)
            f = lambda: ops.color_points(wt.color_xblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=idx, want_nviews=True, mfma="x3")
            rgb, nv = f(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                a, b = ev(), ev(); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            if kern not in base:
                base[kern] = (rgb, nv)
            res.setdefault(f"V{V}_{kern}_sched{sched}_ms", []).append(round(float(np.mean(ts)), 2))
            assert torch.equal(rgb, base[kern][0]) and torch.equal(nv, base[kern][1]), (V, kern, sched)
    res[f"V{V}_pts_vs_tiles_maxdiff"] = float((base["pts"][0] - base["tiles"][0]).abs().max())
    res[f"V{V}_nviews_equal"] = bool(torch.equal(base["pts"][1], base["tiles"][1]))
    nvl = base["pts"][1][idx.long()].float()
    res[f"V{V}_points"] = int(idx.numel()); res[f"V{V}_mean_visible_views"] = float(nvl.mean()); res[f"V{V}_points_without_view"] = int((nvl == 0).sum())
    # (tile-level / point-level visibility statistics: tools/ab_sorted_list.py and the kernel's own work counters, ops.color_stats)
print(json.dumps(res))
