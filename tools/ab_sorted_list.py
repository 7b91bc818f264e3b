"""Does grouping the occupied-point list by view-visibility signature pay?  The colour kernel skips a (32-point tile, view) pair only when NO point of the tile sees
the view; with the list in ray order 80.7 % of the pairs are evaluated although only 72 % of the (point, view) pairs are visible.  Here the list is stably sorted
by the V-bit signature (torch, host-side experiment) and the network kernels are timed on both orders; results must be bit-identical (they are scattered by slot)."""
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
    p = pts[idx.long()]
    P = inp["proj"]
    key = torch.zeros(p.shape[0], dtype=torch.int64, device=dev)
    for v in range(V):
        pr = p @ P[v, :, :3].T + P[v, :, 3]
        z = pr[:, 2].clamp(min=1e-3)
        gx, gy = 2 * (pr[:, 0] / z) / 255 - 1, 2 * (pr[:, 1] / z) / 255 - 1
        key |= ((gx.abs() < 1) & (gy.abs() < 1)).long() << v
    res[f"V{V}_distinct_signatures"] = int(torch.unique(key).numel())
    res[f"V{V}_mean_visible_by_signature"] = float(sum(((key >> v) & 1).float().mean() for v in range(V)))
    order = torch.sort(key, stable=True)[1]
    idx_sorted = idx[order].contiguous()
    nt = idx.numel() // 32
    for name, ii, kk in (("ray_order", idx, key), ("sorted", idx_sorted, key[order])):
        bits = torch.stack([((kk >> v) & 1) for v in range(V)], 0)[:, :nt * 32].reshape(V, nt, 32)
        res[f"V{V}_{name}_tile_view_pairs_with_work(host)"] = float(bits.any(2).float().mean())
    blob = wt.color_xblob
    outs = {}
    for name, ii in (("ray_order", idx), ("sorted", idx_sorted), ("ray_order", idx), ("sorted", idx_sorted)):
        ops.color_stats(True)
        f = lambda: ops.color_points(blob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=ii, want_nviews=True, mfma="x3")
        rgb, nv = f(); torch.cuda.synchronize()
        st = ops.color_stats_read()
        ops.color_stats(False)
        ts = []
        for _ in range(3):
            a, b = ev(), ev(); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        res.setdefault(f"V{V}_color_{name}_ms", []).append(round(float(np.mean(ts)), 2))
        res[f"V{V}_color_{name}_stats"] = {k: int(v) for k, v in st.items()} if isinstance(st, dict) else [int(x) for x in st]
        outs[name] = (rgb, nv)
        o2 = {"sdf": torch.empty(pts.shape[0], device=dev), "grad": torch.empty(pts.shape[0], 3, device=dev)}
        g = lambda: ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, index=ii, out=o2, precision=wt.sdf_precision)
        g(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = ev(), ev(); a.record(); g(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        res.setdefault(f"V{V}_sdf_grad_{name}_ms", []).append(round(float(np.mean(ts)), 2))
        outs[name + "_sdf"] = (o2["sdf"].clone(), o2["grad"].clone())
    res[f"V{V}_color_bit_identical"] = bool(torch.equal(outs["ray_order"][0], outs["sorted"][0]) and torch.equal(outs["ray_order"][1], outs["sorted"][1]))
    m = idx.long()
    res[f"V{V}_sdf_grad_bit_identical"] = bool(torch.equal(outs["ray_order_sdf"][0][m], outs["sorted_sdf"][0][m]) and torch.equal(outs["ray_order_sdf"][1][m], outs["sorted_sdf"][1][m]))
print(json.dumps(res))
