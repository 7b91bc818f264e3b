"""How large is the source-view pixel patch a 32-point tile of k_color_pts touches?  (Round 4, VERDICT item 5: "stage that patch once per (tile, view) in
LDS and serve BOTH passes' taps from it" -- 8 KB of LDS per wave are left next to the operand blobs = 32 pixels x 256 B.)
For BASELINE config 2 (and the 32-view reference configuration) this takes the occupied-point list exactly as o2345_render_rays hands it to the colour
kernel (grouped by visibility signature), cuts it into 32-entry tiles, and for every (tile, view) pair whose view sees the tile's points measures the
bounding box of the bilinear taps: width x height in pixels, the number of DISTINCT pixels actually touched, and what fraction of the pairs a patch
buffer of 32 / 48 / 64 / 96 / 128 pixels would serve.  Host-side statistics with torch ops on the GPU; no kernel is modified."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
res = {}
for name, V, D, scale in (("config2_8views_512x512rays", 8, 128, 2), ("reference_32views_256x256rays", 32, 96, 1)):
    inp = bench.make_inputs(dev, V, 0, scale)
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    idx = bench.render_order_index(out["pm"])
    pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
    n_all = int(idx.numel())
    if out["pm"].numel() >= (1 << 20):
        idx = ops.list_sort_by_visibility(pts, idx, inp["proj"], 256, 256)
    P = pts[idx.long()]
    n = (P.shape[0] // 32) * 32
    P = P[:n].view(-1, 32, 3)                                       # [tiles, 32, 3]
    T = P.shape[0]
    stats = {"tiles": T, "occupied_points": n_all}
    H = W = 256
    box_px, uniq_px, wdt, hgt, npairs = [], [], [], [], 0
    for v in range(V):
        Pm = inp["proj"][v]                                          # [3,4]
        X = P @ Pm[:, :3].T + Pm[:, 3]
        Z = X[..., 2].clamp(min=1e-3)
        px, py = X[..., 0] / Z, X[..., 1] / Z                        # pixel coordinates (align_corners = True grid: gx = 2 px / (W - 1) - 1)
        vis = (px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)
        tile_vis = vis.any(1)
        x0 = torch.floor(px).clamp(0, W - 1); y0 = torch.floor(py).clamp(0, H - 1)
        big = torch.full_like(x0, 1e9)
        xmin = torch.where(vis, x0, big).min(1).values; xmax = torch.where(vis, x0 + 1, -big).max(1).values.clamp(max=W - 1)
        ymin = torch.where(vis, y0, big).min(1).values; ymax = torch.where(vis, y0 + 1, -big).max(1).values.clamp(max=H - 1)
        w_ = (xmax - xmin + 1)[tile_vis]; h_ = (ymax - ymin + 1)[tile_vis]
        box_px.append((w_ * h_).cpu()); wdt.append(w_.cpu()); hgt.append(h_.cpu())
        # distinct pixels touched by the 4 taps of the visible points of a tile
        lin = []
        for dx in (0, 1):
            for dy in (0, 1):
                lin.append(torch.where(vis, (y0 + dy).clamp(max=H - 1) * W + (x0 + dx).clamp(max=W - 1), torch.full_like(x0, -1.0)))
        lin = torch.cat(lin, 1).long()                               # [tiles, 128]
        srt = torch.sort(lin, 1).values
        uniq = ((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum(1) + (srt[:, 0] >= 0).long()
        uniq_px.append(uniq[tile_vis].cpu())
        npairs += int(tile_vis.sum())
    box = torch.cat(box_px).float(); uq = torch.cat(uniq_px).float(); wd = torch.cat(wdt).float(); hg = torch.cat(hgt).float()
    q = lambda t: [float(torch.quantile(t, x)) for x in (0.1, 0.5, 0.9, 0.99)]
    stats.update({"tile_view_pairs_evaluated": npairs, "bbox_pixels_q10_q50_q90_q99": q(box), "bbox_width_q10_q50_q90_q99": q(wd), "bbox_height_q10_q50_q90_q99": q(hg),
                  "distinct_pixels_q10_q50_q90_q99": q(uq), "taps_per_pair": 128,
                  "frac_pairs_bbox_fits": {str(c): float((box <= c).float().mean()) for c in (32, 48, 64, 96, 128, 256)},
                  "frac_pairs_distinct_fits": {str(c): float((uq <= c).float().mean()) for c in (32, 48, 64, 96, 128)},
                  "mean_distinct_pixels": float(uq.mean()), "mean_bbox_pixels": float(box.mean())})
    res[name] = stats
    vol = out = None
print(json.dumps(res))
