"""A/B of the ORDER of the occupied-point list the colour kernel consumes (its results are scattered by slot, so the order is free): (a) what
o2345_render_rays does today (grouped by view-visibility signature, emission order inside a group: 32 consecutive rays of an image row at one sample
index per tile), (b) the same with 2-D ray blocks inside a signature group (8 x 4 rays at one sample index per tile), (c) 2-D blocks only.  Host-side
reordering with torch; times k_color_pts and k_sdf_grad_x3 on each list and checks the colours are bit-identical."""
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
for name, V, D, scale in (("config2_8views", 8, 128, 2), ("ref_32views", 32, 96, 1)):
    inp = bench.make_inputs(dev, V, 0, scale)
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    S, R = out["pm"].shape
    Wimg = 256 * scale
    idx0 = bench.render_order_index(out["pm"])
    pts = (inp["rays_o"][None] + inp["rays_d"][None] * out["mid_z"][..., None]).reshape(-1, 3).contiguous()
    srt, keys = ops.list_sort_by_visibility(pts, idx0, inp["proj"], 256, 256, want_keys=True)
    sig_of_slot = torch.zeros(S * R, dtype=torch.int64, device=dev)
    sig_of_slot[srt.long()] = keys.long() & 0xFFFFFFFF
    def order(bw, bh, use_sig, sample_major=True):
        sl = idx0.long()
        s_, r_ = sl // R, sl % R
        x, y = r_ % Wimg, r_ // Wimg
        blk = (y // bh) * (Wimg // bw) + (x // bw)
        inb = (y % bh) * bw + (x % bw)
        key = (blk * 256 + s_) * (bw * bh) + inb if sample_major else (s_ * (1 << 22) + blk) * (bw * bh) + inb
        if use_sig:
            key = sig_of_slot[sl] * (1 << 40) + key
        return idx0[torch.argsort(key)].contiguous()
    lists = {"a_signature_then_emission (today)": srt, "emission order (no sort)": idx0,
             "b_signature_then_8x4_blocks": order(8, 4, True), "b2_signature_then_4x8_blocks": order(4, 8, True), "b3_signature_then_16x2": order(16, 2, True),
             "c_8x4_blocks_only": order(8, 4, False), "d_signature_then_32x1_rows (= a with an explicit key)": order(32, 1, True)}
    o2 = {"sdf": torch.empty(pts.shape[0], device=dev), "grad": torch.empty(pts.shape[0], 3, device=dev)}
    ref_rgb = None
    r = {}
    for k, ii in lists.items():
        f = lambda: ops.color_points(wt.color_xblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=ii, want_nviews=False, mfma="x3")
        g = lambda: ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, index=ii, out=o2)
        f(); torch.cuda.synchronize()
        ts, tg = [], []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g(); b.record(); torch.cuda.synchronize(); tg.append(a.elapsed_time(b))
        st = ops.color_stats_buffer(dev)
        rgb, _ = ops.color_points(wt.color_xblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts, query_cam=inp["qcam"], index=ii, want_nviews=False, mfma="x3", stats=st)
        if ref_rgb is None: ref_rgb = rgb
        r[k] = {"color_ms": round(float(np.median(ts)), 2), "sdf_grad_ms": round(float(np.median(tg)), 2), "pairs_network": ops.color_stats_read(st)["pairs_network"],
                "bit_identical": bool(torch.equal(rgb, ref_rgb))}
    res[name] = r
    vol = out = None
print(json.dumps(res))
