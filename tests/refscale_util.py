"""HIP path vs the REFERENCE's own outputs at BASELINE scale: the measurements (test infrastructure, shared by tests/test_gpu_refscale.py, which asserts
on them, and bench.py's `parity_reference` block, which reports them).  The golden files come from tests/golden/make_golden_scale.py (the imported
reference modules, build container); nothing here touches the oracle except the marching-cubes checker of `field`."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.join(HERE, "golden") not in sys.path:
    sys.path.insert(0, os.path.join(HERE, "golden"))

pkg = importlib.import_module("one-2-3-45_amd")
ops = importlib.import_module("one-2-3-45_amd.ops")
pipeline = importlib.import_module("one-2-3-45_amd.pipeline")


def relerr(a, b, scale=None):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(1.0, float(np.abs(b).max()) if scale is None else scale))


def load(name, dev=None, precision=None):
    """Golden file `name` ("c1" | "c2" | "ref") + its regenerated seeded inputs + the HIP volume built from the images with the file's weights."""
    import make_golden_scale as MS
    cfg = MS.CONFIGS[name]
    g = np.load(os.path.join(HERE, "golden", cfg["name"]))
    sc, ro, rd, sel, chunk = MS.inputs(cfg)
    for k, v in MS.checksums(sc, ro, rd).items():          # the seeded inputs regenerate bit-identically on this machine
        assert v == g[k], f"input {k} differs from the one the golden file was generated on"
    assert np.array_equal(sel, g["ray_ids"]) and chunk == int(g["chunk"])
    dev = dev or torch.device("cuda:0")
    w = lambda p: {k[len("w:" + p):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:" + p)}
    wt = pipeline.SceneWeights.from_state_dicts(dev, w("sdf."), w("ren."), float(g["w:var.variance"]), featurenet_sd=w("fnet."),
                                                sdf_precision=precision, color_precision=precision)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    D = cfg["D"]
    vol = pipeline.build_volume(wt, T(sc["images"]), T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1))
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    scene = dict(sdf_blob=wt.sdf_blob, vol_cl=vol["vol_cl"], maskvol=vol["maskvol"], cmaps=vol["cmaps"], proj=proj, cam_pos=cam_pos,
                 color_mfma_blob=wt.color_mblob, color_x3_blob=wt.color_xblob, sdf_precision=wt.sdf_precision, color_precision=wt.color_precision)
    return dict(name=name, cfg=cfg, g=g, sc=sc, ro=ro, rd=rd, chunk=chunk, dev=dev, wt=wt, T=T, D=D, vol=vol, scene=scene, MS=MS,
                near=float(sc["query_near_far"][0]), far=float(sc["query_near_far"][1]), qcam=T(sc["query_c2w"][:3, 3].copy()))


def variants(G):
    g = G["g"]
    for vi in range(len(G["cfg"]["variance"])):
        pos = g[f"v{vi}_ray_pos"] if f"v{vi}_ray_pos" in g.files else np.arange(G["ro"].shape[0])
        yield vi, float(g[f"v{vi}_variance"]), pos


def volume(G):
    """get_conditional_volume from the IMAGES: every mask bit, samples of the fused pyramid / compressed maps / dense volume (max abs err / max |reference|)."""
    g, vol = G["g"], G["vol"]
    mask = (vol["maskvol"].view(-1) > 0).cpu().numpy()
    pi = torch.from_numpy(g["pix_idx"]).to(G["dev"])
    vi = torch.from_numpy(g["dense_idx"]).to(G["dev"])
    return {"kept_voxels": int(mask.sum()), "kept_voxels_reference": int(g["kept_voxels"]),
            "mask_bits_exact": bool(np.array_equal(np.packbits(mask), g["mask_bits"])) and int(vol["n_voxels"]) == int(g["kept_voxels"]),
            "fused_pyramid": relerr(vol["cmaps"].view(-1, 64)[pi][:, 3:59], g["fmaps_val"], float(g["fmaps_absmax"])),
            "compressed_maps": relerr(vol["feats_nhwc"].view(-1, 16)[pi], g["feats16_val"], float(g["feats16_absmax"])),
            "dense_volume": relerr(vol["vol_cl"].view(-1, 16)[vi], g["dense_val"], float(g["dense_absmax"])),
            "dense_volume_rms": float((vol["vol_cl"].view(-1, 16)[vi].cpu().double() - torch.from_numpy(g["dense_val"]).double()).pow(2).mean().sqrt() / float(g["dense_absmax"])),
            "dense_voxels_compared": int(vi.numel())}


def sampler(G, bin_frac=5e-3, floor=5e-7):
    """o2345_ray_upsample on the per-round (z, sdf) of the REFERENCE's first chunk vs the reference's new depths."""
    g, dev = G["g"], G["dev"]
    res = {"rounds": 0, "dz_max": 0.0, "dz_over_bin_max": 0.0, "excess_max": -1.0, "samples": 0}
    i = 0
    while f"v0_up{i}_new_z" in g.files:
        z, sdf, new_z = (torch.from_numpy(g[f"v0_up{i}_{k}"]) for k in ("z", "sdf", "new_z"))
        n = z.shape[0]
        nz, _, _ = ops.ray_upsample(G["T"](G["ro"][:n]), G["T"](G["rd"][:n]), z.t().contiguous().to(dev), sdf.t().contiguous().to(dev),
                                    float(g[f"v0_up{i}_inv_s"]), G["vol"]["maskvol"], G["D"], new_z.shape[1])
        dz = (nz.t().cpu() - new_z).abs()
        idx = (torch.searchsorted(z.contiguous(), new_z.contiguous(), right=True) - 1).clamp(0, z.shape[1] - 2)
        width = z.gather(1, idx + 1) - z.gather(1, idx)
        res["dz_max"] = max(res["dz_max"], float(dz.max()))
        res["dz_over_bin_max"] = max(res["dz_over_bin_max"], float((dz / width.clamp(min=1e-9)).max()))
        res["excess_max"] = max(res["excess_max"], float((dz - torch.maximum(bin_frac * width, torch.tensor(floor))).max()))
        res["samples"] += int(dz.numel())
        i += 1
    res["rounds"] = i
    return res


def core(G):
    """render_core on the REFERENCE's own sample lists (ops.render_core = the stage entries) vs the reference's results, chunk by chunk; one dict per variance."""
    g, dev, T, chunk = G["g"], G["dev"], G["T"], G["chunk"]
    sd = (G["far"] - G["near"]) / 64
    out = []
    for vi, variance, pos in variants(G):
        inv_s = float(np.clip(np.exp(10.0 * variance), 1e-6, 1e6))
        zr = torch.from_numpy(g[f"v{vi}_z_vals"])
        acc = {}
        for s in range(0, len(pos), chunk):
            p = pos[s:s + chunk]
            o = ops.render_core(G["scene"], T(G["ro"][p]), T(G["rd"][p]), zr[s:s + chunk].t().contiguous().to(dev), sd, inv_s, 1.0, 1.0, G["qcam"])
            for k in ("color", "depth", "weights_sum", "weights_max", "depth_var", "color_mask"):
                acc.setdefault(k, []).append(o[k].cpu())
            for k in ("weights", "sdf", "pm"):
                acc.setdefault(k, []).append(o[k].t().cpu())
            acc.setdefault("grad", []).append(o["grad"].permute(1, 0, 2).cpu())
        o = {k: torch.cat(v, 0) for k, v in acc.items()}
        e = dict(variance=variance, inv_s=inv_s, rays=int(len(pos)),
                 color=relerr(o["color"], g[f"v{vi}_color_fine"]), depth=relerr(o["depth"][:, None], g[f"v{vi}_depth"]),
                 weights_sum=relerr(o["weights_sum"][:, None], g[f"v{vi}_weights_sum"]), weights_max=relerr(o["weights_max"][:, None], g[f"v{vi}_weights_max"]),
                 depth_var=relerr(o["depth_var"][:, None], g[f"v{vi}_depth_variance"]))
        nw = g[f"v{vi}_weights"].shape[0]
        e["weights"] = relerr(o["weights"][:nw], g[f"v{vi}_weights"])
        nf = g[f"v{vi}_sdf"].shape[0]
        inside = torch.from_numpy(g[f"v{vi}_inside"])
        occ = inside > 0                           # the reference's defaults elsewhere: sdf = 100, gradient = 0 (:231) -- exactly
        sref, gref = torch.from_numpy(g[f"v{vi}_sdf"]), torch.from_numpy(g[f"v{vi}_gradients"])
        e["occupancy_exact"] = bool(torch.equal(o["pm"][:nf], inside))
        e["defaults_exact"] = bool((o["sdf"][:nf][~occ] == 100).all()) and bool((sref[~occ] == 100).all()) and float(o["grad"][:nf][~occ].abs().sum()) == 0
        e["sdf"] = relerr(o["sdf"][:nf][occ], sref[occ].numpy())
        e["grad"] = relerr(o["grad"][:nf][occ], gref[occ].numpy())
        e["color_mask_mismatches"] = int((o["color_mask"].bool()[:, None] != torch.from_numpy(g[f"v{vi}_color_fine_mask"])).sum())
        out.append(e)
    return out


def end_to_end(G):
    """render() per chunk exactly as the trainer's loop calls it vs the reference's images; one dict per variance."""
    g, T, chunk, wt = G["g"], G["T"], G["chunk"], G["wt"]
    out = []
    v0 = (wt.variance, wt.inv_s)
    try:
        for vi, variance, pos in variants(G):
            wt.variance, wt.inv_s = variance, float(np.clip(np.exp(10.0 * variance), 1e-6, 1e6))
            outs = []
            for s in range(0, len(pos), chunk):
                p = pos[s:s + chunk]
                o = pipeline.render(wt, G["vol"], G["scene"]["proj"], G["scene"]["cam_pos"], T(G["ro"][p]), T(G["rd"][p]), G["near"], G["far"], G["qcam"], want_z=True)
                outs.append(dict({k: o[k].cpu() for k in ("color", "depth", "weights_sum", "color_mask")}, z=o["z_vals"].t().cpu()))
            o = {k: torch.cat([x[k] for x in outs], 0) for k in outs[0]}
            zerr = (o["z"] - torch.from_numpy(g[f"v{vi}_z_vals"])).abs().max(1).values
            cerr = (o["color"] - torch.from_numpy(g[f"v{vi}_color_fine"])).abs().max(1).values
            derr = (o["depth"][:, None] - torch.from_numpy(g[f"v{vi}_depth"])).abs()[:, 0]
            same = zerr < 1e-6
            q = lambda t, x: float(torch.quantile(t, x))
            own = None
            pre = "selfsens0" if vi == 0 else f"v{vi}_selfsens0"
            if pre + "_color_err" in g.files:                     # the REFERENCE against itself on a volume that differs by fp32-class noise (make_golden_scale.SELFSENS_*)
                sc_, sz_ = torch.from_numpy(g[pre + "_color_err"]).reshape(-1), torch.from_numpy(g[pre + "_z_err"])
                own = {"volume_noise_sigma_over_absmax": float(g[pre + "_sigma"]), "color_err_q50_q90_q99_max": [q(sc_, 0.5), q(sc_, 0.9), q(sc_, 0.99), float(sc_.max())],
                       "frac_rays_color_gt_1e-3": float((sc_ > 1e-3).float().mean()), "z_err_max": float(sz_.max())}
            own1 = None
            pre1 = "selfsens1" if vi == 0 else f"v{vi}_selfsens1"
            if pre1 + "_color_err" in g.files:                    # the second stored level (SELFSENS_SIGMA[1] = 1e-6)
                s1 = torch.from_numpy(g[pre1 + "_color_err"]).reshape(-1)
                own1 = {"volume_noise_sigma_over_absmax": float(g[pre1 + "_sigma"]), "color_err_q50_q90_q99_max": [q(s1, 0.5), q(s1, 0.9), q(s1, 0.99), float(s1.max())]}
            out.append({"variance": variance, "inv_s": wt.inv_s, "rays": int(len(pos)), "coarse_spacing": (G["far"] - G["near"]) / 63,
                        "reference_vs_itself_on_a_noisy_volume": own, "reference_vs_itself_at_the_lower_noise_level": own1,
                        "color_err_q50_q90_q99_max": [q(cerr, 0.5), q(cerr, 0.9), q(cerr, 0.99), float(cerr.max())],
                        "depth_err_q50_q90_q99_max": [q(derr, 0.5), q(derr, 0.9), q(derr, 0.99), float(derr.max())],
                        "frac_rays_color_gt_1e-4": float((cerr > 1e-4).float().mean()), "frac_rays_color_gt_1e-3": float((cerr > 1e-3).float().mean()),
                        "z_err_max": float(zerr.max()), "rays_with_coinciding_lists": int(same.sum()),
                        "color_err_max_on_coinciding_lists": float(cerr[same].max()) if same.any() else 0.0,
                        "color_mask_mismatches": int((o["color_mask"].bool()[:, None] != torch.from_numpy(g[f"v{vi}_color_fine_mask"])).sum())})
    finally:
        wt.variance, wt.inv_s = v0
    return out


def field(G, check_mesh=True):
    """extract_fields vs the reference's u (config 1: the whole 64^3 lattice through the fused lattice kernel; otherwise the central 64^3 block of the 256^3
    lattice, points built like :887-889)."""
    g, dev, wt = G["g"], G["dev"], G["wt"]
    bmin, bmax, R = G["MS"].grid_box(G["cfg"])
    uref = torch.from_numpy(g["u"])
    assert np.array_equal(np.stack([bmin.numpy(), bmax.numpy()]), g["u_bounds"])
    if G["cfg"]["grid"][0] == "full":
        u = ops.sdf_mlp(wt.sdf_blob, G["vol"]["vol_cl"], None, variant=0, grid_R=R, sign=-1.0, precision=wt.sdf_precision, grid_tables=wt.grid_tables(R))["sdf"].view(R, R, R)
    else:
        ax = [torch.linspace(float(bmin[d]), float(bmax[d]), R) for d in range(3)]                         # :887-889
        pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3).contiguous().to(dev)
        u = ops.sdf_mlp(wt.sdf_blob, G["vol"]["vol_cl"], pts, variant=0, sign=-1.0, precision=wt.sdf_precision)["sdf"].view(R, R, R)
    u = u.cpu()
    flips = (u > 0) != (uref > 0)
    nflip = int(flips.sum())
    res = {"grid": R, "field_err_max": float((u - uref).abs().max()), "field_scale": float(uref.abs().max()), "inside_nodes_reference": int((uref > 0).sum()),
           "sign_flips": nflip, "abs_u_reference_at_flips_max": float(uref[flips].abs().max()) if nflip else 0.0}
    if nflip == 0 and check_mesh:
        from oracle import mc as omc                       # the checker's marching cubes on the REFERENCE's field (tests only: bench.py's block passes check_mesh=False)
        v, t = ops.marching_cubes(u.to(dev).contiguous(), 0.0)
        v_ref, t_ref = omc.marching_cubes(uref.numpy(), 0.0)
        res["triangles"] = int(t.shape[0])
        res["triangles_identical"] = bool(np.array_equal(t.cpu().numpy(), t_ref))
        res["vertex_shift_max_cells"] = float(np.abs(v.cpu().numpy() - v_ref).max()) if v.shape[0] == v_ref.shape[0] else None
    return res


def vertex_colours(G):
    """validate_colored_mesh's colouring (trainer_generic.py:1309-1363: compute_view_independent with the SDF gradient as the query direction + the rendering
    network) on vertices of the reference field's surface vs the reference's colours; valid-view counts exact."""
    g, wt, vol, sc_ = G["g"], G["wt"], G["vol"], G["scene"]
    if "vert_pts" not in g.files:
        return None
    pts = G["T"](g["vert_pts"])
    grad = ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, precision=wt.sdf_precision)["grad"]
    x3 = wt.color_precision == "f16x3"
    rgb, nv = ops.color_points(wt.color_xblob if x3 else wt.color_mblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], sc_["proj"], sc_["cam_pos"], pts, normals=grad,
                               mfma="x3" if x3 else True)
    want_nv = g["vert_mask"].sum(0).astype(np.uint8)
    return {"vertices": int(pts.shape[0]), "rgb": relerr(rgb, g["vert_rgb"]), "valid_view_count_mismatches": int((nv.cpu().numpy() != want_nv).sum()),
            "vertices_seen_by_two_or_more_views": int((want_nv >= 2).sum())}


def report(name, dev=None, precision=None):
    """Everything above for one golden file as one JSON-able dict (bench.py's `parity_reference` block).  Nothing under oracle/ is touched: the comparison is
    HIP against the stored outputs of the reference."""
    G = load(name, dev, precision)
    return {"golden": G["cfg"]["name"], "generator": "tests/golden/make_golden_scale.py (the imported reference modules, CPU)", "volume": volume(G),
            "sampler_on_reference_inputs": sampler(G), "render_core_on_reference_lists": core(G), "render_end_to_end": end_to_end(G), "extract_fields": field(G, check_mesh=False), "vertex_colours": vertex_colours(G)}


# ---- BASELINE config 5's sparse 256^3 level (tests/golden/ref_c5_lod1_sample.npz): the coarse-to-fine path through the MIRROR modules, like the trainer ----
def lod1(dev=None):
    """trainer_generic.py:437-491 on config 2's scene with recon.* : get_sdf_volume -> get_valid_sparse_coords_by_sdf -> x2 -> lod-1 get_conditional_volume
    (256^3 sparse) -> sdf() -> render() at lod 1, each against the reference's outputs.  -> dict of measured numbers."""
    import make_golden_scale as MS
    recon = importlib.import_module("one-2-3-45_amd.recon")
    fn = importlib.import_module("one-2-3-45_amd.featurenet")
    c5, cfg = MS.C5, MS.CONFIGS[MS.C5["base"]]
    g = np.load(os.path.join(HERE, "golden", c5["name"]))
    sc, ro, rd, sel, chunk = MS.inputs(cfg)
    for k, v in MS.checksums(sc, ro, rd).items():
        assert v == g[k], f"input {k} differs from the one the golden file was generated on"
    dev = dev or torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    w = lambda p: {k[len("w:" + p):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:" + p)}
    D, HW = cfg["D"], 256
    D1 = 2 * D

    class Conf(dict):
        def get_int(self, k, default=None):
            return int(self.get(k, default))
    mk = lambda lod, dims, comp: recon.SparseSdfNetwork(lod=lod, ch_in=56, voxel_size=2.0 / (dims - 1), vol_dims=[dims] * 3, hidden_dim=128, cost_type="variance_mean",
                                                        d_pyramid_feature_compress=comp, regnet_d_out=16, num_sdf_layers=4, multires=6).to(dev)
    sdf0, sdf1 = mk(0, D, 16), mk(1, D1, 8)
    for net, sd in ((sdf0, w("sdf.")), (sdf1, w("sdf1."))):
        miss = net.load_state_dict(sd, strict=False)
        assert not miss.unexpected_keys and all("running_" in k or "num_batches" in k for k in miss.missing_keys), miss
    fnet = fn.FeatureNet().to(dev)
    miss = fnet.load_state_dict(w("fnet."), strict=False)
    assert not miss.unexpected_keys
    rnet1 = recon.GeneralRenderingNetwork(16, 56, True).to(dev)
    rnet1.load_state_dict(w("ren1."))
    var1 = recon.SingleVarianceNetwork(0.2).to(dev)
    ren1 = recon.SparseNeuSRenderer(None, sdf1, var1, rnet1, 64, 64, 0, 1.0, alpha_type="div", conf=Conf({"general.base_exp_dir": "/tmp"}))
    res = {"golden": c5["name"]}
    with torch.no_grad():
        imgs = T(sc["images"])
        fmaps = fn.fused_pyramid(fnet, imgs)
        origin, aff = T(sc["partial_vol_origin"])[None], T(sc["affine_mats"])[None]
        cv0 = sdf0.get_conditional_volume(feature_maps=fmaps[None], partial_vol_origin=origin, proj_mats=aff, sizeH=HW, sizeW=HW, lod=0)
        dense, mask, lattice = cv0["dense_volume_scale0"], cv0["valid_mask_volume_scale0"], cv0["coords_scale0"]
        # ---- get_sdf_volume (sparse_sdf_network.py:441-474)
        sv = sdf0.get_sdf_volume(dense, mask, lattice, origin)
        res["l0_sdf_volume"] = relerr(sv.reshape(-1)[torch.from_numpy(g["l0_sdf_idx"]).to(dev)], g["l0_sdf_val"])
        below = (sv.reshape(-1).abs() < 0.02).cpu().numpy()
        ref_below = np.unpackbits(g["l0_sdf_bits_below_thr"])[: below.size].astype(bool)
        diff = np.nonzero(below != ref_below)[0]
        res["l0_voxels_below_threshold"] = [int(below.sum()), int(ref_below.sum())]
        res["l0_threshold_disagreements"] = int(diff.size)
        res["l0_threshold_disagreements_band"] = float((sv.reshape(-1)[torch.from_numpy(diff).to(dev)].abs() - 0.02).abs().max()) if diff.size else 0.0
        # ---- get_valid_sparse_coords_by_sdf (sparse_neus_renderer.py:822-879): the mirror's own selection vs the reference's
        pc, pf = ren1.get_valid_sparse_coords_by_sdf(sv[0], lattice[0], mask[0], dense[0])
        ref_pc = torch.from_numpy(g["pre_coords"].astype(np.int64))
        key = lambda c: (c[:, 0] * D + c[:, 1]) * D + c[:, 2]
        a_, b_ = set(key(pc[:, 1:].long().cpu()).tolist()), set(key(ref_pc).tolist())
        res["pruned_voxels"] = [len(a_), len(b_)]
        res["pruned_voxels_symmetric_difference"] = len(a_ ^ b_)
        res["pruned_order_equal"] = bool(len(a_ ^ b_) == 0 and torch.equal(pc[:, 1:].long().cpu(), ref_pc))
        # ---- the lod-1 volume on the REFERENCE's selection (the discrete part compared exactly), features from HIP's own lod-0 volume
        rp = ref_pc.to(dev)
        pf_ref = dense[0].reshape(16, -1).t()[key(rp)].contiguous()
        pc2 = torch.cat([torch.zeros(rp.shape[0], 1, device=dev), rp.float() * 2], 1)                        # trainer_generic.py:474
        cv1 = sdf1.get_conditional_volume(feature_maps=fmaps[None], partial_vol_origin=origin, proj_mats=aff, sizeH=HW, sizeW=HW, pre_coords=pc2, pre_feats=pf_ref)
        dense1, mask1 = cv1["dense_volume_scale1"], cv1["valid_mask_volume_scale1"]
        m1 = (mask1.reshape(-1) > 0).cpu().numpy()
        res["l1_kept_voxels"] = [int(m1.sum()), int(g["l1_kept_voxels"])]
        res["l1_mask_bits_exact"] = bool(np.array_equal(np.packbits(m1), g["l1_mask_bits"]))
        res["l1_dense_volume"] = relerr(dense1[0].reshape(16, -1)[:, torch.from_numpy(g["l1_dense_idx"]).to(dev)].t(), g["l1_dense_val"], float(g["l1_dense_absmax"]))
        res["l1_sdf"] = relerr(sdf1.sdf(T(g["l1_pts"]), dense1, 1)["sdf_pts_scale1"], g["l1_sdf"])
        # ---- one chunk of the lod-1 val loop
        pos = g["ray_pos"]
        kw = dict(background_rgb=1.0, alpha_inter_ratio=1.0, lod=1, conditional_volume=dense1, conditional_valid_mask_volume=mask1, feature_maps=fmaps,
                  color_maps=imgs, w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW], query_c2w=T(sc["query_c2w"])[None], if_render_with_grad=False)
        near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
        sd = (float(sc["query_near_far"][1]) - float(sc["query_near_far"][0])) / 64
        rc = ren1.render_core(T(ro[pos]), T(rd[pos]), T(g["v0_z_vals"]), sd, 1, sdf1, rnet1, **{k: v for k, v in kw.items() if k != "lod"})
        res["render_core_on_reference_lists"] = {
            "color": relerr(rc["color"], g["v0_color_fine"]), "depth": relerr(rc["depth"], g["v0_depth"]), "weights": relerr(rc["weights"], g["v0_weights"]),
            "weights_sum": relerr(rc["weights_sum"], g["v0_weights_sum"]),
            "color_mask_mismatches": int((rc["color_mask"].cpu() != torch.from_numpy(g["v0_color_fine_mask"])).sum())}
        seen = []
        real = ops.render_rays
        ops.render_rays = lambda *a, **k: (lambda o: (seen.append(o["z_vals"].t().cpu()), o)[1])(real(*a, **dict(k, want_z=True)))
        try:
            out = ren1.render(T(ro[pos]), T(rd[pos]), near, far, sdf1, rnet1, perturb_overwrite=0, **kw)
        finally:
            ops.render_rays = real
        cerr = (out["color_fine"].cpu() - torch.from_numpy(g["v0_color_fine"])).abs().max(1).values
        zerr = (seen[-1] - torch.from_numpy(g["v0_z_vals"])).abs().max(1).values
        own = torch.from_numpy(g["selfsens0_color_err"]).reshape(-1)
        q = lambda t, x: float(torch.quantile(t, x))
        same = zerr < 1e-6
        res["render_end_to_end"] = {"rays": int(len(pos)), "color_err_q50_q90_q99_max": [q(cerr, 0.5), q(cerr, 0.9), q(cerr, 0.99), float(cerr.max())],
                                    "reference_vs_itself_color_err_q50_q90_q99_max": [q(own, 0.5), q(own, 0.9), q(own, 0.99), float(own.max())],
                                    "z_err_max": float(zerr.max()), "reference_vs_itself_z_err_max": float(g["selfsens0_z_err"].max()),
                                    "rays_with_coinciding_lists": int(same.sum()), "color_err_max_on_coinciding_lists": float(cerr[same].max()) if same.any() else 0.0,
                                    "color_mask_mismatches": int((out["color_fine_mask"].cpu() != torch.from_numpy(g["v0_color_fine_mask"])).sum())}
    return res
