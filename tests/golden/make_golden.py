"""Generate tests/golden/ref_small.npz by running the REFERENCE's own modules (imported from /root/reference on CPU,
stubs per SURVEY Appendix E) on a small seeded scene.  Run in the build container only:

    python tests/golden/make_golden.py

Inputs are regenerated from seeds (numpy Generator streams are stable across platforms); the file stores the network
parameters (reference state dicts), the seeds, input checksums and the reference outputs of every stage of the hot path.
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import as RI  # noqa: E402

pkg = importlib.import_module("one-2-3-45_amd")
CFG = dict(V=4, HW=40, D=20, seed=7, n_pts=1500, n_rays=40, grid_R=20)


def inputs(cfg=CFG):
    rng = np.random.default_rng(cfg["seed"])
    sc = pkg.synth.make_scene(cfg["V"], hw=(cfg["HW"], cfg["HW"]), image_seed=cfg["seed"])
    fmaps = rng.standard_normal((cfg["V"], 56, cfg["HW"], cfg["HW"])).astype(np.float32)
    pts = rng.uniform(-1.1, 1.1, (cfg["n_pts"], 3)).astype(np.float32)
    pts[:16] = np.array([[-1.0, 0.3, -0.2]], np.float32)
    pts[16:24] = 1.0
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], cfg["HW"], cfg["HW"])
    HW = cfg["HW"]
    ys, xs = rng.integers(HW // 4, 3 * HW // 4, cfg["n_rays"]), rng.integers(HW // 4, 3 * HW // 4, cfg["n_rays"])
    sel = ys * HW + xs
    return sc, fmaps, pts, ro[sel].copy(), rd[sel].copy()


def networks(cfg=CFG):
    """The seeded reference networks of the golden scene (same construction for every golden file)."""
    D = cfg["D"]
    sdfnet, rnet, var, renderer = RI.build_networks(D, seed=cfg["seed"])
    # the reference zero-initialises latent / PE columns: perturb them (seeded) so that every path is exercised
    g = torch.Generator().manual_seed(cfg["seed"])
    L = sdfnet.sdf_layer
    L.lin0.weight_v.data[:, 3:] += 0.003 * torch.randn(128, 36, generator=g)
    L.lin1.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
    L.lin2.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
    for m in sdfnet.sparse_costreg_net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = 1 + 0.2 * torch.randn(m.weight.shape, generator=g)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
    sdfnet.compress_layer.bn.weight.data = -(1 + 0.2 * torch.randn(16, generator=g))     # negative: exercises |gamma|
    sdfnet.compress_layer.bn.bias.data = 0.1 * torch.randn(16, generator=g)
    return sdfnet, rnet, var, renderer


def main_perturb(seed=1234):
    """tests/golden/ref_perturb.npz: the reference's DEFAULT val path -- render() with perturb = 1.0 (confs/one2345_lod0_val_demo.conf:127,
    trainer_generic.py:505-523 passes no perturb_overwrite) under torch.manual_seed(seed).  The reference draws the jitter with
    torch.rand(z_vals.shape) on the host generator (sparse_neus_renderer.py:506-515), first random draw of the call."""
    torch.set_grad_enabled(False)
    cfg = CFG
    sc, fmaps, pts, ro, rd = inputs()
    HW = cfg["HW"]
    sdfnet, rnet, var, renderer = networks(cfg)
    T = torch.from_numpy
    cv = sdfnet.get_conditional_volume(feature_maps=T(fmaps)[None], partial_vol_origin=T(sc["partial_vol_origin"])[None],
                                       proj_mats=T(sc["affine_mats"])[None], sizeH=HW, sizeW=HW, lod=0)
    dense, mask = cv["dense_volume_scale0"], cv["valid_mask_volume_scale0"]
    here = os.path.dirname(os.path.abspath(__file__))
    g0 = np.load(os.path.join(here, "ref_small.npz"))
    assert np.array_equal(g0["dense"], dense[0].numpy()) and np.array_equal(g0["mask"], mask[0, 0].numpy()), "scene differs from ref_small.npz"
    near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
    assert renderer.perturb == 1.0
    torch.manual_seed(seed)
    ren = renderer.render(T(ro), T(rd), near, far, sdfnet, rnet, background_rgb=1.0, alpha_inter_ratio=1.0,        # perturb_overwrite = -1
                          lod=0, conditional_volume=dense, conditional_valid_mask_volume=mask, feature_maps=T(fmaps),
                          color_maps=T(sc["images"]), w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW],
                          query_c2w=T(sc["query_c2w"])[None], if_render_with_grad=False)
    torch.manual_seed(seed)
    t_rand = torch.rand(len(ro), 64)
    out = {"seed": np.int64(seed), "t_rand": t_rand.numpy()}
    for k in ("color_fine", "depth", "weights", "gradients", "sdf", "weights_sum", "depth_variance", "cdf_fine", "color_fine_mask",
              "weights_max", "inside_sphere"):
        out["ren_" + k] = ren[k].numpy()
    out["ren_alpha_sum"] = np.float32(ren["alpha_sum"])
    path = os.path.join(here, "ref_perturb.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; colour differs from the deterministic render by",
          float(np.abs(out["ren_color_fine"] - g0["ren_color_fine"]).max()))


def main_perturb2(seed=4321):
    """tests/golden/ref_perturb2.npz: TWO consecutive 512-ray render() calls under ONE torch.manual_seed -- the first two chunks of the trainer's loop
    over the 40 x 40 query image (trainer_generic.py:503-524, perturb = 1.0 from the conf).  The reference draws from the HOST generator twice per call:
    t_rand = torch.rand(z_vals.shape) (sparse_neus_renderer.py:506-515) and pts_random = torch.rand([1024, 3]) (:606), so the second chunk's jitter depends
    on the first chunk having drawn both.  Stored: per call the sample lists handed to render_core, the per-ray results and sdf_random."""
    torch.set_grad_enabled(False)
    cfg = CFG
    sc, fmaps, pts, _, _ = inputs()
    HW = cfg["HW"]
    sdfnet, rnet, var, renderer = networks(cfg)
    T = torch.from_numpy
    here = os.path.dirname(os.path.abspath(__file__))
    g0 = np.load(os.path.join(here, "ref_small.npz"))
    dense, mask = T(g0["dense"])[None], T(g0["mask"])[None, None]
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], HW, HW)
    near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
    seen_z = []
    core = renderer.render_core
    renderer.render_core = lambda ro_, rd_, z_, *a, **k: (seen_z.append(z_.clone()), core(ro_, rd_, z_, *a, **k))[1]
    out = {"seed": np.int64(seed), "chk_rays": np.float64(rd.astype(np.float64).sum())}
    torch.manual_seed(seed)
    for c in range(2):
        s = slice(512 * c, 512 * (c + 1))
        ren = renderer.render(T(ro[s]), T(rd[s]), near, far, sdfnet, rnet, background_rgb=1.0, alpha_inter_ratio=1.0, lod=0,      # perturb_overwrite = -1
                              conditional_volume=dense, conditional_valid_mask_volume=mask, feature_maps=T(fmaps), color_maps=T(sc["images"]),
                              w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW], query_c2w=T(sc["query_c2w"])[None],
                              if_render_with_grad=False)
        for k in ("color_fine", "depth", "weights_sum", "sdf_random"):
            out[f"c{c}_{k}"] = ren[k].numpy()
        out[f"c{c}_z_vals"] = seen_z[-1].numpy()
    # the host stream the two calls consumed, restated: the goldens above must be reproducible from it
    torch.manual_seed(seed)
    for c in range(2):
        out[f"c{c}_t_rand"] = torch.rand(512, 64).numpy()
        out[f"c{c}_pts_random"] = (torch.rand([1024, 3]).float() * 2 - 1).numpy()
        chk = sdfnet.sdf(T(out[f"c{c}_pts_random"]), dense, lod=0)["sdf_pts_scale0"].numpy()
        assert np.array_equal(chk, out[f"c{c}_sdf_random"]), "the restated host stream is not what render() drew"
    path = os.path.join(here, "ref_perturb2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; second-chunk colour max", float(out["c1_color_fine"].max()))


TRAINED = ((0.62, 0.5, None), (0.45, 0.0, 1.0), (0.65, 1.0, None), (0.2, 0.5, None))     # (variance, alpha_inter_ratio, background_rgb)
TRAINED_SDF_SHIFT = -0.3      # added to the SDF output bias (sdf_layer.lin2.bias[0]): moves the zero level set of the seeded field into the rays'
                              # valid range (min SDF over the valid samples of ref_small.npz is 0.117), so that sharp sigmoids have a surface to find


def main_trained():
    """tests/golden/ref_trained.npz: the reference's render() in the regime of a TRAINED model -- SingleVarianceNetwork far from its
    init (variance 0.45 / 0.62 / 0.65 -> inv_s = exp(10 v) = 90 / 493 / 665, models/fields.py:179-186), the iter_step-driven
    alpha_inter_ratio of exp_runner_generic_blender_val.py:412-418 (0, 0.5, 1) and background_rgb None / 1.0 (train.use_white_bkgd) --
    on the ref_small.npz scene, deterministic sampling (perturb_overwrite = 0)."""
    torch.set_grad_enabled(False)
    cfg = CFG
    sc, fmaps, pts, ro, rd = inputs()
    HW = cfg["HW"]
    sdfnet, rnet, var, renderer = networks(cfg)
    T = torch.from_numpy
    here = os.path.dirname(os.path.abspath(__file__))
    g0 = np.load(os.path.join(here, "ref_small.npz"))
    dense, mask = T(g0["dense"])[None], T(g0["mask"])[None, None]
    sdfnet.sdf_layer.lin2.bias.data[0] += TRAINED_SDF_SHIFT
    near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
    out = {"sdf_shift": np.float64(TRAINED_SDF_SHIFT), "combos": np.array([[v, a, -1.0 if b is None else b] for v, a, b in TRAINED], np.float64)}
    seen_z = []
    core = renderer.render_core
    renderer.render_core = lambda ro_, rd_, z_, *a, **k: (seen_z.append(z_.clone()), core(ro_, rd_, z_, *a, **k))[1]     # record the sample lists
    for i, (v, air, bg) in enumerate(TRAINED):
        var.variance.data = torch.tensor(v)
        ren = renderer.render(T(ro), T(rd), near, far, sdfnet, rnet, perturb_overwrite=0, background_rgb=bg, alpha_inter_ratio=air,
                              lod=0, conditional_volume=dense, conditional_valid_mask_volume=mask, feature_maps=T(fmaps),
                              color_maps=T(sc["images"]), w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW],
                              query_c2w=T(sc["query_c2w"])[None], if_render_with_grad=False)
        for k in ("color_fine", "depth", "weights", "weights_sum", "depth_variance", "cdf_fine", "color_fine_mask", "weights_max"):
            out[f"c{i}_{k}"] = ren[k].numpy()
        out[f"c{i}_z_vals"] = seen_z[-1].numpy()                  # what render() handed to render_core (sparse_neus_renderer.py:559)
        print(f"variance {v} (inv_s {float(np.exp(10 * v)):.1f}) air {air} bg {bg}: weights_sum max {float(ren['weights_sum'].max()):.4f} "
              f"weights_max max {float(ren['weights_max'].max()):.4f}")
    path = os.path.join(here, "ref_trained.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_featurenet(seed=21):
    """tests/golden/ref_featurenet.npz: the reference's FeatureNet (models/featurenet.py:40-91, InPlaceABN = the functional stand-in of
    oracle/ref_import.py) and the fused pyramid of GenericTrainer.obtain_pyramid_feature_maps (trainer_generic.py:1117-1123) on seeded images."""
    torch.set_grad_enabled(False)
    import torch.nn.functional as F
    R = RI.load()
    from models.featurenet import FeatureNet
    torch.manual_seed(seed)
    net = FeatureNet()
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if type(m).__name__ == "InPlaceABN":
            m.weight.data = (1 + 0.2 * torch.randn(m.weight.shape, generator=g)) * torch.where(torch.rand(m.weight.shape, generator=g) < 0.2, -1.0, 1.0)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
    imgs = torch.rand(2, 3, 48, 64, generator=g)
    f2, s1, s0 = net(imgs)
    fused = torch.cat([F.interpolate(f2, scale_factor=4, mode="bilinear", align_corners=True),
                       F.interpolate(s1, scale_factor=2, mode="bilinear", align_corners=True), s0], dim=1)        # trainer_generic.py:1117-1123
    out = {"imgs": imgs.numpy(), "f2": f2.numpy(), "s1": s1.numpy(), "s0": s0.numpy(), "fused": fused.numpy()}
    for k, v in net.state_dict().items():
        if "running_" not in k and "num_batches" not in k:
            out["w:" + k] = v.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_featurenet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; fused", tuple(fused.shape), "max", float(fused.abs().max()))


def main():
    torch.set_grad_enabled(False)
    cfg = CFG
    sc, fmaps, pts, ro, rd = inputs()
    D, HW = cfg["D"], cfg["HW"]
    sdfnet, rnet, var, renderer = networks(cfg)
    T = torch.from_numpy
    out = {}
    cv = sdfnet.get_conditional_volume(feature_maps=T(fmaps)[None], partial_vol_origin=T(sc["partial_vol_origin"])[None],
                                       proj_mats=T(sc["affine_mats"])[None], sizeH=HW, sizeW=HW, lod=0)
    dense, mask = cv["dense_volume_scale0"], cv["valid_mask_volume_scale0"]
    out["dense"], out["mask"] = dense[0].numpy(), mask[0, 0].numpy()
    # intermediate: compressed feature maps + aggregated cost volume rows via the reference functions
    feats = sdfnet.compress_layer(T(fmaps))
    out["feats16"] = feats.numpy()
    r = sdfnet.sdf(T(pts).clone(), dense, 0)
    out["sdf"] = r["sdf_pts_scale0"].numpy(); out["sdf_feat"] = r["sdf_features_pts_scale0"].numpy(); out["latent"] = r["sampled_latent_scale0"].numpy()
    with torch.enable_grad():
        out["grad"] = sdfnet.gradient(T(pts).clone(), dense, 0).squeeze(1).detach().numpy()
    out["pts_mask"] = renderer.get_pts_mask_for_conditional_volume(T(pts), mask)[:, 0].numpy()
    near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
    ren = renderer.render(T(ro), T(rd), near, far, sdfnet, rnet, perturb_overwrite=0, background_rgb=1.0, alpha_inter_ratio=1.0,
                          lod=0, conditional_volume=dense, conditional_valid_mask_volume=mask, feature_maps=T(fmaps),
                          color_maps=T(sc["images"]), w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW],
                          query_c2w=T(sc["query_c2w"])[None], if_render_with_grad=False)
    for k in ("color_fine", "depth", "weights", "gradients", "sdf", "weights_sum", "depth_variance", "cdf_fine", "color_fine_mask",
              "weights_max", "inside_sphere"):
        out["ren_" + k] = ren[k].numpy()
    out["ren_alpha_sum"] = np.float32(ren["alpha_sum"]); out["ren_grad_err"] = np.float32(ren["gradient_error_fine"])
    R = cfg["grid_R"]
    out["u"] = renderer.extract_fields(torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), R, lambda p, **kw: sdfnet.sdf(p, **kw),
                                       "cpu", conditional_volume=dense, lod=0)
    # vertex colouring path on surface-ish points
    vp = T(pts[:400] * 0.6)
    with torch.enable_grad():
        geo, rf, rdiff, vm, _, _ = renderer.rendering_projector.compute_view_independent(
            vp.clone(), lod=0, geometryVolume=dense[0], geometryVolumeMask=mask[0], sdf_network=sdfnet, rendering_feature_maps=T(fmaps),
            color_maps=T(sc["images"]), w2cs=T(sc["w2cs"]), target_candidate_w2cs=None, intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW],
            query_img_idx=0, query_c2w=T(sc["query_c2w"])[None])
    vcol, _ = rnet(geo.detach(), rf.detach(), rdiff.detach(), vm)
    out["vert_pts"] = vp.numpy(); out["vert_rgb"] = vcol[0].detach().numpy(); out["vert_mask"] = vm[:, 0].numpy()
    # ---- lod 1 (coarse-to-fine): get_sdf_volume -> get_valid_sparse_coords_by_sdf -> lod-1 get_conditional_volume -------------
    R = RI.load()
    from models.sparse_sdf_network import SparseSdfNetwork
    origin = T(sc["partial_vol_origin"])[None]
    sv = sdfnet.get_sdf_volume(dense, mask, cv["coords_scale0"], origin)
    out["l1_sdf_volume"] = sv[0, 0].numpy()
    pc, pf = renderer.get_valid_sparse_coords_by_sdf(sv[0], cv["coords_scale0"][0], mask[0], dense[0], threshold=0.2, maximum_pts=700)
    out["l1_pre_coords"], out["l1_pre_feats"] = pc.numpy(), pf.numpy()
    torch.manual_seed(cfg["seed"] + 1)
    sdf1 = SparseSdfNetwork(lod=1, ch_in=56, voxel_size=2.0 / (2 * D - 1), vol_dims=[2 * D] * 3, hidden_dim=128, cost_type="variance_mean",
                            d_pyramid_feature_compress=8, regnet_d_out=16, num_sdf_layers=4, multires=6)
    g1 = torch.Generator().manual_seed(cfg["seed"] + 1)
    sdf1.sdf_layer.lin1.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g1)
    sdf1.sdf_layer.lin2.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g1)
    for m in sdf1.sparse_costreg_net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = 1 + 0.2 * torch.randn(m.weight.shape, generator=g1)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g1)
    pc2 = pc.clone()
    pc2[:, 1:] = pc2[:, 1:] * 2                                   # trainer_generic.py:889
    cv1 = sdf1.get_conditional_volume(feature_maps=T(fmaps)[None], partial_vol_origin=origin, proj_mats=T(sc["affine_mats"])[None],
                                      sizeH=HW, sizeW=HW, pre_coords=pc2, pre_feats=pf)
    out["l1_dense"], out["l1_mask"] = cv1["dense_volume_scale1"][0].numpy(), cv1["valid_mask_volume_scale1"][0, 0].numpy()
    out["l1_sdf"] = sdf1.sdf(T(pts).clone(), cv1["dense_volume_scale1"], 1)["sdf_pts_scale1"].numpy()
    print("lod1: pruned", pc.shape[0], "children kept", int(out["l1_mask"].sum()))
    sd = {}
    for k, v in sdf1.state_dict().items():
        if "num_batches_tracked" not in k and "running_" not in k:
            sd["w:sdf1." + k] = v.numpy()
    for prefix, net in (("sdf.", sdfnet), ("ren.", rnet), ("var.", var)):
        for k, v in net.state_dict().items():
            if "num_batches_tracked" in k or "running_" in k:
                continue
            sd["w:" + prefix + k] = v.numpy()
    chk = {"chk_fmaps": np.float64(fmaps.astype(np.float64).sum()), "chk_pts": np.float64(pts.astype(np.float64).sum()),
           "chk_rays": np.float64(rd.astype(np.float64).sum()), "chk_aff": np.float64(sc["affine_mats"].astype(np.float64).sum())}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_small.npz")
    np.savez_compressed(path, **out, **sd, **chk, **{"cfg_" + k: np.int64(v) for k, v in cfg.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; weights_sum max", float(ren["weights_sum"].max()),
          "valid voxels", int(mask.sum()))


if __name__ == "__main__":
    if "--perturb" in sys.argv:
        main_perturb()
    elif "--perturb2" in sys.argv:
        main_perturb2()
    elif "--featurenet" in sys.argv:
        main_featurenet()
    elif "--trained" in sys.argv:
        main_trained()
    else:
        main()
