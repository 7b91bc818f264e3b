"""Reference-generated golden vectors AT BASELINE SCALE: the REFERENCE's own modules (imported from /root/reference on CPU,
stubs per SURVEY Appendix E / oracle/ref_import.py) run the whole hot path -- FeatureNet -> fused pyramid -> get_conditional_volume -> render in the
runner's 512-ray chunks -> extract_fields -- on the configurations BASELINE.json names.  Build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden_scale.py c1        # -> ref_c1.npz          BASELINE config 1, exactly: V = 8 ring, 64^3, 64 seeded rays, 64 + 64, perturb 0
    python tests/golden/make_golden_scale.py c2        # -> ref_c2_sample.npz   BASELINE config 2: 128^3, 8 x 512-ray chunks (= 8 rows) of the 512^2 image
    python tests/golden/make_golden_scale.py ref       # -> ref_refcfg_sample.npz   the reference configuration: V = 32, 96^3, 2 chunks of the 256^2 val image
    python tests/golden/make_golden_scale.py c5        # -> ref_c5_lod1_sample.npz  BASELINE config 5's sparse 256^3 level: the coarse-to-fine path on config 2's scene

A file stores seeds, the networks' state dicts and OUTPUTS only: the images / cameras / rays are regenerated from the seeds by `inputs()` on the machine
that runs the comparison (numpy Generator streams and the closed-form camera rig are platform-stable; checksums are stored and checked).  Nothing here is
product code; the HIP path is compared with these files in tests/test_gpu_refscale.py and in bench.py's parity_fullsize block.
"""
import hashlib
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

pkg = importlib.import_module("one-2-3-45_amd")

CHUNK = 512                      # the runner's ray batch (trainer_generic.py:503: rays_o.split(self.batch_size)), conf val batch_size 512
CONFIGS = {
    # rays: "c1" = 64 pixels randint(64, 192) seed 0 of the 256^2 query image; "rows" = whole rows of the query image (one row of the
    # 512^2 image = one 512-ray chunk of the trainer's loop; two rows of the 256^2 image = one chunk)
    "c1": dict(name="ref_c1.npz", V=8, D=64, image_seed=0, net_seed=11, rays="c1", ray_scale=1, grid=("full", 64), n_dense=20000,
               variance=(0.2,)),
    "c2": dict(name="ref_c2_sample.npz", V=8, D=128, image_seed=3, net_seed=12, rays="rows", ray_scale=2, rows=(40, 104, 168, 232, 296, 360, 424, 488),
               grid=("block", 256, (96, 96, 96), 64), n_dense=100000, variance=(0.2, 0.5), trained_rows=(232, 296)),
    "ref": dict(name="ref_refcfg_sample.npz", V=32, D=96, image_seed=5, net_seed=13, rays="rows", ray_scale=1, rows=(100, 101, 140, 141),
                grid=("block", 256, (96, 96, 96), 64), n_dense=20000, variance=(0.2,)),
}
UPSAMPLE_TRACE_RAYS = 512        # the sampler's per-round inputs / outputs are kept for the first chunk only
# The reference's OWN end-to-end sensitivity: render() again on a latent volume that differs from its own by Gaussian noise of SELFSENS_SIGMA x max|volume|
# (3e-6: max ~ 1.5e-5 over the volume -- the class of difference any other fp32 implementation of FeatureNet + the sparse CNN has against it; the HIP
# build measures 0.9 - 1.4e-5 max; 1e-6 for scale).  The hierarchical sampler amplifies such differences into O(1e-2) colour differences on some rays:
# the end-to-end test bounds HIP-vs-reference by this reference-vs-reference distribution.
SELFSENS_SIGMA = (3e-6, 1e-6)
SELFSENS_SEEDS = (1, 2, 3)


def inputs(cfg):
    """Seeded inputs of one configuration: scene dict (pkg.synth.make_scene), rays [R,3] x 2 in chunk order, chunk size."""
    sc = pkg.synth.make_scene(cfg["V"], image_seed=cfg["image_seed"])
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=cfg["ray_scale"])
    W = 256 * cfg["ray_scale"]
    if cfg["rays"] == "c1":
        rng = np.random.default_rng(0)
        ys, xs = rng.integers(64, 192, 64), rng.integers(64, 192, 64)
        sel = ys * W + xs
        chunk = 64
    else:
        sel = np.concatenate([np.arange(r * W, (r + 1) * W) for r in cfg["rows"]])
        chunk = CHUNK
    return sc, ro[sel].copy(), rd[sel].copy(), sel, chunk


def checksums(sc, ro, rd):
    f = lambda a: np.float64(np.asarray(a, np.float64).sum())
    return {"chk_images": f(sc["images"]), "chk_aff": f(sc["affine_mats"]), "chk_rays_o": f(ro), "chk_rays_d": f(rd), "chk_w2cs": f(sc["w2cs"])}


def grid_box(cfg):
    """-> (bound_min [3], bound_max [3], resolution) of the extract_fields call of this configuration, or None."""
    g = cfg["grid"]
    if g is None:
        return None
    if g[0] == "full":
        return torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), g[1]
    _, R, origin, B = g
    lin = torch.linspace(-1, 1, R)                                   # nodes of the R^3 extraction lattice: the block's corners are lattice nodes
    o = torch.tensor(origin)
    return lin[o], lin[o + B - 1], B


def build_reference_networks(cfg):
    """The reference's FeatureNet + lod-0 networks, seeded, with every zero-initialised path perturbed (make_golden.networks) and some negative ABN gammas."""
    import make_golden as MG
    from oracle import ref_import as RI
    RI.load()
    from models.featurenet import FeatureNet
    sdfnet, rnet, var, renderer = MG.networks(dict(D=cfg["D"], seed=cfg["net_seed"]))
    torch.manual_seed(cfg["net_seed"] + 100)
    fnet = FeatureNet()
    g = torch.Generator().manual_seed(cfg["net_seed"] + 100)
    for m in fnet.modules():
        if type(m).__name__ == "InPlaceABN":
            m.weight.data = (1 + 0.2 * torch.randn(m.weight.shape, generator=g)) * torch.where(torch.rand(m.weight.shape, generator=g) < 0.2, -1.0, 1.0)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
    return fnet, sdfnet, rnet, var, renderer


def fused_pyramid(fnet, imgs):
    """GenericTrainer.obtain_pyramid_feature_maps (trainer_generic.py:1104-1125) on the reference's FeatureNet."""
    import torch.nn.functional as F
    f2, s1, s0 = fnet(imgs)
    return torch.cat([F.interpolate(f2, scale_factor=4, mode="bilinear", align_corners=True),
                      F.interpolate(s1, scale_factor=2, mode="bilinear", align_corners=True), s0], dim=1)


REN_KEYS = ("color_fine", "depth", "weights_sum", "depth_variance", "weights_max", "color_fine_mask")


def render_chunks(renderer, sdfnet, rnet, sc, T, ro, rd, chunk, dense, mask, fmaps, HW, trace_first=False):
    """The trainer's chunk loop (trainer_generic.py:503-524) on the given rays; per-ray outputs concatenated, the sample lists render() hands to
    render_core (:559) recorded, optionally the sampler's per-round inputs / outputs of the first chunk."""
    near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
    seen_z, trace = [], []
    core, up = renderer.render_core, renderer.up_sample

    def core_hook(ro_, rd_, z_, *a, **k):
        seen_z.append(z_.clone())
        return core(ro_, rd_, z_, *a, **k)

    def up_hook(ro_, rd_, z_, sdf_, n_imp, inv_s, **k):
        new_z = up(ro_, rd_, z_, sdf_, n_imp, inv_s, **k)
        if trace_first and len(seen_z) == 0 and ro_.shape[0] <= UPSAMPLE_TRACE_RAYS:
            trace.append(dict(z=z_.clone(), sdf=sdf_.reshape(z_.shape).clone(), inv_s=float(inv_s), new_z=new_z.clone()))
        return new_z

    renderer.render_core, renderer.up_sample = core_hook, up_hook
    acc = {k: [] for k in REN_KEYS + ("weights", "sdf", "gradients", "inside_sphere")}
    try:
        for s in range(0, ro.shape[0], chunk):
            r = renderer.render(T(ro[s:s + chunk]), T(rd[s:s + chunk]), near, far, sdfnet, rnet, perturb_overwrite=0, background_rgb=1.0,
                                alpha_inter_ratio=1.0, lod=0, conditional_volume=dense, conditional_valid_mask_volume=mask, feature_maps=fmaps,
                                color_maps=T(sc["images"]), w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW],
                                query_c2w=T(sc["query_c2w"])[None], if_render_with_grad=False)
            n = min(chunk, ro.shape[0] - s)
            for k in acc:
                v = r[k]
                acc[k].append(v.reshape(n, -1) if k == "sdf" else v)
    finally:
        renderer.render_core, renderer.up_sample = core, up
    out = {k: torch.cat(v, 0).numpy() for k, v in acc.items()}
    out["z_vals"] = torch.cat(seen_z, 0).numpy()
    return out, trace


T0 = time.time()


@torch.no_grad()
def main(which):
    cfg = CONFIGS[which]
    sc, ro, rd, sel, chunk = inputs(cfg)
    V, D, HW = cfg["V"], cfg["D"], 256
    T = torch.from_numpy
    fnet, sdfnet, rnet, var, renderer = build_reference_networks(cfg)
    out = {"ray_ids": sel.astype(np.int64), "chunk": np.int64(chunk)}
    out.update(checksums(sc, ro, rd))
    # ---- volume build from the IMAGES
    fmaps = fused_pyramid(fnet, T(sc["images"]))
    print(f"[{which}] fused pyramid {tuple(fmaps.shape)} {time.time() - T0:.0f} s", flush=True)
    cv = sdfnet.get_conditional_volume(feature_maps=fmaps[None], partial_vol_origin=T(sc["partial_vol_origin"])[None],
                                       proj_mats=T(sc["affine_mats"])[None], sizeH=HW, sizeW=HW, lod=0)
    dense, mask = cv["dense_volume_scale0"], cv["valid_mask_volume_scale0"]
    print(f"[{which}] volume: {int(mask.sum())} kept voxels of {D ** 3}, {time.time() - T0:.0f} s", flush=True)
    feats16 = sdfnet.compress_layer(fmaps)
    rng = np.random.default_rng(cfg["net_seed"])
    mflat = mask.reshape(-1).numpy() > 0
    out["kept_voxels"] = np.int64(mflat.sum())
    out["mask_bits"] = np.packbits(mflat)
    out["mask_sha256"] = np.frombuffer(hashlib.sha256(mflat.astype(np.uint8).tobytes()).digest(), np.uint8)
    kept = np.nonzero(mflat)[0]
    vi = np.sort(rng.choice(kept, cfg["n_dense"], replace=False))                 # a sample of the kept voxels: all 16 channels
    out["dense_idx"], out["dense_val"] = vi.astype(np.int64), dense[0].reshape(16, -1)[:, vi].t().contiguous().numpy()
    out["dense_absmax"] = np.float32(dense.abs().max())
    out["dense_sum"] = np.float64(dense.double().sum())
    pi = np.sort(rng.choice(V * HW * HW, 6000, replace=False))                   # pixels (view, y, x): all channels of the fused pyramid / compressed maps
    out["pix_idx"] = pi.astype(np.int64)
    out["fmaps_val"] = fmaps.permute(0, 2, 3, 1).reshape(-1, 56)[pi].numpy()
    out["fmaps_absmax"] = np.float32(fmaps.abs().max())
    out["feats16_val"] = feats16.permute(0, 2, 3, 1).reshape(-1, 16)[pi].numpy()
    out["feats16_absmax"] = np.float32(feats16.abs().max())
    # ---- render() in the runner's chunks
    for vi_, variance in enumerate(cfg["variance"]):
        var.variance.data = torch.tensor(float(variance))
        if vi_ == 0:
            r_ro, r_rd = ro, rd
        else:                                    # a trained model's inv_s = exp(10 variance) on a subset of the chunks
            W = 256 * cfg["ray_scale"]
            keep = np.concatenate([np.nonzero((sel // W) == r)[0] for r in cfg["trained_rows"]])
            out[f"v{vi_}_ray_pos"] = keep.astype(np.int64)
            r_ro, r_rd = ro[keep], rd[keep]
        ren, trace = render_chunks(renderer, sdfnet, rnet, sc, T, r_ro, r_rd, chunk, dense, mask, fmaps, HW, trace_first=(vi_ == 0))
        out[f"v{vi_}_variance"] = np.float64(variance)
        for k in REN_KEYS + ("z_vals",):
            out[f"v{vi_}_{k}"] = ren[k]
        out[f"v{vi_}_weights"] = ren["weights"][:2048]          # per-sample weights of the first four chunks
        n_full = min(r_ro.shape[0], 512)         # per-sample fields of the first chunk (sdf / gradients / occupancy at the final samples)
        out[f"v{vi_}_sdf"] = ren["sdf"][:n_full]
        out[f"v{vi_}_gradients"] = ren["gradients"][:n_full]
        out[f"v{vi_}_inside"] = ren["inside_sphere"][:n_full]
        for i, t in enumerate(trace):
            for k in ("z", "sdf", "new_z"):
                out[f"v{vi_}_up{i}_{k}"] = t[k].numpy()
            out[f"v{vi_}_up{i}_inv_s"] = np.float64(t["inv_s"])
        if True:                                 # every variance: the reference against itself
            amax = float(dense.abs().max())
            for si, sigma in enumerate(SELFSENS_SIGMA):
                ce, ze = [], []
                for seed in SELFSENS_SEEDS:
                    gsd = torch.Generator().manual_seed(1000 * si + seed)
                    noisy = dense + (sigma * amax) * torch.randn(dense.shape, generator=gsd) * mask
                    rn, _ = render_chunks(renderer, sdfnet, rnet, sc, T, r_ro, r_rd, chunk, noisy, mask, fmaps, HW)
                    ce.append(np.abs(rn["color_fine"] - ren["color_fine"]).max(1))
                    ze.append(np.abs(rn["z_vals"] - ren["z_vals"]).max(1))
                pre = f"selfsens{si}" if vi_ == 0 else f"v{vi_}_selfsens{si}"
                out[f"{pre}_sigma"] = np.float64(sigma)
                out[f"{pre}_color_err"] = np.stack(ce).astype(np.float32)
                out[f"{pre}_z_err"] = np.stack(ze).astype(np.float32)
                c = np.stack(ce).reshape(-1)
                print(f"[{which}] variance {variance}: reference vs itself on a volume perturbed by {sigma:g} x max: colour q50 / q90 / q99 / max {np.quantile(c, .5):.2e} "
                      f"{np.quantile(c, .9):.2e} {np.quantile(c, .99):.2e} {c.max():.2e}, > 1e-3: {(c > 1e-3).mean():.3f}, z max {np.stack(ze).max():.3f}", flush=True)
        print(f"[{which}] variance {variance}: {r_ro.shape[0]} rays, weights_sum max {float(ren['weights_sum'].max()):.4f}, "
              f"rays with weight > 0.5: {int((ren['weights_sum'] > 0.5).sum())}, colour-valid rays {int(ren['color_fine_mask'].sum())}, "
              f"{time.time() - T0:.0f} s", flush=True)
    var.variance.data = torch.tensor(float(cfg["variance"][0]))
    # ---- extract_fields
    box = grid_box(cfg)
    if box is not None:
        bmin, bmax, R = box
        u = renderer.extract_fields(bmin, bmax, R, lambda p, **kw: sdfnet.sdf(p, **kw), "cpu", conditional_volume=dense, lod=0)
        out["u"] = u
        out["u_bounds"] = np.stack([bmin.numpy(), bmax.numpy()])
        ins = u > 0
        print(f"[{which}] extract_fields {R}^3: {int(ins.sum())} inside nodes, sign changes along x {int((ins[1:] != ins[:-1]).sum())}, "
              f"{time.time() - T0:.0f} s", flush=True)
        # ---- validate_colored_mesh (trainer_generic.py:1309-1363) on vertices of that field's surface: view-independent projector (normals = the SDF's
        # autograd gradient) + the rendering network.  (mcubes is not installed: the vertices come from the checker's marching cubes on the reference's u.)
        from oracle import mc as omc
        v_idx, _ = omc.marching_cubes(u, 0.0)
        if len(v_idx):
            pick = np.sort(rng.choice(len(v_idx), min(4000, len(v_idx)), replace=False))
            vw = v_idx[pick] / (R - 1.0) * (bmax.numpy() - bmin.numpy())[None, :] + bmin.numpy()[None, :]          # sparse_neus_renderer.py:936
            vp = torch.tensor(vw).to(dense)                                                                        # trainer_generic.py:1327 (float64 -> float32)
            with torch.enable_grad():
                geo, rf, rdiff, vm, _, _ = renderer.rendering_projector.compute_view_independent(
                    vp.clone(), lod=0, geometryVolume=dense[0], geometryVolumeMask=mask[0], sdf_network=sdfnet, rendering_feature_maps=fmaps,
                    color_maps=T(sc["images"]), w2cs=T(sc["w2cs"]), target_candidate_w2cs=None, intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW],
                    query_img_idx=0, query_c2w=T(sc["query_c2w"])[None])
            vcol, _ = rnet(geo.detach(), rf.detach(), rdiff.detach(), vm)
            out["vert_pts"], out["vert_rgb"], out["vert_mask"] = vp.numpy(), vcol[0].detach().numpy(), vm[:, 0].numpy()
            print(f"[{which}] vertex colours: {len(pick)} of {len(v_idx)} vertices, {time.time() - T0:.0f} s", flush=True)
    # ---- state dicts (parameters only: the reference runs every normalisation layer on batch statistics)
    for prefix, net in (("fnet.", fnet), ("sdf.", sdfnet), ("ren.", rnet), ("var.", var)):
        for k, v in net.state_dict().items():
            if "num_batches_tracked" in k or "running_" in k:
                continue
            out["w:" + prefix + k] = v.numpy()
    out["w:var.variance"] = np.float32(cfg["variance"][0])
    for k, v in cfg.items():
        if isinstance(v, (int, float)):
            out["cfg_" + k] = np.float64(v)
    path = os.path.join(HERE, cfg["name"])
    np.savez_compressed(path, **out)
    print(f"wrote {path} {os.path.getsize(path) // 1024} KiB in {time.time() - T0:.0f} s")


# ---- BASELINE config 5's sparse 256^3 level: the reference's coarse-to-fine path (trainer_generic.py:437-491) on config 2's scene ----------------------
C5 = dict(name="ref_c5_lod1_sample.npz", base="c2", net_seed=31, rows=(232,), n_dense=50000, n_pts=20000)


def lod1_networks(D1, seed):
    """The reference's lod-1 SparseSdfNetwork (confs/one2345_lod_train.conf:83-96: 8 compressed channels, parent feature concatenated) + its own rendering /
    variance networks, seeded, zero-initialised paths perturbed like make_golden.main()."""
    from oracle import ref_import as RI
    R = RI.load()
    torch.manual_seed(seed)
    sdf1 = R.SparseSdfNetwork(lod=1, ch_in=56, voxel_size=2.0 / (D1 - 1), vol_dims=[D1] * 3, hidden_dim=128, cost_type="variance_mean",
                              d_pyramid_feature_compress=8, regnet_d_out=16, num_sdf_layers=4, multires=6)
    g1 = torch.Generator().manual_seed(seed)
    sdf1.sdf_layer.lin0.weight_v.data[:, 3:] += 0.003 * torch.randn(128, 36, generator=g1)
    sdf1.sdf_layer.lin1.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g1)
    sdf1.sdf_layer.lin2.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g1)
    for m in sdf1.sparse_costreg_net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = 1 + 0.2 * torch.randn(m.weight.shape, generator=g1)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g1)
    sdf1.compress_layer.bn.weight.data = (1 + 0.2 * torch.randn(8, generator=g1)) * torch.tensor([1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 1.0, 1.0])
    sdf1.compress_layer.bn.bias.data = 0.1 * torch.randn(8, generator=g1)
    rnet1 = R.GeneralRenderingNetwork(in_geometry_feat_ch=16, in_rendering_feat_ch=56, anti_alias_pooling=True)
    var1 = R.SingleVarianceNetwork(0.2)
    conf = RI.Conf({"general.base_exp_dir": "/tmp", "model.h_patch_size": 3})
    ren1 = R.SparseNeuSRenderer(None, sdf1, var1, rnet1, 64, 64, 0, 1.0, alpha_type="div", conf=conf)
    return sdf1, rnet1, var1, ren1


@torch.no_grad()
def main_c5():
    c5, cfg = C5, CONFIGS[C5["base"]]
    sc, ro, rd, sel, chunk = inputs(cfg)
    V, D, HW = cfg["V"], cfg["D"], 256
    D1 = 2 * D
    T = torch.from_numpy
    fnet, sdfnet, rnet, var, renderer = build_reference_networks(cfg)
    out = {}
    out.update(checksums(sc, ro, rd))
    fmaps = fused_pyramid(fnet, T(sc["images"]))
    origin = T(sc["partial_vol_origin"])[None]
    cv = sdfnet.get_conditional_volume(feature_maps=fmaps[None], partial_vol_origin=origin, proj_mats=T(sc["affine_mats"])[None], sizeH=HW, sizeW=HW, lod=0)
    dense, mask, coords = cv["dense_volume_scale0"], cv["valid_mask_volume_scale0"], cv["coords_scale0"]
    print(f"[c5] lod-0 volume {int(mask.sum())} voxels, {time.time() - T0:.0f} s", flush=True)
    # ---- trainer_generic.py:437-440
    sv = sdfnet.get_sdf_volume(dense, mask, coords, origin)
    rng = np.random.default_rng(c5["net_seed"])
    kept = np.nonzero(mask.reshape(-1).numpy() > 0)[0]
    vi = np.sort(rng.choice(kept, 100000, replace=False))
    out["l0_sdf_idx"], out["l0_sdf_val"] = vi.astype(np.int64), sv.reshape(-1)[vi].numpy()
    out["l0_sdf_bits_below_thr"] = np.packbits((sv.reshape(-1).abs() < 0.02).numpy())          # |sdf| < the pruning threshold, every voxel
    # ---- :470-472 (no depth filter): the default threshold 0.02, stepping down while more than 110,000 voxels remain
    np.random.seed(0)
    pc, pf = renderer.get_valid_sparse_coords_by_sdf(sv[0], coords[0], mask[0], dense[0])
    assert pc.shape[0] <= 110000, "the unseeded np.random.choice branch fired: this scene cannot pin the selection"
    out["pre_coords"] = pc[:, 1:].numpy().astype(np.int16)
    # the threshold the loop ended on: the largest one of the sequence 0.02, 0.018, ... that keeps <= 110,000 voxels
    print(f"[c5] pruned lod-0 voxels {pc.shape[0]}, {time.time() - T0:.0f} s", flush=True)
    pc2 = pc.clone()
    pc2[:, 1:] = pc2[:, 1:] * 2                                                                 # :474
    sdf1, rnet1, var1, ren1 = lod1_networks(D1, c5["net_seed"])
    cv1 = sdf1.get_conditional_volume(feature_maps=fmaps[None], partial_vol_origin=origin, proj_mats=T(sc["affine_mats"])[None], sizeH=HW, sizeW=HW,
                                      pre_coords=pc2, pre_feats=pf)
    dense1, mask1 = cv1["dense_volume_scale1"], cv1["valid_mask_volume_scale1"]
    m1 = mask1.reshape(-1).numpy() > 0
    out["l1_kept_voxels"] = np.int64(m1.sum())
    out["l1_mask_bits"] = np.packbits(m1)
    k1 = np.nonzero(m1)[0]
    v1 = np.sort(rng.choice(k1, min(c5["n_dense"], k1.size), replace=False))
    out["l1_dense_idx"], out["l1_dense_val"] = v1.astype(np.int64), dense1[0].reshape(16, -1)[:, v1].t().contiguous().numpy()
    out["l1_dense_absmax"] = np.float32(dense1.abs().max())
    print(f"[c5] lod-1 volume {D1}^3: {int(m1.sum())} children kept of {8 * pc.shape[0]}, {time.time() - T0:.0f} s", flush=True)
    # ---- SDF of the lod-1 network at points around the kept voxels
    ctr = torch.stack(torch.meshgrid(*[torch.arange(D1, dtype=torch.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)[T(k1[rng.choice(k1.size, c5["n_pts"])])]
    pts = (ctr * (2.0 / (D1 - 1)) + origin.reshape(1, 3) + T(rng.uniform(-0.004, 0.004, (c5["n_pts"], 3)).astype(np.float32))).contiguous()
    out["l1_pts"] = pts.numpy()
    out["l1_sdf"] = sdf1.sdf(pts.clone(), dense1, 1)["sdf_pts_scale1"].numpy()
    # ---- one chunk of the lod-1 val loop (trainer_generic.py:540-566)
    W = 256 * cfg["ray_scale"]
    keep = np.concatenate([np.nonzero((sel // W) == r)[0] for r in c5["rows"]])
    out["ray_pos"] = keep.astype(np.int64)
    ren, trace = render_chunks(ren1, sdf1, rnet1, sc, T, ro[keep], rd[keep], chunk, dense1, mask1, fmaps, HW)
    for k in REN_KEYS + ("z_vals", "weights"):
        out["v0_" + k] = ren[k]
    out["v0_variance"] = np.float64(0.2)
    amax = float(dense1.abs().max())
    ce, ze = [], []
    for seed in SELFSENS_SEEDS:
        gsd = torch.Generator().manual_seed(seed)
        noisy = dense1 + (SELFSENS_SIGMA[0] * amax) * torch.randn(dense1.shape, generator=gsd) * mask1
        rn, _ = render_chunks(ren1, sdf1, rnet1, sc, T, ro[keep], rd[keep], chunk, noisy, mask1, fmaps, HW)
        ce.append(np.abs(rn["color_fine"] - ren["color_fine"]).max(1)); ze.append(np.abs(rn["z_vals"] - ren["z_vals"]).max(1))
        noisy = None
    out["selfsens0_sigma"], out["selfsens0_color_err"], out["selfsens0_z_err"] = np.float64(SELFSENS_SIGMA[0]), np.stack(ce).astype(np.float32), np.stack(ze).astype(np.float32)
    print(f"[c5] lod-1 render: {len(keep)} rays, weights_sum max {float(ren['weights_sum'].max()):.4f}, rays with weight > 0.5: {int((ren['weights_sum'] > 0.5).sum())}; "
          f"reference vs itself colour q50 / q99 / max {np.quantile(np.stack(ce), .5):.2e} {np.quantile(np.stack(ce), .99):.2e} {np.stack(ce).max():.2e}, {time.time() - T0:.0f} s", flush=True)
    for prefix, net in (("fnet.", fnet), ("sdf.", sdfnet), ("sdf1.", sdf1), ("ren1.", rnet1)):
        for k, v in net.state_dict().items():
            if "num_batches_tracked" in k or "running_" in k:
                continue
            out["w:" + prefix + k] = v.numpy()
    path = os.path.join(HERE, c5["name"])
    np.savez_compressed(path, **out)
    print(f"wrote {path} {os.path.getsize(path) // 1024} KiB in {time.time() - T0:.0f} s")


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    for w in (sys.argv[1:] or ["c1"]):
        main_c5() if w == "c5" else main(w)
