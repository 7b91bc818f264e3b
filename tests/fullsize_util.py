"""Oracle-vs-HIP comparison at BASELINE config-2 size (test infrastructure; used by tests/test_gpu_edges_and_fullsize.py, bench.py's
cpu_baseline leg and tools/fullsize_parity.py).

The comparison is decomposed so that every ray is accounted for:

  (A) sampler: per-ray sample lists z [R,128] of the HIP path vs the oracle.  The reference's inverse-CDF sampler (render_utils.py:8-51)
      is ill-conditioned for samples that land in (nearly) empty bins: pdf mass p ~ 1e-5 there, so fp32 rounding in the cdf is
      amplified by 1/p.  The oracle reports, per ray, the smallest pdf mass any of its new samples landed in (``diag``); rays are
      classified WELL-conditioned (min mass >= P_WELL) or ILL-conditioned.  Well-conditioned rays must agree tightly; ill-conditioned
      rays may move samples by a fraction of the coarse spacing.
  (B) everything downstream of the sampler: the oracle's render_core evaluated ON THE HIP PATH'S OWN sample depths must reproduce
      the HIP colour / depth / weights for ALL rays at a tight tolerance (no quantile)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import recon as O  # noqa: E402

pkg = importlib.import_module("one-2-3-45_amd")
pipeline = importlib.import_module("one-2-3-45_amd.pipeline")

P_WELL = 1e-3          # a new sample in a bin of pdf mass >= 1e-3 moves by <= 1e-7 / 1e-3 = 1e-4 of a bin under fp32 cdf rounding
CHUNK = 16             # rays per oracle / HIP call: identical per-call quirk semantics (cat_z_vals' "<= 1 valid point" rule) on both sides


def build_full_scene(dev, seed=0, image_seed=3, V=8, D=128):
    wt = pipeline.SceneWeights(dev, seed=seed)
    sc = pkg.synth.make_scene(V, image_seed=image_seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    vol = pipeline.build_volume(wt, T(sc["images"]), T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1))
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    return dict(wt=wt, sc=sc, vol=vol, proj=proj, cam_pos=cam_pos, T=T, D=D, ro=ro, rd=rd)


def select_rays(n_total, n):
    """n rays spread over the image, rounded to whole CHUNKs."""
    n = max(CHUNK, (n // CHUNK) * CHUNK)
    return np.unique(np.linspace(0, n_total - 1, n).astype(np.int64))[: n]


def _oracle_args(full):
    wt, vol, sc, D = full["wt"], full["vol"], full["sc"], full["D"]
    dense = vol["vol_cl"].permute(3, 0, 1, 2).contiguous().cpu()
    mask = vol["maskvol"].view(D, D, D).cpu()
    W = {k: torch.from_numpy(np.asarray(v)) for k, v in wt.sdfW.items()}
    RW = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in wt.color_sd.items()}
    fm = (vol["fmaps"] if vol["fmaps"] is not None else vol["cmaps"][..., 3:59].permute(0, 3, 1, 2)).contiguous().cpu()   # the pipeline keeps the channel-last map only
    H, Wd = sc["images"].shape[2:]
    return dict(volume=dense, maskvol=mask, W=W, RW=RW, variance=torch.tensor(wt.variance, dtype=torch.float32), feat_maps=fm,
                color_maps=torch.from_numpy(sc["images"]), w2cs=torch.from_numpy(sc["w2cs"]), K=torch.from_numpy(sc["intrinsics"]),
                img_wh=(Wd, H), query_c2w=torch.from_numpy(sc["query_c2w"]))


KEYS = ("color_fine", "depth", "weights_sum", "weights", "z_vals", "mid_z_vals", "color_fine_mask")


@torch.no_grad()
def oracle_render_sample(full, n, budget_s=None, sel=None):
    """-> (dict of per-ray oracle outputs incl. 'min_pdf' [R], selected ray ids, seconds).  Stops early after budget_s seconds."""
    import time
    sel = select_rays(full["ro"].shape[0], n) if sel is None else sel
    a = _oracle_args(full)
    near, far = torch.tensor(float(full["sc"]["query_near_far"][0])), torch.tensor(float(full["sc"]["query_near_far"][1]))
    ro, rd = torch.from_numpy(full["ro"][sel]), torch.from_numpy(full["rd"][sel])
    acc = {k: [] for k in KEYS + ("min_pdf",)}
    t0 = time.time()
    done = 0
    while done < len(sel) and (budget_s is None or time.time() - t0 < budget_s):
        diag = []
        r = O.render(ro[done:done + CHUNK], rd[done:done + CHUNK], near, far, a["volume"], a["maskvol"], a["W"], a["RW"], a["variance"],
                     a["feat_maps"], a["color_maps"], a["w2cs"], a["K"], a["img_wh"], a["query_c2w"], diag=diag)
        for k in KEYS:
            acc[k].append(r[k])
        acc["min_pdf"].append(torch.stack([d.min(1).values for d in diag], 0).min(0).values)
        done += CHUNK
    dt = time.time() - t0
    out = {k: torch.cat(v, 0) for k, v in acc.items()}
    return out, sel[:done], dt


@torch.no_grad()
def gpu_render_sample(full, sel, precision):
    wt, T, sc = full["wt"], full["T"], full["sc"]
    old = (wt.sdf_precision, wt.color_precision)
    wt.sdf_precision = wt.color_precision = precision
    acc = {k: [] for k in ("color", "depth", "weights_sum", "weights", "z_vals", "mid_z", "color_mask")}
    try:
        ro, rd = T(full["ro"][sel]), T(full["rd"][sel])
        qc = T(sc["query_c2w"][:3, 3].copy())
        for s in range(0, len(sel), CHUNK):
            o = pipeline.render(wt, full["vol"], full["proj"], full["cam_pos"], ro[s:s + CHUNK].contiguous(), rd[s:s + CHUNK].contiguous(),
                                float(sc["query_near_far"][0]), float(sc["query_near_far"][1]), qc, want_z=True)
            for k in acc:
                v = o[k]
                acc[k].append((v.t() if v.dim() == 2 and k in ("weights", "z_vals", "mid_z") else v).cpu())
    finally:
        wt.sdf_precision, wt.color_precision = old
    return {k: torch.cat(v, 0) for k, v in acc.items()}


@torch.no_grad()
def oracle_core(a, ro, rd, near, far, z, n_samples=64, chunk=CHUNK):
    """The oracle's render_core (everything downstream of the sampler) on GIVEN sample depths z [R,S]; a = oracle argument dict."""
    sd = float((torch.tensor(far) - torch.tensor(near)) / n_samples)
    acc = {k: [] for k in ("color_fine", "depth", "weights_sum", "weights", "gradients", "sdf", "color_fine_mask", "depth_variance")}
    for s in range(0, ro.shape[0], chunk):
        r = O.render_core(ro[s:s + chunk], rd[s:s + chunk], z[s:s + chunk], sd, a["volume"], a["maskvol"], a["W"], a["RW"], a["variance"],
                          a["feat_maps"], a["color_maps"], a["w2cs"], a["K"], a["img_wh"], a["query_c2w"])
        for k in acc:
            acc[k].append(r[k].reshape(min(chunk, ro.shape[0] - s), -1) if k == "sdf" else r[k])
    return {k: torch.cat(v, 0) for k, v in acc.items()}


def oracle_core_on(full, sel, z):
    """oracle_core for the config-2 scene: the HIP path's own sample lists z [R,S] of the selected rays."""
    return oracle_core(_oracle_args(full), torch.from_numpy(full["ro"][sel]), torch.from_numpy(full["rd"][sel]),
                       float(full["sc"]["query_near_far"][0]), float(full["sc"]["query_near_far"][1]), z)


def classify(out, ref, core=None, verbose=False):
    """Per-ray error statistics of one HIP render `out` against the oracle `ref` (+ `core` = oracle downstream on the HIP z lists)."""
    near_far_spacing = float(ref["z_vals"][:, -1].max() - ref["z_vals"][:, 0].min()) / 63
    zerr = (out["z_vals"] - ref["z_vals"]).abs().max(1).values
    well = ref["min_pdf"] >= P_WELL
    cerr = (out["color"] - ref["color_fine"]).abs().max(1).values
    derr = (out["depth"] - ref["depth"][:, 0]).abs()
    res = {
        "rays": int(len(zerr)), "well_conditioned": int(well.sum()), "ill_conditioned": int((~well).sum()),
        "rays_hitting_surface": int((ref["weights_sum"][:, 0] > 0.5).sum()),
        "z_err_max_well": float(zerr[well].max()) if well.any() else 0.0,
        "z_err_max_ill": float(zerr[~well].max()) if (~well).any() else 0.0, "coarse_spacing": near_far_spacing,
        "color_err_max_well": float(cerr[well].max()) if well.any() else 0.0, "color_err_max_all": float(cerr.max()),
        "depth_err_max_well": float(derr[well].max()) if well.any() else 0.0, "depth_err_max_all": float(derr.max()),
        "frac_rays_color_gt_1e-4": float((cerr > 1e-4).float().mean()), "frac_rays_color_gt_1e-3": float((cerr > 1e-3).float().mean()),
        "ill_rays_with_z_err_gt_1e-4": int(((zerr > 1e-4) & ~well).sum()), "well_rays_with_z_err_gt_1e-4": int(((zerr > 1e-4) & well).sum()),
        "color_mask_mismatch": int((out["color_mask"].bool() != ref["color_fine_mask"][:, 0]).sum()),
    }
    if core is not None:
        res["downstream_color_err_max_all"] = float((out["color"] - core["color_fine"]).abs().max())
        res["downstream_depth_err_max_all"] = float((out["depth"] - core["depth"][:, 0]).abs().max())
        res["downstream_weights_err_max_all"] = float((out["weights"] - core["weights"]).abs().max())
    if verbose:
        print({k: (round(v, 8) if isinstance(v, float) else v) for k, v in res.items()}, file=sys.stderr)
    return res


@torch.no_grad()
def sampler_stage_check(ops, dev, ro, rd, near, far, oracle_args, maskvol_dev, D, chunk=CHUNK):
    """Drives the HIP sampler stage (o2345_ray_upsample) with the ORACLE's own per-round inputs (z, sdf) and compares the 16 new depths of
    every ray in every one of the 4 rounds.  Returns per-sample tensors: dz [4,R,16] (|new_z_hip - new_z_oracle|), pdf [4,R,16] (pdf mass of
    the bin the oracle's sample lands in), width [4,R,16] (width of that bin)."""
    a = oracle_args
    R = ro.shape[0]
    dz, pdfs, widths = [], [], []
    for s in range(0, R, chunk):
        diag, trace = [], []
        O.render(ro[s:s + chunk], rd[s:s + chunk], torch.tensor(near), torch.tensor(far), a["volume"], a["maskvol"], a["W"], a["RW"], a["variance"],
                 a["feat_maps"], a["color_maps"], a["w2cs"], a["K"], a["img_wh"], a["query_c2w"], diag=diag, trace=trace)
        dzs, ws = [], []
        for t in trace:
            nz, _, _ = ops.ray_upsample(ro[s:s + chunk].to(dev), rd[s:s + chunk].to(dev), t["z"].t().contiguous().to(dev),
                                        t["sdf"].t().contiguous().to(dev), t["inv_s"], maskvol_dev, D, t["new_z"].shape[1])
            dzs.append((nz.t().cpu() - t["new_z"]).abs())
            # width of the bin each oracle sample lies in
            idx = (torch.searchsorted(t["z"].contiguous(), t["new_z"].contiguous(), right=True) - 1).clamp(0, t["z"].shape[1] - 2)
            ws.append(t["z"].gather(1, idx + 1) - t["z"].gather(1, idx))
        dz.append(torch.stack(dzs)); pdfs.append(torch.stack(diag)); widths.append(torch.stack(ws))
    return torch.cat(dz, 1), torch.cat(pdfs, 1), torch.cat(widths, 1)


@torch.no_grad()
def oracle_self_sensitivity(full, sel, ref, sigma=2e-6, seeds=(1, 2, 3)):
    """How much the ORACLE's own rendered colour / depth move when its SDF values carry fp32-class relative noise (sigma = 2e-6, the
    measured accuracy class of every fp32 SDF evaluation incl. ATen's own, cf. test_sdf_mlp): the error-propagation envelope of the
    reference algorithm (4 up-sampling rounds with sigmoid slopes up to 512 + inverse-CDF sampling).  -> (colour err [K,R], depth err [K,R])."""
    ce, de = [], []
    for sd in seeds:
        O.SDF_NOISE = (sigma, torch.Generator().manual_seed(sd))
        try:
            r, _, _ = oracle_render_sample(full, len(sel), sel=sel)
        finally:
            O.SDF_NOISE = None
        ce.append((r["color_fine"] - ref["color_fine"]).abs().max(1).values)
        de.append((r["depth"] - ref["depth"]).abs()[:, 0])
    return torch.stack(ce), torch.stack(de)


# ------------------------------------------------------------------------------------------------ volume build / mesh field vs the oracle
def oracle_weight_dicts(wt):
    """Reference-format state dicts of a pipeline.SceneWeights for the oracle's FeatureNet / compress layer / sparse CNN."""
    from scene_util import costreg_oracle_weights
    fsd = {k: v.detach().cpu() for k, v in wt.featurenet.state_dict().items()}
    csd = {k: v.detach().cpu() for k, v in wt.compress.state_dict().items()}
    return fsd, csd, costreg_oracle_weights(wt.costreg_sd)


@torch.no_grad()
def oracle_volume(wt, sc, D):
    """The ORACLE's own get_conditional_volume from the IMAGES (FeatureNet -> fused pyramid -> compress layer -> back-projection + aggregation ->
    sparse CNN -> dense scatter; oracle/recon.py:conditional_volume).  ~30 GFLOP at BASELINE config 2: tens of seconds on the host."""
    fsd, csd, cw = oracle_weight_dicts(wt)
    return O.conditional_volume(torch.from_numpy(sc["images"]), fsd, csd, cw, torch.from_numpy(sc["affine_mats"]), [D, D, D], 2.0 / (D - 1),
                                torch.from_numpy(sc["partial_vol_origin"]))


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


def volume_vs_oracle(vol, ov, D):
    """HIP volume build (pipeline.build_volume dict) vs oracle_volume(): integer results as booleans, fp32 tensors as max abs error / max|oracle|."""
    return {"kept_voxels": int(vol["coords"].shape[0]), "kept_set_exact": bool(torch.equal(vol["coords"].cpu(), ov["coords"])),
            "view_counts_exact": bool(torch.equal(vol["cnt"].cpu().long().view(-1), ov["cnt"].long().view(-1))),
            "mask_exact": bool(torch.equal(vol["maskvol"].view(D, D, D).cpu(), ov["mask"][0, 0])),
            "fused_pyramid": _relerr(vol["cmaps"][..., 3:59].permute(0, 3, 1, 2), ov["fmaps"]),
            "compressed_maps": _relerr(vol["feats_nhwc"].permute(0, 3, 1, 2), ov["feats16"]),
            "cost_volume_rows": _relerr(vol["rows"], ov["rows"]), "sparse_cnn_rows": _relerr(vol["rows16"], ov["rows16"]),
            "dense_volume": _relerr(vol["vol_cl"].permute(3, 0, 1, 2)[None], ov["dense"])}


@torch.no_grad()
def mesh_field_vs_oracle(ops, wt, vol, u, R=256, B=64):
    """End-to-end mesh agreement on the B^3 sub-block of the R^3 extraction lattice with the most sign changes: HIP's u (the lattice kernel) vs the
    oracle's extract_fields (sparse_neus_renderer.py:881-905, u = -sdf on linspace(-1,1,R)^3); sign disagreements, IoU of the inside sets, and the
    two meshes of the block (HIP marching cubes on HIP's u, oracle marching cubes on the oracle's u)."""
    from oracle import mc as omc
    ins = (u > 0)
    chg = (ins[1:] != ins[:-1]).float()
    best, origin = -1.0, None
    for x0 in range(0, R - B + 1, 32):
        for y0 in range(0, R - B + 1, 32):
            for z0 in range(0, R - B + 1, 32):
                c = float(chg[x0:x0 + B - 1, y0:y0 + B, z0:z0 + B].sum())
                if c > best:
                    best, origin = c, (x0, y0, z0)
    x0, y0, z0 = origin
    lin = torch.linspace(-1, 1, R)
    gx, gy, gz = torch.meshgrid(lin[x0:x0 + B], lin[y0:y0 + B], lin[z0:z0 + B], indexing="ij")
    pts = torch.stack([gx, gy, gz], -1).reshape(-1, 3)
    dense = vol["vol_cl"].permute(3, 0, 1, 2).contiguous().cpu()
    W = {k: torch.from_numpy(np.asarray(v)) for k, v in wt.sdfW.items()}
    uo = torch.cat([-O.sdf(pts[s:s + (1 << 16)], dense, W)[0][:, 0] for s in range(0, pts.shape[0], 1 << 16)]).view(B, B, B)
    uh = u[x0:x0 + B, y0:y0 + B, z0:z0 + B].cpu()
    flips = (uh > 0) != (uo > 0)
    nflip = int(flips.sum())
    inter, union = int(((uh > 0) & (uo > 0)).sum()), int(((uh > 0) | (uo > 0)).sum())
    vh, th = ops.marching_cubes(uh.to(u.device).contiguous(), 0.0)
    vo, to = omc.marching_cubes(uo.numpy(), 0.0)
    res = {"block_origin": list(origin), "block": B, "grid": R, "sign_changes_in_block": int(best), "field_err_max": float((uh - uo).abs().max()),
           "field_scale": float(uo.abs().max()), "mesh_sign_flips": nflip, "flipped_nodes": torch.nonzero(flips)[:20].tolist(),
           "abs_u_at_flips_max": float(uo[flips].abs().max()) if nflip else 0.0, "inside_nodes_union": union, "iou": inter / max(1, union),
           "hip_mesh": [int(vh.shape[0]), int(th.shape[0])], "oracle_mesh": [int(vo.shape[0]), int(to.shape[0])]}
    if nflip == 0 and th.shape[0] == to.shape[0] and vh.shape[0] == vo.shape[0]:
        res["triangles_identical"] = bool(np.array_equal(th.cpu().numpy(), to))
        d = np.abs(vh.cpu().numpy() - vo).max(1)
        res["vertex_shift_max_mean_cells"] = [float(d.max()), float(d.mean())]
    return res
