"""The three-clause render contract, shared by every GPU render parity test (test infrastructure).

The reference's render() = hierarchical sampler (4 up-sampling rounds, sigmoid slopes 64 ... 512, inverse-CDF sampling) followed by
render_core (SDF + gradient + colour network at the final sample list, NeuS compositing with sigmoid slope inv_s = exp(10 variance)).
Both parts AMPLIFY fp32-class differences of the SDF values -- the sampler by relocating samples of near-empty bins, the compositing
by inv_s (7.4 for the initial variance 0.2, several hundred for a trained model) -- so "HIP == oracle to 1e-5 end to end" is not a
property even two runs of the reference on different hardware have.  What IS asserted, for ALL rays and ALL samples, no quantiles:

  (1) SAMPLER on identical inputs: o2345_ray_upsample driven with the oracle's own per-round (z, sdf): every new depth within 2e-4
      absolute and within max(5e-3 of the width of its bin, 5e-7 = two ulps of a depth in [1, 2)) -- late rounds produce bins only ~1e-5 wide,
      where a one-ulp difference of the depth itself is already 0.7 % of the bin.
  (2) DOWNSTREAM of the sampler on identical sample lists: the oracle's render_core evaluated on the HIP path's OWN sample lists
      reproduces HIP's weights / colour / depth / weight sum / depth variance / colour mask within
          base tolerance  +  4 x  the oracle's OWN sensitivity on these sample lists to fp32-class SDF noise
      (absolute sigma 1e-6 = the accuracy class every fp32 evaluation of the SDF network has, ATen's included; relative 2e-6).
      At inv_s = 7.4 the sensitivity term is ~1e-6 and the base tolerance decides (colour 3e-5, the others 2e-5); at inv_s = 665 it is
      what a 1e-6 SDF difference does to sigmoid(665 sdf).
  (3) END TO END: sample lists differ by at most one coarse section; rays whose lists coincide (1e-6) agree as tightly as in (2); every
      ray that deviates by more than 1e-4 has differing sample lists (ids printed); and the error distribution (q50 / q90 / q99 / mean)
      stays within 4 x the oracle's own end-to-end sensitivity to the same SDF noise.
"""
import sys

import torch

from oracle import recon as O

ABS_SIGMA, REL_SIGMA = 1e-6, 2e-6
SAMPLER_ABS_FLOOR = 5e-7
BASE = dict(color=3e-5, depth=2e-5, weights=2e-5, weights_sum=2e-5, depth_var=2e-5)


def _core(a, ro, rd, z, near, far, variance, air, bg, chunk, n_samples=64):
    sd = float((torch.tensor(far) - torch.tensor(near)) / n_samples)
    keys = ("color_fine", "depth", "weights_sum", "weights", "color_fine_mask", "depth_variance")
    acc = {k: [] for k in keys}
    for s in range(0, ro.shape[0], chunk):
        r = O.render_core(ro[s:s + chunk], rd[s:s + chunk], z[s:s + chunk], sd, a["volume"], a["maskvol"], a["W"], a["RW"], variance,
                          a["feat_maps"], a["color_maps"], a["w2cs"], a["K"], a["img_wh"], a["query_c2w"], alpha_inter_ratio=air,
                          background_rgb=bg)
        for k in keys:
            acc[k].append(r[k])
    return {k: torch.cat(v, 0) for k, v in acc.items()}


def _render(a, ro, rd, near, far, variance, air, bg, chunk, trace=False, n_samples=64, n_importance=64):
    keys = ("color_fine", "depth", "weights_sum", "weights", "color_fine_mask", "depth_variance", "z_vals")
    acc = {k: [] for k in keys}
    traces = []
    for s in range(0, ro.shape[0], chunk):
        tr = [] if trace else None
        r = O.render(ro[s:s + chunk], rd[s:s + chunk], torch.tensor(near), torch.tensor(far), a["volume"], a["maskvol"], a["W"], a["RW"], variance,
                     a["feat_maps"], a["color_maps"], a["w2cs"], a["K"], a["img_wh"], a["query_c2w"], n_samples=n_samples, n_importance=n_importance,
                     alpha_inter_ratio=air, background_rgb=bg, trace=tr)
        for k in keys:
            acc[k].append(r[k])
        traces.append(tr)
    return {k: torch.cat(v, 0) for k, v in acc.items()}, traces


def _noisy(fn, seeds=(1, 2, 3)):
    outs = []
    for sd in seeds:
        O.SDF_NOISE = (REL_SIGMA, torch.Generator().manual_seed(sd), ABS_SIGMA)
        try:
            outs.append(fn())
        finally:
            O.SDF_NOISE = None
    return outs


def _hip(out):
    """ops.render_rays' sample-major dict -> ray-major CPU tensors under the oracle's key names."""
    return dict(color_fine=out["color"].cpu(), depth=out["depth"].cpu()[:, None], weights=out["weights"].t().cpu().contiguous(),
                weights_sum=out["weights_sum"].cpu()[:, None], depth_variance=out["depth_var"].cpu()[:, None],
                color_fine_mask=out["color_mask"].cpu().bool()[:, None], z_vals=out["z_vals"].t().cpu().contiguous())


PAIRS = (("color", "color_fine"), ("depth", "depth"), ("weights", "weights"), ("weights_sum", "weights_sum"), ("depth_var", "depth_variance"))


@torch.no_grad()
def three_clause(ops, dev, scene, a, ro, rd, near, far, variance=0.2, air=1.0, bg=1.0, precision="f16x3", chunk=None, sampler=True,
                 label="", n_samples=64, n_importance=64, quantiles=True, e2e_caps=None, strict_e2e=True):
    """scene: the device dict of ops.render_rays; a: oracle arguments (volume [C,D,D,D], maskvol [D,D,D], W, RW, feat_maps, color_maps, w2cs,
    K, img_wh, query_c2w); ro / rd: CPU float32 [R,3]; bg: None (the reference's background_rgb=None: nothing added) or a float.
    chunk: rays per call on BOTH sides (the reference's per-call quirks -- cat_z_vals' "<= 1 valid point" rule -- then apply identically).
    Returns the measured numbers (dict) after asserting the three clauses."""
    R = ro.shape[0]
    chunk = chunk or R
    var_t = torch.tensor(float(variance))
    inv_s = float(torch.exp(var_t * 10.0).clip(1e-6, 1e6))
    bgv = 0.0 if bg is None else float(bg)
    D = a["volume"].shape[-1]
    scene = dict(scene, sdf_precision=precision, color_precision=precision)
    outs = []
    for s in range(0, R, chunk):
        o = ops.render_rays(scene, ro[s:s + chunk].to(dev).contiguous(), rd[s:s + chunk].to(dev).contiguous(), near, far, n_samples,
                            n_importance, inv_s, float(air), bgv, a["query_c2w"][:3, 3].contiguous().to(dev), want_z=True)
        outs.append(_hip(o))
    hip = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    ref, traces = _render(a, ro, rd, near, far, var_t, air, bgv, chunk, trace=sampler, n_samples=n_samples, n_importance=n_importance)
    res = {"rays": R, "inv_s": inv_s, "alpha_inter_ratio": float(air), "background": bg, "precision": precision,
           "rays_hitting_surface": int((ref["weights_sum"][:, 0] > 0.5).sum())}
    # ---- (1) sampler stage on identical inputs
    if sampler:
        dzs, ws = [], []
        for ci, s in enumerate(range(0, R, chunk)):
            for t in traces[ci]:
                nz, _, _ = ops.ray_upsample(ro[s:s + chunk].to(dev), rd[s:s + chunk].to(dev), t["z"].t().contiguous().to(dev),
                                            t["sdf"].t().contiguous().to(dev), t["inv_s"], scene["maskvol"].reshape(-1), D, t["new_z"].shape[1])
                dzs.append((nz.t().cpu() - t["new_z"]).abs())
                idx = (torch.searchsorted(t["z"].contiguous(), t["new_z"].contiguous(), right=True) - 1).clamp(0, t["z"].shape[1] - 2)
                ws.append(t["z"].gather(1, idx + 1) - t["z"].gather(1, idx))
        dz, width = torch.cat(dzs), torch.cat(ws)
        res["sampler_dz_max"], res["sampler_dz_over_bin_max"] = float(dz.max()), float((dz / width.clamp(min=1e-9)).max())
        res["sampler_excess_max"] = float((dz - torch.maximum(5e-3 * width, torch.tensor(SAMPLER_ABS_FLOOR))).max())
        assert res["sampler_dz_max"] < 2e-4 and res["sampler_excess_max"] <= 0, (label, res)
    # ---- (2) downstream of the sampler on the HIP path's own sample lists
    core = _core(a, ro, rd, hip["z_vals"], near, far, var_t, air, bgv, chunk, n_samples)
    noisy = _noisy(lambda: _core(a, ro, rd, hip["z_vals"], near, far, var_t, air, bgv, chunk, n_samples))
    res["downstream"] = {}
    for name, k in PAIRS:
        err = float((hip[k] - core[k]).abs().max())
        sens = max(float((n[k] - core[k]).abs().max()) for n in noisy)
        scale = max(1.0, float(core[k].abs().max()))
        res["downstream"][name] = {"err": err, "oracle_sensitivity": sens, "bound": BASE[name] * scale + 4 * sens}
        assert err <= BASE[name] * scale + 4 * sens, (label, name, res["downstream"][name])
    # colour mask: ">8 samples seen by >=2 views" is a count of integers decided by projections only
    mm = int((hip["color_fine_mask"] != core["color_fine_mask"]).sum())
    res["downstream"]["color_mask_mismatch"] = mm
    assert mm == 0, (label, "colour mask")
    # ---- (3) end to end
    spacing = (far - near) / (n_samples - 1)
    zerr = (hip["z_vals"] - ref["z_vals"]).abs().max(1).values
    cerr = (hip["color_fine"] - ref["color_fine"]).abs().max(1).values
    derr = (hip["depth"] - ref["depth"]).abs()[:, 0]
    res["e2e"] = {"z_err_max": float(zerr.max()), "coarse_spacing": spacing, "color_max": float(cerr.max()), "depth_max": float(derr.max()),
                  "rays_color_gt_1e-4": int((cerr > 1e-4).sum()), "rays_with_coinciding_lists": int((zerr < 1e-6).sum())}
    assert float(zerr.max()) <= 1.001 * spacing, (label, res["e2e"])
    same = zerr < 1e-6
    if same.any() and strict_e2e:        # (with few samples per ray one mid-point whose nearest voxel flips under a 1e-7 depth difference is visible above the downstream bound)
        b = res["downstream"]["color"]["bound"]
        assert float(cerr[same].max()) <= b, (label, "coinciding lists", float(cerr[same].max()), b)
    dev_rays = torch.nonzero(cerr > max(1e-4, res["downstream"]["color"]["bound"]))[:, 0]
    print(f"[{label} {precision} inv_s={inv_s:.1f} air={air} bg={bg}] {len(dev_rays)} of {R} rays deviate end to end; ray ids {dev_rays.tolist()[:40]}; "
          f"their sample lists differ by {[round(float(x), 7) for x in zerr[dev_rays][:8]]}", file=sys.stderr)
    assert not strict_e2e or bool((zerr[dev_rays] > 1e-6).all()), (label, "a deviating ray has coinciding sample lists")
    # Hard regression caps on the END-TO-END error, independent of the sensitivity argument above (a sampler regression that changes many sample lists
    # must not hide inside "4 x the oracle's own sensitivity").  This comparison runs oracle and HIP on the SAME latent volume (HIP's), so only SDF-evaluation
    # differences (1e-6 class) enter: measured at BASELINE config 2 on the driver's 4,320 rays colour max 4.2e-2, q99 3.1e-3, 2.2 % of the rays above 1e-3.
    # Where the caps stand against the REFERENCE (round 5, tests/golden/ref_c2_sample.npz, make_golden_scale.SELFSENS_*): the reference's own render() against
    # itself on a volume perturbed by 1e-6 x max gives colour max 7.8e-2, q99 1.4e-2, 7.2 % of the rays above 1e-3 at config 2 -- the caps below are TIGHTER than
    # what the reference does to itself under a smaller perturbation than any second implementation of its volume build has.  The comparison from the images
    # (volumes differ by 1.6e-6 rms) is tests/test_gpu_refscale.py::test_render_end_to_end_vs_reference, bounded by the same reference-vs-reference numbers.
    res["e2e"]["frac_color_gt_1e-3"] = float((cerr > 1e-3).float().mean())
    res["e2e"]["frac_rays_with_other_lists_and_color_gt_bound"] = float(len(dev_rays)) / R
    if e2e_caps is None:
        e2e_caps = variance <= 0.3                  # the caps are calibrated in the regime the benchmark runs in (inv_s = 7.4 ... 20); a trained model's inv_s = 90 ... 665
    if e2e_caps:                                    # sharpens every list difference into an O(1) colour difference -- there the sensitivity-scaled clauses are the contract
        assert float(cerr.max()) <= 6e-2, (label, "end-to-end colour error above the 6e-2 cap", res["e2e"])
        if R >= 64:
            assert res["e2e"]["frac_color_gt_1e-3"] <= 0.06, (label, "more than 6 % of the rays deviate by more than 1e-3", res["e2e"])
            assert len(dev_rays) <= 0.25 * R, (label, "more than a quarter of the rays have other sample lists AND deviate", res["e2e"])
        if R >= 1000:
            assert float(torch.quantile(cerr, 0.99)) <= 8e-3, (label, "q99 of the end-to-end colour error above 8e-3", res["e2e"])
    if quantiles and R >= 64:
        sens = _noisy(lambda: _render(a, ro, rd, near, far, var_t, air, bgv, chunk, n_samples=n_samples, n_importance=n_importance)[0])
        ce = torch.stack([(n["color_fine"] - ref["color_fine"]).abs().max(1).values for n in sens])
        q = lambda t, x: float(torch.quantile(t.flatten(), x))
        res["e2e"]["quantiles"] = {str(x): (q(cerr, x), q(ce, x)) for x in (0.5, 0.9, 0.99)}
        res["e2e"]["mean"] = (float(cerr.mean()), float(ce.mean()))
        for x in (0.5, 0.9, 0.99):
            assert q(cerr, x) <= 4 * q(ce, x) + 1e-5, (label, x, q(cerr, x), q(ce, x))
        assert float(cerr.mean()) <= 4 * float(ce.mean()) + 1e-5, (label, res["e2e"]["mean"])
    return res
