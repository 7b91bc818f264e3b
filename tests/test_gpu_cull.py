"""GPU: tolerance-bounded colour work removal (O2345RenderIO.weight_cull) against the exhaustive path (weight_cull = 0: every occupied
sample goes through the colour network, the reference's work).  The compositing weight w = alpha * T of every sample is known after the SDF + gradient
pass; occupied samples with w < weight_cull skip the colour network and keep rgb = 0.  Asserted: everything that does not involve a colour is
BIT-IDENTICAL (depth, weights, weight sums, cdf, SDF, gradients, occupancy, valid-view counts, the per-ray colour mask, the sample lists); the colours of
the kept samples are bit-identical; culled samples are exactly the occupied ones below the threshold; and a ray's colour moves by no more than the sum of
its culled weights <= S * weight_cull (colours lie in [0, 1])."""
import importlib

import numpy as np
import pytest
import torch

from scene_util import rays_for, small_scene
from test_gpu_parity import dev, dev_scene, ops  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("one-2-3-45_amd")

EXACT = ("depth", "weights_sum", "weights_max", "depth_var", "alpha_sum", "grad_err", "color_mask", "mid_z", "dists", "pm", "sdf", "grad", "nviews", "weights", "cdf",
         "z_vals")


def _scene(s, d, dev, shift):
    W = {k: np.array(v) for k, v in s["sdfW"].items()}
    W["b2"][0] += shift                                   # a field whose zero level set the rays cross (tests/test_gpu_parity.py::test_render_trained_regime)
    scene = {k: d[k] for k in ("color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    scene["sdf_blob"] = torch.from_numpy(pkg.weights.pack_sdf_blob(W)).to(dev)
    return scene


@pytest.mark.parametrize("n_rays,inv_s,thr,precision", [(512, 665.0, 2.0 ** -24, "f16x3"), (512, 90.0, 2.0 ** -24, "fp32"), (700, 90.0, 1e-5, "f16x3"),
                                                        (8192, 665.0, 2.0 ** -24, "f16x3"), (8192, 90.0, 1e-4, "f16x3")])
def test_weight_cull_vs_exhaustive(dev, ops, n_rays, inv_s, thr, precision):
    """512 / 700 rays: the sixteen-lane sampler kernels, emission-order list; 8,192 rays: streaming kernels and the culled list grouped by visibility."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    scene = dict(_scene(s, d, dev, -0.2), sdf_precision=precision, color_precision=precision)
    ro, rd = rays_for(s, n_rays, seed=33, center=True)
    tro, trd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    st_a, st_b = ops.color_stats_buffer(dev), ops.color_stats_buffer(dev)
    full = ops.render_rays(scene, tro, trd, near, far, 64, 64, inv_s, 1.0, 1.0, qcam, want_z=True, weight_cull=0.0, color_stats=st_a)
    cut = ops.render_rays(scene, tro, trd, near, far, 64, 64, inv_s, 1.0, 1.0, qcam, want_z=True, weight_cull=thr, color_stats=st_b)
    for k in EXACT:
        assert torch.equal(full[k], cut[k]), k
    occ = full["pm"] > 0
    w = full["weights"]
    culled = occ & (w < thr)
    kept = occ & ~culled
    assert int(culled.sum()) > 0 and int(kept.sum()) > 0, (int(culled.sum()), int(kept.sum()))
    assert torch.equal(cut["rgb"][kept], full["rgb"][kept]), "colours of the kept samples"
    assert float(cut["rgb"][culled].abs().sum()) == 0.0, "culled samples keep colour 0"
    assert float(full["rgb"][occ].min()) >= 0.0 and float(full["rgb"][occ].max()) <= 1.0 + 1e-6       # what the bound rests on
    bound = (w * culled).sum(0)                                                                       # per ray: the culled weights
    assert float(bound.max()) <= 128 * thr
    derr = (full["color"] - cut["color"]).abs().max(1).values
    assert bool((derr <= bound * (1.0 + 1e-6) + 2e-7).all()), (float(derr.max()), float(bound.max()))  # (+ fp32 summation noise of the 128-term colour sums)
    a, b = ops.color_stats_read(st_a), ops.color_stats_read(st_b)
    assert b["pairs_network"] < a["pairs_network"] and b["tiles"] < a["tiles"]
    print(f"[cull] {n_rays} rays inv_s {inv_s} thr {thr:.1e} {precision}: {int(culled.sum())} of {int(occ.sum())} occupied samples culled "
          f"({float(culled.sum()) / float(occ.sum()):.3f}), colour moves by {float(derr.max()):.2e} <= {float(bound.max()):.2e}; "
          f"(tile, view) pairs {a['pairs_network']} -> {b['pairs_network']}")


def test_weight_cull_default_and_knob(dev, ops, monkeypatch):
    """The default threshold is config.WEIGHT_CULL (2^-24 unless O2345_WEIGHT_CULL says otherwise); scene['weight_cull'] and the call argument override it."""
    config = importlib.import_module("one-2-3-45_amd.config")
    assert config.weight_cull() == config.WEIGHT_CULL and config.weight_cull(0) == 0.0
    with pytest.raises(ValueError):
        config.weight_cull(1.5)
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    scene = _scene(s, d, dev, -0.2)
    ro, rd = rays_for(s, 256, seed=3, center=True)
    tro, trd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    r = lambda **kw: ops.render_rays(scene, tro, trd, near, far, 64, 64, 665.0, 1.0, 1.0, qcam, **kw)
    monkeypatch.setattr(config, "WEIGHT_CULL", 2.0 ** -24)
    default, explicit, off = r(), r(weight_cull=2.0 ** -24), r(weight_cull=0.0)
    assert torch.equal(default["rgb"], explicit["rgb"]) and not torch.equal(default["rgb"], off["rgb"])
    assert torch.equal(ops.render_rays(dict(scene, weight_cull=0.0), tro, trd, near, far, 64, 64, 665.0, 1.0, 1.0, qcam)["rgb"], off["rgb"])
    monkeypatch.setattr(config, "WEIGHT_CULL", 0.0)
    assert torch.equal(r()["rgb"], off["rgb"])
