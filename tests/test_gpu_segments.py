"""GPU: O2345RenderIO.segment_rays (ABI 2.1) -- R rays that are CONSECUTIVE render() calls of G rays each, evaluated in one call.
The reference's two per-CALL rules must then hold PER SEGMENT:

  cat_z_vals   (sparse_neus_renderer.py:137)      the SDF of a round's new samples is evaluated only if MORE THAN ONE of the call's new samples lies
                                                  inside the mask, otherwise all of them keep 100;
  render_core  (sparse_neus_renderer.py:222-223)  a call without any occupied mid-point evaluates its first 100 points anyway.

Contract: every output of the segmented call -- per ray, per sample, and the per-call scalars -- is BIT-IDENTICAL to the separate calls, including
segments constructed so that each rule fires in one segment and not in its neighbour (a one-voxel mask: a segment with ONE crossing ray has exactly one
new sample inside the mask per round, a segment with 64 crossing rays has 64, a segment of rays that miss the volume has no occupied point at all)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import recon as O
from scene_util import rays_for, sdfW_t, small_scene
from test_gpu_parity import dev, dev_scene, ops  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("one-2-3-45_amd")

PER_RAY = ("color", "depth", "weights_sum", "weights_max", "depth_var", "alpha_sum", "grad_err", "color_mask")
PER_SAMPLE = ("mid_z", "dists", "pm", "sdf", "grad", "rgb", "nviews", "weights", "cdf", "z_vals")


def _compare(ops_, scene, ro, rd, near, far, ns, ni, inv_s, qcam, G, t_rand=None, label=""):
    R = ro.shape[0]
    whole = ops_.render_rays(scene, ro, rd, near, far, ns, ni, inv_s, 1.0, 1.0, qcam, want_z=True, want_scalars=True, t_rand=t_rand, segment_rays=G)
    parts = []
    for k, c0 in enumerate(range(0, R, G)):
        tr = None if t_rand is None else t_rand[c0:c0 + G].contiguous()
        part = ops_.render_rays(scene, ro[c0:c0 + G].contiguous(), rd[c0:c0 + G].contiguous(), near, far, ns, ni, inv_s, 1.0, 1.0, qcam, want_z=True,
                                want_scalars=True, t_rand=tr)
        for key in PER_RAY:
            assert torch.equal(part[key], whole[key][c0:c0 + G]), (label, key, k)
        for key in PER_SAMPLE:
            assert torch.equal(part[key], whole[key][:, c0:c0 + G]), (label, key, k)
        assert torch.equal(part["scalars"], whole["scalars"][k]), (label, "scalars", k, part["scalars"], whole["scalars"][k])
        parts.append(part)
    return whole, parts


@pytest.mark.parametrize("form", ["group", "stream"])
def test_rules_fire_per_segment(dev, ops, form, lib_instance):
    lib_instance({"O2345_RAY_STREAM_MIN": "1" if form == "stream" else "1000000"})
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc, D = s["sc"], s["D"]
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "cmaps", "proj", "cam_pos")}
    Wt = sdfW_t(s["sdfW"])
    c = (2 * torch.arange(D) + 1) / D - 1                                     # voxel centres of the nearest-mask lookup
    ctr = torch.stack(torch.meshgrid(c, c, c, indexing="ij"), -1).reshape(-1, 3)
    inside = torch.nonzero(O.sdf(ctr, s["dense"][0], Wt)[0][:, 0].reshape(D, D, D) < -0.05)
    ix, iy, iz = (int(v) for v in inside[inside.shape[0] // 3])
    mask = torch.zeros(D, D, D)
    mask[ix, iy, iz] = 1                                                      # ONE occupied voxel (tests/test_gpu_edges_and_fullsize.py's construction)
    scene["maskvol"] = mask.reshape(-1).contiguous().to(dev)
    cross_o = torch.tensor([float(c[ix]) - 1.0 + 0.02, float(c[iy]), float(c[iz])])
    cross_d = torch.tensor([1.0, 0.0, 0.0])
    miss_o, miss_d = torch.tensor([5.0, 5.0, 5.0]), torch.nn.functional.normalize(torch.tensor([1.0, 0.2, 0.1]), dim=0)
    G = 64

    def segment(n_cross):
        o = miss_o[None].repeat(G, 1); dd = miss_d[None].repeat(G, 1)
        o[:n_cross] = cross_o; dd[:n_cross] = cross_d
        return o, dd
    segs = [segment(1), segment(0), segment(64), segment(2), segment(0)[0][:40], None]
    ro = torch.cat([segment(1)[0], segment(0)[0], segment(64)[0], segment(2)[0], segment(1)[0][:40]]).to(dev).contiguous()   # last segment: 40 rays
    rd = torch.cat([segment(1)[1], segment(0)[1], segment(64)[1], segment(2)[1], segment(1)[1][:40]]).to(dev).contiguous()
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    inv_s = float(np.exp(2.0))
    whole, parts = _compare(ops, scene, ro, rd, 0.1, 2.0, 16, 64, inv_s, qcam, G, label=form)
    # the rules did fire, and differently per segment: segment 1 (nothing occupied) evaluated its first 100 points in the reference's ray-major order
    # (pts_mask_bool[:100] on the flattened [N_rays * S] mask, :222-223) -- S = 80 < 100: all 80 samples of its first ray and samples 0..19 of its second ...
    assert float(parts[1]["pm"].sum()) == 0 and float((parts[1]["sdf"][:, 0] != 100).float().mean()) > 0.9 and float(parts[1]["scalars"][3]) == 100.0
    assert float((parts[1]["sdf"][:20, 1] != 100).float().mean()) > 0.9 and float((parts[1]["sdf"][20:, 1] == 100).float().mean()) == 1.0
    assert float((whole["sdf"][:, G + 2:2 * G] == 100).float().mean()) == 1.0
    # ... and the crossing ray's sample list in segment 0 (alone: at most one new sample inside the mask per round -> the new samples keep sdf = 100) differs
    # from the same ray's list in segment 2 (64 copies: the rule does not fire) and in segment 3 (two copies)
    z0, z2, z3 = whole["z_vals"][:, 0], whole["z_vals"][:, 2 * G], whole["z_vals"][:, 3 * G]
    assert float((z0 - z2).abs().max()) > 1e-3, "cat_z_vals' rule must separate the lone crossing ray from the crowd"
    assert torch.equal(z2, z3)
    assert torch.equal(whole["z_vals"][:, 4 * G], z0)                          # the short last segment: one crossing ray again


@pytest.mark.parametrize("R,G", [(1600, 512), (8192, 512), (5000, 1024)])
def test_segmented_call_equals_separate_calls_on_a_real_scene(dev, ops, R, G):
    """An image's rays through the small scene, perturbed coarse samples: R = 1,600 (sixteen-lane kernels), 8,192 / 5,000 (streaming kernels, list grouped by
    visibility, weight culling on): the fused segmented call == the separate 512- / 1,024-ray calls, every output."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    ro, rd = rays_for(s, R, seed=17, center=True)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    t_rand = torch.rand(R, 64, generator=torch.Generator().manual_seed(3)).to(dev)
    whole, parts = _compare(ops, scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), near, far, 64, 64, 90.0, qcam, G, t_rand=t_rand, label=f"{R}/{G}")
    assert float(whole["weights_sum"].max()) > 0.5 and tuple(whole["scalars"].shape) == ((R + G - 1) // G, 4)


def test_segment_argument_checks(dev, ops):
    s = small_scene()
    d = dev_scene(s, dev, ops)
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    ro, rd = rays_for(s, 128, seed=1)
    qcam = torch.zeros(3, device=dev)
    with pytest.raises(ValueError, match="multiple of 64"):
        ops.render_rays(scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), 0.5, 1.8, 64, 64, 7.4, 1.0, 1.0, qcam, segment_rays=100)
    nr = torch.full((128,), 0.5, device=dev)
    with pytest.raises(RuntimeError, match="segment_rays needs one near / far pair"):
        ops.render_rays(scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), nr, nr + 1.3, 64, 64, 7.4, 1.0, 1.0, qcam, sample_dist=1.3 / 64, segment_rays=64)
