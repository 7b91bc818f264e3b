"""GPU: the HIP path against the REFERENCE ITSELF at BASELINE scale -- no oracle in between.

tests/golden/ref_c1.npz (BASELINE config 1, exactly), ref_c2_sample.npz (config 2: 128^3, eight 512-ray chunks of the 512^2 image, + two chunks with a
trained model's variance) and ref_refcfg_sample.npz (the reference configuration: 32 views, 96^3) hold what the reference's own modules computed from the
seeded images (tests/golden/make_golden_scale.py, run in the build container): FeatureNet -> fused pyramid -> get_conditional_volume -> render() in the
runner's chunks -> extract_fields.  Here the same seeds go through the C ABI and every stage is compared with the file directly (tests/refscale_util.py):

  volume      kept-voxel set bit-exact (all D^3 mask bits), fused pyramid / compressed maps / dense volume samples within a relative tolerance
  sampler     o2345_ray_upsample on the reference's own per-round (z, sdf)              -> the reference's 16 new depths per ray and round
  downstream  render_core on the reference's own sample lists (stage entries, ops.render_core) -> the reference's colour / depth / weights / masks
  end to end  render() per chunk -> the reference's images: sample lists within one coarse section, hard caps on the error distribution
  field       extract_fields -> the reference's u: same sign pattern => the same triangles
"""
import json
import sys

import pytest

import refscale_util as RU

pytestmark = pytest.mark.gpu

# ---- the stated tolerances, HIP vs reference at these sizes (max abs error / max(1, max |reference|) unless noted; measured values: DESIGN.md section 4)
TOL = dict(fmaps=2e-5, feats16=2e-5, dense=5e-5, dense_rms=3e-6,
           sampler_abs=2e-4, sampler_bin=5e-3, sampler_floor=5e-7,
           core_color=1e-4, core_depth=6e-5, core_weights=6e-5, core_sdf=5e-5, core_grad=2e-4,
           u=5e-5)
# ---- end to end: HIP-vs-reference against the REFERENCE-VS-ITSELF distributions stored in the golden files (its own latent volume perturbed by 3e-6 and by 1e-6 of
# max|volume| rms; HIP's volume differs from the reference's by 0.7e-6 rms at config 1 / 32 views and 1.6e-6 rms at config 2).  Caps as multiples of the stored
# quantiles (q50, q90, q99), of the maximum, of the fraction of rays above 1e-3 and of the largest sample-list difference.  Measured multiples (round 6, MI355X) in
# the comment of each row; DESIGN.md section 4 has the table.  VERDICT r5 item 3 asked for 1.0 everywhere: config 1 and the reference configuration are there (HIP
# sits 2 - 10x INSIDE the reference's self-sensitivity), config 2's median does not (1.8x: the larger volume's build error is closer to the 3e-6 noise level), so its
# caps are the measured multiples + a third, instead of the 3x / 2x of round 5.
E2E_CAPS = {
    "c1": dict(q=(1.0, 1.0, 1.0), max=1.0, frac=1.0, z=1.0, lower_level=3.0),        # measured 0.09 / 0.23 / 0.13, max 0.15, frac 0.20, z 0.20; vs the 1e-6 level 0.40 / 0.62 / 0.32
    "ref": dict(q=(1.0, 1.0, 1.0), max=1.0, frac=1.0, z=1.0, lower_level=3.0),       # measured 0.40 / 0.39 / 0.45, max 0.39, frac 0.51, z 0.29; vs the 1e-6 level 1.2 / 1.0 / 0.68
    "c2": dict(q=(2.5, 1.5, 2.0), max=1.5, frac=1.5, z=1.0, lower_level=None),       # measured 1.80 / 1.01 / 1.01, max 0.76, frac 1.11, z 0.56; trained variance 1.65 / 0.88 / 1.51, max 1.08
}
LOD1_CAPS = dict(q=(2.0, 1.0, 1.0), max=1.0, z=1.0)                                  # measured 1.54 / 0.80 / 0.49, max 0.48, z 1.00 (both one coarse section)


def show(G, what, res):
    print(f"[refscale {G['name']}] {what}: {json.dumps(res)}", file=sys.stderr)


@pytest.fixture(scope="module", params=["c1", "c2", "ref"])
def G(request):
    return RU.load(request.param)


def test_volume_build_vs_reference(G):
    """get_conditional_volume from the IMAGES (sparse_sdf_network.py:286-400 after FeatureNet + obtain_pyramid_feature_maps): every mask bit, samples of the
    fused pyramid, of the compressed maps and of the dense latent volume."""
    r = RU.volume(G)
    show(G, "volume vs REFERENCE", r)
    assert r["mask_bits_exact"] and r["kept_voxels"] == r["kept_voxels_reference"], "kept-voxel set differs from the reference's valid_mask_volume"
    assert r["fused_pyramid"] < TOL["fmaps"] and r["compressed_maps"] < TOL["feats16"] and r["dense_volume"] < TOL["dense"], r
    # the volume differs from the reference's by no more than the noise the reference's own end-to-end sensitivity was measured with (3e-6 rms)
    assert r["dense_volume_rms"] <= TOL["dense_rms"], r


def test_sampler_stage_on_the_references_own_inputs(G):
    """up_sample + sample_pdf (sparse_neus_renderer.py:73-115, render_utils.py:8-51): o2345_ray_upsample driven with the per-round (z, sdf) the REFERENCE's
    render() had in its first chunk -> the reference's new depths: within 2e-4 absolute and within max(5e-3 of the bin, 5e-7)."""
    r = RU.sampler(G, TOL["sampler_bin"], TOL["sampler_floor"])
    show(G, "sampler on the reference's inputs", r)
    assert r["rounds"] == 4 and r["dz_max"] < TOL["sampler_abs"] and r["excess_max"] <= 0, r


def test_render_core_on_the_references_own_sample_lists(G):
    """Everything downstream of the sampler (render_core, :171-455: SDF + gradient, Projector + GeneralRenderingNetwork, NeuS compositing) evaluated by the HIP
    stage entries ON THE REFERENCE'S sample lists -> the reference's per-ray and per-sample results, chunk by chunk."""
    for e in RU.core(G):
        show(G, "render_core on the reference's lists", e)
        # a trained model's inv_s multiplies every SDF difference inside the sigmoid: the bound scales with it
        amp = max(1.0, e["inv_s"] / 20.0)
        assert e["color_mask_mismatches"] == 0 and e["occupancy_exact"] and e["defaults_exact"], e
        assert e["sdf"] < TOL["core_sdf"] and e["grad"] < TOL["core_grad"], e
        assert e["color"] < TOL["core_color"] * amp and e["depth"] < TOL["core_depth"] * amp, e
        assert max(e["weights"], e["weights_sum"], e["weights_max"], e["depth_var"]) < TOL["core_weights"] * amp, e


def test_render_end_to_end_vs_reference(G):
    """render() (sparse_neus_renderer.py:457-635) per chunk, exactly as the trainer's loop calls it -> the reference's images, both sides FROM THE IMAGES.
    The hierarchical sampler amplifies fp32-class differences: the golden file holds the REFERENCE AGAINST ITSELF on a latent volume perturbed by 3e-6 rms
    (max 1.5e-5 -- HIP's volume differs from the reference's by less: test_volume_build_vs_reference), and that alone moves sample lists by several coarse
    sections and 15 % of the rays by more than 1e-3 in colour at config 2.  Asserted: the colour mask exact; rays whose sample lists coincide agree as
    tightly as the downstream test; and HIP-vs-reference stays inside the reference-vs-reference distribution -- E2E_CAPS above: 1.0x at config 1 and at the
    reference configuration (plus 3x of the 1e-6 level), the measured multiples + a third at config 2 (caps derived from the reference, not from the oracle or
    from HIP's own output)."""
    for e in RU.end_to_end(G):
        show(G, "render() end to end vs REFERENCE", e)
        assert e["color_mask_mismatches"] == 0, e
        amp = max(1.0, e["inv_s"] / 20.0)
        assert e["color_err_max_on_coinciding_lists"] <= TOL["core_color"] * amp, e
        own = e["reference_vs_itself_on_a_noisy_volume"]
        if own is not None:                      # (also for the trained variance: inv_s = 148 turns a list difference into an O(1) colour difference -- in the reference too)
            cap = E2E_CAPS[G["name"]]
            ratios = [hip / ref for hip, ref in zip(e["color_err_q50_q90_q99_max"], own["color_err_q50_q90_q99_max"])]
            show(G, "end to end: multiples of the reference's own sensitivity (q50, q90, q99, max)", [round(r, 3) for r in ratios])
            for hip, ref, c in zip(e["color_err_q50_q90_q99_max"][:3], own["color_err_q50_q90_q99_max"][:3], cap["q"]):
                assert hip <= c * ref + 1e-6, (cap, e)
            assert e["color_err_q50_q90_q99_max"][3] <= cap["max"] * own["color_err_q50_q90_q99_max"][3], (cap, e)
            assert e["frac_rays_color_gt_1e-3"] <= cap["frac"] * own["frac_rays_color_gt_1e-3"] + 1.0 / e["rays"], (cap, e)
            assert e["z_err_max"] <= cap["z"] * own["z_err_max"] + 1e-5, (cap, e)
            low = e["reference_vs_itself_at_the_lower_noise_level"]
            if cap["lower_level"] is not None and low is not None and e["variance"] < 0.3:          # the 1e-6 level as a second bound
                for hip, ref in zip(e["color_err_q50_q90_q99_max"], low["color_err_q50_q90_q99_max"]):
                    assert hip <= cap["lower_level"] * ref + 1e-6, (cap, e)


def test_extract_fields_vs_reference(G):
    """extract_fields (sparse_neus_renderer.py:881-905): u = -sdf on the reference's lattice.  Config 1: the whole 64^3 lattice through the fused lattice
    kernel; config 2 / reference configuration: the central 64^3 block of the 256^3 lattice.  The same sign at every node => marching cubes (exact for an
    identical sign pattern) yields the reference's triangles."""
    r = RU.field(G)
    show(G, "extract_fields vs REFERENCE", r)
    assert r["field_err_max"] < TOL["u"] * max(1.0, r["field_scale"]), r
    assert r["sign_flips"] <= 2 and r["abs_u_reference_at_flips_max"] <= r["field_err_max"], "a lattice node may change sign only inside the field's own error band"
    if r["sign_flips"] == 0:
        assert r["triangles_identical"] and r["vertex_shift_max_cells"] < 0.02, r


def test_vertex_colours_vs_reference(G):
    """validate_colored_mesh (trainer_generic.py:1309-1363; row a25): the view-independent Projector (query direction = the SDF's gradient, autograd in the
    reference, analytic here) + GeneralRenderingNetwork on up to 4,000 vertices of the reference field's surface -> the reference's vertex colours; the
    valid-view count of every vertex exact."""
    r = RU.vertex_colours(G)
    show(G, "vertex colours vs REFERENCE", r)
    assert r is not None and r["vertices"] > 1000 and r["vertices_seen_by_two_or_more_views"] > 100, r
    assert r["valid_view_count_mismatches"] == 0 and r["rgb"] < 2e-4, r


def test_lod1_sparse_256_cubed_vs_reference():
    """BASELINE config 5's sparse 256^3 level: the reference's coarse-to-fine path (trainer_generic.py:437-491; rows a13, a26, f2 of SURVEY 8) on config 2's
    scene, through the mirror modules, against tests/golden/ref_c5_lod1_sample.npz (the imported reference, make_golden_scale.py c5): get_sdf_volume at the
    1.17 M kept voxels, the pruning selection (60,324 voxels: the mirror's own selection may differ from the reference's only where |sdf| is within the
    field's error band of the 0.02 threshold), the lod-1 volume built on the REFERENCE's selection (every one of the 16.7 M mask bits, dense samples), the
    lod-1 SDF, and one 512-ray chunk of the lod-1 val loop."""
    r = RU.lod1()
    print(f"[refscale c5 / lod 1] {json.dumps(r)}", file=sys.stderr)
    assert r["l0_sdf_volume"] < 2e-5, r
    assert r["l0_threshold_disagreements"] <= 20 and r["l0_threshold_disagreements_band"] <= 5e-5, r
    assert r["pruned_voxels_symmetric_difference"] <= 60, r          # (a threshold flip reaches the selection through the 7^3 dilation)
    assert r["l1_mask_bits_exact"] and r["l1_kept_voxels"][0] == r["l1_kept_voxels"][1], r
    assert r["l1_dense_volume"] < 5e-5 and r["l1_sdf"] < 5e-5, r
    c = r["render_core_on_reference_lists"]
    assert c["color_mask_mismatches"] == 0 and c["color"] < TOL["core_color"] and c["depth"] < TOL["core_depth"], r
    # per-sample weights: every occupied sample of the sparse level lies within a few voxels of the surface, where a 1.7e-5 SDF difference (the lod-1 volume
    # differs by 1.8e-5 from the reference's) moves an opacity of order 1e-5 ... 1e-2 by up to 1e-4; the per-ray sums stay at the lod-0 bound x 2
    assert c["weights"] < 5e-4 and c["weights_sum"] < 2 * TOL["core_weights"], r
    e = r["render_end_to_end"]
    assert e["color_mask_mismatches"] == 0 and e["color_err_max_on_coinciding_lists"] <= TOL["core_color"], r
    for hip, ref, c in zip(e["color_err_q50_q90_q99_max"][:3], e["reference_vs_itself_color_err_q50_q90_q99_max"][:3], LOD1_CAPS["q"]):
        assert hip <= c * ref + 1e-6, r
    assert e["color_err_q50_q90_q99_max"][3] <= LOD1_CAPS["max"] * e["reference_vs_itself_color_err_q50_q90_q99_max"][3], r
    assert e["z_err_max"] <= LOD1_CAPS["z"] * e["reference_vs_itself_z_err_max"] + 1e-5, r
