"""GPU: edge cases (empty / single / ragged inputs, the reference's quirk branches) and size-independent properties at
BASELINE config-2 sizes (8 x 256^2 views, 128^3 volume, 512 x 512 rays, 256^3 mesh grid), where the CPU oracle is too slow."""
import importlib

import numpy as np
import pytest
import torch

from oracle import recon as O
from scene_util import small_scene, sdfW_t

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("one-2-3-45_amd")
ops = importlib.import_module("one-2-3-45_amd.ops")
pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
dev = torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------------------ edge cases
def test_empty_volume_and_single_view():
    s = small_scene()
    D, V, H, W = s["D"], s["V"], s["H"], s["W"]
    aff = torch.from_numpy(s["sc"]["affine_mats"]).to(dev).clone()
    aff[:, 2, :] *= -1                                  # every voxel behind every camera -> nothing is kept
    cnt, row, coords, n = ops.costvol_index(aff, V, H, W, (D, D, D), s["voxel_size"], s["sc"]["partial_vol_origin"])
    assert n == 0 and coords.shape == (0, 4) and int(cnt.max()) <= 1 and int((row >= 0).sum()) == 0
    nhwc = ops.nchw_to_nhwc(torch.from_numpy(s["f16"]).to(dev))
    rows = ops.costvol_gather(nhwc, aff, (D, D, D), s["voxel_size"], s["sc"]["partial_vol_origin"], cnt, coords)
    assert rows.shape == (0, 32)
    costreg = importlib.import_module("one-2-3-45_amd.costreg")
    out = costreg.CostRegNet(s["costreg_sd"], dev).forward(rows, coords, row, (D, D, D))
    assert out.shape == (0, 16)
    cl, cf, mask = ops.scatter_dense(out, row, (D, D, D))
    assert float(mask.sum()) == 0 and float(cl.abs().sum()) == 0
    # one view: minimum_visible_views = min(1, V-1) = 0 (sparse_sdf_network.py:303)
    aff1 = torch.from_numpy(s["sc"]["affine_mats"][:1]).to(dev)
    c_ref, v_ref, _ = O.costvol(torch.from_numpy(s["f16"][:1]), aff1.cpu(), [D, D, D], s["voxel_size"], torch.from_numpy(s["sc"]["partial_vol_origin"]),
                                min_views=0)
    cnt, row, coords, n = ops.costvol_index(aff1, 1, H, W, (D, D, D), s["voxel_size"], s["sc"]["partial_vol_origin"], min_views=0)
    assert torch.equal(coords.cpu(), c_ref)
    rows = ops.costvol_gather(nhwc[:1].contiguous(), aff1, (D, D, D), s["voxel_size"], s["sc"]["partial_vol_origin"], cnt, coords)
    assert (rows.cpu() - v_ref).abs().max() < 2e-5 * max(1.0, v_ref.abs().max().item())


def test_non_cubic_volume_index():
    s = small_scene()
    V, H, W = s["V"], s["H"], s["W"]
    dims = (12, 20, 9)
    aff = torch.from_numpy(s["sc"]["affine_mats"]).to(dev)
    c_ref, v_ref, cnt_ref = O.costvol(torch.from_numpy(s["f16"]), aff.cpu(), list(dims), 0.11, torch.tensor([-0.6, -1.0, -0.5]))
    cnt, row, coords, n = ops.costvol_index(aff, V, H, W, dims, 0.11, [-0.6, -1.0, -0.5])
    assert torch.equal(coords.cpu(), c_ref) and torch.equal(cnt.cpu(), cnt_ref.to(torch.uint8))
    rows = ops.costvol_gather(ops.nchw_to_nhwc(torch.from_numpy(s["f16"]).to(dev)), aff, dims, 0.11, [-0.6, -1.0, -0.5], cnt, coords)
    assert (rows.cpu() - v_ref).abs().max() < 2e-5 * max(1.0, v_ref.abs().max().item())


def _tiny_scene(s):
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev)
    sc = s["sc"]
    proj, cam_pos = pipeline.camera_terms(t(sc["intrinsics"]), t(sc["w2cs"]))
    return dict(sdf_blob=t(pkg.weights.pack_sdf_blob(s["sdfW"])),
                color_mfma_blob=t(pkg.weights.pack_color_mfma_blob(s["color_sd"])), vol_cl=s["dense"][0].permute(1, 2, 3, 0).contiguous().to(dev),
                maskvol=s["mask"][0, 0].reshape(-1).contiguous().to(dev), cmaps=ops.pack_color_maps(t(s["fmaps"]), t(sc["images"])), proj=proj,
                cam_pos=cam_pos)


def test_rays_that_miss_everything_and_single_ray():
    s = small_scene()
    scene = _tiny_scene(s)
    qcam = torch.from_numpy(s["sc"]["query_c2w"][:3, 3].copy()).to(dev)
    # rays far outside the volume: no sample is occupied -> render_core's "first 100 points" quirk, colour = background
    ro = torch.tensor([[5.0, 5.0, 5.0]] * 3, device=dev)
    rd = torch.nn.functional.normalize(torch.tensor([[1.0, 0.2, 0.1], [0.0, 1.0, 0.0], [0.3, 0.3, 0.9]], device=dev), dim=-1)
    o = ops.render_rays(scene, ro, rd, 0.1, 2.0, 64, 64, 7.4, 1.0, 1.0, qcam)
    assert float(o["weights"].abs().max()) == 0 and torch.allclose(o["color"], torch.ones(3, 3, device=dev))
    assert float(o["pm"].sum()) == 0 and int(o["color_mask"].sum()) == 0
    assert float((o["sdf"][:100, 0] != 100).float().mean()) > 0.9          # ray 0's first 100 samples were evaluated anyway (:222-223)
    assert float((o["sdf"][:, 1:] == 100).float().mean()) == 1.0
    # the same rule in the large-batch form of the round kernels (one lane per ray, >= O2345_RAY_STREAM_MIN rays): applied by the workgroup that finishes last
    n_big = 4096 + 37
    ob = ops.render_rays(scene, ro[:1].repeat(n_big, 1).contiguous(), rd[:1].repeat(n_big, 1).contiguous(), 0.1, 2.0, 64, 64, 7.4, 1.0, 1.0, qcam, want_scalars=True)
    assert float(ob["pm"].sum()) == 0 and torch.allclose(ob["color"], torch.ones(n_big, 3, device=dev))
    assert float((ob["sdf"][:100, 0] != 100).float().mean()) > 0.9 and float((ob["sdf"][:, 1:] == 100).float().mean()) == 1.0
    assert torch.equal(ob["sdf"][:, 0], o["sdf"][:, 0]) and float(ob["scalars"][3]) == 100.0          # evaluated points: the 100 forced ones
    # a single ray through the object equals the same ray inside a batch
    ro1, rd1 = torch.from_numpy(s["sc"]["query_c2w"][:3, 3].copy())[None].to(dev), torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    near, far = float(s["sc"]["query_near_far"][0]), float(s["sc"]["query_near_far"][1])
    a = ops.render_rays(scene, ro1, rd1, near, far, 64, 64, 7.4, 1.0, 1.0, qcam)
    b = ops.render_rays(scene, ro1.repeat(5, 1), rd1.repeat(5, 1), near, far, 64, 64, 7.4, 1.0, 1.0, qcam)
    assert torch.allclose(a["color"][0], b["color"][3], atol=1e-6) and torch.allclose(a["depth"][0], b["depth"][3], atol=1e-6)


@pytest.mark.parametrize("form", ["group", "stream"])
def test_cat_z_vals_rule_for_a_single_occupied_new_sample(form, lib_instance):
    """cat_z_vals (:137) evaluates the SDF of a round's new samples only if MORE THAN ONE of them is inside the mask; otherwise all sixteen keep 100.
    The render call applies the rule inside the round kernel (the workgroup that finishes last, csrc/render.hip round_epilogue).  Constructed case: the
    mask holds ONE voxel inside the object (SDF < 0) and one ray crosses it along x with 16 coarse samples (spacing 0.127 > the voxel's 0.1): one
    coarse sample and, in every round, exactly one new sample lie in the voxel.  The oracle WITHOUT the rule puts its samples elsewhere (lists differ
    by ~0.6), so agreement with the oracle proper shows that the rule fired -- in both forms of the round kernels."""
    from scene_util import color_t
    lib_instance({"O2345_RAY_STREAM_MIN": "1" if form == "stream" else "1000000"})
    s = small_scene()
    scene = _tiny_scene(s)
    sc, D = s["sc"], s["D"]
    Wt = sdfW_t(s["sdfW"])
    c = (2 * torch.arange(D) + 1) / D - 1                                     # voxel centres of the nearest-mask lookup
    ctr = torch.stack(torch.meshgrid(c, c, c, indexing="ij"), -1).reshape(-1, 3)
    inside = torch.nonzero(O.sdf(ctr, s["dense"][0], Wt)[0][:, 0].reshape(D, D, D) < -0.05)
    assert inside.shape[0] >= 4
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy())
    a = dict(volume=s["dense"][0], W=Wt, RW=color_t(s["color_sd"]), variance=torch.tensor(0.2), feat_maps=torch.from_numpy(s["fmaps"]),
             color_maps=torch.from_numpy(sc["images"]), w2cs=torch.from_numpy(sc["w2cs"]), K=torch.from_numpy(sc["intrinsics"]), img_wh=(s["W"], s["H"]),
             query_c2w=torch.from_numpy(sc["query_c2w"]))
    inv_s = float(torch.exp(a["variance"] * 10.0))

    def oracle(mask, ro, rd, min_valid):
        O.CAT_Z_MIN_VALID = min_valid
        try:
            trace = []
            r = O.render(ro, rd, torch.tensor(0.1), torch.tensor(2.0), a["volume"], mask, a["W"], a["RW"], a["variance"], a["feat_maps"], a["color_maps"],
                         a["w2cs"], a["K"], a["img_wh"], a["query_c2w"], n_samples=16, n_importance=64, trace=trace)
        finally:
            O.CAT_Z_MIN_VALID = 1
        return r, trace

    fired = 0
    for vi, off in ((0, 0.0), (inside.shape[0] // 3, 0.02), (2 * inside.shape[0] // 3, -0.03), (inside.shape[0] - 1, 0.0)):
        ix, iy, iz = (int(v) for v in inside[vi])
        mask = torch.zeros(D, D, D)
        mask[ix, iy, iz] = 1
        ro = torch.tensor([[float(c[ix]) - 1.0 + off, float(c[iy]), float(c[iz])]])
        rd = torch.tensor([[1.0, 0.0, 0.0]])
        ref, trace = oracle(mask, ro, rd, 1)
        per_round = [int(O.mask_nearest(mask, (ro[:, None] + rd[:, None] * t["new_z"][..., None]).reshape(-1, 3)).sum()) for t in trace]
        without, _ = oracle(mask, ro, rd, 0)
        sensitive = float((ref["z_vals"] - without["z_vals"]).abs().max())
        sm = dict(scene, maskvol=mask.reshape(-1).contiguous().to(dev))
        o = ops.render_rays(sm, ro.to(dev), rd.to(dev), 0.1, 2.0, 16, 64, inv_s, 1.0, 1.0, qcam.to(dev), want_z=True)
        dz = float((o["z_vals"].t().cpu() - ref["z_vals"]).abs().max())
        print(f"[{form}] voxel {ix, iy, iz} offset {off}: occupied new samples per round {per_round}, lists differ by {dz:.2e} from the oracle, "
              f"{sensitive:.2e} between the oracle with and without the rule")
        assert dz < 2e-4
        assert (o["color"].cpu() - ref["color_fine"]).abs().max() < 1e-3 and (o["depth"].cpu()[:, None] - ref["depth"]).abs().max() < 1e-3
        if 1 in per_round and sensitive > 0.1:
            fired += 1
    assert fired >= 2, "the constructed cases must exercise the rule"


def test_zero_points_and_tiny_grids():
    s = small_scene()
    scene = _tiny_scene(s)
    r = ops.sdf_mlp(scene["sdf_blob"], scene["vol_cl"], torch.zeros(0, 3, device=dev), variant=2)
    assert r["sdf"].shape == (0,) and r["grad"].shape == (0, 3)
    rgb, nv = ops.color_points(scene["color_mfma_blob"], scene["vol_cl"], scene["maskvol"], scene["cmaps"], scene["proj"], scene["cam_pos"],
                               torch.zeros(0, 3, device=dev), query_cam=torch.zeros(3, device=dev), mfma=True)
    assert rgb.shape == (0, 3)
    u = torch.tensor([[[-1.0, 1.0], [1.0, 1.0]], [[1.0, 1.0], [1.0, 1.0]]], device=dev)      # one inside corner in a 2x2x2 grid
    v, t = ops.marching_cubes(u, 0.0)
    assert v.shape == (3, 3) and t.shape == (1, 3)
    idx = torch.zeros(10, dtype=torch.int32, device=dev)
    out = {"sdf": torch.full((10,), 100.0, device=dev)}
    ops.sdf_mlp(scene["sdf_blob"], scene["vol_cl"], torch.zeros(10, 3, device=dev), variant=0, index=idx,
                n_dev=torch.zeros(1, dtype=torch.int32, device=dev), out=out)
    assert float((out["sdf"] == 100).float().mean()) == 1.0          # device-side count 0: nothing evaluated


@pytest.mark.parametrize("kernel", ["pts", "tiles"])
@pytest.mark.parametrize("V", [1, 2, 3, 7])
def test_colour_kernels_odd_view_counts_and_degenerate_points(V, kernel, lib_instance):
    """View counts that are not a power of two (V = 1: mean = the view, variance 0), a single point, a device-side count of zero, points outside
    the volume / seen by no view (all pooling weights 0 -> uniform softmax over zero colours), in both matrix-core kernels, vs the oracle."""
    from scene_util import color_t
    if kernel == "tiles":                                       # test-only build variant (libo2345_hip_tiles.so) since round 4
        lib_instance({"O2345_COLOR_KERNEL": "tiles"}, variant="tiles")
    s = small_scene(V=8, HW=40, D=16)
    sc = s["sc"]
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev)
    sel = list(range(V))
    proj, cam_pos = pipeline.camera_terms(t(sc["intrinsics"][sel]), t(sc["w2cs"][sel]))
    cmaps = ops.pack_color_maps(t(s["fmaps"][sel]).contiguous(), t(sc["images"][sel]).contiguous())
    vol_cl = s["dense"][0].permute(1, 2, 3, 0).contiguous().to(dev)
    maskvol = s["mask"][0, 0].reshape(-1).contiguous().to(dev)
    xblob = t(pkg.weights.pack_color_x3_blob(s["color_sd"]))
    rng = np.random.default_rng(V)
    pts = torch.from_numpy(rng.uniform(-0.9, 0.9, (77, 3)).astype(np.float32))
    pts[:5] = torch.tensor([1.5, 0.2, 0.0])                      # outside the volume
    pts[5:8] = torch.tensor([0.0, 0.0, -0.999])                  # inside the cube, far from what the cameras see
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy())
    Kt, w2c = torch.from_numpy(sc["intrinsics"][sel]), torch.from_numpy(sc["w2cs"][sel])
    geo, rf, rd, vm = O.projector(pts, s["dense"][0], s["mask"][0, 0], torch.from_numpy(s["fmaps"][sel]), torch.from_numpy(sc["images"][sel]), w2c, Kt,
                                  (s["W"], s["H"]), query_cam=qcam)
    rgb_ref, nv_ref = O.rendering_network(color_t(s["color_sd"]), geo, rf, rd, vm)
    rgb, nv = ops.color_points(xblob, vol_cl, maskvol, cmaps, proj, cam_pos, pts.to(dev), query_cam=qcam.to(dev), mfma="x3")
    assert torch.equal(nv.cpu().float(), nv_ref)
    assert float((rgb.cpu() - rgb_ref).abs().max()) < 1e-4
    # one point, and the same through an index list with a device-side count
    r1, _ = ops.color_points(xblob, vol_cl, maskvol, cmaps, proj, cam_pos, pts[40:41].to(dev).contiguous(), query_cam=qcam.to(dev), mfma="x3")
    assert float((r1.cpu() - rgb_ref[40:41]).abs().max()) < 1e-4
    idx = torch.tensor([40, 3, 76], dtype=torch.int32, device=dev)
    n2 = torch.tensor([2], dtype=torch.int32, device=dev)
    r2, _ = ops.color_points(xblob, vol_cl, maskvol, cmaps, proj, cam_pos, pts.to(dev), query_cam=qcam.to(dev), index=idx, n_dev=n2, mfma="x3")
    assert float((r2[[40, 3]].cpu() - rgb_ref[[40, 3]]).abs().max()) < 1e-4 and float(r2[76].abs().sum()) == 0      # third list entry not evaluated
    n0 = torch.zeros(1, dtype=torch.int32, device=dev)
    r0, _ = ops.color_points(xblob, vol_cl, maskvol, cmaps, proj, cam_pos, pts.to(dev), query_cam=qcam.to(dev), index=idx, n_dev=n0, mfma="x3")
    assert float(r0.abs().sum()) == 0


# ------------------------------------------------------------------------------------------------- full-size properties
@pytest.fixture(scope="module")
def full():
    torch.manual_seed(0)
    wt = pipeline.SceneWeights(dev, seed=0)
    sc = pkg.synth.make_scene(8, image_seed=3)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    D = 128
    vol = pipeline.build_volume(wt, T(sc["images"]), T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1))
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    return dict(wt=wt, sc=sc, vol=vol, proj=proj, cam_pos=cam_pos, T=T, D=D)


def test_fullsize_cost_volume_properties(full):
    sc, T, D, vol = full["sc"], full["T"], full["D"], full["vol"]
    n = vol["n_voxels"]
    assert 0.45 * D ** 3 < n < 0.65 * D ** 3                                  # SURVEY: 55.5 % occupancy for the 8-view ring
    c = vol["coords"].long()
    lin = (c[:, 0] * D + c[:, 1]) * D + c[:, 2]
    assert bool((lin[1:] > lin[:-1]).all())                                    # x-major order, no duplicates
    assert torch.equal(vol["row_of_voxel"][lin], torch.arange(n, dtype=torch.int32, device=dev))
    assert bool((vol["cnt"][lin] > 1).all()) and int((vol["cnt"] > 1).sum()) == n
    # (no sign property for the "variance" half: the reference divides sums over ALL views by the count of VALID views, A.2)
    # aggregation is symmetric in the views: permuting them changes neither the kept set nor (beyond summation order) the rows
    perm = torch.tensor([3, 7, 0, 5, 1, 6, 2, 4], device=dev)
    aff = T(sc["affine_mats"])[perm].contiguous()
    feats = vol["feats_nhwc"][perm].contiguous()
    cnt2, row2, coords2, n2 = ops.costvol_index(aff, 8, 256, 256, (D, D, D), 2.0 / (D - 1), sc["partial_vol_origin"])
    assert n2 == n and torch.equal(coords2, vol["coords"]) and torch.equal(cnt2, vol["cnt"])
    rows2 = ops.costvol_gather(feats, aff, (D, D, D), 2.0 / (D - 1), sc["partial_vol_origin"], cnt2, coords2)
    assert float((rows2 - vol["rows"]).abs().max()) < 1e-4 * float(vol["rows"].abs().max())
    # scatter -> gather round trip
    back = vol["vol_cl"].view(-1, 16)[lin]
    assert torch.equal(back, vol["rows16"])
    assert float(vol["maskvol"].sum()) == n


def test_fullsize_mesh_properties(full):
    R = 256
    verts, tris, rgb, u = pipeline.extract_mesh(full["wt"], full["vol"], full["proj"], full["cam_pos"], R)
    assert verts.shape[0] > 10000 and tris.shape[0] > 20000 and rgb.shape == (verts.shape[0], 3)
    assert float(verts.abs().max()) <= 1.0 and int(tris.max()) == verts.shape[0] - 1 and int(tris.min()) == 0
    # lattice consistency of the fused grid kernel: random lattice nodes re-evaluated as explicit points
    ii = torch.randint(0, R, (50000, 3), device=dev)
    lin = torch.linspace(-1, 1, R).to(dev)            # created on the CPU like extract_fields does (:888-890)
    pts = torch.stack([lin[ii[:, 0]], lin[ii[:, 1]], lin[ii[:, 2]]], -1).contiguous()
    s2 = ops.sdf_mlp(full["wt"].sdf_blob, full["vol"]["vol_cl"], pts, variant=0)["sdf"]
    # the lattice kernel that evaluates the embedding per point: same code, bit-identical coordinates -> bit-identical SDF
    u_pts = ops.sdf_mlp(full["wt"].sdf_blob, full["vol"]["vol_cl"], None, variant=0, grid_R=R, sign=-1.0).get("sdf").view(R, R, R)
    assert torch.equal(-s2, u_pts[ii[:, 0], ii[:, 1], ii[:, 2]])
    # the extraction path reads layer 0 from per-axis tables (fp64-evaluated, csrc/sdf_mlp_x3.hip TAB form): same function to rounding
    d = (-s2 - u[ii[:, 0], ii[:, 1], ii[:, 2]]).abs()
    assert float(d.max()) < 2e-5 * max(1.0, float(s2.abs().max())), float(d.max())
    # topology: every vertex is used, every edge is shared by exactly two triangles unless it lies on the grid boundary
    assert int(torch.unique(tris).numel()) == verts.shape[0]
    t = tris.cpu().numpy()
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), 1)
    _, cnt = np.unique(e[:, 0].astype(np.int64) * verts.shape[0] + e[:, 1], return_counts=True)
    v = verts.cpu().numpy()
    assert set(np.unique(cnt)) <= {1, 2}
    if (cnt == 1).any():                       # open edges only where the surface leaves the (-1,1)^3 box
        uu, inv = np.unique(e[:, 0].astype(np.int64) * verts.shape[0] + e[:, 1], return_inverse=True)
        open_e = e[np.nonzero(cnt[inv] == 1)[0]]
        assert (np.abs(v[open_e.reshape(-1)]).max(1) > 1 - 1e-9).all()
    assert float(rgb.min()) >= -1e-4 and float(rgb.max()) <= 1 + 1e-4          # convex blends of colours in [0,1]
    # vertices sit on sign changes of the sampled field: the true SDF there is small compared with a cell for most of them
    # (the seeded random-weight latent volume is rough, so this is a statistical property, not a bound)
    sv = ops.sdf_mlp(full["wt"].sdf_blob, full["vol"]["vol_cl"], verts.float().contiguous(), variant=0)["sdf"]
    assert float(sv.abs().median()) < 2.0 / (R - 1)


def test_fullsize_render_properties(full):
    sc, T = full["sc"], full["T"]
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    o = pipeline.render(full["wt"], full["vol"], full["proj"], full["cam_pos"], T(ro), T(rd), float(sc["query_near_far"][0]),
                        float(sc["query_near_far"][1]), T(sc["query_c2w"][:3, 3].copy()), want_z=True)
    R = ro.shape[0]
    assert o["weights"].shape == (128, R)
    assert bool(torch.isfinite(o["color"]).all()) and bool(torch.isfinite(o["depth"]).all())
    assert float(o["weights"].min()) >= 0 and float(o["weights_sum"].max()) <= 1 + 1e-4
    assert float(o["color"].min()) >= -1e-4 and float(o["color"].max()) <= 1 + 1e-4
    z = o["z_vals"]
    assert bool((z[1:] >= z[:-1]).all())                                       # merged sample lists are sorted
    assert float(o["weights_sum"].max()) > 0.9                                 # the sphere-like SDF is hit
    assert bool(((o["pm"] == 0) | (o["pm"] == 1)).all()) and bool((o["weights"][o["pm"] == 0] == 0).all())
    assert bool((o["sdf"][o["pm"] == 0] == 100).all())                         # masked points keep the reference's sentinel
    assert float((o["depth"] - (o["mid_z"] * o["weights"]).sum(0)).abs().max()) < 1e-4
    hit = o["weights_sum"] > 0.99
    assert float((o["depth_var"][hit] >= -1e-6).float().mean()) == 1.0


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_fullsize_oracle_parity(full, precision):
    """ORACLE vs HIP at the BENCHMARKED configuration (BASELINE config 2), 256 rays spread over the 512^2 image, every ray and every
    sample accounted for -- no quantiles:
      (1) sampler stage (o2345_ray_upsample) driven with the oracle's OWN per-round inputs: all 4 x 256 x 16 new depths agree;
      (2) everything downstream of the sampler, evaluated by the oracle ON THE HIP PATH'S OWN sample lists: colour / depth / weights of
          ALL rays agree tightly;
      (3) end to end the two differ only through error propagation inside the reference algorithm (4 up-sampling rounds, sigmoid slopes up
          to 512, inverse-CDF sampling): rays whose sample lists coincide agree tightly, and the error distribution of the others matches
          the oracle's OWN sensitivity to fp32-class (2e-6 relative) noise on its SDF values.  The ids of the deviating rays are printed."""
    import fullsize_util as FU
    sc = full["sc"]
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    fu = dict(full, ro=ro, rd=rd)
    ref, sel, _ = FU.oracle_render_sample(fu, 256)
    assert float(ref["weights_sum"].max()) > 0.9
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    # (1) sampler stage, identical inputs
    dz, pdf, width = FU.sampler_stage_check(ops, dev, torch.from_numpy(fu["ro"][sel]), torch.from_numpy(fu["rd"][sel]), near, far,
                                            FU._oracle_args(fu), full["vol"]["maskvol"], full["D"])
    assert dz.shape == (4, len(sel), 16)
    assert float(dz.max()) < 2e-4 and float((dz / width.clamp(min=1e-9)).max()) < 5e-3, (float(dz.max()), float((dz / width.clamp(min=1e-9)).max()))
    # (2) downstream of the sampler on the HIP path's own sample lists: ALL rays
    out = FU.gpu_render_sample(fu, sel, precision)
    core = FU.oracle_core_on(fu, sel, out["z_vals"])
    assert float((out["color"] - core["color_fine"]).abs().max()) < 3e-5
    assert float((out["depth"] - core["depth"][:, 0]).abs().max()) < 1e-5
    assert float((out["weights"] - core["weights"]).abs().max()) < 1e-5
    assert float((out["weights_sum"] - core["weights_sum"][:, 0]).abs().max()) < 1e-5
    assert bool((out["color_mask"].bool() == core["color_fine_mask"][:, 0]).all())
    # (3) end to end
    cerr = (out["color"] - ref["color_fine"]).abs().max(1).values
    zerr = (out["z_vals"] - ref["z_vals"]).abs().max(1).values
    same = zerr < 1e-6
    assert float(cerr[same].max()) < 3e-5 if same.any() else True
    dev_rays = torch.nonzero(cerr > 1e-4)[:, 0]
    print(f"[{precision}] {len(dev_rays)} of {len(sel)} rays deviate by > 1e-4 in colour; ray ids {sel[dev_rays.numpy()].tolist()}; "
          f"their sample lists differ by {zerr[dev_rays].min().item() if len(dev_rays) else 0:.2e} .. {zerr[dev_rays].max().item() if len(dev_rays) else 0:.2e}")
    assert bool((zerr[dev_rays] > 1e-6).all())                       # every deviation is attributable to the sample lists (2 is tight for all rays)
    assert float(zerr.max()) <= 1.001 * (far - near) / 63            # never more than one coarse section (a relocated sample shifts the sorted list by one)
    ce, _ = FU.oracle_self_sensitivity(fu, sel, ref)
    q = lambda t, x: float(torch.quantile(t.flatten(), x))
    for x in (0.5, 0.9, 0.99):
        assert q(cerr, x) <= 4 * q(ce, x) + 1e-5, (x, q(cerr, x), q(ce, x))
    assert float(cerr.mean()) <= 4 * float(ce.mean()) + 1e-5


def test_fullsize_numerical_forms_agree(full):
    """BASELINE-size scene, 512^2 rays: the default split-f16 form and the exact fp32 MFMA form of the network kernels render
    the same image.  Per-sample lists differ for the few rays whose importance samples fall into empty bins (the reference's
    ill-conditioned sample_pdf, see test_gpu_parity.test_render), so the statement is about the rendered quantities."""
    sc, T = full["sc"], full["T"]
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    args = (full["vol"], full["proj"], full["cam_pos"], T(ro), T(rd), float(sc["query_near_far"][0]), float(sc["query_near_far"][1]),
            T(sc["query_c2w"][:3, 3].copy()))
    outs = {}
    for prec in ("f16x3", "fp32"):
        wt = full["wt"]
        old = (wt.sdf_precision, wt.color_precision)
        wt.sdf_precision = wt.color_precision = prec
        try:
            outs[prec] = pipeline.render(wt, *args)
        finally:
            wt.sdf_precision, wt.color_precision = old
    a, b = outs["f16x3"], outs["fp32"]
    for k, tol_mean in (("color", 2e-4), ("depth", 2e-4), ("weights_sum", 2e-4)):
        d = (a[k].float() - b[k].float()).abs()
        print(k, "mean", float(d.mean()), "within 1e-4/1e-3/1e-2/1e-1:", [round(float((d < t).float().mean()), 5) for t in (1e-4, 1e-3, 1e-2, 1e-1)], "max", float(d.max()))
        assert float(d.mean()) < tol_mean, (k, float(d.mean()))
        # measured: 94-96 % of the rays within 1e-4, 98.7-99.1 % within 1e-3, 99.8 % within 1e-2 (random-weight scene, inv_s = 7.4: broad
        # weight distributions, i.e. many importance samples in near-empty bins where sample_pdf amplifies 1e-6 SDF differences)
        assert float((d < 1e-3).float().mean()) >= 0.98 and float((d < 1e-2).float().mean()) >= 0.995, k
    assert float((a["color_mask"] != b["color_mask"]).float().mean()) < 1e-3
    # the SDF of the final pass (same points wherever the sample lists coincide): fp32-class agreement on the bulk
    same = (a["mid_z"] - b["mid_z"]).abs() < 1e-6
    ds = (a["sdf"] - b["sdf"]).abs()[same & (a["pm"] > 0)]
    # 29 M points: the maximum approaches the worst case of the split form (144 same-signed terms x 2^-22), the mean is fp32 rounding noise
    assert float(ds.max()) < 2e-4 and float(ds.mean()) < 3e-6 and float(same.float().mean()) > 0.9


# ------------------------------------------------------------------------------- oracle parity of the VOLUME BUILD at the benchmarked size
@pytest.fixture(scope="module")
def oracle_volume(full):
    """The ORACLE's own get_conditional_volume from the IMAGES at BASELINE config-2 size (8 x 256^2 views, 128^3): FeatureNet -> fused pyramid ->
    compress layer -> back-projection + aggregation over 2.1 M voxels -> sparse CNN on 1.17 M voxels -> dense scatter (oracle/recon.py,
    ~30 GFLOP on the host: tens of seconds)."""
    import fullsize_util as FU
    return FU.oracle_volume(full["wt"], full["sc"], full["D"])


def test_fullsize_volume_vs_oracle(full, oracle_volume):
    """HIP volume build vs the oracle at the BENCHMARKED size (sparse_sdf_network.py:286-400, tsparse/modules.py:259-304): kept-voxel set and
    visible-view counts exact, every intermediate tensor and the dense latent volume within fp32 tolerances."""
    import fullsize_util as FU
    r = FU.volume_vs_oracle(full["vol"], oracle_volume, full["D"])
    print("full-size volume build vs oracle (max abs error / max|oracle|):", r)
    assert r["kept_set_exact"] and r["view_counts_exact"] and r["mask_exact"] and r["kept_voxels"] > 1_000_000, r     # 1,166,970 kept voxels, same order
    assert r["fused_pyramid"] < 5e-5 and r["compressed_maps"] < 5e-5, r
    assert r["cost_volume_rows"] < 3e-4, r           # var = E[x^2] - mean^2 over the views (sparse_sdf_network.py:221-250): cancellation amplifies the 3e-6 of the maps
    assert r["sparse_cnn_rows"] < 1e-4 and r["dense_volume"] < 1e-4, r


def _oracle_args_from(ov_dense, ov_mask, fmaps, wt, sc):
    """Oracle argument dict for tests/render_check.py from CPU tensors (dense [1,16,D,D,D], mask [1,1,D,D,D], fmaps [V,56,H,W])."""
    H, Wd = sc["images"].shape[2:]
    return dict(volume=ov_dense[0].contiguous(), maskvol=ov_mask[0, 0].contiguous(), W={k: torch.from_numpy(np.asarray(v)) for k, v in wt.sdfW.items()},
                RW={k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in wt.color_sd.items()}, feat_maps=fmaps.contiguous(),
                color_maps=torch.from_numpy(sc["images"]), w2cs=torch.from_numpy(sc["w2cs"]), K=torch.from_numpy(sc["intrinsics"]),
                img_wh=(Wd, H), query_c2w=torch.from_numpy(sc["query_c2w"]))


def _scene_from(wt, vol_cl, maskvol, cmaps, proj, cam_pos):
    return dict(sdf_blob=wt.sdf_blob, color_mfma_blob=wt.color_mblob, color_x3_blob=wt.color_xblob, vol_cl=vol_cl,
                maskvol=maskvol, cmaps=cmaps, proj=proj, cam_pos=cam_pos)


def _spread(n_total, n):
    return torch.from_numpy(np.unique(np.linspace(0, n_total - 1, n).astype(np.int64)))


def test_fullsize_render_on_the_oracle_volume(full, oracle_volume):
    """The three-clause render contract at BASELINE config 2 with the ORACLE's volume, mask and feature maps on BOTH sides (the other
    full-size test hands the oracle HIP's volume): the render comparison then isolates the renderer, and together with
    test_fullsize_volume_vs_oracle the whole path images -> colours is oracle-checked at the benchmarked size."""
    import render_check as RC
    sc, wt, ov, T = full["sc"], full["wt"], oracle_volume, full["T"]
    a = _oracle_args_from(ov["dense"], ov["mask"], ov["fmaps"], wt, sc)
    cmaps = ops.pack_color_maps(ov["fmaps"].to(dev).contiguous(), T(sc["images"]))
    scene = _scene_from(wt, ov["dense"][0].permute(1, 2, 3, 0).contiguous().to(dev), ov["mask"].reshape(-1).contiguous().to(dev), cmaps, full["proj"], full["cam_pos"])
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    sel = _spread(ro.shape[0], 192)
    res = RC.three_clause(ops, dev, scene, a, torch.from_numpy(ro)[sel], torch.from_numpy(rd)[sel], float(sc["query_near_far"][0]),
                          float(sc["query_near_far"][1]), 0.2, 1.0, 1.0, "f16x3", chunk=64, label="C2 on the oracle's volume")
    assert res["rays_hitting_surface"] > 20, res


# ------------------------------------------------------------------------------- the other timed configurations, with parity
@pytest.mark.parametrize("variance", [0.2, 0.6])
def test_reference_configuration_render_parity(variance):
    """The REFERENCE's own configuration (confs/one2345_lod0_val_demo.conf: 32 source views, 96^3 volume, one 256 x 256 val image) -- the
    `ref_config` block of bench.py: three-clause render parity on rays spread over the image (k_color_mfma<32,true>, 32-view cost volume, sparse
    CNN on ~760 k voxels), at the initial variance and at a trained model's (inv_s = 403)."""
    import render_check as RC
    V, D = 32, 96
    wt = pipeline.SceneWeights(dev, seed=0, variance=variance)
    sc = pkg.synth.make_scene(V, image_seed=4)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    vol = pipeline.build_volume(wt, T(sc["images"]), T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1))
    assert 0.80 * D ** 3 < vol["n_voxels"] < 0.92 * D ** 3                     # SURVEY: 85.9 % of the voxels valid with 32 views
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    fm = vol["cmaps"][..., 3:59].permute(0, 3, 1, 2).contiguous().cpu()
    a = _oracle_args_from(vol["vol_cl"].permute(3, 0, 1, 2)[None].cpu(), vol["maskvol"].view(1, 1, D, D, D).cpu(), fm, wt, sc)
    scene = _scene_from(wt, vol["vol_cl"], vol["maskvol"], vol["cmaps"], proj, cam_pos)
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256)
    sel = _spread(ro.shape[0], 96)
    res = RC.three_clause(ops, dev, scene, a, torch.from_numpy(ro)[sel], torch.from_numpy(rd)[sel], float(sc["query_near_far"][0]),
                          float(sc["query_near_far"][1]), variance, 1.0, 1.0, "f16x3", chunk=32, label="REF V=32 96^3")
    assert res["rays_hitting_surface"] > 10, res
    # vertex colours of the export path at this view count (k_color_mfma<32> with normals): a sample of surface points vs the oracle
    verts, tris, rgb, u = pipeline.extract_mesh(wt, vol, proj, cam_pos, 128)
    pick = _spread(verts.shape[0], 600).to(dev)
    p = verts[pick].float().contiguous().cpu()
    g = O.sdf_grad(p, a["volume"], a["W"])
    geo, rf, rdf, vm = O.projector(p, a["volume"], a["maskvol"], a["feat_maps"], a["color_maps"], a["w2cs"], a["K"], a["img_wh"],
                                   normals=torch.nn.functional.normalize(g, p=2, dim=-1, eps=1e-6))
    want, _ = O.rendering_network(a["RW"], geo, rf, rdf, vm)
    assert float((rgb[pick].cpu() - want).abs().max()) < 2e-4


def test_config5_render_parity():
    """BASELINE config 5 shape (256^3 volume -- 9.4 M kept voxels through the sparse CNN --, 1024^2 virtual camera): the `config5` block of
    bench.py.  64 rays spread over the image under the three-clause contract (the oracle reads HIP's 256^3 volume: its own volume build at this
    size would take minutes), plus the 256^3 kept-voxel set against the oracle's projection test (exact)."""
    import render_check as RC
    V, D = 8, 256
    wt = pipeline.SceneWeights(dev, seed=0)
    sc = pkg.synth.make_scene(V, image_seed=6)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    vol = pipeline.build_volume(wt, T(sc["images"]), T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1))
    assert 0.45 * D ** 3 < vol["n_voxels"] < 0.65 * D ** 3
    # kept set: the oracle's projection / visibility rule on the full 256^3 lattice (integer result: exact)
    lat = O.voxel_lattice([D, D, D])
    cnt = torch.cat([O.project(lat[s:s + (1 << 20)] * (2.0 / (D - 1)) + torch.from_numpy(sc["partial_vol_origin"])[None],
                               torch.from_numpy(sc["affine_mats"]), 256, 256)[3].sum(1) for s in range(0, lat.shape[0], 1 << 20)])
    assert torch.equal(vol["cnt"].cpu().long().view(-1), cnt.long())
    assert torch.equal(vol["coords"][:, :3].cpu().long(), lat[cnt > 1].long())
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    fm = vol["cmaps"][..., 3:59].permute(0, 3, 1, 2).contiguous().cpu()
    a = _oracle_args_from(vol["vol_cl"].permute(3, 0, 1, 2)[None].cpu(), vol["maskvol"].view(1, 1, D, D, D).cpu(), fm, wt, sc)
    scene = _scene_from(wt, vol["vol_cl"], vol["maskvol"], vol["cmaps"], proj, cam_pos)
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=4)
    sel = _spread(ro.shape[0], 64)
    res = RC.three_clause(ops, dev, scene, a, torch.from_numpy(ro)[sel], torch.from_numpy(rd)[sel], float(sc["query_near_far"][0]),
                          float(sc["query_near_far"][1]), 0.2, 1.0, 1.0, "f16x3", chunk=32, label="C5 256^3")
    assert res["rays_hitting_surface"] > 5, res
    del vol, scene
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------- end-to-end mesh agreement
def test_fullsize_mesh_vs_oracle_field(full):
    """north_star: "mesh topology exactly ... at matched mesh IoU".  Marching cubes is exact GIVEN u (tests/test_gpu_parity.py); this test closes
    the loop over the field: on a 64^3 sub-block of the 256^3 extraction lattice that the surface crosses, HIP's u (tabulated-layer-0 lattice
    kernel, f16x3) vs the oracle's extract_fields (sparse_neus_renderer.py:881-905) -- sign disagreements counted and bounded by the field
    tolerance (a node may only flip where |u| is inside the error band), volumetric IoU of the two inside-sets >= 0.999, and the two meshes of the
    sub-block (HIP marching cubes on HIP's u, the oracle's marching cubes on the oracle's u): without a flipped node every cell has the same
    8-corner sign pattern, hence identical triangles, vertex for vertex."""
    import fullsize_util as FU
    R = 256
    verts, tris, rgb, u = pipeline.extract_mesh(full["wt"], full["vol"], full["proj"], full["cam_pos"], R)
    r = FU.mesh_field_vs_oracle(ops, full["wt"], full["vol"], u, R, 64)
    print("mesh field vs oracle:", r)
    assert r["sign_changes_in_block"] > 1000 and r["inside_nodes_union"] > 1000
    assert r["field_err_max"] < 2e-5 * max(1.0, r["field_scale"])
    assert r["mesh_sign_flips"] == 0 or r["abs_u_at_flips_max"] <= r["field_err_max"]            # a node flips only inside the error band
    assert r["iou"] >= 0.999
    if r["mesh_sign_flips"] == 0:
        assert r["hip_mesh"] == r["oracle_mesh"] and r["triangles_identical"]
        assert r["vertex_shift_max_mean_cells"][0] < 0.5 and r["vertex_shift_max_mean_cells"][1] < 1e-3, r
    else:
        n = r["mesh_sign_flips"]
        assert abs(r["hip_mesh"][1] - r["oracle_mesh"][1]) <= 16 * n and abs(r["hip_mesh"][0] - r["oracle_mesh"][0]) <= 12 * n
