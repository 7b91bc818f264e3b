"""Build container only: the reference's UNCHANGED GenericTrainer (models/trainer_generic.py) driven through the drop-in import hook.

``dropin.install()`` serves this package's mirrors / shims under the reference's module names; ``models.trainer_generic``, ``utils.*``
and ``loss.*`` are the reference's own files.  There is no GPU here, so the ops layer is replaced by oracle-backed CPU stand-ins
(tests/fake_ops.py, test infrastructure): what this test pins is the TRAINER'S OWN control flow on top of the mirrors --
``export_mesh_step`` (trainer_generic.py:827-979 -> validate_colored_mesh :1309-1382 -> trimesh export) and ``val_step``
(:359-622, the reference's default perturb = 1.0 path, 512-ray chunks, save_visualization, validate_mesh) -- i.e. dict plumbing,
tensor / numpy round trips, returned keys and shapes, files written.  The arithmetic behind the same mirror calls is pinned on the
GPU by tests/test_gpu_mirror.py against reference-generated golden vectors."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import ref_import as RI

pytestmark = [pytest.mark.reference, pytest.mark.skipif(not RI.available(), reason="/root/reference not present")]
pkg = importlib.import_module("one-2-3-45_amd")


class Conf(dict):
    def _get(self, k, default=None):
        if k in self:
            return self[k]
        if default is None:
            raise KeyError(k)
        return default

    def get_int(self, k, default=None):
        return int(self._get(k, default))

    def get_float(self, k, default=None):
        return float(self._get(k, default))

    def get_bool(self, k, default=None):
        return bool(self._get(k, default))


@pytest.fixture()
def ref_trainer(monkeypatch, tmp_path):
    """-> (GenericTrainer class of the reference, mirror module namespace); every sys.modules / meta_path change is undone afterwards."""
    import fake_ops
    dropin = importlib.import_module("one-2-3-45_amd.dropin")
    mine = ("torchsparse", "inplace_abn", "mcubes", "trimesh", "models", "utils", "loss", "cv2", "torchvision", "icecream", "tsparse")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in mine}
    written = []

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    # third-party packages of the reference's utility modules that are not installed here and not on the path under test
    mod("cv2", COLORMAP_JET=2, applyColorMap=lambda x, cmap: np.repeat(np.asarray(x)[..., None], 3, -1),
        imwrite=lambda path, img: written.append((path, np.asarray(img).shape)) or True)
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils")
    tv.transforms = mod("torchvision.transforms")
    mod("icecream", ic=lambda *a, **k: None)
    old_path = list(sys.path)
    sys.path.insert(0, RI.REF)
    old_dwb = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        dropin.install()
        fake_ops.install(monkeypatch)
        from models.trainer_generic import GenericTrainer
        import models.featurenet as mf
        import models.rendering_network as mr
        import models.sparse_sdf_network as ms
        assert GenericTrainer.__module__ == "models.trainer_generic" and "/root/reference" in sys.modules["models.trainer_generic"].__file__
        assert ms.SparseSdfNetwork.__module__.startswith("one-2-3-45_amd") and mf.FeatureNet.__module__.startswith("one-2-3-45_amd")
        fields = importlib.import_module("one-2-3-45_amd.recon.fields")
        yield GenericTrainer, types.SimpleNamespace(sdf=ms, ren=mr, feat=mf, fields=fields, written=written, tmp=str(tmp_path))
    finally:
        sys.meta_path[:] = [f for f in sys.meta_path if type(f).__name__ != "_AliasFinder"]
        for k in list(sys.modules):
            if k.split(".")[0] in mine:
                del sys.modules[k]
        sys.modules.update(saved)
        sys.path[:] = old_path
        sys.dont_write_bytecode = old_dwb


def _build(GenericTrainer, M, D, seed=3):
    torch.manual_seed(seed)
    conf = Conf({"model.num_lods": 1, "train.if_fix_lod0_networks": True, "train.sdf_igr_weight": 0.1, "train.val_mesh_freq": 1,
                 "general.base_exp_dir": M.tmp, "model.h_patch_size": 3})
    feat = M.feat.FeatureNet()
    sdf = M.sdf.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=2.0 / (D - 1), vol_dims=[D, D, D], hidden_dim=128, cost_type="variance_mean",
                                 d_pyramid_feature_compress=16, regnet_d_out=16, num_sdf_layers=4, multires=6)
    # the geometric initialisation zeroes the latent / PE columns; perturb them so that the volume matters (seeded)
    g = torch.Generator().manual_seed(seed)
    L = sdf.sdf_layer
    L.lin1.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
    L.lin2.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
    ren = M.ren.GeneralRenderingNetwork(in_geometry_feat_ch=16, in_rendering_feat_ch=56, anti_alias_pooling=True)
    var = M.fields.SingleVarianceNetwork(0.2)
    tr = GenericTrainer(None, feat, None, sdf, None, var, None, ren, None, n_samples_lod0=64, n_importance_lod0=64, n_samples_lod1=64,
                        n_importance_lod1=64, n_outside=0, perturb=1.0, alpha_type="div", conf=conf, timestamp="", base_exp_dir=M.tmp)
    return tr


def _sample(V, HW):
    """The dict the reference's dataset hands to the trainer (data/One2345_eval_new_data.py:300-377), batch dimension 1."""
    sc = pkg.synth.make_scene(V, hw=(HW, HW), image_seed=5)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None]
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], HW, HW)
    ys, xs = np.meshgrid(np.linspace(0, HW - 1, HW), np.linspace(0, HW - 1, HW), indexing="ij")
    uv = np.stack([2 * xs / (HW - 1) - 1, 2 * ys / (HW - 1) - 1], -1).reshape(-1, 2).astype(np.float32)
    return {"batch_idx": torch.tensor([0]), "meta": ["synthetic_scene"], "img_wh": torch.tensor([[HW, HW]]), "partial_vol_origin": T(sc["partial_vol_origin"]),
            "query_near_far": T(sc["query_near_far"]), "rays": {"rays_o": T(ro), "rays_v": T(rd), "rays_ndc_uv": T(uv)},
            "images": T(sc["images"]), "intrinsics": T(sc["intrinsics"]), "w2cs": T(sc["w2cs"]), "c2ws": T(sc["c2ws"]),
            "affine_mats": T(sc["affine_mats"]), "scale_mat": T(sc["scale_mat"]), "trans_mat": T(sc["trans_mat"]),
            "query_c2w": T(sc["query_c2w"]), "query_w2c": T(sc["query_w2c"]), "query_image": T(sc["images"][0]),
            "scale_factor": torch.tensor([1.0])}, sc


def test_export_mesh_step_of_the_unchanged_trainer(ref_trainer):
    GenericTrainer, M = ref_trainer
    mesh_io = importlib.import_module("one-2-3-45_amd.mesh_io")
    D, HW, R = 14, 24, 20
    tr = _build(GenericTrainer, M, D)
    sample, sc = _sample(4, HW)
    tr.export_mesh_step(sample, iter_step=0, chunk_size=512, resolution=R)
    path = os.path.join(M.tmp, "mesh.ply")
    assert os.path.exists(path), "validate_colored_mesh must have written <base_exp_dir>/mesh.ply"
    v, f, c = mesh_io.read_ply(path)
    assert v.shape[0] > 0 and f.shape[0] > 0 and c is not None and c.shape == (v.shape[0], 4) and (c[:, 3] == 255).all()
    assert f.min() >= 0 and f.max() < v.shape[0]
    # the vertices are in the original (un-normalised) frame: inverse transforms bring them back into the unit cube
    tm, sm = sc["trans_mat"].astype(np.float64), sc["scale_mat"].astype(np.float64)
    w = (np.linalg.inv(tm) @ np.concatenate([v.astype(np.float64), np.ones((len(v), 1))], 1).T).T[:, :3]
    n = (w - sm[:3, 3][None]) / sm[0, 0]
    assert np.abs(n).max() <= 1.0 + 1e-4
    # the same path through the mirror API directly gives the same mesh (the trainer added nothing but plumbing)
    with torch.no_grad():
        fm = tr.obtain_pyramid_feature_maps(sample["images"][0], lod=0)
        cv = tr.sdf_network_lod0.get_conditional_volume(feature_maps=fm[None], partial_vol_origin=sample["partial_vol_origin"],
                                                        proj_mats=sample["affine_mats"], sizeH=HW, sizeW=HW, lod=0)
        verts, tris, u = tr.sdf_renderer_lod0.extract_geometry(tr.sdf_network_lod0, torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), resolution=R,
                                                               threshold=0, device="cpu", conditional_volume=cv["dense_volume_scale0"], lod=0)
    assert verts.shape[0] == v.shape[0] and np.array_equal(tris, f) and np.abs(verts - n).max() < 1e-4


def test_val_step_of_the_unchanged_trainer_default_perturb(ref_trainer):
    """--mode val: val_step passes NO perturb_overwrite, the conf's perturb = 1.0 applies (confs/one2345_lod0_val_demo.conf:127)."""
    GenericTrainer, M = ref_trainer
    mesh_io = importlib.import_module("one-2-3-45_amd.mesh_io")
    D, HW = 14, 24
    tr = _build(GenericTrainer, M, D)
    sample, sc = _sample(4, HW)
    seen = []
    ren = tr.sdf_renderer_lod0
    orig = ren.render

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append((a[0].shape[0], k.get("perturb_overwrite", -1), {kk: (tuple(v.shape) if torch.is_tensor(v) else v) for kk, v in out.items()}))
        return out
    ren.render = spy
    # val_step calls validate_mesh with its default 360^3 grid (46.6 M SDF evaluations: minutes on the CPU stand-ins): same call, 20^3 grid
    vm = tr.validate_mesh
    tr.validate_mesh = lambda *a, **k: vm(*a, **dict(k, resolution=20))
    torch.manual_seed(11)
    tr.val_step(sample, background_rgb=None, alpha_inter_ratio_lod0=1.0, iter_step=0, chunk_size=512, save_vis=True)
    # 24 x 24 = 576 rays in 512-ray chunks, default (stochastic) sampling
    assert [s[0] for s in seen] == [512, 64] and all(s[1] == -1 for s in seen)
    keys = seen[0][2]
    for k, shp in (("color_fine", (512, 3)), ("depth", (512, 1)), ("weights", (512, 128)), ("gradients", (512, 128, 3)), ("inside_sphere", (512, 128)),
                   ("weights_sum", (512, 1)), ("sdf", (512 * 128, 1)), ("color_fine_mask", (512, 1)), ("cdf_fine", (512, 128))):
        assert keys[k] == shp, (k, keys[k])
    # the reference's return dictionary, key for key (sparse_neus_renderer.py:594-635)
    assert set(keys) >= {"depth", "color_fine", "color_fine_mask", "color_outside", "color_outside_mask", "color_mlp", "color_mlp_mask", "variance",
                         "cdf_fine", "depth_variance", "weights_sum", "weights_max", "alpha_sum", "alpha_mean", "gradients", "weights",
                         "gradient_error_fine", "inside_sphere", "sdf", "sdf_random", "blended_color_patch", "blended_color_patch_mask",
                         "weights_sum_fg"}
    # save_visualization wrote the three images, validate_mesh the uncoloured mesh
    names = sorted(os.path.relpath(p, M.tmp) for p, _ in M.written)
    assert names == ["depths_val_lod0/00000000_synthetic_scene.png", "normals_val_lod0/00000000_synthetic_scene.png",
                     "synthesized_color_val_lod0/00000000_synthetic_scene.png"]
    shapes = {os.path.relpath(p, M.tmp).split("/")[0]: s for p, s in M.written}
    assert shapes["synthesized_color_val_lod0"] == (2 * HW, HW, 3) and shapes["normals_val_lod0"] == (HW, HW, 3)
    ply = os.path.join(M.tmp, "meshes_val_bg", "mesh_00000000_synthetic_scene_lod0.ply")
    assert os.path.exists(ply)
    v, f, c = mesh_io.read_ply(ply)
    assert v.shape[0] > 0 and f.shape[0] > 0 and c is None
    # the stochastic path is seeded by torch's host generator exactly like the reference: same seed -> same image, other seed -> another
    imgs = []
    for seed in (11, 11, 12):
        torch.manual_seed(seed)
        imgs.append(orig(sample["rays"]["rays_o"][0][:64], sample["rays"]["rays_v"][0][:64], sample["query_near_far"][0, :1],
                         sample["query_near_far"][0, 1:], tr.sdf_network_lod0, tr.rendering_network_lod0, lod=0, alpha_inter_ratio=1.0,
                         conditional_volume=tr.sdf_network_lod0.get_conditional_volume(
                             feature_maps=tr.obtain_pyramid_feature_maps(sample["images"][0])[None], partial_vol_origin=sample["partial_vol_origin"],
                             proj_mats=sample["affine_mats"], sizeH=HW, sizeW=HW, lod=0)["dense_volume_scale0"],
                         conditional_valid_mask_volume=torch.ones(1, 1, D, D, D), feature_maps=tr.obtain_pyramid_feature_maps(sample["images"][0]),
                         color_maps=sample["images"][0], w2cs=sample["w2cs"][0], intrinsics=sample["intrinsics"][0], img_wh=[HW, HW],
                         query_c2w=sample["query_c2w"], if_render_with_grad=False)["color_fine"])
    assert torch.equal(imgs[0], imgs[1]) and not torch.equal(imgs[0], imgs[2])


def test_whole_image_mode_under_the_unchanged_trainer_loop(ref_trainer):
    """The reference's OWN val_step loop (trainer_generic.py:503-524: `rays_o.reshape(-1, 3).split(chunk_size)`, one render() per chunk) drives the mirror's
    whole-image mode: the chunks are views of one tensor, the first call renders every segment (O2345RenderIO.segment_rays; here through the CPU stand-in,
    which evaluates the segments one oracle call each), the second call is a slice.  Pinned here: the HOST LOGIC -- view detection on the trainer's real
    tensors, the interleaved host random stream (t_rand, pts_random per chunk), slicing, cache release -- by comparing every returned entry of every chunk
    and the final generator state with the plain per-chunk calls (O2345_WHOLE_IMAGE=0) under the same seed.  The kernels' side of the same contract
    (per-segment rules) is tests/test_gpu_segments.py."""
    GenericTrainer, M = ref_trainer
    D, HW = 14, 24
    tr = _build(GenericTrainer, M, D)
    sample, sc = _sample(4, HW)
    ren = tr.sdf_renderer_lod0
    orig = ren.render
    vm = tr.validate_mesh
    tr.validate_mesh = lambda *a, **k: vm(*a, **dict(k, resolution=12))
    runs = {}
    for mode in (False, True):
        ren.whole_image, ren._abandoned, ren._image = mode, 0, None
        outs, served = [], []

        def spy(*a, **k):
            had = ren._image is not None
            out = orig(*a, **k)
            outs.append({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in out.items()})
            served.append(had and ren._image is None or (had and ren._image is not None and ren._image["next"] > 1))
            return out
        ren.render = spy
        torch.manual_seed(5)
        tr.val_step(sample, background_rgb=None, alpha_inter_ratio_lod0=1.0, iter_step=0, chunk_size=512, save_vis=True)
        runs[mode] = (outs, torch.get_rng_state(), served)
        ren.render = orig
    plain, fused = runs[False], runs[True]
    assert len(plain[0]) == len(fused[0]) == 2 and plain[0][1]["depth"].shape[0] == 64
    assert fused[2] == [False, True] and plain[2] == [False, False], "the second chunk of the image was served from the first call's fused render"
    assert torch.equal(plain[1], fused[1]), "host generator state after the image"
    for k, (a, b) in enumerate(zip(plain[0], fused[0])):
        assert set(a) == set(b) and len(a) == 23
        for key in a:
            if a[key] is None:
                assert b[key] is None
            else:
                assert a[key].shape == b[key].shape and torch.equal(a[key], b[key]), (k, key)
    assert ren._image is None


def test_reference_costregnet_on_the_torchsparse_shim(ref_trainer):
    """INTEGRATION.md's "shims only" level with the REFERENCE's own module code: tsparse/modules.py's SparseCostRegNet (:259-304, built from
    its BasicSparse{Conv,Deconv}olutionBlock) imports `torchsparse` = our shim (dropin hook) and runs through the shim's Conv3d / BatchNorm /
    ReLU / SparseTensor.__add__ (kernel-map reuse by the transposed convolutions, coordinate bookkeeping per stride) -- against
    oracle.sparse_costreg.  CPU stand-ins replace the three ops calls underneath (sparse_downsample, sparse_conv3d, bn_act_rows)."""
    _, M = ref_trainer
    import torchsparse
    from tsparse.modules import SparseCostRegNet
    from oracle import recon as O
    from scene_util import costreg_oracle_weights, small_scene
    assert torchsparse.__name__.startswith("one-2-3-45_amd") or "o2345" in torchsparse.__version__
    assert "/root/reference" in sys.modules["tsparse.modules"].__file__
    s = small_scene()
    net = SparseCostRegNet(d_in=32, d_out=16)
    sd = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in s["costreg_sd"].items()}
    miss = net.load_state_dict(sd, strict=False)
    assert not miss.unexpected_keys and all("running_" in k or "num_batches" in k for k in miss.missing_keys), miss
    x = torchsparse.SparseTensor(s["vol"].clone(), s["coords"].clone())
    got = net(x)
    want, extra = O.sparse_costreg(s["vol"], s["coords"], costreg_oracle_weights(s["costreg_sd"]))
    assert got.shape == want.shape and float((got - want).abs().max()) < 2e-4 * max(1.0, float(want.abs().max()))
    for lv, key in zip(extra["levels"][1:], (2, 4, 8)):
        assert torch.equal(x.cmaps[key][0][:, :3].long(), lv.xyz)
