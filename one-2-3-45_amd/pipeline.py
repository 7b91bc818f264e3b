"""Per-scene reconstruction on one MI355X: the hot path of export_mesh_step / val_step (trainer_generic.py:359-622,
827-979) as a sequence of HIP calls with no host synchronisation except the three size read-backs
(kept-voxel count, coarse-level sizes, mesh size)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import config, ops, weights
from .costreg import CostRegNet
from .featurenet import ConvBnReLU, FeatureNet, fused_pyramid, set_precision


def _load_checked(module, sd, what):
    """load_state_dict that only tolerates absent BatchNorm running buffers (the reference runs in training mode and this back end
    never reads them): a renamed / missing parameter must not silently leave seeded stand-in weights in place."""
    res = module.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if not ("running_" in k or "num_batches_tracked" in k)]
    if missing or res.unexpected_keys:
        raise KeyError(f"{what}: checkpoint does not match (missing {missing}, unexpected {list(res.unexpected_keys)})")


class SceneWeights:
    """All network parameters of one lod-0 model on the device (seeded stand-ins unless state dicts are given)."""

    def __init__(self, device, seed=0, sdf=None, color_sd=None, costreg_sd=None, variance=0.2, sdf_precision=None, color_precision=None):
        self.device = device
        sdf_precision = config.sdf_precision(sdf_precision)
        color_precision = config.color_precision(color_precision)
        self.color_precision = color_precision    # "f16x3" (default) | "fp32": see config.py
        self.sdf_precision = sdf_precision        # "f16x3" (default) | "fp32": see config.py
        with torch.random.fork_rng(devices=[]):         # seeded stand-in initialisation must not reset the caller's global RNG
            torch.manual_seed(seed)
            self.featurenet = set_precision(FeatureNet().to(device), color_precision)      # 2-D convolutions follow the same mode as the
            self.compress = set_precision(ConvBnReLU(56, 16).to(device), color_precision)   # sparse ones: strict fp32 means fp32 everywhere
        self.sdfW = sdf or weights.init_sdf_weights(seed)
        self.color_sd = color_sd or weights.init_color_state_dict(seed)
        self.costreg_sd = costreg_sd or weights.init_costreg_state_dict(seed)
        t = lambda a: torch.from_numpy(a).to(device)
        self.sdf_blob = t(weights.pack_sdf_blob(self.sdfW))
        self.color_mblob = t(weights.pack_color_mfma_blob(self.color_sd))
        self.color_xblob = t(weights.pack_color_x3_blob(self.color_sd))
        self.costreg = CostRegNet(self.costreg_sd, device, precision=color_precision)      # sparse convolutions follow the same mode
        self._grid_tabs = {}
        self.variance = float(variance)
        self.inv_s = float(np.clip(np.exp(10.0 * variance), 1e-6, 1e6))

    def grid_tables(self, resolution):
        """Layer 0 of the SDF network tabulated for the extraction lattice of this resolution (built once per resolution; f16x3 mode only)."""
        if self.sdf_precision != "f16x3":
            return None
        R = int(resolution)
        if R not in self._grid_tabs:
            axes, bias = weights.sdf_grid_tables(self.sdfW, R)
            dev = self.sdf_blob.device
            self._grid_tabs[R] = ops.sdf_grid_tables(torch.from_numpy(axes).to(dev), torch.from_numpy(bias).to(dev))
        return self._grid_tabs[R]

    @classmethod
    def from_state_dicts(cls, device, sdf_network_sd, rendering_network_sd, variance, featurenet_sd=None, sdf_precision=None, color_precision=None):
        """Build from the reference checkpoint's per-network state dicts (exp_runner_generic_blender_val.py:485-512:
        keys ``sdf_network_lod0``, ``rendering_network_lod0``, ``variance_network_lod0``, ``pyramid_feature_network``)."""
        t = lambda v: torch.as_tensor(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v))
        sd = {k: t(v) for k, v in sdf_network_sd.items()}
        costreg = {k[len("sparse_costreg_net."):]: v for k, v in sd.items() if k.startswith("sparse_costreg_net.")}
        self = cls(device, seed=0, sdf=weights.sdf_weights_from_state_dict(sd, "sdf_layer."),
                   color_sd={k: t(v).numpy() for k, v in rendering_network_sd.items()}, costreg_sd=costreg, variance=float(variance),
                   sdf_precision=sdf_precision, color_precision=color_precision)
        comp = {k[len("compress_layer."):]: v for k, v in sd.items() if k.startswith("compress_layer.")}
        _load_checked(self.compress, comp, "compress_layer")
        if featurenet_sd is not None:
            _load_checked(self.featurenet, {k: t(v) for k, v in featurenet_sd.items()}, "pyramid_feature_network")
        return self


    @staticmethod
    def _read_checkpoint(path, names, allow_pickle=None):
        import os
        if allow_pickle is None:
            allow_pickle = os.environ.get("O2345_ALLOW_PICKLE", "0") not in ("", "0")
        try:                                   # only float tensors are kept, so the restricted unpickler is enough for a well-formed checkpoint
            ck = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:                 # checkpoints that carry optimizer / numpy scalars need the full unpickler (the reference's own loader uses it)
            if not allow_pickle:
                raise RuntimeError(f"o2345 SceneWeights.from_checkpoint: {path!r} does not load with the restricted unpickler ({type(e).__name__}: {e}); "
                                   "pass allow_pickle=True (or O2345_ALLOW_PICKLE=1) only for a checkpoint you trust -- the full unpickler runs code "
                                   "from the file") from e
            import warnings
            warnings.warn(f"o2345: loading {path!r} with the FULL unpickler (allow_pickle): code in the file is executed")
            ck = torch.load(path, map_location="cpu", weights_only=False)
        return {n: {k: v for k, v in ck[n].items() if torch.is_tensor(v) and v.is_floating_point()} for n in names}

    @classmethod
    def from_checkpoint(cls, device, path, broadcast=False, sdf_precision=None, color_precision=None, allow_pickle=None):
        """Every rank builds its weights from ONE checkpoint file in the reference's format (exp_runner_generic_blender_val.py:514-541: keys
        ``sdf_network_lod0``, ``rendering_network_lod0``, ``variance_network_lod0``, ``pyramid_feature_network``).  Default: each rank reads the file
        (< 4 MB); ``broadcast=True``: rank 0 OF THE PROCESS GROUP reads it and the state dicts reach the other ranks through
        sharding.broadcast_state_dicts (one RCCL broadcast) -- ``path`` may then be None on the other ranks; needs an initialised process group.
        The file is read with torch's restricted unpickler.  A checkpoint that needs the full unpickler (optimizer state with numpy scalars, as the
        reference's own trainer writes) executes arbitrary code from the file when loaded: that is opt-in -- ``allow_pickle=True`` or O2345_ALLOW_PICKLE=1
        -- for files you trust, never a silent fallback."""
        import torch.distributed as dist
        from . import sharding
        names = ("sdf_network_lod0", "rendering_network_lod0", "variance_network_lod0", "pyramid_feature_network")
        if broadcast and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("SceneWeights.from_checkpoint(broadcast=True) needs an initialised torch.distributed process group (sharding.init)")
        state = None
        if not broadcast or dist.get_rank() == 0:
            try:
                state = cls._read_checkpoint(path, names, allow_pickle)
            except Exception as e:                 # with broadcast=True the other ranks are about to enter the collective: hand them the failure instead of a hang
                if not broadcast:
                    raise
                state = e
        if broadcast:
            state = sharding.broadcast_state_dicts(state, device)
        return cls.from_state_dicts(device, state["sdf_network_lod0"], state["rendering_network_lod0"], state["variance_network_lod0"]["variance"],
                                    featurenet_sd=state["pyramid_feature_network"], sdf_precision=sdf_precision, color_precision=color_precision)


@torch.no_grad()
def build_volume(wt, imgs, affine_mats, origin, D, voxel_size, fmaps=None):
    """imgs [V,3,H,W] cuda -> scene dict (dense latent volume, occupancy, colour maps...) -- steps (a),(b) of 3.2.
    ``fmaps`` [V,56,H,W] may be supplied to skip FeatureNet."""
    V, _, H, W = imgs.shape
    cmaps = None
    if fmaps is None:
        # the fused pyramid is written once, as the channel-last colour map [V,H,W,64] (rgb | 56 features | pad); the compress layer reads its
        # features from there (channel offset 3): the channel-first [V,56,H,W] tensor of the reference API is never materialised on this path
        _, cmaps = fused_pyramid(wt.featurenet, imgs, want_cmaps=True, want_nchw=False)
        feats_nhwc = wt.compress.forward_nhwc(cmaps, nhwc_offset=3)            # Conv3x3 56->16 + batch stats, then ABN + channel-last re-layout (HIP)
    else:
        feats_nhwc = wt.compress.forward_nhwc(fmaps)
    cnt, row, coords, n = ops.costvol_index(affine_mats, V, H, W, (D, D, D), voxel_size, origin)
    rows = ops.costvol_gather(feats_nhwc, affine_mats, (D, D, D), voxel_size, origin, cnt, coords)
    rows16 = wt.costreg.forward(rows, coords, row, (D, D, D))
    vol_cl, vol_cf, mask = ops.scatter_dense(rows16, row, (D, D, D), want_cf=False)
    if cmaps is None:
        cmaps = ops.pack_color_maps(fmaps, imgs.contiguous())
    return dict(vol_cl=vol_cl, maskvol=mask.view(-1), cmaps=cmaps, n_voxels=n, rows16=rows16, fmaps=fmaps, rows=rows, coords=coords,
                row_of_voxel=row, cnt=cnt, feats_nhwc=feats_nhwc)


def camera_terms(intrinsics, w2cs):
    """proj = intrinsics @ w2cs[:, :3, :] (render_utils.py:106), cam_pos = inverse(w2cs)[:, :3, 3]: one HIP launch (ops.camera_terms), no BLAS / solver library."""
    return ops.camera_terms(intrinsics, w2cs)


@torch.no_grad()
def render(wt, vol, proj, cam_pos, rays_o, rays_d, near, far, query_cam, n_samples=64, n_importance=64, want_z=False, t_rand=None):
    """One call renders all rays (no 512-ray chunks).  ``t_rand`` [R, n_samples]: the reference's perturb > 0 jitter (drawn by the caller).
    Note: the reference's per-512-ray-chunk quirks (cat_z_vals skipped when <= 1 new point of the CHUNK is valid; "first 100 points"
    when a chunk has no valid point) apply per CALL here -- identical when called per chunk, as the drop-in mirror does (or per segment of one
    call: ops.render_rays(segment_rays=...)).  The colour network skips occupied samples whose compositing weight is below config.WEIGHT_CULL
    (2^-24: a ray's colour moves by <= 7.6e-6, nothing else changes; O2345_WEIGHT_CULL=0 = every occupied sample, like the reference)."""
    scene = dict(sdf_blob=wt.sdf_blob, vol_cl=vol["vol_cl"], maskvol=vol["maskvol"],
                 cmaps=vol["cmaps"], proj=proj, cam_pos=cam_pos, color_mfma_blob=wt.color_mblob,
                 sdf_precision=wt.sdf_precision, color_precision=wt.color_precision,
                 color_x3_blob=wt.color_xblob)
    return ops.render_rays(scene, rays_o, rays_d, near, far, n_samples, n_importance, wt.inv_s, 1.0, 1.0, query_cam, want_z, t_rand=t_rand)


@torch.no_grad()
def render_scene_split(wt, imgs, affine_mats, origin, D, voxel_size, proj, cam_pos, rays_o, rays_d, near, far, query_cam, src=0,
                       keys=("color", "depth", "weights_sum", "color_mask"), **render_kw):
    """ONE image rendered by all ranks of the process group (intra-scene split; single-scene latency instead of scene throughput).
    ``imgs`` [V,3,H,W]: the scene's source images on rank ``src`` (None elsewhere) -- one broadcast; every rank builds the volume itself and renders a
    contiguous block of the rays (sharding.ray_block); one all-gather returns ``keys`` of pipeline.render for ALL rays on every rank.  A render call's
    results do not depend on how the rays are batched (tests/test_gpu_parity.py::test_chunked_render_equals_one_call), so the result equals the
    one-GPU call bit for bit wherever the reference's per-call rules cannot fire (a block without any occupied sample).  Without a process group this
    is build_volume + render."""
    from . import sharding as sh
    dev = rays_o.device
    imgs = sh.broadcast_tensor(imgs, src=src, device=dev)
    vol = build_volume(wt, imgs, affine_mats, origin, D, voxel_size)
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if torch.distributed.is_initialized() else (0, 1)
    lo, hi, per = sh.ray_block(rays_o.shape[0], rank, world)
    R_all = rays_o.shape[0]
    per_ray = lambda a, b: {k: (v[a:b].contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == R_all else v)     # t_rand [R, n_samples] (perturb > 0)
                            for k, v in render_kw.items()}                                                                  # follows its rays
    if hi > lo:
        o = render(wt, vol, proj, cam_pos, rays_o[lo:hi].contiguous(), rays_d[lo:hi].contiguous(), near, far, query_cam, **per_ray(lo, hi))
        block = {k: o[k] for k in keys}
    else:                                                # more ranks than 64-ray blocks: this rank contributes nothing
        o = render(wt, vol, proj, cam_pos, rays_o[:1].contiguous(), rays_d[:1].contiguous(), near, far, query_cam, **per_ray(0, 1))
        block = {k: o[k][:0] for k in keys}
    return sh.gather_ray_blocks(block, rays_o.shape[0], per, device=dev), vol


@torch.no_grad()
def extract_mesh(wt, vol, proj, cam_pos, resolution, return_index_verts=False):
    """extract_fields + marching cubes + vertex colouring (trainer_generic.py:1309-1363), all on the device."""
    prec = wt.sdf_precision
    u = ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], None, variant=0, grid_R=resolution, sign=-1.0, precision=prec, grid_tables=wt.grid_tables(resolution))["sdf"]
    u = u.view(resolution, resolution, resolution)
    verts_idx, tris = ops.marching_cubes(u, 0.0)
    verts = (verts_idx / (resolution - 1.0) * 2.0 - 1.0)                      # sparse_neus_renderer.py:936
    pts = verts.to(torch.float32).contiguous()
    if return_index_verts:
        verts = verts_idx
    if pts.shape[0] == 0:
        return verts, tris, torch.zeros(0, 3, device=pts.device), u
    g = ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, precision=prec)["grad"]
    x3 = wt.color_precision == "f16x3"
    rgb, _ = ops.color_points(wt.color_xblob if x3 else wt.color_mblob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], proj, cam_pos, pts, normals=g,
                              want_nviews=False, mfma="x3" if x3 else True)
    return verts, tris, rgb, u


@torch.no_grad()
def export_mesh_ply(path, wt, vol, proj, cam_pos, resolution, scale_mat=None, trans_mat=None):
    """validate_colored_mesh end to end (trainer_generic.py:1309-1382): SDF grid, marching cubes, vertex colours, frame transforms,
    uint8 colours and the binary PLY -- records packed on the device (csrc/mesh_pack.hip), one D2H copy, one file write.
    Returns (n_vertices, n_triangles)."""
    from . import mesh_io
    verts_idx, tris, rgb, _ = extract_mesh(wt, vol, proj, cam_pos, resolution, return_index_verts=True)
    return mesh_io.export_mesh(path, verts_idx, tris, resolution, scale_mat=scale_mat, trans_mat=trans_mat,
                               vertex_colors=rgb if verts_idx.shape[0] else None)


@torch.no_grad()
def reconstruct_folder(root_dir, name, wt, out_ply, D=96, resolution=256, render_val_image=False):
    """run.py's reconstruction stage without the reference tree: Zero123-style folder (dataset.SceneFolder) -> coloured mesh (binary PLY in
    the original frame), optionally the val image of the target view.  Returns dict(vertices, triangles, kept_voxels[, color, depth])."""
    from . import dataset
    s = dataset.SceneFolder(root_dir, "export_mesh", specific_dataset_name=name)[0]
    dev = wt.device
    T = lambda t: t.to(dev).contiguous().float()
    vol = build_volume(wt, T(s["images"]), T(s["affine_mats"]), s["partial_vol_origin"].numpy(), D, 2.0 / (D - 1))
    proj, cam_pos = camera_terms(T(s["intrinsics"]), T(s["w2cs"]))
    nv, nt = export_mesh_ply(out_ply, wt, vol, proj, cam_pos, resolution, scale_mat=s["scale_mat"], trans_mat=s["trans_mat"])
    out = {"vertices": nv, "triangles": nt, "kept_voxels": int(vol["n_voxels"])}
    if render_val_image:
        r = render(wt, vol, proj, cam_pos, T(s["rays"]["rays_o"]), T(s["rays"]["rays_v"]), float(s["query_near_far"][0]), float(s["query_near_far"][1]),
                   T(s["query_c2w"][:3, 3]))
        H, W = int(s["img_wh"][1]), int(s["img_wh"][0])
        out["color"], out["depth"] = r["color"].view(H, W, 3), r["depth"].view(H, W)
    return out
