"""SparseNeuSRenderer + Projector on the HIP back end (mirror of models/sparse_neus_renderer.py:22-937 and
models/projector.py:11-425; general rendering, lod 0)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .rendering_network import DeferredColour
from .sparse_sdf_network import channel_last


def _scene_maps(feature_maps, color_maps, w2cs, intrinsics):
    cm = getattr(feature_maps, "_o2345_cmaps", None)
    if cm is None:
        cm = ops.pack_color_maps(feature_maps.detach().contiguous().float(), color_maps.detach().contiguous().float())
        try:
            feature_maps._o2345_cmaps = cm
        except Exception:
            pass
    proj = torch.matmul(intrinsics, w2cs[:, :3, :]).contiguous().float()
    cam_pos = torch.inverse(w2cs)[:, :3, 3].contiguous().float()
    return cm, proj, cam_pos


class Projector:
    """compute / compute_view_independent return (DeferredColour, None, None, None, None, None): the reference's callers pass the
    first four entries straight to GeneralRenderingNetwork.forward (sparse_neus_renderer.py:311-338, trainer_generic.py:1330-1352).
    With ``materialise = True`` they return the reference's own four tensors instead (for a rendering network that is not ours)."""

    materialise = False

    def _result(self, h):
        if self.materialise:
            return h.materialise() + (None, None)
        return h, None, None, None, None, None

    def _handle(self, pts, geometryVolume, geometryVolumeMask, rendering_feature_maps, color_maps, w2cs, intrinsics, **kw):
        if pts.dim() == 2:
            pts = pts[None]
        R, S, _ = pts.shape
        vol = geometryVolume[None] if geometryVolume.dim() == 4 else geometryVolume
        cm, proj, cam_pos = _scene_maps(rendering_feature_maps, color_maps, w2cs, intrinsics)
        return DeferredColour(pts=pts.reshape(-1, 3).contiguous().float(), shape=(R, S), vol_cl=channel_last(vol),
                              maskvol=geometryVolumeMask.reshape(-1).contiguous().float(), cmaps=cm, proj=proj, cam_pos=cam_pos, **kw)

    def compute(self, pts, geometryVolume=None, geometryVolumeMask=None, vol_dims=None, partial_vol_origin=None, vol_size=None,
                rendering_feature_maps=None, color_maps=None, w2cs=None, intrinsics=None, img_wh=None, query_img_idx=0, query_c2w=None,
                pred_depth_maps=None, pred_depth_masks=None):
        if query_c2w is None:
            raise NotImplementedError("o2345 Projector.compute: pass query_c2w (the runner always does)")
        h = self._handle(pts, geometryVolume, geometryVolumeMask, rendering_feature_maps, color_maps, w2cs, intrinsics,
                         query_cam=query_c2w.reshape(-1, 4, 4)[0, :3, 3].contiguous().float(), normals=None)
        return self._result(h)

    def compute_view_independent(self, pts, geometryVolume=None, geometryVolumeMask=None, sdf_network=None, lod=0, vol_dims=None,
                                 partial_vol_origin=None, vol_size=None, rendering_feature_maps=None, color_maps=None, w2cs=None,
                                 target_candidate_w2cs=None, intrinsics=None, img_wh=None, query_img_idx=0, query_c2w=None,
                                 pred_depth_maps=None, pred_depth_masks=None):
        p = pts if pts.dim() == 2 else pts.reshape(-1, 3)
        grad = sdf_network.gradient(p.contiguous().float(), geometryVolume[None] if geometryVolume.dim() == 4 else geometryVolume, lod).squeeze(1)
        h = self._handle(pts, geometryVolume, geometryVolumeMask, rendering_feature_maps, color_maps, w2cs, intrinsics,
                         query_cam=None, normals=grad.contiguous())
        return self._result(h)


class SparseNeuSRenderer(nn.Module):
    def __init__(self, rendering_network_outside, sdf_network, variance_network, rendering_network, n_samples, n_importance, n_outside,
                 perturb, alpha_type="div", conf=None):
        super().__init__()
        if n_outside != 0 or alpha_type != "div":
            raise NotImplementedError("o2345 SparseNeuSRenderer: n_outside = 0 and alpha_type = 'div' (the released configuration)")
        self.conf = conf
        self.base_exp_dir = conf["general.base_exp_dir"] if conf is not None else None
        self.rendering_network_outside, self.sdf_network = rendering_network_outside, sdf_network
        self.variance_network, self.rendering_network = variance_network, rendering_network
        self.n_samples, self.n_importance, self.n_outside, self.perturb, self.alpha_type = n_samples, n_importance, n_outside, perturb, alpha_type
        self.rendering_projector = Projector()
        self.if_fitted_rendering = False

    @torch.no_grad()
    def get_pts_mask_for_conditional_volume(self, pts, mask_volume):
        D = mask_volume.shape[-1]
        idx = torch.round(((pts + 1) * D - 1) / 2)
        ok = ((idx >= 0) & (idx <= D - 1)).all(1)
        ci = idx.clamp(0, D - 1).long()
        m = mask_volume.reshape(D, D, D)[ci[:, 0], ci[:, 1], ci[:, 2]]
        return torch.where(ok, m, torch.zeros_like(m))[:, None]

    @torch.no_grad()
    def get_valid_sparse_coords_by_sdf(self, sdf_volume, coords_volume, mask_volume, feature_volume, threshold=0.02, maximum_pts=110000):
        """lod-0 -> lod-1 pruning (:822-879): voxels with |sdf| < threshold, dilated by a 7^3 box, AND the valid mask; the
        threshold is lowered by 0.002 while more than maximum_pts voxels remain.  Returns (coords [N,4] float (b,x,y,z),
        features [N,C]) in x-major order.  If the count is STILL above maximum_pts the reference drops voxels with an unseeded
        np.random.choice; here the drop is a seeded permutation (documented deviation: same distribution, reproducible)."""
        C, D = feature_volume.shape[0], feature_volume.shape[1]
        sv = sdf_volume.reshape(-1).contiguous().float()
        mv = mask_volume.reshape(-1).contiguous().float()
        thr = float(threshold)
        flag = ops.prune_dilate(sv, mv, D, thr)
        while int(flag.sum()) > maximum_pts and thr > 0.003:
            thr -= 0.002
            flag = ops.prune_dilate(sv, mv, D, thr)
        idx = torch.nonzero(flag)[:, 0]
        if idx.numel() > maximum_pts:
            g = torch.Generator(device="cpu").manual_seed(0)
            keep = torch.sort(torch.randperm(idx.numel(), generator=g)[:maximum_pts]).values.to(idx.device)
            idx = idx[keep]
        coords = coords_volume.reshape(3, -1).t()[idx]
        feat = feature_volume.reshape(C, -1).t()[idx] if getattr(feature_volume, "_o2345_cl", None) is None else feature_volume._o2345_cl.reshape(-1, C)[idx]
        return torch.cat([torch.zeros(idx.numel(), 1, device=coords.device, dtype=coords.dtype), coords], dim=1), feat.contiguous()

    @torch.no_grad()
    def render(self, rays_o, rays_d, near, far, sdf_network, rendering_network, perturb_overwrite=-1, background_rgb=None,
               alpha_inter_ratio=0.0, lod=None, conditional_volume=None, conditional_valid_mask_volume=None, feature_maps=None,
               color_maps=None, w2cs=None, intrinsics=None, img_wh=None, query_c2w=None, if_general_rendering=True,
               if_render_with_grad=True, img_index=None, rays_uv=None, pre_sample=False, bg_ratio=0.0):
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        if pre_sample or bg_ratio > 0 or not if_general_rendering:
            raise NotImplementedError("o2345 render: general rendering without pre_sample / bg_ratio (the released val / export configuration)")
        cm, proj, cam_pos = _scene_maps(feature_maps, color_maps, w2cs, intrinsics)
        R = rays_o.shape[0]
        # stratified jitter exactly as the reference draws it (:506-515): torch.rand(z_vals.shape) on the HOST generator, then moved to
        # the device -> the same numbers as the reference under the same torch.manual_seed; the kernel applies lower + (upper-lower)*t
        t_rand = torch.rand(R, self.n_samples).to(rays_o.device) if perturb > 0 else None
        scene = dict(sdf_blob=sdf_network.sdf_layer.blob(), color_blob=rendering_network.blob(), vol_cl=channel_last(conditional_volume),
                     maskvol=conditional_valid_mask_volume.reshape(-1).contiguous().float(), cmaps=cm, proj=proj, cam_pos=cam_pos,
                     color_mfma_blob=rendering_network.mfma_blob(),
                     color_x3_blob=rendering_network.x3_blob())
        inv_s = float(torch.exp(self.variance_network.variance.detach() * 10.0).clip(1e-6, 1e6))
        nt, ft = torch.as_tensor(near).reshape(-1).float(), torch.as_tensor(far).reshape(-1).float()
        if nt.numel() > 1 and (bool((nt != nt[0]).any()) or bool((ft != ft[0]).any())):
            raise NotImplementedError("o2345 render: one near / far pair per call (the runner passes the query view's [1] tensors); "
                                      "per-ray near / far are not supported")
        nr, fr = float(nt[0]), float(ft[0])
        o = ops.render_rays(scene, rays_o.contiguous().float(), rays_d.contiguous().float(), nr, fr, self.n_samples, self.n_importance,
                            inv_s, float(alpha_inter_ratio), 0.0 if background_rgb is None else float(background_rgb),    # None: nothing is added (:430-431)
                            query_c2w.reshape(-1, 4, 4)[0, :3, 3].contiguous().float(), t_rand=t_rand)
        S = self.n_samples + self.n_importance
        pm = o["pm"].t()
        ge = o["grad_err"].sum(0)
        pts_random = torch.rand([1024, 3], device=rays_o.device) * 2 - 1
        sdf_random = sdf_network.sdf(pts_random, conditional_volume, lod=lod)["sdf_pts_scale%d" % lod]
        color = o["color"]
        return {"depth": o["depth"][:, None], "color_fine": color, "color_fine_mask": o["color_mask"].bool()[:, None], "color_outside": None,
                "color_outside_mask": None, "color_mlp": None, "color_mlp_mask": None, "variance": torch.tensor(1.0 / inv_s, device=rays_o.device),
                "cdf_fine": o["cdf"].t(), "depth_variance": o["depth_var"][:, None], "weights_sum": o["weights_sum"][:, None],
                "weights_max": o["weights_max"][:, None], "alpha_sum": o["alpha_sum"].mean(), "alpha_mean": o["alpha_sum"].sum() / (R * S),
                "gradients": o["grad"].permute(1, 0, 2), "weights": o["weights"].t(), "gradient_error_fine": ge[0] / (ge[1] + 1e-5),
                "inside_sphere": pm, "sdf": o["sdf"].t().reshape(-1, 1), "sdf_random": sdf_random, "blended_color_patch": None,
                "blended_color_patch_mask": None, "weights_sum_fg": o["weights_sum"][:, None]}

    @torch.no_grad()
    def extract_fields(self, bound_min, bound_max, resolution, query_func, device, **kwargs):
        """u = -sdf on linspace(bound_min, bound_max, resolution)^3 (:881-905).  One fused launch instead of 64 chunks + host syncs;
        the lattice is generated in-kernel, which requires the reference's own bounds (-1, 1)."""
        if not (float(torch.as_tensor(bound_min).min()) == -1.0 and float(torch.as_tensor(bound_max).max()) == 1.0):
            raise NotImplementedError("o2345 extract_fields: bounds (-1, 1) only")
        vol = kwargs["conditional_volume"]
        layer = self.sdf_network.sdf_layer
        u = ops.sdf_mlp(layer.blob(), channel_last(vol), None, variant=0, grid_R=resolution, sign=-1.0, grid_tables=layer.grid_tables(resolution))["sdf"]
        return u.view(resolution, resolution, resolution)

    @torch.no_grad()
    def extract_geometry(self, sdf_network, bound_min, bound_max, resolution, threshold, device, occupancy_mask=None, **kwargs):
        """-> (vertices float64 [Nv,3] in world units, triangles int64 [Nt,3], u float32 [R,R,R]) as numpy (:907-937)."""
        u = self.extract_fields(bound_min, bound_max, resolution, None, device, **kwargs)
        if occupancy_mask is not None:
            e = torch.nn.functional.interpolate((1 - occupancy_mask)[None, None].float(), [resolution] * 3, mode="nearest")[0, 0] > 0
            u = torch.where(e.to(u.device), torch.full_like(u, -100.0), u)
        v, t = ops.marching_cubes(u.contiguous(), float(threshold))
        bmin = torch.as_tensor(bound_min).double().cpu().numpy()
        bmax = torch.as_tensor(bound_max).double().cpu().numpy()
        verts = v.cpu().numpy() / (resolution - 1.0) * (bmax - bmin)[None, :] + bmin[None, :]
        return verts, t.cpu().numpy(), u.cpu().numpy()
