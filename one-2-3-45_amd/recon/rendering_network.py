"""GeneralRenderingNetwork (mirror of models/rendering_network.py:26-129): same parameters / state-dict keys; the forward
pass of projector outputs produced by our Projector runs fused in csrc/color.hip."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config, ops, weights


class DeferredColour:
    """What our Projector returns in place of the four big tensors: everything csrc/color.hip needs.  It is passed through
    the unchanged trainer code (trainer_generic.py:1330-1361) straight into GeneralRenderingNetwork.forward."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, *a, **k):
        return self

    def detach(self):
        return self

    def materialise(self):
        """The reference Projector's own four tensors for these points (models/projector.py:96-425): geometry_feat [R,S,16], rgb_feat [V,R,S,59],
        ray_diff [V,R,S,4], mask [V,R,S] -- for a caller that feeds a rendering network other than ours."""
        R, S = self.shape
        geo, rf, rd, m = ops.project_features(self.vol_cl, self.maskvol, self.cmaps, self.proj, self.cam_pos, self.pts,
                                              query_cam=getattr(self, "query_cam", None), normals=getattr(self, "normals", None))
        V = rf.shape[0]
        return geo.view(R, S, 16), rf.view(V, R, S, 59), rd.view(V, R, S, 4), m.view(V, R, S)


class GeneralRenderingNetwork(nn.Module):
    def __init__(self, in_geometry_feat_ch=8, in_rendering_feat_ch=56, anti_alias_pooling=True):
        super().__init__()
        if not (in_geometry_feat_ch == 16 and in_rendering_feat_ch == 56 and anti_alias_pooling):
            raise NotImplementedError("o2345 GeneralRenderingNetwork: only the released configuration (16 / 56 / pooling) is built")
        self.in_geometry_feat_ch, self.in_rendering_feat_ch, self.anti_alias_pooling = 16, 56, True
        self.s = nn.Parameter(torch.tensor(0.2), requires_grad=True)
        act = nn.ELU(inplace=True)
        self.ray_dir_fc = nn.Sequential(nn.Linear(4, 16), act, nn.Linear(16, 59), act)
        self.base_fc = nn.Sequential(nn.Linear(193, 64), act, nn.Linear(64, 32), act)
        self.vis_fc = nn.Sequential(nn.Linear(32, 32), act, nn.Linear(32, 33), act)
        self.vis_fc2 = nn.Sequential(nn.Linear(32, 32), act, nn.Linear(32, 1), nn.Sigmoid())
        self.rgb_fc = nn.Sequential(nn.Linear(37, 16), act, nn.Linear(16, 8), act, nn.Linear(8, 1))
        for seq in (self.base_fc, self.vis_fc2, self.vis_fc, self.rgb_fc):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.kaiming_normal_(m.weight.data)
                    nn.init.zeros_(m.bias.data)
        self._blob, self._key = None, None

    def blob(self):
        ps = [p for _, p in sorted(self.named_parameters())]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._blob is None or key != self._key:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._blob = torch.from_numpy(weights.pack_color_blob(sd)).to(ps[0].device)
            self._mblob = torch.from_numpy(weights.pack_color_mfma_blob(sd)).to(ps[0].device)
            self._xblob = torch.from_numpy(weights.pack_color_x3_blob(sd)).to(ps[0].device)
            self._key = key
        return self._blob

    def mfma_blob(self):
        self.blob()
        return self._mblob

    def x3_blob(self):
        self.blob()
        return self._xblob

    @torch.no_grad()
    def forward(self, geometry_feat, rgb_feat=None, ray_diff=None, mask=None):
        """Either the reference's four tensors or a DeferredColour produced by our Projector.  Returns
        (rgb [n_rays, n_samples, 3], valid_mask [n_rays]) like rendering_network.py:122-129."""
        if isinstance(geometry_feat, DeferredColour):
            d = geometry_feat
            mf = True                          # matrix-core kernels for every view count (k_color_pts beyond 32 views)
            if mf and config.color_precision() == "f16x3":
                blob, mode = self.x3_blob(), "x3"
            else:
                blob, mode = (self.mfma_blob() if mf else self.blob()), mf
            rgb, nv = ops.color_points(blob, d.vol_cl, d.maskvol, d.cmaps, d.proj, d.cam_pos, d.pts,
                                       query_cam=d.query_cam, normals=d.normals, want_nviews=True, mfma=mode)
            R, S = d.shape
            valid = ((nv.view(R, S) >= 2).float().sum(1) > 8)
            return rgb.view(R, S, 3), valid
        # the reference's own call form (rendering_network.py:75-83): geometry_feat [R,S,16], rgb_feat [V,R,S,59], ray_diff [V,R,S,4], mask [V,R,S]
        # -- any Projector's materialised tensors.  Same network kernel, inputs read instead of gathered (slower than the fused path, same results)
        if rgb_feat is None or ray_diff is None or mask is None:
            raise ValueError("o2345 GeneralRenderingNetwork.forward: pass a DeferredColour or the reference's four tensors")
        R, S = geometry_feat.shape[:2]
        V = rgb_feat.shape[0]
        x3 = config.color_precision() == "f16x3"
        rgb, nv = ops.color_from_features(self.x3_blob() if x3 else self.mfma_blob(), geometry_feat.reshape(R * S, -1), rgb_feat.reshape(V, R * S, -1),
                                          ray_diff.reshape(V, R * S, 4), mask.reshape(V, R * S).float(), x3=x3)
        valid = ((nv.view(R, S) >= 2).float().sum(1) > 8)
        return rgb.view(R, S, 3), valid
