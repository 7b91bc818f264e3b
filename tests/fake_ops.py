"""CPU stand-ins for ``one-2-3-45_amd.ops`` backed by the oracle -- TEST INFRASTRUCTURE ONLY.

Purpose: run the reference's UNCHANGED ``GenericTrainer`` (models/trainer_generic.py) through ``dropin.install()`` in the build
container, where ``/root/reference`` exists but no GPU does, so that the trainer's own control flow (dict plumbing, ``.to()``,
chunking, ``validate_colored_mesh``'s host round trips, file outputs) is exercised against the mirror modules and shims.  Every
stand-in takes exactly the arguments of the ops function it replaces and returns tensors of the same shape / dtype / layout; the
arithmetic is the oracle's (oracle/recon.py, oracle/mc.c).  Nothing here is imported by the product."""
import importlib

import numpy as np
import torch
import torch.nn.functional as F

from oracle import mc as omc
from oracle import recon as O

pkg = importlib.import_module("one-2-3-45_amd")
_SDF, _COL, _KEEP = {}, {}, []


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32)


def _origin(o):
    return torch.as_tensor(np.asarray(o.detach().cpu() if torch.is_tensor(o) else o, np.float32)).reshape(-1)[:3]


# ------------------------------------------------------------------------------------------------ cost volume
def costvol_index(proj, V, H, W, dims, voxel_size, origin, min_views=1):
    lat = O.voxel_lattice([int(d) for d in dims])
    _, _, _, m = O.project(lat * voxel_size + _origin(origin)[None], proj, H, W)
    cnt = m.sum(1)
    idx = torch.nonzero(cnt > min_views)[:, 0]
    coords = torch.cat([lat[idx].to(torch.int32), torch.zeros(len(idx), 1, dtype=torch.int32)], 1)
    row = torch.full((lat.shape[0],), -1, dtype=torch.int32)
    row[idx] = torch.arange(len(idx), dtype=torch.int32)
    return cnt.to(torch.uint8), row, coords, int(len(idx))


def _rows(feats_nhwc, proj, voxel_size, origin, coords):
    feats = feats_nhwc.permute(0, 3, 1, 2)
    H, W = feats.shape[2:]
    gx, gy, _, m = O.project(coords[:, :3].float() * voxel_size + _origin(origin)[None], proj, H, W)
    f = O.bilinear_zeros(feats, gx, gy)
    c = (1.0 / (m.sum(1).float() + 1e-5))[:, None]
    s1, s2 = f.sum(1), (f * f).sum(1)
    return torch.cat([s2 * c - (s1 * c) ** 2, s1 * c], 1)


def costvol_gather(feats_nhwc, proj, dims, voxel_size, origin, cnt, coords):
    return _rows(feats_nhwc, proj, voxel_size, origin, coords)


def visible_count_list(proj, H, W, voxel_size, origin, coords):
    _, _, _, m = O.project(coords[:, :3].float() * voxel_size + _origin(origin)[None], proj, H, W)
    return m.sum(1).to(torch.uint8)


def costvol_gather_list(feats_nhwc, proj, voxel_size, origin, cnt_row, coords):
    return _rows(feats_nhwc, proj, voxel_size, origin, coords)


def build_index_grid(coords, ts, cells):
    nx, ny, nz = (int(c) for c in cells)
    g = torch.full((nx * ny * nz,), -1, dtype=torch.int32)
    c = coords.long() // ts
    g[(c[:, 0] * ny + c[:, 1]) * nz + c[:, 2]] = torch.arange(coords.shape[0], dtype=torch.int32)
    return g


def scatter_dense(rows, row_of_voxel, dims, want_cf=True):
    dx, dy, dz = (int(d) for d in dims)
    C = rows.shape[1]
    cl = torch.zeros(dx * dy * dz, C)
    v = row_of_voxel >= 0
    cl[v] = rows[row_of_voxel[v].long()]
    cl = cl.view(dx, dy, dz, C)
    cf = cl.permute(3, 0, 1, 2).contiguous()[None] if want_cf else None
    return cl, cf, v.float().view(1, 1, dx, dy, dz)


def _ss_act(x, ss, slope):
    C = x.shape[1]
    t = x * ss[:C].view(1, -1, 1, 1) + ss[C:].view(1, -1, 1, 1)
    return torch.where(t >= 0, t, t * slope)


def conv_x3(precision=None):
    return False


def conv2d_pack(weight, precision=None):
    return weight


def conv2d(x, weight, bias=None, stride=1, in_scale_shift=None, slope=0.01, bn=None, packed=None, precision=None, nhwc_offset=None):
    """ops.conv2d: raw convolution output + this layer's InPlaceABN (scale | shift) from the batch statistics."""
    if nhwc_offset is not None:
        x = x[..., nhwc_offset:nhwc_offset + weight.shape[1]].permute(0, 3, 1, 2)
    if in_scale_shift is not None:
        x = _ss_act(x, in_scale_shift, slope)
    out = F.conv2d(x, weight, bias, stride, weight.shape[-1] // 2)
    ss = None
    if bn is not None:
        gamma, beta, eps, abs_gamma = bn
        mean, var = out.mean((0, 2, 3)), out.var((0, 2, 3), unbiased=False)
        g = gamma.abs() + eps if abs_gamma else gamma
        scale = g / torch.sqrt(var + eps)
        ss = torch.cat([scale, beta - mean * scale])
    return out, ss


def scale_shift_act(x, scale_shift, slope=0.01, want_nchw=True, want_nhwc=False):
    y = _ss_act(x, scale_shift, slope)
    return (y if want_nchw else None), (y.permute(0, 2, 3, 1).contiguous() if want_nhwc else None)


def fpn_level(fine, coarse, weight, bias, fine_scale_shift=None, slope=0.01):
    if fine_scale_shift is not None:
        fine = _ss_act(fine, fine_scale_shift, slope)
    return F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True) + F.conv2d(fine, weight.reshape(32, -1, 1, 1), bias)


def pyramid_pack(f2, s1, s0, rgb, want_nchw=True):
    fm = torch.cat([F.interpolate(f2, scale_factor=4, mode="bilinear", align_corners=True), F.interpolate(s1, scale_factor=2, mode="bilinear", align_corners=True), s0], 1)
    return (fm if want_nchw else None), pack_color_maps(fm, rgb)


def sparse_downsample(coords, ts, fine_cells):
    """-> (index grid of the coarse level, coarse coords int32 [n,4], n, cells per axis) -- oracle.downsample_coords on the lattice ops uses."""
    nc = tuple((int(c) + 1) // 2 + 1 for c in fine_cells)
    if coords.shape[0] == 0:
        return torch.full((nc[0] * nc[1] * nc[2],), -1, dtype=torch.int32), torch.zeros(0, 4, dtype=torch.int32), 0, nc
    lc = O.downsample_coords(O.SparseLevel(coords[:, :3], ts))
    cc = torch.cat([lc.xyz.to(torch.int32), torch.zeros(len(lc.xyz), 1, dtype=torch.int32)], 1)
    return build_index_grid(cc, 2 * ts, nc), cc, int(len(lc.xyz)), nc


def sparse_conv3d(mode, x, in_grid, in_cells, out_coords, ts_out, kernel):
    """mode 0 same stride, 1 strided (output stride = 2 x input stride), 2 transposed (output stride = input stride / 2); gather form over the
    input level's index grid, kernel offsets with x fastest (oracle.kernel_offsets)."""
    ts_in = {0: ts_out, 1: ts_out // 2, 2: ts_out * 2}[int(mode)]
    step = ts_in if mode != 2 else ts_out                       # spacing of the kernel offsets = the FINER of the two strides
    nx, ny, nz = (int(c) for c in in_cells)
    q = out_coords[:, :3].long()
    out = torch.zeros(q.shape[0], kernel.shape[2], dtype=x.dtype)
    for k, off in enumerate(O.kernel_offsets(step)):
        src = q + off[None] if mode != 2 else q - off[None]      # transposed: out[p] += x[q'] W[k] for q' + off_k = p
        ok = (src % ts_in == 0).all(1) & (src >= 0).all(1)
        c = src // ts_in
        ok &= (c[:, 0] < nx) & (c[:, 1] < ny) & (c[:, 2] < nz)
        lin = ((c[:, 0] * ny + c[:, 1]) * nz + c[:, 2]).clamp(0, nx * ny * nz - 1)
        row = torch.where(ok, in_grid[lin].long(), torch.full_like(lin, -1))
        v = row >= 0
        if v.any():
            out[v] += x[row[v]] @ kernel[k]
    return out


def bn_act_rows(x, gamma, beta, eps=1e-5, slope=0.0, abs_gamma=False, skip=None, want_stats=False):
    mu, var = x.mean(0), x.var(0, unbiased=False)
    g = (gamma.abs() + eps) if abs_gamma else gamma
    y = (x - mu) / torch.sqrt(var + eps) * g + beta
    y = torch.where(y >= 0, y, y * slope)
    if skip is not None:
        y = y + skip
    return (y, torch.stack([mu, var])) if want_stats else y


def costreg_forward(self, feat, coords, grid0, dims):
    """CostRegNet.forward: the same rows in the same order from the oracle's sparse U-Net."""
    w = {n: (K.cpu(), g.cpu(), b.cpu()) for n, (K, g, b) in self.p.items()}
    out, extra = O.sparse_costreg(feat, coords, w)
    self.level_sizes = tuple(len(lv.xyz) for lv in extra["levels"])
    return out


def abn_forward(self, x, want_nhwc=False):
    """featurenet.InPlaceABN.forward."""
    y = O.abn_train(x, self.weight.detach(), self.bias.detach(), self.eps, self.slope, self.abs_gamma)
    return (y, y.permute(0, 2, 3, 1).contiguous()) if want_nhwc else y


# ------------------------------------------------------------------------------------------------ networks
def sdf_grid_tables(tab_axes, bias_lane_order):
    return tab_axes, bias_lane_order


def sdf_mlp(blob, vol_cl, pts=None, variant=0, grid_R=0, sign=1.0, index=None, n_dev=None, want_lat=False, out=None, lat_in=None,
            precision=None, grid_tables=None):
    W = _SDF[blob.data_ptr()]
    volume = vol_cl.permute(3, 0, 1, 2)
    if pts is None:
        lin = torch.linspace(-1, 1, int(grid_R))
        pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    P = pts.shape[0]
    res = out or {}
    res.setdefault("sdf", torch.zeros(P))
    if variant == 1:
        res.setdefault("feat", torch.zeros(P, 128))
    if variant == 2:
        res.setdefault("grad", torch.zeros(P, 3))
    if want_lat:
        res.setdefault("lat", torch.zeros(P, 16))
    sel = torch.arange(P) if index is None else index.long()
    if sel.numel() == 0:
        return res
    p = pts[sel]
    if lat_in is not None:
        lat = lat_in[sel]
        y = O.sdf_mlp(p, lat, W)
    else:
        y, lat = O.sdf(p, volume, W)
    res["sdf"][sel] = sign * y[:, 0]
    if variant == 1:
        res["feat"][sel] = y
    if variant == 2:
        res["grad"][sel] = O.sdf_grad(p, volume, W)
    if want_lat:
        res["lat"][sel] = lat
    return res


def pack_color_maps(feat_nchw, color_nchw):
    V, _, H, W = feat_nchw.shape
    return torch.cat([color_nchw, feat_nchw, torch.zeros(V, 5, H, W)], 1).permute(0, 2, 3, 1).contiguous()


def _maps(cmaps):
    return cmaps[..., 3:59].permute(0, 3, 1, 2).contiguous(), cmaps[..., :3].permute(0, 3, 1, 2).contiguous()


def color_points(blob, vol_cl, maskvol, cmaps, proj, cam_pos, pts, query_cam=None, normals=None, index=None, n_dev=None,
                 want_nviews=True, mfma="x3", stats=None):
    RW = _COL[blob.data_ptr()]
    D = vol_cl.shape[0]
    fm, cm = _maps(cmaps)
    P = pts.shape[0]
    rgb = torch.zeros(P, 3)
    nv = torch.zeros(P, dtype=torch.uint8) if want_nviews else None
    sel = torch.arange(P) if index is None else index.long()
    if sel.numel():
        nrm = None if normals is None else F.normalize(normals[sel], p=2, dim=-1, eps=1e-6)
        geo, rf, rd, vm = O.projector(pts[sel], vol_cl.permute(3, 0, 1, 2), maskvol.view(D, D, D), fm, cm, None, None, (cmaps.shape[2], cmaps.shape[1]),
                                      query_cam=query_cam, normals=nrm, proj=proj, cam_pos=cam_pos)
        c, n = O.rendering_network(RW, geo, rf, rd, vm)
        rgb[sel] = c
        if want_nviews:
            nv[sel] = n.to(torch.uint8)
    return rgb, nv


def project_features(vol_cl, maskvol, cmaps, proj, cam_pos, pts, query_cam=None, normals=None):
    D = vol_cl.shape[0]
    fm, cm = _maps(cmaps)
    nrm = None if normals is None else F.normalize(normals, p=2, dim=-1, eps=1e-6)
    geo, rf, rd, vm = O.projector(pts, vol_cl.permute(3, 0, 1, 2), maskvol.view(D, D, D), fm, cm, None, None, (cmaps.shape[2], cmaps.shape[1]),
                                  query_cam=query_cam, normals=nrm, proj=proj, cam_pos=cam_pos)
    return geo, rf.contiguous(), rd.contiguous(), vm.float()


def color_from_features(blob, geometry_feat, rgb_feat, ray_diff, mask, x3=True, want_nviews=True):
    c, n = O.rendering_network(_COL[blob.data_ptr()], geometry_feat, rgb_feat, ray_diff, mask > 0)
    return c, (n.to(torch.uint8) if want_nviews else None)


def render_rays(scene, rays_o, rays_d, near, far, n_samples=64, n_importance=64, inv_s=None, alpha_inter_ratio=1.0, background=1.0,
                query_cam=None, want_z=False, t_rand=None, sample_dist=None, want_scalars=False, color_stats=None, weight_cull=None, segment_rays=0):
    if segment_rays:                       # O2345RenderIO.segment_rays: consecutive render() calls evaluated "together" -- here literally one oracle call per segment
        parts = [render_rays(scene, rays_o[a:a + segment_rays], rays_d[a:a + segment_rays], near, far, n_samples, n_importance, inv_s, alpha_inter_ratio, background,
                             query_cam, want_z, None if t_rand is None else t_rand[a:a + segment_rays], sample_dist, want_scalars, color_stats)
                 for a in range(0, rays_o.shape[0], segment_rays)]
        out = {}
        for k in parts[0]:
            if k == "scalars":
                out[k] = torch.stack([p[k] for p in parts])
            else:
                sample_major = k in ("mid_z", "pm", "sdf", "grad", "weights", "cdf", "z_vals")
                out[k] = torch.cat([p[k] for p in parts], dim=1 if sample_major else 0).contiguous()
        return out
    W, RW = _SDF[scene["sdf_blob"].data_ptr()], _COL[scene["color_x3_blob"].data_ptr()]
    D = scene["vol_cl"].shape[0]
    fm, cm = _maps(scene["cmaps"])
    q = torch.eye(4)
    q[:3, 3] = query_cam
    r = O.render(rays_o, rays_d, torch.tensor(float(near)), torch.tensor(float(far)), scene["vol_cl"].permute(3, 0, 1, 2), scene["maskvol"].view(D, D, D),
                 W, RW, torch.tensor(float(np.log(inv_s) / 10.0)), fm, cm, None, None, (scene["cmaps"].shape[2], scene["cmaps"].shape[1]), q,
                 n_samples, n_importance, alpha_inter_ratio, background, t_rand=t_rand, proj=scene["proj"], cam_pos=scene["cam_pos"])
    R, S = rays_o.shape[0], n_samples + n_importance
    pm = r["inside_sphere"]
    ge = torch.zeros(R, 2)
    ge[0, 1] = pm.sum()
    ge[0, 0] = r["gradient_error_fine"] * (pm.sum() + 1e-5)
    o = dict(mid_z=r["mid_z_vals"].t().contiguous(), pm=pm.t().contiguous(), sdf=r["sdf"].view(R, S).t().contiguous(),
             grad=r["gradients"].permute(1, 0, 2).contiguous(), color=r["color_fine"], depth=r["depth"][:, 0], weights=r["weights"].t().contiguous(),
             cdf=r["cdf_fine"].t().contiguous(), weights_sum=r["weights_sum"][:, 0], weights_max=r["weights_max"][:, 0],
             depth_var=r["depth_variance"][:, 0], alpha_sum=torch.full((R,), float(r["alpha_sum"])), grad_err=ge,
             color_mask=r["color_fine_mask"][:, 0].to(torch.uint8))
    if want_z:
        o["z_vals"] = r["z_vals"].t().contiguous()
    if want_scalars:
        o["scalars"] = torch.stack([o["alpha_sum"].mean(), o["alpha_sum"].sum() / (R * S), ge[:, 0].sum() / (ge[:, 1].sum() + 1e-5), pm.sum()]).float()
    return o


def mc_verts_to_world(verts_idx, grid_R, bound_min, bound_max):
    bmin, bmax = np.asarray(bound_min, np.float32).reshape(3), np.asarray(bound_max, np.float32).reshape(3)
    verts_idx.copy_(torch.from_numpy(verts_idx.numpy() / (grid_R - 1.0) * (bmax - bmin)[None, :] + bmin[None, :]))
    return verts_idx


def preload(device):
    return None


def camera_terms(intrinsics, w2cs):
    return torch.matmul(intrinsics, w2cs[:, :3, :]).contiguous().float(), torch.inverse(w2cs)[:, :3, 3].contiguous().float()


def marching_cubes(u, iso=0.0, index_dtype=torch.int64):
    v, t = omc.marching_cubes(u.detach().cpu().numpy(), float(iso))
    return torch.from_numpy(v), torch.from_numpy(t).to(index_dtype)


def prune_dilate(sdf_vol, mask_vol, D, threshold, radius=3):
    occ = (sdf_vol.view(D, D, D).abs() < threshold).float()[None, None]
    occ = F.max_pool3d(occ, 2 * radius + 1, stride=1, padding=radius)[0, 0] > 0
    return (occ & (mask_vol.view(D, D, D) > 0)).reshape(-1).to(torch.uint8)


# ------------------------------------------------------------------------------------------------ installation
def install(monkeypatch):
    """Patch the ops layer (and the three places that reach the HIP library without going through it) for one test."""
    ops = importlib.import_module("one-2-3-45_amd.ops")
    weights = importlib.import_module("one-2-3-45_amd.weights")
    costreg = importlib.import_module("one-2-3-45_amd.costreg")
    featurenet = importlib.import_module("one-2-3-45_amd.featurenet")
    for name in ("costvol_index", "costvol_gather", "visible_count_list", "costvol_gather_list", "build_index_grid", "scatter_dense", "sdf_mlp",
                 "pack_color_maps", "color_points", "color_from_features", "project_features", "render_rays", "camera_terms", "mc_verts_to_world", "preload", "marching_cubes", "prune_dilate", "sparse_downsample", "sparse_conv3d", "bn_act_rows", "fpn_level", "pyramid_pack", "conv2d", "conv2d_pack", "conv_x3", "scale_shift_act", "sdf_grid_tables"):
        monkeypatch.setattr(ops, name, globals()[name])
    spnn = importlib.import_module("one-2-3-45_amd.shims.torchsparse.nn")
    monkeypatch.setattr(spnn, "_require_device", lambda t: None)
    monkeypatch.setattr(costreg.CostRegNet, "forward", costreg_forward)
    monkeypatch.setattr(featurenet.InPlaceABN, "forward", abn_forward)
    # the mirrors hand packed blobs to ops: remember which parameters each blob was packed from
    pack_sdf = weights.pack_sdf_blob

    def pack_sdf_reg(W):
        arr = pack_sdf(W)
        _KEEP.append(arr)
        _SDF[arr.ctypes.data] = {k: _t(v) for k, v in W.items()}
        return arr
    monkeypatch.setattr(weights, "pack_sdf_blob", pack_sdf_reg)
    monkeypatch.setattr(weights, "CACHE_ENABLED", False)      # the registry below is keyed by the address of the array a packer returned
    for fn in ("pack_color_mfma_blob", "pack_color_x3_blob"):
        orig = getattr(weights, fn)

        def reg(sd, _orig=orig):
            arr = _orig(sd)
            _KEEP.append(arr)
            _COL[arr.ctypes.data] = {k: _t(v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in sd.items()}
            return arr
        monkeypatch.setattr(weights, fn, reg)
