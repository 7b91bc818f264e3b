"""CPU oracle for the One-2-3-45 reconstruction hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  Nothing under ``one-2-3-45_amd/`` imports it; the product path is the HIP library.

Every function is a plain torch-CPU fp32 restatement of one reference function (cited as
``file:line`` relative to ``/root/reference/reconstruction``), written independently of the
reference's code (different decomposition, no ``grid_sample`` / ``gather`` tricks) so that it is a
second opinion on the HIP kernels, and pinned against the reference itself by
``tests/test_oracle_vs_reference.py`` (runs where ``/root/reference`` exists) and by the golden
vectors under ``tests/golden/`` (generated from the reference by ``tests/golden/make_golden.py``).

Parity status of the three third-party pieces that are NOT in ``/root/reference`` (torchsparse
v1.4.0, inplace_abn, PyMCubes): restated from their published algorithms -> "parity unpinned"
(see DESIGN.md); everything else is pinned by reference-generated vectors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# a3: back projection (ops/back_project.py:5-86) -- per voxel x view projection + bilinear tap
# ----------------------------------------------------------------------------------------------
def project(world, P, H, W):
    """world [N,3], P [V,4,4] -> gx, gy, z  each [N,V] and validity mask [N,V] (ops/back_project.py:49-63)."""
    R, t = P[:, :3, :3], P[:, :3, 3]
    p = torch.einsum("vij,nj->nvi", R, world) + t[None]
    x, y, z = p[..., 0], p[..., 1], p[..., 2].clone()
    pos = z >= 0
    z[pos] = z[pos].clamp(min=1e-6)                       # negative z untouched (A.1)
    gx = 2 * (x / z) / (W - 1) - 1
    gy = 2 * (y / z) / (H - 1) - 1
    mask = (gx.abs() <= 1) & (gy.abs() <= 1) & (z > 0)
    return gx, gy, z, mask


def bilinear_zeros(maps, gx, gy):
    """maps [V,C,H,W]; gx,gy [N,V] in [-1,1] (align_corners=True, zero padding) -> [N,V,C].
    Same arithmetic as ATen's grid_sample 2-D bilinear (ops/back_project.py:73)."""
    V, C, H, W = maps.shape
    ix = (gx + 1) / 2 * (W - 1)
    iy = (gy + 1) / 2 * (H - 1)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = torch.zeros(gx.shape[0], V, C, dtype=maps.dtype)
    vi = torch.arange(V)[None].expand_as(gx)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wgt = (1 - (ix - xx).abs()) * (1 - (iy - yy).abs())
            ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            xi = xx.clamp(0, W - 1).long()
            yi = yy.clamp(0, H - 1).long()
            val = maps[vi, :, yi, xi]                       # [N,V,C]
            out += torch.where(ok, wgt, torch.zeros_like(wgt))[..., None] * val
    return out


def voxel_lattice(dims):
    """ops/generate_grids.py:4-19 -- x-major lattice [D^3,3] (grid[:,:,x,y,z]=(x,y,z))."""
    g = torch.stack(torch.meshgrid(*[torch.arange(d, dtype=torch.float32) for d in dims], indexing="ij"), -1)
    return g.reshape(-1, 3)


def costvol(feats, P, dims, voxel_size, origin, min_views=1, chunk=1 << 18):
    """Fused a1+a3+a4+a5 (sparse_sdf_network.py:286-362).  feats [V,C,H,W], P [V,4,4].
    Returns coords int32 [N,4] (x,y,z,b), volume [N,2C] = cat(var, mean), visible-view count [D^3]."""
    V, C, H, W = feats.shape
    lat = voxel_lattice(dims)
    world = lat * voxel_size + origin[None]
    cnt_all, rows, keep_idx = [], [], []
    for s in range(0, lat.shape[0], chunk):
        gx, gy, z, m = project(world[s:s + chunk], P, H, W)
        cnt = m.sum(1)
        cnt_all.append(cnt)
        keep = cnt > min_views
        if keep.any():
            f = bilinear_zeros(feats, gx[keep], gy[keep])   # all views, masked or not (A.2)
            c = 1.0 / (cnt[keep].float() + 1e-5)
            s1, s2 = f.sum(1), (f * f).sum(1)
            rows.append(torch.cat([s2 * c[:, None] - (s1 * c[:, None]) ** 2, s1 * c[:, None]], 1))
            keep_idx.append(torch.nonzero(keep)[:, 0] + s)
    cnt_all = torch.cat(cnt_all)
    idx = torch.cat(keep_idx) if keep_idx else torch.zeros(0, dtype=torch.long)
    xyz = lat[idx].to(torch.int32)
    coords = torch.cat([xyz, torch.zeros(len(idx), 1, dtype=torch.int32)], 1)
    vol = torch.cat(rows) if rows else torch.zeros(0, 2 * C)
    return coords, vol, cnt_all


# ----------------------------------------------------------------------------------------------
# a6: sparse cost-regularisation U-Net on torchsparse v1.4.0 semantics (tsparse/modules.py:94-124,
#     259-304).  torchsparse is not in /root/reference: restated from its published algorithm
#     (nn/functional/{conv,downsample}.py, nn/utils/kernel.py @ v1.4.0) -- PARITY UNPINNED.
# ----------------------------------------------------------------------------------------------
def kernel_offsets(ts):
    """get_kernel_offsets(3, stride=ts): x fastest, then y, then z (odd kernel volume)."""
    r = [-ts, 0, ts]
    return torch.tensor([[x, y, z] for z in r for y in r for x in r], dtype=torch.int64)


def _key(c):            # c int64 [N,3] with values >= -8
    c = c + 8
    return (c[:, 0] * 4096 + c[:, 1]) * 4096 + c[:, 2]


class SparseLevel:
    """Active coordinates of one tensor stride (all batch 0), sorted lookup by key."""

    def __init__(self, xyz, ts):
        self.xyz, self.ts = xyz.long(), ts
        k = _key(self.xyz)
        self.sorted_key, self.order = torch.sort(k)

    def lookup(self, q):
        k = _key(q)
        pos = torch.searchsorted(self.sorted_key, k).clamp(max=len(self.sorted_key) - 1)
        hit = self.sorted_key[pos] == k
        return torch.where(hit, self.order[pos], torch.full_like(pos, -1))


def downsample_coords(level):
    """spdownsample(coords, stride=2, kernel_size=3, tensor_stride=ts): out = unique{c+off : all comps
    % (2 ts) == 0 and >= per-axis min of the input coords}, sorted by (b,x,y,z)."""
    ts = level.ts
    off = kernel_offsets(ts)
    cand = (level.xyz[:, None, :] + off[None]).reshape(-1, 3)
    ok = ((cand % (2 * ts)) == 0).all(1) & (cand >= level.xyz.min(0).values[None]).all(1)
    cand = torch.unique(cand[ok], dim=0)                   # lexicographic (x,y,z) order
    return SparseLevel(cand, 2 * ts)


def build_kmap(lin, lout):
    """For every output row q and offset k: the input row at q + off_k * ts_in (or -1).  [Nout,27]."""
    off = kernel_offsets(lin.ts)
    nb = torch.stack([lin.lookup(lout.xyz + off[k][None]) for k in range(27)], 1)
    return nb


def sparse_conv(x, kmap, Wk, transposed=False, n_out=None):
    """x [Nin,Cin], Wk [27,Cin,Cout].  Forward: out[q] += x[kmap[q,k]] @ Wk[k].
    Transposed (maps cached from the matching down conv, roles swapped): out[kmap[q,k]] += x[q] @ Wk[k]."""
    if not transposed:
        out = torch.zeros(kmap.shape[0], Wk.shape[2], dtype=x.dtype)
        for k in range(27):
            idx = kmap[:, k]
            v = idx >= 0
            if v.any():
                out[v] += x[idx[v]] @ Wk[k]
        return out
    out = torch.zeros(n_out, Wk.shape[2], dtype=x.dtype)
    for k in range(27):
        idx = kmap[:, k]
        v = idx >= 0
        if v.any():
            out.index_add_(0, idx[v], x[v] @ Wk[k])
    return out


def bn_relu_rows(x, gamma, beta, eps=1e-5):
    """spnn.BatchNorm (= BatchNorm1d over rows) in training mode (the runner never calls .eval(),
    SURVEY finding 5) + ReLU.  Biased batch variance."""
    mu = x.mean(0)
    var = ((x - mu) ** 2).mean(0)
    return torch.relu((x - mu) / torch.sqrt(var + eps) * gamma + beta)


def sparse_costreg(feat, coords, w):
    """SparseCostRegNet.forward (tsparse/modules.py:287-304).  ``w``: dict name -> (kernel[27,Ci,Co], gamma, beta)
    for conv0,1,2,3,4,5,6,7,9,11.  coords int [N,4] (x,y,z,b), batch 0 only.  Returns [N,16] in input row order."""
    L0 = SparseLevel(coords[:, :3], 1)
    L1 = downsample_coords(L0)
    L2 = downsample_coords(L1)
    L3 = downsample_coords(L2)
    k00, k11, k22, k33 = (build_kmap(L, L) for L in (L0, L1, L2, L3))
    k01, k12, k23 = build_kmap(L0, L1), build_kmap(L1, L2), build_kmap(L2, L3)

    def blk(name, x, kmap, transposed=False, n_out=None):
        K, g, b = w[name]
        return bn_relu_rows(sparse_conv(x, kmap, K, transposed, n_out), g, b)

    c0 = blk("conv0", feat, k00)
    c2 = blk("conv2", blk("conv1", c0, k01), k11)
    c4 = blk("conv4", blk("conv3", c2, k12), k22)
    x = blk("conv6", blk("conv5", c4, k23), k33)
    x = c4 + blk("conv7", x, k23, True, len(L2.xyz))
    x = c2 + blk("conv9", x, k12, True, len(L1.xyz))
    x = c0 + blk("conv11", x, k01, True, len(L0.xyz))
    return x, dict(levels=(L0, L1, L2, L3))


# a7: sparse_to_dense_volume (sparse_sdf_network.py:252-284, tsparse/torchsparse_utils.py:125-130)
def scatter_dense(coords, feat, dims):
    D0, D1, D2 = dims
    C = feat.shape[1]
    dense = torch.zeros(D0, D1, D2, C)
    mask = torch.zeros(D0, D1, D2, 1)
    x, y, z = coords[:, 0].long(), coords[:, 1].long(), coords[:, 2].long()
    dense[x, y, z] = feat
    mask[x, y, z] = 1
    return dense.permute(3, 0, 1, 2).contiguous()[None], mask.permute(3, 0, 1, 2).contiguous()[None]


# ----------------------------------------------------------------------------------------------
# a2: compress layer = Conv3x3 (no bias) + InPlaceABN in training mode (featurenet.py:12-22,
#     sparse_sdf_network.py:171-173,312).  inplace_abn is not in /root/reference -- PARITY UNPINNED
#     for the |gamma|+eps detail, exposed as a flag.
# ----------------------------------------------------------------------------------------------
def abn_train(x, gamma, beta, eps=1e-5, slope=0.01, abs_gamma=True):
    mu = x.mean((0, 2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean((0, 2, 3), keepdim=True)
    g = (gamma.abs() + eps) if abs_gamma else gamma
    y = (x - mu) / torch.sqrt(var + eps) * g.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    return torch.where(y >= 0, y, y * slope)


# ----------------------------------------------------------------------------------------------
# a8: the reference's own trilinear sampler (ops/grid_sampler.py:64-216) incl. its edge rules (A.3)
# ----------------------------------------------------------------------------------------------
def trilinear_ref(volume, pts):
    """volume [C,D,D,D] (axes x,y,z; cubic), pts [P,3] (x,y,z in [-1,1]) -> [P,C].
    zero unless 0 < i < D on every axis; corner indices clamped, weights from unclamped corners."""
    C, D = volume.shape[0], volume.shape[1]
    assert volume.shape[1] == volume.shape[2] == volume.shape[3], "reference linear index is cubic-only (A.3)"
    i = (pts + 1) / 2 * (D - 1)
    ok = ((i > 0) & (i < D)).all(1)
    i0 = torch.floor(i)
    out = torch.zeros(pts.shape[0], C, dtype=volume.dtype)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                c = i0 + torch.tensor([dx, dy, dz], dtype=i.dtype)
                opp = i0 + torch.tensor([1 - dx, 1 - dy, 1 - dz], dtype=i.dtype)
                wgt = ((opp - i) * torch.tensor([1.0 if d == 0 else -1.0 for d in (dx, dy, dz)])).prod(1)
                ci = c.clamp(0, D - 1).long()
                out += wgt[:, None] * volume[:, ci[:, 0], ci[:, 1], ci[:, 2]].T
    return torch.where(ok[:, None], out, torch.zeros_like(out))


def trilinear_ref_jac(volume, pts):
    """d(trilinear_ref)/d pts -> [P,C,3] (what autograd sees through ops/grid_sampler.py: floor/clamp are
    constants, the 8 weights are linear in the coordinate; masked points have zero gradient)."""
    C, D = volume.shape[0], volume.shape[1]
    i = (pts + 1) / 2 * (D - 1)
    ok = ((i > 0) & (i < D)).all(1)
    i0 = torch.floor(i)
    jac = torch.zeros(pts.shape[0], C, 3, dtype=volume.dtype)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                d = (dx, dy, dz)
                opp = i0 + torch.tensor([1 - a for a in d], dtype=i.dtype)
                sgn = torch.tensor([1.0 if a == 0 else -1.0 for a in d])
                f = (opp - i) * sgn                              # per-axis weight factor
                ci = (i0 + torch.tensor(d, dtype=i.dtype)).clamp(0, D - 1).long()
                val = volume[:, ci[:, 0], ci[:, 1], ci[:, 2]].T   # [P,C]
                for ax in range(3):
                    oth = [a for a in range(3) if a != ax]
                    dw = -sgn[ax] * f[:, oth[0]] * f[:, oth[1]] * (D - 1) / 2
                    jac[:, :, ax] += dw[:, None] * val
    return torch.where(ok[:, None, None], jac, torch.zeros_like(jac))


# a18: occupancy lookup, F.grid_sample(mode='nearest', align_corners=False) (sparse_neus_renderer.py:153-169)
def mask_nearest(maskvol, pts):
    """maskvol [D,D,D] (x,y,z), pts [P,3] -> [P]; index = rne(((g+1) D - 1)/2), out of range -> 0."""
    D = maskvol.shape[0]
    idx = torch.round(((pts + 1) * D - 1) / 2)              # torch.round = half-to-even = nearbyint
    ok = ((idx >= 0) & (idx <= D - 1)).all(1)
    ci = idx.clamp(0, D - 1).long()
    return torch.where(ok, maskvol[ci[:, 0], ci[:, 1], ci[:, 2]], torch.zeros(()))


# ATen trilinear, zeros padding, align_corners=True (render_utils.py:54-85)
def trilinear_zeros(volume, pts):
    C, D0, D1, D2 = volume.shape
    dims = torch.tensor([D0, D1, D2], dtype=pts.dtype)
    i = (pts + 1) / 2 * (dims - 1)
    i0 = torch.floor(i)
    out = torch.zeros(pts.shape[0], C, dtype=volume.dtype)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                c = i0 + torch.tensor([dx, dy, dz], dtype=i.dtype)
                wgt = (1 - (i - c).abs()).prod(1)
                ok = ((c >= 0) & (c <= dims - 1)).all(1)
                ci = torch.minimum(c.clamp(min=0), dims - 1).long()
                out += torch.where(ok, wgt, torch.zeros_like(wgt))[:, None] * volume[:, ci[:, 0], ci[:, 1], ci[:, 2]].T
    return out


# ----------------------------------------------------------------------------------------------
# a9-a12: embedding + LatentSDFLayer (embedder.py:63-101, sparse_sdf_network.py:35-136,402-420,476-499)
# ----------------------------------------------------------------------------------------------
def embed(x, n_freq=6):
    out = [x]
    for k in range(n_freq):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def softplus100(x):
    return torch.where(x * 100 > 20, x, torch.log1p(torch.exp(x * 100)) / 100)


def fold_weight_norm(g, v):
    """nn.utils.weight_norm(dim=0): w = g * v / ||v||_row."""
    return g * v / v.norm(dim=1, keepdim=True)


def sdf_mlp(pts, latent, W):
    """W = dict(w0[128,39], b0, w1[128,144], b1, w2[128,144], b2) (weight-norm already folded) -> [P,128]."""
    h = softplus100(embed(pts) @ W["w0"].T + W["b0"])
    h = softplus100(torch.cat([h, latent], 1) @ W["w1"].T + W["b1"])
    return torch.cat([h, latent], 1) @ W["w2"].T + W["b2"]


SDF_NOISE = None      # (relative sigma, torch.Generator[, absolute sigma]) or None.  Sensitivity probe ONLY (tests/fullsize_util.py,
                      # tests/render_check.py): multiplies the SDF output by (1 + sigma * N(0,1)) and adds abs_sigma * N(0,1) to measure how the
                      # reference ALGORITHM (hierarchical sampler, sigmoids of slope inv_s) amplifies fp32-class SDF differences.


def sdf(pts, volume, W):
    """SparseSdfNetwork.sdf (sparse_sdf_network.py:402-420): volume [C,D,D,D] -> (y[P,128], latent[P,16])."""
    lat = trilinear_ref(volume, pts)
    y = sdf_mlp(pts, lat, W)
    if SDF_NOISE is not None:
        sigma, gen = SDF_NOISE[:2]
        y0 = y[:, :1] * (1 + sigma * torch.randn(y.shape[0], 1, generator=gen))
        if len(SDF_NOISE) > 2:
            y0 = y0 + SDF_NOISE[2] * torch.randn(y.shape[0], 1, generator=gen)
        y = torch.cat([y0, y[:, 1:]], 1)
    return y, lat


def sdf_grad(pts, volume, W):
    """Analytic d sdf / d x, equal to autograd through a8+a10 (sparse_sdf_network.py:476-499)."""
    lat = trilinear_ref(volume, pts)
    pe = embed(pts)
    a0 = pe @ W["w0"].T + W["b0"]
    h0 = softplus100(a0)
    a1 = torch.cat([h0, lat], 1) @ W["w1"].T + W["b1"]
    s0 = torch.where(a0 * 100 > 20, torch.ones_like(a0), torch.sigmoid(a0 * 100))
    s1 = torch.where(a1 * 100 > 20, torch.ones_like(a1), torch.sigmoid(a1 * 100))
    g1 = W["w2"][0, :128] * s1                              # d y0 / d a1
    gin1 = g1 @ W["w1"]                                     # [P,144]
    glat = gin1[:, 128:] + W["w2"][0, 128:]
    gpe = (gin1[:, :128] * s0) @ W["w0"]                    # [P,39]
    gx = gpe[:, :3].clone()
    for k in range(6):
        f = 2.0 ** k
        gx += gpe[:, 3 + 6 * k:6 + 6 * k] * torch.cos(pts * f) * f
        gx -= gpe[:, 6 + 6 * k:9 + 6 * k] * torch.sin(pts * f) * f
    gx += torch.einsum("pc,pcx->px", glat, trilinear_ref_jac(volume, pts))
    return gx


# ----------------------------------------------------------------------------------------------
# a17, a19: hierarchical sampling (sparse_neus_renderer.py:73-151, render_utils.py:8-51)
# ----------------------------------------------------------------------------------------------
def sample_pdf_det(bins, weights, n, diag=None):
    """diag (list or None): receives, per call, the pdf mass [R,n] of the bin every new sample lands in.  A sample in a
    bin of mass p is placed with a sensitivity of (bin width / p) to the cdf: p ~ 1e-5 ("empty" bin of a ray that hits the
    surface) amplifies fp32 rounding in the cdf (1e-7) to ~1 % of a bin -- the conditioning measure used by the full-size test."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.5 / n, 1 - 0.5 / n, n).expand(cdf.shape[0], n).contiguous()
    ind = torch.searchsorted(cdf, u, right=True)
    lo = (ind - 1).clamp(min=0)
    hi = ind.clamp(max=cdf.shape[1] - 1)
    if diag is not None:
        diag.append(pdf.gather(1, lo.clamp(max=pdf.shape[1] - 1)))          # [R,n]: pdf mass of the bin each new sample lands in
    c0, c1 = cdf.gather(1, lo), cdf.gather(1, hi)
    b0, b1 = bins.gather(1, lo), bins.gather(1, hi)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def up_sample(rays_o, rays_d, z, sdf_v, n_imp, inv_s, maskvol, diag=None):
    pts = rays_o[:, None] + rays_d[:, None] * z[..., None]
    m = mask_nearest(maskvol, pts.reshape(-1, 3)).reshape(z.shape)
    m = m[:, :-1] * m[:, 1:]
    ps, ns, pz, nz = sdf_v[:, :-1], sdf_v[:, 1:], z[:, :-1], z[:, 1:]
    mid = (ps + ns) * 0.5
    dot = (ns - ps) / (nz - pz + 1e-5)
    prev = torch.cat([torch.zeros_like(dot[:, :1]), dot[:, :-1]], 1)
    dot = torch.minimum(prev, dot).clip(-10.0, 0.0) * m
    dist = nz - pz
    pc = torch.sigmoid((mid - dot * dist * 0.5) * inv_s)
    nc = torch.sigmoid((mid + dot * dist * 0.5) * inv_s)
    alpha = m * ((pc - nc + 1e-5) / (pc + 1e-5))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-7], 1), 1)[:, :-1]
    return sample_pdf_det(z, alpha * T, n_imp, diag)


CAT_Z_MIN_VALID = 1      # the reference's rule (:137).  Tests set 0 to show that a case they construct is SENSITIVE to the rule (never used otherwise)


def cat_z(rays_o, rays_d, z, new_z, sdf_v, volume, maskvol, W):
    pts = (rays_o[:, None] + rays_d[:, None] * new_z[..., None]).reshape(-1, 3)
    m = mask_nearest(maskvol, pts) > 0
    new_sdf = torch.full((pts.shape[0],), 100.0)
    if m.sum() > CAT_Z_MIN_VALID:                            # quirk: skipped entirely if <= 1 valid point
        new_sdf[m] = sdf(pts[m], volume, W)[0][:, 0]
    zz = torch.cat([z, new_z], 1)
    ss = torch.cat([sdf_v, new_sdf.reshape(new_z.shape)], 1)
    zz, idx = torch.sort(zz, dim=1, stable=True)
    return zz, ss.gather(1, idx)


# ----------------------------------------------------------------------------------------------
# a20: Projector.compute / compute_view_independent (projector.py:15-62,96-228,231-425,
#      render_utils.py:54-120, ops/back_project.py:89-129)
# ----------------------------------------------------------------------------------------------
def project_pts(pts, K, w2c, W_img, H_img, Pm=None):
    """cam2pixel with padding 'zeros': Z.clamp(min=1e-3), out-of-range coordinate -> 2.  pts [P,3] -> gx,gy [V,P].
    Pm [V,3,4]: the product K @ w2c[:, :3] if the caller already has it (the C ABI's `proj`)."""
    if Pm is None:
        Pm = K @ w2c[:, :3, :]
    p = torch.einsum("vij,pj->vpi", Pm[:, :, :3], pts) + Pm[:, None, :, 3]
    Z = p[..., 2].clamp(min=1e-3)
    gx = 2 * (p[..., 0] / Z) / (W_img - 1) - 1
    gy = 2 * (p[..., 1] / Z) / (H_img - 1) - 1
    gx = torch.where((gx > 1) | (gx < -1), torch.full_like(gx, 2.0), gx)
    gy = torch.where((gy > 1) | (gy < -1), torch.full_like(gy, 2.0), gy)
    return gx, gy


def ray_diff(pts, query_dir, cam_pos):
    """query_dir [P,3] (already normalised the reference's way), cam_pos [V,3] -> [V,P,4]."""
    sup = cam_pos[:, None] - pts[None]
    sup = sup / (sup.norm(dim=-1, keepdim=True) + 1e-6)
    d = query_dir[None] - sup
    n = d.norm(dim=-1, keepdim=True)
    dot = (query_dir[None] * sup).sum(-1, keepdim=True)
    return torch.cat([d / n.clamp(min=1e-6), dot], -1)


def projector(pts, volume, maskvol, feat_maps, color_maps, w2cs, K, img_wh, query_cam=None, normals=None, proj=None, cam_pos=None):
    """pts [P,3] -> geo [P,16], rgb_feat [V,P,59] (colour first), ray_diff [V,P,4], mask [V,P].
    proj [V,3,4] / cam_pos [V,3] may be given instead of (K, w2cs) (the form the C ABI takes)."""
    geo = trilinear_zeros(volume, pts)
    inside = (pts.abs() < 1).all(1)
    gmask = inside & (trilinear_zeros(maskvol[None], pts)[:, 0] > 0)
    gx, gy = project_pts(pts, K, w2cs, img_wh[0], img_wh[1], Pm=proj)
    pmask = (gx.abs() < 1) & (gy.abs() < 1)
    feats = bilinear_zeros(feat_maps, gx.T, gy.T).permute(1, 0, 2)      # [V,P,56]
    cols = bilinear_zeros(color_maps, gx.T, gy.T).permute(1, 0, 2)
    if cam_pos is None:
        cam_pos = torch.inverse(w2cs)[:, :3, 3]
    if normals is None:
        q = query_cam[None] - pts
        q = q / (q.norm(dim=-1, keepdim=True) + 1e-6)
    else:
        q = normals
    return geo, torch.cat([cols, feats], -1), ray_diff(pts, q, cam_pos), gmask[None] & pmask


# a22: GeneralRenderingNetwork.forward (rendering_network.py:75-129); RW = dict of Linear weights
def elu(x):
    return torch.where(x > 0, x, torch.expm1(x))


def _lin(RW, name, x):
    return x @ RW[name + ".weight"].T + RW[name + ".bias"]


def rendering_network(RW, geo, rgb_feat, rdiff, mask):
    """geo [P,16], rgb_feat [V,P,59], rdiff [V,P,4], mask [V,P] -> rgb [P,3], n_valid_views [P]."""
    rf = rgb_feat.permute(1, 0, 2)
    rd = rdiff.permute(1, 0, 2)
    m = mask.permute(1, 0)[..., None].float()
    V = rf.shape[1]
    dfeat = elu(_lin(RW, "ray_dir_fc.2", elu(_lin(RW, "ray_dir_fc.0", rd))))
    rgb_in = rf[..., :3]
    rf = rf + dfeat
    e = torch.exp(RW["s"].abs() * (rd[..., 3:] - 1))
    wgt = (e - e.min(1, keepdim=True).values) * m
    wgt = wgt / (wgt.sum(1, keepdim=True) + 1e-8)
    mean = (rf * wgt).sum(1, keepdim=True)
    var = (wgt * (rf - mean) ** 2).sum(1, keepdim=True)
    x = torch.cat([geo[:, None].expand(-1, V, -1), mean.expand(-1, V, -1), var.expand(-1, V, -1), rf], -1)
    x = elu(_lin(RW, "base_fc.2", elu(_lin(RW, "base_fc.0", x))))
    xv = elu(_lin(RW, "vis_fc.2", elu(_lin(RW, "vis_fc.0", x * wgt))))
    vis = torch.sigmoid(xv[..., 32:]) * m
    x = x + xv[..., :32]
    vis = torch.sigmoid(_lin(RW, "vis_fc2.2", elu(_lin(RW, "vis_fc2.0", x * vis)))) * m
    x = torch.cat([x, vis, rd], -1)
    x = _lin(RW, "rgb_fc.4", elu(_lin(RW, "rgb_fc.2", elu(_lin(RW, "rgb_fc.0", x)))))
    x = x.masked_fill(m == 0, -1e9)
    rgb = (rgb_in * torch.softmax(x, 1)).sum(1)
    return rgb, m.sum(1)[:, 0]


# ----------------------------------------------------------------------------------------------
# a16, a21: render / render_core (sparse_neus_renderer.py:171-635), general rendering; perturb > 0 when t_rand is given
# ----------------------------------------------------------------------------------------------
def render(rays_o, rays_d, near, far, volume, maskvol, W, RW, variance, feat_maps, color_maps, w2cs, K, img_wh,
           query_c2w, n_samples=64, n_importance=64, alpha_inter_ratio=1.0, background_rgb=1.0, t_rand=None, diag=None, trace=None,
           proj=None, cam_pos=None):
    """t_rand [R, n_samples]: the stratified jitter of :506-515 (the reference draws torch.rand(z_vals.shape) on the host).
    diag: see sample_pdf_det.  trace (list or None): receives per up-sampling round the sampler's input state and output
    dict(z [R,S], sdf [R,S], inv_s, new_z [R,n]) -- lets a test drive the HIP sampler stage with IDENTICAL inputs."""
    R = rays_o.shape[0]
    sample_dist = float((far - near) / n_samples)
    z = (near + (far - near) * torch.linspace(0, 1, n_samples))[None].repeat(R, 1)
    if t_rand is not None:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], 1)
        lower = torch.cat([z[:, :1], mids], 1)
        z = lower + (upper - lower) * t_rand
    pts = (rays_o[:, None] + rays_d[:, None] * z[..., None]).reshape(-1, 3)
    s = sdf(pts, volume, W)[0][:, 0].reshape(R, n_samples)     # coarse pass is NOT masked (:525-528)
    for i in range(4):
        nz = up_sample(rays_o, rays_d, z, s, n_importance // 4, 64.0 * 2 ** i, maskvol, diag)
        if trace is not None:
            trace.append(dict(z=z.clone(), sdf=s.clone(), inv_s=64.0 * 2 ** i, new_z=nz.clone()))
        z, s = cat_z(rays_o, rays_d, z, nz, s, volume, maskvol, W)
    return render_core(rays_o, rays_d, z, sample_dist, volume, maskvol, W, RW, variance, feat_maps, color_maps, w2cs, K, img_wh,
                       query_c2w, alpha_inter_ratio, background_rgb, proj=proj, cam_pos=cam_pos)


def render_core(rays_o, rays_d, z, sample_dist, volume, maskvol, W, RW, variance, feat_maps, color_maps, w2cs, K, img_wh, query_c2w,
                alpha_inter_ratio=1.0, background_rgb=1.0, proj=None, cam_pos=None):
    """Everything of render() after the hierarchical sampling (sparse_neus_renderer.py:555-635 + render_core :171-455) for GIVEN
    sorted sample depths z [R,S]."""
    R = rays_o.shape[0]
    S = z.shape[1]
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), sample_dist)], 1)
    mid = z + dists * 0.5
    pts = (rays_o[:, None] + rays_d[:, None] * mid[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None].expand(R, S, 3).reshape(-1, 3)
    pm = mask_nearest(maskvol, pts)
    mb = pm > 0
    if mb.sum() < 1:
        mb[:100] = True
    sdf_v = torch.full((R * S,), 100.0)
    grad = torch.zeros(R * S, 3)
    sdf_v[mb] = sdf(pts[mb], volume, W)[0][:, 0]
    grad[mb] = sdf_grad(pts[mb], volume, W)
    geo, rf, rdiff, vmask = projector(pts, volume, maskvol, feat_maps, color_maps, w2cs, K, img_wh,
                                      query_cam=query_c2w[:3, 3], proj=proj, cam_pos=cam_pos)
    rgb, nvalid = rendering_network(RW, geo, rf, rdiff, vmask)
    inv_s = torch.exp(variance * 10.0).clip(1e-6, 1e6)
    tdot = (dirs * grad).sum(-1)
    icos = -(torch.relu(-tdot * 0.5 + 0.5) * (1 - alpha_inter_ratio) + torch.relu(-tdot) * alpha_inter_ratio) * pm
    half = icos.clip(-10, 10) * dists.reshape(-1) * 0.5
    pc = torch.sigmoid((sdf_v - half) * inv_s)
    nc = torch.sigmoid((sdf_v + half) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).reshape(R, S).clip(0, 1) * pm.reshape(R, S)
    T = torch.cumprod(torch.cat([torch.ones(R, 1), 1 - alpha + 1e-7], 1), 1)[:, :-1]
    wts = alpha * T
    wsum = wts.sum(1, keepdim=True)
    color = (rgb.reshape(R, S, 3) * wts[..., None]).sum(1) + background_rgb * (1 - wsum)
    depth = (mid * wts).sum(1, keepdim=True)
    cmask = ((nvalid.reshape(R, S) >= 2).float().sum(1) > 8)[:, None]      # (N_rays, 1) like the reference
    gerr = (pm.reshape(R, S) * (grad.reshape(R, S, 3).norm(dim=-1) - 1) ** 2).sum() / (pm.sum() + 1e-5)
    return dict(color_fine=color, color_fine_mask=cmask, depth=depth, weights=wts, weights_sum=wsum,
                gradients=grad.reshape(R, S, 3), sdf=sdf_v.reshape(-1, 1), z_vals=z, mid_z_vals=mid,
                depth_variance=((mid - depth) ** 2 * wts).sum(1, keepdim=True), cdf_fine=pc.reshape(R, S),
                alpha_sum=alpha.sum(1).mean(), alpha_mean=alpha.mean(), gradient_error_fine=gerr,
                weights_max=wts.max(1, keepdim=True).values, inside_sphere=pm.reshape(R, S))


# a23: extract_fields (sparse_neus_renderer.py:881-905): u = -sdf on linspace(-1,1,R)^3
def sdf_grid(volume, W, R, chunk=1 << 18):
    lin = torch.linspace(-1, 1, R)
    u = torch.empty(R ** 3)
    g = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    for s in range(0, R ** 3, chunk):
        u[s:s + chunk] = -sdf(g[s:s + chunk], volume, W)[0][:, 0]
    return u.reshape(R, R, R)


# ----------------------------------------------------------------------------------------------
# lod 1 (coarse-to-fine): a13 get_sdf_volume (sparse_sdf_network.py:441-474), a26
# get_valid_sparse_coords_by_sdf (sparse_neus_renderer.py:822-879), upsample (sparse_sdf_network.py:198-219) and the
# lod > 0 branch of get_conditional_volume (sparse_sdf_network.py:335-372)
# ----------------------------------------------------------------------------------------------
def sdf_volume(dense, mask, W, voxel_size, origin):
    """SDF at every valid voxel centre using the voxel's OWN latent (no interpolation); invalid voxels = 1.
    dense [C,D,D,D], mask [D,D,D] -> [D,D,D]."""
    C, D0, D1, D2 = dense.shape
    m = mask.reshape(-1) > 0
    pts = voxel_lattice([D0, D1, D2]) * voxel_size + origin[None]
    lat = dense.reshape(C, -1).T
    out = torch.ones(D0 * D1 * D2)
    out[m] = sdf_mlp(pts[m], lat[m], W)[:, 0]
    return out.reshape(D0, D1, D2)


def prune_by_sdf(sdf_vol, mask, threshold=0.02, maximum_pts=110000):
    """|sdf| < thr, dilated by a 7^3 box, AND the valid mask; thr lowered by 0.002 while too many voxels remain.
    Returns (x-major boolean mask [D,D,D], final threshold).  The reference then drops random voxels (np.random.choice)
    if the count is still above maximum_pts -- not reproducible, reported via the returned count instead."""
    def prune(thr):
        occ = (sdf_vol.abs() < thr).float()[None, None]
        occ = F.avg_pool3d(occ, kernel_size=7, stride=1, padding=3)[0, 0] > 0
        return occ & (mask > 0)
    thr = threshold
    fm = prune(thr)
    while fm.sum() > maximum_pts and thr > 0.003:
        thr = thr - 0.002
        fm = prune(thr)
    return fm, thr


def upsample8(pre_feat, pre_coords):
    """Each parent (b,x,y,z) -> 8 children at +{0,1}^3 in the reference's order: base, +x, +y, +z, +xy, +xz, +yz, +xyz."""
    off = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]], dtype=pre_coords.dtype)
    c = pre_coords[:, None, :].repeat(1, 8, 1)
    c[:, :, 1:] += off[None]
    return pre_feat[:, None, :].expand(-1, 8, -1).reshape(-1, pre_feat.shape[1]), c.reshape(-1, 4)


def costvol_list(feats, P, xyz, voxel_size, origin):
    """Back-projection + aggregation for an explicit voxel list xyz [M,3] (float lattice coords): keeps voxels seen by > 1
    views (lod > 0 filter, sparse_sdf_network.py:352-357).  Returns (keep mask [M], rows [N,2C])."""
    V, C, H, W = feats.shape
    gx, gy, z, m = project(xyz * voxel_size + origin[None], P, H, W)
    cnt = m.sum(1)
    keep = cnt > 1
    f = bilinear_zeros(feats, gx[keep], gy[keep])
    c = 1.0 / (cnt[keep].float() + 1e-5)
    s1, s2 = f.sum(1), (f * f).sum(1)
    return keep, torch.cat([s2 * c[:, None] - (s1 * c[:, None]) ** 2, s1 * c[:, None]], 1)


def sparse_costreg_unordered(feat, coords, w):
    """SparseCostRegNet on an arbitrary-order coordinate list (lod 1): same maths as sparse_costreg, rows stay in input order."""
    return sparse_costreg(feat, coords, w)


# ----------------------------------------------------------------------------------------------
# f1: FeatureNet (models/featurenet.py:40-91) + the fused pyramid of GenericTrainer.obtain_pyramid_feature_maps
#     (models/trainer_generic.py:1104-1125) + the compress layer (sparse_sdf_network.py:171-173, 312).
#     sd = the reference module's own state dict (conv0.0.conv.weight, conv0.0.bn.{weight,bias}, toplayer.{weight,bias}, ...).
#     Pinned to the reference's FeatureNet by tests/golden/ref_featurenet.npz (tests/test_golden_oracle.py).
# ----------------------------------------------------------------------------------------------
def conv_abn(x, sd, prefix, stride=1):
    """ConvBnReLU (featurenet.py:12-22): Conv2d without bias, padding = k // 2, then InPlaceABN with batch statistics (the runner
    never calls .eval(): SURVEY finding 5)."""
    w = sd[prefix + ".conv.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    return abn_train(y, sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"])


def featurenet(imgs, sd):
    """[V,3,H,W] -> [feat2 (32 @ H/4), feat1 (16 @ H/2), feat0 (8 @ H)], featurenet.py:68-91."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    lin = lambda t, name, pad=0: F.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], padding=pad)
    c0 = conv_abn(conv_abn(imgs, sd, "conv0.0"), sd, "conv0.1")
    c1 = conv_abn(conv_abn(conv_abn(c0, sd, "conv1.0", 2), sd, "conv1.1"), sd, "conv1.2")
    c2 = conv_abn(conv_abn(conv_abn(c1, sd, "conv2.0", 2), sd, "conv2.1"), sd, "conv2.2")
    f2 = lin(c2, "toplayer")
    f1 = up(f2) + lin(c1, "lat1")
    f0 = up(f1) + lin(c0, "lat0")
    return [f2, lin(f1, "smooth1", 1), lin(f0, "smooth0", 1)]


def fused_pyramid(imgs, sd):
    """trainer_generic.py:1117-1123: cat(x4 bilinear of feat2, x2 bilinear of feat1, feat0) -> [V,56,H,W]."""
    f2, s1, s0 = featurenet(imgs, sd)
    return torch.cat([F.interpolate(f2, scale_factor=4, mode="bilinear", align_corners=True),
                      F.interpolate(s1, scale_factor=2, mode="bilinear", align_corners=True), s0], 1)


def conditional_volume(imgs, feat_sd, compress_sd, costreg_w, P, dims, voxel_size, origin, fmaps=None):
    """get_conditional_volume at lod 0 from the IMAGES (sparse_sdf_network.py:286-400 behind trainer_generic.py:1104-1125):
    FeatureNet -> fused pyramid -> compress layer -> back-projection + aggregation -> SparseCostRegNet -> dense scatter.
    -> dict(fmaps [V,56,H,W], feats16 [V,16,H,W], coords [N,3], rows [N,32], rows16 [N,16], dense [1,16,D,D,D], mask [1,1,D,D,D])."""
    fmaps = fused_pyramid(imgs, feat_sd) if fmaps is None else fmaps
    f16 = abn_train(F.conv2d(fmaps, compress_sd["conv.weight"], None, padding=1), compress_sd["bn.weight"], compress_sd["bn.bias"])
    coords, rows, cnt = costvol(f16, P, dims, voxel_size, origin)
    rows16, extra = sparse_costreg(rows, coords, costreg_w)
    dense, mask = scatter_dense(coords, rows16, dims)
    return dict(fmaps=fmaps, feats16=f16, coords=coords, rows=rows, cnt=cnt, rows16=rows16, dense=dense, mask=mask, levels=extra["levels"])
