"""GPU parity: every HIP entry point (through the C ABI) against the CPU oracle on the same seeded inputs.
Tolerances: fp32 -- 2e-5 relative to the tensor's max magnitude unless stated; integer / index work bit-exact."""
import importlib

import numpy as np
import pytest
import torch

from oracle import mc as omc
from oracle import recon as O
from scene_util import color_t, costreg_oracle_weights, rays_for, sdfW_t, small_scene

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("one-2-3-45_amd")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from importlib import import_module
    import_module("one-2-3-45_amd._lib").lib()          # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("one-2-3-45_amd.ops")


def close(a, b, rel=2e-5, what=""):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(b).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3g}"


def dev_scene(s, dev, ops):
    """Device-side copies of a small_scene (volumes, maps, packed weights)."""
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev)
    feats = t(s["f16"]).contiguous()
    vol_cl = s["dense"][0].permute(1, 2, 3, 0).contiguous().to(dev)
    maskvol = s["mask"][0, 0].contiguous().to(dev)
    sc = s["sc"]
    Kt, w2c = torch.from_numpy(sc["intrinsics"]), torch.from_numpy(sc["w2cs"])
    proj = (Kt @ w2c[:, :3, :]).contiguous().to(dev)
    cam_pos = torch.inverse(w2c)[:, :3, 3].contiguous().to(dev)
    cmaps = ops.pack_color_maps(t(s["fmaps"]).contiguous(), t(sc["images"]).contiguous())
    return dict(feats=feats, vol_cl=vol_cl, maskvol=maskvol, proj=proj, cam_pos=cam_pos, cmaps=cmaps,
                sdf_blob=t(pkg.weights.pack_sdf_blob(s["sdfW"])),
                color_mfma_blob=t(pkg.weights.pack_color_mfma_blob(s["color_sd"])),
                color_x3_blob=t(pkg.weights.pack_color_x3_blob(s["color_sd"])),
                aff=t(sc["affine_mats"]).contiguous())


def test_costvol(dev, ops):
    s = small_scene()
    d = dev_scene(s, dev, ops)
    D, V, H, W = s["D"], s["V"], s["H"], s["W"]
    nhwc = ops.nchw_to_nhwc(d["feats"])
    assert torch.equal(nhwc.cpu(), torch.from_numpy(s["f16"]).permute(0, 2, 3, 1))
    cnt, row, coords, n = ops.costvol_index(d["aff"], V, H, W, (D, D, D), s["voxel_size"], s["sc"]["partial_vol_origin"])
    assert n == s["coords"].shape[0]
    assert torch.equal(cnt.cpu(), s["cnt"].to(torch.uint8))
    assert torch.equal(coords.cpu(), s["coords"])
    rows = ops.costvol_gather(nhwc, d["aff"], (D, D, D), s["voxel_size"], s["sc"]["partial_vol_origin"], cnt, coords)
    close(rows, s["vol"], what="cost volume rows")
    # row_of_voxel is the inverse map
    lin = (s["coords"][:, 0].long() * D + s["coords"][:, 1].long()) * D + s["coords"][:, 2].long()
    assert torch.equal(row.cpu()[lin], torch.arange(n, dtype=torch.int32))
    assert int((row.cpu() >= 0).sum()) == n


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_sparse_cnn_and_scatter(dev, ops, precision):
    """f16x3: convolutions on the matrix cores (csrc/sparse_mfma.hip); fp32: thread-per-row VALU kernel.  Same tolerance."""
    s = small_scene()
    D = s["D"]
    costreg = importlib.import_module("one-2-3-45_amd.costreg")
    net = costreg.CostRegNet(s["costreg_sd"], dev, precision=precision)
    coords = s["coords"].to(dev)
    lin = (s["coords"][:, 0].long() * D + s["coords"][:, 1].long()) * D + s["coords"][:, 2].long()
    grid0 = torch.full((D ** 3,), -1, dtype=torch.int32)
    grid0[lin] = torch.arange(len(lin), dtype=torch.int32)
    out = net.forward(s["vol"].to(dev), coords, grid0.to(dev), (D, D, D))
    for lv, co in zip(s["levels"], net.levels):           # coarse coordinate sets, in torch.unique order
        assert torch.equal(co.cpu()[:, :3].long(), lv.xyz)
    close(out, s["rows16"], rel=1e-4, what="sparse CNN output")
    cl, cf, mask = ops.scatter_dense(out, grid0.to(dev), (D, D, D))
    assert torch.equal(mask.cpu(), s["mask"])
    close(cf, s["dense"], rel=1e-4, what="dense volume")
    assert torch.equal(cl.cpu(), cf[0].permute(1, 2, 3, 0).cpu())


@pytest.mark.parametrize("cin,cout,mode", [(32, 16, 0), (16, 32, 1), (64, 64, 0), (64, 32, 2), (48, 16, 0), (32, 64, 1), (16, 16, 2)])
def test_sparse_conv_x3_single_layer(dev, ops, cin, cout, mode):
    """One layer of csrc/sparse_mfma.hip (and of the fp32 VALU kernel) against the ORACLE's sparse convolution (oracle.sparse_conv over oracle.build_kmap,
    torchsparse v1.4.0 semantics) on a random sparse set incl. a ragged last tile -- every (cin, cout, mode) the two lod networks use, (48, 16) = the lod-1
    input width included."""
    Wn = importlib.import_module("one-2-3-45_amd.weights")
    rng = np.random.default_rng(cin * 100 + cout + mode)
    D = 24
    occ = rng.random((D, D, D)) < 0.3
    xyz = np.argwhere(occ).astype(np.int32)                             # x-major order
    coords = torch.from_numpy(np.ascontiguousarray(np.concatenate([xyz, np.zeros((len(xyz), 1), np.int32)], 1))).to(dev).contiguous()
    grid0 = ops.build_index_grid(coords, 1, (D, D, D))
    g1, co1, n1, cells1 = ops.sparse_downsample(coords, 1, (D, D, D))
    K = torch.from_numpy(rng.normal(0, 0.2, (27, cin, cout)).astype(np.float32)).to(dev)
    blob = torch.from_numpy(Wn.pack_sparse_conv_x3(K)).to(dev)
    assert blob.numel() == ops._lib.lib().o2345_sparse_conv_x3_blob_floats(cin, cout)
    if mode == 0:
        x = torch.from_numpy(rng.normal(0, 1, (len(xyz), cin)).astype(np.float32)).to(dev)
        args = (x, grid0, (D, D, D), coords, 1)
    elif mode == 1:
        x = torch.from_numpy(rng.normal(0, 1, (len(xyz), cin)).astype(np.float32)).to(dev)
        args = (x, grid0, (D, D, D), co1, 2)
    else:
        x = torch.from_numpy(rng.normal(0, 1, (n1, cin)).astype(np.float32)).to(dev)
        args = (x, g1, cells1, coords, 1)
    ref = ops.sparse_conv3d(mode, *args, K)
    got = ops.sparse_conv3d_x3(mode, *args, blob, cout, identity_rows=(mode == 0))       # mode 0: the level's own list -> the brick kernel for 32 -> 16
    close(got, ref, rel=2e-5, what=f"sparse conv x3 {cin}->{cout} mode {mode}")
    if mode == 0:
        # without the caller's identity-row guarantee the gather form runs and honours ANY out_coords: a reversed subset of the sites
        sub = torch.arange(len(xyz) - 1, -1, -3, device=dev)
        got_sub = ops.sparse_conv3d_x3(0, x, grid0, (D, D, D), coords[sub].contiguous(), 1, blob, cout)
        close(got_sub, ref[sub], rel=2e-5, what=f"sparse conv x3 {cin}->{cout} on a reordered subset of the output sites")
        with pytest.raises(ValueError):
            ops.sparse_conv3d_x3(0, x, grid0, (D, D, D), coords[sub].contiguous(), 1, blob, cout, identity_rows=True)
    # the oracle: level objects + kernel maps of oracle/recon.py (down: outputs on the coarse level; up: the down map with roles swapped)
    L0 = O.SparseLevel(torch.from_numpy(xyz).long(), 1)
    L1 = O.downsample_coords(L0)
    assert torch.equal(L1.xyz, co1[:, :3].cpu().long())
    kmap = O.build_kmap(L0, L0) if mode == 0 else O.build_kmap(L0, L1)
    want = O.sparse_conv(x.cpu(), kmap, K.cpu(), transposed=(mode == 2), n_out=len(xyz) if mode == 2 else None)
    close(got, want, rel=2e-5, what=f"sparse conv x3 vs oracle {cin}->{cout} mode {mode}")
    close(ref, want, rel=2e-5, what=f"sparse conv fp32 vs oracle {cin}->{cout} mode {mode}")


def test_bn_and_abn(dev, ops):
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(0.3, 2.0, (5000, 32)).astype(np.float32))
    g = torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)); b = torch.from_numpy(rng.normal(0, 0.2, 32).astype(np.float32))
    y = ops.bn_act_rows(x.to(dev), g.to(dev), b.to(dev))
    close(y, O.bn_relu_rows(x, g, b), what="bn+relu rows")
    xi = torch.from_numpy(rng.normal(-0.2, 1.5, (3, 16, 40, 36)).astype(np.float32))
    g = -g[:16]; b = b[:16]
    y1, y2 = ops.abn_nchw(xi.to(dev), g.to(dev), b.to(dev), want_nhwc=True)
    ref = O.abn_train(xi, g, b)
    close(y1, ref, what="abn nchw")
    close(y2, ref.permute(0, 2, 3, 1), what="abn nhwc")
    for C in (8, 32):                                        # FeatureNet's other widths (HW not a multiple of the 64-pixel tile)
        xc = torch.from_numpy(rng.normal(0.1, 1.2, (2, C, 23, 19)).astype(np.float32))
        gc = torch.from_numpy(rng.uniform(-1.5, 1.5, C).astype(np.float32)); bc = torch.from_numpy(rng.normal(0, 0.2, C).astype(np.float32))
        y1, y2 = ops.abn_nchw(xc.to(dev), gc.to(dev), bc.to(dev), want_nhwc=True)
        ref = O.abn_train(xc, gc, bc)
        close(y1, ref, what=f"abn nchw C={C}")
        close(y2, ref.permute(0, 2, 3, 1), what=f"abn nhwc C={C}")


@pytest.mark.parametrize("R", [33, 64])
def test_sdf_lattice_with_tabulated_layer0(dev, ops, R):
    """extract_fields with layer 0 of the SDF network read from per-axis tables (o2345_sdf_grid_x3) against the oracle and against the kernel that
    evaluates the embedding per point; the tables themselves against W0 . PE + b0 of the oracle's embedding."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    W = sdfW_t(s["sdfW"])
    Wn = pkg.weights
    axes, bias = Wn.sdf_grid_tables(s["sdfW"], R)
    lin = torch.linspace(-1, 1, R)
    order = np.array([Wn.neuron_of(nb, r, h) for h in (0, 1) for nb in range(4) for r in range(16)])
    rng = np.random.default_rng(R)
    idx = torch.from_numpy(rng.integers(0, R, (500, 3)))
    pe = O.embed(torch.stack([lin[idx[:, 0]], lin[idx[:, 1]], lin[idx[:, 2]]], -1), 6).double()
    ref = (pe @ W["w0"].double().T + W["b0"].double())[:, order] * Wn.SOFTPLUS_SCALE   # W0 . PE(x, y, z) + b0 in the kernels' lane order and t domain (x 100 / ln 2)
    got = (torch.from_numpy(axes[0])[idx[:, 0]].double() + torch.from_numpy(axes[1])[idx[:, 1]].double() + torch.from_numpy(axes[2])[idx[:, 2]].double()
           + torch.from_numpy(bias).double())
    assert float((got - ref).abs().max()) < 5e-6 * Wn.SOFTPLUS_SCALE
    tabs = ops.sdf_grid_tables(torch.from_numpy(axes).to(dev), torch.from_numpy(bias).to(dev))
    assert float((tabs[0].view(R, R, 128)[idx[:, 0], idx[:, 1]].cpu().double() + tabs[1][idx[:, 2]].cpu().double() - ref).abs().max()) < 5e-6 * Wn.SOFTPLUS_SCALE
    u_tab = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], None, variant=0, grid_R=R, sign=-1.0, precision="f16x3", grid_tables=tabs)["sdf"]
    u_pts = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], None, variant=0, grid_R=R, sign=-1.0, precision="f16x3")["sdf"]
    ref = O.sdf_grid(s["dense"][0], W, R)
    close(u_tab.view(R, R, R), ref, rel=2e-5, what="tabulated lattice vs oracle")
    close(u_tab, u_pts, rel=2e-5, what="tabulated vs per-point embedding")
    with pytest.raises(ValueError, match="another resolution"):
        ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], None, variant=0, grid_R=R + 1, sign=-1.0, precision="f16x3", grid_tables=tabs)


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
@pytest.mark.parametrize("V", [4, 5])
def test_color_from_materialised_features(dev, ops, prec, V):
    """o2345_color_from_features = GeneralRenderingNetwork.forward on the reference's own four tensors (any Projector's output): against the oracle
    network on the same tensors, and equal to the fused Projector + network kernel on the points those tensors were projected from."""
    s = small_scene(V=V, HW=40, D=16) if V != 4 else small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    rng = np.random.default_rng(V)
    pts = torch.from_numpy(rng.uniform(-0.9, 0.9, (1501, 3)).astype(np.float32))
    pts[:20] = torch.tensor([1.5, 0.0, 0.0])
    RW = color_t(s["color_sd"])
    Kt, w2c = torch.from_numpy(sc["intrinsics"]), torch.from_numpy(sc["w2cs"])
    fm, im = torch.from_numpy(s["fmaps"]), torch.from_numpy(sc["images"])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy())
    geo, rf, rd, vm = O.projector(pts, s["dense"][0], s["mask"][0, 0], fm, im, w2c, Kt, (s["W"], s["H"]), query_cam=qcam)
    rgb_ref, nv_ref = O.rendering_network(RW, geo, rf, rd, vm)
    x3 = prec == "f16x3"
    blob = d["color_x3_blob"] if x3 else d["color_mfma_blob"]
    rgb, nv = ops.color_from_features(blob, geo.to(dev), rf.contiguous().to(dev), rd.contiguous().to(dev), vm.float().to(dev), x3=x3)
    assert torch.equal(nv.cpu().float(), nv_ref)
    close(rgb, rgb_ref, rel=1e-4, what="colour network on materialised tensors")
    fused, _ = ops.color_points(blob, d["vol_cl"], d["maskvol"], d["cmaps"], d["proj"], d["cam_pos"], pts.to(dev), query_cam=qcam.to(dev), mfma="x3" if x3 else True)
    close(rgb, fused, rel=2e-5, what="materialised vs fused inputs")
    with pytest.raises(ValueError, match="expected geometry_feat"):
        ops.color_from_features(blob, geo.to(dev), rf[..., :58].contiguous().to(dev), rd.contiguous().to(dev), vm.float().to(dev), x3=x3)
    # the materialising Projector (o2345_project_features) reproduces the oracle projector's four tensors, and feeds the network kernel
    g2, rf2, rd2, m2 = ops.project_features(d["vol_cl"], d["maskvol"], d["cmaps"], d["proj"], d["cam_pos"], pts.to(dev), query_cam=qcam.to(dev))
    assert torch.equal(m2.cpu() > 0, vm)
    close(g2, geo, rel=2e-5, what="geometry_feat"); close(rf2, rf, rel=2e-5, what="rgb_feat"); close(rd2, rd, rel=2e-5, what="ray_diff")
    rgb3, nv3 = ops.color_from_features(blob, g2, rf2, rd2, m2, x3=x3)
    assert torch.equal(nv3.cpu().float(), nv_ref)
    close(rgb3, fused, rel=2e-5, what="materialised by the HIP projector vs fused")
    nrm = torch.from_numpy(rng.normal(0, 1, (pts.shape[0], 3)).astype(np.float32))
    g4, rf4, rd4, m4 = ops.project_features(d["vol_cl"], d["maskvol"], d["cmaps"], d["proj"], d["cam_pos"], pts.to(dev), normals=nrm.to(dev))
    _, _, rd_ref, _ = O.projector(pts, s["dense"][0], s["mask"][0, 0], fm, im, w2c, Kt, (s["W"], s["H"]), normals=torch.nn.functional.normalize(nrm, p=2, dim=-1, eps=1e-6))
    close(rd4, rd_ref, rel=2e-5, what="ray_diff (view-independent)")


CONV_SHAPES = [(3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1), (32, 32, 1, 1), (32, 16, 3, 1), (32, 8, 3, 1),
               (56, 16, 3, 1), (56, 8, 3, 1)]


@pytest.mark.parametrize("cin,cout,k,stride", CONV_SHAPES)
@pytest.mark.parametrize("hw", [(40, 36), (21, 67)])
@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_conv2d_vs_torch(dev, ops, cin, cout, k, stride, hw, prec):
    """csrc/convnet.hip (every FeatureNet / compress-layer shape; ragged tiles, odd sizes; matrix-core form and strict fp32 VALU form) against
    ATen's fp32 conv2d on the CPU: plain convolution with bias, then the fused form -- producer ABN applied on load, batch statistics of the
    output reduced into this layer's (scale | shift)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(cin * 100 + cout + k)
    V = 3
    x = torch.from_numpy(rng.normal(0.2, 1.0, (V, cin) + hw).astype(np.float32))
    w = torch.from_numpy((rng.normal(0, 1.0, (cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.3, cout).astype(np.float32))
    y, ss = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), stride, precision=prec)
    assert ss is None
    ref = F.conv2d(x, w, b, stride, k // 2)
    assert y.shape == ref.shape
    close(y, ref, rel=1e-5, what="conv2d + bias")
    # fused form
    in_ss = torch.from_numpy(np.concatenate([rng.uniform(-1.5, 1.5, cin), rng.normal(0, 0.3, cin)]).astype(np.float32))
    gamma = torch.from_numpy(rng.uniform(-1.5, 1.5, cout).astype(np.float32)); beta = torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32))
    y, ss = ops.conv2d(x.to(dev), w.to(dev), None, stride, in_ss.to(dev), 0.01, bn=(gamma.to(dev), beta.to(dev), 1e-5, True), precision=prec)
    t = x * in_ss[:cin].view(1, -1, 1, 1) + in_ss[cin:].view(1, -1, 1, 1)
    xa = torch.where(t >= 0, t, t * 0.01)
    ref = F.conv2d(xa, w, None, stride, k // 2)
    close(y, ref, rel=1e-5, what="conv2d of the activated input")
    rd = ref.double()
    mean, var = rd.mean((0, 2, 3)), rd.var((0, 2, 3), unbiased=False)
    scale = (gamma.double().abs() + 1e-5) / torch.sqrt(var + 1e-5)
    close(ss[:cout], scale.float(), rel=2e-5, what="ABN scale from the fused statistics")
    close(ss[cout:], (beta.double() - mean * scale).float(), rel=2e-5, what="ABN shift from the fused statistics")
    if cout in (8, 16, 32):
        y1, y2 = ops.scale_shift_act(y, ss, 0.01, want_nchw=True, want_nhwc=True)
        close(y1, O.abn_train(ref, gamma, beta), rel=2e-5, what="conv + ABN")
        assert torch.equal(y2, y1.permute(0, 2, 3, 1))


@pytest.mark.parametrize("cin,cout,k,stride", [(3, 8, 3, 1), (8, 16, 5, 2), (32, 32, 1, 1), (56, 16, 3, 1)])
@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_conv2d_large_maps(dev, ops, cin, cout, k, stride, prec):
    """The launch variants chosen for large maps (2 and 4 pixels per thread in the fp32 form; many tiles and partial-sum blocks in both forms):
    3 views of 260 x 516 and of 130 x 258 against ATen's fp32 conv2d, incl. the batch statistics."""
    import torch.nn.functional as F
    rng = np.random.default_rng(cin + cout)
    for hw in ((260, 516), (130, 258)):
        x = torch.from_numpy(rng.normal(0.1, 1.0, (3, cin) + hw).astype(np.float32))
        w = torch.from_numpy((rng.normal(0, 1.0, (cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
        gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)); beta = torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32))
        y, ss = ops.conv2d(x.to(dev), w.to(dev), None, stride, bn=(gamma.to(dev), beta.to(dev), 1e-5, True), precision=prec)
        ref = F.conv2d(x, w, None, stride, k // 2)
        close(y, ref, rel=1e-5, what=f"conv2d {hw}")
        rd = ref.double()
        scale = (gamma.double().abs() + 1e-5) / torch.sqrt(rd.var((0, 2, 3), unbiased=False) + 1e-5)
        close(ss[:cout], scale.float(), rel=2e-5, what="scale"); close(ss[cout:], (beta.double() - rd.mean((0, 2, 3)) * scale).float(), rel=2e-5, what="shift")


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_conv2d_channel_last_input(dev, ops, prec):
    """The compress layer reading its 56 input channels out of the channel-last colour map [V,H,W,64] (offset 3) == the same convolution on the
    channel-first tensor, bit for bit (same kernel, other addressing), and the pyramid kernel's channel-last output carries those features."""
    rng = np.random.default_rng(5)
    V, H, W = 3, 24, 44
    cm = torch.from_numpy(rng.normal(0, 1, (V, H, W, 64)).astype(np.float32)).to(dev)
    x = cm[..., 3:59].permute(0, 3, 1, 2).contiguous()
    w = torch.from_numpy((rng.normal(0, 1, (16, 56, 3, 3)) / 22).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.uniform(0.5, 1.5, 16).astype(np.float32)).to(dev); b = torch.from_numpy(rng.normal(0, 0.2, 16).astype(np.float32)).to(dev)
    y0, ss0 = ops.conv2d(x, w, bn=(g, b, 1e-5, True), precision=prec)
    y1, ss1 = ops.conv2d(cm, w, bn=(g, b, 1e-5, True), precision=prec, nhwc_offset=3)
    assert torch.equal(y0, y1) and torch.equal(ss0, ss1)
    with pytest.raises(ValueError, match="do not fit"):
        ops.conv2d(cm, w, precision=prec, nhwc_offset=9)
    # a 32-channel layer reading channels 5..37 of the same map (the whole-tile kernel's channel-last addressing)
    w32 = torch.from_numpy((rng.normal(0, 1, (16, 32, 3, 3)) / 17).astype(np.float32)).to(dev)
    y2, _ = ops.conv2d(cm[..., 5:37].permute(0, 3, 1, 2).contiguous(), w32, precision=prec)
    y3, _ = ops.conv2d(cm, w32, precision=prec, nhwc_offset=5)
    assert torch.equal(y2, y3)
    # k_pyramid_pack: channel-last map alone (no channel-first tensor) == the map written together with it
    f2 = torch.from_numpy(rng.normal(0, 1, (V, 32, H // 4, W // 4)).astype(np.float32)).to(dev)
    s1 = torch.from_numpy(rng.normal(0, 1, (V, 16, H // 2, W // 2)).astype(np.float32)).to(dev)
    s0 = torch.from_numpy(rng.normal(0, 1, (V, 8, H, W)).astype(np.float32)).to(dev)
    rgb = torch.from_numpy(rng.uniform(0, 1, (V, 3, H, W)).astype(np.float32)).to(dev)
    fm, cm2 = ops.pyramid_pack(f2, s1, s0, rgb)
    none, cm3 = ops.pyramid_pack(f2, s1, s0, rgb, want_nchw=False)
    assert none is None and torch.equal(cm2, cm3) and torch.equal(cm2[..., 3:59].permute(0, 3, 1, 2), fm) and torch.equal(cm2[..., :3].permute(0, 3, 1, 2), rgb)


def test_conv2d_rejects_other_shapes(dev, ops):
    with pytest.raises(Exception, match="no kernel for"):
        ops.conv2d(torch.zeros(1, 4, 8, 8, device=dev), torch.zeros(8, 4, 3, 3, device=dev), precision="fp32")
    with pytest.raises(Exception, match="no kernel for"):
        ops.conv2d(torch.zeros(1, 16, 8, 8, device=dev), torch.zeros(8, 16, 7, 7, device=dev), precision="f16x3")
    with pytest.raises(ValueError, match="CUDA"):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(8, 3, 3, 3))


def _pts(n, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-1.1, 1.1, (n, 3)).astype(np.float32)
    p[:32] = np.array([[-1.0, 0.1, 0.2]], np.float32); p[32:48] = 1.0; p[48:64] = 1.03
    return torch.from_numpy(p)


@pytest.mark.parametrize("n", [1, 31, 1000, 20011])
def test_sdf_mlp(dev, ops, n):
    """The exact fp32 MFMA kernel, all three variants (the default precision is f16x3: test_sdf_mlp_x3)."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    W = sdfW_t(s["sdfW"])
    pts = _pts(n)
    y, lat = O.sdf(pts, s["dense"][0], W)
    import functools
    ops = type("fp32ops", (), {"sdf_mlp": staticmethod(functools.partial(ops.sdf_mlp, precision="fp32"))})
    r0 = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=0, want_lat=True)
    close(r0["sdf"], y[:, 0], what="sdf (variant 0)")
    close(r0["lat"], lat, what="latent")
    r1 = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=1)
    close(r1["sdf"], y[:, 0], what="sdf (variant 1)")
    close(r1["feat"], y, what="128 features")
    r2 = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=2)
    close(r2["sdf"], y[:, 0], what="sdf (variant 2)")
    close(r2["grad"], O.sdf_grad(pts, s["dense"][0], W), rel=1e-4, what="gradient")




@pytest.mark.parametrize("n", [31, 20011])
def test_sdf_mlp_x3(dev, ops, n):
    """Split-f16 ("f16x3") forward kernel: same tolerance as the exact fp32 MFMA kernel (default `close`), and within 2e-6 of it."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    W = sdfW_t(s["sdfW"])
    pts = _pts(n)
    y, _ = O.sdf(pts, s["dense"][0], W)
    r = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=0, precision="f16x3")
    close(r["sdf"], y[:, 0], what="sdf (f16x3)")
    exact = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=0, precision="fp32")["sdf"]
    err = float((r["sdf"] - exact).abs().max())
    assert err <= 2e-6 * max(1.0, float(exact.abs().max())), err
    g = O.sdf_grad(pts, s["dense"][0], W)
    r2 = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=2, precision="f16x3")
    close(r2["sdf"], y[:, 0], what="sdf (f16x3 gradient kernel)")
    close(r2["grad"], g, rel=1e-4, what="gradient (f16x3)")                  # same tolerance as the fp32 kernel
    g32 = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=2, precision="fp32")["grad"]
    print(f"f16x3 grad: vs oracle {float((r2['grad'].cpu() - g).abs().max()):.3e}, fp32 kernel vs oracle "
          f"{float((g32.cpu() - g).abs().max()):.3e} (max|grad| {float(g.abs().max()):.3f})")
    print(f"f16x3 vs fp32 MFMA: max|diff| {err:.3e}; vs oracle {float((r['sdf'].cpu() - y[:, 0]).abs().max()):.3e}; "
          f"fp32 MFMA vs oracle {float((exact.cpu() - y[:, 0]).abs().max()):.3e}")


def test_sdf_mlp_x3_large_activations(dev, ops):
    """The split form's domain: operands up to +-65504.  A latent volume scaled by 100 (latents up to ~+-300, pre-activations of
    the same order) must still agree with the exact fp32 MFMA kernel to fp32-class RELATIVE accuracy, and stay finite."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    pts = _pts(20011, seed=5) * 1.7
    vol = (d["vol_cl"] * 100.0).contiguous()
    a = ops.sdf_mlp(d["sdf_blob"], vol, pts.to(dev), variant=2, precision="f16x3")
    b = ops.sdf_mlp(d["sdf_blob"], vol, pts.to(dev), variant=2, precision="fp32")
    assert bool(torch.isfinite(a["sdf"]).all()) and bool(torch.isfinite(a["grad"]).all())
    scale = float(b["sdf"].abs().max())
    assert scale > 5.0                                                   # the scaling did reach the network
    assert float((a["sdf"] - b["sdf"]).abs().max()) <= 2e-5 * scale
    assert float((a["grad"] - b["grad"]).abs().max()) <= 1e-4 * float(b["grad"].abs().max())


def test_sdf_mlp_indexed_and_grid(dev, ops):
    s = small_scene()
    d = dev_scene(s, dev, ops)
    W = sdfW_t(s["sdfW"])
    pts = _pts(5000, 3)
    idx = torch.from_numpy(np.random.default_rng(1).permutation(5000)[:1234].astype(np.int32))
    out = {"sdf": torch.full((5000,), 100.0, device=dev)}
    n_dev = torch.tensor([1000], dtype=torch.int32, device=dev)
    ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], pts.to(dev), variant=0, index=idx.to(dev), n_dev=n_dev, out=out)
    ref = torch.full((5000,), 100.0)
    ref[idx[:1000].long()] = O.sdf(pts[idx[:1000].long()], s["dense"][0], W)[0][:, 0]
    close(out["sdf"], ref, what="indexed sdf with device count")
    R = 24
    g = ops.sdf_mlp(d["sdf_blob"], d["vol_cl"], None, variant=0, grid_R=R, sign=-1.0)
    close(g["sdf"].reshape(R, R, R), O.sdf_grid(s["dense"][0], W, R), what="sdf grid (u = -sdf)")


def test_marching_cubes(dev, ops):
    s = small_scene()
    W = sdfW_t(s["sdfW"])
    for R in (24, 37):
        u = O.sdf_grid(s["dense"][0], W, R).contiguous()
        v_ref, t_ref = omc.marching_cubes(u.numpy(), 0.0)
        assert len(t_ref) > 100
        v, t = ops.marching_cubes(u.to(dev), 0.0)
        assert torch.equal(t.cpu(), torch.from_numpy(t_ref))
        assert np.abs(v.cpu().numpy() - v_ref).max() < 1e-12
    # a surface that touches the boundary + an empty field
    lin = np.linspace(-1, 1, 20, dtype=np.float32)
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing="ij")
    u = (1.2 - np.sqrt(X ** 2 + Y ** 2 + Z ** 2)).astype(np.float32)
    v_ref, t_ref = omc.marching_cubes(u, 0.0)
    v, t = ops.marching_cubes(torch.from_numpy(u).to(dev), 0.0)
    assert torch.equal(t.cpu(), torch.from_numpy(t_ref)) and np.abs(v.cpu().numpy() - v_ref).max() < 1e-12
    v, t = ops.marching_cubes(torch.full((8, 9, 10), -1.0, device=dev), 0.0)
    assert v.shape[0] == 0 and t.shape[0] == 0


@pytest.mark.parametrize("mfma,V,kernel", [(True, 4, "tiles"), (True, 8, "tiles"), (True, 12, "tiles"), (True, 32, "tiles"),
                                           ("x3", 4, "tiles"), ("x3", 8, "tiles"), ("x3", 12, "tiles"), ("x3", 32, "tiles"),
                                           (True, 4, "pts"), (True, 5, "pts"), (True, 8, "pts"), (True, 32, "pts"),
                                           ("x3", 4, "pts"), ("x3", 5, "pts"), ("x3", 8, "pts"), ("x3", 12, "pts"), ("x3", 32, "pts")])
def test_color_points(dev, ops, mfma, V, kernel, lib_instance):
    """The colour kernel in both numerical forms: "pts" = k_color_pts (columns = points, any V: the product kernel); "tiles" = k_color_mfma (columns =
    (point, view) pairs; V = 4 / 8 / 12 / 32 exercise its G = 4 / 8 / 16 / 32 lane groups incl. padded views for V = 12) -- a TEST-ONLY build variant
    since round 4 (libo2345_hip_tiles.so, -DO2345_TILES_KERNEL, selected by O2345_COLOR_KERNEL=tiles), kept as the independent second implementation."""
    if kernel == "tiles":
        assert b"color_tiles=1" in lib_instance({"O2345_COLOR_KERNEL": "tiles"}, variant="tiles").o2345_knobs()
    s = small_scene(V=V, HW=40, D=16) if V != 4 else small_scene()
    d = dev_scene(s, dev, ops)
    blob = d["color_x3_blob"] if mfma == "x3" else d["color_mfma_blob"]
    sc = s["sc"]
    rng = np.random.default_rng(2)
    pts = torch.from_numpy(rng.uniform(-0.9, 0.9, (3000, 3)).astype(np.float32))
    pts[:20] = torch.tensor([1.5, 0.0, 0.0])
    RW = color_t(s["color_sd"])
    Kt, w2c = torch.from_numpy(sc["intrinsics"]), torch.from_numpy(sc["w2cs"])
    fm, im = torch.from_numpy(s["fmaps"]), torch.from_numpy(sc["images"])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy())
    geo, rf, rd, vm = O.projector(pts, s["dense"][0], s["mask"][0, 0], fm, im, w2c, Kt, (s["W"], s["H"]), query_cam=qcam)
    rgb_ref, nv_ref = O.rendering_network(RW, geo, rf, rd, vm)
    rgb, nv = ops.color_points(blob, d["vol_cl"], d["maskvol"], d["cmaps"], d["proj"], d["cam_pos"], pts.to(dev),
                               query_cam=qcam.to(dev), mfma=mfma)
    assert torch.equal(nv.cpu().float(), nv_ref)
    close(rgb, rgb_ref, rel=1e-4, what="blended colour")
    assert torch.equal(ops.view_count(pts.to(dev), d["maskvol"], s["D"], d["proj"], s["V"], s["H"], s["W"]).cpu().float(), nv_ref)
    # view-independent variant: direction = normalised SDF gradient
    g = O.sdf_grad(pts, s["dense"][0], sdfW_t(s["sdfW"]))
    nrm = torch.nn.functional.normalize(g, p=2, dim=-1, eps=1e-6)
    geo, rf, rd, vm = O.projector(pts, s["dense"][0], s["mask"][0, 0], fm, im, w2c, Kt, (s["W"], s["H"]), normals=nrm)
    rgb_ref, _ = O.rendering_network(RW, geo, rf, rd, vm)
    rgb, _ = ops.color_points(blob, d["vol_cl"], d["maskvol"], d["cmaps"], d["proj"], d["cam_pos"], pts.to(dev), normals=g.to(dev), mfma=mfma)
    close(rgb, rgb_ref, rel=1e-4, what="vertex colour")


@pytest.mark.parametrize("nrays,precision", [(7, "fp32"), (300, "fp32"), (7, "f16x3"), (300, "f16x3")])
def test_render(dev, ops, nrays, precision):
    """render() against the oracle in its DEFAULT (ATen) mode, every ray and every sample accounted for (no quantiles):
      (1) the sampler stage driven with the oracle's own per-round inputs -> all new depths agree;
      (2) everything downstream of the sampler, evaluated by the oracle on the HIP path's OWN sample lists -> all rays agree tightly
          (weights, SDF, gradients, colour, depth, depth variance, colour mask);
      (3) end to end, the hierarchical sampler propagates fp32-class SDF differences (sigmoid slopes up to 512, inverse CDF): rendered
          quantities of all rays within 1e-3, sample lists within one coarse section, rays with coinciding sample lists tight."""
    import fullsize_util as FU
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    ro, rd = rays_for(s, nrays, seed=nrays)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    variance = torch.tensor(0.2)
    inv_s = float(torch.exp(variance * 10.0).clip(1e-6, 1e6))
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy())
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    scene["sdf_precision"] = scene["color_precision"] = precision
    out = ops.render_rays(scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), near, far, 64, 64, inv_s, 1.0, 1.0,
                          qcam.to(dev), want_z=True)
    a = dict(volume=s["dense"][0], maskvol=s["mask"][0, 0], W=sdfW_t(s["sdfW"]), RW=color_t(s["color_sd"]), variance=variance,
             feat_maps=torch.from_numpy(s["fmaps"]), color_maps=torch.from_numpy(sc["images"]), w2cs=torch.from_numpy(sc["w2cs"]),
             K=torch.from_numpy(sc["intrinsics"]), img_wh=(s["W"], s["H"]), query_c2w=torch.from_numpy(sc["query_c2w"]))
    tro, trd = torch.from_numpy(ro), torch.from_numpy(rd)
    ref = O.render(tro, trd, torch.tensor(near), torch.tensor(far), a["volume"], a["maskvol"], a["W"], a["RW"], variance, a["feat_maps"],
                   a["color_maps"], a["w2cs"], a["K"], a["img_wh"], a["query_c2w"])
    assert ref["weights_sum"].max() > 0.5, "test scene must contain a surface"
    # (1) sampler stage with identical inputs (one call for all rays: same per-call quirk semantics as the render call above)
    dz, pdf, width = FU.sampler_stage_check(ops, dev, tro, trd, near, far, a, d["maskvol"].reshape(-1), s["D"], chunk=nrays)
    assert float(dz.max()) < 2e-4 and float((dz / width.clamp(min=1e-9)).max()) < 5e-3, (float(dz.max()), float((dz / width.clamp(min=1e-9)).max()))
    # (2) downstream of the sampler on the HIP path's own sample lists: ALL rays, tight
    z_gpu = out["z_vals"].t().cpu().contiguous()
    core = FU.oracle_core(a, tro, trd, near, far, z_gpu, chunk=nrays)
    close(out["weights"].t().cpu(), core["weights"], rel=2e-5, what="weights")
    close(out["sdf"].t().cpu(), core["sdf"], rel=2e-5, what="sdf")
    close(out["grad"].permute(1, 0, 2).cpu(), core["gradients"], rel=1e-4, what="gradients")
    close(out["color"], core["color_fine"], rel=3e-5, what="colour")
    close(out["depth"][:, None], core["depth"], rel=2e-5, what="depth")
    close(out["weights_sum"][:, None], core["weights_sum"], rel=2e-5, what="weights_sum")
    close(out["depth_var"][:, None], core["depth_variance"], rel=2e-5, what="depth variance")
    assert torch.equal(out["color_mask"].cpu().bool(), core["color_fine_mask"][:, 0])
    # (3) end to end against the oracle's own render
    z_ref = ref["z_vals"]
    spacing = (far - near) / 63
    zerr = (z_gpu - z_ref).abs().max(1).values
    assert zerr.max() <= 1.001 * spacing, f"sample lists differ by {zerr.max():.3e} (coarse spacing {spacing:.3e})"
    same = zerr < 1e-6
    if same.any():
        close(out["color"].cpu()[same], ref["color_fine"][same], rel=3e-5, what="colour (coinciding sample lists)")
    close(out["color"], ref["color_fine"], rel=1e-3, what="colour")
    close(out["depth"][:, None], ref["depth"], rel=1e-3, what="depth")
    close(out["weights_sum"][:, None], ref["weights_sum"], rel=1e-3, what="weights_sum")
    close(out["depth_var"][:, None], ref["depth_variance"], rel=1e-3, what="depth variance")
    bad = torch.nonzero((out["color"].cpu() - ref["color_fine"]).abs().max(1).values > 1e-4)[:, 0]
    print(f"[{precision}, {nrays} rays] colour deviates by > 1e-4 on rays {bad.tolist()} (sample lists differ by {zerr[bad].tolist()})")
    assert bool((zerr[bad] > 1e-6).all())                      # attributable to the sample lists, since (2) is tight for all rays


def _oracle_args(s, sdf_shift=0.0):
    sc = s["sc"]
    W = sdfW_t(s["sdfW"])
    if sdf_shift:
        W = {k: v.clone() for k, v in W.items()}
        W["b2"][0] += sdf_shift
    return dict(volume=s["dense"][0], maskvol=s["mask"][0, 0], W=W, RW=color_t(s["color_sd"]), feat_maps=torch.from_numpy(s["fmaps"]),
                color_maps=torch.from_numpy(sc["images"]), w2cs=torch.from_numpy(sc["w2cs"]), K=torch.from_numpy(sc["intrinsics"]),
                img_wh=(s["W"], s["H"]), query_c2w=torch.from_numpy(sc["query_c2w"]))


@pytest.mark.parametrize("variance,air,bg,shift,precision", [
    (0.2, 0.0, None, 0.0, "f16x3"), (0.2, 0.5, 1.0, 0.0, "f16x3"), (0.45, 0.5, None, 0.0, "f16x3"), (0.45, 1.0, 1.0, 0.0, "f16x3"),
    (0.65, 0.0, 1.0, 0.0, "f16x3"), (0.65, 0.5, None, 0.0, "f16x3"), (0.65, 1.0, None, -0.2, "f16x3"), (0.45, 0.0, 1.0, -0.2, "f16x3"),
    (0.65, 1.0, None, -0.2, "fp32"), (0.45, 0.5, 1.0, 0.0, "fp32")])
def test_render_trained_regime(dev, ops, variance, air, bg, shift, precision):
    """render() in the regime of a TRAINED model: SingleVarianceNetwork far from its init (variance 0.45 / 0.65 -> inv_s = 90 / 665,
    models/fields.py:179-186), the iter_step-driven alpha_inter_ratio of exp_runner_generic_blender_val.py:412-418 (0 / 0.5 / 1),
    background_rgb None / 1.0 (train.use_white_bkgd), and -- shift = -0.2 -- a field whose zero level set every ray crosses (single-sample
    weights up to 1.0 at inv_s = 665).  The three-clause contract of tests/render_check.py, all rays, no quantile thresholds."""
    import render_check as RC
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    a = _oracle_args(s, shift)
    W = {k: np.array(v) for k, v in s["sdfW"].items()}
    W["b2"][0] += shift
    scene = {k: d[k] for k in ("color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    scene["sdf_blob"] = torch.from_numpy(pkg.weights.pack_sdf_blob(W)).to(dev)
    ro, rd = rays_for(s, 96, seed=5)
    res = RC.three_clause(ops, dev, scene, a, torch.from_numpy(ro), torch.from_numpy(rd), float(sc["query_near_far"][0]),
                          float(sc["query_near_far"][1]), variance, air, bg, precision, label="small_scene")
    assert res["rays_hitting_surface"] > 10, res


@pytest.mark.parametrize("variance,air,bg,shift", [(1.0, 1.0, 1.0, -0.2), (1.5, 1.0, None, -0.2), (1.5, 0.5, 1.0, 0.0), (-1.5, 0.5, 1.0, -0.2)])
def test_render_at_the_clip_limits_of_the_variance(dev, ops, variance, air, bg, shift):
    """The ends of SingleVarianceNetwork's range as the renderer uses it (sparse_neus_renderer.py:340: inv_s = exp(10 variance) clipped to [1e-6, 1e6]):
    variance 1.0 -> inv_s 22,026; 1.5 -> the upper clip 1e6 (every section's opacity is a step function of the SDF's sign); -1.5 -> 3e-7, below the lower
    clip 1e-6 (every section transparent but for the reference's +1e-5).  Same three-clause contract as test_render_trained_regime, every output finite."""
    import render_check as RC
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    a = _oracle_args(s, shift)
    W = {k: np.array(v) for k, v in s["sdfW"].items()}
    W["b2"][0] += shift
    scene = {k: d[k] for k in ("color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    scene["sdf_blob"] = torch.from_numpy(pkg.weights.pack_sdf_blob(W)).to(dev)
    ro, rd = rays_for(s, 96, seed=5)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    inv_s = float(np.clip(np.exp(10.0 * variance), 1e-6, 1e6))
    assert inv_s in (1e-6, 1e6) or variance == 1.0
    o = ops.render_rays(scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), near, far, 64, 64, inv_s, air, 0.0 if bg is None else bg,
                        torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev))
    for k in ("color", "depth", "weights", "weights_sum", "weights_max", "depth_var", "cdf", "sdf", "grad"):
        assert bool(torch.isfinite(o[k]).all()), k
    assert float(o["weights_sum"].max()) <= 1.0 + 1e-5 and float(o["weights"].min()) >= 0.0
    if inv_s == 1e-6:
        assert float(o["weights_sum"].max()) < 0.01                       # nothing but the reference's +1e-5 per section
    res = RC.three_clause(ops, dev, scene, a, torch.from_numpy(ro), torch.from_numpy(rd), near, far, variance, air, bg, "f16x3", label="clip_limits")
    if shift < 0 and inv_s > 1:
        assert res["rays_hitting_surface"] > 10, res


def test_ray_finalize_stage_initialises_every_slot(dev, ops):
    """The public stage entry o2345_ray_finalize: the reference's defaults (sdf = 100, gradients = colours = 0, sparse_neus_renderer.py:231) in EVERY
    slot -- occupied ones included, a caller may evaluate only part of the list --, mid points / section lengths / occupancy vs the oracle, and the
    list = exactly the occupied slots."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    ro, rd = rays_for(s, 70, seed=3, center=False)
    S, R = 24, 70
    near, far = float(s["sc"]["query_near_far"][0]), float(s["sc"]["query_near_far"][1])
    z = torch.sort(torch.rand(R, S, generator=torch.Generator().manual_seed(1)) * (far - near) + near, 1).values
    sd = (far - near) / 64
    o = ops.ray_finalize(torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), z.t().contiguous().to(dev), sd, d["maskvol"].reshape(-1), s["D"])
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), sd)], 1)
    mid = z + dists * 0.5
    pts = torch.from_numpy(ro)[:, None] + torch.from_numpy(rd)[:, None] * mid[..., None]
    pm = O.mask_nearest(s["mask"][0, 0], pts.reshape(-1, 3)).reshape(R, S)
    assert torch.equal(o["pm"].t().cpu(), pm) and 0 < int(pm.sum()) < R * S
    close(o["mid_z"].t(), mid, rel=1e-6, what="mid_z"); close(o["dists"].t(), dists, rel=1e-6, what="dists")
    close(o["pts"].permute(1, 0, 2), pts, rel=1e-6, what="mid points")
    assert bool((o["sdf"] == 100).all()) and float(o["grad"].abs().sum()) == 0 and float(o["rgb"].abs().sum()) == 0
    n = int(o["count"])
    assert n == int(pm.sum())
    slots = torch.sort(o["list"][:n].cpu().long()).values
    assert torch.equal(slots, torch.nonzero(pm.t().reshape(-1) > 0)[:, 0])


def test_convolution_precision_is_per_object(dev, ops, monkeypatch):
    """The numerical mode of FeatureNet / the compress layer belongs to the object (SceneWeights(color_precision=...), featurenet.set_precision), not
    only to the global O2345_PRECISION: an fp32 object under the default global mode is bit-equal to the same network under a global fp32 mode."""
    fn = importlib.import_module("one-2-3-45_amd.featurenet")
    config = importlib.import_module("one-2-3-45_amd.config")
    pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
    torch.manual_seed(0)
    net = fn.FeatureNet().to(dev)
    comp = fn.ConvBnReLU(56, 16).to(dev)
    imgs = torch.rand(2, 3, 48, 64, device=dev)
    with torch.no_grad():
        monkeypatch.setattr(config, "PRECISION", "f16x3")
        fn.set_precision(net, "fp32"); fn.set_precision(comp, "fp32")
        a = [t.clone() for t in net(imgs)] + [comp(fn.fused_pyramid(net, imgs)).clone()]
        fn.set_precision(net, None); fn.set_precision(comp, None)
        c = [t.clone() for t in net(imgs)] + [comp(fn.fused_pyramid(net, imgs)).clone()]            # global default: split-f16 matrix cores
        monkeypatch.setattr(config, "PRECISION", "fp32")
        b = [t.clone() for t in net(imgs)] + [comp(fn.fused_pyramid(net, imgs)).clone()]
    for x, y, w in zip(a, b, c):
        assert torch.equal(x, y)
        assert not torch.equal(x, w) and float((x - w).abs().max()) < 5e-5 * max(1.0, float(x.abs().max()))
    monkeypatch.setattr(config, "PRECISION", "f16x3")
    wt = pipeline.SceneWeights(dev, seed=0, color_precision="fp32")
    assert wt.featurenet.precision == "fp32" and wt.compress.precision == "fp32" and all(m.precision == "fp32" for m in wt.featurenet.modules() if isinstance(m, fn.ConvBnReLU))


@pytest.mark.parametrize("V", [5, 8, 32])
def test_view_skipping_is_bit_identical(dev, ops, V, lib_instance):
    """k_color_pts skips the views that see none of a tile's 32 points (wave-uniform).  The claim is BIT-identity with the kernel that evaluates every
    view (O2345_COLOR_SCHED bit 2), for colours and valid-view counts, on a point set that contains every case: points seen by many views, by one, by
    none (all-masked tiles blend every view uniformly), points outside the volume, tiles whose points disagree about a view, a ragged last tile."""
    s = small_scene(V=V, HW=40, D=16) if V != 4 else small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    rng = np.random.default_rng(V)
    # tiles of the render path are 32 neighbouring points: clusters of 32 around random centres (coherent visibility), some tiles of unrelated points
    centres = rng.uniform(-0.95, 0.95, (128, 3)).astype(np.float32)
    centres[:8] = rng.uniform(0.8, 0.99, (8, 3)).astype(np.float32) * np.sign(rng.standard_normal((8, 3))).astype(np.float32)      # corners: few / no views
    pts = (centres[:, None] + rng.normal(0, 2e-3, (128, 32, 3)).astype(np.float32)).reshape(-1, 3)
    pts[64:96] = np.array([1.5, 0.0, 0.0], np.float32)                                                                            # a whole tile outside
    pts[1024:1280] = rng.uniform(-0.95, 0.95, (256, 3)).astype(np.float32)                                                        # incoherent tiles
    pts = np.concatenate([pts, rng.uniform(-0.5, 0.5, (3, 3)).astype(np.float32)])                                               # ragged last tile
    p = torch.from_numpy(pts).to(dev)
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    outs = {}
    for sched in ("4", "0", "10", "14"):
        lib_instance({"O2345_COLOR_SCHED": sched})          # a library instance per knob value: the knobs are read once per instance
        st = ops.color_stats_buffer(dev)                    # caller-owned work counters (no library state)
        outs[sched] = ops.color_points(d["color_x3_blob"], d["vol_cl"], d["maskvol"], d["cmaps"], d["proj"], d["cam_pos"], p, query_cam=qcam, mfma="x3", stats=st)
        outs[sched] += (ops.color_stats_read(st),)
    ref = outs["4"]
    assert ref[2]["pairs_network"] == ref[2]["tiles"] * V                      # bit 2: every (tile, view) pair evaluated
    for k in ("0", "10"):
        assert torch.equal(outs[k][0], ref[0]) and torch.equal(outs[k][1], ref[1]), k
        assert outs[k][2]["pairs_network"] < ref[2]["pairs_network"] and outs[k][2]["pairs_pooling"] < ref[2]["pairs_pooling"]   # something WAS skipped
    assert torch.equal(outs["14"][0], ref[0])
    assert int((ref[1] == 0).sum()) >= 32 and int((ref[1] >= 2).sum()) > 500
    # and against the oracle (colours of the no-visible-view points included)
    Kt, w2c = torch.from_numpy(sc["intrinsics"]), torch.from_numpy(sc["w2cs"])
    geo, rf, rdf, vm = O.projector(torch.from_numpy(pts), s["dense"][0], s["mask"][0, 0], torch.from_numpy(s["fmaps"]), torch.from_numpy(sc["images"]), w2c, Kt,
                                   (s["W"], s["H"]), query_cam=qcam.cpu())
    rgb_ref, nv_ref = O.rendering_network(color_t(s["color_sd"]), geo, rf, rdf, vm)
    assert torch.equal(ref[1].cpu().float(), nv_ref)
    close(outs["10"][0], rgb_ref, rel=1e-4, what="blended colour with view skipping")


@pytest.mark.gpu
@pytest.mark.parametrize("V", [5, 8, 19, 32])
def test_list_sort_by_visibility(pkg, ops, V):
    """o2345_list_sort_by_visibility: a stable permutation of the list, signatures ascending, signatures = 'projects strictly inside view v' (checked against torch on
    the same points, borderline points aside), entries past the device-side count untouched; and the colour kernel's results do not depend on the order."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(V)
    P_ = 300000
    pts = (torch.rand(P_, 3, generator=g) * 2 - 1).to(dev)
    sc = pkg.synth.make_scene(V, image_seed=1)
    pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
    proj, _ = pipeline.camera_terms(torch.from_numpy(sc["intrinsics"]).to(dev), torch.from_numpy(sc["w2cs"]).to(dev))
    proj = proj.float().contiguous()
    keep = torch.rand(P_, generator=g) < 0.7
    index = torch.nonzero(keep).flatten().to(torch.int32).to(dev)                 # ascending slots: stability is visible in the values
    n = index.numel()
    n_valid = n - 1234
    count = torch.tensor([n_valid], dtype=torch.int32, device=dev)
    out, keys = ops.list_sort_by_visibility(pts, index, proj, 256, 256, count=count, want_keys=True)
    o, k = out[:n_valid].long(), keys[:n_valid].long() & 0xFFFFFFFF
    assert torch.equal(torch.sort(o)[0], index[:n_valid].long()), "not a permutation of the valid entries"
    assert bool((k[1:] >= k[:-1]).all()), "signatures not ascending"
    same = k[1:] == k[:-1]
    assert bool((o[1:][same] > o[:-1][same]).all()), "not stable inside a signature"
    p = pts[o]
    ref = torch.zeros_like(k)
    for v in range(V):
        pr = p @ proj[v, :, :3].T + proj[v, :, 3]
        z = pr[:, 2].clamp(min=1e-3)
        ref |= ((pr[:, 0] > 0) & (pr[:, 0] < 255 * z) & (pr[:, 1] > 0) & (pr[:, 1] < 255 * z)).long() << v
    assert float((ref != k).float().mean()) < 1e-3
    assert int(torch.unique(k).numel()) > 3


def test_ray_kernels_streaming_and_lds_forms_are_bit_identical(dev, ops, lib_instance):
    """csrc/render.hip has two forms of every sampler round and of the compositing kernel: sixteen lanes per ray (small batches, e.g. the 512-ray chunks of
    the reference's val loop: only the two scans of a round are serial) and streaming (one lane per ray on the global lists; large batches), selected by
    the ray count (O2345_RAY_STREAM_MIN).  They share render_math.h and must agree BIT FOR BIT: a whole render call in both forms (every output), and the
    up-sampling stage entry on 5,000 rays."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    ro, rd = rays_for(s, 333, seed=11, center=False)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    t_rand = torch.rand(333, 64, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = {}
    for name, thr in (("group", "1000000000"), ("stream", "0")):
        assert ("ray_stream_min=" + thr).encode() in lib_instance({"O2345_RAY_STREAM_MIN": thr}).o2345_knobs()
        outs[name] = ops.render_rays(scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), near, far, 64, 64, 7.4, 0.5, 1.0, qcam, want_z=True,
                                     t_rand=t_rand, want_scalars=True)
    for k, v in outs["group"].items():
        assert torch.equal(v, outs["stream"][k]), k
    assert float(outs["group"]["weights_sum"].max()) > 0.5
    # stage entry, default library: 5,000 rays take the streaming kernel when scratch is handed over, the sixteen-lane kernel without
    lib_instance({})
    R, S = 5000, 80
    g = torch.Generator().manual_seed(5)
    sel = torch.randint(0, ro.shape[0], (R,), generator=g)
    tro, trd = torch.from_numpy(ro)[sel].contiguous().to(dev), torch.from_numpy(rd)[sel].contiguous().to(dev)
    z = torch.sort(torch.rand(S, R, generator=g) * (far - near) + near, 0).values.contiguous().to(dev)
    sdf = ((1.4 - z.cpu()) * (0.5 + torch.rand(S, R, generator=g))).contiguous().to(dev)
    a = ops.ray_upsample(tro, trd, z, sdf, 256.0, d["maskvol"].reshape(-1), s["D"], 16)
    b = ops.ray_upsample(tro, trd, z, sdf, 256.0, d["maskvol"].reshape(-1), s["D"], 16, streaming=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(torch.sort(a[2]).values, torch.sort(b[2]).values) and a[2].numel() > 100


def test_chunked_render_equals_one_call(dev, ops):
    """The reference renders an image in 512-ray chunks (trainer_generic.py:365, 415-416), the fused pipeline in one call.  A ray's result must not depend on
    which rays share its call: 8,192 rays in ONE call (streaming sampler kernels, list grouped by visibility when long enough) == the same rays in sixteen
    512-ray calls (sixteen-lane kernels, emission order), bit for bit, every per-ray and per-sample output -- on rays where the reference's per-CALL quirks
    cannot fire (every chunk has more than one occupied new sample per round and at least one occupied mid-point: asserted)."""
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    ro, rd = rays_for(s, 8192, seed=21, center=True)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    tro, trd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    t_rand = torch.rand(8192, 64, generator=torch.Generator().manual_seed(9)).to(dev)
    whole = ops.render_rays(scene, tro, trd, near, far, 64, 64, 90.0, 1.0, 1.0, qcam, want_z=True, t_rand=t_rand)
    per_ray = ("color", "depth", "weights_sum", "weights_max", "depth_var", "alpha_sum", "grad_err", "color_mask")
    per_sample = ("mid_z", "dists", "pm", "sdf", "grad", "rgb", "nviews", "weights", "cdf", "z_vals")
    for c0 in range(0, 8192, 512):
        part = ops.render_rays(scene, tro[c0:c0 + 512].contiguous(), trd[c0:c0 + 512].contiguous(), near, far, 64, 64, 90.0, 1.0, 1.0, qcam, want_z=True,
                               t_rand=t_rand[c0:c0 + 512].contiguous())
        assert int((part["pm"] > 0).sum()) > 512            # the quirks need (almost) empty chunks
        for k in per_ray:
            assert torch.equal(part[k], whole[k][c0:c0 + 512]), (k, c0)
        for k in per_sample:
            assert torch.equal(part[k], whole[k][:, c0:c0 + 512]), (k, c0)
    assert float(whole["weights_sum"].max()) > 0.5


def test_camera_terms_and_world_vertices(dev, ops):
    """o2345_camera_terms == intrinsics @ w2cs[:, :3, :] and inverse(w2cs)[:, :3, 3] (what the reference computes with torch.matmul / torch.inverse,
    models/render_utils.py:106, models/projector.py:60-70) -- products as fp32 FMA chains in k order, the inverse by fp64 cofactors -- on rigid poses, on a
    general (sheared, scaled) affine matrix and for V = 1 / 32 / 70; o2345_mc_verts_to_world == numpy's fp64 expression, bit for bit; o2345_preload loads."""
    rng = np.random.default_rng(4)
    for V in (1, 32, 70):
        sc = pkg.synth.make_scene(32 if V > 8 else 8)
        K = np.tile(sc["intrinsics"][:1], (V, 1, 1)).astype(np.float32) * rng.uniform(0.5, 2.0, (V, 1, 1)).astype(np.float32)
        w2c = np.stack([sc["w2cs"][i % sc["w2cs"].shape[0]] for i in range(V)]).astype(np.float32)
        if V == 70:                                           # general invertible matrices, not just rigid poses
            w2c = w2c + rng.normal(0, 0.05, w2c.shape).astype(np.float32)
        proj, cam = ops.camera_terms(torch.from_numpy(K).to(dev), torch.from_numpy(w2c).to(dev))
        want_p = K.astype(np.float64) @ w2c[:, :3, :].astype(np.float64)
        want_c = np.linalg.inv(w2c.astype(np.float64))[:, :3, 3]
        assert np.abs(proj.cpu().numpy() - want_p).max() <= 3e-7 * np.abs(want_p).max()
        assert np.abs(cam.cpu().numpy() - want_c).max() <= 2e-7 * max(1.0, np.abs(want_c).max())
        tp = torch.matmul(torch.from_numpy(K), torch.from_numpy(w2c)[:, :3, :])            # ATen on the CPU evaluates the same FMA chain
        assert float((proj.cpu() - tp).abs().max()) <= 1e-6 * float(tp.abs().max())
    v = torch.from_numpy(rng.uniform(0, 255, (100003, 3))).to(dev)
    b0, b1 = np.array([-1.0, -0.53, 0.251], np.float32), np.array([1.0, 1.47, 2.003], np.float32)       # float32 bounds, like the reference's tensors
    ref = v.cpu().numpy() / 255.0 * (b1 - b0)[None, :] + b0[None, :]
    got = ops.mc_verts_to_world(v.clone(), 256, torch.from_numpy(b0), torch.from_numpy(b1))
    assert np.array_equal(got.cpu().numpy(), ref)
    ops._preloaded.discard(dev)
    ops.preload(dev)
    assert dev in ops._preloaded


@pytest.mark.parametrize("ns,ni", [(32, 32), (48, 80)])
def test_render_with_other_sample_counts(dev, ops, ns, ni, lib_instance):
    """n_importance / 4 != 16 new samples per round (the renderer is built around 16: fused merge with the block in registers): the merge then runs as its
    own launch on the global lists (k_ray_merge_any, the pointer-walk form of render_math.h).  Three-clause contract against the oracle with the same
    sample counts, and the two kernel forms still bit-identical."""
    import render_check as RC
    s = small_scene()
    d = dev_scene(s, dev, ops)
    sc = s["sc"]
    a = _oracle_args(s, 0.0)
    scene = {k: d[k] for k in ("sdf_blob", "color_mfma_blob", "color_x3_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")}
    ro, rd = rays_for(s, 80, seed=ns + ni)
    near, far = float(sc["query_near_far"][0]), float(sc["query_near_far"][1])
    res = RC.three_clause(ops, dev, scene, a, torch.from_numpy(ro), torch.from_numpy(rd), near, far, 0.2, 1.0, 1.0, "f16x3", label=f"{ns}+{ni} samples",
                          n_samples=ns, n_importance=ni, strict_e2e=False)
    assert res["rays_hitting_surface"] > 10 and res["e2e"]["color_max"] < 6e-2, res
    qcam = torch.from_numpy(sc["query_c2w"][:3, 3].copy()).to(dev)
    outs = {}
    for name, thr in (("group", "1000000000"), ("stream", "0")):
        lib_instance({"O2345_RAY_STREAM_MIN": thr})
        outs[name] = ops.render_rays(scene, torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), near, far, ns, ni, 7.4, 1.0, 1.0, qcam, want_z=True)
    for k, v in outs["group"].items():
        assert torch.equal(v, outs["stream"][k]), k
