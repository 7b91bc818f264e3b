"""CPU: the MFMA weight-blob packing of csrc/sdf_mlp.hip, checked by emulating the kernel's dataflow in numpy
(documented lane layouts of v_mfma_f32_32x32x2_f32) against the oracle MLP and its analytic gradient."""
import numpy as np

import weights_emulators as EMU
import torch

from oracle import recon as O


def _weights(pkg):
    W = pkg.weights.init_sdf_weights(seed=3, latent_scale=0.1, pe_scale=0.02)
    rng = np.random.default_rng(1)
    for k in ("b0", "b1", "b2"):
        W[k] = (W[k] + rng.normal(0, 0.05, 128)).astype(np.float32)
    return W


def test_blob_emulation_matches_oracle(pkg):
    W = _weights(pkg)
    blob = pkg.weights.pack_sdf_blob(W)
    assert blob.size == pkg.weights.SDF_BLOB_FLOATS
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (29, 3)).astype(np.float32)
    lat = rng.normal(0, 1, (29, 16)).astype(np.float32)
    y, gpe, glat = EMU.emulate_sdf_blob(blob, pts, lat)
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    ref = O.sdf_mlp(torch.from_numpy(pts), torch.from_numpy(lat), Wt).numpy()
    assert np.abs(y - ref).max() < 2e-5
    # gradient pieces: d sdf / d pe and d sdf / d latent via autograd on the oracle MLP
    torch.set_grad_enabled(True)
    p = torch.from_numpy(pts)
    pe = O.embed(p).requires_grad_(True)
    lt = torch.from_numpy(lat).requires_grad_(True)
    sp = lambda t: torch.nn.functional.softplus(t, beta=100)
    h = sp(pe @ Wt["w0"].T + Wt["b0"])
    h = sp(torch.cat([h, lt], 1) @ Wt["w1"].T + Wt["b1"])
    out = (torch.cat([h, lt], 1) @ Wt["w2"].T + Wt["b2"])[:, 0].sum()
    gpe_ref, glat_ref = torch.autograd.grad(out, [pe, lt])
    assert np.abs(gpe - gpe_ref.numpy()).max() < 2e-4 * max(1.0, gpe_ref.abs().max().item())
    assert np.abs(glat - glat_ref.numpy()).max() < 2e-4 * max(1.0, glat_ref.abs().max().item())


def test_x3_blob_emulation(pkg):
    """csrc/sdf_mlp_x3.hip (split-f16 operands, three MFMAs per product) reproduces the fp32 network to fp32-class accuracy."""
    Wn = pkg.weights
    W = _weights(pkg)
    blob = Wn.pack_sdf_blob(W)
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (32, 3)).astype(np.float32)
    lat = rng.normal(0, 1, (32, 16)).astype(np.float32)
    lat[:4] *= 1e-3                                          # small operands exercise the f16-subnormal lo halves
    sdf = EMU.emulate_sdf_blob_x3(blob, pts, lat)
    w0, w1, w2 = (W[k].astype(np.float64) for k in ("w0", "w1", "w2"))
    sp = lambda a: np.where(a * 100 > 20, a, np.log1p(np.exp(np.minimum(a * 100, 50))) / 100)
    pe = O.embed(torch.from_numpy(pts)).numpy().astype(np.float64)
    h0 = sp(pe @ w0.T + W["b0"])
    h1 = sp(np.concatenate([h0, lat], 1) @ w1.T + W["b1"])
    ref = np.concatenate([h1, lat], 1) @ w2[0] + W["b2"][0]
    err = np.abs(sdf - ref).max()
    assert err < 2e-6 * max(1.0, np.abs(ref).max()), err
    # the split itself: hi + lo recovers x to 2^-20 relative (or the f16 subnormal spacing)
    x = rng.normal(0, 1, 4096).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 2, 4096).astype(np.float32)
    hi, lo = Wn.f16_split_device(x)
    assert np.all(np.abs(hi.astype(np.float64) + lo - x) <= np.maximum(np.abs(x) * 2.0 ** -20, 6e-8))


def test_cached_pack_is_content_keyed(pkg, tmp_path, monkeypatch):
    """weights.cached_pack: the same parameters -> the same blob without re-packing (process memo, then the disk cache of a later process); changed
    parameters -> a different key; a truncated cache file is recomputed, never trusted."""
    import os
    W = pkg.weights
    monkeypatch.setenv("O2345_CACHE_DIR", str(tmp_path))
    monkeypatch.setattr(W, "_MEM_CACHE", {})
    sd = W.init_color_state_dict(3)
    calls = []
    orig = W.pack_color_x3_blob
    monkeypatch.setattr(W, "pack_color_x3_blob", lambda s_: calls.append(1) or orig(s_))
    a = W.packed_color_x3_blob(sd)
    b = W.packed_color_x3_blob({k: np.array(v) for k, v in sd.items()})          # equal content, different objects
    assert len(calls) == 1 and a is b and np.array_equal(a, orig(sd))
    files = [f for f in os.listdir(tmp_path) if f.startswith("color_x3_")]
    assert len(files) == 1
    W._MEM_CACHE.clear()                                                         # "a later process": served from disk
    c = W.packed_color_x3_blob(sd)
    assert len(calls) == 1 and np.array_equal(c, a)
    sd2 = dict(sd)
    sd2["s"] = np.float32(0.25)
    d = W.packed_color_x3_blob(sd2)
    assert len(calls) == 2 and not np.array_equal(d, a)
    W._MEM_CACHE.clear()
    with open(os.path.join(tmp_path, files[0]), "r+b") as f:
        f.truncate(100)
    e = W.packed_color_x3_blob(sd)
    assert len(calls) == 3 and np.array_equal(e, a)
    # a well-formed .npy of the wrong size / dtype (stale layout, planted file) is not trusted either: the kernels read the blob at fixed offsets
    for bad in (a[:-8].copy(), a.astype(np.float64), np.full_like(a, np.nan)):
        W._MEM_CACHE.clear()
        np.save(os.path.join(tmp_path, files[0]), bad)
        n = len(calls)
        e = W.packed_color_x3_blob(sd)
        assert len(calls) == n + 1 and e.dtype == np.float32 and np.array_equal(e, a)
    # equal arrays under other key names are a different key
    W._MEM_CACHE.clear()
    k1 = W.cached_pack("t", [np.ones(3, np.float32)], lambda: np.zeros(4, np.float32), 4, ("a",))
    k2 = W.cached_pack("t", [np.ones(3, np.float32)], lambda: np.ones(4, np.float32), 4, ("b",))
    assert not np.array_equal(k1, k2)
    monkeypatch.setenv("O2345_CACHE_DIR", "off")
    W._MEM_CACHE.clear()
    assert W.cache_dir() is None and np.array_equal(W.packed_sdf_blob(W.init_sdf_weights(1)), W.pack_sdf_blob(W.init_sdf_weights(1)))


def test_color_mfma_blob_emulation_matches_oracle(pkg):
    """The MFMA formulation of GeneralRenderingNetwork (csrc/color_mfma.hip), emulated lane by lane, vs the oracle network."""
    from scene_util import color_t
    sd = pkg.weights.init_color_state_dict(5)
    rng = np.random.default_rng(2)
    for k in list(sd):                      # non-zero biases everywhere
        if k.endswith(".bias"):
            sd[k] = (sd[k] + rng.normal(0, 0.1, sd[k].shape)).astype(np.float32)
    blob = pkg.weights.pack_color_mfma_blob(sd)
    RW = color_t(sd)
    for G in (8, 4, 32):
        P = 32 // G
        geo = rng.normal(0, 1, (P, 16)).astype(np.float32)
        rf = rng.normal(0, 1, (P, G, 59)).astype(np.float32)
        rd = rng.normal(0, 1, (P, G, 4)).astype(np.float32)
        rd[..., 3] = rng.uniform(-1, 1, (P, G))
        m = (rng.uniform(0, 1, (P, G)) > 0.3)
        if P > 1:
            m[0] = False                                     # a point that no view sees
        rf64 = np.concatenate([rf, np.zeros((P, G, 5), np.float32)], -1)
        got = EMU.emulate_color_mfma(blob, geo, rf64, rd, m.astype(np.float32), G)
        ref, _ = O.rendering_network(RW, torch.from_numpy(geo), torch.from_numpy(rf).permute(1, 0, 2), torch.from_numpy(rd).permute(1, 0, 2),
                                     torch.from_numpy(m).permute(1, 0))
        assert np.abs(got - ref.numpy()).max() < 2e-5, G
        # split-f16 instantiation: regrouped blob, activations split as on the device
        xblob = pkg.weights.pack_color_x3_blob(sd)
        assert xblob.size == pkg.weights.CX_BLOB_FLOATS
        gotx = EMU.emulate_color_mfma(blob, geo, rf64, rd, m.astype(np.float32), G, x3_blob=xblob)
        assert np.abs(gotx - got).max() < 5e-6, (G, np.abs(gotx - got).max())


def test_shared_rows_matrix_operand_of_color_pts(pkg):
    """A_S (csrc/color_pts.hip): the view-independent rows of base_fc.0 as an MFMA A operand -- every (block, k slot, lane) entry is the W_S entry
    of (output neuron 32 b + lane % 32, operand row of slot s in half lane // 32); the split-f16 copy reproduces it to 2^-21."""
    W = pkg.weights
    sd = W.init_color_state_dict(5)
    b32, bx = W.pack_color_mfma_blob(sd), W.pack_color_x3_blob(sd)
    assert b32.size == W.CM_BLOB_FLOATS and bx.size == W.CX_BLOB_FLOATS
    off = W.CM_LAYOUT["A_S"][0]
    A = b32[off:off + 2 * 72 * 64].reshape(2, 72, 64)
    ws_off = W.CM_LAYOUT["W_S"][0]
    ws = b32[ws_off:ws_off + 144 * 64].reshape(144, 64)
    n_checked = 0
    for b in range(2):
        for s in range(72):
            for lane in range(64):
                i, h = lane & 31, lane >> 5
                if s < 8:
                    row = 8 * h + s
                else:
                    f = 32 * h + (s - 8) % 32
                    row = None if f >= 59 else (16 + f if s < 40 else 80 + f)
                want = 0.0 if row is None else ws[row, 32 * b + i]
                assert A[b, s, lane] == want
                n_checked += row is not None
    assert n_checked == 2 * 32 * (16 + 59 + 59) and np.abs(A).max() > 0
    sec = bx[W.CX_A_S:W.CX_A_S + 2 * 9 * 512].view(np.float16).reshape(2, 9, 2, 64, 8).astype(np.float32)
    rec = (sec[:, :, 0] + sec[:, :, 1]).transpose(0, 1, 3, 2).reshape(2, 72, 64)              # [b][k-step][lane][t] -> [b][8 s + t][lane]
    assert np.abs(rec - A).max() <= 2.0 ** -21 * max(1.0, np.abs(A).max())
    # the prefix read by k_color_mfma is unchanged by the appended segment
    assert off == W.CM_LAYOUT["S_SCALAR"][0] + 4 and W.CX_A_S == W.CX_A_END + (off - W.CM_TAIL0)


def test_sdf_grid_tables_are_the_separable_first_layer(pkg):
    """weights.sdf_grid_tables (the TAB form of csrc/sdf_mlp_x3.hip): on the lattice linspace(-1,1,R)^3 the first layer of the SDF network is
    b0 + Tx[ix] + Ty[iy] + Tz[iz] -- stored in the kernels' t domain, i.e. times weights.SOFTPLUS_SCALE = 100 / ln 2; checked against W0 . embed(p) + b0 of
    the oracle's embedding, columns in the kernels' lane order."""
    W = _weights(pkg)
    for R in (17, 64):
        axes, bias = pkg.weights.sdf_grid_tables(W, R)
        assert axes.shape == (3, R, 128) and axes.dtype == np.float32 and bias.shape == (128,)
        order = np.array([pkg.weights.neuron_of(nb, r, h) for h in (0, 1) for nb in range(4) for r in range(16)])
        assert sorted(order.tolist()) == list(range(128))
        lin = torch.linspace(-1, 1, R)
        rng = np.random.default_rng(R)
        idx = rng.integers(0, R, (300, 3))
        p = torch.stack([lin[idx[:, 0]], lin[idx[:, 1]], lin[idx[:, 2]]], -1)
        ref = (O.embed(p).double() @ torch.from_numpy(W["w0"]).double().T + torch.from_numpy(W["b0"]).double()).numpy()[:, order]
        got = (axes[0][idx[:, 0]].astype(np.float64) + axes[1][idx[:, 1]] + axes[2][idx[:, 2]] + bias) / pkg.weights.SOFTPLUS_SCALE
        assert np.abs(got - ref).max() < 5e-6

