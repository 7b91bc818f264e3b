"""CPU: the per-element device math (csrc/*_math.h, compiled for the host by tests/hostcheck) against the oracle.
This is the same source the GPU executes; it pins indexing / edge semantics before any GPU time is spent."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import recon as O
from scene_util import small_scene, sdfW_t, rays_for

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HERE, "hostcheck", "libhostcheck.so")
    src = os.path.join(HERE, "hostcheck", "hostcheck.hip")
    csrc = os.path.join(os.path.dirname(HERE), "one-2-3-45_amd", "csrc")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(".h")])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
                               "-fPIC", "-shared", src, "-o", so], stderr=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    L.hc_linspace.restype = ctypes.c_float
    L.hc_linspace.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    return L


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_linspace_matches_torch(hc):
    for a, b, n in ((0.0, 1.0, 64), (-1.0, 1.0, 256), (-1.0, 1.0, 360), (-1.0, 1.0, 37), (0.03125, 0.96875, 16), (-0.147, 1.937, 64)):
        ref = torch.linspace(a, b, n)
        got = np.array([hc.hc_linspace(a, b, n, i) for i in range(n)], np.float32)
        assert np.array_equal(got, ref.numpy()), (a, b, n)          # bit-exact with ATen's CPU linspace


def test_sincos_pe(hc):
    """csrc/pe_math.h (the SDF kernels' positional-encoding sin/cos) vs a double reference over every argument the
    embedding can produce: 2^k * x, k = 0..5, |x| <= 2 (the scene box is [-1,1]^3; rays sample slightly outside)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-2, 2, 200000), np.linspace(-2, 2, 100001), [0.0, 1.0, -1.0, np.pi / 4, np.pi / 2, -np.pi]]).astype(np.float32)
    worst = 0.0
    for k in range(6):
        a = np.ascontiguousarray(x * np.float32(2 ** k))
        s_ = np.empty_like(a); c_ = np.empty_like(a)
        hc.hc_sincos_pe(P(a), a.size, P(s_), P(c_))
        worst = max(worst, np.abs(s_ - np.sin(a.astype(np.float64))).max(), np.abs(c_ - np.cos(a.astype(np.float64))).max())
        # and never worse than 2 ulp(1) from the fp32 libm the reference's torch.sin / torch.cos resolve to
        assert np.abs(s_ - np.sin(a)).max() <= 2.4e-7 and np.abs(c_ - np.cos(a)).max() <= 2.4e-7
    assert worst <= 1.2e-7, worst                 # 1 ulp of 1.0


def test_costvol_rows(hc):
    s = small_scene()
    V, H, W, D = s["V"], s["H"], s["W"], s["D"]
    nhwc = np.ascontiguousarray(s["f16"].transpose(0, 2, 3, 1))
    proj = s["sc"]["affine_mats"]
    nvox = D ** 3
    cnt = np.zeros(nvox, np.uint8); row = np.zeros(nvox, np.int32); coords = np.zeros((nvox, 4), np.int32)
    rows = np.zeros((nvox, 32), np.float32)
    n = hc.hc_costvol(P(nhwc), P(proj), V, H, W, D, D, D, ctypes.c_float(s["voxel_size"]), P(s["sc"]["partial_vol_origin"]),
                      1, P(cnt), P(row), P(coords), P(rows))
    assert n == s["coords"].shape[0]
    assert np.array_equal(cnt, s["cnt"].numpy().astype(np.uint8))
    assert np.array_equal(coords[:n], s["coords"].numpy())
    ref = s["vol"].numpy()
    assert np.abs(rows[:n] - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_trilinear_and_mask(hc):
    s = small_scene()
    D = s["D"]
    vol_cl = np.ascontiguousarray(s["dense"][0].permute(1, 2, 3, 0).numpy())
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1.15, 1.15, (4000, 3)).astype(np.float32)
    pts[:40] = np.array([[-1.0, 0.2, 0.3]], np.float32); pts[40:60] = 1.0; pts[60:70] = 1.04
    pts[70:80, 0] = np.float32(2.0 * 3.5 / D - 1 + 1.0 / D)       # exactly half-way between two mask cells
    out = np.zeros((len(pts), 16), np.float32)
    hc.hc_trilinear_ref(P(vol_cl), D, P(pts), len(pts), P(out))
    ref = O.trilinear_ref(s["dense"][0], torch.from_numpy(pts)).numpy()
    assert np.abs(out - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    m = np.zeros(len(pts), np.float32)
    maskvol = np.ascontiguousarray(s["mask"][0, 0].numpy())
    hc.hc_mask_nearest(P(maskvol), D, P(pts), len(pts), P(m))
    assert np.array_equal(m, O.mask_nearest(s["mask"][0, 0], torch.from_numpy(pts)).numpy())


def _coarse(s, ro, rd, n_samples=64):
    near, far = s["sc"]["query_near_far"]
    z = (near + (far - near) * torch.linspace(0, 1, n_samples))[None].repeat(len(ro), 1)
    pts = (torch.from_numpy(ro)[:, None] + torch.from_numpy(rd)[:, None] * z[..., None]).reshape(-1, 3)
    sd = O.sdf(pts, s["dense"][0], sdfW_t(s["sdfW"]))[0][:, 0].reshape(len(ro), n_samples)
    return z, sd


def test_upsample_and_merge(hc):
    s = small_scene()
    D = s["D"]
    ro, rd = rays_for(s, 40)
    z, sd = _coarse(s, ro, rd)
    maskvol = np.ascontiguousarray(s["mask"][0, 0].numpy())
    W = sdfW_t(s["sdfW"])
    R = len(ro)
    zc = np.zeros((128, R), np.float32); sc_ = np.zeros((128, R), np.float32)
    zc[:64] = z.numpy().T; sc_[:64] = sd.numpy().T
    zt, st = z.clone(), sd.clone()
    S = 64
    for i in range(4):
        nz_ref = O.up_sample(torch.from_numpy(ro), torch.from_numpy(rd), zt, st, 16, 64.0 * 2 ** i, s["mask"][0, 0])
        wbuf = np.zeros((128, R), np.float32); nz = np.zeros((16, R), np.float32)
        hc.hc_upsample(P(ro), P(rd), R, P(zc), P(sc_), S, ctypes.c_float(64.0 * 2 ** i), P(maskvol), D, P(wbuf), 16, P(nz))
        assert np.abs(nz.T - nz_ref.numpy()).max() < 2e-5, f"round {i}"
        # feed BOTH sides the same new samples (the oracle's) so later rounds stay comparable
        zt2, st2 = O.cat_z(torch.from_numpy(ro), torch.from_numpy(rd), zt, nz_ref, st, s["dense"][0], s["mask"][0, 0], W)
        pts = (torch.from_numpy(ro)[:, None] + torch.from_numpy(rd)[:, None] * nz_ref[..., None]).reshape(-1, 3)
        m = O.mask_nearest(s["mask"][0, 0], pts) > 0
        nsdf = torch.full((pts.shape[0],), 100.0)
        if m.sum() > 1:
            nsdf[m] = O.sdf(pts[m], s["dense"][0], W)[0][:, 0]
        nzc = np.ascontiguousarray(nz_ref.numpy().T); nsc = np.ascontiguousarray(nsdf.reshape(R, 16).numpy().T)
        hc.hc_merge(R, P(zc), P(sc_), S, P(nzc), P(nsc), 16)
        S += 16
        assert np.array_equal(zc[:S].T, zt2.numpy())
        assert np.array_equal(sc_[:S].T, st2.numpy())
        zt, st = zt2, st2


def test_composite(hc):
    s = small_scene()
    rng = np.random.default_rng(5)
    R, S = 33, 128
    ro, rd = rays_for(s, R)
    mid = np.sort(rng.uniform(-0.1, 1.9, (S, R)).astype(np.float32), 0)
    dists = np.diff(mid, axis=0, append=mid[-1:] + 0.03).astype(np.float32)
    pm = (rng.uniform(0, 1, (S, R)) > 0.4).astype(np.float32)
    sdf = rng.normal(0, 0.05, (S, R)).astype(np.float32)
    grad = rng.normal(0, 1, (S, R, 3)).astype(np.float32)
    rgb = rng.uniform(0, 1, (S, R, 3)).astype(np.float32)
    nv = rng.integers(0, 4, (S, R)).astype(np.uint8)
    inv_s, air, bg = float(np.exp(2.0)), 1.0, 1.0
    f = lambda *sh: np.zeros(sh, np.float32)
    color, depth, w, cdf, ws, wm, dv, asum, ge = f(R, 3), f(R), f(S, R), f(S, R), f(R), f(R), f(R), f(R), f(R, 2)
    cm = np.zeros(R, np.uint8)
    hc.hc_composite(P(ro), P(rd), R, S, P(mid), P(dists), P(pm), P(sdf), P(grad), P(rgb), P(nv), ctypes.c_float(inv_s),
                    ctypes.c_float(air), ctypes.c_float(bg), P(color), P(depth), P(w), P(cdf), P(ws), P(wm), P(dv), P(asum), P(ge), P(cm))
    # oracle formulae (render_core tail) on the same inputs, ray-major
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.moveaxis(a, 0, 1)))
    dirs = torch.from_numpy(rd)[:, None].expand(R, S, 3)
    tdot = (dirs * t(grad)).sum(-1)
    icos = -(torch.relu(-tdot * 0.5 + 0.5) * (1 - air) + torch.relu(-tdot) * air) * t(pm)
    half = icos.clip(-10, 10) * t(dists) * 0.5
    pc = torch.sigmoid((t(sdf) - half) * inv_s); nc = torch.sigmoid((t(sdf) + half) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0, 1) * t(pm)
    T = torch.cumprod(torch.cat([torch.ones(R, 1), 1 - alpha + 1e-7], 1), 1)[:, :-1]
    wr = alpha * T
    assert np.abs(w.T - wr.numpy()).max() < 1e-6
    assert np.abs(color - ((t(rgb) * wr[..., None]).sum(1) + bg * (1 - wr.sum(1, keepdim=True))).numpy()).max() < 2e-6
    dref = (t(mid) * wr).sum(1)
    assert np.abs(depth - dref.numpy()).max() < 2e-6
    assert np.abs(dv - ((t(mid) - dref[:, None]) ** 2 * wr).sum(1).numpy()).max() < 2e-6
    assert np.array_equal(cm, ((t(nv) >= 2).sum(1) > 8).numpy().astype(np.uint8))
    assert np.abs(cdf.T - pc.numpy()).max() < 1e-6


def test_visible_views_256_cubed_matches_oracle_exactly(hc):
    """The kept-voxel set is an INTEGER result: the device projection (csrc/geom_math.h:project_voxel, compiled for the host) against the oracle
    on every voxel of a 256^3 lattice (BASELINE config 5), 8 views: exact.  With a mul / add sequence instead of the GEMM's FMA chain 3 of the
    16.7 M voxels sat on the other side of a frustum boundary (|g| within one ulp of 1); 128^3 happened to have none."""
    pkg = __import__("importlib").import_module("one-2-3-45_amd")
    D = 256
    sc = pkg.synth.make_scene(8, image_seed=6)
    cnt = np.empty(D ** 3, np.uint8)
    aff = np.ascontiguousarray(sc["affine_mats"], np.float32)
    org = np.ascontiguousarray(sc["partial_vol_origin"], np.float32)
    hc.hc_visible_views(P(aff), 8, 256, 256, D, D, D, ctypes.c_float(2.0 / (D - 1)), P(org), P(cnt))
    lat = O.voxel_lattice([D, D, D])
    ref = torch.cat([O.project(lat[s:s + (1 << 20)] * (2.0 / (D - 1)) + torch.from_numpy(org)[None], torch.from_numpy(aff), 256, 256)[3].sum(1)
                     for s in range(0, lat.shape[0], 1 << 20)])
    bad = np.nonzero(cnt.astype(np.int64) != ref.numpy())[0]
    assert bad.size == 0, (bad[:10], cnt[bad[:10]], ref.numpy()[bad[:10]])
    assert 0.45 * D ** 3 < int((ref > 1).sum()) < 0.65 * D ** 3
