"""GPU: out-of-bounds WRITE check of the C ABI's kernels.

GPU AddressSanitizer needs code objects built for page-fault retry and the matching runtime switch, which this GPU pool refuses (profiles/NOTES.md, "sanitizer"), so the check is done with
guard bands instead: every device buffer the Python side hands to the library -- outputs, caller-sized workspaces (o2345_render_workspace_bytes,
o2345_conv2d_workspace_bytes, the two-call marching-cubes protocol), scratch -- is carved out of a larger allocation whose bytes immediately before and
immediately after the buffer (no rounding: the first byte past numel * itemsize is a canary) are filled with a pattern; after whole scene passes at awkward
sizes (ray counts that are not multiples of 64 or of a wavefront, odd volume sizes, empty lists) every canary must be intact.  A kernel that rounds a store up
to a vector width, writes a tail element, or mis-sizes a workspace fails here; torch's own allocator would have hidden it behind its 512-byte rounding."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

pkg = importlib.import_module("one-2-3-45_amd")
ops = importlib.import_module("one-2-3-45_amd.ops")
pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
costreg = importlib.import_module("one-2-3-45_amd.costreg")
featurenet = importlib.import_module("one-2-3-45_amd.featurenet")

PRE, POST, PATTERN = 512, 4096, 0xA5             # PRE keeps the buffer at the alignment torch's allocator gives (512 B); the kernels assume 16 B


class GuardedTorch:
    """Stands in for the `torch` module inside the product modules: the allocation functions return guarded device buffers, everything else is torch's."""

    def __init__(self):
        self.live = []                            # (raw uint8 tensor, payload bytes, what)

    def __getattr__(self, name):
        return getattr(torch, name)

    def _guarded(self, shape, dtype, device, what):
        dtype = dtype or torch.float32
        shape = tuple(int(x) for x in (shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape))
        n = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size() if len(shape) else torch.empty((), dtype=dtype).element_size()
        raw = torch.full((PRE + n + POST,), PATTERN, dtype=torch.uint8, device=device)
        self.live.append((raw, n, what))
        return raw[PRE:PRE + n].view(dtype).view(shape)

    @staticmethod
    def _is_cuda(device):
        return device is not None and torch.device(device).type == "cuda"

    def empty(self, *shape, dtype=None, device=None, **kw):
        if not self._is_cuda(device) or kw.get("pin_memory"):
            return torch.empty(*shape, dtype=dtype, device=device, **kw)
        return self._guarded(shape, dtype, device, "empty")

    def zeros(self, *shape, dtype=None, device=None, **kw):
        if not self._is_cuda(device):
            return torch.zeros(*shape, dtype=dtype, device=device, **kw)
        return self._guarded(shape, dtype, device, "zeros").zero_()

    def ones(self, *shape, dtype=None, device=None, **kw):
        if not self._is_cuda(device):
            return torch.ones(*shape, dtype=dtype, device=device, **kw)
        return self._guarded(shape, dtype, device, "ones").fill_(1)

    def full(self, shape, value, dtype=None, device=None, **kw):
        if not self._is_cuda(device):
            return torch.full(shape, value, dtype=dtype, device=device, **kw)
        if dtype is None:
            dtype = torch.float32 if isinstance(value, float) else torch.int64
        return self._guarded((tuple(shape),), dtype, device, "full").fill_(value)

    def empty_like(self, t, **kw):
        return self.empty(*t.shape, dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device)) if t.is_cuda else torch.empty_like(t, **kw)

    def zeros_like(self, t, **kw):
        return self.zeros(*t.shape, dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device)) if t.is_cuda else torch.zeros_like(t, **kw)

    def full_like(self, t, value, **kw):
        return self.full(tuple(t.shape), value, dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device)) if t.is_cuda else torch.full_like(t, value, **kw)

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for raw, n, what in self.live:
            pre_ok = bool((raw[:PRE] == PATTERN).all())
            post = raw[PRE + n:]
            post_ok = bool((post == PATTERN).all())
            if not (pre_ok and post_ok):
                first = int(torch.nonzero(post != PATTERN)[0]) if not post_ok else None
                bad.append((what, n, "underrun" if not pre_ok else f"overrun: first damaged byte {first} past the end"))
        return bad


@pytest.fixture()
def guard(monkeypatch):
    g = GuardedTorch()
    for mod in (ops, pipeline, costreg, featurenet):
        monkeypatch.setattr(mod, "torch", g)
    monkeypatch.setattr(ops, "_ws_cache", {})            # workspaces are re-allocated through the guard
    yield g


def _scene(dev, V, seed):
    sc = pkg.synth.make_scene(V, image_seed=seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    return sc, T, proj, cam_pos


@pytest.mark.parametrize("D,R_mesh,V,precision", [(33, 47, 3, "f16x3"), (48, 64, 8, "f16x3"), (20, 21, 2, "fp32")])
def test_no_kernel_writes_outside_its_buffers(guard, D, R_mesh, V, precision):
    dev = torch.device("cuda:0")
    wt = pipeline.SceneWeights(dev, seed=0, sdf_precision=precision, color_precision=precision)
    sc, T, proj, cam_pos = _scene(dev, V, seed=3)
    vol = pipeline.build_volume(wt, T(sc["images"]), T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1))
    assert not guard.check(), guard.check()
    n_alloc = len(guard.live)
    assert n_alloc > 30, "the guard must actually see the volume build's buffers"
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256)
    near, far, qcam = float(sc["query_near_far"][0]), float(sc["query_near_far"][1]), T(sc["query_c2w"][:3, 3].copy())
    rng = np.random.default_rng(0)
    for R in (1, 63, 64, 65, 333, 512, 4097, 5000):          # sixteen-lane form below 4,096 rays, streaming form above; tails of every width
        sel = rng.integers(0, ro.shape[0], R)
        o = pipeline.render(wt, vol, proj, cam_pos, T(ro[sel]), T(rd[sel]), near, far, qcam, want_z=True)
        assert torch.isfinite(o["color"]).all()
        bad = guard.check()
        assert not bad, (R, bad)
    # segment mode (a whole image behind the chunk loop), jittered, with per-segment scalars; last segment short
    scene = dict(sdf_blob=wt.sdf_blob, vol_cl=vol["vol_cl"], maskvol=vol["maskvol"], cmaps=vol["cmaps"], proj=proj, cam_pos=cam_pos,
                 color_mfma_blob=wt.color_mblob, color_x3_blob=wt.color_xblob, sdf_precision=wt.sdf_precision, color_precision=wt.color_precision)
    sel = rng.integers(0, ro.shape[0], 1600)
    ops.render_rays(scene, T(ro[sel]), T(rd[sel]), near, far, 64, 64, wt.inv_s, 1.0, 1.0, qcam, t_rand=torch.rand(1600, 64).to(dev), want_scalars=True,
                    segment_rays=512, color_stats=ops.color_stats_buffer(dev))
    ops.render_rays(scene, T(ro[sel[:100]]), T(rd[sel[:100]]), near, far, 16, 64, wt.inv_s, 0.5, 0.0, qcam, weight_cull=0.0)      # S = 80: the "first 100" rule's short form
    # rays that miss everything: the reference's two per-call rules write their own list entries
    miss_o = torch.tensor([[5.0, 5.0, 5.0]] * 70, device=dev)
    miss_d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.2, 0.1]] * 70, device=dev), dim=-1)
    ops.render_rays(scene, miss_o, miss_d, 0.1, 2.0, 64, 64, wt.inv_s, 1.0, 1.0, qcam)
    ops.render_rays(scene, miss_o, miss_d, 0.1, 2.0, 16, 16, wt.inv_s, 1.0, 1.0, qcam)                                               # S = 32 < 100
    bad = guard.check()
    assert not bad, bad
    # stage entry points on the same scene
    z = torch.sort(torch.rand(64, 333, device=dev) * (far - near) + near, dim=0).values.contiguous()
    sel = rng.integers(0, ro.shape[0], 333)
    rc = ops.render_core(scene, T(ro[sel]), T(rd[sel]), z, (far - near) / 64, wt.inv_s, 1.0, 1.0, qcam)
    up = ops.ray_upsample(T(ro[sel]), T(rd[sel]), z, rc["sdf"], 64.0, vol["maskvol"], D, 16)
    assert up[0].shape == (16, 333)
    pts = (torch.rand(1001, 3, device=dev) * 2 - 1).contiguous()
    geo, rf, rdiff, m = ops.project_features(vol["vol_cl"], vol["maskvol"], vol["cmaps"], proj, cam_pos, pts, query_cam=qcam)
    ops.color_from_features(wt.color_xblob if precision == "f16x3" else wt.color_mblob, geo, rf, rdiff, m, x3=precision == "f16x3")
    ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=1, want_lat=True, precision=precision)
    ops.view_count(pts, vol["maskvol"], D, proj, V, 256, 256)
    bad = guard.check()
    assert not bad, bad
    # mesh extraction: lattice SDF, the two-call marching cubes, vertex colours, PLY record packing
    verts_idx, tris, rgb, u = pipeline.extract_mesh(wt, vol, proj, cam_pos, R_mesh, return_index_verts=True)
    assert tris.shape[0] > 0
    ops.mesh_pack(verts_idx, tris, R_mesh, scale_mat=np.eye(4, dtype=np.float32), trans_mat=np.eye(4, dtype=np.float32), rgb=rgb)
    ops.marching_cubes(torch.full((5, 5, 5), -1.0, device=dev), 0.0)                       # no crossing at all: empty outputs
    ops.marching_cubes((u * 0 + torch.linspace(-1, 1, R_mesh, device=dev)[:, None, None]).contiguous(), 0.0, index_dtype=torch.int32)   # a plane
    bad = guard.check()
    assert not bad, bad
    assert len(guard.live) > n_alloc + 100


def test_the_guard_itself_catches_a_one_byte_overrun(guard):
    """The checker checks: a one-element overrun (what a kernel rounding its last store up would do) is reported."""
    dev = torch.device("cuda:0")
    t = guard.empty(7, dtype=torch.uint8, device=dev)
    base = guard.live[-1][0]
    assert t.data_ptr() == base.data_ptr() + PRE and not guard.check()
    base[PRE + 7] = 0
    bad = guard.check()
    assert len(bad) == 1 and "overrun: first damaged byte 0" in bad[0][2]
    base[PRE + 7] = PATTERN
    base[PRE - 1] = 0
    assert guard.check()[0][2] == "underrun"
