"""SparseNeuSRenderer + Projector on the HIP back end (mirror of models/sparse_neus_renderer.py:22-937 and
models/projector.py:11-425; general rendering, lod 0)."""
import contextlib
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .rendering_network import DeferredColour
from .sparse_sdf_network import _attr_cache, channel_last


# process-wide sums of every renderer's whole-image counters (SparseNeuSRenderer.whole_image_stats); dropin.py prints them once at exit
WHOLE_IMAGE_TOTALS = dict(images=0, chunks_served=0, plain_calls=0, fallbacks_by_reason={})


def _scene_maps(feature_maps, color_maps, w2cs, intrinsics):
    """-> (colour map [V,H,W,64] = rgb | features | pad, proj [V,3,4], cam_pos [V,3]) of the scene's source views (models/projector.py:96-228 gathers from
    them); cached per tensor object + version."""
    # (the partner tensor enters the key by object, storage and version: a bare id() can be reused by a later temporary with other content)
    cm = _attr_cache(feature_maps, "_o2345_cmaps", (id(color_maps), color_maps.data_ptr(), color_maps._version),
                     lambda: ops.pack_color_maps(feature_maps.detach().contiguous().float(), color_maps.detach().contiguous().float()))
    proj, cam_pos = _attr_cache(w2cs, "_o2345_cam", (id(intrinsics), intrinsics.data_ptr(), intrinsics._version),
                                lambda: ops.camera_terms(intrinsics.detach(), w2cs.detach()))
    return cm, proj, cam_pos


def _inv_s(var, cached=True):
    """SingleVarianceNetwork's inv_s = exp(10 variance) clipped to [1e-6, 1e6] (models/fields.py:179-186, sparse_neus_renderer.py:340) as a host float, once per
    (parameter object, version): ONE scalar read-back, then the reference's fp32 expression on the host (two ATen launches -- exp, clip -- cost 27 ms the first
    time a process runs them, inside the reference's val_step bracket).  ``cached=False`` (render_core, a handful of calls): read the parameter every time."""
    def make():
        v = np.float32(var.detach().reshape(-1)[0].item())
        return float(np.clip(np.exp(np.float32(10.0) * v, dtype=np.float32), np.float32(1e-6), np.float32(1e6)))
    if not cached:
        return make()
    return _attr_cache(var, "_o2345_inv_s", (var.data_ptr(),), make)       # (`p.data = t` swaps the storage without bumping the version counter)


def _variance_tensor(var, inv_s, dev):
    """The returned dict's `variance` entry (1 / inv_s as a 0-d device tensor, :609-633), once per (parameter, version, device); a fill, not a host copy."""
    return _attr_cache(var, "_o2345_var_t", (str(dev), inv_s), lambda: torch.full((), 1.0 / inv_s, dtype=torch.float32, device=dev))


def _host_scalar(t):
    """float(t) for a one-element tensor, read back ONCE per (tensor object, version): near / far / the variance parameter are device tensors that the
    trainer passes unchanged to every chunk; reading them per call is a device synchronisation per chunk."""
    return _attr_cache(t, "_o2345_host", (), lambda: float(t.reshape(-1)[0]))


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
        self._image, self._side = None, None                             # whole-image mode (render())
        # counters and the abandonment count live in ONE dict object: nn.DataParallel re-creates its per-device replicas from this module on every forward
        # (shallow copies of __dict__), so anything a replica must remember for the next image -- "the mode switched itself off" -- has to be shared by reference
        self._stats = dict(images=0, chunks_served=0, plain_calls=0, fallbacks_by_reason={}, abandoned=0)
        # the side stream of the whole-image mode: created with the renderer (the first stream a process creates costs 6 ms in the HIP runtime -- not inside
        # the first val_step bracket); the networks are on their device when the trainer builds the renderer (exp_runner_generic_blender_val.py:93-129)
        p = next(iter(sdf_network.parameters()), None) if isinstance(sdf_network, nn.Module) else None
        if self.whole_image and p is not None and p.is_cuda:
            self._side = torch.cuda.Stream(device=p.device)

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
        feat = feature_volume.reshape(C, -1).t()[idx]
        return torch.cat([torch.zeros(idx.numel(), 1, device=coords.device, dtype=coords.dtype), coords], dim=1), feat.contiguous()

    # ---- the whole image behind the trainer's unchanged chunk loop ------------------------------------------------------------
    # GenericTrainer.val_step renders an image as `for ro, rd in zip(rays_o.split(512), rays_d.split(512)): render(ro, rd, ...)` (trainer_generic.py:503-524):
    # 128 calls of 16 launches each for a 256^2 image, every launch latency-bound.  The chunks are VIEWS of one ray tensor, so the first call can see the
    # whole image: it renders EVERY 512-ray segment in one fused call (O2345RenderIO.segment_rays: the reference's two per-call rules and the per-call
    # scalars per segment -- bit-identical to the separate calls, tests/test_gpu_segments.py), draws the host random numbers of every chunk in the
    # reference's interleaved order (t_rand, pts_random, t_rand, ...), and the later calls return slices.  What a later call checks before it trusts the
    # cache: the same ray tensors (object, storage, version), the same scene / network / scalar arguments, the expected position in the image, and that
    # torch's host generator is exactly where the previous chunk left it (then it is advanced as the chunk's own draws would have).  Anything else falls
    # back to a plain call.  O2345_WHOLE_IMAGE=0 disables the mode.
    # Which path an image took is observable: ``whole_image_stats()`` (per renderer) / ``WHOLE_IMAGE_TOTALS`` (process-wide, printed once at exit by dropin.py):
    # images rendered whole, chunks served as slices, plain calls, and the fallbacks by reason -- "rng" (somebody else drew from torch's host generator, e.g.
    # nn.DataParallel's other device threads), "args" (other scene / scalar arguments), "weights", "order" (not the next chunk), "rays" (other ray tensors),
    # "oom" (the image did not fit: the mode switches itself off for this renderer and the call is served as a plain call).
    whole_image = os.environ.get("O2345_WHOLE_IMAGE", "1") not in ("", "0")
    WHOLE_IMAGE_MAX_RAYS = 1 << 21
    WHOLE_IMAGE_MAX_ABANDONED = 1            # images rendered whole of which only the first chunk was ever asked for, before the mode switches itself off: a caller
                                             # that passes rays[:512] of a bigger tensor once, or nn.DataParallel's device threads moving the shared host
                                             # generator (every image would fall back after its first chunk), pays for ONE speculative image, not for every image
    WHOLE_IMAGE_BYTES_PER_RAY = 12288        # outputs 6.3 KB + workspace ~5 KB per ray at 64 + 64 samples (ops.render_rays), rounded up
    WHOLE_IMAGE_MEMORY_FRACTION = 0.5        # of the device memory that is free (or cached by torch and unused) when the first chunk arrives
    image_batches = int(os.environ.get("O2345_IMAGE_BATCHES", "4"))

    @property
    def _abandoned(self):
        return self._stats["abandoned"]

    @_abandoned.setter
    def _abandoned(self, v):
        self._stats["abandoned"] = int(v)

    def whole_image_stats(self):
        """{"images", "chunks_served", "plain_calls", "fallbacks_by_reason": {...}, "enabled"} of this renderer since construction."""
        st = self._stats
        return dict(images=st["images"], chunks_served=st["chunks_served"], plain_calls=st["plain_calls"], fallbacks_by_reason=dict(st["fallbacks_by_reason"]),
                    enabled=bool(self.whole_image and self._abandoned < self.WHOLE_IMAGE_MAX_ABANDONED))

    def _count(self, key, reason=None):
        for st in (self._stats, WHOLE_IMAGE_TOTALS):
            if reason is None:
                st[key] += 1
            else:
                st[key][reason] = st[key].get(reason, 0) + 1

    def _max_image_rays(self, dev):
        """The largest image rendered whole: WHOLE_IMAGE_MAX_RAYS, and no more than fits in WHOLE_IMAGE_MEMORY_FRACTION of the memory that is available now
        (the reference's 512-ray chunk loop exists to bound memory: an image that fits chunk by chunk must not die here)."""
        if dev.type != "cuda":
            return self.WHOLE_IMAGE_MAX_RAYS
        free, _ = torch.cuda.mem_get_info(dev)
        avail = free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        return int(min(self.WHOLE_IMAGE_MAX_RAYS, self.WHOLE_IMAGE_MEMORY_FRACTION * avail / self.WHOLE_IMAGE_BYTES_PER_RAY))

    def _drop_image(self):
        """Release the cached image.  Batches still running on the side stream read the scene's tensors and the ray storage, which were allocated on the
        caller's stream: the caller's stream waits for the side stream first, so nothing the caller frees next can be reused under those kernels."""
        c, self._image = self._image, None
        if c is not None and self._side is not None and c["dev"].type == "cuda" and any(not bt["joined"] for bt in c["batches"]):
            torch.cuda.current_stream(c["dev"]).wait_stream(self._side)

    @staticmethod
    def _chunk_of_image(t):
        """t [n,3] that is a block of rows of a larger contiguous float32 tensor (`rays.reshape(-1, 3).split(chunk)`) -> (image [R,3] view of that tensor,
        first row, that tensor) or None."""
        b = t._base
        if (b is None or t.dim() != 2 or t.shape[1] != 3 or t.dtype != torch.float32 or not t.is_contiguous() or b.dtype != torch.float32
                or not b.is_contiguous() or b.numel() % 3):
            return None
        off = t.storage_offset() - b.storage_offset()
        if off < 0 or off % 3 or off // 3 + t.shape[0] > b.numel() // 3:
            return None
        return b.view(-1, 3), off // 3, b

    def _pack_rows(self, c, a, b, k):
        """The returned dict of chunk k = rays [a, b) of the cached image c."""
        bt = c["batches"][c["batch_of"][k]]
        if not bt["joined"]:                       # first chunk of a batch rendered on the side stream: the caller's stream waits for THAT batch only, and the
            bt["joined"] = True                    # caching allocator learns that the batch's buffers are used on the caller's stream too
            if bt["event"] is not None:
                cur = torch.cuda.current_stream(c["dev"])
                cur.wait_event(bt["event"])
                for t in list(bt["o"].values()) + [bt["sdf_random"]]:
                    t.record_stream(cur)
        a, b, k = a - bt["a0"], b - bt["a0"], k - bt["k0"]
        r, sc, var, dev, inv_s = bt["rows"], bt["o"]["scalars"][k], c["var"][0], c["dev"], c["inv_s"]
        ws = r["weights_sum"][a:b]
        return {"depth": r["depth"][a:b], "color_fine": r["color"][a:b], "color_fine_mask": r["mask"][a:b], "color_outside": None,
                "color_outside_mask": None, "color_mlp": None, "color_mlp_mask": None,
                "variance": _variance_tensor(var, inv_s, dev),
                "cdf_fine": r["cdf"][a:b], "depth_variance": r["depth_var"][a:b], "weights_sum": ws, "weights_max": r["weights_max"][a:b],
                "alpha_sum": sc[0], "alpha_mean": sc[1], "gradients": r["grad"][a:b], "weights": r["weights"][a:b], "gradient_error_fine": sc[2],
                "inside_sphere": r["pm"][a:b], "sdf": r["sdf"][a:b].reshape(-1, 1), "sdf_random": bt["sdf_random"][k], "blended_color_patch": None,
                "blended_color_patch_mask": None, "weights_sum_fg": ws}

    def _serve_chunk(self, rays_o, rays_d, near, far, sdf_network, rendering_network, perturb, background_rgb, alpha_inter_ratio, args):
        """-> the returned dict of this call if it is the NEXT chunk of the cached image under unchanged arguments, weights and host-generator state; else
        None (the cache is dropped and the caller makes a plain call)."""
        c = self._image
        R = rays_o.shape[0]
        co, cd = self._chunk_of_image(rays_o), self._chunk_of_image(rays_d)
        store = lambda t: (t.data_ptr(), t._version, t.numel())
        same = lambda t, rec: (t is rec[0] and getattr(t, "_version", None) == rec[1]) or (not torch.is_tensor(t) and not torch.is_tensor(rec[0]) and t == rec[0])
        why = None
        if co is None or cd is None or co[1] != cd[1] or store(co[2]) != c["store"][0] or store(cd[2]) != c["store"][1]:
            why = "rays"
        elif co[1] != c["next"] * c["n"] or R != min(c["n"], c["R"] - co[1]):
            why = "order"
        elif not ((float(perturb) > 0) == c["perturb"] and alpha_inter_ratio == c["air"] and background_rgb == c["bg"] and (self.n_samples, self.n_importance) == c["ns"]
                  and all(t is o and t._version == v for t, (o, v) in zip(args, c["args"])) and same(near, c["near"]) and same(far, c["far"])):
            why = "args"
        elif not (self.variance_network.variance is c["var"][0] and c["var"][0]._version == c["var"][1] and c["var"][0].data_ptr() == c["var"][2]
                  and sdf_network is c["nets"][0] and rendering_network is c["nets"][1]
                  and sdf_network.sdf_layer.weights_key() == c["wkeys"][0] and rendering_network.weights_key() == c["wkeys"][1]):
            why = "weights"
        elif not torch.equal(torch.get_rng_state(), c["states"][c["next"] - 1]):
            why = "rng"
        if why is not None:
            self._drop_image()                               # another image, other arguments, out of order, or somebody drew from the host generator
            self._count("fallbacks_by_reason", why)
            if c["next"] <= 1:
                self._abandoned += 1                         # rendered whole, read once: the mode switches itself off for this renderer (WHOLE_IMAGE_MAX_ABANDONED)
            return None
        k = c["next"]
        torch.set_rng_state(c["states"][k])                  # the host generator advances as this chunk's own draws (t_rand, pts_random) would have
        c["next"] = k + 1
        self._abandoned = 0
        self._count("chunks_served")
        a = co[1]
        out = self._pack_rows(c, a, a + R, k)
        if a + R >= c["R"]:
            self._image = None                               # last chunk served (its batch is joined): release the image's buffers
        return out

    def _pack(self, o, sl, sc, sdf_random, var, inv_s, dev):
        """The reference's returned dict (:609-633) for the rays ``sl`` of the call's sample-major outputs ``o``."""
        return {"depth": o["depth"][sl, None], "color_fine": o["color"][sl], "color_fine_mask": o["color_mask"].view(torch.bool)[sl, None], "color_outside": None,
                "color_outside_mask": None, "color_mlp": None, "color_mlp_mask": None,
                "variance": _variance_tensor(var, inv_s, dev),
                "cdf_fine": o["cdf"][:, sl].t(), "depth_variance": o["depth_var"][sl, None], "weights_sum": o["weights_sum"][sl, None],
                "weights_max": o["weights_max"][sl, None], "alpha_sum": sc[0], "alpha_mean": sc[1],
                "gradients": o["grad"][:, sl].permute(1, 0, 2), "weights": o["weights"][:, sl].t(), "gradient_error_fine": sc[2],
                "inside_sphere": o["pm"][:, sl].t(), "sdf": o["sdf"][:, sl].t().reshape(-1, 1), "sdf_random": sdf_random, "blended_color_patch": None,
                "blended_color_patch_mask": None, "weights_sum_fg": o["weights_sum"][sl, None]}

    def _render_image(self, img, R, dev, scene, nr, fr, inv_s, air, bg, qcam, pin, perturb, near, far, sdf_network, rendering_network, var,
                      alpha_inter_ratio, background_rgb, args):
        """The first chunk of an image: every R-ray segment of it in one fused call per batch; caches the image and returns chunk 0's dict."""
        io_, id_, bases = img
        Ri = io_.shape[0]
        K = (Ri + R - 1) // R
        # An image of >= 8,192 rays goes out in up to `image_batches` batches of whole segments on a SIDE stream: the host draws a batch's random numbers, launches it,
        # draws the next -- and later, while the trainer pulls chunk after chunk to the host (a .cpu() per chunk on ITS stream), the GPU is still rendering
        # the following batches.  A chunk waits only for its own batch (one event per batch), not for the image.
        nb = max(1, min(self.image_batches, Ri // 4096)) if dev.type == "cuda" else 1          # (a batch of >= 4,096 rays runs the streaming sampler kernels)
        # batch boundaries in chunks: equal batches, except that the LAST one is a third of the others -- what the host does with a batch's chunks (serve + the
        # trainer's own .cpu() calls, ~0.11 ms per chunk) only overlaps the rendering of LATER batches, so the last batch's share is pure tail
        if nb > 1 and os.environ.get("O2345_IMAGE_TAIL", "1") not in ("", "0"):
            big = -(-K * 3 // (3 * nb - 2))
            starts = [min(K, i * big) for i in range(nb)]
        else:
            starts = [min(K, i * ((K + nb - 1) // nb)) for i in range(nb)]
        starts = sorted(set(starts))
        bounds = [(k0, k1) for k0, k1 in zip(starts, starts[1:] + [K]) if k1 > k0]
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        if nb > 1:
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            self._side.wait_stream(cur)                                     # the scene's tensors were produced on the caller's stream
        # the host stream of the K calls of the trainer's loop, in the reference's order: per call t_rand = torch.rand(z_vals.shape) (:506-515, only when
        # perturb > 0), then pts_random = torch.rand([1024, 3]) (:606); the generator is then put back to where it stands after the FIRST call
        states, batches = [], []
        rng_before = torch.get_rng_state()
        try:
            for bi, k1 in bounds:
                a0, a1 = bi * R, min(Ri, k1 * R)
                t_b = torch.empty(a1 - a0, self.n_samples, pin_memory=pin) if perturb > 0 else None
                p_b = torch.empty(k1 - bi, 1024, 3, pin_memory=pin)
                for k in range(bi, k1):
                    ra, rb = k * R - a0, min(Ri, (k + 1) * R) - a0
                    if perturb > 0:
                        t_b[ra:rb] = torch.rand(rb - ra, self.n_samples)
                    p_b[k - bi] = torch.rand([1024, 3])
                    states.append(torch.get_rng_state())
                with (torch.cuda.stream(self._side) if nb > 1 else contextlib.nullcontext()):
                    o = ops.render_rays(scene, io_[a0:a1], id_[a0:a1], nr, fr, self.n_samples, self.n_importance, inv_s, air, bg, qcam,
                                        t_rand=t_b.to(dev, non_blocking=True) if perturb > 0 else None, want_scalars=True, segment_rays=R)
                    pts_random = p_b.to(dev, non_blocking=True).view(-1, 3) * 2 - 1
                    sdf_random = ops.sdf_mlp(scene["sdf_blob"], scene["vol_cl"], pts_random, variant=0)["sdf"].view(k1 - bi, 1024, 1)
                    for dead in ("mid_z", "dists", "rgb", "nviews", "alpha_sum", "grad_err"):      # per-sample outputs no returned entry reads: 2.7 of 6.3 KB per ray
                        o.pop(dead, None)
                    rows = dict(depth=o["depth"][:, None], color=o["color"], mask=o["color_mask"].view(torch.bool)[:, None], cdf=o["cdf"].t(),
                                depth_var=o["depth_var"][:, None], weights_sum=o["weights_sum"][:, None], weights_max=o["weights_max"][:, None],
                                grad=o["grad"].permute(1, 0, 2), weights=o["weights"].t(), pm=o["pm"].t(), sdf=o["sdf"].t())      # ray-major views: a chunk is a row range
                    ev = None
                    if nb > 1:
                        ev = torch.cuda.Event()
                        ev.record(self._side)
                batches.append(dict(a0=a0, k0=bi, o=o, rows=rows, sdf_random=sdf_random, event=ev, joined=False))
        except BaseException:
            torch.set_rng_state(rng_before)                                  # a failed call must not leave the host generator advanced by a whole image
            raise
        torch.set_rng_state(states[0])
        store = lambda t: (t.data_ptr(), t._version, t.numel())     # (the Python object of a view's base is not guaranteed to be the same one twice)
        # (scene / bases: the packed weights and ray storages stay alive while cached, so their addresses cannot be handed to other tensors)
        batch_of = [b for b, (k0, k1) in enumerate(bounds) for _ in range(k0, k1)]
        self._image = dict(n=R, R=Ri, batch_of=batch_of, next=1, states=states, batches=batches, scene=scene, bases=bases, store=(store(bases[0]), store(bases[1])),
                           args=[(t, t._version) for t in args], near=(near, getattr(near, "_version", None)), far=(far, getattr(far, "_version", None)),
                           perturb=float(perturb) > 0, air=alpha_inter_ratio, bg=background_rgb, var=(var, var._version, var.data_ptr()), inv_s=inv_s,
                           wkeys=(sdf_network.sdf_layer.weights_key(), rendering_network.weights_key()), nets=(sdf_network, rendering_network),
                           ns=(self.n_samples, self.n_importance), dev=dev)
        self._count("images")
        self._count("chunks_served")
        return self._pack_rows(self._image, 0, R, 0)

    @torch.no_grad()
    def render(self, rays_o, rays_d, near, far, sdf_network, rendering_network, perturb_overwrite=-1, background_rgb=None,
               alpha_inter_ratio=0.0, lod=None, conditional_volume=None, conditional_valid_mask_volume=None, feature_maps=None,
               color_maps=None, w2cs=None, intrinsics=None, img_wh=None, query_c2w=None, if_general_rendering=True,
               if_render_with_grad=True, img_index=None, rays_uv=None, pre_sample=False, bg_ratio=0.0):
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        if pre_sample or bg_ratio > 0 or not if_general_rendering:
            raise NotImplementedError("o2345 render: general rendering without pre_sample / bg_ratio (the released val / export configuration)")
        if rays_o.shape[0] == 0:
            raise ValueError("o2345 render: empty ray batch")
        if self._image is not None:                          # a later chunk of an image the first chunk rendered whole: a slice, after the checks of _serve_chunk
            hit = self._serve_chunk(rays_o, rays_d, near, far, sdf_network, rendering_network, perturb, background_rgb, alpha_inter_ratio,
                                    (conditional_volume, conditional_valid_mask_volume, feature_maps, color_maps, w2cs, intrinsics, query_c2w))
            if hit is not None:
                return hit
        cm, proj, cam_pos = _scene_maps(feature_maps, color_maps, w2cs, intrinsics)
        R = rays_o.shape[0]
        dev = rays_o.device
        scene = dict(sdf_blob=sdf_network.sdf_layer.blob(), vol_cl=channel_last(conditional_volume),
                     maskvol=_attr_cache(conditional_valid_mask_volume, "_o2345_flat", (), lambda: conditional_valid_mask_volume.reshape(-1).contiguous().float()),
                     cmaps=cm, proj=proj, cam_pos=cam_pos, color_mfma_blob=rendering_network.mfma_blob(), color_x3_blob=rendering_network.x3_blob())
        var = self.variance_network.variance
        inv_s = _inv_s(var)
        # near / far: one value each (the runner passes the query view's [1] tensors) or one per ray ([N_rays, 1], :486-490)
        nt, ft = torch.as_tensor(near), torch.as_tensor(far)
        sample_dist = None
        if nt.numel() == 1 and ft.numel() == 1:
            nr, fr = _host_scalar(nt), _host_scalar(ft)
        else:
            nr = nt.to(dev).float().reshape(-1).expand(R).contiguous() if nt.numel() in (1, R) else None
            fr = ft.to(dev).float().reshape(-1).expand(R).contiguous() if ft.numel() in (1, R) else None
            if nr is None or fr is None:
                raise ValueError(f"o2345 render: near / far must have 1 or N_rays = {R} elements (got {nt.numel()}, {ft.numel()})")
            sample_dist = float(((fr - nr) / self.n_samples).mean())                   # :484
        air, bg = float(alpha_inter_ratio), 0.0 if background_rgb is None else float(background_rgb)        # None: nothing is added (:430-431)
        qcam = _attr_cache(query_c2w, "_o2345_qcam", (), lambda: query_c2w.reshape(-1, 4, 4)[0, :3, 3].contiguous().float())
        pin = dev.type == "cuda"
        # ---- the FIRST chunk of an image: render every segment now (later chunks: _serve_chunk above)
        img = None
        if sample_dist is None and self.whole_image and self._abandoned < self.WHOLE_IMAGE_MAX_ABANDONED and R % 64 == 0:
            co, cd = self._chunk_of_image(rays_o), self._chunk_of_image(rays_d)
            if (co is not None and cd is not None and co[1] == cd[1] == 0 and co[0].shape == cd[0].shape and R < co[0].shape[0] <= self._max_image_rays(dev)):
                img = (co[0], cd[0], (co[2], cd[2]))
        if img is not None:
            try:
                return self._render_image(img, R, dev, scene, nr, fr, inv_s, air, bg, qcam, pin, perturb, near, far, sdf_network, rendering_network, var,
                                          alpha_inter_ratio, background_rgb,
                                          (conditional_volume, conditional_valid_mask_volume, feature_maps, color_maps, w2cs, intrinsics, query_c2w))
            except torch.cuda.OutOfMemoryError:
                # the image did not fit after all (the host generator is back where the call found it): this renderer stops speculating, the blocks the attempt
                # left in torch's cache go back to the driver, and THIS call is served the way the reference serves it -- one 512-ray chunk
                self._image, self.whole_image = None, False
                self._count("fallbacks_by_reason", "oom")
                if self._side is not None:
                    torch.cuda.current_stream(dev).wait_stream(self._side)
                torch.cuda.empty_cache()
        # ---- one plain call
        self._count("plain_calls")
        # stratified jitter exactly as the reference draws it (:506-515): torch.rand(z_vals.shape) on the HOST generator, then moved to the device
        # -> the same numbers as the reference under the same torch.manual_seed; drawn into pinned memory and copied asynchronously (torch's
        # caching host allocator keeps the block until the copy has run), the kernel applies lower + (upper - lower) * t
        t_rand = torch.rand(R, self.n_samples, pin_memory=pin).to(dev, non_blocking=True) if perturb > 0 else None
        o = ops.render_rays(scene, rays_o.contiguous().float(), rays_d.contiguous().float(), nr, fr, self.n_samples, self.n_importance,
                            inv_s, air, bg, qcam, t_rand=t_rand, sample_dist=sample_dist, want_scalars=True)
        # the 1,024 random points of every call (:606): torch.rand([1024, 3]) on the HOST generator -- the second and last host draw of a call, after
        # t_rand -- moved to the device and mapped to (-1, 1) there, exactly the reference's expression: under one torch.manual_seed the host stream
        # advances by R * n_samples + 3,072 numbers per call, so every later chunk of an image draws the reference's jitter too.  Evaluated by the
        # SDF-only kernel (the reference's sdf() also returns 128 features nobody reads here)
        pts_random = torch.rand([1024, 3], pin_memory=pin).to(dev, non_blocking=True) * 2 - 1
        sdf_random = ops.sdf_mlp(scene["sdf_blob"], scene["vol_cl"], pts_random, variant=0)["sdf"][:, None]
        # scalars: [alpha_sum.mean(), alpha_sum.sum() / (R S), gradient error, evaluated points]: one tiny kernel inside the call
        return self._pack(o, slice(0, R), o["scalars"], sdf_random, var, inv_s, dev)

    @torch.no_grad()
    def render_core(self, rays_o, rays_d, z_vals, sample_dist, lod, sdf_network, rendering_network, background_alpha=None,
                    background_sampled_color=None, background_rgb=None, alpha_inter_ratio=0.0, conditional_volume=None,
                    conditional_valid_mask_volume=None, feature_maps=None, color_maps=None, w2cs=None, intrinsics=None, img_wh=None, query_c2w=None,
                    if_general_rendering=True, if_render_with_grad=True, img_index=None, rays_uv=None, bg_num=0):
        """render_core on GIVEN sample depths z_vals [N_rays, n_samples] (:171-455): the part of render() downstream of the hierarchical sampler, in the
        reference's call form with the reference's returned keys.  Composed from the C ABI's stage entries (ops.render_core)."""
        if not if_general_rendering or bg_num or self.if_fitted_rendering:
            raise NotImplementedError("o2345 render_core: general rendering, bg_num = 0 (the released val / export configuration)")
        cm, proj, cam_pos = _scene_maps(feature_maps, color_maps, w2cs, intrinsics)
        scene = dict(sdf_blob=sdf_network.sdf_layer.blob(), vol_cl=channel_last(conditional_volume),
                     maskvol=_attr_cache(conditional_valid_mask_volume, "_o2345_flat", (), lambda: conditional_valid_mask_volume.reshape(-1).contiguous().float()),
                     cmaps=cm, proj=proj, cam_pos=cam_pos, color_mfma_blob=rendering_network.mfma_blob(), color_x3_blob=rendering_network.x3_blob())
        var = self.variance_network.variance
        inv_s = _inv_s(var, cached=False)
        o = ops.render_core(scene, rays_o.contiguous().float(), rays_d.contiguous().float(), z_vals.t().contiguous().float(), float(sample_dist), inv_s,
                            float(alpha_inter_ratio), 0.0 if background_rgb is None else float(background_rgb),
                            query_c2w.reshape(-1, 4, 4)[0, :3, 3].contiguous().float())
        R, S = z_vals.shape
        dev = rays_o.device
        ge = o["grad_err"].double().sum(0)
        return {"color": o["color"], "color_mask": o["color_mask"].view(torch.bool)[:, None], "color_mlp": None, "color_mlp_mask": None,
                "sdf": o["sdf"].t().reshape(-1, 1), "depth": o["depth"][:, None], "dists": o["dists"].t(), "gradients": o["grad"].permute(1, 0, 2),
                "variance": torch.full((R * S, 1), 1.0 / inv_s, device=dev), "mid_z_vals": o["mid_z"].t(), "weights": o["weights"].t(),
                "weights_sum": o["weights_sum"][:, None], "alpha_sum": o["alpha_sum"][:, None], "alpha_mean": o["alpha_sum"].sum() / (R * S),
                "cdf": o["cdf"].t(), "gradient_error": (ge[0] / (ge[1] + 1e-5)).float(), "inside_sphere": o["pm"].t(), "blended_color_patch": None,
                "blended_color_patch_mask": None, "weights_sum_fg": o["weights_sum"][:, None]}

    @torch.no_grad()
    def extract_fields(self, bound_min, bound_max, resolution, query_func, device, **kwargs):
        """u = -sdf on linspace(bound_min, bound_max, resolution)^3 (:881-905), as a DEVICE tensor [R,R,R].  The reference's own bounds (-1, 1): one fused
        launch, lattice generated in-kernel, layer 0 from per-axis tables.  Any other box: the three axes from torch.linspace on the host exactly as the
        reference builds them (:887-889), the lattice points materialised once on the device, one launch of the point kernel (instead of 64 chunks with a
        device -> host copy each)."""
        vol = kwargs["conditional_volume"]
        layer = self.sdf_network.sdf_layer
        R = int(resolution)
        b0 = torch.as_tensor(bound_min, dtype=torch.float32).reshape(-1).cpu()
        b1 = torch.as_tensor(bound_max, dtype=torch.float32).reshape(-1).cpu()
        if bool((b0 == -1.0).all()) and bool((b1 == 1.0).all()):
            u = ops.sdf_mlp(layer.blob(), channel_last(vol), None, variant=0, grid_R=R, sign=-1.0, grid_tables=layer.grid_tables(R))["sdf"]
            return u.view(R, R, R)
        if R ** 3 >= 2 ** 31:
            raise ValueError("o2345 extract_fields: resolution^3 must stay below 2^31")
        ax = [torch.linspace(float(b0[d]), float(b1[d]), R).to(vol.device) for d in range(3)]
        pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3).contiguous()
        u = ops.sdf_mlp(layer.blob(), channel_last(vol), pts, variant=0, sign=-1.0)["sdf"]
        return u.view(R, R, R)

    @torch.no_grad()
    def extract_geometry(self, sdf_network, bound_min, bound_max, resolution, threshold, device, occupancy_mask=None, **kwargs):
        """-> (vertices float64 [Nv,3] in world units, triangles int64 [Nt,3], u float32 [R,R,R]) as numpy (:907-937)."""
        u = self.extract_fields(bound_min, bound_max, resolution, None, device, **kwargs)
        if occupancy_mask is not None:
            e = torch.nn.functional.interpolate((1 - occupancy_mask)[None, None].float(), [resolution] * 3, mode="nearest")[0, 0] > 0
            u = torch.where(e.to(u.device), torch.full_like(u, -100.0), u)
        v, t = ops.marching_cubes(u.contiguous(), float(threshold))
        # the reference returns numpy (vertices float64 in world units, triangles, u): index -> world on the device (fp64, the expression numpy would
        # evaluate), then three copies into pinned memory and one synchronisation
        if v.shape[0]:
            ops.mc_verts_to_world(v, resolution, torch.as_tensor(bound_min, dtype=torch.float32), torch.as_tensor(bound_max, dtype=torch.float32))
        vh, th, uh = ops.to_host_numpy(v, t, u)
        return vh, th, uh
