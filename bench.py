#!/usr/bin/env python
"""Benchmark of the reconstruction hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" = one full pass of the hot path over one synthetic scene of BASELINE config 2
(8 views x 256^2, 128^3 volume, 512 x 512 rays, mesh extraction on a 256^3 grid):
   FeatureNet + compress layer (HIP convolutions on the matrix cores, ABN folded in) -> cost volume (HIP) -> sparse CNN (HIP) -> dense volume ->
   render 262,144 rays (HIP: hierarchical sampling, SDF / colour networks, compositing) ->
   SDF grid + marching cubes + vertex colours (HIP).
Inputs (images, cameras, weights) are resident in HBM before the timed region.  `value` = rays rendered by all ranks /
wall time of the K steps (whole step, i.e. including the volume build and the mesh extraction -- conservative);
`render_rays_per_s` and `mesh_extract_ms` give the two halves of BASELINE's metric separately.

Every step reconstructs a DIFFERENT scene (seeded images; scene index = rank + world * step, the `scenes_for_rank` deal), so no
step benefits from the previous step's L2 / MALL contents.

N > 1: one process per GPU, every rank reconstructs its own scenes: embarrassingly parallel, no data-path collective
-> "scaling": "weak".  `python bench.py --gpus N` works bare: when WORLD_SIZE is not set it re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (rendezvous on 127.0.0.1); under an existing torchrun launch it uses that.

Output (rank 0): the LAST stdout line is the contract line, compact (<= 4 KB, contract_line()): metric / value / ms_per_step / config / dtype, `roofline`
of the dominant kernel, {frac, ms} of the other three, `cpu_baseline` (the reference's own modules as timed on this host type when the stamped file of this
round matches, kind "reference"; else the oracle port timed live, kind "port") and one max-error number per stage vs the reference (`parity`).  The full
record -- `parity_fullsize`, `c3` (BASELINE config 3), `ref_config` (V=32 / 96^3 / 256^3), `config5` (256^3 / 1024^2 / 512^3), `fp32_whole_step_ms`,
`cpu_baseline_reference`, `parity_reference` detail, `trained_regime`, `pipelined`, `dropin` -- is the stdout line BEFORE it and bench_extra.json.
`--quick` skips the extra blocks.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("one-2-3-45_amd")
pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
ops = importlib.import_module("one-2-3-45_amd.ops")
sharding = importlib.import_module("one-2-3-45_amd.sharding")

config = importlib.import_module("one-2-3-45_amd.config")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3      # v_mfma_f32_32x32x2_f32 = fp32 vector rate
F16_MFMA_PEAK_TF = 2516.6      # v_mfma_f32_32x32x16_f16 / bf16, dense (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)
MFMA_PEAK = {"fp32": FP32_MFMA_PEAK_TF, "f16x3": F16_MFMA_PEAK_TF}
# FLOP the matrix pipe actually executes per unit (padding and, for f16x3, the three partial products included)
COLOR_MFMA_FLOP_PER_PAIR = {"fp32": 189 * 32 * 32 * 2 * 2 / 32, "f16x3": 75 * 32 * 32 * 16 * 2 / 32}
SDF_MFMA_FLOP_PER_POINT = {"fp32": 368 * 4096 / 32, "f16x3": 144 * 32768 / 32}
GRAD_MFMA_FLOP_PER_POINT = {"fp32": 896 * 4096 / 32, "f16x3": 348 * 32768 / 32}
# algorithmic FLOP per unit
SDF_FLOP_SDF_ONLY = 2 * (39 * 128 + 144 * 128 + 144)            # 47,136 per point (SDF-only forward)
SDF_FLOP_GRAD = 2 * SDF_FLOP_SDF_ONLY                          # + ~47,136 for the input gradient (transposed GEMMs)
COLOR_FLOP_PER_PAIR = 38544                                    # per (point, view)
COLOR_KERNEL_PREFIX = "k_color_pts"                             # the colour kernel the default configuration launches (profiles/*_pmc_*.json key prefix)
# matrix instructions of k_color_pts per evaluated (32-point tile, view) pair in the pooling pass / the network pass, and per tile (shared rows)
COLOR_PTS_MFMA = {"f16x3": (9, 75, 54, 32 * 32 * 16 * 2), "fp32": (18, 189, 144, 32 * 32 * 2 * 2)}


class Timer:
    """HIP-event stage timer on torch's current stream (the stream every o2345 kernel is launched on)."""

    def __init__(self):
        self.acc, self.pending = {}, []

    def start(self, name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pending.append((name, e0, e1))

    def stop(self):
        self.pending[-1][2].record()

    def collect(self):
        torch.cuda.synchronize()
        for name, e0, e1 in self.pending:
            self.acc.setdefault(name, []).append(e0.elapsed_time(e1))
        self.pending = []

    def mean(self, name):
        v = self.acc.get(name, [])
        return float(np.mean(v)) if v else 0.0


def scene_images(V, seed):
    """The per-scene input: V source images [V,3,256,256] float32 (what Zero123 hands over), seeded."""
    return np.random.default_rng(seed).random((V, 3, 256, 256), dtype=np.float32)


def make_inputs(dev, V, seed, ray_scale):
    """Camera rig / rays (identical for every scene of the rig, like the reference's fixed 8-view ring) + the images of scene `seed`."""
    sc = pkg.synth.make_scene(V, image_seed=seed)
    assert np.array_equal(sc["images"], scene_images(V, seed))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=ray_scale)
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    return dict(sc=sc, imgs=T(sc["images"]), aff=T(sc["affine_mats"]), origin=sc["partial_vol_origin"], rays_o=T(ro), rays_d=T(rd),
                proj=proj, cam_pos=cam_pos, qcam=T(sc["query_c2w"][:3, 3].copy()), near=float(sc["query_near_far"][0]),
                far=float(sc["query_near_far"][1]), ro_host=ro, rd_host=rd)


def step(wt, inp, D, R_mesh, tm, chunk, imgs=None):
    vs = 2.0 / (D - 1)
    imgs = inp["imgs"] if imgs is None else imgs
    tm.start("volume"); vol = pipeline.build_volume(wt, imgs, inp["aff"], inp["origin"], D, vs); tm.stop()
    tm.start("render")
    n = inp["rays_o"].shape[0]
    outs = []
    for s in range(0, n, chunk):
        outs.append(pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"][s:s + chunk], inp["rays_d"][s:s + chunk],
                                    inp["near"], inp["far"], inp["qcam"]))
    tm.stop()
    tm.start("mesh"); mesh = pipeline.extract_mesh(wt, vol, inp["proj"], inp["cam_pos"], R_mesh); tm.stop()
    return vol, outs, mesh


def pipelined_block(dev, a, inp, imgs_list, n_streams):
    """The SAME scenes once more, dealt round-robin to ``n_streams`` host threads, each with its own HIP stream and its own weights object (scratch and packed-weight
    caches are per object): the latency-bound phases of one scene (volume build: ~45 short kernels and two size read-backs; marching cubes) overlap the long
    render kernels of another.  Reported next to the contract's one-stream number, never instead of it; results are checked bit-identical to the sequential pass
    (they were not before the library was built without packed-FP32 instructions -- profiles/NOTES.md, "co-resident MFMA")."""
    import threading
    K = len(imgs_list)
    wts = [pipeline.SceneWeights(dev, seed=0, sdf_precision=a.precision, color_precision=a.precision) for _ in range(n_streams)]
    for w in wts:
        w.grid_tables(a.mesh_res)

    def digest(vol, outs, mesh):
        return torch.stack([outs[0]["color"].double().sum(), outs[0]["depth"].double().sum(), outs[0]["weights_sum"].double().sum(),
                            vol["rows"].double().sum(), vol["vol_cl"].double().sum(), mesh[0].double().sum(), mesh[2].double().sum()])

    def run(n):
        streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        dig, errs = [None] * K, []

        def worker(i):
            try:
                torch.cuda.set_device(dev)
                tm = Timer()
                with torch.cuda.stream(streams[i]):
                    for k in range(i, K, n):
                        out = step(wts[i], inp, a.vol, a.mesh_res, tm, a.ray_chunk, imgs=imgs_list[k])
                        dig[k] = digest(*out)
                        out = None
                    streams[i].synchronize()
            except Exception as e:                                   # noqa: BLE001
                errs.append(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if errs:
            raise errs[0]
        return dt / K * 1e3, torch.stack(dig).cpu()
    run(n_streams)                                                   # every stream's memory pools filled before anything is timed
    ms1, d1 = run(1)
    msn, dn = run(n_streams)
    n_rays = inp["rays_o"].shape[0]
    return {"streams": n_streams, "scenes": K, "ms_per_scene": msn, "rays_per_s": n_rays / (msn * 1e-3), "ms_per_scene_one_thread_same_harness": ms1,
            "bit_identical_to_sequential": bool(torch.equal(d1, dn)),
            "note": "throughput of K scenes in flight on several streams of ONE GPU; the contract's value / ms_per_step above is the one-stream number"}


def trained_regime_block(dev, wt, inp, D):
    """The regime of a TRAINED model on the benchmark's geometry: SingleVarianceNetwork far from its 0.2 initialisation (variance 0.45 /
    0.65 -> inv_s = exp(10 v) = 90 / 665, models/fields.py:179-186; ref_trained.npz proves the kernels handle it).  Reports, from the per-sample compositing
    weights of one whole render, which fraction of the OCCUPIED samples (the ones the colour network is evaluated on) carry a weight below 2^-24 -- colour work
    that cannot change the image by more than 128 x 2^-24 = 7.6e-6 -- and the render time per regime.  The seeded stand-in SDF (geometric initialisation:
    a sphere-like level set inside the volume) is crossed by the rays: rays_hitting_surface counts weights_sum > 0.5."""
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
    old = (wt.variance, wt.inv_s)
    out = {"note": "occupied = sample mid-points inside kept voxels (sparse_neus_renderer.py:216-221); free-space samples carry the reference's +1e-5 and are never dropped"}
    try:
        for v in (0.2, 0.45, 0.65):
            wt.variance, wt.inv_s = v, float(np.clip(np.exp(10.0 * v), 1e-6, 1e6))
            scene = dict(sdf_blob=wt.sdf_blob, vol_cl=vol["vol_cl"], maskvol=vol["maskvol"], cmaps=vol["cmaps"], proj=inp["proj"], cam_pos=inp["cam_pos"],
                         color_mfma_blob=wt.color_mblob, color_x3_blob=wt.color_xblob, sdf_precision=wt.sdf_precision, color_precision=wt.color_precision)
            call = lambda cull, stats=None: ops.render_rays(scene, inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], 64, 64, wt.inv_s, 1.0, 1.0, inp["qcam"],
                                                            weight_cull=cull, color_stats=stats)
            fn = lambda: call(None)                    # the default: config.WEIGHT_CULL
            sa, sb = ops.color_stats_buffer(dev), ops.color_stats_buffer(dev)
            o_full = call(0.0, sa)
            o = call(None, sb)
            pa, pb = ops.color_stats_read(sa)["pairs_network"], ops.color_stats_read(sb)["pairs_network"]
            dcol = float((o["color"] - o_full["color"]).abs().max())
            same = all(bool(torch.equal(o[k], o_full[k])) for k in ("depth", "weights", "weights_sum", "color_mask", "sdf", "grad", "nviews"))
            o_full = None
            occ = o["pm"] > 0
            w = o["weights"][occ]
            n = int(w.numel())
            fr = lambda t: float((w < t).sum()) / max(1, n)
            out[f"variance_{v}"] = {"inv_s": wt.inv_s, "occupied_samples": n, "rays_hitting_surface": int((o["weights_sum"] > 0.5).sum()),
                                    "frac_occupied_with_weight_below": {"2^-24": fr(2.0 ** -24), "1e-6": fr(1e-6), "1e-5": fr(1e-5), "1e-4": fr(1e-4), "1e-3": fr(1e-3)},
                                    "render_ms_median": median_ms(fn, reps=3), "render_ms_median_exhaustive": median_ms(lambda: call(0.0), reps=3),
                                    "weight_cull": config.weight_cull(), "colour_tile_view_pairs_exhaustive_vs_culled": [pa, pb],
                                    "colour_max_abs_difference_vs_exhaustive": dcol, "bound": 128 * config.weight_cull(),
                                    "everything_but_colour_bit_identical": same}
            o = w = occ = None
    finally:
        wt.variance, wt.inv_s = old
    return out


def render_order_index(pm):
    """Slots (s * R + r) of the occupied mid-points in the order k_ray_finalize writes its list: wave-major (64 consecutive rays),
    sample-major inside a wave.  pm [S, R]."""
    S, R = pm.shape
    pad = (-R) % 64
    occ = torch.nn.functional.pad(pm > 0, (0, pad)).view(S, -1, 64).permute(1, 0, 2)          # [waves, S, 64]
    w, s_, l = torch.nonzero(occ, as_tuple=True)
    return (s_ * R + w * 64 + l).to(torch.int32).contiguous()


def kernel_times(wt, vol, inp, outs, D, reps=5, sdf_precision=None, color_precision=None):
    """Per-kernel timings (HIP events on the launch stream) of the kernels of one render call, for the roofline blocks."""
    sdf_precision = sdf_precision or wt.sdf_precision
    color_precision = color_precision or wt.color_precision
    res = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = ev(), ev()
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.mean(ts))
    dev = inp["imgs"].device
    vs = 2.0 / (D - 1)
    res["costvol_gather_ms"] = timed(lambda: ops.costvol_gather(vol["feats_nhwc"], inp["aff"], (D, D, D), vs, inp["origin"], vol["cnt"], vol["coords"]))
    # the occupied mid-points of the first ray chunk = what the SDF-gradient and colour kernels of a render call process
    o = outs[0]
    R = o["pm"].shape[1]
    idx = render_order_index(o["pm"])
    pts = (inp["rays_o"][None, :R] + inp["rays_d"][None, :R] * o["mid_z"][..., None]).reshape(-1, 3).contiguous()
    if b"list_sort=1" in ops._lib.lib().o2345_knobs():           # what o2345_render_rays does to its list before the network kernels (csrc/list_sort.hip)
        res["list_sort_ms"] = timed(lambda: ops.list_sort_by_visibility(pts, idx, inp["proj"], 256, 256))
        idx = ops.list_sort_by_visibility(pts, idx, inp["proj"], 256, 256)
    res["n_valid_points"] = int(idx.numel())
    res["n_points"] = int(pts.shape[0])
    o2 = {"sdf": torch.empty(pts.shape[0], device=dev), "grad": torch.empty(pts.shape[0], 3, device=dev)}
    res["sdf_grad_ms"] = timed(lambda: ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=2, index=idx, out=o2, precision=sdf_precision))
    res["sdf_mlp_ms"] = timed(lambda: ops.sdf_mlp(wt.sdf_blob, vol["vol_cl"], pts, variant=0, out={"sdf": o2["sdf"]}, precision=sdf_precision))
    V = inp["imgs"].shape[0]
    if color_precision == "f16x3":
        blob, mode = wt.color_xblob, "x3"
    else:
        blob, mode = wt.color_mblob, True
    color = lambda stats=None: ops.color_points(blob, vol["vol_cl"], vol["maskvol"], vol["cmaps"], inp["proj"], inp["cam_pos"], pts,
                                                query_cam=inp["qcam"], index=idx, want_nviews=False, mfma=mode, stats=stats)
    res["color_ms"] = timed(color)
    # how much of the (tile, view) grid the kernel evaluated (views that see none of a tile's 32 points are skipped): caller-owned device counters
    st = ops.color_stats_buffer(dev); color(st); res["color_work"] = ops.color_stats_read(st)
    return res


def network_rooflines(kt, V, sdf_p, col_p):
    """roofline blocks of the three network kernels for one precision mode.  `achieved` = ALGORITHMIC FLOP / HIP-event
    time; `peak` = dense MFMA peak of the type the matrix pipe runs in; `mfma_pipe_util` = FLOP the pipe actually executes
    (padding and the three partial products of the split form included) / time / that peak."""
    nvp, npts = kt["n_valid_points"], kt["n_points"]

    def blk(kernel, units, flop_alg, flop_pipe, ms, peak, extra=None):
        ach = units * flop_alg / (ms * 1e-3) / 1e12
        d = {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "units": units,
             "flop_per_unit": flop_alg, "ms": ms, "mfma_pipe_util": units * flop_pipe / (ms * 1e-3) / 1e12 / peak}
        d.update(extra or {})
        return d
    cname = {"fp32": "k_color_pts<false,false> (points as columns, fp32 MFMA)", "f16x3": "k_color_pts<true,false> (points as columns, split-f16 MFMA, fp32 accumulate)"}[col_p]
    cw = kt["color_work"]
    na, nb_, nt, fl = COLOR_PTS_MFMA[col_p]
    color_pipe_flop = (na * cw["pairs_pooling"] + nb_ * cw["pairs_network"] + nt * cw["tiles"]) * fl        # what the matrix pipe executed in that launch
    sname = {"fp32": ("k_sdf_mlp<0>", "k_sdf_mlp<2>"), "f16x3": ("k_sdf_mlp_x3", "k_sdf_grad_x3")}[sdf_p]
    return {
        "color": blk(cname + ": Projector + GeneralRenderingNetwork", nvp * V, COLOR_FLOP_PER_PAIR, color_pipe_flop / max(1, nvp * V),
                     kt["color_ms"], MFMA_PEAK[col_p],
                     {"tile_view_pairs": cw["tiles"] * V, "pairs_evaluated_pooling_pass": cw["pairs_pooling"], "pairs_evaluated_network_pass": cw["pairs_network"],
                      "tiles_evaluating_every_view": cw["tiles_all_views"],
                      "note": "units = occupied points x views = the reference's work (it evaluates every pair); the kernel skips, bit-identically, the views "
                              "that see none of a tile's 32 points -- `achieved` counts the reference's FLOP, `mfma_pipe_util` the instructions actually executed"}),
        "sdf": blk(sname[0] + ": SDF forward on all sample points", npts, SDF_FLOP_SDF_ONLY, SDF_MFMA_FLOP_PER_POINT[sdf_p], kt["sdf_mlp_ms"],
                   MFMA_PEAK[sdf_p]),
        "sdf_grad": blk(sname[1] + ": SDF + analytic gradient, occupied points", nvp, SDF_FLOP_GRAD, GRAD_MFMA_FLOP_PER_POINT[sdf_p],
                        kt["sdf_grad_ms"], MFMA_PEAK[sdf_p]),
    }


def _pmc_file():
    """The newest committed counter summary (profiles/rNN_pmc_f16x3.json) and whether it was measured on THIS tree's kernels: the file carries the
    hash of csrc/* it was collected on (tools/summarize_rocprof.py:provenance); counter-derived numbers are printed only on a match."""
    import glob
    build = importlib.import_module("one-2-3-45_amd.build")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_pmc_f16x3.json")))
    if not files:
        return None, None, "no profiles/rNN_pmc_f16x3.json"
    path = files[-1]
    d = json.load(open(path))
    want, have = build.sources_sha(), (d.get("_meta") or {}).get("kernel_sources_sha")
    rel = os.path.relpath(path, ROOT)
    if have != want:
        return rel, None, f"{rel} was collected on kernel sources {have} but this tree's csrc/ hashes to {want}: counters not printed (re-run tools/profile_round.sh)"
    return rel, d, None


PMC_FILE, PMC_DATA, PMC_STALE = _pmc_file()


def pmc_traffic(kernel_prefix):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (same workload, separate FETCH_SIZE / WRITE_SIZE runs;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  None if absent or measured on other kernel sources."""
    if PMC_DATA is None:
        return None
    for k, v in PMC_DATA.items():
        if k.startswith(kernel_prefix) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    return None


def pmc_mfma_busy(kernel_prefix):
    """rocprofv3's matrix-pipe utilisation of a kernel from the committed PMC passes: SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x SIMDs), active cycles =
    SQ_BUSY_CYCLES / 32 (the counter is summed over the 32 shader engines; it reproduces launch duration x ~2.3 GHz for every kernel of the profile).  This is
    the "MfmaUtil" the north star asks for; unlike `mfma_pipe_util` (FLOP at the 2.4 GHz peak) it is not diluted by the clock the chip actually sustains."""
    if PMC_DATA is None:
        return None
    for k, v in PMC_DATA.items():
        if k.startswith(kernel_prefix) and v.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            active = v["SQ_BUSY_CYCLES"] / 32.0
            return {"source": PMC_FILE, "kernel": k, "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * 1024.0),
                    "valu_issue_frac": 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / (active * 1024.0), "active_cycles": active}
    return None


def pmc_issue_model(kernel_prefix, ms):
    """What the SIMDs of the dominant kernel spend their time on, from the committed PMC passes (profiles/r02_ubench_issue_model.md: a gfx950 SIMD
    issues EITHER vector OR matrix work, so the matrix-pipe utilisation of an issue-bound kernel is MFMA time / (MFMA + VALU time))."""
    if PMC_DATA is None:
        return None
    for k, v in PMC_DATA.items():
        if k.startswith(kernel_prefix) and "SQ_WAVE_CYCLES" in v:
            waves_per_simd = 3.0 if "mfma<" in k else 2.0
            simd_quads = v["SQ_WAVE_CYCLES"] / waves_per_simd                       # summed over the 1024 SIMDs, in quad-cycles
            return {"source": PMC_FILE, "kernel": k, "valu_wave_instructions": v.get("SQ_INSTS_VALU"), "mfma_busy_cycles": v.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                    "simd_time_valu_issue_frac": v["SQ_ACTIVE_INST_VALU"] / simd_quads,
                    "simd_time_mfma_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / simd_quads,
                    "wave_time_waiting_frac": v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"],
                    "note": "fractions of the SIMD time (sum of the resident waves / waves per SIMD); vector and matrix work of one SIMD do not overlap on gfx950 (tools/ubench, profiles/r02_ubench_issue_model.md)"}
    return None


def cpu_baseline_and_parity(wt, vol, inp, D, n_rays, budget_s=15.0):
    """cpu_baseline: the oracle's render() (CPU restatement of the reference, oracle/recon.py) on a bounded sample of the same rays.
    parity_fullsize: the SAME oracle outputs compared with the HIP path on the same rays (both numerical forms), see tests/fullsize_util.py:
    sampler stage with identical inputs, everything downstream on the HIP path's own sample lists (all rays), end-to-end distribution."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullsize_util as FU                                     # test infrastructure (oracle side of the comparison)
    FU.CHUNK = 512                                                  # the runner's ray batch (trainer_generic.py:503) on BOTH sides: the oracle is timed the way the reference
    torch.set_num_threads(min(32, os.cpu_count() or 1))             # more threads only add fork/join overhead on these op sizes
    dev = inp["imgs"].device
    full = dict(wt=wt, sc=inp["sc"], vol=vol, proj=inp["proj"], cam_pos=inp["cam_pos"], D=D, ro=inp["ro_host"], rd=inp["rd_host"],
                T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    n = full["ro"].shape[0]
    ref, sel, dt = FU.oracle_render_sample(full, n_rays, budget_s=budget_s)
    cpu = {"value": len(sel) / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": this_host()["cpu"],
           "sample": f"{len(sel)} of the {n} rays of the same scene (oracle.render, {FU.CHUNK}-ray chunks, fp32), {dt:.1f} s"}
    par = {"rays": int(len(sel)), "oracle": "oracle/recon.py (pinned to the reference by tests/golden + tests/test_oracle_vs_reference.py)"}
    sub = slice(0, min(len(sel), 512))                             # the downstream / stage checks re-run oracle stages: bounded subset
    for prec in ("f16x3", "fp32"):
        out = FU.gpu_render_sample(full, sel, prec)
        cerr = (out["color"] - ref["color_fine"]).abs().max(1).values
        derr = (out["depth"] - ref["depth"][:, 0]).abs()
        core = FU.oracle_core_on(full, sel[sub], out["z_vals"][sub])
        q = lambda t: [float(torch.quantile(t, x)) for x in (0.5, 0.9, 0.99)] + [float(t.max())]
        par[prec] = {"color_err_q50_q90_q99_max": q(cerr), "depth_err_q50_q90_q99_max": q(derr),
                     "frac_rays_color_gt_1e-4": float((cerr > 1e-4).float().mean()), "frac_rays_color_gt_1e-3": float((cerr > 1e-3).float().mean()),
                     "color_mask_mismatches": int((out["color_mask"].bool() != ref["color_fine_mask"][:, 0]).sum()),
                     "downstream_on_hip_samples": {"rays": int(core["depth"].shape[0]),
                                                   "color_max": float((out["color"][sub] - core["color_fine"]).abs().max()),
                                                   "depth_max": float((out["depth"][sub] - core["depth"][:, 0]).abs().max()),
                                                   "weights_max": float((out["weights"][sub] - core["weights"]).abs().max())}}
    ops_ = ops
    s256 = sel[:256]
    dz, pdf, width = FU.sampler_stage_check(ops_, dev, torch.from_numpy(full["ro"][s256]), torch.from_numpy(full["rd"][s256]), inp["near"], inp["far"],
                                            FU._oracle_args(full), vol["maskvol"], D)
    par["sampler_stage_identical_inputs"] = {"samples": int(dz.numel()), "dz_max": float(dz.max()), "dz_over_bin_width_max": float((dz / width.clamp(min=1e-9)).max())}
    ce, _ = FU.oracle_self_sensitivity(full, s256, {k: v[:256] for k, v in ref.items()}, seeds=(1,))
    par["oracle_self_sensitivity_to_2e-6_sdf_noise_color_q50_q90_q99_max"] = [float(torch.quantile(ce.flatten(), x)) for x in (0.5, 0.9, 0.99)] + [float(ce.max())]
    # the volume build itself (FeatureNet -> compress -> cost volume -> sparse CNN -> scatter) against the oracle's, from the images, at this size
    t0 = time.time()
    ov = FU.oracle_volume(wt, inp["sc"], D)
    par["volume_vs_oracle"] = dict(FU.volume_vs_oracle(vol, ov, D), oracle_seconds=time.time() - t0,
                                   note="max abs error / max|oracle| per tensor; integer results exact (tests/test_gpu_edges_and_fullsize.py::test_fullsize_volume_vs_oracle)")
    ov = None
    # end-to-end mesh agreement: HIP's extraction field vs the oracle's on the 64^3 sub-block of the 256^3 lattice with the most sign changes
    u = pipeline.extract_mesh(wt, vol, inp["proj"], inp["cam_pos"], 256)[3]
    mf = FU.mesh_field_vs_oracle(ops, wt, vol, u, 256, 64)
    par["mesh_sign_flips"] = mf["mesh_sign_flips"]
    par["mesh_vs_oracle_field"] = mf
    return cpu, par


CURRENT_ROUND = 6


def this_host():
    cpu = "?"
    try:
        cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    except OSError:
        pass
    return {"cpu": cpu, "nproc": os.cpu_count()}


def cpu_reference_file():
    """The reference's OWN modules timed on CPU (tools/time_reference_cpu.py; the GPU box has no /root/reference at bench time, so the number is a
    provenance-stamped file).  Files of THIS round only (profiles/rNN_cpu_reference*.json carry round / commit / date / host / CPU model / core count; a
    stale file is refused with the reason instead of being re-attached to every line forever).  Preferred: the file measured on a box with THIS box's CPU
    model and core count (tools/reference_cpu_on_gpu_box.sh ships an untracked copy of the reference to a GPU box and times it on its host cores --
    SURVEY 8(d)); otherwise the newest file, with `same_host_type: false`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r{CURRENT_ROUND:02d}_cpu_reference*.json")))
    if not files:
        old = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_cpu_reference*.json")))
        why = f"newest is {os.path.relpath(old[-1], ROOT)}" if old else "none found"
        return {"refused": f"no profiles/r{CURRENT_ROUND:02d}_cpu_reference*.json from this round ({why}): re-run tools/time_reference_cpu.py / tools/reference_cpu_on_gpu_box.sh"}
    here = this_host()
    cands = []
    for f in files:
        d = json.load(open(f))
        meta = d.get("_meta") or {}
        if meta.get("round") != f"r{CURRENT_ROUND:02d}":
            continue
        same = meta.get("cpu") == here["cpu"] and meta.get("nproc") == here["nproc"]
        cands.append((same, f, d))
    if not cands:
        return {"refused": "the cpu_reference files of this round carry no matching _meta stamp"}
    cands.sort(key=lambda c: (c[0], c[1]))
    same, f, d = cands[-1]
    return dict(d, source=os.path.relpath(f, ROOT), same_host_type=bool(same), this_host=here)


def merge_cpu_baseline(port, ref):
    """The `cpu_baseline` object of the line.  Preferred: the REFERENCE's own modules (kind "reference") as timed on a host of THIS box's CPU model and
    core count (profiles/rNN_cpu_reference_gpubox.json, tools/reference_cpu_on_gpu_box.sh -- the reference tree does not exist on the box at bench time);
    the oracle port timed live in this run rides along as port_rays_per_s.  When the stamped file is missing, stale or from another host type the live
    port measurement is the value (kind "port") and `reference` says why the reference number was refused."""
    port = port or {}
    if ref and ref.get("value") and ref.get("same_host_type"):
        meta = ref.get("_meta") or {}
        return {"value": ref["value"], "unit": ref.get("unit", "rays/s"), "cores": ref.get("torch_threads") or ref.get("cores"), "torch_threads": ref.get("torch_threads"),
                "host_cores": ref.get("cores"), "kind": "reference", "cpu": meta.get("cpu"), "same_host_type": True, "sample": ref.get("sample"),
                "source": ref.get("source"), "port_rays_per_s": port.get("value"), "port_sample": port.get("sample")}
    why = (ref or {}).get("refused") or (f"{ref.get('source')} was measured on {(ref.get('_meta') or {}).get('cpu')} x{(ref.get('_meta') or {}).get('nproc')}, this box is "
                                        f"{ref.get('this_host')}: {ref.get('value'):.0f} rays/s there" if ref and ref.get("value") else "no stamped reference timing")
    return dict(port, same_host_type=False, reference={"value": None, "refused": why})


def parity_reference_block(dev, precision):
    """HIP vs the REFERENCE's own outputs at BASELINE scale (tests/golden/ref_c2_sample.npz = config 2, ref_c1.npz = config 1 exactly; generated by
    tests/golden/make_golden_scale.py from the imported reference modules): volume build, sampler stage on the reference's inputs, render_core on the
    reference's sample lists, render() end to end, extract_fields.  Test infrastructure on the checker side only (tests/refscale_util.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    out = {}
    try:
        import refscale_util as RU
        for name in ("c2", "c1"):
            out[name] = RU.report(name, dev, precision)
    except FileNotFoundError as e:
        out["refused"] = f"golden file missing: {e}"
    return out


def median_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def ref_config_block(dev, wt):
    """The reference's own configuration (confs/one2345_lod0_val_demo.conf: 32 source views, 96^3 volume; export on a 256^3 grid): the
    only published wall-clock for this path is export_mesh_step = 2.4887 s (example.ipynb:478, authors' GPU, incl. image decode + PLY write)."""
    inp = make_inputs(dev, 32, 0, 1)
    D, R = 96, 256
    st = {}

    def export():
        st["vol"] = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
        st["mesh"] = pipeline.extract_mesh(wt, st["vol"], inp["proj"], inp["cam_pos"], R)
    t_export = median_ms(export)
    t_val = median_ms(lambda: pipeline.render(wt, st["vol"], inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"]), reps=3)
    return {"workload": "V=32 views 256^2, 96^3 volume, 256^3 mesh grid (export_mesh_step body) + one 256x256 val image (64+64 samples)",
            "export_mesh_ms_median": t_export, "val_image_ms_median": t_val, "val_rays_per_s": 65536 / (t_val * 1e-3),
            "kept_voxels": int(st["vol"]["n_voxels"]), "vertices": int(st["mesh"][0].shape[0]), "triangles": int(st["mesh"][1].shape[0]),
            "reference_published_export_mesh_s": 2.4887, "note": "random-weight stand-in scene: its surface (vertex count) is larger than a trained model's"}


def config5_block(dev, wt, a):
    """BASELINE config 5 shape: 256^3 sparse volume, 1024^2 rays, 512^3 mesh grid, one scene."""
    inp = make_inputs(dev, a.views, 0, 4)
    tm = Timer()
    out = None
    for _ in range(2):
        out = None
        out = step(wt, inp, 256, 512, tm, 1 << 20)
    torch.cuda.synchronize(); tm.collect()
    vol, outs, mesh = out
    n = inp["rays_o"].shape[0]
    res = {"workload": f"{a.views} views 256^2, 256^3 volume, {n} rays, 512^3 mesh grid", "volume_build_ms": tm.acc["volume"][-1], "render_ms": tm.acc["render"][-1],
           "mesh_extract_ms": tm.acc["mesh"][-1], "scene_ms": tm.acc["volume"][-1] + tm.acc["render"][-1] + tm.acc["mesh"][-1],
           "render_rays_per_s": n / (tm.acc["render"][-1] * 1e-3), "kept_voxels": int(vol["n_voxels"]), "vertices": int(mesh[0].shape[0]),
           "hbm_peak_gb": torch.cuda.max_memory_allocated() / 1e9}
    return res


def c3_block(dev, wt, inp, a, rank, world):
    """BASELINE config 3: 32 distinct scenes dealt over the ranks (scenes_for_rank: scene k on rank k mod world), each scene's images uploaded
    host -> device INSIDE the step (pinned staging buffer, 6.3 MB), whole scene pass per scene.  Every rank also reports its own clock, scene rate and
    the time its uploads took (HIP events around the copies), so that the first real 8-GPU run attributes a loss to PCIe / host / kernels without a second run."""
    n_scenes = 32
    mine = sharding.scenes_for_rank(n_scenes, rank, world)
    host = [torch.from_numpy(scene_images(a.views, 1000 + k)).pin_memory() for k in mine]
    tm = Timer()
    out = step(wt, inp, a.vol, a.mesh_res, tm, a.ray_chunk)           # warm
    out = None
    tm.collect(); tm.acc = {}
    ev = []
    sharding.barrier(dev)
    t0 = time.perf_counter()
    for h in host:
        out = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        imgs = h.to(dev, non_blocking=True)
        e1.record()
        ev.append((e0, e1))
        out = step(wt, inp, a.vol, a.mesh_res, tm, a.ray_chunk, imgs=imgs)
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0
    sharding.barrier(dev)
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev)
    tm.collect()
    h2d_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
    n_rays = inp["rays_o"].shape[0]
    mine_report = {"rank": rank, "scenes": len(mine), "seconds_own_clock": t_own, "scenes_per_s_own_clock": len(mine) / t_own if t_own > 0 else None,
                   "h2d_ms_per_scene_mean_max": [float(np.mean(h2d_ms)), float(np.max(h2d_ms))] if h2d_ms else None,
                   "h2d_gb_per_s": (host[0].numel() * 4 / 1e9) / (np.mean(h2d_ms) * 1e-3) if h2d_ms else None,
                   "kernel_ms_per_scene": {k: float(np.mean(v)) for k, v in tm.acc.items()}}
    per_rank = sharding.gather_objects(mine_report)
    return {"workload": f"{n_scenes} distinct scenes, {len(mine)} per GPU on {world} GPU(s), images uploaded inside the step",
            "scenes": n_scenes, "scenes_per_s": n_scenes / dt, "rays_per_s": n_scenes * n_rays / dt, "seconds": dt, "h2d_bytes_per_scene": int(host[0].numel() * 4),
            "per_rank": per_rank}


def dropin_block(dev):
    """The reference's OWN timing brackets (trainer_generic.py:1072-1094: "export mesh time", "val_step time") taken on the drop-in surface -- the recon/*
    mirrors + shims driven in the unchanged trainer's call order with its host round trips (tools/dropin_bench.py), at the reference configuration
    (V = 32, 96^3, 256^3 grid, 512-ray chunks).  ``warm``: in this process; ``fresh_process`` x 3: `python tools/dropin_bench.py --cold` three times -- run.py
    starts one process per shape (run.py:61-67), so the first bracket of a fresh process IS the product's latency; the first of the three also fills the
    on-disk cache of packed weights (weights.cached_pack), the others are what every later process on the machine sees (summary: the better of the two;
    the host is shared with other tenants and the trainer's own numpy block varies with their load)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dropin_bench as DB
    out = {"warm": DB.run(dev, reps=5)}
    keep = ("import_torch_ms", "hip_context_ms", "load_library_ms", "construct_networks_ms", "construct_stages_ms", "construct_incl_checkpoint_stand_in_ms", "export_mesh_first_call_ms", "export_mesh_first_call_stages_ms",
            "val_step_first_call_ms", "export_mesh_warm_ms_median", "val_step_warm_ms_median", "val_step_warm_ms_median_per_chunk_calls", "process_total_s", "cpu_threads", "vertices", "error", "rc")
    names = ("fresh_process_1", "fresh_process_2", "fresh_process_3")
    for name in names:
        d = DB.cold_subprocess()
        out[name] = {k: d[k] for k in keep if k in d}
    w = out["warm"]
    firsts = sorted(out[n]["export_mesh_first_call_ms"] for n in names[1:] if out[n].get("export_mesh_first_call_ms"))     # process 1 also fills the disk cache of packed weights
    out["summary"] = {"export_mesh_warm_ms": w["export_mesh_warm_ms_median"], "export_mesh_fresh_process_ms": (firsts[0] if firsts else None),
                      "export_mesh_fresh_process_ms_all": [out[n].get("export_mesh_first_call_ms") for n in names],
                      "construct_networks_ms_all": [out[n].get("construct_networks_ms") for n in names],
                      "val_step_warm_ms": w.get("val_step_warm_ms_median"), "val_step_warm_ms_per_chunk_calls": w.get("val_step_warm_ms_median_per_chunk_calls"),
                      "val_step_with_validate_mesh_360_warm_ms": w.get("val_step_with_validate_mesh_360_ms"),
                      "reference_published_export_mesh_ms": 2488.7,
                      "speedup_vs_published_fresh_process": (2488.7 / firsts[0]) if firsts else None}
    return out


LINE_BUDGET = 4096             # bytes: the driver keeps only the tail of stdout (BENCH_r05.json: a 23 KB line -> `parsed: null`)


def _r(x, n=5):
    """Numbers rounded to n significant digits (the compact line is for a parser and a reader, not for bit-level provenance: that is bench_extra.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, (list, tuple)):
        return [_r(v, n) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, n) for k, v in x.items()}
    return x


def _parity_summary(ref):
    """One max-error number per stage from `parity_reference` (HIP vs the reference's own outputs, tests/golden/ref_c{1,2}*.npz)."""
    out = {}
    for name in ("c1", "c2"):
        p = (ref or {}).get(name)
        if not isinstance(p, dict):
            continue
        rc = (p.get("render_core_on_reference_lists") or [{}])[0]
        e2e = (p.get("render_end_to_end") or [{}])[0]
        ef, vc, vol, smp = p.get("extract_fields") or {}, p.get("vertex_colours") or {}, p.get("volume") or {}, p.get("sampler_on_reference_inputs") or {}
        out[name] = {"mask_bits_exact": vol.get("mask_bits_exact"), "dense_volume": vol.get("dense_volume"), "sampler_dz": smp.get("dz_max"),
                     "sdf": rc.get("sdf"), "grad": rc.get("grad"), "core_color": rc.get("color"), "core_depth": rc.get("depth"), "core_weights": rc.get("weights"),
                     "e2e_color_q99": (e2e.get("color_err_q50_q90_q99_max") or [None] * 4)[2],
                     "ref_self_sensitivity_q99": ((e2e.get("reference_vs_itself_on_a_noisy_volume") or {}).get("color_err_q50_q90_q99_max") or [None] * 4)[2],
                     "field": ef.get("field_err_max"), "sign_flips": ef.get("sign_flips"), "triangles_identical": ef.get("triangles_identical"),
                     "vertex_rgb": vc.get("rgb")}
        out[name] = {k: v for k, v in out[name].items() if v is not None}
    if (ref or {}).get("refused"):
        out["refused"] = ref["refused"]
    return out


def contract_line(result):
    """The ONE line the driver parses: the contract's keys + roofline + cpu_baseline + one parity number per stage, <= LINE_BUDGET bytes.  Everything else
    of `result` (c3, pipelined, trained_regime, ref_config, config5, dropin, parity detail, issue model ...) goes to bench_extra.json and to an EARLIER stdout line."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "shared_gpu_functional_run", "rccl_ranks", "backend")
    line = {k: result[k] for k in keep if k in result}
    cfg = result.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "views", "volume", "rays", "mesh_res", "parallelism", "precision", "variance", "weight_cull", "weights") if k in cfg}
    if len(str(line["config"].get("workload", ""))) > 400:
        line["config"]["workload"] = line["config"]["workload"][:397] + "..."
    for k in ("render_rays_per_s", "render_ms", "mesh_extract_ms", "volume_build_ms"):
        if k in result:
            line[k] = result[k]
    if result.get("per_rank"):                              # N > 1: every rank's own clock (a straggler GPU shows here)
        line["per_rank_ms"] = [r.get("ms_per_step_own_clock") for r in result["per_rank"]]
    if "exhaustive_colour" in result:                       # the reference's work item for item (weight_cull = 0), same scenes and steps
        line["exhaustive"] = {"value": result["exhaustive_colour"]["value"], "ms_per_step": result["exhaustive_colour"]["ms_per_step"]}
    rl = result.get("roofline")
    if rl:
        ev = rl.get("pairs_evaluated_network_pass")
        line["roofline"] = {"kernel": rl["kernel"].split(" ")[0], "bound": rl["bound"], "achieved": rl["achieved"], "peak": rl["peak"], "unit": rl["unit"],
                            "frac": rl["frac"],
                            "frac_on_evaluated_pairs": (rl["frac"] * ev * 32.0 / rl["units"]) if ev and rl.get("units") else None,
                            "ms": rl["ms"], "units": rl.get("units"), "flop_per_unit": rl.get("flop_per_unit"),
                            "mfma_busy_frac": (rl.get("rocprof") or {}).get("mfma_busy_frac"), "traffic": rl.get("traffic"),
                            "algorithmic_bytes": rl.get("algorithmic_bytes")}
    for k in ("roofline_sdf", "roofline_sdf_grad", "roofline_costvol"):
        if k in result:
            line[k] = {"frac": result[k]["frac"], "ms": result[k]["ms"]}
    if "cpu_baseline" in result:
        line["cpu_baseline"] = result["cpu_baseline"]
    dsum = (result.get("dropin") or {}).get("summary")
    if dsum:                                                # the reference's own brackets on the drop-in surface (V = 32 / 96^3 / 256^3): warm, fresh process, chunked val image
        line["dropin"] = {k: dsum.get(k) for k in ("export_mesh_warm_ms", "export_mesh_fresh_process_ms", "val_step_warm_ms", "reference_published_export_mesh_ms")}
    par = _parity_summary(result.get("parity_reference"))
    if par:
        line["parity"] = par
    line["extra"] = "bench_extra.json + the previous stdout line"
    line = _r(line)
    line["value"], line["ms_per_step"] = result["value"], result["ms_per_step"]          # the two contract numbers unrounded
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("parity", "dropin", "exhaustive", "roofline_costvol", "roofline_sdf_grad", "roofline_sdf"):          # never exceed the budget: shed detail, keep the contract
        if len(s) <= LINE_BUDGET:
            break
        line.pop(drop, None)
        line["dropped_for_size"] = line.get("dropped_for_size", []) + [drop]
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_BUDGET, len(s)
    return s


def emit(result):
    """bench_extra.json (next to bench.py; $O2345_BENCH_EXTRA_FILE overrides the path), the full record as an earlier stdout line, the contract line LAST."""
    full = json.dumps(result)
    path = os.environ.get("O2345_BENCH_EXTRA_FILE", os.path.join(ROOT, "bench_extra.json"))       # "" = no file (tests that run bench.py as a subprocess)
    if path:
        try:
            with open(path, "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
    sys.stderr.flush()
    print(full)
    print(contract_line(result), flush=True)


def reexec_under_torchrun(n):
    """`python bench.py --gpus N` invoked bare: start N ranks on this node (one per GPU) through torch.distributed.run."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--vol", type=int, default=128)
    ap.add_argument("--ray-scale", type=int, default=2, help="rays = (256*scale)^2")
    ap.add_argument("--mesh-res", type=int, default=256)
    ap.add_argument("--ray-chunk", type=int, default=1 << 18)
    ap.add_argument("--cpu-rays", type=int, default=8192)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--quick", action="store_true", help="only the contract's line (no c3 / ref_config / config5 / fp32 / dropin blocks)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the `dropin` block (the reference's own timing brackets on the drop-in surface, warm and in fresh processes)")
    ap.add_argument("--same-scene", action="store_true", help="re-reconstruct one scene every step (round-1 behaviour; A/B knob)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for the CPU plumbing test)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous + sharding only, no GPU work (CPU plumbing test)")
    ap.add_argument("--share-gpu", action="store_true", help="FUNCTIONAL TEST ONLY: ranks may share a device (rank r on cuda:r mod count; use with "
                                                             "--backend gloo, RCCL refuses duplicate devices).  The number it prints is not a scaling measurement")
    ap.add_argument("--precision", choices=config.PRECISIONS, default=config.PRECISION,
                    help="network kernels: f16x3 (default; split-f16 MFMA, fp32-class accuracy), fp32 (exact fp32 MFMA)")
    ap.add_argument("--streams", type=int, default=2, help="extra block \"pipelined\": the timed scenes once more on this many streams (0/1 = skip); N = 1 only")
    ap.add_argument("--ckpt", default=None, help="checkpoint in the reference's format (exp_runner...:514-541); default: seeded stand-in weights, identical on every rank")
    ap.add_argument("--broadcast-weights", action="store_true", help="with --ckpt: rank 0 reads the file, ONE RCCL broadcast hands the weights to the other ranks "
                                                                      "(the north star's optional shared-backbone broadcast); default: every rank reads the file")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        reexec_under_torchrun(a.gpus)
    rank, world, local = sharding.init(a.backend)            # RCCL ("nccl") when WORLD_SIZE > 1; only used for the clock
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    if a.dry_run:
        # what the N > 1 path adds to the single-GPU path, without a GPU: rendezvous, scene deal, barrier, max-over-ranks clock, one line
        mine = [rank + world * k for k in range(a.steps)]
        sharding.barrier()
        dt = sharding.max_over_ranks(0.01 * (rank + 1))
        tot = sharding.sum_over_ranks(len(mine))
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": a.steps, "scenes_total": tot, "slowest_rank_s": dt, "rank0_scenes": mine}))
        sharding.shutdown()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the reconstruction path has no CPU fallback)"
    if a.share_gpu:
        local = local % torch.cuda.device_count()
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} device(s) are visible")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 and not a.share_gpu:                        # every rank on its own GPU
        ids = [None] * world
        torch.distributed.all_gather_object(ids, (os.uname().nodename, local))
        assert len(set(ids)) == world, f"ranks share a device: {ids}"
    if a.ckpt:
        wt = pipeline.SceneWeights.from_checkpoint(dev, a.ckpt, broadcast=a.broadcast_weights)
    else:
        wt = pipeline.SceneWeights(dev, seed=0, sdf_precision=a.precision, color_precision=a.precision)
    wt.grid_tables(a.mesh_res)      # per-(network, resolution) tables of the lattice SDF kernel: part of loading the weights, like the operand blobs
    inp = make_inputs(dev, a.views, seed=rank, ray_scale=a.ray_scale)
    # distinct scene per step: scene index = rank + world * step (images resident in HBM before the timed region starts)
    n_total = a.warmup + a.steps
    scene_imgs = [inp["imgs"] if a.same_scene else torch.from_numpy(scene_images(a.views, rank + world * k)).to(dev) for k in range(n_total)]
    tm = Timer()
    vol = outs = mesh = None
    for k in range(a.warmup):
        vol = outs = mesh = None                 # release the previous scene's outputs first: every step then reuses the same
        vol, outs, mesh = step(wt, inp, a.vol, a.mesh_res, tm, a.ray_chunk, imgs=scene_imgs[k])      # cached blocks (no hipMalloc inside a timed step)
    tm.collect(); tm.acc = {}

    # a full (generation-2) Python GC pass over the ~10^6 objects that `import torch` creates takes 30-40 ms and would land
    # inside one of the timed steps; collect now and freeze the survivors (standard practice for latency-sensitive services)
    import gc
    gc.collect()
    gc.freeze()
    sharding.barrier(dev)
    t0 = time.perf_counter()
    alloc_log = []
    for k in range(a.steps):
        vol = outs = mesh = None
        vol, outs, mesh = step(wt, inp, a.vol, a.mesh_res, tm, a.ray_chunk, imgs=scene_imgs[a.warmup + k])
        if os.environ.get("O2345_BENCH_VERBOSE"):
            st = torch.cuda.memory_stats()
            alloc_log.append((st["num_device_alloc"], st["num_device_free"], round((time.perf_counter() - t0) * 1e3, 1)))
    t_own = time.perf_counter() - t0                         # this rank's own clock, before it waits for the others
    sharding.barrier(dev)
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev)
    tm.collect()
    # per-rank view of the same K steps (a straggler GPU shows up here the first time --gpus N runs on real hardware): gathered as small objects
    props = torch.cuda.get_device_properties(dev)
    per_rank = sharding.gather_objects({
        "rank": rank, "device": f"cuda:{local}", "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
        "host": os.uname().nodename, "ms_per_step_own_clock": t_own / a.steps * 1e3,
        "step_ms_min_median_max": [float(np.min(s_)), float(np.median(s_)), float(np.max(s_))] if (s_ := [sum(x) for x in zip(tm.acc["volume"], tm.acc["render"], tm.acc["mesh"])]) else None})
    if os.environ.get("O2345_BENCH_VERBOSE"):
        print({k: [round(x, 2) for x in v] for k, v in tm.acc.items()}, file=sys.stderr)
        print("per step (device allocs, frees, host ms since start):", alloc_log, file=sys.stderr)
        st = torch.cuda.memory_stats()
        print({k: st[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.peak", "allocated_bytes.all.peak")}, file=sys.stderr)
    n_rays = inp["rays_o"].shape[0]
    ms_step = dt / a.steps * 1e3
    c3 = piped = None
    exhaustive = None
    if world == 1 and config.weight_cull() > 0:
        # the same K scenes once more with the colour network on EVERY occupied sample (weight_cull = 0: the reference's work, the value of rounds 1-4):
        # the contract's `value` above runs with the tolerance-bounded removal (a ray's colour moves by <= 7.6e-6); both are on the line
        old_cull, config.WEIGHT_CULL = config.WEIGHT_CULL, 0.0
        try:
            tx = Timer()
            v_ = o_ = m_ = None
            v_, o_, m_ = step(wt, inp, a.vol, a.mesh_res, tx, a.ray_chunk, imgs=scene_imgs[0])
            tx.collect(); tx.acc = {}
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for k in range(a.steps):
                v_ = o_ = m_ = None
                v_, o_, m_ = step(wt, inp, a.vol, a.mesh_res, tx, a.ray_chunk, imgs=scene_imgs[a.warmup + k])
            torch.cuda.synchronize(dev)
            ms_x = (time.perf_counter() - t1) / a.steps * 1e3
            tx.collect()
            exhaustive = {"weight_cull": 0.0, "ms_per_step": ms_x, "value": n_rays / (ms_x * 1e-3), "render_ms": tx.mean("render"),
                          "note": "same scenes, same steps; colour network on every occupied sample like the reference (O2345_WEIGHT_CULL=0)"}
            v_ = o_ = m_ = None
        finally:
            config.WEIGHT_CULL = old_cull
    if not a.quick and world == 1 and a.streams > 1 and not a.same_scene and not a.ckpt:
        vol_keep = (vol, outs, mesh)
        piped = pipelined_block(dev, a, inp, scene_imgs[a.warmup:], a.streams)
        vol, outs, mesh = vol_keep
    if not a.quick:
        scene_imgs = None
        c3 = c3_block(dev, wt, inp, a, rank, world)          # all ranks take part (barrier + max-over-ranks clock inside)
    result = None
    if rank == 0:
        kt = kernel_times(wt, vol, inp, outs, a.vol)
        V, C = a.views, 16
        n_vox = int(vol["n_voxels"])
        cv_bytes = V * C * 256 * 256 * 4 + n_vox * (2 * C * 4 + 16) + a.vol ** 3
        nvp, npts = kt["n_valid_points"], kt["n_points"]
        rl = network_rooflines(kt, V, wt.sdf_precision, wt.color_precision)
        dtype = {"f16x3": "f32 (matrix products as 3 f16 MFMAs on split operands, fp32 accumulate)", "fp32": "f32"}[a.precision]
        result = {
            "metric": "rays/sec + mesh-extract wall-clock per scene (8x256^2 views, 128^3 vol)", "value": world * n_rays / (ms_step * 1e-3),
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            **({"shared_gpu_functional_run": True} if a.share_gpu else {}),
            **({"rccl_ranks": world, "backend": torch.distributed.get_backend(), "per_rank": per_rank,
                "weights": ("checkpoint, broadcast from rank 0" if a.broadcast_weights else "checkpoint, read by every rank") if a.ckpt else "seeded stand-ins, built per rank"}
               if world > 1 else {}),
            "config": {"workload": f"BASELINE config 2: 1 scene/GPU/step ({'the same scene' if a.same_scene else 'a different seeded scene'} every step), "
                                   f"{a.views} views 256x256, {a.vol}^3 volume, "
                                   f"{n_rays} rays (64+64 samples), mesh grid {a.mesh_res}^3; whole scene pass per step",
                       "views": a.views, "volume": a.vol, "rays": n_rays, "mesh_res": a.mesh_res, "parallelism": f"scenes x{world}",
                       "precision": a.precision, "variance": wt.variance, "inv_s": wt.inv_s, "weight_cull": config.weight_cull(),
                       "weights": "checkpoint" if a.ckpt else "seeded stand-ins (untrained: variance 0.2 -> inv_s 7.4; the trained regime is the `trained_regime` block)"},
            "render_rays_per_s": n_rays / (tm.mean("render") * 1e-3), "mesh_extract_ms": tm.mean("mesh"),
            "volume_build_ms": tm.mean("volume"), "render_ms": tm.mean("render"),
            "mesh": {"vertices": int(mesh[0].shape[0]), "triangles": int(mesh[1].shape[0])}, "kept_voxels": n_vox,
            "occupied_points": nvp, "sampled_points": npts,
            "list_sort_ms": kt.get("list_sort_ms"),      # csrc/list_sort.hip: the occupied-point list grouped by view-visibility signature (inside every render call)
            # dominant kernel of a step = the colour network: ALGORITHMIC FLOP (38,544 per (point, view)) x occupied
            # points x views / HIP-event time of that launch, against the dense MFMA peak of the type the matrix pipe runs in
            # algorithmic_bytes of the colour kernel = every input byte once: 28 B per occupied point (xyz, list index, rgb out) + the latent volume + mask + colour/feature maps
            "roofline": dict(rl["color"], traffic=pmc_traffic(COLOR_KERNEL_PREFIX),
                             algorithmic_bytes=int(nvp * 28 + vol["vol_cl"].numel() * vol["vol_cl"].element_size() + vol["maskvol"].numel() * vol["maskvol"].element_size()
                                                   + vol["cmaps"].numel() * vol["cmaps"].element_size()),
                             traffic_source=(f"{PMC_FILE} (bytes, 2*FETCH_SIZE+WRITE_SIZE; collected on these kernel sources)" if PMC_DATA is not None else PMC_STALE),
                             issue_model=pmc_issue_model(COLOR_KERNEL_PREFIX, kt["color_ms"]), rocprof=pmc_mfma_busy(COLOR_KERNEL_PREFIX + "<true")),
            "roofline_sdf": dict(rl["sdf"], rocprof=pmc_mfma_busy("k_sdf_mlp_x3<false>" if wt.sdf_precision == "f16x3" else "k_sdf_mlp<0>"),
                                 traffic=pmc_traffic("k_sdf_mlp_x3<false>" if wt.sdf_precision == "f16x3" else "k_sdf_mlp<0>")),
            "roofline_sdf_grad": dict(rl["sdf_grad"], rocprof=pmc_mfma_busy("k_sdf_grad_x3" if wt.sdf_precision == "f16x3" else "k_sdf_mlp<2>"),
                                      traffic=pmc_traffic("k_sdf_grad_x3" if wt.sdf_precision == "f16x3" else "k_sdf_mlp<2>")),
            "roofline_costvol": {"kernel": "k_costvol_gather<16>", "bound": "hbm", "achieved": cv_bytes / (kt["costvol_gather_ms"] * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cv_bytes / (kt["costvol_gather_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic("k_costvol_gather"), "algorithmic_bytes": cv_bytes, "ms": kt["costvol_gather_ms"]},
        }
        if c3 is not None:
            result["c3"] = c3
        if piped is not None:
            result["pipelined"] = piped
        if exhaustive is not None:
            result["exhaustive_colour"] = exhaustive
        if a.precision != "fp32":
            # the same three kernels in the exact fp32 MFMA form, priced against the fp32 matrix peak (strict mode of the library)
            kf = kernel_times(wt, vol, inp, outs, a.vol, reps=3, sdf_precision="fp32", color_precision="fp32")
            result["roofline_fp32_mode"] = network_rooflines(kf, V, "fp32", "fp32")
        if world == 1 and not a.no_cpu:
            vol0 = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], a.vol, 2.0 / (a.vol - 1))     # the scene cpu_baseline's images belong to
            port, result["parity_fullsize"] = cpu_baseline_and_parity(wt, vol0, inp, a.vol, a.cpu_rays)
            result["cpu_baseline_reference"] = cpu_reference_file()
            result["cpu_baseline_port"] = port
            result["cpu_baseline"] = merge_cpu_baseline(port, result["cpu_baseline_reference"])
            result["parity_reference"] = parity_reference_block(dev, a.precision)
            vol0 = None
        if world == 1 and not a.quick:
            vol = outs = mesh = None
            if a.precision != "fp32":
                wf = pipeline.SceneWeights(dev, seed=0, sdf_precision="fp32", color_precision="fp32")
                tf = Timer()
                result["fp32_whole_step_ms"] = median_ms(lambda: step(wf, inp, a.vol, a.mesh_res, tf, a.ray_chunk), reps=3)
                wf = None
            result["trained_regime"] = trained_regime_block(dev, wt, inp, a.vol)
            result["ref_config"] = ref_config_block(dev, wt)
            torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
            result["config5"] = config5_block(dev, wt, a)
            wt = None
            torch.cuda.empty_cache()
            if not a.no_dropin:
                result["dropin"] = dropin_block(dev)
    sharding.shutdown()
    if result is not None:
        emit(result)                                # nothing may follow the contract line on stdout


if __name__ == "__main__":
    main()
