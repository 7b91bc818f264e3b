#!/usr/bin/env python
"""The reference's OWN timing brackets, taken on the drop-in surface (measurement harness; imported by bench.py, runnable as a script).

The only wall-clock the reference publishes for this path is the ``print("export mesh time: ", ...)`` around ``GenericTrainer.export_mesh_step``
(models/trainer_generic.py:1086-1094; 2.4887 s in example.ipynb:478) and ``val_step time`` (:1072-1083).  ``bench.py``'s contract line times the fused
``pipeline.*`` calls; THIS file times what the UNCHANGED trainer does on top of the mirrors (``recon/*``) and shims: the same objects, the same
call order, the same host round trips (numpy returns, ``torch.tensor(vertices)``, the trainer's own numpy frame transforms, ``trimesh`` export,
512-ray chunks with ``.cpu().numpy()`` per chunk).  ``/root/reference`` does not exist on the GPU box, so ``TrainerLike`` below RESTATES the trainer's
control flow (each block cites the lines it follows); the real, unchanged ``GenericTrainer`` is run through the same mirrors by
``tests/test_trainer_dropin.py`` (CPU stand-ins) and ``tools/runner_on_gpu.sh`` (MI355X, untracked working copy of the reference).

Reference configuration (confs/one2345_lod0_val_demo.conf): V = 32 source views 256^2, 96^3 volume, export on a 256^3 grid, 64 + 64 samples.

  python tools/dropin_bench.py                 # warm numbers + stage decomposition, JSON on stdout
  python tools/dropin_bench.py --cold          # what a FRESH process pays (run.py:61-67 starts one per shape): import / lib load / first calls
"""
import argparse
import importlib
import json
import os
import sys
import time

T_START = time.perf_counter()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class Stages:
    """Named host-clock sections.  ``sync=True`` brackets every section with a device synchronisation (decomposition run); ``sync=False`` only
    accumulates host time between the marks (the headline run must not add synchronisations the trainer does not have)."""

    def __init__(self, sync):
        import torch
        self.torch, self.sync, self.acc, self._t = torch, sync, {}, None

    def mark(self, name=None):
        if self.sync:
            self.torch.cuda.synchronize()
        now = time.perf_counter()
        if self._name is not None:
            self.acc[self._name] = self.acc.get(self._name, 0.0) + (now - self._t) * 1e3
        self._name, self._t = name, now

    _name = None

    def ms(self):
        return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in self.acc.items()}


class Conf(dict):
    def get_int(self, k, default=None):
        return int(self.get(k, default))


class TrainerLike:
    """The members and methods of GenericTrainer that run.py / the val configuration reach (trainer_generic.py:29-118 for the members), lod 0 only
    (``num_lods = 1`` in confs/one2345_lod0_val_demo.conf)."""

    def __init__(self, dev, D=96, out_dir="/tmp/o2345_dropin", variance=0.3, seed=0):
        """Same order as Runner.__init__ (exp_runner_generic_blender_val.py:93-160, 485-512): networks built from the conf's kwargs, moved to the device,
        THEN the checkpoint's state dicts loaded into them (the synthetic checkpoint of tests/run_reference_runner.py: seeded mirrors with non-zero
        latent columns, so that the volume matters) -- the weights reach the kernels' packed form in the load hooks, before any timed call, exactly as
        they do under the unchanged runner."""
        import torch
        recon = importlib.import_module("one-2-3-45_amd.recon")
        fn = importlib.import_module("one-2-3-45_amd.featurenet")
        self.torch, self.dev = torch, dev

        def nets():
            return {"pyramid_feature_network": fn.FeatureNet(),
                    "sdf_network_lod0": recon.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=2.0 / (D - 1), vol_dims=[D, D, D], hidden_dim=128, cost_type="variance_mean",
                                                               d_pyramid_feature_compress=16, regnet_d_out=16, num_sdf_layers=4, multires=6),
                    "variance_network_lod0": recon.SingleVarianceNetwork(0.3), "rendering_network_lod0": recon.GeneralRenderingNetwork(16, 56, True)}
        self.construct_stages_ms = cs = {}
        t_ = [time.perf_counter()]

        def lap(name):
            if dev.type == "cuda":
                torch.cuda.synchronize()
            now = time.perf_counter()
            cs[name] = round(cs.get(name, 0.0) + (now - t_[0]) * 1e3, 3)
            t_[0] = now
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            ck = nets()
            lap("checkpoint_stand_in_modules_cpu")
            g = torch.Generator().manual_seed(seed + 1)
            L = ck["sdf_network_lod0"].sdf_layer
            L.lin1.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
            L.lin2.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
            ck["variance_network_lod0"].variance.data = torch.tensor(float(variance))
            ckpt = {k: {kk: vv.clone() for kk, vv in n.state_dict().items()} for k, n in ck.items()}
            lap("checkpoint_stand_in_state_dicts")
            mine = nets()
            lap("build_modules_cpu")
            mine = {k: n.to(dev) for k, n in mine.items()}
            lap("modules_to_device")
        for k, n in mine.items():
            n.load_state_dict(ckpt[k])
            lap("load_state_dict+prepack:" + k)
        self.pyramid_feature_network_geometry_lod0 = mine["pyramid_feature_network"]
        self.sdf_network_lod0, self.rendering_network_lod0 = mine["sdf_network_lod0"], mine["rendering_network_lod0"]
        self.variance_network_lod0 = mine["variance_network_lod0"]
        self.base_exp_dir = out_dir
        os.makedirs(out_dir, exist_ok=True)
        self.sdf_renderer_lod0 = recon.SparseNeuSRenderer(None, self.sdf_network_lod0, self.variance_network_lod0, self.rendering_network_lod0,
                                                          64, 64, 0, 1.0, alpha_type="div", conf=Conf({"general.base_exp_dir": out_dir}))
        self.n_samples_lod0, self.n_importance_lod0 = 64, 64
        lap("renderer")

    # trainer_generic.py:1104-1125
    def obtain_pyramid_feature_maps(self, imgs):
        F = self.torch.nn.functional
        p = self.pyramid_feature_network_geometry_lod0(imgs)
        return self.torch.cat([F.interpolate(p[0], scale_factor=4, mode="bilinear", align_corners=True),
                               F.interpolate(p[1], scale_factor=2, mode="bilinear", align_corners=True), p[2]], dim=1)

    def _volume(self, sample, st):
        """The head both steps share (:369-435 / :835-897)."""
        torch = self.torch
        st.mark("unpack_sample")
        sizeW, sizeH = sample["img_wh"][0][0], sample["img_wh"][0][1]
        imgs = sample["images"][0]
        intrinsics_l_4x = sample["intrinsics"][0].clone()                                       # :853-854 (unused afterwards in the released configuration, but it runs)
        intrinsics_l_4x[:, :2] *= 0.25
        true_img = np.uint8(sample["query_image"][0].permute(1, 2, 0).cpu().numpy() * 255)      # noqa: F841  (the trainer converts it before anything else)
        st.mark("featurenet_pyramid")
        with torch.no_grad():
            fmaps = self.obtain_pyramid_feature_maps(imgs)
            st.mark("get_conditional_volume")
            cf = self.sdf_network_lod0.get_conditional_volume(feature_maps=fmaps[None], partial_vol_origin=sample["partial_vol_origin"],
                                                              proj_mats=sample["affine_mats"], sizeH=sizeH, sizeW=sizeW, lod=0)
        return imgs, fmaps, cf["dense_volume_scale0"], cf["valid_mask_volume_scale0"]

    # trainer_generic.py:827-979 with validate_colored_mesh :1309-1382 inlined
    def export_mesh_step(self, sample, resolution=256, st=None):
        torch = self.torch
        st = st or Stages(False)
        trimesh = importlib.import_module("one-2-3-45_amd.shims.trimesh")
        imgs, fmaps, vol, mask = self._volume(sample, st)
        st.mark("empty_cache")
        torch.cuda.empty_cache()
        st.mark("extract_geometry")
        bmin, bmax = torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3)
        vertices, triangles, _fields = self.sdf_renderer_lod0.extract_geometry(self.sdf_network_lod0, bmin, bmax, resolution=resolution, threshold=0,
                                                                               device=vol.device, conditional_volume=vol, lod=0, occupancy_mask=None)
        with torch.no_grad():
            st.mark("vertices_to_device")
            vt = torch.tensor(vertices).to(vol)
            st.mark("compute_view_independent")
            a, b, c, d, _, _ = self.sdf_renderer_lod0.rendering_projector.compute_view_independent(
                vt, lod=0, geometryVolume=vol[0], geometryVolumeMask=mask[0], sdf_network=self.sdf_network_lod0, rendering_feature_maps=fmaps,
                color_maps=imgs, w2cs=sample["w2cs"][0], target_candidate_w2cs=None, intrinsics=sample["intrinsics"][0], img_wh=[256, 256],
                query_img_idx=0, query_c2w=sample["query_c2w"])
            st.mark("rendering_network")
            vertices_color, _valid = self.rendering_network_lod0(a, b, c, d)
        st.mark("host_frame_transforms")
        fine = os.environ.get("O2345_DROPIN_FINE_TIMERS")
        sub = (lambda n: st.mark("host_frame_transforms:" + n)) if fine else (lambda n: None)
        sub("scale_mat.cpu")
        sm = sample["scale_mat"].cpu().numpy()
        sub("scale+shift")
        vertices = vertices * sm[0][0, 0] + sm[0][:3, 3][None]
        sub("trans_mat.cpu")
        tm = sample["trans_mat"].cpu().numpy()
        sub("concatenate")
        vh = np.concatenate([vertices, np.ones_like(vertices[:, :1])], axis=1)
        sub("matmul")
        vertices = np.matmul(tm, vh[:, :, None])[:, :3, 0]
        st.mark("colours_to_host")
        vertices_color = np.array(vertices_color.squeeze(0).cpu() * 255, dtype=np.uint8)
        st.mark("trimesh_export")
        mesh = trimesh.Trimesh(vertices, triangles, vertex_colors=vertices_color)
        mesh.export(os.path.join(self.base_exp_dir, "mesh.ply"))
        st.mark("empty_cache")
        torch.cuda.empty_cache()
        st.mark(None)
        return vertices.shape[0], triangles.shape[0]

    # trainer_generic.py:359-622 (save_vis = True, num_lods = 1; the PNG writes of save_visualization are cv2 calls of the reference, not ours)
    def val_step(self, sample, chunk_size=512, st=None, background_rgb=None, alpha_inter_ratio=1.0, mesh_resolution=None):
        torch = self.torch
        st = st or Stages(False)
        imgs, fmaps, vol, mask = self._volume(sample, st)
        st.mark("split_rays")
        near, far = sample["query_near_far"][0, :1], sample["query_near_far"][0, 1:]
        rays_o = sample["rays"]["rays_o"][0].reshape(-1, 3).split(chunk_size)
        rays_d = sample["rays"]["rays_v"][0].reshape(-1, 3).split(chunk_size)
        rgb, nrm, dep = [], [], []
        S = self.n_samples_lod0 + self.n_importance_lod0
        w2cs, intrinsics, query_c2w = sample["w2cs"][0], sample["intrinsics"][0], sample["query_c2w"]      # sliced once, before the loop (:385-400)
        for ro, rd in zip(rays_o, rays_d):
            st.mark("render_chunks")
            out = self.sdf_renderer_lod0.render(ro, rd, near, far, self.sdf_network_lod0, self.rendering_network_lod0, background_rgb=background_rgb,
                                                alpha_inter_ratio=alpha_inter_ratio, lod=0, conditional_volume=vol, conditional_valid_mask_volume=mask,
                                                feature_maps=fmaps, color_maps=imgs, w2cs=w2cs, intrinsics=intrinsics,
                                                img_wh=[256, 256], query_c2w=query_c2w, if_render_with_grad=False)
            st.mark("chunk_outputs_to_host")
            dep.append(out["depth"].detach().cpu().numpy())
            rgb.append(out["color_fine"].detach().cpu().numpy())
            nrm.append((out["gradients"] * out["weights"][:, :S, None] * out["inside_sphere"][..., None]).sum(dim=1).detach().cpu().numpy())
            del out
        st.mark("concat_image")
        img = np.concatenate(rgb, axis=0)
        depth = np.concatenate(dep, axis=0)
        normals = np.concatenate(nrm, axis=0)
        res = {"color": img, "depth": depth, "normals": normals}
        if mesh_resolution:                                           # validate_mesh (:1255-1303): geometry only, no colours, frame transforms, PLY
            trimesh = importlib.import_module("one-2-3-45_amd.shims.trimesh")
            st.mark("validate_mesh")
            torch.cuda.empty_cache()
            v, t, _ = self.sdf_renderer_lod0.extract_geometry(self.sdf_network_lod0, torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), resolution=mesh_resolution,
                                                              threshold=0, device=vol.device, conditional_volume=vol, lod=0, occupancy_mask=None)
            sm = sample["scale_mat"].cpu().numpy()
            v = v * sm[0][0, 0] + sm[0][:3, 3][None]
            tm = sample["trans_mat"].cpu().numpy()
            v = np.matmul(tm, np.concatenate([v, np.ones_like(v[:, :1])], axis=1)[:, :, None])[:, :3, 0]
            os.makedirs(os.path.join(self.base_exp_dir, "meshes_val_bg"), exist_ok=True)
            trimesh.Trimesh(v, t).export(os.path.join(self.base_exp_dir, "meshes_val_bg", "mesh_lod0.ply"))
            torch.cuda.empty_cache()
        st.mark(None)
        return res


def make_sample(dev, V=32, seed=0):
    """The sample dict as the runner hands it to the trainer: default collate (batch 1) of the dataset's item, moved to the device
    (exp_runner_generic_blender_val.py:563-566 ``tocuda``); SURVEY 3.5."""
    import torch
    synth = importlib.import_module("one-2-3-45_amd.synth")
    sc = synth.make_scene(V, image_seed=seed)
    ro, rd = synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].to(dev)
    rng = np.random.default_rng(seed + 7)
    return {"images": T(sc["images"]), "intrinsics": T(sc["intrinsics"]), "w2cs": T(sc["w2cs"]), "c2ws": T(sc["c2ws"]), "affine_mats": T(sc["affine_mats"]),
            "partial_vol_origin": T(sc["partial_vol_origin"]), "query_near_far": T(sc["query_near_far"]), "query_c2w": T(sc["query_c2w"]),
            "query_w2c": T(sc["query_w2c"]), "scale_mat": T(sc["scale_mat"]), "trans_mat": T(sc["trans_mat"]),
            "query_image": T(rng.random((3, 256, 256), dtype=np.float32)), "rays": {"rays_o": T(ro), "rays_v": T(rd)},
            "img_wh": torch.tensor([[256, 256]]), "meta": ["scene0_refview0"], "batch_idx": torch.tensor([0]), "scale_factor": T(np.float32(1.0))}


def _bracket(fn, reps):
    """-> (list of wall-clock ms of ``fn`` exactly as the trainer's ``time.time()`` pair sees it: no synchronisation added)."""
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


def run(dev, reps=3, resolution=256, cold=False, val=True, out_dir="/tmp/o2345_dropin"):
    """-> dict for bench.py's ``dropin`` block.  ``cold``: the caller is a fresh process; the first bracket is reported separately.
    The host-side tensor ops of the trainer run with the intra-op thread count the drop-in launcher sets (dropin.cpu_threads(): min(8, cores) unless
    $O2345_CPU_THREADS says otherwise) -- restored afterwards."""
    import torch
    threads = importlib.import_module("one-2-3-45_amd.dropin").cpu_threads()
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        res = _run(dev, reps, resolution, cold, val, out_dir)
    finally:
        torch.set_num_threads(old_threads)
    res["cpu_threads"] = threads
    res["cpu_cores"] = os.cpu_count()
    return res


def _run(dev, reps, resolution, cold, val, out_dir):
    import torch
    res = {"workload": "reference configuration: V=32 views 256^2, 96^3 volume, 256^3 extraction grid, 64+64 samples; the trainer's own call order on recon/* + shims",
           "reference_published_export_mesh_s": 2.4887, "reference_bracket": "models/trainer_generic.py:1086-1094 (export mesh time), :1072-1083 (val_step time)"}
    t0 = time.perf_counter()
    tr = TrainerLike(dev, out_dir=out_dir)
    torch.cuda.synchronize()
    # what the runner does (exp_runner_generic_blender_val.py:93-160, 485-512): build the modules, move them to the device, load_state_dict (-> prepack hooks),
    # build the renderer.  Building the synthetic CHECKPOINT (a second set of modules on the CPU + their state dicts) stands in for torch.load of a file and is
    # reported separately.
    res["construct_stages_ms"] = tr.construct_stages_ms
    res["construct_incl_checkpoint_stand_in_ms"] = (time.perf_counter() - t0) * 1e3
    res["construct_networks_ms"] = sum(v for k, v in tr.construct_stages_ms.items() if not k.startswith("checkpoint_stand_in"))
    t0 = time.perf_counter()
    sample = make_sample(dev)
    torch.cuda.synchronize()
    res["make_sample_ms"] = (time.perf_counter() - t0) * 1e3
    if cold:
        import gc
        st = Stages(True)
        g0 = [s_["collections"] for s_ in gc.get_stats()]
        t0 = time.perf_counter()
        nv, nt = tr.export_mesh_step(sample, resolution, st)
        res["export_mesh_first_call_ms"] = (time.perf_counter() - t0) * 1e3
        res["export_mesh_first_call_gc_collections_gen0_1_2"] = [s_["collections"] - a for s_, a in zip(gc.get_stats(), g0)]
        res["export_mesh_first_call_stages_ms"] = st.ms()
    else:
        nv, nt = tr.export_mesh_step(sample, resolution)
    res["vertices"], res["triangles"] = int(nv), int(nt)
    ts = _bracket(lambda: tr.export_mesh_step(sample, resolution), reps)
    res["export_mesh_warm_ms"] = [round(t, 2) for t in ts]
    res["export_mesh_warm_ms_median"] = float(np.median(ts))
    st = Stages(True)
    tr.export_mesh_step(sample, resolution, st)
    res["export_mesh_warm_stages_ms"] = st.ms()
    res["speedup_vs_published_warm"] = 2488.7 / res["export_mesh_warm_ms_median"]
    if val:
        if cold:
            st = Stages(True)
            t0 = time.perf_counter()
            tr.val_step(sample, st=st)
            res["val_step_first_call_ms"] = (time.perf_counter() - t0) * 1e3
            res["val_step_first_call_stages_ms"] = st.ms()
        else:
            tr.val_step(sample)
        ts = _bracket(lambda: tr.val_step(sample), reps)
        res["val_step_warm_ms"] = [round(t, 2) for t in ts]
        res["val_step_warm_ms_median"] = float(np.median(ts))
        st = Stages(True)
        tr.val_step(sample, st=st)
        res["val_step_warm_stages_ms"] = st.ms()
        res["val_rays_per_s_chunked"] = 65536 / (res["val_step_warm_ms_median"] * 1e-3)
        # A/B: the same loop with the whole-image mode off (O2345_WHOLE_IMAGE=0): every 512-ray chunk is its own render call of 16 launches (round 4's path)
        tr.sdf_renderer_lod0.whole_image = False
        try:
            tr.val_step(sample)
            ts = _bracket(lambda: tr.val_step(sample), reps)
        finally:
            tr.sdf_renderer_lod0.whole_image = True
        res["val_step_warm_ms_median_per_chunk_calls"] = float(np.median(ts))
        res["val_step_note"] = ("the trainer's unchanged loop over 128 chunks of 512 rays (trainer_generic.py:503-524); whole-image mode: the first render() call "
                                "renders every chunk in four fused segmented calls on a side stream, the other 127 calls return slices")
        # what the runner's "val_step time" additionally contains: validate_mesh at its default resolution of 360 (trainer_generic.py:1271, :598-606)
        tr.val_step(sample, mesh_resolution=360)
        st = Stages(True)
        t0 = time.perf_counter()
        tr.val_step(sample, st=st, mesh_resolution=360)
        res["val_step_with_validate_mesh_360_ms"] = (time.perf_counter() - t0) * 1e3
        res["val_step_with_validate_mesh_360_stages_ms"] = st.ms()
    return res


def cold_process(profile=False):
    """Everything a fresh process pays before and inside its first bracket (run.py:61-67 spawns one process per shape)."""
    res = {"python_start_to_main_ms": (time.perf_counter() - T_START) * 1e3}
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):           # what `python -m o2345_amd.dropin <script>` does before torch is imported
        os.environ.setdefault(var, str(importlib.import_module("one-2-3-45_amd.dropin").cpu_threads()))
    res["malloc_tuned"] = importlib.import_module("one-2-3-45_amd.dropin").tune_host_allocator()      # as the launcher does (dropin.main)
    t0 = time.perf_counter()
    import torch
    res["import_torch_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    importlib.import_module("one-2-3-45_amd.recon")
    res["import_package_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    importlib.import_module("one-2-3-45_amd._lib").lib()
    res["load_library_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    res["hip_context_ms"] = (time.perf_counter() - t0) * 1e3
    if profile:                                   # host-side view of the FIRST export_mesh_step of the process -> stderr
        import cProfile
        import pstats
        tr, sample = TrainerLike(dev), make_sample(dev)
        pr = cProfile.Profile()
        pr.enable(); tr.export_mesh_step(sample); pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(70)
        pr = cProfile.Profile()
        pr.enable(); tr.val_step(sample); pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(50)
        return res
    res.update(run(dev, reps=3, cold=True))
    res["process_total_s"] = time.perf_counter() - T_START
    return res


def cold_subprocess(timeout=600):
    """bench.py side: run ``--cold`` in a fresh interpreter, return its JSON (or the error)."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    for line in reversed(p.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": (p.stderr or p.stdout)[-2000:], "rc": p.returncode}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-val", action="store_true")
    ap.add_argument("--profile", action="store_true", help="cProfile of one warm val_step and one warm export_mesh_step (host side) -> stderr")
    a = ap.parse_args()
    if a.cold:
        print(json.dumps(cold_process(a.profile)))
        return
    import torch
    dev = torch.device("cuda:0")
    res = run(dev, reps=a.reps, val=not a.no_val)
    if a.profile:
        import cProfile
        import pstats
        tr = TrainerLike(dev)
        sample = make_sample(dev)
        tr.export_mesh_step(sample); tr.val_step(sample)
        for name, fn in (("val_step", lambda: tr.val_step(sample)), ("export_mesh_step", lambda: tr.export_mesh_step(sample))):
            pr = cProfile.Profile()
            pr.enable(); fn(); pr.disable()
            print(f"==== cProfile {name}", file=sys.stderr)
            pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
