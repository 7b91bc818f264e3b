"""Times the REFERENCE's own modules (imported from /root/reference, CPU, stubs per oracle/ref_import.py) on BASELINE config-2 inputs:
SparseNeuSRenderer.render (512-ray chunks, like the runner) and extract_fields.  Needs the reference tree: /root/reference in the build container, or
O2345_REFERENCE_DIR = an untracked working copy shipped to the GPU box (tools/reference_cpu_on_gpu_box.sh -- SURVEY 8(d): "the reference's CPU path timed
on the host cores of the same box").  Writes <out dir>/rNN_cpu_reference[_gpubox].json stamped with the commit, date, host, CPU model and core count it
was measured on; bench.py attaches the file whose CPU model and core count are those of the box it runs on (else the newest, with the mismatch stated)
and refuses files of an older round.

    python tools/time_reference_cpu.py [seconds] [round tag, default r06] [out dir, default profiles] [file suffix, default ""]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recon as O  # noqa: E402
from oracle import ref_import as RI  # noqa: E402

pkg = importlib.import_module("one-2-3-45_amd")


@torch.no_grad()
def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
    out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
    suffix = sys.argv[4] if len(sys.argv) > 4 else ""
    torch.set_num_threads(min(32, os.cpu_count() or 1))            # the same cap as bench.py's oracle leg: more threads only add fork / join overhead on these op sizes
    V, HW, D = 8, 256, 128
    sc = pkg.synth.make_scene(V, image_seed=0)
    sdfnet, rnet, var, renderer = RI.build_networks(D, seed=0)
    T = torch.from_numpy
    rng = np.random.default_rng(0)
    # geometry of the real scene (which voxels are valid) from the visibility count; latents random -- the timing does not depend on them
    feats16 = T(rng.standard_normal((V, 1, HW, HW)).astype(np.float32))
    coords, _, _ = O.costvol(feats16, T(sc["affine_mats"]), [D, D, D], 2.0 / (D - 1), T(sc["partial_vol_origin"]))
    mask = torch.zeros(D, D, D)
    mask[coords[:, 0].long(), coords[:, 1].long(), coords[:, 2].long()] = 1
    dense = (torch.from_numpy(rng.standard_normal((16, D, D, D)).astype(np.float32)) * 0.1 * mask[None])[None]
    mask = mask[None, None]
    fmaps = T(rng.standard_normal((V, 56, HW, HW)).astype(np.float32))
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    near, far = T(sc["query_near_far"][:1]), T(sc["query_near_far"][1:])
    sel = np.linspace(0, len(ro) - 1, 1 << 14).astype(np.int64)
    done, t0 = 0, time.time()
    while time.time() - t0 < budget and done + 512 <= len(sel):
        s = sel[done:done + 512]
        renderer.render(T(ro[s]), T(rd[s]), near, far, sdfnet, rnet, perturb_overwrite=0, background_rgb=1.0, alpha_inter_ratio=1.0, lod=0,
                        conditional_volume=dense, conditional_valid_mask_volume=mask, feature_maps=fmaps, color_maps=T(sc["images"]),
                        w2cs=T(sc["w2cs"]), intrinsics=T(sc["intrinsics"]), img_wh=[HW, HW], query_c2w=T(sc["query_c2w"])[None],
                        if_render_with_grad=False)
        done += 512
    dt = time.time() - t0
    R = 64
    t1 = time.time()
    renderer.extract_fields(torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), R, lambda p, **kw: sdfnet.sdf(p, **kw), "cpu",
                            conditional_volume=dense, lod=0)
    de = time.time() - t1
    # ---- the volume build of the same configuration, per stage (FeatureNet, pyramid, get_conditional_volume), on seeded images
    stages = {}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden_scale as MS
        cfg = MS.CONFIGS["c2"]
        fnet, sdf2, rnet2, var2, ren2 = MS.build_reference_networks(cfg)
        imgs = T(sc["images"])
        tick = time.time()
        fm2 = MS.fused_pyramid(fnet, imgs)
        stages["featurenet_fused_pyramid_s"] = time.time() - tick
        tick = time.time()
        f16 = sdf2.compress_layer(fm2)
        stages["compress_layer_s"] = time.time() - tick
        tick = time.time()
        cv2 = sdf2.get_conditional_volume(feature_maps=fm2[None], partial_vol_origin=T(sc["partial_vol_origin"])[None], proj_mats=T(sc["affine_mats"])[None],
                                          sizeH=HW, sizeW=HW, lod=0)
        stages["get_conditional_volume_s"] = time.time() - tick
        stages["kept_voxels"] = int(cv2["valid_mask_volume_scale0"].sum())
        stages["note"] = ("get_conditional_volume = compress layer + both back_project_sparse_type passes + aggregation + SparseCostRegNet (torchsparse stand-in: "
                          "oracle/recon.py's gather-GEMM-scatter restatement) + dense scatter")
        del cv2, fm2, f16
    except Exception as e:                               # noqa: BLE001  (the render timing above is the number bench.py needs)
        stages["error"] = f"{type(e).__name__}: {e}"
    out = {"what": "the reference's own SparseNeuSRenderer.render / extract_fields (models/sparse_neus_renderer.py:457-635, 881-905) on CPU, "
                   "BASELINE config-2 inputs (8 views 256^2, 128^3 volume); where it was measured: _meta.host / _meta.cpu / cores",
           "value": done / dt, "unit": "rays/s", "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "kind": "reference",
           "sample": f"{done} rays in 512-ray chunks (the runner's batch size), {dt:.1f} s",
           "extract_fields_points_per_s": R ** 3 / de, "extract_fields_sample": f"{R}^3 grid, {de:.1f} s (a 256^3 grid is 64x that)",
           "volume_build_stages": stages}
    import datetime
    import platform
    import subprocess
    git = lambda *a: subprocess.run(["git", "-C", ROOT] + list(a), capture_output=True, text=True).stdout.strip()
    out["_meta"] = {"round": tag, "commit": git("rev-parse", "--short", "HEAD") or os.environ.get("O2345_COMMIT", ""), "dirty": bool(git("status", "--porcelain", "--", "oracle", "one-2-3-45_amd")),
                    "date_utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"), "host": platform.node(), "nproc": os.cpu_count(),
                    "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"), "torch": torch.__version__,
                    "script": "tools/time_reference_cpu.py"}
    out["_meta"]["reference_dir"] = RI.REF
    out["_meta"]["has_gpu"] = bool(torch.cuda.is_available())
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{tag}_cpu_reference{suffix}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
