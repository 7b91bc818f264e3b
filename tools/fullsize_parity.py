"""Oracle-vs-HIP parity AT THE BENCHMARKED CONFIGURATION (BASELINE config 2: 8 views 256^2, 128^3 volume, 512^2 rays): renders a
sample of the config-2 rays on the GPU (both numerical forms) and with the CPU oracle, and prints / writes per-ray error statistics
together with the per-ray classification used by tests/test_gpu_edges_and_fullsize.py::test_fullsize_oracle_parity.

    gpurun -- python tools/fullsize_parity.py [n_rays] > gpurun_out/fullsize_parity.json"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("one-2-3-45_amd")
pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
from fullsize_util import (_oracle_args, build_full_scene, classify, gpu_render_sample, oracle_core_on, oracle_render_sample,  # noqa: E402
                           oracle_self_sensitivity, sampler_stage_check)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    full = build_full_scene(dev)
    ref, sel, dt = oracle_render_sample(full, n)
    res = {"n_rays": int(len(sel)), "oracle_s": dt}
    for prec in ("f16x3", "fp32"):
        out = gpu_render_sample(full, sel, prec)
        res[prec] = classify(out, ref, core=oracle_core_on(full, sel, out["z_vals"]), verbose=True)
        if os.environ.get("O2345_DUMP"):
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fullsize_parity_{prec}.npz"), sel=sel, z_gpu=out["z_vals"].numpy(),
                                z_ref=ref["z_vals"].numpy(), min_pdf=ref["min_pdf"].numpy(), color_gpu=out["color"].numpy(),
                                color_ref=ref["color_fine"].numpy(), w_gpu=out["weights"].numpy(), w_ref=ref["weights"].numpy())
    ce, de = oracle_self_sensitivity(full, sel, ref)
    q = lambda t: [float(torch.quantile(t.flatten(), x)) for x in (0.5, 0.9, 0.99, 1.0)]
    res["oracle_self_sensitivity_color_q50_90_99_max"] = q(ce)
    for prec in ("f16x3", "fp32"):
        out = gpu_render_sample(full, sel, prec)
        res[prec]["color_err_q50_90_99_max"] = q((out["color"] - ref["color_fine"]).abs().max(1).values)
    print({k: v for k, v in res.items() if "sens" in k}, res["f16x3"]["color_err_q50_90_99_max"], res["fp32"]["color_err_q50_90_99_max"], file=sys.stderr)
    ops = importlib.import_module("one-2-3-45_amd.ops")
    sc = full["sc"]
    dz, pdf, width = sampler_stage_check(ops, dev, torch.from_numpy(full["ro"][sel]), torch.from_numpy(full["rd"][sel]), float(sc["query_near_far"][0]),
                                         float(sc["query_near_far"][1]), _oracle_args(full), full["vol"]["maskvol"], full["D"])
    st = {}
    for lo, hi in ((1e-2, 10), (1e-3, 1e-2), (1e-4, 1e-3), (0, 1e-4)):
        m = (pdf >= lo) & (pdf < hi)
        if m.any():
            st[f"pdf[{lo:g},{hi:g})"] = {"n": int(m.sum()), "dz_max": float(dz[m].max()), "dz_over_width_max": float((dz[m] / width[m].clamp(min=1e-9)).max()),
                                         "dz_times_pdf_over_width_max": float((dz[m] * pdf[m] / width[m].clamp(min=1e-9)).max())}
    res["sampler_stage"] = st
    print(st, file=sys.stderr)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
