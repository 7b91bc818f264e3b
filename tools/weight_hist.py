"""Distribution of the compositing weights over the occupied sample points of one BASELINE-config-2 render (diagnostic for colour-list culling)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline = bench.pipeline
dev = torch.device("cuda:0")
res = {}
for seed in (0, 1):
    wt = pipeline.SceneWeights(dev, seed=0)
    inp = bench.make_inputs(dev, 8, seed, 2)
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], 128, 2.0 / 127)
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    w, pm = out["weights"].reshape(-1), out["pm"].reshape(-1) > 0
    wo = w[pm]
    r = {"occupied": int(pm.sum()), "rays_weights_sum_mean": float(out["weights_sum"].mean())}
    for th in (0.0, 1e-20, 1e-12, 1e-10, 1e-8, 1e-6, 1e-5, 1e-4):
        r[f"frac_w_le_{th:g}"] = float((wo <= th).float().mean())
        r[f"lost_weight_per_ray_max_{th:g}"] = float((torch.where((w <= th) & pm, w, torch.zeros_like(w)).reshape(out["weights"].shape).sum(0)).max())
    res[f"scene{seed}"] = r
print(json.dumps(res, indent=1))
