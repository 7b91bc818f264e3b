"""Times the cost-volume gather of the library selected by O2345_LIB at 128^3 / 8 views (BASELINE config 2) and 256^3 (config 5)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
res = {"lib": os.path.basename(ops._lib.lib()._name)}
for D in (128, 256):
    wt = pipeline.SceneWeights(dev, seed=0)
    inp = bench.make_inputs(dev, 8, 0, 1)
    vs = 2.0 / (D - 1)
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, vs)
    f = lambda: ops.costvol_gather(vol["feats_nhwc"], inp["aff"], (D, D, D), vs, inp["origin"], vol["cnt"], vol["coords"])
    y = f(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    res[f"D{D}"] = {"rows": int(vol["coords"].shape[0]), "ms_min": round(min(ts), 4), "ms_median": round(float(np.median(ts)), 4),
                    "equals_product_rows": bool((y == vol["rows"]).all()), "checksum": float(y.double().sum())}
print(json.dumps(res))
