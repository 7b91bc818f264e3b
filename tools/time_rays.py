"""Per-kernel times of one config-2 render through torch's profiler-free route: HIP events around ops.render_rays stages are not exposed,
so this simply times whole renders (diagnostic for the ray kernels: everything else is unchanged between variants)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], 128, 2.0 / 127)
f = lambda: pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
o = f(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); o = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(json.dumps({"render_ms_min": min(ts), "render_ms_mean": float(np.mean(ts)), "color_checksum": float(o["color"].double().sum()), "depth_checksum": float(o["depth"].double().sum())}))
