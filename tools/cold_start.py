"""Cold-start cost of the reconstruction path in a fresh process on a fresh box: first scene vs second scene, and -- for comparison -- what the
first call of ATen's conv2d (MIOpen solver search) costs for the ten FeatureNet / compress-layer shapes the path no longer sends to a library."""
import json, os, sys, time
t0 = time.time()
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
res = {"import_and_context_s": time.time() - t0}
wt = pipeline.SceneWeights(dev, seed=0)
for name, seed in (("first_scene", 0), ("second_scene", 1)):
    inp = bench.make_inputs(dev, 8, seed, 2)
    torch.cuda.synchronize(); t = time.time()
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], 128, 2.0 / 127)
    torch.cuda.synchronize(); tv = time.time() - t
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    torch.cuda.synchronize(); tr = time.time() - t - tv
    res[name] = {"volume_build_ms": tv * 1e3, "render_ms": tr * 1e3}
import torch.nn.functional as F
shapes = [(3, 8, 3, 1, 256), (8, 8, 3, 1, 256), (8, 16, 5, 2, 256), (16, 16, 3, 1, 128), (16, 32, 5, 2, 128), (32, 32, 3, 1, 64), (32, 32, 1, 1, 64), (32, 16, 3, 1, 128),
          (32, 8, 3, 1, 256), (56, 16, 3, 1, 256)]
tot = 0.0
for cin, cout, k, s, hw in shapes:
    x = torch.randn(8, cin, hw, hw, device=dev); w = torch.randn(cout, cin, k, k, device=dev)
    torch.cuda.synchronize(); t = time.time(); F.conv2d(x, w, None, s, k // 2); torch.cuda.synchronize(); tot += time.time() - t
res["aten_conv2d_first_calls_ten_shapes_s"] = tot
print(json.dumps(res))
