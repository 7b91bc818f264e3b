"""Times the finest sparse convolution (conv0: 32 -> 16 channels, same resolution) in its two forms (O2345_SPARSE_BRICK=0: gather form) at 128^3 and 256^3."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
res = {}
for D in (128, 256):
    wt = pipeline.SceneWeights(dev, seed=0)
    inp = bench.make_inputs(dev, 8, 0, 1)
    vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1))
    cr = wt.costreg
    blob = None
    for name in dir(cr):
        pass
    rows, coords, grid = vol["rows"], vol["coords"], vol["row_of_voxel"]
    K = torch.as_tensor(np.asarray(wt.costreg_sd["conv0.net.0.kernel"]), dtype=torch.float32, device=dev)
    import importlib
    W = importlib.import_module("one-2-3-45_amd.weights")
    blob = torch.from_numpy(W.pack_sparse_conv_x3(K.cpu().numpy())).to(dev)
    outs = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["O2345_SPARSE_BRICK"] = mode
        f = lambda: ops.sparse_conv3d_x3(0, rows, grid, (D, D, D), coords, 1, blob, 16)
        y = f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        res.setdefault(f"D{D}_{'brick' if mode == '1' else 'gather'}_ms", []).append(round(min(ts), 4))
        outs[mode] = y
    res[f"D{D}_maxdiff_rel"] = float((outs["1"] - outs["0"]).abs().max() / outs["0"].abs().max())
    res[f"D{D}_rows"] = int(rows.shape[0])
    os.environ.pop("O2345_SPARSE_BRICK")
    tv = []
    for _ in range(5):
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
        vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, 2.0 / (D - 1)); b.record(); torch.cuda.synchronize(); tv.append(a.elapsed_time(b))
    res[f"D{D}_volume_build_ms"] = round(min(tv), 3)
    del vol
print(json.dumps(res))
