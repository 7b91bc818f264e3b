"""Every phase of the scene pass as the victim of a co-running matrix kernel of another stream (the pure-MFMA loop of tools/ubench/poison.hip, and the library's own
brick sparse convolution): launches whose outputs differ bitwise from the phase's idle result.  Companion of tools/stress_gather.py; expected: all zeros."""
import ctypes, json, os, subprocess, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
so = os.path.join(ROOT, "tools", "ubench", "libpoison.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "ubench", "poison.hip"), "-o", so],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
P = ctypes.CDLL(so)
P.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wt, wt2 = pipeline.SceneWeights(dev, seed=0), pipeline.SceneWeights(dev, seed=0)
wt.grid_tables(256)
inp = bench.make_inputs(dev, 8, 0, 2)
D, vs = 128, 2.0 / 127
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, vs)
torch.cuda.synchronize()
fsink = torch.zeros(16, device=dev)
stop = False


def flat(x):
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, dict):
        return [t for k in sorted(x) for t in flat(x[k])]
    if isinstance(x, (list, tuple)):
        return [t for y in x for t in flat(y)]
    return []


K0 = wt2.costreg.p["conv0"][0]
AGGR = {
    "mfma_loop": lambda s: P.aggr_launch(11, 40000, 0, 256, ctypes.c_void_p(fsink.data_ptr()), ctypes.c_void_p(s.cuda_stream)),
    "brick_conv": lambda s: ops.sparse_conv3d_x3(0, vol["rows"], vol["row_of_voxel"], (D, D, D), vol["coords"], 1, wt2.costreg.xblob["conv0"], K0.shape[2]),
}


def co(fn):
    def run():
        torch.cuda.set_device(dev)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            while not stop:
                fn(s)
                s.synchronize()
    return run


VICT = {
    "featurenet+compress": (lambda: wt.compress.forward_nhwc(pipeline.fused_pyramid(wt.featurenet, inp["imgs"], want_cmaps=True, want_nchw=False)[1], nhwc_offset=3), 40),
    "costvol_index": (lambda: ops.costvol_index(inp["aff"], 8, 256, 256, (D, D, D), vs, inp["origin"])[:3], 40),
    "costvol_gather": (lambda: ops.costvol_gather(vol["feats_nhwc"], inp["aff"], (D, D, D), vs, inp["origin"], vol["cnt"], vol["coords"]), 100),
    "costreg": (lambda: wt.costreg.forward(vol["rows"], vol["coords"], vol["row_of_voxel"], (D, D, D)), 40),
    "scatter_dense": (lambda: ops.scatter_dense(vol["rows16"], vol["row_of_voxel"], (D, D, D), want_cf=False), 40),
    "render": (lambda: pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"]), 8),
    "mesh": (lambda: pipeline.extract_mesh(wt, vol, inp["proj"], inp["cam_pos"], 256), 12),
}
res = {}
for aname, afn in AGGR.items():
    for name, (fn, iters) in VICT.items():
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            ref = [t.clone() for t in flat(fn())]
            s.synchronize()
            stop = False
            th = threading.Thread(target=co(afn))
            th.start()
            bad = 0
            for _ in range(iters):
                got = flat(fn())
                if any(a.shape != b.shape or bool((a != b).any()) for a, b in zip(ref, got)):
                    bad += 1
            s.synchronize()
            stop = True
            th.join()
        torch.cuda.synchronize()
        res[f"{name}|next_to_{aname}"] = f"{bad}/{iters}"
print(json.dumps(res))
