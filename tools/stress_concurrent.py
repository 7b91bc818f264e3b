"""Which co-running kernel corrupts the cost-volume gather?  Thread 0 loops the gather on fixed inputs and counts rows that differ from a reference
taken with the device otherwise idle; the other threads (own streams) loop one candidate workload."""
import json, os, sys, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wts = [pipeline.SceneWeights(dev, seed=0) for _ in range(3)]
for w_ in wts:
    w_.grid_tables(256)
inp = bench.make_inputs(dev, 8, 0, 2)
D, vs = 128, 2.0 / 127
vol = pipeline.build_volume(wts[0], inp["imgs"], inp["aff"], inp["origin"], D, vs)
out0 = pipeline.render(wts[0], vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
torch.cuda.synchronize()
gather = lambda: ops.costvol_gather(vol["feats_nhwc"], inp["aff"], (D, D, D), vs, inp["origin"], vol["cnt"], vol["coords"])
ref = gather()
torch.cuda.synchronize()
assert bool((ref == vol["rows"]).all())
stop = False


def victim(res, iters, fn, refv):
    s = torch.cuda.Stream(device=dev)
    bad = []
    with torch.cuda.stream(s):
        for _ in range(iters):
            r = fn()
            r = r if torch.is_tensor(r) else r[0]
            ne = (r != refv).reshape(r.shape[0], -1).any(1)
            n = int(ne.sum())
            if n:
                idx = torch.nonzero(ne).flatten()
                bad.append((n, sorted(set((idx % 16).tolist()))))
        s.synchronize()
    res.append(bad)


def co(fn):
    def run():
        torch.cuda.set_device(dev)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            while not stop:
                fn()
                s.synchronize()
    return run


WORK = {
    "idle": None,
    "gather": lambda w: (lambda: gather()),
    "featurenet": lambda w: (lambda: pipeline.fused_pyramid(w.featurenet, inp["imgs"], want_cmaps=True, want_nchw=False)),
    "costreg": lambda w: (lambda: w.costreg.forward(vol["rows"], vol["coords"], vol["row_of_voxel"], (D, D, D))),
    "render": lambda w: (lambda: pipeline.render(w, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])),
    "mesh": lambda w: (lambda: pipeline.extract_mesh(w, vol, inp["proj"], inp["cam_pos"], 256)),
    "torch_matmul": lambda w: (lambda a=torch.randn(4096, 4096, device=dev): a @ a),
    "torch_copy": lambda w: (lambda a=torch.randn(1 << 26, device=dev): a.clone()),
}
res = {}
for name, mk in WORK.items():
    for nco in ((0,) if mk is None else (1, 2)):
        stop = False
        ths = [threading.Thread(target=co(mk(wts[1 + i]))) for i in range(nco)]
        [t.start() for t in ths]
        r = []
        victim(r, 40, gather, ref)
        stop = True
        [t.join() for t in ths]
        torch.cuda.synchronize()
        res[f"gather_vs_{name}_x{nco}"] = {"bad_iters": len(r[0]), "detail": r[0][:4]}
# the other direction: is the gather the only victim?  costreg forward as the victim of a co-running render / gather
ref16 = wts[0].costreg.forward(vol["rows"], vol["coords"], vol["row_of_voxel"], (D, D, D)).clone()
for name in ("render", "gather", "featurenet"):
    stop = False
    ths = [threading.Thread(target=co(WORK[name](wts[1 + i]))) for i in range(2)]
    [t.start() for t in ths]
    r = []
    victim(r, 15, lambda: wts[0].costreg.forward(vol["rows"], vol["coords"], vol["row_of_voxel"], (D, D, D)), ref16)
    stop = True
    [t.join() for t in ths]
    res[f"costreg_vs_{name}_x2"] = {"bad_iters": len(r[0]), "detail": r[0][:3]}
print(json.dumps(res))
