"""Whole-scene steps of BASELINE config 2, K scenes processed (a) one after the other on one HIP stream (bench.py's timed loop) and (b) by N host threads,
each with its own stream, scenes dealt round-robin: does overlapping the latency-bound phases of one scene (volume build: ~45 small kernels + two size
read-backs; marching cubes) with the render kernels of another buy throughput?  Same K, same scenes, wall clock around all of them."""
import json, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline = bench.pipeline
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wts = [pipeline.SceneWeights(dev, seed=0) for _ in range(4)]        # one weights object per stream: they carry per-object caches / scratch
for w_ in wts:
    w_.grid_tables(256)
inp = bench.make_inputs(dev, 8, 0, 2)
K = 12
imgs = [torch.from_numpy(bench.scene_images(8, 100 + k)).to(dev) for k in range(K)]


def run(n_threads):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_threads)]
    sums = [None] * K

    def worker(i):
        torch.cuda.set_device(dev)
        tm = bench.Timer()
        with torch.cuda.stream(streams[i]):
            out = None
            for k in range(i, K, n_threads):
                out = None
                out = bench.step(wts[i], inp, 128, 256, tm, 1 << 18, imgs=imgs[k])
                v_, o_, m_ = out
                sums[k] = (o_[0]["color"].double().sum(), m_[0].shape[0], int(v_["n_voxels"]), v_["vol_cl"].double().sum(), o_[0]["depth"].double().sum(),
                           v_["rows"].double().sum(), v_["feats_nhwc"].double().sum(), o_[0]["weights_sum"].double().sum())
            streams[i].synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if n_threads == 1:
        worker(0)
    else:
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / K * 1e3, [tuple(float(x) for x in s) for s in sums]


res = {}
run(1); run(2)                                   # warm every stream's pools
for n in (1, 2, 3, 4, 1, 2, 3):
    ms, sums = run(n)
    res.setdefault(f"threads{n}_ms_per_scene", []).append(round(ms, 2))
    if n == 1:
        ref = sums
    else:
        res[f"threads{n}_same_results"] = bool(sums == ref)
        bad = [(k, [j for j in range(len(ref[k])) if sums[k][j] != ref[k][j]]) for k in range(K) if sums[k] != ref[k]]
        if bad:
            res.setdefault(f"threads{n}_mismatch(scene,[fields: color,verts,nvox,vol,depth,rows,feats,wsum])", []).append(bad[:6])
print(json.dumps(res))
