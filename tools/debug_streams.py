"""Locate the first tensor that differs between a sequential run and an N-stream run of the same scenes (see profiles/NOTES.md, multi-stream)."""
import json, os, sys, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.argv = [sys.argv[0]]
import bench
pipeline = bench.pipeline
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wts = [pipeline.SceneWeights(dev, seed=0) for _ in range(4)]
inp = bench.make_inputs(dev, 8, 0, 2)
K = 8
imgs = [torch.from_numpy(bench.scene_images(8, 100 + k)).to(dev) for k in range(K)]
KEYS = ("feats_nhwc", "cnt", "coords", "row_of_voxel", "rows", "rows16", "vol_cl", "maskvol")


def run(n_threads, volume_only):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_threads)]
    keep = [None] * K

    def worker(i):
        torch.cuda.set_device(dev)
        tm = bench.Timer()
        with torch.cuda.stream(streams[i]):
            for k in range(i, K, n_threads):
                if volume_only:
                    v_ = pipeline.build_volume(wts[i], imgs[k], inp["aff"], inp["origin"], 128, 2.0 / 127)
                else:
                    v_, o_, m_ = bench.step(wts[i], inp, 128, 256, tm, 1 << 18, imgs=imgs[k])
                keep[k] = {key: v_[key].clone() for key in KEYS}
            streams[i].synchronize()
    torch.cuda.synchronize()
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize()
    return keep


res = {}
ref = run(1, False)
for vo in (False, True):
    for rep in range(2):
        got = run(N, vo)
        for k in range(K):
            for key in KEYS:
                a, b = ref[k][key], got[k][key]
                if a.shape != b.shape:
                    res.setdefault(f"vo{int(vo)}_rep{rep}", []).append((k, key, "shape", list(a.shape), list(b.shape)))
                    continue
                ne = (a != b)
                if bool(ne.any()):
                    nz = torch.nonzero(ne.reshape(a.shape[0], -1).any(1)).flatten()
                    d = (a.double() - b.double()).abs().max().item()
                    res.setdefault(f"vo{int(vo)}_rep{rep}", []).append((k, key, int(ne.sum()), int(nz.numel()), int(nz[0]), int(nz[-1]), d))
                    if key == "rows":
                        bad = nz
                        info = {"scene": k, "bad_rows_head": bad[:24].tolist(), "runs": int((bad[1:] != bad[:-1] + 1).sum()) + 1}
                        info["matches_other_scene"] = {j: int((ref[j]["rows"][bad] == b[bad]).all(1).sum()) for j in range(K)}
                        info["cols_bad_per_row_head"] = ne[bad[:8]].sum(1).tolist()
                        r0 = int(bad[0])
                        info["row0_ref"] = [round(float(x), 4) for x in a[r0, :6]] + [round(float(x), 4) for x in a[r0, 16:20]]
                        info["row0_got"] = [round(float(x), 4) for x in b[r0, :6]] + [round(float(x), 4) for x in b[r0, 16:20]]
                        again = bench.ops.costvol_gather(got[k]["feats_nhwc"], inp["aff"], (128,) * 3, 2.0 / 127, inp["origin"], got[k]["cnt"], got[k]["coords"])
                        info["regather_equals_ref"] = bool((again == a).all())
                        res.setdefault("detail", []).append(info)
        res.setdefault(f"vo{int(vo)}_rep{rep}", [])
print(json.dumps(res))
