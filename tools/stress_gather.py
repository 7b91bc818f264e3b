"""The cost-volume gather (build selected by O2345_LIB / O2345_GATHER_DEBUG) next to the pure-MFMA co-runner of tools/ubench/poison.hip:
iterations whose rows differ from the idle result.  usage: stress_gather.py [iters]"""
import ctypes, json, os, sys, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
P = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libpoison.so"))
P.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
D, vs = 128, 2.0 / 127
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, vs)
torch.cuda.synchronize()
fsink = torch.zeros(16, device=dev)
stop = False
gather = lambda: ops.costvol_gather(vol["feats_nhwc"], inp["aff"], (D, D, D), vs, inp["origin"], vol["cnt"], vol["coords"])


def co():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(device=dev)
    while not stop:
        P.aggr_launch(11, 40000, 0, 256, ctypes.c_void_p(fsink.data_ptr()), ctypes.c_void_p(s.cuda_stream))
        s.synchronize()


s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    ref = gather().clone()
    s.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(s)
    for _ in range(20):
        gather()
    ev1.record(s)
    s.synchronize()
    idle_ms = ev0.elapsed_time(ev1) / 20
    th = threading.Thread(target=co)
    th.start()
    bad, rows = 0, []
    for _ in range(ITERS):
        ne = (gather() != ref).any(1)
        n = int(ne.sum())
        if n:
            bad += 1
            rows.append(n)
    s.synchronize()
    stop = True
    th.join()
print(json.dumps({"lib": os.path.basename(ops._lib.lib()._name), "mode": os.environ.get("O2345_GATHER_DEBUG", "0"), "idle_ms": round(idle_ms, 4),
                  "bad_iters": bad, "of": ITERS, "rows": rows[:8], "idle_equals_product_rows": bool((ref == vol["rows"]).all())}))
