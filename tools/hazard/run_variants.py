"""Runs every code object of tools/hazard/out/ (make_variants.py) as the cost-volume gather next to the pure-MFMA loop of tools/ubench/poison.hip on another
stream: launches whose rows differ from the variant's own idle result.  usage: run_variants.py [iters] [name ...]"""
import ctypes, json, os, struct, subprocess, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
ITERS = int(args[0]) if args else 200
NAMES = args[1:]
sys.argv = [sys.argv[0]]
import bench
pipeline, ops = bench.pipeline, bench.ops
OUT = os.path.join(ROOT, "tools", "hazard", "out")
KERNEL = b"_ZN5o234516k_costvol_gatherILi16EEEvPKfS2_iiiNS_7VolGeomEPKhPKiiPf"
so = os.path.join(ROOT, "tools", "ubench", "libpoison.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "ubench", "poison.hip"), "-o", so],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
P = ctypes.CDLL(so)
P.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wt = pipeline.SceneWeights(dev, seed=0)
inp = bench.make_inputs(dev, 8, 0, 2)
D, vs = 128, 2.0 / 127
vol = pipeline.build_volume(wt, inp["imgs"], inp["aff"], inp["origin"], D, vs)
torch.cuda.synchronize()
n = vol["coords"].shape[0]
org = [float(x) for x in inp["origin"][:3]] if not torch.is_tensor(inp["origin"]) else [float(x) for x in inp["origin"].cpu()[:3]]
fsink = torch.zeros(16, device=dev)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


def load(path):
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    check(hip.hipModuleLoad(ctypes.byref(mod), path.encode()), "hipModuleLoad " + path)
    check(hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL), "hipModuleGetFunction")
    return fn


hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 6 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def launch(fn, out, stream):
    karg = struct.pack("<QQiiiiiiffffQQiiQ", vol["feats_nhwc"].data_ptr(), inp["aff"].data_ptr(), 8, 256, 256, D, D, D, vs, org[0], org[1], org[2],
                       vol["cnt"].data_ptr(), vol["coords"].data_ptr(), n, 0, out.data_ptr())
    assert len(karg) == 88
    buf = ctypes.create_string_buffer(karg, 88)
    size = ctypes.c_size_t(88)
    extra = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p), 2, ctypes.cast(ctypes.pointer(size), ctypes.c_void_p), 3)
    check(hip.hipModuleLaunchKernel(fn, (n * 4 + 255) // 256, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(stream.cuda_stream), None, extra), "launch")


stop = False


def co():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(device=dev)
    while not stop:
        P.aggr_launch(11, 40000, 0, 256, ctypes.c_void_p(fsink.data_ptr()), ctypes.c_void_p(s.cuda_stream))
        s.synchronize()


names = NAMES or sorted(f[:-6] for f in os.listdir(OUT) if f.endswith(".hsaco"))
fns = {nm: load(os.path.join(OUT, nm + ".hsaco")) for nm in names}
res = {}
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    refs = {}
    for nm in names:
        o = torch.zeros(n, 32, device=dev)
        launch(fns[nm], o, s)
        s.synchronize()
        refs[nm] = o
        res[nm] = {"idle_equals_product": bool((o == vol["rows"]).all()), "bad": 0, "of": 0, "rows": []}
    th = threading.Thread(target=co)
    th.start()
    for rep in range(4):                                         # variants interleaved: box / clock drift hits all of them alike
        for nm in names:
            for _ in range(ITERS // 4):
                o = torch.empty(n, 32, device=dev)
                launch(fns[nm], o, s)
                ne = (o != refs[nm]).any(1)
                k = int(ne.sum())
                res[nm]["of"] += 1
                if k:
                    res[nm]["bad"] += 1
                    if len(res[nm]["rows"]) < 6:
                        res[nm]["rows"].append(k)
    s.synchronize()
    stop = True
    th.join()
print(json.dumps(res))
