"""Minimal reproducer: tools/ubench/pk_victim.hip (a chain of v_pk_*_f32 + dependent v_floor_f32) next to a pure-MFMA kernel of another stream.
No part of the product library is involved.  Prints, per build of the victim (with / without packed-FP32 instructions), how many launches differ from the
idle result and which lanes of a wave the differing threads are."""
import ctypes, json, os, sys, threading
import torch
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench")
P = ctypes.CDLL(os.path.join(HERE, "libpoison.so"))
P.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
fsink = torch.zeros(16, device=dev)
stop = False
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def co(kind):
    def run():
        torch.cuda.set_device(dev)
        s = torch.cuda.Stream(device=dev)
        while not stop:
            P.aggr_launch(kind, 40000 if kind == 11 else 100000, 0, 256, ctypes.c_void_p(fsink.data_ptr()), ctypes.c_void_p(s.cuda_stream))
            s.synchronize()
    return run


res = {}
for lib in ("libpk_victim.so", "libpk_victim_nopk.so"):
    Vl = ctypes.CDLL(os.path.join(HERE, lib))
    Vl.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    Vl.pk_load_victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    blocks = 8192
    table = torch.rand(1 << 20, 4, device=dev)                       # 16 MB of float4: L2 / MALL resident like the feature maps
    for victim, kind, kname in (("alu", 11, "mfma"), ("load", 11, "mfma"), ("load", 12, "alu_spin")):
        if victim == "alu":
            launch = lambda o, st: Vl.pk_victim_launch(500, blocks, ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(st.cuda_stream))
            nout = blocks * 256
        else:
            launch = lambda o, st: Vl.pk_load_victim_launch(ctypes.c_void_p(table.data_ptr()), (1 << 20) - 1, 8, blocks, ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(st.cuda_stream))
            nout = blocks * 256 * 8
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            ref = torch.zeros(nout, device=dev)
            launch(ref, s)
            s.synchronize()
            stop = False
            th = threading.Thread(target=co(kind))
            th.start()
            bad, lanes, nthreads = 0, set(), []
            for _ in range(ITERS):
                out = torch.zeros(nout, device=dev)
                launch(out, s)
                ne = (out != ref).view(blocks * 256, -1).any(1)
                n = int(ne.sum())
                if n:
                    bad += 1
                    idx = torch.nonzero(ne).flatten()
                    lanes |= set((idx % 64).tolist())
                    nthreads.append(n)
            s.synchronize()
            stop = True
            th.join()
        res[f"{lib}|{victim}|co_{kname}"] = {"bad_launches": bad, "of": ITERS, "lanes": sorted(lanes), "threads_wrong": nthreads[:8]}
print(json.dumps(res))
