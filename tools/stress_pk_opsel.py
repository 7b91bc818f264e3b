"""Standalone reproducer (no product code): tools/ubench/pk_opsel.hip -- packed-FP32 instructions with and without cross-half source selection, each checked
in-kernel against two plain instructions -- alone and next to the pure-MFMA loop of tools/ubench/poison.hip on another stream.  Prints, per instruction form,
the number of wrong results and the lanes (of a wave) they came from."""
import ctypes, json, os, subprocess, sys, threading
import torch
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench")


def build(name):
    so = os.path.join(HERE, "lib" + name + ".so")
    src = os.path.join(HERE, name + ".hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ctypes.CDLL(so)


P, V = build("poison"), build("pk_opsel")
P.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
V.pk_opsel_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
fsink = torch.zeros(16, device=dev)
stop = False
FORMS = {0: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]  (D.lo=A.lo-B.hi, D.hi=A.hi-B.lo; the gather's form)",
         1: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]                            (swizzle only)",
         2: "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]                               (negation only)",
         3: "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]                            (multiply, swizzle)",
         4: "v_pk_add_f32 op_sel_hi:[1,0]                                         (B.lo broadcast)",
         5: "v_pk_add_f32                                                          (plain)",
         6: "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]                        (fma, src1 swizzled)",
         7: "v_pk_mov_b32 op_sel:[1,0]                                            (D.lo=A.hi, D.hi=B.lo)",
         8: "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]                            (swizzle on src0)"}
LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def co(kind):
    def run():
        torch.cuda.set_device(dev)
        s = torch.cuda.Stream(device=dev)
        while not stop:
            P.aggr_launch(kind, 40000 if kind == 11 else 100000, 0, 256, ctypes.c_void_p(fsink.data_ptr()), ctypes.c_void_p(s.cuda_stream))
            s.synchronize()
    return run


res = {}
for corun, kind in (("alone", None), ("next_to_mfma_loop", 11), ("next_to_alu_spin", 12)):
    for form, desc in FORMS.items():
        errs = torch.zeros(65, dtype=torch.int32, device=dev)
        sample = torch.zeros(8, device=dev)
        s = torch.cuda.Stream(device=dev)
        stop = False
        th = None
        if kind is not None:
            th = threading.Thread(target=co(kind))
            th.start()
        with torch.cuda.stream(s):
            for _ in range(LAUNCHES):
                assert V.pk_opsel_launch(form, 2000, 4096, ctypes.c_void_p(errs.data_ptr()), ctypes.c_void_p(sample.data_ptr()), ctypes.c_void_p(s.cuda_stream)) == 0
            s.synchronize()
        stop = True
        if th:
            th.join()
        e = errs.cpu()
        lanes = [i for i in range(64) if int(e[i])]
        rec = {"wrong_results": int(e[:64].sum()), "of": LAUNCHES * 4096 * 256 * 2000, "lanes": (f"{lanes[0]}..{lanes[-1]} ({len(lanes)} lanes)" if lanes else "-")}
        if lanes:
            sm = [float(x) for x in sample.cpu()]
            rec["first_failure"] = {"A": sm[0:2], "B": sm[2:4], "packed_result": sm[4:6], "expected": sm[6:8]}
        res[f"{corun} | form {form}: {desc.split('(')[0].strip()}"] = rec
for k, v in res.items():
    print(k, "->", json.dumps(v))
