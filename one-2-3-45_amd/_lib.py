"""ctypes loader for libo2345_hip.so.  Prototypes are generated from include/o2345.h so that the binding can never
drift from the declared C ABI.  There is NO fallback: if the library is missing or a call fails, this raises."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "o2345.h")
LIB_PATH = os.environ.get("O2345_LIB") or os.path.join(HERE, "libo2345_hip.so")       # O2345_LIB: an A/B build variant (build.build_variant)

_CT = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong, "size_t": ctypes.c_size_t,
       "void": None, "double": ctypes.c_double}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct.*?\}\s*\w+;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(o2345_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        def ctype(decl):
            decl = decl.strip()
            if "*" in decl:
                return ctypes.c_char_p if decl.startswith("const char") else ctypes.c_void_p
            base = re.sub(r"\b(const|unsigned)\b", "", decl).strip()
            base = " ".join(base.split()[:-1]) if len(base.split()) > 1 and base.split()[-1] not in ("long", "int") else base
            for k in ("long long", "size_t", "float", "double", "int"):
                if base.startswith(k):
                    return _CT[k]
            raise ValueError(f"cannot map C type {decl!r} in {name}")
        argt = [] if args in ("", "void") else [ctype(a) for a in args.split(",")]
        rest = ctypes.c_char_p if (ret.startswith("const char") or name == "o2345_last_error") else (_CT["size_t"] if ret == "size_t" else ctypes.c_int)
        protos[name] = (rest, argt)
    return protos


class RenderIO(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("sdf_blob", "color_blob", "vol_cl", "maskvol")] + [("D", ctypes.c_int)] +
                [(n, ctypes.c_void_p) for n in ("cmaps", "proj", "cam_pos")] +
                [("V", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int)] +
                [("rays_o", ctypes.c_void_p), ("rays_d", ctypes.c_void_p), ("R", ctypes.c_int), ("near", ctypes.c_float),
                 ("far", ctypes.c_float), ("n_samples", ctypes.c_int), ("n_importance", ctypes.c_int),
                 ("inv_s", ctypes.c_float), ("alpha_inter_ratio", ctypes.c_float), ("background", ctypes.c_float),
                 ("query_cam", ctypes.c_void_p)] +
                [(n, ctypes.c_void_p) for n in ("mid_z", "dists", "pm", "sdf", "grad", "rgb", "nviews", "color", "depth",
                                                "weights", "cdf", "weights_sum", "weights_max", "depth_var", "alpha_sum",
                                                "grad_err", "color_mask", "z_vals", "color_mfma_blob")] +
                [("sdf_bf16", ctypes.c_int), ("color_x3_blob", ctypes.c_void_p), ("t_rand", ctypes.c_void_p)])


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                               "There is no CPU fallback for the reconstruction path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (rest, argt) in parse_header().items():
            fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = rest, argt
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"o2345 {what} failed ({rc}): {lib().o2345_last_error().decode()}")
