"""ctypes loader for libo2345_hip.so.  Prototypes are generated from include/o2345.h so that the binding can never
drift from the declared C ABI.  There is NO fallback: if the library is missing or a call fails, this raises."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "o2345.h")
LIB_PATH = os.environ.get("O2345_LIB") or os.path.join(HERE, "libo2345_hip.so")       # O2345_LIB: an A/B build variant (build.build_variant)

ABI_VERSION = 210          # include/o2345.h: o2345_version()

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
        rest = ctypes.c_char_p if (ret.startswith("const char") or name in ("o2345_last_error", "o2345_knobs")) else (_CT["size_t"] if ret == "size_t" else ctypes.c_int)
        protos[name] = (rest, argt)
    return protos


def parse_struct(name, path=HEADER):
    """-> [(field, ctypes type)] of ``typedef struct <name> {...} <name>;`` in the header, in declaration order: the binding's struct is GENERATED from
    the one declaration the C side compiles (csrc/ includes the same header), never kept by hand."""
    src = open(path).read()
    m = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), src, flags=re.S)
    if not m:
        raise ValueError(f"{name} is not declared in {path}")
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        if "*" in decl:
            base, names = decl.rsplit("*", 1)
            if "," in names:
                raise ValueError(f"{name}: one pointer per declaration, got {decl!r}")
            fields.append((names.strip(), ctypes.c_void_p))
            continue
        typ, names = decl.split(" ", 1)
        ct = {"int": ctypes.c_int, "float": ctypes.c_float}.get(typ)
        if ct is None:
            raise ValueError(f"{name}: cannot map field declaration {decl!r}")
        fields += [(n.strip(), ct) for n in names.split(",")]
    return fields


class RenderIO(ctypes.Structure):
    _fields_ = parse_struct("O2345RenderIO")


def check_render_io_layout(L, path=LIB_PATH):
    """The loaded library's own sizeof / offsetof table against the generated Structure: a field added to, removed from or reordered in only one of the
    header the library was compiled with and the header this binding parsed fails HERE, at load time, instead of corrupting render calls."""
    want = (ctypes.c_size_t * 256)()
    n = L.o2345_render_io_layout(want, 256)
    mine = [(f, getattr(RenderIO, f).offset) for f, _ in RenderIO._fields_]
    if n != len(mine) or L.o2345_render_io_size() != ctypes.sizeof(RenderIO) or any(int(want[i]) != off for i, (_, off) in enumerate(mine)):
        theirs = [int(want[i]) for i in range(min(n, 256))]
        raise RuntimeError(f"O2345RenderIO layout mismatch between {path} ({n} fields, {L.o2345_render_io_size()} bytes, offsets {theirs}) and "
                           f"{HEADER} ({len(mine)} fields, {ctypes.sizeof(RenderIO)} bytes, offsets {[o for _, o in mine]}): rebuild the library")


_LIB = None


def load_library(path):
    """dlopen ``path``, attach the header's prototypes, check ABI version and struct layout.  ``lib()`` does this once for the product library; tests load
    build variants (libo2345_hip_<tag>.so) next to it with this."""
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                           "There is no CPU fallback for the reconstruction path.")
    L = ctypes.CDLL(path)
    for name, (rest, argt) in parse_header().items():
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = rest, argt
    if L.o2345_version() != ABI_VERSION:
        raise RuntimeError(f"{path} implements ABI {L.o2345_version()}, this binding expects {ABI_VERSION}: rebuild the library")
    check_render_io_layout(L, path)
    L.o2345_knobs()                    # the environment knobs are read now, once (csrc/common.h)
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load_library(LIB_PATH)
    return _LIB


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"o2345 {what} failed ({rc}): {lib().o2345_last_error().decode()}")
