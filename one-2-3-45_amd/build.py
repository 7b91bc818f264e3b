"""Builds libo2345_hip.so (the C-ABI HIP library, gfx950 only) in-tree with hipcc.  No torch involved."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libo2345_hip.so")
SOURCES = ["api.cpp", "costvol.hip", "sparse.hip", "sparse_mfma.hip", "sdf_mlp.hip", "sdf_mlp_x3.hip", "render.hip", "list_sort.hip", "color_maps.hip", "color_mfma.hip", "color_pts.hip", "mcubes.hip", "mesh_pack.hip", "featmaps.hip", "convnet.hip"]
# No packed-FP32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  Measured on MI355X (profiles/NOTES.md, "co-resident MFMA"): the cost-volume
# gather built WITH them returns garbage in lanes 48..63 of some waves whenever a kernel of ANOTHER stream that issues MFMA shares its SIMDs (23 of 400 launches next
# to a pure-MFMA loop, 77 of 80 next to the brick sparse convolution); built without them: 0 of 400, same speed, and whole scenes on 2-4 streams become bit-identical
# to the sequential run.  A single stream never co-schedules two kernels, so the old build was only wrong under concurrency -- but a library must not depend on that.
# The whole-step time is unchanged (65.6 ms either way).  -Xclang reaches the host pass too, which prints "not a recognized feature" (filtered in _compile).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=default"] + NO_PACKED_FP32
# The network kernels are VALU-bound and take max(x, .) of raw MFMA accumulators hundreds of times per tile; under IEEE NaN
# rules every such max is preceded by a quieting v_max x,x,x.  Their inputs are finite by construction.
EXTRA_FLAGS = {name: ["-fno-honor-nans"] for name in ("sdf_mlp.hip", "sdf_mlp_x3.hip", "color_mfma.hip", "color_pts.hip")}


def sources_sha():
    """sha256 (first 16 hex digits) over the kernel sources and headers (csrc/*, sorted by name): what a profile / counter file was measured on.
    tools/summarize_rocprof.py stamps it into every profiles/*.json; bench.py prints counter-derived numbers only when it matches."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + [k + ":" + " ".join(v) for k, v in sorted(EXTRA_FLAGS.items())]).encode())      # a flag change is a different binary
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(HERE), "include", "o2345.h"), "rb").read())       # csrc/common.h includes the public header
    return h.hexdigest()[:16]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(cmd, verbose=False):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    out = "\n".join(l for l in r.stdout.splitlines() if "is not a recognized feature for this target" not in l)
    if out.strip():
        print(out, file=sys.stderr)
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "o2345.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        if force or _stale(obj, [path] + headers + [os.path.abspath(__file__)]):      # build.py holds the flags
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
            jobs.append(cmd)
    def run(cmd):
        _compile(cmd, verbose)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_variant(tag, defines, packed_fp32=False):
    """A second library for A/B runs on ONE box: every source recompiled with extra -D flags into build_<tag>/,
    linked as libo2345_hip_<tag>.so next to the product library (git-ignored; travels with the gpurun snapshot).  ``packed_fp32=True`` drops
    NO_PACKED_FP32 (the reproducer of the co-resident-MFMA corruption, tools/stress_gather.py)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build_" + tag)
    os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(HERE, f"libo2345_hip_{tag}.so")

    def run(src):
        flags = [f for f in FLAGS if not (packed_fp32 and f in NO_PACKED_FP32)]
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(src, []) + list(defines) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", os.path.join(objdir, src + ".o")]
        _compile(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, SOURCES))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [os.path.join(objdir, s + ".o") for s in SOURCES])
    return lib


def build_tiles_variant():
    """libo2345_hip_tiles.so: the product objects + csrc/color_mfma.hip recompiled with -DO2345_TILES_KERNEL (k_color_mfma, the (point, view)-column colour
    kernel that lost against k_color_pts: test-only since round 4).  tests/test_gpu_parity.py::test_color_points loads it next to the product library."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    build()
    objdir, vdir = os.path.join(HERE, "build"), os.path.join(HERE, "build_tiles")
    os.makedirs(vdir, exist_ok=True)
    src, obj = os.path.join(CSRC, "color_mfma.hip"), os.path.join(vdir, "color_mfma.hip.o")
    lib = os.path.join(HERE, "libo2345_hip_tiles.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "o2345.h")]
    if _stale(obj, [src] + headers + [os.path.abspath(__file__)]):
        _compile([hipcc] + FLAGS + EXTRA_FLAGS.get("color_mfma.hip", []) + ["-DO2345_TILES_KERNEL", "-c", src, "-o", obj])
    objs = [obj if s_ == "color_mfma.hip" else os.path.join(objdir, s_ + ".o") for s_ in SOURCES]
    if _stale(lib, objs):
        _compile([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
