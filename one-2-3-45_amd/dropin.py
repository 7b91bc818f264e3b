"""Run an unmodified reference script on the HIP back end:

    cd /path/to/One-2-3-45/reconstruction
    python -m o2345_amd.dropin exp_runner_generic_blender_val.py --mode export_mesh --conf confs/one2345_lod0_val_demo.conf ...

An import hook serves the reference's module NAMES from this package: the third-party packages (torchsparse,
inplace_abn, mcubes, trimesh's PLY export) and the four L1 modules whose classes the runner imports (exp_runner_generic_blender_val.py:16-20).
Nothing in the reference tree is modified; everything else (trainer_generic, data, confs) is imported from the reference.

Without touching the command line (run.py:61-67 builds `python exp_runner_generic_blender_val.py ...` itself and runs it with os.system): put
``one-2-3-45_amd/autoload`` first on PYTHONPATH -- its sitecustomize.py calls activate() in every interpreter whose script is the reference's runner
(and in no other: run.py's own process keeps the real trimesh / mcubes)."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import runpy
import sys

PKG = "one-2-3-45_amd"
ALIASES = {
    "torchsparse": f"{PKG}.shims.torchsparse", "torchsparse.tensor": f"{PKG}.shims.torchsparse.tensor",
    "torchsparse.nn": f"{PKG}.shims.torchsparse.nn", "torchsparse.nn.functional": f"{PKG}.shims.torchsparse.nn.functional",
    "torchsparse.nn.utils": f"{PKG}.shims.torchsparse.nn.utils", "inplace_abn": f"{PKG}.shims.inplace_abn",
    "mcubes": f"{PKG}.shims.mcubes", "trimesh": f"{PKG}.shims.trimesh",
    "models.sparse_sdf_network": f"{PKG}.recon.sparse_sdf_network", "models.sparse_neus_renderer": f"{PKG}.recon.sparse_neus_renderer",
    "models.rendering_network": f"{PKG}.recon.rendering_network", "models.featurenet": f"{PKG}.featurenet",
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)

    def exec_module(self, module):
        pass


class _EmptyPackageLoader(importlib.abc.Loader):
    def create_module(self, spec):
        return None

    def exec_module(self, module):
        module.__path__ = []


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name in ALIASES:
            return importlib.util.spec_from_loader(name, _AliasLoader(ALIASES[name]))
        if name == "models":
            # normally the reference's own `models` package (cwd = reconstruction/); when it is not importable (tests),
            # provide an empty parent so that the aliased sub-modules can still be imported by their reference names
            if importlib.machinery.PathFinder.find_spec(name, sys.path) is None:
                return importlib.util.spec_from_loader(name, _EmptyPackageLoader(), is_package=True)
        return None


def cpu_threads():
    """Intra-op CPU threads for a drop-in run: $O2345_CPU_THREADS, else min(8, cores).  The reference's trainer does a handful of SMALL tensor ops on the
    host inside its timing brackets (torch.tensor(vertices).to(...), colours * 255, ...); with torch's default of one thread per core the OpenMP
    fork / join of each costs 50 - 70 ms on a 256-core MI355X host (measured: export_mesh_step 33 ms with 8 threads, 35 - 100 ms with 128)."""
    n = os.environ.get("O2345_CPU_THREADS")
    return max(1, int(n)) if n else max(1, min(8, os.cpu_count() or 1))


def tune_host_allocator():
    """glibc serves allocations above M_MMAP_THRESHOLD (128 KB at start) with mmap and returns them with munmap.  In a ROCm process every munmap runs the
    GPU driver's MMU-notifier callbacks for the unmapped range: measured 10 - 15 ms per 8 MB numpy temporary of the trainer's own frame transforms in the
    first "export mesh time" bracket of a fresh process (`vertices * scale + shift` on 332 k vertices: 24 - 30 ms instead of 2.5; later calls are fast only
    because glibc raises its threshold after the first such free).  Raise the threshold to glibc's maximum (32 MB) and the trim threshold to 256 MB up front,
    so that the trainer's medium-sized temporaries come from the heap.  $O2345_MALLOC_TUNE=0 leaves the allocator alone.  -> True if applied."""
    if os.environ.get("O2345_MALLOC_TUNE", "1") in ("", "0"):
        return False
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        ok = libc.mallopt(M_MMAP_THRESHOLD, 32 << 20)
        ok = libc.mallopt(M_TRIM_THRESHOLD, 256 << 20) and ok
        return bool(ok)
    except (OSError, AttributeError):       # not glibc
        return False


def install():
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())


def _report_whole_image():
    """One line at exit: which path the val images took (recon.sparse_neus_renderer.WHOLE_IMAGE_TOTALS) -- rendered whole behind the trainer's chunk loop, or
    fallen back to per-chunk calls and why (e.g. "rng" under nn.DataParallel on several devices: the device threads share torch's host generator)."""
    mod = sys.modules.get(f"{PKG}.recon.sparse_neus_renderer") or sys.modules.get("models.sparse_neus_renderer")
    st = getattr(mod, "WHOLE_IMAGE_TOTALS", None)
    if st and (st["images"] or st["plain_calls"]):
        print(f"o2345 render(): {st['images']} image(s) rendered whole, {st['chunks_served']} chunk(s) served from them, {st['plain_calls']} plain call(s), "
              f"fallbacks by reason: {st['fallbacks_by_reason'] or 'none'}", file=sys.stderr)


def activate():
    """Everything the launcher does before the reference script starts: thread-pool sizes (before anything imports torch: the pools read them at start-up;
    an explicit OMP_NUM_THREADS of the user wins), the host allocator setting, the import hook, the exit report.  Idempotent."""
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(cpu_threads()))
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        tune_host_allocator()
        install()
        import atexit
        atexit.register(_report_whole_image)


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    activate()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
