"""run.py drop-in WITHOUT editing run.py or its command line.

run.py starts the reconstruction as a child process, `python exp_runner_generic_blender_val.py --mode export_mesh ...` (run.py:61-67, os.system), so there is no
place to write `python -m o2345_amd.dropin`.  Put this directory FIRST on PYTHONPATH (children inherit it):

    PYTHONPATH=/path/to/repo/one-2-3-45_amd/autoload:/path/to/repo  python run.py --img_path ... --half_precision

Python imports `sitecustomize` at start-up of every interpreter; this one activates the HIP back end (o2345_amd.dropin.activate(): import hook for
torchsparse / inplace_abn / mcubes / trimesh-export / the four L1 model modules, thread-pool sizes, host allocator setting) ONLY when the interpreter's
script is the reference's reconstruction runner -- `exp_runner_generic_blender_*.py`, or a name listed in $O2345_AUTOLOAD_SCRIPTS (comma-separated basenames)
-- so run.py's own process (Zero123, SAM, `convert_mesh_format` with the real trimesh) is left exactly as it is.  $O2345_AUTOLOAD=0 switches it off.
A `sitecustomize` further down sys.path (e.g. the distribution's) is chained to."""
import fnmatch
import os
import sys


def _wanted():
    if os.environ.get("O2345_AUTOLOAD", "1") in ("", "0"):
        return False
    argv0 = os.path.basename((getattr(sys, "argv", None) or [""])[0] or "")
    names = [n.strip() for n in os.environ.get("O2345_AUTOLOAD_SCRIPTS", "").split(",") if n.strip()]
    return fnmatch.fnmatch(argv0, "exp_runner_generic_blender_*.py") or argv0 in names


def _chain():
    """Import the next `sitecustomize` on sys.path (this file shadows it), so that putting this directory first changes nothing else."""
    here = os.path.dirname(os.path.abspath(__file__))
    import importlib.machinery
    import importlib.util
    paths = [p for p in sys.path if os.path.abspath(p or ".") != here]
    spec = importlib.machinery.PathFinder.find_spec("sitecustomize", paths)
    if spec is not None and spec.origin and os.path.abspath(spec.origin) != os.path.abspath(__file__):
        mod = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(mod)
        except Exception:                      # noqa: BLE001  (a broken site hook of the distribution must not stop the interpreter; site.py would print and continue too)
            pass


if _wanted():
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # the repository root (holds o2345_amd.py)
    if _root not in sys.path:
        sys.path.insert(1, _root)
    import importlib
    importlib.import_module("one-2-3-45_amd.dropin").activate()
_chain()
