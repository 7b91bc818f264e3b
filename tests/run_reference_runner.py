"""TEST INFRASTRUCTURE: run the reference's UNCHANGED entry point (reconstruction/exp_runner_generic_blender_val.py -- Runner: HOCON conf ->
network construction from conf['model.*'] -> checkpoint load -> BlenderPerView + DataLoader -> GenericTrainer under nn.DataParallel ->
export_mesh / val) on this back end through ``python -m o2345_amd.dropin``-style aliasing.

    python tests/run_reference_runner.py --ref <.../reconstruction> --work <scratch dir> [--fake-ops] [--views-hw 256] -- \\
        --mode export_mesh --conf confs/one2345_lod0_val_demo.conf --resolution 64 --specific_dataset_name scene0

What this script adds around the reference files (none of which is modified or copied into the repository):
  * a scratch tree ``<work>/reconstruction`` whose entries are SYMLINKS to the reference's files (the Runner needs cwd = reconstruction/ and
    writes ``exp/lod0`` relative to it, exp_runner...:46-52, conf :6), a synthetic Zero123-style folder ``<work>/<name>`` (dataset.write_synthetic_folder)
    and a synthetic ``exp/lod0/checkpoints/ckpt_XXXXXX.pth`` with the reference's checkpoint keys (``sdf_network_lod0``, ``rendering_network_lod0``,
    ``variance_network_lod0``, ``pyramid_feature_network``, ``optimizer``, ``iter_step``; :485-512, :514-541) -- the released ckpt_215000.pth cannot be
    downloaded here;
  * import stubs for third-party packages missing from this image (tests/stubs: pyhocon, cv2, kornia, torchvision, icecream; tensorboard's
    SummaryWriter);
  * with ``--fake-ops`` (build container, no GPU): oracle-backed CPU stand-ins for the ops layer (tests/fake_ops.py) and a cuda -> cpu redirection
    of the Runner's hard-coded ``torch.device('cuda:%d')`` (:31).  WITHOUT the flag (MI355X) nothing is patched: the Runner's tensors reach the
    real ctypes ops of libo2345_hip.so.

Prints one JSON line ``RUNNER_RESULT {...}`` (mesh path, vertex / triangle counts, whether "load fails" was printed, native library loaded)."""
import argparse
import contextlib
import importlib
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build_tree(ref, work):
    rec = os.path.join(work, "reconstruction")
    os.makedirs(rec, exist_ok=True)
    for name in os.listdir(ref):
        if name in ("exp", "__pycache__"):
            continue
        dst = os.path.join(rec, name)
        if not os.path.lexists(dst):
            os.symlink(os.path.join(ref, name), dst)
    os.makedirs(os.path.join(rec, "exp", "lod0", "checkpoints"), exist_ok=True)
    return rec


def synth_checkpoint(rec, conf_path, iter_step, variance, seed=0):
    """A seeded checkpoint in the reference's format, built from OUR mirrors constructed with the conf's kwargs exactly like the Runner does
    (exp_runner...:93-106): if a mirror's state-dict keys or shapes differed from what the Runner's load path expects, "load fails" would be printed."""
    import torch
    from pyhocon import ConfigFactory
    pkg = "one-2-3-45_amd"
    conf = ConfigFactory.parse_file(conf_path)
    feat = importlib.import_module(f"{pkg}.featurenet")
    sdfm = importlib.import_module(f"{pkg}.recon.sparse_sdf_network")
    renm = importlib.import_module(f"{pkg}.recon.rendering_network")
    fld = importlib.import_module(f"{pkg}.recon.fields")
    torch.manual_seed(seed)
    nets = {"pyramid_feature_network": feat.FeatureNet(), "sdf_network_lod0": sdfm.SparseSdfNetwork(**conf["model.sdf_network_lod0"]),
            "variance_network_lod0": fld.SingleVarianceNetwork(**conf["model.variance_network"]),
            "rendering_network_lod0": renm.GeneralRenderingNetwork(**conf["model.rendering_network"])}
    g = torch.Generator().manual_seed(seed + 1)
    L = nets["sdf_network_lod0"].sdf_layer                          # geometric init zeroes the latent columns: make the volume matter (seeded)
    L.lin1.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
    L.lin2.weight_v.data[:, 128:] += 0.03 * torch.randn(128, 16, generator=g)
    nets["variance_network_lod0"].variance.data = torch.tensor(float(variance))
    params = [p for k in ("pyramid_feature_network", "sdf_network_lod0", "variance_network_lod0", "rendering_network_lod0") for p in nets[k].parameters()]
    ckpt = {k: n.state_dict() for k, n in nets.items()}
    ckpt["optimizer"] = torch.optim.Adam(params, lr=conf.get_float("train.learning_rate")).state_dict()     # trainer_generic.py:133-156 order
    ckpt["iter_step"], ckpt["val_step"] = int(iter_step), 0
    path = os.path.join(rec, "exp", "lod0", "checkpoints", "ckpt_{:0>6d}.pth".format(int(iter_step)))
    torch.save(ckpt, path)
    return path, {k: {kk: tuple(v.shape) for kk, v in n.state_dict().items()} for k, n in nets.items()}


class _Patch:
    """monkeypatch-like setattr (no undo needed: the process ends with the run)."""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def redirect_cuda_to_cpu():
    """Build container only: the Runner hard-codes cuda devices (exp_runner...:31, :612); keep its code path, land on the CPU."""
    import torch
    cpu = torch.device("cpu")
    is_cuda = lambda d: (isinstance(d, torch.device) and d.type == "cuda") or (isinstance(d, str) and d.startswith("cuda"))
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    torch.cuda.empty_cache = lambda: None
    m_to, t_to, load = torch.nn.Module.to, torch.Tensor.to, torch.load
    fix = lambda a: tuple(cpu if is_cuda(x) else x for x in a)
    fixk = lambda k: {kk: (cpu if kk == "device" and is_cuda(v) else v) for kk, v in k.items()}
    torch.nn.Module.to = lambda self, *a, **k: m_to(self, *fix(a), **fixk(k))
    torch.Tensor.to = lambda self, *a, **k: t_to(self, *fix(a), **fixk(k))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.load = lambda f, *a, **k: load(f, *a, **dict(k, map_location=cpu))
    dl = torch.utils.data.DataLoader
    orig_init = dl.__init__

    def init(self, *a, **k):
        k["pin_memory"] = False
        orig_init(self, *a, **k)
    dl.__init__ = init


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True, help="the reference's reconstruction/ directory")
    ap.add_argument("--work", required=True)
    ap.add_argument("--fake-ops", action="store_true")
    ap.add_argument("--iter-step", type=int, default=30000, help="checkpoint iter_step: drives alpha_inter_ratio (exp_runner...:412-418; >= 25000 -> 1.0)")
    ap.add_argument("--variance", type=float, default=0.2)
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--workers", type=int, default=None, help="override the Runner's 4 x batch_size DataLoader workers (0 on the CPU stand-ins)")
    ap.add_argument("--via-autoload", action="store_true", help="start the runner the way run.py does (run.py:61-67): a child `python exp_runner_generic_blender_val.py ...` "
                                                                "with one-2-3-45_amd/autoload first on PYTHONPATH -- no `-m o2345_amd.dropin` on the command line")
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
    rp = argparse.ArgumentParser()
    rp.add_argument("--conf", default="./confs/base.conf")
    rp.add_argument("--mode", default="train")
    rp.add_argument("--specific_dataset_name", default="GSO")
    ra, _ = rp.parse_known_args(rest)

    for p in (ROOT, HERE, os.path.join(HERE, "stubs")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True
    import torch
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:                                           # tensorboard is not installed; only train() writes to it
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb

    rec = build_tree(os.path.abspath(a.ref), os.path.abspath(a.work))
    ds = importlib.import_module("one-2-3-45_amd.dataset")
    ds.write_synthetic_folder(os.path.abspath(a.work), ra.specific_dataset_name, seed=0, hw=(a.hw, a.hw))
    ckpt, shapes = synth_checkpoint(rec, os.path.join(rec, ra.conf), a.iter_step, a.variance)

    dropin = importlib.import_module("one-2-3-45_amd.dropin")
    if a.fake_ops:
        import fake_ops
        fake_ops.install(_Patch())
        redirect_cuda_to_cpu()
    if a.workers is not None:
        dl = torch.utils.data.DataLoader
        oi = dl.__init__

        def init(self, *aa, **kk):
            kk["num_workers"] = a.workers
            oi(self, *aa, **kk)
        dl.__init__ = init

    os.chdir(rec)
    out_dir = os.path.join(os.path.abspath(a.work), ra.specific_dataset_name)
    if a.via_autoload:
        import subprocess
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "one-2-3-45_amd", "autoload"), ROOT, os.path.join(HERE, "stubs"),
                                                          os.path.join(HERE, "stubs_site")] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
        cmd = [sys.executable, "exp_runner_generic_blender_val.py"] + rest                      # run.py's command, verbatim
        print("RUNNER_COMMAND " + " ".join(cmd) + "   (PYTHONPATH=" + env["PYTHONPATH"] + ")")
        p = subprocess.run(cmd, cwd=rec, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
        log = p.stdout
        print(log)
        res = {"mode": ra.mode, "via_autoload": True, "rc": p.returncode, "checkpoint": os.path.relpath(ckpt, rec), "iter_step": a.iter_step,
               "load_fails_printed": "load fails" in log, "optimizer_load_fails_printed": "load optimizer fails" in log, "fake_ops": False,
               "whole_image_report_printed": "o2345 render():" in log}
        mesh = os.path.join(out_dir, "mesh.ply")
        if os.path.exists(mesh):
            mio = importlib.import_module("one-2-3-45_amd.mesh_io")
            v, f, c = mio.read_ply(mesh)
            res.update(mesh=mesh, vertices=int(v.shape[0]), triangles=int(f.shape[0]), has_vertex_colours=c is not None)
        print("RUNNER_RESULT " + json.dumps(res))
        raise SystemExit(p.returncode)
    sys.argv = ["o2345_amd.dropin", "exp_runner_generic_blender_val.py"] + rest
    buf = io.StringIO()

    class Tee(io.TextIOBase):
        def __init__(self, *s):
            self.s = s

        def write(self, x):
            for s in self.s:
                s.write(x)
            return len(x)

        def flush(self):
            for s in self.s:
                s.flush()
    import rich
    import builtins
    rich_print = rich.print
    rich.print = lambda *aa, **kk: builtins.print(*aa, **{k: v for k, v in kk.items() if k in ("sep", "end", "file", "flush")})   # plain text into the tee
    with contextlib.redirect_stdout(Tee(sys.__stdout__, buf)):
        if os.environ.get("O2345_RUNNER_PROFILE"):                  # host-side view of the trainer's own brackets: what the first step of a fresh process spends where
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            try:
                dropin.main()
            finally:
                pr.disable()
                st = pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative")
                for fn in ("export_mesh_step", "validate_colored_mesh", "val_step", "extract_geometry", "compute_view_independent"):
                    st.print_callees(fn)
        else:
            dropin.main()
    rich.print = rich_print
    log = buf.getvalue()
    res = {"mode": ra.mode, "checkpoint": os.path.relpath(ckpt, rec), "iter_step": a.iter_step, "load_fails_printed": "load fails" in log,
           "optimizer_load_fails_printed": "load optimizer fails" in log, "fake_ops": bool(a.fake_ops), "cuda": bool(torch.cuda.is_available()) and not a.fake_ops}
    mesh = os.path.join(out_dir, "mesh.ply")
    if os.path.exists(mesh):
        mio = importlib.import_module("one-2-3-45_amd.mesh_io")
        v, f, c = mio.read_ply(mesh)
        res.update(mesh=mesh, vertices=int(v.shape[0]), triangles=int(f.shape[0]), has_vertex_colours=c is not None)
    res["files_written"] = sorted(os.path.relpath(os.path.join(dp, fn), out_dir) for dp, _, fns in os.walk(out_dir) for fn in fns
                                  if not fn.endswith(".png") or "stage" not in dp)[:40]
    if not a.fake_ops:
        lib = importlib.import_module("one-2-3-45_amd._lib")
        res["native_library"] = os.path.basename(lib.lib()._name)
    print("RUNNER_RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
