"""CPU: the drop-in boundary -- the C-ABI library loads and exports every symbol declared in include/o2345.h, the shim
packages expose the names the reference imports, state-dict keys equal the reference's, and the host-side logic
(header parser, weight packing, import hook) works without a GPU.  No compute call is made here."""
import importlib
import sys

import numpy as np
import pytest
import torch

from golden_util import load
from oracle import ref_import as RI

pkg = importlib.import_module("one-2-3-45_amd")


def test_cabi_exports_every_declared_symbol():
    L = importlib.import_module("one-2-3-45_amd._lib")
    protos = L.parse_header()
    assert len(protos) >= 30 and "o2345_render_rays" in protos and "o2345_marching_cubes_emit" in protos
    lib = L.lib()                                    # raises if the .so is missing or a declared symbol is not exported
    assert lib.o2345_version() == L.ABI_VERSION == 210
    assert lib.o2345_sdf_blob_floats() == pkg.weights.SDF_BLOB_FLOATS
    assert not hasattr(lib, "o2345_color_points") and "o2345_color_stats_enable" not in protos      # ABI 2.0: VALU colour kernel and library-global counters are gone
    assert lib.o2345_color_mfma_blob_floats() == pkg.weights.CM_BLOB_FLOATS
    assert lib.o2345_color_x3_blob_floats() == pkg.weights.CX_BLOB_FLOATS
    assert lib.o2345_costvol_workspace_bytes(128, 128, 128) > 0
    # error convention: non-zero status + message, no exception from C
    rc = lib.o2345_sdf_mlp(0, None, None, 8, None, None, None, 10, 0, 1.0, None, None, None, None, None)
    assert rc != 0 and b"null pointer" in lib.o2345_last_error()


def test_render_call_validates_its_arguments_before_touching_the_device():
    """o2345_render_rays rejects what it cannot do with a message that says why (the checks run on the host, ahead of every HIP call): the reference's
    n_importance is split into four rounds, sample slots are 32-bit, the caller sizes the workspace with o2345_render_workspace_bytes."""
    import ctypes
    L = importlib.import_module("one-2-3-45_amd._lib")
    lib = L.lib()
    lib.o2345_render_workspace_bytes.restype = ctypes.c_size_t
    ws_bytes = lib.o2345_render_workspace_bytes(512, 64, 64, 8)
    assert ws_bytes >= 512 * 128 * (4 * 2 + 12 + 4 + 1)                     # z, sdf, pts, list, occupancy for every sample slot
    assert lib.o2345_render_workspace_bytes(262144, 64, 64, 8) > lib.o2345_render_workspace_bytes(262144, 64, 64, 40)      # > 32 views: no list sort, no sort buffers
    dummy = ctypes.c_void_p(256)                                            # a non-null pointer that is never dereferenced: every call below fails in the checks

    def call(**kw):
        io = L.RenderIO()
        io.R, io.n_samples, io.n_importance, io.V, io.sdf_mode = 512, 64, 64, 8, 2
        io.color_x3_blob = 256
        for k, v in kw.items():
            setattr(io, k, v)
        rc = lib.o2345_render_rays(ctypes.byref(io), dummy, ctypes.c_size_t(ws_bytes), None)
        return rc, lib.o2345_last_error()

    rc, msg = call(n_importance=62)
    assert rc != 0 and b"multiple of 4" in msg
    rc, msg = call(R=1 << 24, n_samples=128, n_importance=128)
    assert rc != 0 and b"2^31" in msg
    rc, msg = call(n_samples=200, n_importance=64, R=4)
    assert rc != 0 and b"at most 256 samples" in msg
    rc, msg = call(color_x3_blob=None)
    assert rc != 0 and b"colour network blob" in msg
    rc, msg = call(sdf_mode=1)
    assert rc != 0 and b"SDF mode 1" in msg
    io = L.RenderIO()
    io.R, io.n_samples, io.n_importance, io.V, io.sdf_mode = 512, 64, 64, 8, 2
    io.color_x3_blob = 256
    rc = lib.o2345_render_rays(ctypes.byref(io), dummy, ctypes.c_size_t(ws_bytes - 1), None)
    assert rc != 0 and b"workspace too small" in lib.o2345_last_error()
    assert lib.o2345_render_rays(None, dummy, ctypes.c_size_t(ws_bytes), None) != 0 and b"null pointer" in lib.o2345_last_error()


def test_render_io_has_one_declaration_and_the_binding_checks_it(tmp_path):
    """O2345RenderIO is declared in include/o2345.h only: csrc/ compiles that header (common.h includes it), the ctypes Structure is generated from its
    text, and the loaded library's own sizeof / offsetof table is compared at load time.  A field added to ONE side only must fail loudly."""
    import ctypes
    import re
    L = importlib.import_module("one-2-3-45_amd._lib")
    lib = L.lib()
    fields = L.parse_struct("O2345RenderIO")
    names = [f for f, _ in fields]
    assert "sdf_mode" in names and "sdf_bf16" not in names and "color_blob" not in names and names[0] == "sdf_blob" and names[-1] == "color_stats"
    assert lib.o2345_render_io_layout(None, 0) == len(fields) and lib.o2345_render_io_size() == ctypes.sizeof(L.RenderIO)
    # no private copy of the struct anywhere in the library sources
    import glob
    import os
    for f in glob.glob(os.path.join(L.HERE, "csrc", "*")):
        assert not re.search(r"struct\s+O2345RenderIO\s*\{", open(f).read()), f
    # a header that gained a field the library was not compiled with (or lost one, or reordered two): the load-time check raises
    src = open(L.HEADER).read()
    for mutate in (lambda t: t.replace("int R, n_samples, n_importance;", "int R, n_samples, n_importance; int new_field;"),
                   lambda t: t.replace("    const float* t_rand;", "    /* removed */", 1),
                   lambda t: t.replace("float near, far;", "float far, near; float extra;")):
        hp = tmp_path / "o2345_mut.h"
        hp.write_text(mutate(src))
        assert hp.read_text() != src

        class Mut(ctypes.Structure):
            _fields_ = L.parse_struct("O2345RenderIO", str(hp))
        orig = L.RenderIO
        L.RenderIO = Mut
        try:
            with pytest.raises(RuntimeError, match="layout mismatch"):
                L.check_render_io_layout(lib)
        finally:
            L.RenderIO = orig
    L.check_render_io_layout(lib)


def test_import_hook_and_shims():
    dropin = importlib.import_module("one-2-3-45_amd.dropin")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("torchsparse", "inplace_abn", "mcubes", "models")}
    try:
        dropin.install()
        import torchsparse
        import torchsparse.nn as spnn
        from torchsparse.tensor import PointTensor, SparseTensor  # noqa: F401
        from torchsparse.nn.utils import get_kernel_offsets
        import torchsparse.nn.functional as F  # noqa: F401
        from inplace_abn import InPlaceABN
        import mcubes
        from models.sparse_sdf_network import SparseSdfNetwork
        from models.sparse_neus_renderer import SparseNeuSRenderer  # noqa: F401
        from models.rendering_network import GeneralRenderingNetwork  # noqa: F401
        assert SparseSdfNetwork.__module__.startswith("one-2-3-45_amd")
        assert callable(mcubes.marching_cubes) and hasattr(spnn, "Conv3d") and hasattr(spnn, "BatchNorm") and hasattr(spnn, "ReLU")
        off = get_kernel_offsets(3, 2)
        assert off.shape == (27, 3) and off[0].tolist() == [-2, -2, -2] and off[1].tolist() == [0, -2, -2]      # x fastest
        conv = spnn.Conv3d(32, 16, kernel_size=3, stride=2)
        assert tuple(conv.kernel.shape) == (27, 32, 16)
        abn = InPlaceABN(16)
        assert set(abn.state_dict()) == {"weight", "bias", "running_mean", "running_var"}
        with pytest.raises(RuntimeError):
            abn(torch.zeros(1, 16, 4, 4))           # HIP-only: must fail loudly on CPU tensors, not fall back
        assert torchsparse.__version__.startswith("1.4.0")
    finally:
        sys.meta_path[:] = [f for f in sys.meta_path if type(f).__name__ != "_AliasFinder"]
        for k in list(sys.modules):
            if k.split(".")[0] in ("torchsparse", "inplace_abn", "mcubes", "models"):
                del sys.modules[k]
        sys.modules.update(saved)


def _mirror():
    recon = importlib.import_module("one-2-3-45_amd.recon")
    sdf = recon.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=2 / 95, vol_dims=[96, 96, 96], hidden_dim=128, cost_type="variance_mean",
                                 d_pyramid_feature_compress=16, regnet_d_out=16, num_sdf_layers=4, multires=6)
    return recon, sdf, recon.GeneralRenderingNetwork(16, 56, True), recon.SingleVarianceNetwork(0.2)


def test_state_dict_keys_match_golden_reference_weights():
    G = load()
    _, sdf, rnet, var = _mirror()
    mine = {k: tuple(v.shape) for k, v in sdf.state_dict().items()}
    for k, v in G["sdf_sd"].items():                 # every reference parameter exists with the same shape
        assert mine.get(k) == tuple(v.shape), k
    assert {k: tuple(v.shape) for k, v in rnet.state_dict().items()} == {k: tuple(v.shape) for k, v in G["ren_sd"].items()}
    assert set(var.state_dict()) == set(G["var_sd"])
    sdf.load_state_dict(G["sdf_sd"], strict=False)
    rnet.load_state_dict(G["ren_sd"])


@pytest.mark.reference
@pytest.mark.skipif(not RI.available(), reason="/root/reference not present")
def test_state_dict_keys_match_reference_modules():
    ref_sdf, ref_rnet, ref_var, _ = RI.build_networks(96, seed=0, voxel_size=2 / 95)
    _, sdf, rnet, var = _mirror()
    shapes = lambda m: {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes(sdf) == shapes(ref_sdf)
    assert shapes(rnet) == shapes(ref_rnet) and shapes(var) == shapes(ref_var)
    # same geometric initialisation statistics (sparse_sdf_network.py:75-100)
    assert abs(float(sdf.sdf_layer.lin2.weight_v.mean()) - float(ref_sdf.sdf_layer.lin2.weight_v.mean())) < 1e-3
    assert float(sdf.sdf_layer.lin0.weight_v[:, 3:].abs().max()) == 0.0


def test_synthetic_scene_contract():
    sc = pkg.synth.make_scene(32)
    assert sc["images"].shape == (32, 3, 256, 256) and sc["affine_mats"].shape == (32, 4, 4)
    assert np.allclose(sc["intrinsics"][0], [[280, 0, 128], [0, 280, 128], [0, 0, 1]])
    assert abs(float(sc["scale_mat"][0, 0]) - 1.42) < 0.01 and sc["query_near_far"][0] < 0
    sc8 = pkg.synth.make_scene(8)
    assert np.array_equal(sc8["w2cs"], sc["w2cs"][0::4])                                            # one stage-2 view per stage-1 view
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256, scale=2)
    assert ro.shape == (512 * 512, 3) and np.allclose(np.linalg.norm(rd, axis=1), 1, atol=1e-5)


def test_precision_config(pkg, monkeypatch):
    """config.py: the numerical mode of the network kernels (default f16x3; fp32 strict)."""
    import importlib
    cfg = importlib.import_module("one-2-3-45_amd.config")
    assert cfg.PRECISION in cfg.PRECISIONS
    assert cfg.sdf_precision("fp32") == "fp32" and cfg.color_precision("fp32") == "fp32"
    assert cfg.color_precision(None) in ("f16x3", "fp32")
    with __import__("pytest").raises(ValueError):
        cfg.sdf_precision("fp8")
    # packing of every blob the default mode needs works without a GPU and has the size the library expects
    W = pkg.weights
    L = pkg._lib.lib()
    sd = W.init_color_state_dict(1)
    assert W.pack_color_x3_blob(sd).size == L.o2345_color_x3_blob_floats()
    cs = W.init_costreg_state_dict(1)
    for name, ci, co in W.COSTREG_LAYERS:
        K = cs[f"{name}.net.0.kernel"]
        assert W.pack_sparse_conv_x3(K).size == L.o2345_sparse_conv_x3_blob_floats(K.shape[1], K.shape[2])


def test_invalidate_packed_drops_every_weight_cache():
    """Parameter updates through `.data` (EMA, weight surgery) bump no version counter: featurenet.invalidate_packed is the documented hook.  It must
    reset the cache key of every module that keeps packed operands (conv packings, SDF blob, colour blobs, packed sparse CNN)."""
    import importlib
    import torch
    fn = importlib.import_module("one-2-3-45_amd.featurenet")
    sdfm = importlib.import_module("one-2-3-45_amd.recon.sparse_sdf_network")
    renm = importlib.import_module("one-2-3-45_amd.recon.rendering_network")
    net = torch.nn.ModuleList([fn.FeatureNet(), sdfm.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=0.1, vol_dims=[8, 8, 8], regnet_d_out=16),
                               renm.GeneralRenderingNetwork(in_geometry_feat_ch=16)])
    keys = ("_o2345_packed_key", "_blob_key", "_key", "_costreg_key")
    for m in net.modules():
        for k in keys:
            if hasattr(m, k):
                setattr(m, k, ("stale",))
    net[0].conv0[0].conv._o2345_packed_key = ("stale",)
    fn.invalidate_packed(net)
    assert all(getattr(m, k, None) is None for m in net.modules() for k in keys)


def test_replaced_parameter_objects_are_repacked():
    """load_state_dict(assign=True) / `m.weight = nn.Parameter(...)` REPLACE the Parameter objects.  The packed operand caches (colour blobs, SDF blob,
    packed sparse CNN) must follow the modules' CURRENT parameters -- a cached parameter list keyed on the dead objects served stale weights."""
    import torch
    sdfm = importlib.import_module("one-2-3-45_amd.recon.sparse_sdf_network")
    renm = importlib.import_module("one-2-3-45_amd.recon.rendering_network")
    torch.manual_seed(0)
    rn = renm.GeneralRenderingNetwork(in_geometry_feat_ch=16)
    assert sorted(id(p) for p in rn._params()) == sorted(id(p) for p in rn.parameters())          # the fixed-order list is complete
    b0 = rn.x3_blob().clone()
    torch.manual_seed(1)
    other = renm.GeneralRenderingNetwork(in_geometry_feat_ch=16)
    rn.load_state_dict(other.state_dict(), assign=True)
    assert torch.equal(rn.x3_blob(), other.x3_blob()) and not torch.equal(rn.x3_blob(), b0)
    rn.rgb_fc[4].weight = torch.nn.Parameter(rn.rgb_fc[4].weight.detach() * 2)                    # direct replacement of one object
    fresh = renm.GeneralRenderingNetwork(in_geometry_feat_ch=16)
    fresh.load_state_dict(rn.state_dict())
    assert torch.equal(rn.x3_blob(), fresh.x3_blob()) and torch.equal(rn.mfma_blob(), fresh.mfma_blob())
    # SDF layer + sparse CNN
    torch.manual_seed(2)
    sn = sdfm.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=0.1, vol_dims=[8, 8, 8], regnet_d_out=16)
    torch.manual_seed(3)
    so = sdfm.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=0.1, vol_dims=[8, 8, 8], regnet_d_out=16)
    so.sdf_layer.lin1.weight_v.data[:, 128:] += 0.05
    b0 = sn.sdf_layer.blob().clone()
    c0 = sn._costreg("cpu")
    assert sn._costreg("cpu") is c0                                                               # unchanged parameters: no re-pack
    sn.load_state_dict(so.state_dict(), assign=True)
    assert torch.equal(sn.sdf_layer.blob(), so.sdf_layer.blob()) and not torch.equal(sn.sdf_layer.blob(), b0)
    c1 = sn._costreg("cpu")
    assert c1 is not c0
    for name in c1.p:
        assert all(torch.equal(a, b) for a, b in zip(c1.p[name], so._costreg("cpu").p[name]))


def test_launcher_tunes_the_host_allocator(monkeypatch):
    """dropin.tune_host_allocator(): glibc's mmap threshold raised to its maximum so that the trainer's 8 MB numpy temporaries are not munmap'ed under the GPU
    driver's MMU notifiers (24 - 30 ms per first use on the MI355X host); O2345_MALLOC_TUNE=0 leaves the allocator alone."""
    import importlib
    dropin = importlib.import_module("one-2-3-45_amd.dropin")
    monkeypatch.setenv("O2345_MALLOC_TUNE", "0")
    assert dropin.tune_host_allocator() is False
    monkeypatch.delenv("O2345_MALLOC_TUNE")
    assert dropin.tune_host_allocator() is True                  # glibc in this image
    import numpy as np
    a = np.ones(1 << 20)                                          # 8 MB: served and released without trouble afterwards
    assert float((a * 2 + 1).sum()) == 3.0 * (1 << 20)


def test_whole_image_switch_off_survives_data_parallel_replication():
    """nn.DataParallel on several devices rebuilds its per-device replicas from the module on EVERY forward (shallow __dict__ copies): the renderer's
    "whole-image mode switched itself off" state and its counters are shared by reference, so an image abandoned on a replica (the device threads share
    torch's host generator -> reason "rng") disables the speculation for the following forwards too instead of costing one wasted image per forward."""
    import importlib
    recon = importlib.import_module("one-2-3-45_amd.recon")

    class Conf(dict):
        def get_int(self, k, default=None):
            return int(self.get(k, default))
    sdf = recon.SparseSdfNetwork(lod=0, ch_in=56, voxel_size=2.0 / 95, vol_dims=[96] * 3, hidden_dim=128, cost_type="variance_mean",
                                 d_pyramid_feature_compress=16, regnet_d_out=16, num_sdf_layers=4, multires=6)
    ren = recon.SparseNeuSRenderer(None, sdf, recon.SingleVarianceNetwork(0.2), recon.GeneralRenderingNetwork(16, 56, True), 64, 64, 0, 1.0,
                                   alpha_type="div", conf=Conf({"general.base_exp_dir": "/tmp"}))
    assert ren.whole_image_stats() == dict(images=0, chunks_served=0, plain_calls=0, fallbacks_by_reason={}, enabled=True)
    rep = ren._replicate_for_data_parallel()
    rep._count("fallbacks_by_reason", "rng")
    rep._abandoned += 1
    assert ren._abandoned == 1 and not ren.whole_image_stats()["enabled"] and ren.whole_image_stats()["fallbacks_by_reason"] == {"rng": 1}
    rep2 = ren._replicate_for_data_parallel()                     # the next forward's replica starts switched off
    assert rep2._abandoned == 1 and not rep2.whole_image_stats()["enabled"]


def test_autoload_sitecustomize_activates_only_for_the_reference_runner(tmp_path):
    """run.py builds `python exp_runner_generic_blender_val.py ...` itself and starts it with os.system (run.py:61-67): with one-2-3-45_amd/autoload first on
    PYTHONPATH the child interpreter gets the import hook (the reference's module names resolve to this package, thread-pool sizes set), while any other
    script -- run.py's own process with Zero123, SAM and the real trimesh -- is left alone; O2345_AUTOLOAD=0 switches it off."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = textwrap.dedent("""
        import os, sys
        try:
            import torchsparse, torchsparse.nn, inplace_abn, mcubes, trimesh
            from models.sparse_sdf_network import SparseSdfNetwork
            from models.sparse_neus_renderer import SparseNeuSRenderer
            from models.rendering_network import GeneralRenderingNetwork
            from models.featurenet import FeatureNet
            print("HOOKED", torchsparse.__name__, SparseSdfNetwork.__module__, FeatureNet.__module__, os.environ.get("OMP_NUM_THREADS"), "sitecustomize_chained" in os.environ)
        except ImportError as e:
            print("PLAIN", type(e).__name__)
    """)
    (tmp_path / "exp_runner_generic_blender_val.py").write_text(probe)
    (tmp_path / "run.py").write_text(probe)
    later = tmp_path / "later"
    later.mkdir()
    (later / "sitecustomize.py").write_text("import os\nos.environ['sitecustomize_chained'] = '1'\n")      # a site hook further down the path still runs
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "O2345_AUTOLOAD")}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "one-2-3-45_amd", "autoload"), root, str(later)])
    run = lambda script, **kw: subprocess.run([sys.executable, str(tmp_path / script)], capture_output=True, text=True, timeout=300, cwd=tmp_path, env=dict(env, **kw))
    r = run("exp_runner_generic_blender_val.py")
    assert r.returncode == 0 and r.stdout.split()[:4] == ["HOOKED", "one-2-3-45_amd.shims.torchsparse", "one-2-3-45_amd.recon.sparse_sdf_network", "one-2-3-45_amd.featurenet"], (r.stdout, r.stderr[-800:])
    assert r.stdout.split()[4] != "None" and r.stdout.split()[5] == "True"
    assert run("run.py").stdout.split()[0] == "PLAIN"
    assert run("exp_runner_generic_blender_val.py", O2345_AUTOLOAD="0").stdout.split()[0] == "PLAIN"
    assert run("run.py", O2345_AUTOLOAD_SCRIPTS="run.py").stdout.split()[0] == "HOOKED"
