"""Build container only: the reference's UNCHANGED entry point -- reconstruction/exp_runner_generic_blender_val.py, `--mode export_mesh`, the way
run.py invokes it (run.py:61-67) -- through the drop-in import hook, in its own process (tests/run_reference_runner.py).

What runs that is the reference's own code: argument parsing, Runner.__init__ (HOCON conf -> networks built from conf['model.*'] kwargs ->
GenericTrainer -> BlenderPerView datasets + DataLoaders -> Adam -> checkpoint discovery + load_checkpoint with its try/except "load fails" ->
nn.DataParallel), Runner.export_mesh, GenericTrainer.forward / export_mesh_step / validate_colored_mesh, the dataset's __getitem__.
What is ours: the five classes the Runner imports (mirrors), the third-party shims, the PLY writer.  There is no GPU here: the ops layer
is the oracle-backed CPU stand-in (tests/fake_ops.py) and cuda devices are redirected to the CPU; the SAME script without --fake-ops is
run on the MI355X against the real ctypes ops (tools/runner_on_gpu.sh, log under profiles/)."""
import json
import os
import subprocess
import sys

import pytest

from oracle import ref_import as RI

pytestmark = [pytest.mark.reference, pytest.mark.skipif(not RI.available(), reason="/root/reference not present")]
HERE = os.path.dirname(os.path.abspath(__file__))


def test_export_mesh_through_the_unchanged_runner(tmp_path):
    cmd = [sys.executable, os.path.join(HERE, "run_reference_runner.py"), "--ref", RI.REF, "--work", str(tmp_path), "--fake-ops", "--workers", "0", "--",
           "--mode", "export_mesh", "--conf", "confs/one2345_lod0_val_demo.conf", "--resolution", "24", "--specific_dataset_name", "scene0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RUNNER_RESULT ")]
    assert line, p.stdout[-2000:]
    res = json.loads(line[-1][len("RUNNER_RESULT "):])
    # the Runner's own load path accepted every network of the synthetic checkpoint (exp_runner...:437-451 prints "<name> load fails" otherwise
    # and silently continues with random weights) and the optimizer state
    assert not res["load_fails_printed"] and not res["optimizer_load_fails_printed"], res
    assert "Find checkpoint: ckpt_030000.pth" in p.stderr + p.stdout and "iter_step:  30000" in p.stdout
    assert res["mesh"].endswith("scene0/mesh.ply") and res["vertices"] > 100 and res["triangles"] > 200 and res["has_vertex_colours"], res
    # nothing of the reference was modified: the scratch tree consists of symlinks + exp/
    rec = os.path.join(str(tmp_path), "reconstruction")
    assert all(os.path.islink(os.path.join(rec, n)) for n in os.listdir(rec) if n != "exp")


def test_hocon_stub_reads_the_reference_confs():
    sys.path.insert(0, os.path.join(HERE, "stubs"))
    try:
        import importlib
        ph = importlib.import_module("pyhocon")
        c = ph.ConfigFactory.parse_file(os.path.join(RI.REF, "confs", "one2345_lod0_val_demo.conf"))
        assert c["general.base_exp_dir"] == "exp/lod0" and c.get_int("model.num_lods") == 1 and c.get_bool("train.use_white_bkgd") is True
        assert dict(c["model.sdf_network_lod0"]) == {"lod": 0, "ch_in": 56, "voxel_size": 0.02105263, "vol_dims": [96, 96, 96], "hidden_dim": 128,
                                                      "cost_type": "variance_mean", "d_pyramid_feature_compress": 16, "regnet_d_out": 16,
                                                      "num_sdf_layers": 4, "multires": 6}
        assert c.get_list("general.recording") == ["./", "./data", "./ops", "./models", "./loss"] and c.get_float("train.anneal_end", default=0) == 25000.0
        assert c.get_string("dataset.test_split", default="test") == "test" and c.get_float("train.anneal_start_lod1", default=0) == 0.0
        c["general.base_exp_dir"] = "elsewhere"
        assert c["general.base_exp_dir"] == "elsewhere"
    finally:
        sys.path.remove(os.path.join(HERE, "stubs"))
        for k in [k for k in sys.modules if k.split(".")[0] == "pyhocon"]:
            del sys.modules[k]
