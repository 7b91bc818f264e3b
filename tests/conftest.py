import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name is not a python identifier)."""
    return importlib.import_module("one-2-3-45_amd")


_LIB_INSTANCES = {}


@pytest.fixture
def lib_instance(monkeypatch, tmp_path_factory):
    """``use(env={...}, variant=None)``: make a SEPARATE instance of the HIP library current for the rest of the test -- the product library (or the
    build variant libo2345_hip_<variant>.so) copied to a private path, so that it has its own statics, loaded with ``env`` set: the library reads its
    O2345_* debug knobs from the environment once per instance (csrc/common.h knobs()).  Skips when the variant library was not built."""
    import shutil
    L = importlib.import_module("one-2-3-45_amd._lib")

    def use(env=None, variant=None):
        env = dict(env or {})
        key = (variant, tuple(sorted(env.items())))
        if key not in _LIB_INSTANCES:
            src = L.LIB_PATH if variant is None else os.path.join(L.HERE, f"libo2345_hip_{variant}.so")
            if not os.path.exists(src):
                pytest.skip(f"{src} is not built (python __graft_entry__.py build)")
            dst = os.path.join(str(tmp_path_factory.mktemp("libvariant")), os.path.basename(src))
            shutil.copy(src, dst)
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                _LIB_INSTANCES[key] = L.load_library(dst)       # load_library reads the knobs (o2345_knobs) while env is set
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        monkeypatch.setattr(L, "_LIB", _LIB_INSTANCES[key])
        return _LIB_INSTANCES[key]
    return use
