"""TEST INFRASTRUCTURE (tests/run_reference_runner.py --via-autoload): the second `sitecustomize` on the path, chained to by
one-2-3-45_amd/autoload/sitecustomize.py.  Provides the one import the unchanged runner needs that neither this image nor tests/stubs can supply as a
plain package: `from torch.utils.tensorboard import SummaryWriter` (tensorboard is not installed; only train() writes to it)."""
import sys
import types

_tb = types.ModuleType("torch.utils.tensorboard")


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


_tb.SummaryWriter = SummaryWriter
sys.modules["torch.utils.tensorboard"] = _tb
