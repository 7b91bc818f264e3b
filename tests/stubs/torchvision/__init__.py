"""TEST INFRASTRUCTURE: the two torchvision entry points the reference's data / utility modules touch (torchvision is not installed here)."""
from . import transforms, utils  # noqa: F401

__version__ = "0.0-o2345-test-stub"
