"""TEST INFRASTRUCTURE: kornia.create_meshgrid as data/One2345_eval_new_data.py:24 uses it (kornia is not installed in this image)."""
import torch

__version__ = "0.0-o2345-test-stub"


def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs, ys = (xs / (width - 1) - 0.5) * 2, (ys / (height - 1) - 0.5) * 2
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    return torch.stack([gx, gy], -1)[None]
