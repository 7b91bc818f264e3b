import torch


def make_grid(tensor, nrow=8, padding=0, **kw):
    t = tensor if torch.is_tensor(tensor) else torch.stack(list(tensor))
    return torch.cat(list(t), -1) if t.dim() == 4 else t


def save_image(*a, **k):
    raise NotImplementedError("torchvision stub: save_image")
