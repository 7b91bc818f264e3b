import numpy as np
import torch


class ToTensor:
    def __call__(self, img):
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[..., None]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

    def __call__(self, x):
        return (x - self.mean) / self.std
