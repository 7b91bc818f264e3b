"""TEST INFRASTRUCTURE: the handful of OpenCV calls the reference's data / visualisation code makes (cv2 is not installed in this image),
on numpy / scipy / PIL.  Used only to run the reference's unchanged Runner in tests (tests/run_reference_runner.py)."""
import numpy as np

INTER_NEAREST, INTER_LINEAR, COLORMAP_JET, IMREAD_GRAYSCALE = 0, 1, 2, 0
FONT_HERSHEY_SIMPLEX = 0
__version__ = "0.0-o2345-test-stub"


def decomposeProjectionMatrix(P):
    """RQ decomposition of P[:, :3] with a positive diagonal + homogeneous camera centre (One2345_eval_new_data.py:42)."""
    from scipy.linalg import rq
    P = np.asarray(P, np.float64)
    K, R = rq(P[:, :3])
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R
    c = -np.linalg.inv(P[:, :3]) @ P[:, 3]
    return K, R, np.concatenate([c, [1.0]])[:, None]


def imread(path, flags=1):
    from PIL import Image
    try:
        im = Image.open(path)
    except Exception:
        return None
    if flags == 0:
        return np.asarray(im.convert("L"))
    return np.asarray(im.convert("RGB"))[..., ::-1].copy()


def imwrite(path, img):
    from PIL import Image
    a = np.asarray(img)
    if a.dtype != np.uint8:
        a = np.clip(a, 0, 255).astype(np.uint8)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[..., ::-1]
    Image.fromarray(a).save(path)
    return True


def resize(img, dsize, fx=None, fy=None, interpolation=INTER_NEAREST):
    a = np.asarray(img)
    if dsize:
        w, h = dsize
    else:
        h, w = int(round(a.shape[0] * fy)), int(round(a.shape[1] * fx))
    ys = np.minimum((np.arange(h) * (a.shape[0] / h)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum((np.arange(w) * (a.shape[1] / w)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


def applyColorMap(x, cmap=COLORMAP_JET):
    v = np.asarray(x).astype(np.float32) / 255.0
    r = np.clip(1.5 - np.abs(4 * v - 3), 0, 1)
    g = np.clip(1.5 - np.abs(4 * v - 2), 0, 1)
    b = np.clip(1.5 - np.abs(4 * v - 1), 0, 1)
    return (np.stack([b, g, r], -1) * 255).astype(np.uint8)


def putText(img, *a, **k):
    return img
