"""Torch-tensor wrappers over the C ABI (include/o2345.h).  PyTorch is plumbing only: device memory + streams.
Every function validates dtype / device / contiguity, passes raw pointers, and raises on a non-zero status."""
import ctypes
import functools

import numpy as np
import torch

from . import _lib, config
from ._lib import check


def _stream():
    """The current HIP stream of the CURRENT device; every public op runs under _on_device, which makes the device of its
    tensor arguments current first (the C side sizes persistent grids from hipGetDevice())."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _first_cuda_tensor(objs):
    for o in objs:
        if torch.is_tensor(o):
            if o.is_cuda:
                return o
        elif isinstance(o, dict):
            t = _first_cuda_tensor(o.values())
            if t is not None:
                return t
    return None


def _on_device(fn):
    """Run the op with the device of its (first) CUDA tensor argument current, so that the launch stream, the workspace and
    the C side's device queries all refer to the tensors' device even when the caller never called torch.cuda.set_device."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        t = _first_cuda_tensor(list(args) + list(kw.values()))
        if t is None:
            raise ValueError(f"o2345 ops.{fn.__name__}: needs CUDA tensors (the HIP library has no CPU fallback)")
        with torch.cuda.device(t.device):
            return fn(*args, **kw)
    return wrapped


def _p(t, dtype=torch.float32):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise ValueError(f"expected contiguous cuda {dtype} tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return ctypes.c_void_p(t.data_ptr())


def _f(t):
    return t.detach().to(torch.float32).contiguous()


def _host3(v):
    a = np.ascontiguousarray(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v, np.float32).reshape(-1)[:3])
    return a, a.ctypes.data_as(ctypes.c_void_p)


def to_host_numpy(*tensors):
    """Device tensors -> numpy arrays: the copies go into pinned blocks of torch's caching host allocator, queued back to back, ONE stream
    synchronisation at the end (a pageable ``.cpu()`` per tensor synchronises per tensor and stages through a bounce buffer: the 67 MB extraction field
    took 6 - 60 ms that way, depending on what was still in flight).  Tensors that already live on the host are returned as they are."""
    out, dev = [], None
    for t in tensors:
        if t.is_cuda:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            dev = t.device
            out.append(h)
        else:
            out.append(t)
    if dev is not None:
        torch.cuda.current_stream(dev).synchronize()
    return [h.numpy() for h in out]


_preloaded = set()


def preload(device):
    """Load every code object of the library on ``device`` now (o2345_preload) -- once per device and process.  The mirrors call it when weights are loaded,
    so that the HIP runtime's lazy per-translation-unit loading (5 - 60 ms each) does not land inside the first timed call of a fresh process."""
    device = torch.device(device)
    if device.type != "cuda" or device in _preloaded:
        return
    with torch.cuda.device(device):
        check(_lib.lib().o2345_preload(), "preload")
        if config.warm_aten():
            _warm_aten(device)
    _preloaded.add(device)


def _warm_aten(device):
    """The handful of ATen kernels that the UNCHANGED trainer (and the mirrors' glue) launch inside the reference's timing brackets, each once on a tiny tensor:
    the first launch of an ATen kernel in a process loads its code object (a few ms each: the valid-ray rule's four kernels cost 20 ms in the first
    rendering_network call of a fresh process without this, tools/dropin_bench.py --cold with O2345_WARM_ATEN=0).  Same operators and dtypes as the call sites:
    trainer_generic.py:1119-1123 (bilinear x4 / x2 up-sampling + cat of the pyramid), :1322 (float64 vertices -> float32 on the device),
    rendering_network.py:122-129 (valid-ray rule), generate_grids.py:4-19 (voxel lattice), the per-chunk normal map of val_step (:528-543)."""
    F = torch.nn.functional
    z = torch.zeros(1, 8, 4, 4, device=device)
    torch.cat([F.interpolate(z, scale_factor=4, mode="bilinear", align_corners=True), F.interpolate(z.repeat(1, 1, 2, 2), scale_factor=2, mode="bilinear", align_corners=True),
               z.repeat(1, 1, 4, 4)], dim=1)
    torch.zeros(1, 4, 2, 2, 2, device=device)[0].permute(1, 2, 3, 0).contiguous()
    nv = torch.zeros(2, 64, dtype=torch.uint8, device=device)
    ((nv >= 2).float().sum(1) > 8).cpu()
    torch.stack(torch.meshgrid(*[torch.arange(2, dtype=torch.float32, device=device)] * 3, indexing="ij"))[None]
    torch.zeros(4, 3, dtype=torch.float64).to(z)
    g, w = torch.zeros(2, 4, 3, device=device), torch.zeros(2, 4, device=device)
    (g * w[:, :4, None] * w[..., None]).sum(dim=1).detach().cpu()
    (torch.rand(2, 3, pin_memory=True).to(device, non_blocking=True) * 2 - 1).sum()
    k = torch.eye(3, device=device)[None].repeat(2, 1, 1).clone()              # trainer_generic.py:853-854: intrinsics.clone(); [:, :2] *= 0.25
    k[:, :2] *= 0.25


_ws_cache = {}


def _workspace(nbytes, device, tag="ws"):
    """Scratch buffer per (purpose, device, stream): two streams of one device never share scratch memory."""
    key = (tag, str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


# ---------------------------------------------------------------------------------------------------------- cost volume
@_on_device
def nchw_to_nhwc(x):
    V, C, H, W = x.shape
    out = torch.empty(V, H, W, C, device=x.device, dtype=torch.float32)
    check(_lib.lib().o2345_nchw_to_nhwc(_p(x), _p(out), V, C, H, W, _stream()), "nchw_to_nhwc")
    return out


def conv_x3(precision=None):
    """True when the convolutions run on the matrix cores (split-f16 form, the default numerical mode), False for the strict fp32 VALU kernels."""
    return config.color_precision(precision) == "f16x3"


@_on_device
def conv2d_pack(weight, precision=None):
    """nn.Conv2d weight [cout,cin,k,k] -> the operand packing of the convolution kernels for the numerical mode in force (one tiny launch; callers
    that own the parameter cache the result, see featurenet.packed_weight): split-f16 MFMA A operands [tap][cin/16][hi|lo][64 lanes][8 f16]
    or, in fp32 mode, [cin][k][k][cout] for the scalar-load VALU kernels."""
    cout, cin, k, k2 = weight.shape
    if k != k2:
        raise ValueError("conv2d: square kernels only")
    L = _lib.lib()
    if conv_x3(precision):
        t = torch.empty(L.o2345_conv2d_x3_weight_floats(cin, k), dtype=torch.float32, device=weight.device)
        check(L.o2345_conv2d_pack_weights_x3(_p(_f(weight)), cout, cin, k, _p(t), _stream()), "conv2d_pack_weights_x3")
    else:
        t = torch.empty(cin * k * k * cout, dtype=torch.float32, device=weight.device)
        check(L.o2345_conv2d_pack_weights(_p(_f(weight)), cout, cin, k, _p(t), _stream()), "conv2d_pack_weights")
    return t


@_on_device
def conv2d(x, weight, bias=None, stride=1, in_scale_shift=None, slope=0.01, bn=None, packed=None, precision=None, nhwc_offset=None):
    """nn.Conv2d (padding k // 2) of FeatureNet / the compress layer as a HIP kernel (csrc/convnet.hip): implicit GEMM on the matrix cores in the
    default mode, direct fp32 VALU convolution in fp32 mode.
    ``in_scale_shift`` [2*cin]: x is the RAW output of a convolution whose InPlaceABN (scale | shift, leaky ``slope``) is applied while x is read.
    ``bn`` = (gamma, beta, eps, abs_gamma): also reduce the batch statistics of the output and return this layer's own (scale | shift) [2*cout]
    for the consumer to apply.  ``packed``: conv2d_pack(weight, precision) kept by the caller.  ``nhwc_offset`` = c0: x is a channel-last map
    [V,Hi,Wi,C] and the convolution reads its channels c0 .. c0 + cin (the compress layer on the [V,H,W,64] colour map, c0 = 3).
    -> (raw output [V,cout,Ho,Wo], scale_shift or None)."""
    cout, cin, k, _ = weight.shape
    if nhwc_offset is None:
        V, cx, Hi, Wi = x.shape
        pix_stride, c0 = 0, 0
        if cx != cin:
            raise ValueError(f"conv2d: weight expects {cin} input channels, got {cx}")
    else:
        V, Hi, Wi, pix_stride = x.shape
        c0 = int(nhwc_offset)
        if c0 < 0 or c0 + cin > pix_stride:
            raise ValueError(f"conv2d: channels {c0}..{c0 + cin} do not fit a channel-last map with {pix_stride} channels")
    pad = k // 2
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    L = _lib.lib()
    out = torch.empty(V, cout, Ho, Wo, dtype=torch.float32, device=x.device)
    ss = gamma = beta = ws = None
    eps, abs_gamma, wsb = 0.0, 0, 0
    if bn is not None:
        gamma, beta, eps, abs_gamma = bn
        ss = torch.empty(2 * cout, dtype=torch.float32, device=x.device)
        wsb = L.o2345_conv2d_workspace_bytes(V, cout, Ho, Wo)
        ws = _workspace(wsb, x.device, "conv2d")
    fn = L.o2345_conv2d_x3 if conv_x3(precision) else L.o2345_conv2d
    check(fn(_p(x), V, cin, Hi, Wi, int(pix_stride), c0, _p(in_scale_shift), float(slope), _p(packed if packed is not None else conv2d_pack(weight, precision)),
             _p(None if bias is None else _f(bias)), cout, k, int(stride), _p(out), _p(None if gamma is None else _f(gamma)),
             _p(None if beta is None else _f(beta)), float(eps), int(abs_gamma), _p(ss), _p(ws, torch.uint8), wsb, _stream()), "conv2d")
    return out, ss


@_on_device
def scale_shift_act(x, scale_shift, slope=0.01, want_nchw=True, want_nhwc=False):
    """leaky_relu(x * scale + shift) of a raw convolution output [V,C,H,W] -> (NCHW or None, channel-last NHWC or None)."""
    V, C, H, W = x.shape
    y1 = torch.empty_like(x) if want_nchw else None
    y2 = torch.empty(V, H, W, C, dtype=torch.float32, device=x.device) if want_nhwc else None
    check(_lib.lib().o2345_scale_shift_act(_p(x), V, C, H, W, _p(scale_shift), float(slope), _p(y1), _p(y2), _stream()), "scale_shift_act")
    return y1, y2


@_on_device
def fpn_level(fine, coarse, weight, bias, fine_scale_shift=None, slope=0.01):
    """FeatureNet._upsample_add(coarse, lateral_1x1(fine)) in one kernel: fine [V,C,H,W] (C = 8 | 16), coarse [V,32,H/2,W/2] -> [V,32,H,W].
    ``fine_scale_shift``: fine is a raw convolution output, its InPlaceABN is applied on load."""
    V, C, H, W = fine.shape
    if tuple(coarse.shape) != (V, 32, H // 2, W // 2):
        raise ValueError(f"fpn_level: coarse map must be [V,32,H/2,W/2], got {tuple(coarse.shape)} for fine {tuple(fine.shape)}")
    out = torch.empty(V, 32, H, W, dtype=torch.float32, device=fine.device)
    check(_lib.lib().o2345_fpn_level_act(_p(fine), _p(fine_scale_shift), float(slope), C, _p(coarse), _p(_f(weight).reshape(32, C)), _p(_f(bias)), V, H, W,
                                         _p(out), _stream()), "fpn_level")
    return out


@_on_device
def pyramid_pack(f2, s1, s0, rgb, want_nchw=True):
    """Fused pyramid -> (fmaps [V,56,H,W] or None, cmaps [V,H,W,64] = rgb | 56 features | pad)."""
    V, _, H, W = s0.shape
    if tuple(f2.shape) != (V, 32, H // 4, W // 4) or tuple(s1.shape) != (V, 16, H // 2, W // 2) or tuple(rgb.shape) != (V, 3, H, W) or s0.shape[1] != 8:
        raise ValueError("pyramid_pack: expected f2 [V,32,H/4,W/4], s1 [V,16,H/2,W/2], s0 [V,8,H,W], rgb [V,3,H,W]")
    fm = torch.empty(V, 56, H, W, dtype=torch.float32, device=s0.device) if want_nchw else None
    cm = torch.empty(V, H, W, 64, dtype=torch.float32, device=s0.device)
    check(_lib.lib().o2345_pyramid_pack(_p(f2), _p(s1), _p(s0), _p(rgb), V, H, W, _p(fm), _p(cm), _stream()), "pyramid_pack")
    return fm, cm


@_on_device
def costvol_index(proj, V, H, W, dims, voxel_size, origin, min_views=1):
    """-> cnt u8 [D^3], row_of_voxel i32 [D^3], coords i32 [N,4] (x,y,z,b), N (python int; one 4-byte D2H read)."""
    L = _lib.lib()
    dx, dy, dz = (int(d) for d in dims)
    nvox = dx * dy * dz
    dev = proj.device
    cnt = torch.empty(nvox, dtype=torch.uint8, device=dev)
    row = torch.empty(nvox, dtype=torch.int32, device=dev)
    coords = torch.empty(nvox, 4, dtype=torch.int32, device=dev)
    n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = L.o2345_costvol_workspace_bytes(dx, dy, dz)
    ws = _workspace(wsb, dev)
    oh, ohp = _host3(origin)
    check(L.o2345_costvol_index(_p(proj), V, H, W, dx, dy, dz, float(voxel_size), ohp, int(min_views), _p(cnt, torch.uint8),
                                _p(row, torch.int32), _p(coords, torch.int32), _p(n_dev, torch.int32), _p(ws, torch.uint8),
                                wsb, _stream()), "costvol_index")
    n = int(n_dev.item())
    return cnt, row, coords[:n], n


@_on_device
def costvol_gather(feats_nhwc, proj, dims, voxel_size, origin, cnt, coords):
    V, H, W, C = feats_nhwc.shape
    n = coords.shape[0]
    out = torch.empty(n, 2 * C, dtype=torch.float32, device=feats_nhwc.device)
    if n == 0:
        return out
    oh, ohp = _host3(origin)
    dx, dy, dz = (int(d) for d in dims)
    check(_lib.lib().o2345_costvol_gather(_p(feats_nhwc), _p(proj), V, H, W, C, dx, dy, dz, float(voxel_size), ohp,
                                          _p(cnt, torch.uint8), _p(coords, torch.int32), n, _p(out), _stream()), "costvol_gather")
    return out


@_on_device
def visible_count_list(proj, H, W, voxel_size, origin, coords):
    """coords [M,4] int32 (x,y,z,b), any order -> number of views that see each listed voxel (u8 [M])."""
    M = coords.shape[0]
    cnt = torch.empty(M, dtype=torch.uint8, device=coords.device)
    if M == 0:
        return cnt
    oh, ohp = _host3(origin)
    check(_lib.lib().o2345_visible_count_list(_p(proj), proj.shape[0], H, W, float(voxel_size), ohp, _p(coords, torch.int32), M,
                                             _p(cnt, torch.uint8), _stream()), "visible_count_list")
    return cnt


@_on_device
def costvol_gather_list(feats_nhwc, proj, voxel_size, origin, cnt_row, coords):
    V, H, W, C = feats_nhwc.shape
    n = coords.shape[0]
    out = torch.empty(n, 2 * C, dtype=torch.float32, device=feats_nhwc.device)
    if n == 0:
        return out
    oh, ohp = _host3(origin)
    check(_lib.lib().o2345_costvol_gather_list(_p(feats_nhwc), _p(proj), V, H, W, C, float(voxel_size), ohp, _p(cnt_row, torch.uint8),
                                              _p(coords, torch.int32), n, _p(out), _stream()), "costvol_gather_list")
    return out


@_on_device
def build_index_grid(coords, ts, cells):
    nx, ny, nz = (int(c) for c in cells)
    grid = torch.empty(nx * ny * nz, dtype=torch.int32, device=coords.device)
    check(_lib.lib().o2345_build_index_grid(_p(coords, torch.int32) if coords.shape[0] else None, coords.shape[0], int(ts), nx, ny, nz,
                                           _p(grid, torch.int32), _stream()), "build_index_grid")
    return grid


@_on_device
def prune_dilate(sdf_vol, mask_vol, D, threshold, radius=3):
    out = torch.empty(D * D * D, dtype=torch.uint8, device=sdf_vol.device)
    check(_lib.lib().o2345_prune_dilate(_p(sdf_vol), _p(mask_vol), D, float(threshold), int(radius), _p(out, torch.uint8), _stream()), "prune_dilate")
    return out


@_on_device
def scatter_dense(rows, row_of_voxel, dims, want_cf=True):
    dx, dy, dz = (int(d) for d in dims)
    nvox, C = dx * dy * dz, rows.shape[1]
    dev = rows.device
    if rows.shape[0] == 0:           # nothing kept: all-zero volumes (the kernel needs a non-null rows pointer)
        z = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=dev)
        return z(dx, dy, dz, C), (z(1, C, dx, dy, dz) if want_cf else None), z(1, 1, dx, dy, dz)
    cl = torch.empty(dx, dy, dz, C, dtype=torch.float32, device=dev)
    cf = torch.empty(1, C, dx, dy, dz, dtype=torch.float32, device=dev) if want_cf else None
    mask = torch.empty(1, 1, dx, dy, dz, dtype=torch.float32, device=dev)
    check(_lib.lib().o2345_scatter_dense(_p(rows), _p(row_of_voxel, torch.int32), C, nvox, _p(cl), _p(cf), _p(mask), _stream()), "scatter_dense")
    return cl, cf, mask


# ---------------------------------------------------------------------------------------------------------- sparse CNN
@_on_device
def sparse_downsample(coords, ts, fine_cells):
    """coords [n,4] int32 at tensor stride ts on a lattice of fine_cells cells/axis -> (grid, coords_c, n_c, cells_c)."""
    L = _lib.lib()
    nc = tuple((int(c) + 1) // 2 + 1 for c in fine_cells)
    dev = coords.device
    if coords.shape[0] == 0:
        ncell = nc[0] * nc[1] * nc[2]
        return torch.full((ncell,), -1, dtype=torch.int32, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), 0, nc
    ncell = nc[0] * nc[1] * nc[2]
    grid = torch.empty(ncell, dtype=torch.int32, device=dev)
    cc = torch.empty(ncell, 4, dtype=torch.int32, device=dev)
    n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = L.o2345_sparse_downsample_workspace_bytes(*nc)
    ws = _workspace(wsb, dev)
    check(L.o2345_sparse_downsample(_p(coords, torch.int32), coords.shape[0], int(ts), nc[0], nc[1], nc[2],
                                    _p(grid, torch.int32), _p(cc, torch.int32), _p(n_dev, torch.int32), _p(ws, torch.uint8),
                                    wsb, _stream()), "sparse_downsample")
    n = int(n_dev.item())
    return grid, cc[:n], n, nc


@_on_device
def sparse_conv3d(mode, x, in_grid, in_cells, out_coords, ts_out, kernel):
    n_out, cin, cout = out_coords.shape[0], x.shape[1], kernel.shape[2]
    out = torch.empty(n_out, cout, dtype=torch.float32, device=x.device)
    if n_out == 0 or x.shape[0] == 0:
        return out.zero_()
    check(_lib.lib().o2345_sparse_conv3d(int(mode), _p(x), cin, _p(in_grid, torch.int32), in_cells[0], in_cells[1], in_cells[2],
                                         _p(out_coords, torch.int32), n_out, int(ts_out), _p(kernel), cout, _p(out), _stream()), "sparse_conv3d")
    return out


@_on_device
def sparse_conv3d_x3(mode, x, in_grid, in_cells, out_coords, ts_out, wblob, cout, identity_rows=False):
    """Matrix-core form of sparse_conv3d (csrc/sparse_mfma.hip); wblob = weights.pack_sparse_conv_x3(kernel) on the device.
    identity_rows=True (mode 0): the caller guarantees out_coords IS the coordinate list in_grid was built from, in the same order (a stride-1
    spnn.Conv3d) -- the precondition of the LDS-tiled brick kernel (32 -> 16 channels), which never reads out_coords.  Default: the gather form,
    correct for any subset / order of out_coords."""
    n_out, cin = out_coords.shape[0], x.shape[1]
    out = torch.empty(n_out, cout, dtype=torch.float32, device=x.device)
    if n_out == 0 or x.shape[0] == 0:
        return out.zero_()
    if identity_rows and (int(mode) != 0 or n_out != x.shape[0]):
        raise ValueError(f"sparse_conv3d_x3: identity_rows needs mode 0 and one output row per input row (mode {mode}, {n_out} vs {x.shape[0]} rows)")
    check(_lib.lib().o2345_sparse_conv3d_x3(int(mode), _p(x), cin, _p(in_grid, torch.int32), in_cells[0], in_cells[1], in_cells[2],
                                            _p(out_coords, torch.int32), n_out, int(ts_out), _p(wblob), int(cout), int(bool(identity_rows)), _p(out),
                                            _stream()), "sparse_conv3d_x3")
    return out


@_on_device
def bn_act_rows(x, gamma, beta, eps=1e-5, slope=0.0, abs_gamma=False, skip=None, want_stats=False):
    L = _lib.lib()
    n, C = x.shape
    y = torch.empty_like(x)
    if n == 0:
        return (y, torch.zeros(2, C, device=x.device)) if want_stats else y
    wsb = L.o2345_bn_workspace_bytes(C)
    ws = _workspace(wsb, x.device, "bn")
    mv = torch.empty(2, C, dtype=torch.float32, device=x.device) if want_stats else None
    check(L.o2345_bn_act_rows(_p(x), n, C, _p(gamma), _p(beta), float(eps), float(slope), int(abs_gamma), _p(skip), _p(y),
                              _p(mv), _p(ws, torch.uint8), wsb, _stream()), "bn_act_rows")
    return (y, mv) if want_stats else y


@_on_device
def abn_nchw(x, gamma, beta, eps=1e-5, slope=0.01, abs_gamma=True, want_nchw=True, want_nhwc=False):
    L = _lib.lib()
    V, C, H, W = x.shape
    wsb = L.o2345_abn_workspace_bytes(C)
    ws = _workspace(wsb, x.device, "abn")
    y1 = torch.empty_like(x) if want_nchw else None
    y2 = torch.empty(V, H, W, C, dtype=torch.float32, device=x.device) if want_nhwc else None
    check(L.o2345_abn_nchw(_p(x), V, C, H, W, _p(gamma), _p(beta), float(eps), float(slope), int(abs_gamma), _p(y1), _p(y2),
                           _p(ws, torch.uint8), wsb, _stream()), "abn_nchw")
    return y1, y2


# ---------------------------------------------------------------------------------------------------------- SDF network
@_on_device
def sdf_grid_tables(tab_axes, bias_lane_order):
    """(tab_axes [3,R,128], bias [128]) from weights.sdf_grid_tables, on the device -> (tab_xy [R*R,128] = x + y + bias, tab_z [R,128])."""
    R = tab_axes.shape[1]
    tab_xy = torch.empty(R * R, 128, dtype=torch.float32, device=tab_axes.device)
    check(_lib.lib().o2345_sdf_grid_tables(_p(tab_axes), _p(bias_lane_order), R, _p(tab_xy), _stream()), "sdf_grid_tables")
    return tab_xy, tab_axes[2]


@_on_device
def sdf_mlp(blob, vol_cl, pts=None, variant=0, grid_R=0, sign=1.0, index=None, n_dev=None, want_lat=False, out=None, lat_in=None,
            precision=None, grid_tables=None):
    """variant 0: sdf; 1: sdf + 128 features; 2: sdf + gradient.  pts [P,3] or grid_R.  lat_in [P,16]: given latents instead of
    sampling the volume (get_sdf_volume).  precision "f16x3" (default): split-f16 MFMA at fp32-class accuracy (variants 0 and 2; variant 1 / lat_in
    run on the fp32 kernel); "fp32": the exact fp32 MFMA kernels.  grid_tables (f16x3, variant 0, lattice mode): (tab_xy, tab_z) from
    ops.sdf_grid_tables -- layer 0 read from per-axis tables instead of being evaluated per point.  Returns dict of tensors."""
    precision = config.sdf_precision(precision)
    D = vol_cl.shape[0]
    dev = vol_cl.device
    if pts is not None:
        P = pts.shape[0]
    else:
        P = int(grid_R) ** 3
    n = P if index is None else index.shape[0]
    res = out or {}
    if "sdf" not in res:
        res["sdf"] = torch.empty(P, dtype=torch.float32, device=dev)
    if variant == 1 and "feat" not in res:
        res["feat"] = torch.empty(P, 128, dtype=torch.float32, device=dev)
    if variant == 2 and "grad" not in res:
        res["grad"] = torch.empty(P, 3, dtype=torch.float32, device=dev)
    if want_lat and "lat" not in res:
        res["lat"] = torch.empty(P, 16, dtype=torch.float32, device=dev)
    if P == 0 or (n == 0 and n_dev is None):
        return res
    if precision == "f16x3" and variant == 0 and not want_lat and lat_in is None:
        if grid_tables is not None and pts is None and index is None and n_dev is None:
            tab_xy, tab_z = grid_tables                  # lattice mode with layer 0 tabulated (weights.sdf_grid_tables + ops.sdf_grid_tables)
            if tuple(tab_xy.shape) != (int(grid_R) ** 2, 128) or tuple(tab_z.shape) != (int(grid_R), 128):
                raise ValueError(f"sdf_mlp: grid tables were built for another resolution than {grid_R}")
            check(_lib.lib().o2345_sdf_grid_x3(_p(blob), _p(vol_cl), D, int(grid_R), float(sign), _p(tab_xy), _p(tab_z), _p(res["sdf"]), _stream()),
                  "sdf_grid_x3")
            return res
        check(_lib.lib().o2345_sdf_mlp_x3(_p(blob), _p(vol_cl), D, _p(pts), _p(index, torch.int32), _p(n_dev, torch.int32), n, int(grid_R),
                                          float(sign), _p(res["sdf"]), _stream()), "sdf_mlp_x3")
        return res
    if precision == "f16x3" and variant == 2 and not want_lat and lat_in is None:
        check(_lib.lib().o2345_sdf_grad_x3(_p(blob), _p(vol_cl), D, _p(pts), _p(index, torch.int32), _p(n_dev, torch.int32), n, int(grid_R),
                                           float(sign), _p(res["sdf"]), _p(res["grad"]), _stream()), "sdf_grad_x3")
        return res
    check(_lib.lib().o2345_sdf_mlp_ex(int(variant), _p(blob), _p(vol_cl), D, _p(pts), _p(index, torch.int32), _p(n_dev, torch.int32),
                                      n, int(grid_R), float(sign), _p(lat_in), _p(res["sdf"]), _p(res.get("feat")), _p(res.get("lat")),
                                      _p(res.get("grad")), _stream()), "sdf_mlp")
    return res


# ---------------------------------------------------------------------------------------------------------- colour
@_on_device
def pack_color_maps(feat_nchw, color_nchw):
    V, _, H, W = feat_nchw.shape
    out = torch.empty(V, H, W, 64, dtype=torch.float32, device=feat_nchw.device)
    check(_lib.lib().o2345_pack_color_maps(_p(feat_nchw), _p(color_nchw), V, H, W, _p(out), _stream()), "pack_color_maps")
    return out


@_on_device
def camera_terms(intrinsics, w2cs):
    """intrinsics [V,3,3], w2cs [V,4,4] -> (proj [V,3,4] = K @ w2c[:3] (render_utils.py:106), cam_pos [V,3] = inverse(w2c)[:3, 3]) in one launch --
    no BLAS / solver library (the first torch.matmul + torch.inverse of a process initialise rocBLAS and rocSOLVER: 180 ms)."""
    V = intrinsics.shape[0]
    if tuple(intrinsics.shape) != (V, 3, 3) or tuple(w2cs.shape) != (V, 4, 4):
        raise ValueError(f"camera_terms: intrinsics [V,3,3] and w2cs [V,4,4] expected, got {tuple(intrinsics.shape)}, {tuple(w2cs.shape)}")
    proj = torch.empty(V, 3, 4, dtype=torch.float32, device=w2cs.device)
    cam = torch.empty(V, 3, dtype=torch.float32, device=w2cs.device)
    check(_lib.lib().o2345_camera_terms(_p(_f(intrinsics)), _p(_f(w2cs)), V, _p(proj), _p(cam), _stream()), "camera_terms")
    return proj, cam


def color_stats_buffer(device):
    """Zeroed work counters [4] (int64 on the device) for ``color_points(..., stats=buf)`` / ``render_rays(..., color_stats=buf)``: the launches ADD
    (32-point tile, view) pairs evaluated in the pooling pass / the network pass, tiles, tiles that evaluated every view.  Caller-owned: no library state."""
    return torch.zeros(4, dtype=torch.int64, device=device)


def color_stats_read(buf):
    out = [int(x) for x in buf.cpu().tolist()]
    return dict(pairs_pooling=out[0], pairs_network=out[1], tiles=out[2], tiles_all_views=out[3])


@_on_device
def color_points(blob, vol_cl, maskvol, cmaps, proj, cam_pos, pts, query_cam=None, normals=None, index=None, n_dev=None,
                 want_nviews=True, mfma="x3", stats=None):
    """Projector + GeneralRenderingNetwork fused (k_color_pts).  mfma="x3": split-f16 matrix steps (blob from weights.pack_color_x3_blob, the default
    numerical mode); mfma=True: fp32 matrix-core form (pack_color_mfma_blob).  Any view count up to 255.  stats: color_stats_buffer()."""
    if mfma is False or mfma is None:
        raise ValueError("color_points: the pure-VALU colour kernel was removed in ABI 2.0; pass mfma='x3' (split-f16) or mfma=True (fp32 MFMA)")
    V, H, W, _ = cmaps.shape
    P = pts.shape[0]
    n = P if index is None else index.shape[0]
    rgb = torch.zeros(P, 3, dtype=torch.float32, device=pts.device)
    nv = torch.zeros(P, dtype=torch.uint8, device=pts.device) if want_nviews else None
    if P == 0 or (n == 0 and n_dev is None):
        return rgb, nv
    L = _lib.lib()
    fn = L.o2345_color_points_x3 if mfma == "x3" else L.o2345_color_points_mfma
    check(fn(_p(blob), _p(vol_cl), _p(maskvol), vol_cl.shape[0], _p(cmaps), _p(proj), _p(cam_pos),
             V, H, W, _p(pts), _p(index, torch.int32), _p(n_dev, torch.int32), n, _p(query_cam),
             _p(normals), _p(rgb), _p(nv, torch.uint8), _p(stats, torch.int64), _stream()), "color_points")
    return rgb, nv


@_on_device
def project_features(vol_cl, maskvol, cmaps, proj, cam_pos, pts, query_cam=None, normals=None):
    """Projector.compute (query_cam) / compute_view_independent (normals) materialised in the reference's layout:
    -> geometry_feat [P,16], rgb_feat [V,P,59], ray_diff [V,P,4], mask [V,P] (1 / 0)."""
    V, H, W, _ = cmaps.shape
    P = pts.shape[0]
    dev = pts.device
    geo = torch.empty(P, 16, dtype=torch.float32, device=dev)
    rf = torch.empty(V, P, 59, dtype=torch.float32, device=dev)
    rd = torch.empty(V, P, 4, dtype=torch.float32, device=dev)
    m = torch.empty(V, P, dtype=torch.float32, device=dev)
    check(_lib.lib().o2345_project_features(_p(vol_cl), _p(maskvol), vol_cl.shape[0], _p(cmaps), _p(proj), _p(cam_pos), V, H, W, _p(pts), P, _p(query_cam),
                                            _p(normals), _p(geo), _p(rf), _p(rd), _p(m), _stream()), "project_features")
    return geo, rf, rd, m


@_on_device
def color_from_features(blob, geometry_feat, rgb_feat, ray_diff, mask, x3=True, want_nviews=True):
    """GeneralRenderingNetwork.forward on materialised tensors in the reference's layout: geometry_feat [P,16], rgb_feat [V,P,59],
    ray_diff [V,P,4], mask [V,P] -> (rgb [P,3], valid views uint8 [P]).  blob: pack_color_x3_blob (x3) or pack_color_mfma_blob."""
    V, P, C = rgb_feat.shape
    if C != 59 or tuple(geometry_feat.shape) != (P, 16) or tuple(ray_diff.shape) != (V, P, 4) or tuple(mask.shape) != (V, P):
        raise ValueError(f"color_from_features: expected geometry_feat [P,16], rgb_feat [V,P,59], ray_diff [V,P,4], mask [V,P]; got "
                         f"{tuple(geometry_feat.shape)}, {tuple(rgb_feat.shape)}, {tuple(ray_diff.shape)}, {tuple(mask.shape)}")
    rgb = torch.empty(P, 3, dtype=torch.float32, device=rgb_feat.device)
    nv = torch.empty(P, dtype=torch.uint8, device=rgb_feat.device) if want_nviews else None
    if P:
        check(_lib.lib().o2345_color_from_features(_p(blob), int(bool(x3)), _p(_f(geometry_feat)), _p(_f(rgb_feat)), _p(_f(ray_diff)), _p(_f(mask)), V, P,
                                                   _p(rgb), _p(nv, torch.uint8), _stream()), "color_from_features")
    return rgb, nv


@_on_device
def view_count(pts, maskvol, D, proj, V, H, W):
    out = torch.empty(pts.shape[0], dtype=torch.uint8, device=pts.device)
    check(_lib.lib().o2345_view_count(_p(pts), pts.shape[0], _p(maskvol), D, _p(proj), V, H, W, _p(out, torch.uint8), _stream()), "view_count")
    return out


@_on_device
def list_sort_by_visibility(pts, index, proj, H, W, count=None, want_keys=False):
    """index [n] int32 slots into pts [P,3] -> the same entries grouped, stably, by view-visibility signature (what o2345_render_rays does to its
    occupied-point list before the network kernels; csrc/list_sort.hip).  count (optional int32[1] on the device): number of valid entries."""
    L = _lib.lib()
    n = int(index.shape[0])
    V = int(proj.shape[0])
    out = torch.empty_like(index)
    if n == 0:
        return (out, torch.empty(0, dtype=torch.int32, device=index.device)) if want_keys else out
    if count is None:
        count = torch.tensor([n], dtype=torch.int32, device=index.device)
    keys = torch.empty(n, dtype=torch.int32, device=index.device) if want_keys else None
    wsb = L.o2345_list_sort_workspace_bytes(n, V)
    ws = _workspace(wsb, index.device, "lsort")
    check(L.o2345_list_sort_by_visibility(_p(pts), _p(index, torch.int32), _p(count, torch.int32), n, _p(proj), V, int(H), int(W), _p(out, torch.int32),
                                          _p(keys, torch.int32), _p(ws, torch.uint8), wsb, _stream()), "list_sort_by_visibility")
    return (out, keys) if want_keys else out


# ---------------------------------------------------------------------------------------------------------- rays
_SCENE_KEYS = ("sdf_blob", "vol_cl", "maskvol", "cmaps", "proj", "cam_pos")


@_on_device
def render_rays(scene, rays_o, rays_d, near, far, n_samples=64, n_importance=64, inv_s=None, alpha_inter_ratio=1.0,
                background=1.0, query_cam=None, want_z=False, t_rand=None, sample_dist=None, want_scalars=False, color_stats=None,
                weight_cull=None, segment_rays=0):
    """scene: dict(sdf_blob, color_x3_blob and / or color_mfma_blob, vol_cl, maskvol [D^3], cmaps, proj [V,3,4], cam_pos [V,3]).
    near / far: python floats, or two float32 tensors [R] on the device (the reference's per-ray [N_rays, 1] form; then ``sample_dist`` =
    ((far - near) / n_samples).mean() must be given, sparse_neus_renderer.py:484).
    t_rand [R, n_samples] (optional): the reference's stratified jitter of the coarse samples (perturb > 0, :506-515), drawn by the caller.
    want_scalars: also ``scalars`` [4] = (alpha_sum.mean(), alpha_sum.sum() / (R S), gradient error, evaluated points).  color_stats: color_stats_buffer().
    weight_cull (None: config.weight_cull(), default 2^-24; 0: exhaustive like the reference): the colour network is evaluated only on occupied samples whose
    compositing weight is >= weight_cull -- a ray's colour moves by <= (n_samples + n_importance) * weight_cull (include/o2345.h, O2345RenderIO.weight_cull).
    segment_rays (0: one render() call): the rays are consecutive render() calls of that many rays evaluated together (O2345RenderIO.segment_rays).
    Returns dict of SAMPLE-MAJOR tensors ([S,R,...]) + per-ray results."""
    L = _lib.lib()
    R = rays_o.shape[0]
    if query_cam is None:
        raise ValueError("render_rays: query_cam [3] (the query camera centre, query_c2w[:3, 3]) is required")
    if inv_s is None:
        raise ValueError("render_rays: inv_s is required")
    if rays_d.shape != rays_o.shape or rays_o.dim() != 2 or rays_o.shape[1] != 3:
        raise ValueError(f"render_rays: rays_o / rays_d must both be [R,3] (got {tuple(rays_o.shape)}, {tuple(rays_d.shape)})")
    if (n_samples + n_importance) * R >= 2 ** 31:
        raise ValueError("render_rays: R * (n_samples + n_importance) must stay below 2^31; split the ray batch")
    dev = rays_o.device
    for k in _SCENE_KEYS:
        _p(scene[k])                                   # contiguous cuda float32, or ValueError
        if scene[k].device != dev:
            raise ValueError(f"render_rays: scene[{k!r}] is on {scene[k].device}, the rays on {dev}")
    Dv = scene["vol_cl"].shape[0]
    cm = scene["cmaps"]
    if scene["vol_cl"].dim() != 4 or scene["maskvol"].numel() != Dv ** 3 or cm.dim() != 4 or cm.shape[-1] != 64:
        raise ValueError("render_rays: vol_cl [D,D,D,C], maskvol [D^3], cmaps [V,H,W,64] expected")
    V, H, W, _ = cm.shape
    if tuple(scene["proj"].shape) != (V, 3, 4) or tuple(scene["cam_pos"].shape) != (V, 3):
        raise ValueError("render_rays: proj [V,3,4] and cam_pos [V,3] must match the V of cmaps")
    if t_rand is not None and tuple(t_rand.shape) != (R, n_samples):
        raise ValueError(f"render_rays: t_rand must be [R, n_samples] = {(R, n_samples)}, got {tuple(t_rand.shape)}")
    S = n_samples + n_importance
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    o = dict(mid_z=f(S, R), dists=f(S, R), pm=f(S, R), sdf=f(S, R), grad=f(S, R, 3), rgb=f(S, R, 3),
             nviews=torch.empty(S, R, dtype=torch.uint8, device=dev), color=f(R, 3), depth=f(R), weights=f(S, R), cdf=f(S, R),
             weights_sum=f(R), weights_max=f(R), depth_var=f(R), alpha_sum=f(R), grad_err=f(R, 2),
             color_mask=torch.empty(R, dtype=torch.uint8, device=dev))
    if want_z:
        o["z_vals"] = f(S, R)
    segment_rays = int(segment_rays)
    if segment_rays < 0 or segment_rays % 64:
        raise ValueError(f"render_rays: segment_rays must be 0 or a multiple of 64, got {segment_rays}")
    n_seg = (R + segment_rays - 1) // segment_rays if segment_rays else 0
    if want_scalars:
        o["scalars"] = f(n_seg, 4) if segment_rays else f(4)
    io = _lib.RenderIO()
    for k in _SCENE_KEYS:
        setattr(io, k, scene[k].data_ptr())
    xb, mb = scene.get("color_x3_blob"), scene.get("color_mfma_blob")
    use_x3 = xb is not None and config.color_precision(scene.get("color_precision")) == "f16x3"
    if not use_x3 and mb is None:
        raise ValueError("render_rays: scene needs color_x3_blob (f16x3 mode) or color_mfma_blob (fp32 mode)")
    io.color_x3_blob = _p(xb).value if use_x3 else None
    io.color_mfma_blob = _p(mb).value if mb is not None else None
    io.t_rand = _p(t_rand).value if t_rand is not None else None
    io.sdf_mode = 2 if config.sdf_precision(scene.get("sdf_precision")) == "f16x3" else 0
    io.D, io.V, io.H, io.W = Dv, V, H, W
    io.rays_o, io.rays_d, io.R = _p(rays_o).value, _p(rays_d).value, R
    if torch.is_tensor(near) or torch.is_tensor(far):
        if not (torch.is_tensor(near) and torch.is_tensor(far)) or near.numel() != R or far.numel() != R or sample_dist is None:
            raise ValueError("render_rays: per-ray near / far are two tensors of R elements, and sample_dist must be given")
        io.near_ray, io.far_ray = _p(near).value, _p(far).value
        io.near = io.far = 0.0
    else:
        io.near_ray = io.far_ray = None
        io.near, io.far = float(near), float(far)
    io.sample_dist = float(sample_dist) if sample_dist is not None else 0.0
    io.n_samples, io.n_importance = n_samples, n_importance
    io.inv_s, io.alpha_inter_ratio, io.background = float(inv_s), float(alpha_inter_ratio), float(background)
    io.weight_cull = float(config.weight_cull(scene.get("weight_cull") if weight_cull is None else weight_cull))
    io.segment_rays = int(segment_rays)
    io.query_cam = _p(query_cam).value
    for k, t in o.items():
        setattr(io, k, t.data_ptr())
    if not want_z:
        io.z_vals = None
    if not want_scalars:
        io.scalars = None
    io.color_stats = _p(color_stats, torch.int64).value if color_stats is not None else None
    wsb = L.o2345_render_workspace_bytes(R, n_samples, n_importance, V)
    ws = _workspace(wsb, dev, "render")
    check(L.o2345_render_rays(ctypes.byref(io), _p(ws, torch.uint8), wsb, _stream()), "render_rays")
    return o


@_on_device
def ray_upsample(rays_o, rays_d, z, sdf, inv_s, maskvol, D, n_imp, streaming=None):
    """Stage entry point of the hierarchical sampler (up_sample + sample_pdf, sparse_neus_renderer.py:73-115, render_utils.py:8-51):
    z, sdf SAMPLE-MAJOR [S,R] -> (new_z [n_imp,R], new_pts [n_imp,R,3], valid-point list int32 [<= n_imp*R] of slots t*R + r)."""
    S, R = z.shape
    dev = z.device
    # streaming=None: scratch is handed over and the library picks the kernel by R (O2345_RAY_STREAM_MIN); False: no scratch -> the LDS-staged kernel
    wbuf = None if streaming is False else torch.empty(S, R, dtype=torch.float32, device=dev)
    new_z = torch.empty(n_imp, R, dtype=torch.float32, device=dev)
    new_pts = torch.empty(n_imp, R, 3, dtype=torch.float32, device=dev)
    new_sdf = torch.empty(n_imp, R, dtype=torch.float32, device=dev)
    lst = torch.empty(n_imp * R, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    check(_lib.lib().o2345_ray_upsample(_p(rays_o), _p(rays_d), R, _p(z), _p(sdf), S, float(inv_s), _p(maskvol), int(D), _p(wbuf), int(n_imp),
                                        _p(new_z), _p(new_pts), _p(new_sdf), _p(lst, torch.int32), _p(cnt, torch.int32), _stream()), "ray_upsample")
    return new_z, new_pts, lst[:int(cnt.item())]


@_on_device
def ray_finalize(rays_o, rays_d, z, sample_dist, maskvol, D):
    """Stage entry point of render_core's head (sparse_neus_renderer.py:204-231): z SAMPLE-MAJOR [S,R] -> dict(mid_z, dists, pm [S,R], pts [S,R,3],
    sdf [S,R] = 100, grad / rgb [S,R,3] = 0 (the reference's defaults in EVERY slot), list int32 (occupied slots s*R + r, wave-major), count)."""
    S, R = z.shape
    dev = z.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    o = dict(mid_z=f(S, R), dists=f(S, R), pts=f(S, R, 3), pm=f(S, R), sdf=f(S, R), grad=f(S, R, 3), rgb=f(S, R, 3),
             list=torch.empty(S * R, dtype=torch.int32, device=dev), count=torch.zeros(1, dtype=torch.int32, device=dev))
    check(_lib.lib().o2345_ray_finalize(_p(rays_o), _p(rays_d), R, _p(z), S, float(sample_dist), _p(maskvol), int(D), _p(o["mid_z"]), _p(o["dists"]),
                                        _p(o["pts"]), _p(o["pm"]), _p(o["sdf"]), _p(o["grad"]), _p(o["rgb"]), _p(o["list"], torch.int32),
                                        _p(o["count"], torch.int32), _stream()), "ray_finalize")
    return o


@_on_device
def ray_composite(rays_o, rays_d, mid_z, dists, pm, sdf, grad, rgb, nviews, inv_s, alpha_inter_ratio=1.0, background=1.0):
    """Stage entry point of render_core's tail (sparse_neus_renderer.py:340-455): NeuS opacities from (sdf, gradient, section length), transmittance,
    weights, colour / depth / depth variance, the per-ray colour mask (> 8 samples seen by >= 2 views).  Per-sample inputs SAMPLE-MAJOR [S,R(,3)],
    nviews uint8 [S,R].  -> dict(color [R,3], depth, weights_sum, weights_max, depth_var, alpha_sum [R], grad_err [R,2], color_mask uint8 [R],
    weights / cdf [S,R])."""
    S, R = mid_z.shape
    dev = mid_z.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    o = dict(color=f(R, 3), depth=f(R), weights=f(S, R), cdf=f(S, R), weights_sum=f(R), weights_max=f(R), depth_var=f(R), alpha_sum=f(R),
             grad_err=f(R, 2), color_mask=torch.empty(R, dtype=torch.uint8, device=dev))
    check(_lib.lib().o2345_ray_composite(_p(rays_o), _p(rays_d), R, S, _p(mid_z), _p(dists), _p(pm), _p(sdf), _p(grad), _p(rgb), _p(nviews, torch.uint8),
                                         float(inv_s), float(alpha_inter_ratio), float(background), _p(o["color"]), _p(o["depth"]), _p(o["weights"]),
                                         _p(o["cdf"]), _p(o["weights_sum"]), _p(o["weights_max"]), _p(o["depth_var"]), _p(o["alpha_sum"]),
                                         _p(o["grad_err"]), _p(o["color_mask"], torch.uint8), _stream()), "ray_composite")
    return o


@_on_device
def render_core(scene, rays_o, rays_d, z, sample_dist, inv_s, alpha_inter_ratio=1.0, background=1.0, query_cam=None):
    """The reference's render_core (sparse_neus_renderer.py:171-455) on GIVEN sample depths ``z`` SAMPLE-MAJOR [S,R] -- everything of a render() call that
    lies downstream of the hierarchical sampler, composed from the public stage entries in the order o2345_render_rays runs them: o2345_ray_finalize
    (mid points, section lengths, occupancy, defaults, occupied-point list; the "first 100 points" rule of :222-223 when no point is occupied) ->
    SDF + analytic gradient on the listed points -> valid-view counts of the unlisted points -> Projector + GeneralRenderingNetwork on the listed points
    -> o2345_ray_composite.  scene: the dict of render_rays.  -> dict: render_rays' per-sample / per-ray keys (+ ``list``)."""
    S, R = z.shape
    D = scene["vol_cl"].shape[0]
    V, H, W, _ = scene["cmaps"].shape
    o = ray_finalize(rays_o, rays_d, z, sample_dist, scene["maskvol"], D)
    n = int(o["count"])                                        # (a stage composition on the host: one read-back; o2345_render_rays keeps the count on the device)
    if n < 1:                                                  # :222-223 -- the first 100 points in the reference's ray-major order: point t = (ray t // S, sample t % S)
        n = min(100, S * R)
        t = torch.arange(n, dtype=torch.int32, device=z.device)
        o["list"][:n] = (t % S) * R + t // S
    lst = o["list"][:n].contiguous()
    pts = o["pts"].view(-1, 3)
    sdf_precision = config.sdf_precision(scene.get("sdf_precision"))
    res = {"sdf": o["sdf"].view(-1), "grad": o["grad"].view(-1, 3)}
    sdf_mlp(scene["sdf_blob"], scene["vol_cl"], pts, variant=2, index=lst, out=res, precision=sdf_precision)
    nviews = torch.zeros(S * R, dtype=torch.uint8, device=z.device)
    check(_lib.lib().o2345_view_count_unlisted(_p(pts), S * R, _p(o["pm"]), _p(scene["maskvol"]), D, _p(scene["proj"]), V, H, W,
                                               _p(nviews, torch.uint8), _stream()), "view_count_unlisted")
    xb, mb = scene.get("color_x3_blob"), scene.get("color_mfma_blob")
    use_x3 = xb is not None and config.color_precision(scene.get("color_precision")) == "f16x3"
    L = _lib.lib()
    fn = L.o2345_color_points_x3 if use_x3 else L.o2345_color_points_mfma
    check(fn(_p(xb if use_x3 else mb), _p(scene["vol_cl"]), _p(scene["maskvol"]), D, _p(scene["cmaps"]), _p(scene["proj"]), _p(scene["cam_pos"]), V, H, W,
             _p(pts), _p(lst, torch.int32), None, n, _p(query_cam), None, _p(o["rgb"]), _p(nviews, torch.uint8), None, _stream()), "color_points")
    c = ray_composite(rays_o, rays_d, o["mid_z"], o["dists"], o["pm"], o["sdf"], o["grad"], o["rgb"], nviews.view(S, R), inv_s, alpha_inter_ratio, background)
    c.update(mid_z=o["mid_z"], dists=o["dists"], pm=o["pm"], sdf=o["sdf"], grad=o["grad"], rgb=o["rgb"], nviews=nviews.view(S, R), list=lst, z_vals=z)
    return c


# ---------------------------------------------------------------------------------------------------------- marching cubes
@_on_device
def mesh_pack(verts_idx, tris, grid_R, bound_min=(-1.0, -1.0, -1.0), bound_max=(1.0, 1.0, 1.0), scale_mat=None, trans_mat=None, rgb=None):
    """Index-space vertices (fp64 [N,3]) + triangles -> (vertex records uint8 [N,16|12], face records uint8 [M,13]) of a binary PLY;
    frame transforms and colour quantisation as in trainer_generic.py:1365-1377.  Matrices / bounds are small host arrays."""
    dev = verts_idx.device
    n, m = verts_idx.shape[0], tris.shape[0]
    host = lambda a: None if a is None else np.ascontiguousarray((a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)), np.float32)
    bmin, bmax = host(bound_min).reshape(3), host(bound_max).reshape(3)
    sm, tm = host(scale_mat), host(trans_mat)
    sm = None if sm is None else sm.reshape(-1, 4, 4)[0].copy()
    tm = None if tm is None else tm.reshape(-1, 4, 4)[0].copy()
    cp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    vrec = torch.empty(n, 16 if rgb is not None else 12, dtype=torch.uint8, device=dev)
    frec = torch.empty(m, 13, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    check(L.o2345_mesh_pack_vertices(_p(verts_idx, torch.float64), n, int(grid_R), cp(bmin), cp(bmax), cp(sm), cp(tm), _p(rgb),
                                     _p(vrec, torch.uint8), _stream()), "mesh_pack_vertices")
    check(L.o2345_mesh_pack_faces(_p(tris, tris.dtype), 8 if tris.dtype == torch.int64 else 4, m, _p(frec, torch.uint8), _stream()), "mesh_pack_faces")
    return vrec, frec


@_on_device
def mc_verts_to_world(verts_idx, grid_R, bound_min, bound_max):
    """Index-space marching-cubes vertices (fp64 [N,3] on the device) -> world coordinates IN PLACE: v / (R - 1) * (bound_max - bound_min) + bound_min in
    fp64 (sparse_neus_renderer.py:936), the same IEEE expression numpy evaluates on the host.  bound_min / bound_max: three float32 numbers each."""
    b0 = np.asarray(bound_min.detach().cpu() if torch.is_tensor(bound_min) else bound_min, np.float32).reshape(3)
    b1 = np.asarray(bound_max.detach().cpu() if torch.is_tensor(bound_max) else bound_max, np.float32).reshape(3)
    ext = np.ascontiguousarray((b1 - b0).astype(np.float64))          # the reference subtracts its float32 bound arrays, numpy promotes afterwards
    off = np.ascontiguousarray(b0.astype(np.float64))
    check(_lib.lib().o2345_mc_verts_to_world(_p(verts_idx, torch.float64), verts_idx.shape[0], int(grid_R), ext.ctypes.data_as(ctypes.c_void_p),
                                             off.ctypes.data_as(ctypes.c_void_p), _stream()), "mc_verts_to_world")
    return verts_idx


@_on_device
def marching_cubes(u, iso=0.0, index_dtype=torch.int64):
    """u: float32 cuda tensor [n0,n1,n2].  Returns (verts float64 [Nv,3] index coords, tris [Nt,3]) on the device."""
    L = _lib.lib()
    n0, n1, n2 = u.shape
    wsb = L.o2345_mc_workspace_bytes(n0, n1, n2)
    ws = _workspace(wsb, u.device, "mc")
    nv, nt = ctypes.c_longlong(), ctypes.c_longlong()
    check(L.o2345_marching_cubes_count(_p(u), n0, n1, n2, float(iso), _p(ws, torch.uint8), wsb, ctypes.byref(nv), ctypes.byref(nt),
                                       _stream()), "marching_cubes_count")
    verts = torch.empty(nv.value, 3, dtype=torch.float64, device=u.device)
    tris = torch.empty(nt.value, 3, dtype=index_dtype, device=u.device)
    check(L.o2345_marching_cubes_emit(_p(u), n0, n1, n2, float(iso), _p(ws, torch.uint8), _p(verts, torch.float64),
                                      _p(tris, index_dtype), 8 if index_dtype == torch.int64 else 4, _stream()), "marching_cubes_emit")
    return verts, tris
