"""a6: independent cross-check of the oracle's sparse convolutions (oracle/recon.py: sparse_conv / downsample_coords /
build_kmap, restated from torchsparse v1.4.0's published algorithm) against ATen's DENSE convolutions on the voxel grid
(SURVEY section 4 (ii)).  torchsparse is not vendored in /root/reference, so this is the strongest pin available: a
sparse convolution on an active set is by definition the dense convolution of the zero-filled grid, read at the active
output sites.

  stride 1 (conv0/2/4/6, tsparse/modules.py:94-107)   == F.conv3d(padding=1)         at the input's own active set
  stride 2 (conv1/3/5)                                 == F.conv3d(stride=2, padding=1) at the down-sampled active set
  transposed stride 2 (conv7/9/11, modules.py:110-124) == F.conv_transpose3d(stride=2, padding=1) at the fine active set

Kernel-offset enumeration: offset k = (dx, dy, dz) with x fastest (get_kernel_offsets, odd kernel volume), i.e. dense
weight w[co, ci, dx+1, dy+1, dz+1] = K[k][ci, co].  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import recon as O


def _active_set(D, n, seed, lo=0):
    rng = np.random.default_rng(seed)
    # a blob + scattered voxels: both dense neighbourhoods and isolated sites, none on the first `lo` planes
    c = rng.integers(lo, D, (n, 3))
    blob = np.stack(np.meshgrid(*[np.arange(D // 3, D // 3 + 4)] * 3, indexing="ij"), -1).reshape(-1, 3)
    xyz = np.unique(np.concatenate([c, blob]), axis=0)             # lexicographic (x,y,z) = the reference's row order
    return torch.from_numpy(xyz).long()


def _dense(xyz, feat, cells, ts=1):
    C = feat.shape[1]
    g = torch.zeros(1, C, *cells, dtype=feat.dtype)
    i = xyz // ts
    g[0, :, i[:, 0], i[:, 1], i[:, 2]] = feat.T
    return g


def _dense_weight(K):
    """K [27,Ci,Co] (x fastest) -> conv3d weight [Co,Ci,3,3,3] indexed (dx,dy,dz)."""
    Ci, Co = K.shape[1:]
    return K.reshape(3, 3, 3, Ci, Co).permute(4, 3, 2, 1, 0).contiguous()      # [z][y][x] -> [.., x, y, z]


@pytest.mark.parametrize("seed,D,Ci,Co", [(0, 12, 5, 7), (1, 17, 16, 8), (2, 9, 3, 4)])
def test_stride1_equals_masked_conv3d(seed, D, Ci, Co):
    g = torch.Generator().manual_seed(seed)
    xyz = _active_set(D, 80, seed)
    x = torch.randn(len(xyz), Ci, generator=g, dtype=torch.float64)
    K = torch.randn(27, Ci, Co, generator=g, dtype=torch.float64)
    L = O.SparseLevel(xyz, 1)
    got = O.sparse_conv(x, O.build_kmap(L, L), K)
    ref = F.conv3d(_dense(xyz, x, (D, D, D)), _dense_weight(K), padding=1)[0]
    want = ref[:, xyz[:, 0], xyz[:, 1], xyz[:, 2]].T
    assert torch.allclose(got, want, rtol=0, atol=1e-10)


@pytest.mark.parametrize("seed,D,Ci,Co,lo", [(3, 12, 4, 6, 0), (4, 15, 8, 8, 1), (5, 10, 3, 5, 3)])
def test_stride2_down_and_transposed_up(seed, D, Ci, Co, lo):
    """`lo` > 0 leaves the first planes empty so that the `>= per-axis minimum` rule of spdownsample is exercised."""
    g = torch.Generator().manual_seed(seed)
    xyz = _active_set(D, 60, seed, lo=lo)
    xyz = xyz[(xyz >= lo).all(1)]
    x = torch.randn(len(xyz), Ci, generator=g, dtype=torch.float64)
    K = torch.randn(27, Ci, Co, generator=g, dtype=torch.float64)
    L0 = O.SparseLevel(xyz, 1)
    L1 = O.downsample_coords(L0)
    G = D + 2 + D % 2                   # even dense extent with a free plane behind D-1 (coarse sites may sit at D)
    # (1) coordinate set: even-lattice sites with an active fine voxel within +-1, not below the per-axis minimum, sorted
    occ = F.max_pool3d(_dense(xyz, torch.ones(len(xyz), 1, dtype=torch.float64), (G, G, G)), 3, stride=1, padding=1)[0, 0]
    q = torch.nonzero(occ[::2, ::2, ::2] > 0) * 2
    q = q[(q >= xyz.min(0).values[None]).all(1)]
    assert torch.equal(L1.xyz, q), "down-sampled coordinate set / order"
    assert (L1.xyz % 2 == 0).all() and L1.ts == 2
    # (2) values of the strided convolution
    k01 = O.build_kmap(L0, L1)
    down = O.sparse_conv(x, k01, K)
    ref = F.conv3d(_dense(xyz, x, (G, G, G)), _dense_weight(K), stride=2, padding=1)[0]
    c = L1.xyz // 2
    assert torch.allclose(down, ref[:, c[:, 0], c[:, 1], c[:, 2]].T, rtol=0, atol=1e-10)
    # (3) transposed convolution back to the fine set with the cached maps (roles swapped)
    Kt = torch.randn(27, Co, Ci, generator=g, dtype=torch.float64)
    up = O.sparse_conv(down, k01, Kt, transposed=True, n_out=len(xyz))
    nc = ref.shape[1:]
    wt = Kt.reshape(3, 3, 3, Co, Ci).permute(3, 4, 2, 1, 0).contiguous()       # conv_transpose3d weight [Cin, Cout, x, y, z]
    dense_up = F.conv_transpose3d(_dense(L1.xyz, down, nc, ts=2), wt, stride=2, padding=1, output_padding=1)[0]
    assert torch.allclose(up, dense_up[:, xyz[:, 0], xyz[:, 1], xyz[:, 2]].T, rtol=0, atol=1e-10)


def test_coarse_levels_and_unet_against_dense():
    """The whole SparseCostRegNet (modules.py:259-304) evaluated with dense ATen convolutions + masked batch statistics on the
    active sets equals oracle.sparse_costreg."""
    import importlib
    pkg = importlib.import_module("one-2-3-45_amd")
    from scene_util import costreg_oracle_weights
    D = 14
    xyz = _active_set(D, 150, 7)
    g = torch.Generator().manual_seed(7)
    feat = torch.randn(len(xyz), 32, generator=g)
    w = costreg_oracle_weights(pkg.weights.init_costreg_state_dict(3))
    coords = torch.cat([xyz.int(), torch.zeros(len(xyz), 1, dtype=torch.int32)], 1)
    got, extra = O.sparse_costreg(feat, coords, w)
    L = extra["levels"]

    def bn(rows, gm, bt):
        mu, var = rows.mean(0), rows.var(0, unbiased=False)
        return torch.relu((rows - mu) / torch.sqrt(var + 1e-5) * gm + bt)

    def cells(lv):
        return (32 // lv.ts,) * 3            # even dense extents 32/16/8/4 >= every level's largest coordinate + 1

    def conv(name, rows, lin, lout, mode):
        K, gm, bt = w[name]
        xin = _dense(lin.xyz, rows, cells(lin), lin.ts)
        if mode == "same":
            y = F.conv3d(xin, _dense_weight(K), padding=1)
        elif mode == "down":
            y = F.conv3d(xin, _dense_weight(K), stride=2, padding=1)
        else:
            wt = K.reshape(3, 3, 3, K.shape[1], K.shape[2]).permute(3, 4, 2, 1, 0).contiguous()
            y = F.conv_transpose3d(xin, wt, stride=2, padding=1, output_padding=1)
        c = lout.xyz // lout.ts
        return bn(y[0][:, c[:, 0], c[:, 1], c[:, 2]].T, gm, bt)

    c0 = conv("conv0", feat, L[0], L[0], "same")
    c2 = conv("conv2", conv("conv1", c0, L[0], L[1], "down"), L[1], L[1], "same")
    c4 = conv("conv4", conv("conv3", c2, L[1], L[2], "down"), L[2], L[2], "same")
    x = conv("conv6", conv("conv5", c4, L[2], L[3], "down"), L[3], L[3], "same")
    x = c4 + conv("conv7", x, L[3], L[2], "up")
    x = c2 + conv("conv9", x, L[2], L[1], "up")
    x = c0 + conv("conv11", x, L[1], L[0], "up")
    assert (got - x).abs().max().item() < 2e-4 * max(1.0, x.abs().max().item())
