"""numpy emulators of the network kernels' register-level data flow (TEST INFRASTRUCTURE, used by tests/test_weights_packing.py only):
they walk the packed operand blobs of one-2-3-45_amd/weights.py lane by lane exactly as csrc/sdf_mlp*.hip / csrc/color_mfma.hip do, so a
packing error shows up on the CPU.  Moved out of the product package (they are not needed to run it)."""
import importlib

import numpy as np

_w = importlib.import_module("one-2-3-45_amd.weights")
globals().update({k: getattr(_w, k) for k in dir(_w) if not k.startswith("__")})


def emulate_sdf_blob(blob, pts, lat, grad_lat_jac=None):
    """Numpy emulation of csrc/sdf_mlp.hip's dataflow (one wave, 32 points) using the documented lane layouts of
    v_mfma_f32_32x32x2_f32.  Used by the CPU tests to pin the packing; returns (y[128] per point, dsdf/dpe, dsdf/dlat)."""
    pts = np.asarray(pts, np.float64)
    P = pts.shape[0]
    assert P <= 32
    lane = np.arange(64)
    j, h = lane & 31, lane >> 5
    live = j < P
    jj = np.minimum(j, P - 1)

    def mfma(a, b, c):
        # a[lane] = A[i=lane&31][k=lane>>5], b[lane] = B[k=lane>>5][j=lane&31]; c[lane][r] = D[(r&3)+8(r>>2)+4(lane>>5)][lane&31]
        A = np.zeros((32, 2)); B = np.zeros((2, 32))
        A[lane & 31, lane >> 5] = a
        B[lane >> 5, lane & 31] = b
        D = A @ B
        out = c.copy()
        for r in range(16):
            out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
        return out

    def softplus(a):
        t = a * 100
        z = np.exp(np.minimum(t, 50))
        return np.where(t > 20, a, np.log1p(z) / 100), np.where(t > 20, 1.0, z / (z + 1))

    misc = blob[OFF_MISC:OFF_MISC + MISC_SIZE].astype(np.float64)
    pe = np.zeros((64, 20))
    for t in range(9):
        c = 9 * h + t
        f = 2.0 ** (c // 3)
        x = pts[jj, t % 3]
        pe[:, t], pe[:, 9 + t] = np.sin(x * f), np.cos(x * f)
    pe[:, 18] = np.where(h == 1, pts[jj, 2], pts[jj, 0])
    pe[:, 19] = np.where(h == 1, 0.0, pts[jj, 1])
    latl = np.stack([lat[jj, 8 * h + t] for t in range(8)], 1)

    def layer(off, nst, bias_off, bsrc):
        acc = [np.stack([misc[bias_off + (nb * 16 + r) * 2 + h] for r in range(16)], 1) for nb in range(4)]
        A = blob[off:off + 4 * nst * 64].reshape(4, nst, 64).astype(np.float64)
        for s in range(nst):
            for nb in range(4):
                acc[nb] = mfma(A[nb, s], bsrc(s), acc[nb])
        return acc

    a0 = layer(OFF_A0, ST0, MISC_B0, lambda s: pe[:, s])
    h0, s0 = zip(*[softplus(a) for a in a0])
    a1 = layer(OFF_A1, ST1, MISC_B1, lambda s: h0[s // 16][:, s % 16] if s < 64 else latl[:, s - 64])
    h1, s1 = zip(*[softplus(a) for a in a1])
    a2 = layer(OFF_A2, ST1, MISC_B2, lambda s: h1[s // 16][:, s % 16] if s < 64 else latl[:, s - 64])
    y = np.zeros((P, 128))
    for nb in range(4):
        for r in range(16):
            for l in lane[live]:
                y[j[l], neuron_of(nb, r, h[l])] = a2[nb][l, r]
    # backward
    g1 = [np.stack([misc[MISC_W2H + (nb * 16 + r) * 2 + h] for r in range(16)], 1) * s1[nb] for nb in range(4)]
    A1T = blob[OFF_A1T:OFF_A0T].reshape(5, STB, 64).astype(np.float64)
    g = [np.zeros((64, 16)) for _ in range(5)]
    for s in range(STB):
        for nb in range(5):
            g[nb] = mfma(A1T[nb, s], g1[s // 16][:, s % 16], g[nb])
    g0 = [g[nb] * s0[nb] for nb in range(4)]
    A0T = blob[OFF_A0T:OFF_MISC].reshape(2, STB, 64).astype(np.float64)
    gp = [np.zeros((64, 16)) for _ in range(2)]
    for s in range(STB):
        for nb in range(2):
            gp[nb] = mfma(A0T[nb, s], g0[s // 16][:, s % 16], gp[nb])
    gpe = np.zeros((P, 39)); glat = np.zeros((P, 16))
    for l in lane[live]:
        for t in range(20):
            col = pe_index(t, int(h[l]))
            if col >= 0:
                gpe[j[l], col] = gp[0][l, t] if t < 16 else gp[1][l, t - 16]
        for t in range(8):
            glat[j[l], 8 * h[l] + t] = g[4][l, t] + misc[MISC_W2L + 8 * h[l] + t]
    return y, gpe, glat


def emulate_sdf_blob_x3(blob, pts, lat):
    """Numpy emulation of csrc/sdf_mlp_x3.hip's forward pass (one wave, 32 points): split-f16 operands, exact products,
    hi*hi + hi*lo + lo*hi, v_mfma_f32_32x32x16_f16 lane layout.  Returns sdf[P]."""
    pts = np.asarray(pts, np.float64)
    P = pts.shape[0]
    assert P <= 32
    lane = np.arange(64)
    j, h = lane & 31, lane >> 5
    live = j < P
    jj = np.minimum(j, P - 1)

    def mfma16(a8, b8, c):
        A = np.zeros((32, 16)); B = np.zeros((16, 32))
        for t in range(8):
            A[j, 8 * h + t] = a8[:, t]; B[8 * h + t, j] = b8[:, t]
        D = A @ B
        out = c.copy()
        for r in range(16):
            out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * h, j]
        return out

    def softplus(a):
        t = a * 100
        return np.where(t > 20, a, np.log1p(np.exp(np.minimum(t, 50))) / 100)

    def sec(off, nb, nst):
        return blob[off:off + nb * nst * 512].view(np.float16).reshape(nb, nst, 2, 64, 8).astype(np.float64)

    def layer(A, nst, bias_off, bsrc):
        acc = [np.stack([misc[bias_off + (nb * 16 + r) * 2 + h] for r in range(16)], 1) for nb in range(4)]
        for st in range(nst):
            bh, bl = (v.astype(np.float64) for v in f16_split_device(bsrc(st)))
            for nb in range(4):
                acc[nb] = mfma16(A[nb, st, 1], bh, acc[nb])
                acc[nb] = mfma16(A[nb, st, 0], bl, acc[nb])
                acc[nb] = mfma16(A[nb, st, 0], bh, acc[nb])
        return acc

    from importlib import import_module
    Wm = import_module("one-2-3-45_amd.weights")
    misc = blob[Wm.OFFX_MISC:Wm.OFFX_MISC + MISC_SIZE].astype(np.float64)          # b0, b1 in the t domain (weights.SOFTPLUS_SCALE)
    c_scale = Wm.SOFTPLUS_SCALE

    def softplus(t):                            # s' = softplus(a) * 100 / ln 2 from t = 100 a / ln 2: max(t, 0) + log2(1 + 2^-|t|)
        return np.maximum(t, 0.0) + np.log2(1.0 + np.exp2(-np.abs(t)))
    pe = np.zeros((64, 24))
    for t in range(9):
        c = 9 * h + t
        f = 2.0 ** (c // 3)
        x = pts[jj, t % 3]
        pe[:, t], pe[:, 9 + t] = np.sin(x * f), np.cos(x * f)
    pe[:, 18] = np.where(h == 1, pts[jj, 2], pts[jj, 0])
    pe[:, 19] = np.where(h == 1, 0.0, pts[jj, 1])
    latl = np.stack([lat[jj, 8 * h + t] for t in range(8)], 1)
    a0 = layer(sec(OFFX_A0, 4, STX0), STX0, MISC_B0, lambda st: pe[:, 8 * st:8 * st + 8])
    h0 = [softplus(a) for a in a0]
    a1 = layer(sec(OFFX_A1, 4, STH1), STH1, MISC_B1,
               lambda st: h0[st >> 1][:, 8 * (st & 1):8 * (st & 1) + 8] if st < 8 else latl)
    h1 = [softplus(a) for a in a1]
    w2h = [np.stack([misc[MISC_W2H + (nb * 16 + r) * 2 + h] for r in range(16)], 1) for nb in range(4)]
    part = sum((w2h[nb] * h1[nb]).sum(1) for nb in range(4)) / c_scale + sum(misc[MISC_W2L + 8 * h + t] * latl[:, t] for t in range(8))
    sdf = np.zeros(P)
    for l in lane[live & (h == 0)]:
        sdf[j[l]] = part[l] + part[l + 32] + misc[MISC_B2]
    return sdf



def emulate_color_mfma(blob, geo, rf64, rd, m, G, x3_blob=None):
    """Numpy emulation of csrc/color_mfma.hip for ONE wave tile (32 columns = 32/G points x G views), fp64.
    geo [P,16], rf64 [P,G,64] (pixel floats: rgb | feat | pad), rd [P,G,4], m [P,G] -> rgb [P,3].
    x3_blob: emulate the split-f16 instantiation instead (A operands from pack_color_x3_blob, activations split as on the device)."""
    P = 32 // G
    lane = np.arange(64)
    j, h = lane & 31, lane >> 5
    pt, v = j // G, j % G

    def mfma(a, b, c):
        A = np.zeros((32, 2)); B = np.zeros((2, 32))
        A[lane & 31, lane >> 5] = a
        B[lane >> 5, lane & 31] = b
        D = A @ B
        out = c.copy()
        for r in range(16):
            out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
        return out

    def mfma16(a8, b8, c):                      # v_mfma_f32_32x32x16_f16: lane supplies k = 8*(lane>>5) + t
        A = np.zeros((32, 16)); B = np.zeros((16, 32))
        for t in range(8):
            A[j, 8 * h + t] = a8[:, t]; B[8 * h + t, j] = b8[:, t]
        D = A @ B
        out = c.copy()
        for r in range(16):
            out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * h, j]
        return out

    def layer(aname, bname, bsrc):
        off, nb, ns = CM_LAYOUT[aname]
        boff = CM_LAYOUT[bname][0]
        Bv = blob[boff:boff + nb * 32].reshape(nb, 2, 16).astype(np.float64)
        acc = [np.stack([Bv[b, h, r] for r in range(16)], 1) for b in range(nb)]
        if x3_blob is None:
            A = blob[off:off + nb * ns * 64].reshape(nb, ns, 64).astype(np.float64)
            for s in range(ns):
                for b in range(nb):
                    acc[b] = mfma(A[b, s], bsrc(s), acc[b])
            return acc
        offx, _, nsx = CX_LAYOUT[aname]
        A = x3_blob[offx:offx + nb * nsx * 512].view(np.float16).reshape(nb, nsx, 2, 64, 8).astype(np.float64)
        for s in range(nsx):
            b8 = np.stack([bsrc(8 * s + t) if 8 * s + t < ns else np.zeros(64) for t in range(8)], 1)
            bh, bl = (x.astype(np.float64) for x in f16_split_device(b8))
            for b in range(nb):
                acc[b] = mfma16(A[b, s, 1], bh, acc[b])
                acc[b] = mfma16(A[b, s, 0], bl, acc[b])
                acc[b] = mfma16(A[b, s, 0], bh, acc[b])
        return acc

    LG = 1.4426950408889634                               # log2(e): the kernel's scaled domain
    elu = lambda y: np.maximum(y, LG * (np.minimum(np.exp2(y), 1.0) - 1.0))      # ELU_y(y) = log2e * ELU(y / log2e)
    sig = lambda z: 1 / (1 + np.exp2(-z))                 # sigmoid(z / log2e)
    gsum = lambda x: np.array([x[(pt == pt[l]) & (h == h[l])].sum() for l in lane])
    gmin = lambda x: np.array([x[(pt == pt[l]) & (h == h[l])].min() for l in lane])
    gmax = lambda x: np.array([x[(pt == pt[l]) & (h == h[l])].max() for l in lane])
    rdl = rd[pt, v].astype(np.float64)                         # [64,4]
    ml = m[pt, v].astype(np.float64)
    rf = np.stack([rf64[pt, v, 32 * h + t] for t in range(32)], 1).astype(np.float64) * LG  # lane's half pixel, scaled domain
    d16 = elu(layer("A_RD0", "B_RD0", lambda s: rdl[lane, 2 * s + h])[0])
    dfe = layer("A_RD1", "B_RD1", lambda s: d16[:, s])
    for b in range(2):
        for r in range(16):
            rf[:, 16 * b + r] += elu(dfe[b][:, r])
    rgb_in = rf64[pt, v, :3].astype(np.float64)              # colours BEFORE the direction feature
    s_par = float(blob[CM_LAYOUT["S_SCALAR"][0]])
    e = np.exp2(abs(s_par) * LG * (rdl[:, 3] - 1))
    wgt = (e - gmin(e)) * ml
    wgt = wgt / (gsum(wgt) + 1e-8)
    mean = np.stack([gsum(rf[:, t] * wgt) for t in range(32)], 1)
    var = np.stack([gsum(wgt * (rf[:, t] - mean[:, t]) ** 2) for t in range(32)], 1)
    off = CM_LAYOUT["W_S"][0]
    WS = blob[off:off + 144 * 64].reshape(144, 64).astype(np.float64)
    S = np.zeros((P, 64))
    for p in range(P):
        l0 = np.nonzero((pt == p) & (v == 0) & (h == 0))[0][0]
        l1 = np.nonzero((pt == p) & (v == 0) & (h == 1))[0][0]
        S[p] = geo[p].astype(np.float64) @ WS[:16] + mean[l0] @ WS[16:48] + mean[l1] @ WS[48:80] + var[l0] @ WS[80:112] + var[l1] @ WS[112:144]
    a0 = layer("A_B0", "B_B0", lambda s: rf[:, s])
    for b in range(2):
        for r in range(16):
            a0[b][:, r] += S[pt, [neuron_of(b, r, hh) for hh in h]]
    h64 = [elu(a) for a in a0]
    x32 = elu(layer("A_B1", "B_B1", lambda s: h64[s // 16][:, s % 16])[0])
    t32 = elu(layer("A_V0", "B_V0", lambda s: x32[:, s] * wgt)[0])
    v1 = layer("A_V1", "B_V1", lambda s: t32[:, s])
    so = CM_LAYOUT["S_SCALAR"][0]

    def dot_all(name, regs, nreg, bias_v):                    # per-lane partial dot over its registers + the other half's
        off = CM_LAYOUT[name][0]
        wv = blob[off:off + 32].reshape(2, 16).astype(np.float64)
        part = sum(regs[:, r] * wv[h, r] for r in range(nreg))
        return part + part[lane ^ 32] + bias_v
    vis = sig(elu(dot_all("V_V1X", t32, 16, float(blob[so + 1])))) * ml
    x32 = x32 + elu(v1[0])
    t32 = elu(layer("A_V20", "B_V20", lambda s: x32[:, s] * vis)[0])
    vis2 = sig(dot_all("V_V21", t32, 16, float(blob[so + 2]))) * ml
    extra = [np.where(h == 0, vis2, rdl[:, 0]), np.where(h == 0, rdl[:, 1], rdl[:, 2]), np.where(h == 0, rdl[:, 3], 0.0)]
    r16 = elu(layer("A_R0", "B_R0", lambda s: x32[:, s] if s < 16 else extra[s - 16])[0])
    r8 = elu(layer("A_R1", "B_R1", lambda s: r16[:, s])[0])
    score = dot_all("V_R2", r8, 4, float(blob[so + 3]))
    score = np.where(ml == 0, -1e9, score)
    ex = np.exp2(score - gmax(score))
    bw = ex / gsum(ex)
    out = np.zeros((P, 3))
    for p in range(P):
        sel = (pt == p) & (h == 0)
        out[p] = (rgb_in[sel] * bw[sel, None]).sum(0)
    return out


