// Per-element geometry shared by the HIP kernels (device) and the host-side check build (tests/hostcheck):
// voxel -> image projection, 2-D bilinear taps, the reference's trilinear sampler and the nearest-mask lookup.
// Each routine cites the reference lines it reproduces (paths relative to reconstruction/).
#pragma once
#include "common.h"

namespace o2345 {

// ops/back_project.py:44-63.  P = rows of the 4x4 affine matrix (row-major, 16 floats).
// z >= 0 is clamped to >= 1e-6, negative z untouched; valid = |gx|<=1 & |gy|<=1 & z>0.
O2345_HD void project_voxel(const float* __restrict__ P, float wx, float wy, float wz, int H, int W,
                            float& gx, float& gy, bool& valid) {
    // The reference forms these with a GEMM over k (proj_batch @ rs_grid, ops/back_project.py:52): BLAS kernels (MKL on the host, cuBLAS / rocBLAS on a GPU)
    // accumulate a tiny-K product as ONE FMA chain in k order, acc = fma(a_k, b_k, acc) -- which is also what ATen's CPU matmul does (checked on
    // 16.7 M voxels x 8 views: bit-identical to this chain, while the mul / add sequence differs in the last bit of ~15 % of the values and flips the
    // frustum test of 3 voxels in 256^3).  The translation enters with b_3 = 1, so the last step is an exact add.
    float x = fmaf(P[2], wz, fmaf(P[1], wy, P[0] * wx)) + P[3];
    float y = fmaf(P[6], wz, fmaf(P[5], wy, P[4] * wx)) + P[7];
    float z = fmaf(P[10], wz, fmaf(P[9], wy, P[8] * wx)) + P[11];
    if (z >= 0.f) z = fmaxf(z, 1e-6f);
    gx = 2.f * (x / z) / (float)(W - 1) - 1.f;
    gy = 2.f * (y / z) / (float)(H - 1) - 1.f;
    valid = (fabsf(gx) <= 1.f) && (fabsf(gy) <= 1.f) && (z > 0.f);
}

// ATen grid_sample 2-D, bilinear, padding zeros, align_corners=True (ops/back_project.py:73, render_utils.py:115):
// pixel = (g+1)/2*(size-1); the four taps and their weights.  Out-of-image taps get weight 0 and a clamped index.
struct Taps2D {
    int idx[4];     // y*W + x of nw, ne, sw, se (clamped into the image)
    float w[4];     // 0 where the tap is outside
};

O2345_HD Taps2D bilinear_taps(float gx, float gy, int H, int W) {
    Taps2D t;
    float ix = (gx + 1.f) / 2.f * (float)(W - 1);
    float iy = (gy + 1.f) / 2.f * (float)(H - 1);
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.f, y1 = y0 + 1.f;
    float wx1 = ix - x0, wx0 = x1 - ix, wy1 = iy - y0, wy0 = y1 - iy;
    bool bx0 = (x0 >= 0.f) && (x0 <= (float)(W - 1)), bx1 = (x1 >= 0.f) && (x1 <= (float)(W - 1));
    bool by0 = (y0 >= 0.f) && (y0 <= (float)(H - 1)), by1 = (y1 >= 0.f) && (y1 <= (float)(H - 1));
    int xi0 = (int)fminf(fmaxf(x0, 0.f), (float)(W - 1)), xi1 = (int)fminf(fmaxf(x1, 0.f), (float)(W - 1));
    int yi0 = (int)fminf(fmaxf(y0, 0.f), (float)(H - 1)), yi1 = (int)fminf(fmaxf(y1, 0.f), (float)(H - 1));
    t.idx[0] = yi0 * W + xi0; t.w[0] = (bx0 && by0) ? wx0 * wy0 : 0.f;
    t.idx[1] = yi0 * W + xi1; t.w[1] = (bx1 && by0) ? wx1 * wy0 : 0.f;
    t.idx[2] = yi1 * W + xi0; t.w[2] = (bx0 && by1) ? wx0 * wy1 : 0.f;
    t.idx[3] = yi1 * W + xi1; t.w[3] = (bx1 && by1) ? wx1 * wy1 : 0.f;
    return t;
}

// ops/grid_sampler.py:64-216 -- the reference's own trilinear sampler (cubic volume of side D):
// i = (g+1)/2*(D-1); output is zero unless 0 < i < D on all three axes; the 8 corner indices are clamped to
// [0,D-1] but the weights use the UNclamped corners.
struct Taps3D {
    int ix[2], iy[2], iz[2];   // clamped corner indices per axis (volume axes x,y,z)
    float fx[2], fy[2], fz[2]; // per-axis weight factors: f[0] = (i0+1-i), f[1] = (i-i0)
    bool ok;
};

O2345_HD Taps3D trilinear_ref_taps(float px, float py, float pz, int D) {
    Taps3D t;
    const float s = (float)(D - 1);
    float ix = (px + 1.f) / 2.f * s, iy = (py + 1.f) / 2.f * s, iz = (pz + 1.f) / 2.f * s;
    const float Df = (float)D;
    t.ok = (ix > 0.f) && (ix < Df) && (iy > 0.f) && (iy < Df) && (iz > 0.f) && (iz < Df);
    float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    t.fx[0] = (x0 + 1.f) - ix; t.fx[1] = ix - x0;
    t.fy[0] = (y0 + 1.f) - iy; t.fy[1] = iy - y0;
    t.fz[0] = (z0 + 1.f) - iz; t.fz[1] = iz - z0;
    // clamp in float first: masked-out points may carry huge / NaN coordinates
    auto cl = [s](float v) { return (int)fminf(fmaxf(v, 0.f), s); };
    t.ix[0] = cl(x0); t.ix[1] = cl(x0 + 1.f);
    t.iy[0] = cl(y0); t.iy[1] = cl(y0 + 1.f);
    t.iz[0] = cl(z0); t.iz[1] = cl(z0 + 1.f);
    return t;
}

// F.grid_sample(mode='nearest', align_corners=False, zeros) on the occupancy volume
// (sparse_neus_renderer.py:153-169): index = nearbyint(((g+1)*D-1)/2) (half-to-even); out of range -> -1.
O2345_HD int nearest_index(float g, int D) {
    float i = nearbyintf(((g + 1.f) * (float)D - 1.f) / 2.f);
    return (i >= 0.f && i <= (float)(D - 1)) ? (int)i : -1;
}

O2345_HD int nearest_voxel(float px, float py, float pz, int D) {
    int x = nearest_index(px, D), y = nearest_index(py, D), z = nearest_index(pz, D);
    return (x < 0 || y < 0 || z < 0) ? -1 : (x * D + y) * D + z;
}

// ATen grid_sample 3-D trilinear, zeros, align_corners=True (render_utils.py:54-85): per-axis corner + weight,
// weight 0 when the corner is outside.
struct Axis2 {
    int i[2];
    float w[2];
};
O2345_HD Axis2 axis_taps_zeros(float g, int D) {
    Axis2 a;
    float s = (float)(D - 1);
    float f = (g + 1.f) / 2.f * s;
    float f0 = floorf(f), f1 = f0 + 1.f;
    a.w[0] = (f0 >= 0.f && f0 <= s) ? (f1 - f) : 0.f;
    a.w[1] = (f1 >= 0.f && f1 <= s) ? (f - f0) : 0.f;
    a.i[0] = (int)fminf(fmaxf(f0, 0.f), s);
    a.i[1] = (int)fminf(fmaxf(f1, 0.f), s);
    return a;
}

}  // namespace o2345
