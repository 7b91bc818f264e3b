// TEST-ONLY host build of the per-element math that the HIP kernels execute on the device
// (one-2-3-45_amd/csrc/*_math.h are __host__ __device__).  This library is never loaded by the product package;
// it lets the CPU test-suite compare the exact source the GPU runs against the oracle before any GPU time is spent.
#include "../../one-2-3-45_amd/csrc/costvol_math.h"
#include "../../one-2-3-45_amd/csrc/render_math.h"
#include "../../one-2-3-45_amd/csrc/pe_math.h"

using namespace o2345;
namespace o2345 { void set_error(const char*, ...) {} }

extern "C" {

void hc_sincos_pe(const float* x, int n, float* s, float* c) {
    for (int i = 0; i < n; ++i) sincos_pe(x[i], s[i], c[i]);
}

int hc_costvol(const float* feats_nhwc, const float* proj, int V, int H, int W, int dx, int dy, int dz, float vs,
               const float* origin, int min_views, uint8_t* cnt, int* row_of_voxel, int* coords, float* rows) {
    VolGeom g{dx, dy, dz, vs, origin[0], origin[1], origin[2]};
    const long long nvox = (long long)dx * dy * dz;
    int n = 0;
    for (long long v = 0; v < nvox; ++v) {
        int x, y, z;
        voxel_xyz(v, g, x, y, z);
        int c = visible_views(proj, V, H, W, g, x, y, z);
        cnt[v] = (uint8_t)c;
        if (c > min_views) { row_of_voxel[v] = n; coords[4 * n] = x; coords[4 * n + 1] = y; coords[4 * n + 2] = z; coords[4 * n + 3] = 0; ++n; }
        else row_of_voxel[v] = -1;
    }
    if (rows)
        for (int r = 0; r < n; ++r)
            for (int q = 0; q < 4; ++q) costvol_row<16>(feats_nhwc, proj, V, H, W, g, cnt, coords, r, q, rows);
    return n;
}

// visible-view counts of every voxel of a dx x dy x dz lattice (the integer result that decides the kept-voxel set)
void hc_visible_views(const float* proj, int V, int H, int W, int dx, int dy, int dz, float vs, const float* origin, uint8_t* cnt) {
    VolGeom g{dx, dy, dz, vs, origin[0], origin[1], origin[2]};
    const long long nvox = (long long)dx * dy * dz;
    for (long long v = 0; v < nvox; ++v) {
        int x, y, z;
        voxel_xyz(v, g, x, y, z);
        cnt[v] = (uint8_t)visible_views(proj, V, H, W, g, x, y, z);
    }
}

void hc_trilinear_ref(const float* vol_cl, int D, const float* pts, int P, float* out /*[P,16]*/) {
    for (int p = 0; p < P; ++p) {
        Taps3D t = trilinear_ref_taps(pts[3 * p], pts[3 * p + 1], pts[3 * p + 2], D);
        for (int c = 0; c < 16; ++c) out[16 * p + c] = 0.f;
        if (!t.ok) continue;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int d = 0; d < 2; ++d) {
            const float w = t.fz[d] * t.fy[b] * t.fx[a];
            const float* v = vol_cl + (((size_t)t.ix[a] * D + t.iy[b]) * D + t.iz[d]) * 16;
            for (int c = 0; c < 16; ++c) out[16 * p + c] += v[c] * w;
        }
    }
}

void hc_mask_nearest(const float* maskvol, int D, const float* pts, int P, float* out) {
    for (int p = 0; p < P; ++p) out[p] = mask_at(maskvol, D, pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]);
}

void hc_upsample(const float* ro, const float* rd, int R, const float* z, const float* sdf, int S, float inv_s,
                 const float* maskvol, int D, float* wbuf, int n_imp, float* new_z) {
    RayGeom g{ro, rd, R};
    for (int r = 0; r < R; ++r) upsample_ray(g, r, z, sdf, S, inv_s, maskvol, D, wbuf, n_imp, new_z);
}

void hc_merge(int R, float* z, float* sdf, int S, float* new_z, float* new_sdf, int n_new) {
    for (int r = 0; r < R; ++r) merge_ray(r, R, z, sdf, S, new_z, new_sdf, n_new);
}

void hc_composite(const float* ro, const float* rd, int R, int S, const float* mid_z, const float* dists, const float* pm,
                  const float* sdf, const float* grad, const float* rgb, const uint8_t* nviews, float inv_s, float air, float bg,
                  float* color, float* depth, float* weights, float* cdf, float* wsum, float* wmax, float* dvar, float* asum,
                  float* gerr, uint8_t* cmask) {
    RayGeom g{ro, rd, R};
    CompositeOut o{color, depth, weights, cdf, wsum, wmax, dvar, asum, gerr, cmask};
    for (int r = 0; r < R; ++r) composite_ray(g, r, S, mid_z, dists, pm, sdf, grad, rgb, nviews, inv_s, air, bg, o);
}

float hc_linspace(float a, float b, int n, int i) { return linspace_at(a, b, n, i); }

}  // extern "C"
