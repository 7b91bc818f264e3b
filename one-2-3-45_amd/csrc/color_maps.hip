// Small kernels around the colour path (SURVEY 8a rows a20 / a22 helpers): the per-point valid-view count that feeds the per-ray colour mask
// (models/rendering_network.py:124-128), the channel-last colour map [V,H,W,64] = rgb | 56 features | pad from the reference's channel-first tensors
// (models/projector.py:96-228 gathers from it), and the camera terms of the Projector (proj = K @ w2c[:3], camera centres) in one launch.
#include "common.h"
#include "geom_math.h"

namespace o2345 {

// cam2pixel (ops/back_project.py:89-129) with padding 'zeros': Z clamped to >= 1e-3, out-of-range coordinate -> 2
__device__ __forceinline__ void project_point(const float* __restrict__ P /*[3][4]*/, float x, float y, float z, int H, int W,
                                              float& gx, float& gy) {
    const float X = P[0] * x + P[1] * y + P[2] * z + P[3];
    const float Y = P[4] * x + P[5] * y + P[6] * z + P[7];
    const float Z = fmaxf(P[8] * x + P[9] * y + P[10] * z + P[11], 1e-3f);
    gx = 2.f * (X / Z) / (float)(W - 1) - 1.f;
    gy = 2.f * (Y / Z) / (float)(H - 1) - 1.f;
    if (gx > 1.f || gx < -1.f) gx = 2.f;
    if (gy > 1.f || gy < -1.f) gy = 2.f;
}

// geometry validity of a point: |p| < 1 on all axes and trilinear (zeros, align_corners) mask sample > 0
__device__ __forceinline__ bool geo_valid(const float* __restrict__ maskvol, int D, float x, float y, float z) {
    if (!(fabsf(x) < 1.f && fabsf(y) < 1.f && fabsf(z) < 1.f)) return false;
    const Axis2 ax = axis_taps_zeros(x, D), ay = axis_taps_zeros(y, D), az = axis_taps_zeros(z, D);
    float m = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // all eight taps are requested together (indices are clamped, a zero weight adds an exact zero): no branch per tap
                m += ax.w[a] * ay.w[b] * az.w[c] * maskvol[((size_t)ax.i[a] * D + ay.i[b]) * D + az.i[c]];
            }
    return m > 0.f;
}

// number of source views whose projection of a point is valid (and the point geometrically valid): all points,
// cheap -- feeds the per-ray colour mask (rendering_network.py:124-128)
__global__ __launch_bounds__(256) void k_view_count(const float* __restrict__ pts, long long n, const float* __restrict__ maskvol,
                                                    int D, const float* __restrict__ proj, int V, int H, int W,
                                                    uint8_t* __restrict__ out, const float* __restrict__ skip /*or null*/) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (skip && skip[i] > 0.f) return;                       // the colour kernel that evaluates this point writes the same count (out_nviews)
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    int c = 0;
    if (geo_valid(maskvol, D, x, y, z)) {
        for (int v = 0; v < V; ++v) {
            float gx, gy;
            project_point(proj + 12 * v, x, y, z, H, W, gx, gy);
            c += (fabsf(gx) < 1.f && fabsf(gy) < 1.f) ? 1 : 0;
        }
    }
    out[i] = (uint8_t)c;
}

// [V,56,H,W] features + [V,3,H,W] colours -> [V,H,W,64] (rgb | feat | 0)
__global__ __launch_bounds__(256) void k_pack_cmaps(const float* __restrict__ feat, const float* __restrict__ col, int HW,
                                                    float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int v = blockIdx.y, p0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i / 64, p = i % 64;
        float t = 0.f;
        if (p0 + p < HW) {
            if (c < 3) t = col[((size_t)v * 3 + c) * HW + p0 + p];
            else if (c < 59) t = feat[((size_t)v * 56 + (c - 3)) * HW + p0 + p];
        }
        tile[c][p] = t;
    }
    __syncthreads();
    float* dst = out + (size_t)v * HW * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int p = i / 64, c = i % 64;
        if (p0 + p < HW) dst[(size_t)(p0 + p) * 64 + c] = tile[c][p];
    }
}


// proj[v] = K[v] (3x3) @ w2c[v][:3, :] (3x4) as fp32 FMA chains in k order; cam_pos[v] = inverse(w2c[v])[:3, 3] by cofactors in fp64 (general 4x4,
// no assumption that w2c is rigid), rounded once.  One thread per view.
__global__ void k_camera_terms(const float* __restrict__ K, const float* __restrict__ w2c, int V, float* __restrict__ proj, float* __restrict__ cam) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float* k = K + 9 * v;
    const float* m = w2c + 16 * v;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = k[3 * i] * m[j];
            acc = fmaf(k[3 * i + 1], m[4 + j], acc);
            acc = fmaf(k[3 * i + 2], m[8 + j], acc);
            proj[12 * v + 4 * i + j] = acc;
        }
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = (double)m[i];
    // 2x2 sub-determinants of the two lower / upper row pairs
    const double s0 = a[0] * a[5] - a[1] * a[4], s1 = a[0] * a[6] - a[2] * a[4], s2 = a[0] * a[7] - a[3] * a[4];
    const double s3 = a[1] * a[6] - a[2] * a[5], s4 = a[1] * a[7] - a[3] * a[5], s5 = a[2] * a[7] - a[3] * a[6];
    const double c5 = a[10] * a[15] - a[11] * a[14], c4 = a[9] * a[15] - a[11] * a[13], c3 = a[9] * a[14] - a[10] * a[13];
    const double c2 = a[8] * a[15] - a[11] * a[12], c1 = a[8] * a[14] - a[10] * a[12], c0 = a[8] * a[13] - a[9] * a[12];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double id = 1.0 / det;
    // fourth column of the inverse: inv[0][3], inv[1][3], inv[2][3]
    cam[3 * v + 0] = (float)((-a[9] * s5 + a[10] * s4 - a[11] * s3) * id);
    cam[3 * v + 1] = (float)((a[8] * s5 - a[10] * s2 + a[11] * s1) * id);
    cam[3 * v + 2] = (float)((-a[8] * s4 + a[9] * s2 - a[11] * s0) * id);
}

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_pack_color_maps(const float* feat_nchw, const float* color_nchw, int V, int H, int W, float* out_nhwc64, void* stream) {
    O2345_REQUIRE(feat_nchw && color_nchw && out_nhwc64, "pack_color_maps: null pointer");
    hipLaunchKernelGGL(k_pack_cmaps, dim3(cdiv((long long)H * W, 64), V), dim3(256), 0, (hipStream_t)stream, feat_nchw, color_nchw, H * W, out_nhwc64);
    return check_launch("pack_color_maps");
}

int o2345_view_count_unlisted(const float* pts, long long n, const float* skip_if_positive, const float* maskvol, int D, const float* proj, int V,
                              int H, int W, uint8_t* out, void* stream) {
    O2345_REQUIRE(pts && maskvol && proj && out, "view_count: null pointer");
    O2345_REQUIRE(V >= 1 && V <= 255, "view_count: V must be in [1,255] (counts are stored as uint8; got %d)", V);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_view_count, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts, n, maskvol, D, proj, V, H, W, out, skip_if_positive);
    return check_launch("view_count");
}

int o2345_view_count(const float* pts, long long n, const float* maskvol, int D, const float* proj, int V, int H, int W,
                     uint8_t* out, void* stream) {
    return o2345_view_count_unlisted(pts, n, nullptr, maskvol, D, proj, V, H, W, out, stream);
}

int o2345_camera_terms(const float* intrinsics, const float* w2cs, int V, float* proj_out, float* cam_pos_out, void* stream) {
    O2345_REQUIRE(intrinsics && w2cs && proj_out && cam_pos_out && V >= 1, "camera_terms: bad arguments");
    hipLaunchKernelGGL(k_camera_terms, dim3(cdiv(V, 64)), dim3(64), 0, (hipStream_t)stream, intrinsics, w2cs, V, proj_out, cam_pos_out);
    return check_launch("camera_terms");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_color_maps() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_pack_cmaps));
}
}  // namespace o2345
