// Image-based colour blending: Projector.compute / compute_view_independent
// (models/projector.py:96-228,231-425; render_utils.py:54-120; ops/back_project.py:89-129) fused with
// GeneralRenderingNetwork.forward (models/rendering_network.py:75-129).
//
// The reference materialises [R,S,V,193] inputs and ~12 GEMM outputs (1.6 GB per 512-ray chunk at V=32).  Here a
// group of G lanes (G = pow2 >= V) owns one point, one lane per source view: the lane projects the point into its
// view, gathers its 59-channel colour+feature tap from a channel-last map [V,H,W,64] (colour first, 256-byte
// pixels), and runs the per-(point,view) MLPs with wave-uniform weights (scalar loads); the reductions over views
// (min, weighted mean/variance, softmax) are xor-shuffles inside the group.  Per-lane vectors that must be
// indexed at run time live in an LDS column (stride = block size: conflict-free, no barriers needed).
#include "common.h"
#include "geom_math.h"

namespace o2345 {

// ---- weight blob offsets (floats); matrices are [in][out] row-major unless noted (must match weights.py) ----------
constexpr int CW_S = 0;
constexpr int CW_RD0_W = 4;                       // [4][16]
constexpr int CW_RD0_B = CW_RD0_W + 64;           // [16]
constexpr int CW_RD1_WT = CW_RD0_B + 16;          // [59][16]  (OUT-major)
constexpr int CW_RD1_B = CW_RD1_WT + 59 * 16;     // [59] (+1 pad)
constexpr int CW_BASE0_W = CW_RD1_B + 60;         // [193][64]  rows: geo16 | mean59 | var59 | feat59
constexpr int CW_BASE0_B = CW_BASE0_W + 193 * 64;
constexpr int CW_BASE1_W = CW_BASE0_B + 64;       // [64][32]
constexpr int CW_BASE1_B = CW_BASE1_W + 64 * 32;
constexpr int CW_VIS0_W = CW_BASE1_B + 32;        // [32][32]
constexpr int CW_VIS0_B = CW_VIS0_W + 1024;
constexpr int CW_VIS1_W = CW_VIS0_B + 32;         // [32][36] (33 used, padded to 36)
constexpr int CW_VIS1_B = CW_VIS1_W + 32 * 36;    // [36]
constexpr int CW_VIS20_W = CW_VIS1_B + 36;        // [32][32]
constexpr int CW_VIS20_B = CW_VIS20_W + 1024;
constexpr int CW_VIS21_W = CW_VIS20_B + 32;       // [32][4] (1 used)
constexpr int CW_VIS21_B = CW_VIS21_W + 128;      // [4]
constexpr int CW_RGB0_W = CW_VIS21_B + 4;         // [37][16]
constexpr int CW_RGB0_B = CW_RGB0_W + 37 * 16;
constexpr int CW_RGB1_W = CW_RGB0_B + 16;         // [16][8]
constexpr int CW_RGB1_B = CW_RGB1_W + 128;
constexpr int CW_RGB2_W = CW_RGB1_B + 8;          // [8][4] (1 used)
constexpr int CW_RGB2_B = CW_RGB2_W + 32;         // [4]
constexpr int CW_TOTAL = CW_RGB2_B + 4;

struct ColorArgs {
    const float* W;          // CW_TOTAL floats
    const float* vol_cl;     // [D,D,D,16]
    const float* maskvol;    // [D^3]
    int D;
    const float* cmaps;      // [V,H,W,64]: rgb(3) | features(56) | pad(5)
    const float* proj;       // [V,3,4] = K @ w2c[:3]  (render_utils.py:106)
    const float* cam_pos;    // [V,3] camera centres (inverse(w2c)[:3,3])
    int V, H, W_img;
    const float* pts;        // [P,3]
    const int* index;        // optional list of point slots
    const int* n_dev;        // optional device count
    long long n;
    const float* query_cam;  // [3] (Projector.compute) or null
    const float* normals;    // [P,3] un-normalised SDF gradients (compute_view_independent) or null
    float* out_rgb;          // [P,3]
    uint8_t* out_nviews;     // [P] number of valid views, or null
};

// ELU / sigmoid are evaluated ~300 times per (point, view): libm's expm1f (~180 VALU ops) would cost 3x the MLP's FMAs.
// expm1(x), x <= 0: degree-7 Taylor polynomial for x > -0.35 (truncation < 3e-9), hardware exp2 minus one below that
// (absolute error ~1e-7 on a result of magnitude >= 0.29).
__device__ __forceinline__ float elu1(float x) {
    if (x > 0.f) return x;
    if (x > -0.35f) {
        float p = 1.f / 5040.f;
        p = fmaf(p, x, 1.f / 720.f); p = fmaf(p, x, 1.f / 120.f); p = fmaf(p, x, 1.f / 24.f);
        p = fmaf(p, x, 1.f / 6.f); p = fmaf(p, x, 0.5f); p = fmaf(p, x, 1.f);
        return p * x;
    }
    return __expf(x) - 1.f;
}
__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.f + __expf(-x)); }

// acc[o] += sum_i xcol[i] * W[i*LD + o]; xcol[i] = xbuf[i*256 + tid]
template <int IN, int OUT, int LD>
__device__ __forceinline__ void dense_lds(const float* __restrict__ W, const float* xbuf, int tid, float (&acc)[OUT]) {
    for (int i = 0; i < IN; ++i) {
        const float x = xbuf[i * 256 + tid];
        const float* w = W + i * LD;
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = fmaf(x, w[o], acc[o]);
    }
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
template <int G>
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

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

template <int G>
__global__ __launch_bounds__(256) void k_color_points(ColorArgs a, const float* __restrict__ Wt) {
    extern __shared__ __attribute__((aligned(16))) float xbuf[];      // [64][256] columns + [256/G][64] shared rows
    const int tid = threadIdx.x;
    const int v = tid % G;
    constexpr int PPB = 256 / G;
    const long long n = a.n_dev ? (long long)*a.n_dev : a.n;
    for (long long base = (long long)blockIdx.x * PPB; base < n; base += (long long)gridDim.x * PPB) {
        const long long i = base + tid / G;
        const bool live = i < n;
        const long long slot = live ? (a.index ? (long long)a.index[i] : i) : 0;
        const float px = live ? a.pts[3 * slot] : 0.f, py = live ? a.pts[3 * slot + 1] : 0.f, pz = live ? a.pts[3 * slot + 2] : 0.f;
        const bool view_ok = v < a.V;
        const int vv = view_ok ? v : 0;
        // ---- geometry feature (trilinear, zeros, align_corners=True) and validity ----------------------------------
        float geo[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) geo[c] = 0.f;
        {
            const Axis2 ax = axis_taps_zeros(px, a.D), ay = axis_taps_zeros(py, a.D), az = axis_taps_zeros(pz, a.D);
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int ic = 0; ic < 2; ++ic) {
                        const float w = ax.w[ia] * ay.w[ib] * az.w[ic];
                        if (w != 0.f) {
                            const float4* p4 = reinterpret_cast<const float4*>(
                                a.vol_cl + (((size_t)ax.i[ia] * a.D + ay.i[ib]) * a.D + az.i[ic]) * 16);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 t = p4[q];
                                geo[4 * q] += t.x * w; geo[4 * q + 1] += t.y * w; geo[4 * q + 2] += t.z * w; geo[4 * q + 3] += t.w * w;
                            }
                        }
                    }
        }
        const bool gvalid = geo_valid(a.maskvol, a.D, px, py, pz);
        // ---- this lane's view: projection, 59-channel tap -> LDS column ------------------------------------------------
        float gx, gy;
        project_point(a.proj + 12 * vv, px, py, pz, a.H, a.W_img, gx, gy);
        const bool pvalid = fabsf(gx) < 1.f && fabsf(gy) < 1.f;
        const float m = (view_ok && gvalid && pvalid) ? 1.f : 0.f;
        float rgb_in[3];
        {
            const Taps2D tp = bilinear_taps(gx, gy, a.H, a.W_img);
            const float4* img = reinterpret_cast<const float4*>(a.cmaps + (size_t)vv * a.H * a.W_img * 64);
#pragma unroll 1
            for (int q = 0; q < 15; ++q) {
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tp.w[k] != 0.f) {
                        const float4 t = img[(size_t)tp.idx[k] * 16 + q];
                        f.x += t.x * tp.w[k]; f.y += t.y * tp.w[k]; f.z += t.z * tp.w[k]; f.w += t.w * tp.w[k];
                    }
                if (q == 0) { rgb_in[0] = f.x; rgb_in[1] = f.y; rgb_in[2] = f.z; }
                xbuf[(4 * q + 0) * 256 + tid] = f.x; xbuf[(4 * q + 1) * 256 + tid] = f.y;
                xbuf[(4 * q + 2) * 256 + tid] = f.z; xbuf[(4 * q + 3) * 256 + tid] = f.w;
            }
        }
        // ---- ray direction difference (projector.py:15-62) -------------------------------------------------------------
        float rd[4];
        {
            float qx, qy, qz;
            if (a.normals) {           // safe_l2_normalize(gradient): g / max(|g|, 1e-6)
                const float nx = a.normals[3 * slot], ny = a.normals[3 * slot + 1], nz = a.normals[3 * slot + 2];
                const float nn = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f);
                qx = nx / nn; qy = ny / nn; qz = nz / nn;
            } else {
                const float tx = a.query_cam[0] - px, ty = a.query_cam[1] - py, tz = a.query_cam[2] - pz;
                const float tn = sqrtf(tx * tx + ty * ty + tz * tz) + 1e-6f;
                qx = tx / tn; qy = ty / tn; qz = tz / tn;
            }
            const float sx = a.cam_pos[3 * vv] - px, sy = a.cam_pos[3 * vv + 1] - py, sz = a.cam_pos[3 * vv + 2] - pz;
            const float sn = sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f;
            const float ux = sx / sn, uy = sy / sn, uz = sz / sn;
            const float dx = qx - ux, dy = qy - uy, dz = qz - uz;
            const float dn = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
            rd[0] = dx / dn; rd[1] = dy / dn; rd[2] = dz / dn;
            rd[3] = qx * ux + qy * uy + qz * uz;
        }
        // ---- ray_dir_fc: 4 -> 16 (ELU) -> 59 (ELU), added to the sampled features -------------------------------------
        float d16[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float t = Wt[CW_RD0_B + o];
#pragma unroll
            for (int k = 0; k < 4; ++k) t = fmaf(rd[k], Wt[CW_RD0_W + k * 16 + o], t);
            d16[o] = elu1(t);
        }
        for (int c = 0; c < 59; ++c) {
            float t = Wt[CW_RD1_B + c];
            const float* w = Wt + CW_RD1_WT + c * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) t = fmaf(d16[k], w[k], t);
            xbuf[c * 256 + tid] += elu1(t);
        }
        // ---- anti-alias pooling weights over views ------------------------------------------------------------------------
        const float e = __expf(fabsf(Wt[CW_S]) * (rd[3] - 1.f));
        const float emin = group_min<G>(view_ok ? e : INFINITY);
        float wgt = (e - emin) * m;
        wgt = wgt / (group_sum<G>(wgt) + 1e-8f);
        // ---- base_fc layer 1: [geo | mean | var | feat] (193) -> 64 ---------------------------------------------------
        // The geo | mean | var rows (134 of the 193 inputs) are the same for all views of a point: the G lanes of the group
        // split the 64 outputs of that part (OPL each, per-lane weight slices through the vector L1), exchange the result
        // through LDS, and only the per-view feature rows use the wave-uniform (scalar-loaded) weights.
        constexpr int OPL = 64 / G;
        float acc[64], sacc[OPL];
#pragma unroll
        for (int o = 0; o < 64; ++o) acc[o] = Wt[CW_BASE0_B + o];
#pragma unroll
        for (int j = 0; j < OPL; ++j) sacc[j] = 0.f;
        const float* Wv = a.W + CW_BASE0_W + v * OPL;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
#pragma unroll
            for (int j = 0; j < OPL; ++j) sacc[j] = fmaf(geo[c], Wv[c * 64 + j], sacc[j]);
        }
        for (int c = 0; c < 59; ++c) {
            const float x = xbuf[c * 256 + tid];
            const float mean = group_sum<G>(x * wgt);
            const float dd = x - mean;
            const float var = group_sum<G>(wgt * dd * dd);
            const float* w2 = Wt + CW_BASE0_W + (134 + c) * 64;
#pragma unroll
            for (int j = 0; j < OPL; ++j) sacc[j] = fmaf(var, Wv[(75 + c) * 64 + j], fmaf(mean, Wv[(16 + c) * 64 + j], sacc[j]));
#pragma unroll
            for (int o = 0; o < 64; ++o) acc[o] = fmaf(x, w2[o], acc[o]);
        }
        {
            float* sbuf = xbuf + 64 * 256 + (tid / G) * 64;     // this point's 64 shared pre-activations
#pragma unroll
            for (int j = 0; j < OPL; ++j) sbuf[v * OPL + j] = sacc[j];
            __builtin_amdgcn_wave_barrier();                      // the G lanes of a group are in one wave
#pragma unroll
            for (int o = 0; o < 64; ++o) acc[o] += sbuf[o];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int o = 0; o < 64; ++o) xbuf[o * 256 + tid] = elu1(acc[o]);
        // ---- base_fc layer 2: 64 -> 32 -------------------------------------------------------------------------------------
        float x32[32];
#pragma unroll
        for (int o = 0; o < 32; ++o) x32[o] = Wt[CW_BASE1_B + o];
        dense_lds<64, 32, 32>(Wt + CW_BASE1_W, xbuf, tid, x32);
#pragma unroll
        for (int o = 0; o < 32; ++o) { x32[o] = elu1(x32[o]); xbuf[o * 256 + tid] = x32[o] * wgt; }
        // ---- vis_fc: 32 -> 32 -> 33 ------------------------------------------------------------------------------------------
        float t32[32];
#pragma unroll
        for (int o = 0; o < 32; ++o) t32[o] = Wt[CW_VIS0_B + o];
        dense_lds<32, 32, 32>(Wt + CW_VIS0_W, xbuf, tid, t32);
#pragma unroll
        for (int o = 0; o < 32; ++o) xbuf[o * 256 + tid] = elu1(t32[o]);
        float v36[36];
#pragma unroll
        for (int o = 0; o < 36; ++o) v36[o] = Wt[CW_VIS1_B + o];
        dense_lds<32, 36, 36>(Wt + CW_VIS1_W, xbuf, tid, v36);
        float vis = sigm(elu1(v36[32])) * m;
#pragma unroll
        for (int o = 0; o < 32; ++o) { x32[o] += elu1(v36[o]); xbuf[o * 256 + tid] = x32[o] * vis; }
        // ---- vis_fc2: 32 -> 32 -> 1 (sigmoid) ------------------------------------------------------------------------------
#pragma unroll
        for (int o = 0; o < 32; ++o) t32[o] = Wt[CW_VIS20_B + o];
        dense_lds<32, 32, 32>(Wt + CW_VIS20_W, xbuf, tid, t32);
#pragma unroll
        for (int o = 0; o < 32; ++o) xbuf[o * 256 + tid] = elu1(t32[o]);
        float v4[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) v4[o] = Wt[CW_VIS21_B + o];
        dense_lds<32, 4, 4>(Wt + CW_VIS21_W, xbuf, tid, v4);
        vis = sigm(v4[0]) * m;
        // ---- rgb_fc: [x(32) | vis | ray_diff(4)] -> 16 -> 8 -> 1 ------------------------------------------------------------
#pragma unroll
        for (int o = 0; o < 32; ++o) xbuf[o * 256 + tid] = x32[o];
        xbuf[32 * 256 + tid] = vis;
#pragma unroll
        for (int k = 0; k < 4; ++k) xbuf[(33 + k) * 256 + tid] = rd[k];
        float r16[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) r16[o] = Wt[CW_RGB0_B + o];
        dense_lds<37, 16, 16>(Wt + CW_RGB0_W, xbuf, tid, r16);
#pragma unroll
        for (int o = 0; o < 16; ++o) xbuf[o * 256 + tid] = elu1(r16[o]);
        float r8[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) r8[o] = Wt[CW_RGB1_B + o];
        dense_lds<16, 8, 8>(Wt + CW_RGB1_W, xbuf, tid, r8);
#pragma unroll
        for (int o = 0; o < 8; ++o) xbuf[o * 256 + tid] = elu1(r8[o]);
#pragma unroll
        for (int o = 0; o < 4; ++o) v4[o] = Wt[CW_RGB2_B + o];
        dense_lds<8, 4, 4>(Wt + CW_RGB2_W, xbuf, tid, v4);
        // ---- masked softmax over views, blended colour ------------------------------------------------------------------------
        float score = (m == 0.f) ? -1e9f : v4[0];
        if (!view_ok) score = -INFINITY;
        const float smax = group_max<G>(score);
        const float ex = view_ok ? __expf(score - smax) : 0.f;
        const float den = group_sum<G>(ex);
        const float bw = ex / den;
        const float c0 = group_sum<G>(rgb_in[0] * bw), c1 = group_sum<G>(rgb_in[1] * bw), c2 = group_sum<G>(rgb_in[2] * bw);
        const float nv = group_sum<G>(m);
        if (live && v == 0) {
            a.out_rgb[3 * slot] = c0; a.out_rgb[3 * slot + 1] = c1; a.out_rgb[3 * slot + 2] = c2;
            if (a.out_nviews) a.out_nviews[slot] = (uint8_t)(nv + 0.5f);
        }
    }
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

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_color_blob_floats(void) { return CW_TOTAL; }

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

int o2345_color_points(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                       const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                       const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                       const float* normals, float* out_rgb, uint8_t* out_nviews, void* stream) {
    O2345_REQUIRE(blob && vol_cl && maskvol && cmaps && proj && cam_pos && pts && out_rgb, "color_points: null pointer");
    O2345_REQUIRE((query_cam != nullptr) != (normals != nullptr), "color_points: give exactly one of query_cam / normals");
    O2345_REQUIRE(V >= 1 && V <= 64, "color_points: V must be in [1,64] (got %d)", V);
    if (n <= 0 && !n_dev) return 0;
    ColorArgs a{blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews};
    int G = 1;
    while (G < V) G <<= 1;
    if (G < 4) G = 4;
    const long long ppb = 256 / G;
    long long want = n_dev ? 2048 : (n + ppb - 1) / ppb;
    if (want > 4096) want = 4096;
    const size_t lds = (64 * 256 + (256 / G) * 64) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define O2345_COLOR_CASE(GG)                                                                                          \
    if (G == GG) {                                                                                                    \
        O2345_HIP(hipFuncSetAttribute((const void*)k_color_points<GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_color_points<GG>, dim3((unsigned)want), dim3(256), lds, s, a, blob);                           \
    }
    O2345_COLOR_CASE(4) O2345_COLOR_CASE(8) O2345_COLOR_CASE(16) O2345_COLOR_CASE(32) O2345_COLOR_CASE(64)
    return check_launch("color_points");
}

}  // extern "C"
