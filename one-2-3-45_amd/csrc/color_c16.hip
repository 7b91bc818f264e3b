// Image-based colour blending, 16-column form: the same function as csrc/color_pts.hip / csrc/color_mfma.hip
// (Projector.compute / compute_view_independent + GeneralRenderingNetwork.forward, models/projector.py:96-425,
// models/rendering_network.py:75-129), split-f16 numerical form, for up to 8 source views.
//
// csrc/color_pts.hip (32 points per wave on v_mfma_f32_32x32x16_f16) has to gather every source pixel and evaluate ray_dir_fc TWICE: the
// weighted mean / variance over the views must be complete before any view's network can start, and the 59 floats x V of a point do not fit
// in the registers of the point's two lanes (32 x V per lane).  Here a wave owns 16 POINTS on v_mfma_f32_16x16x32_f16: column n = lane & 15,
// the four lane groups g = lane >> 4 supply 8 of a step's 32 k values each and receive output rows 4g .. 4g+3 of every 16-row block, and a
// lane owns 16 of its point's 64 pixel floats (16g .. 16g+15).  A lane then holds 16 floats per (point, view): the per-view network inputs of
// 8 views are 128 registers (kept as the split-f16 operands base_fc.0 consumes), and the second pass neither gathers nor runs ray_dir_fc again.
// As in color_pts.hip the matrix result layout is the next layer's k enumeration (weights pre-permuted on the host,
// weights.pack_color_c16_blob; lane-by-lane emulation tests/weights_emulators.py:emulate_color_c16), views that see none of the tile's points
// are skipped (bit-identical, see color_pts.hip), the network runs in the log2(e)-scaled domain, the softmax over views is online.
//   pass 0   geometry feature (4 channels per lane), validity, query direction, min over views of the pooling exponent
//   pass A   per view: gather (4 taps x 64 bytes per lane), ray_dir_fc, Welford update of the weighted mean / M2, operands cached
//   shared   view-independent rows of base_fc.0 once per point (5 k-steps), kept in a lane-private LDS slot
//   pass B   per view: base_fc, vis_fc, vis_fc2, rgb_fc on the cached operands; single-output layers are 16-row blocks whose rows are
//            all the same output row, so the value lands in every lane without a cross-lane step
#include "color_net.h"

namespace o2345 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// blob layout (floats) -- must match weights.C16_SEGS / C16_LAYOUT
constexpr int C16_A_RD0 = 0, C16_A_RD1 = C16_A_RD0 + 1 * 1 * 512, C16_A_B0 = C16_A_RD1 + 4 * 1 * 512, C16_A_B1 = C16_A_B0 + 4 * 2 * 512,
              C16_A_V0 = C16_A_B1 + 2 * 2 * 512, C16_A_V1 = C16_A_V0 + 2 * 1 * 512, C16_A_V20 = C16_A_V1 + 3 * 1 * 512,
              C16_A_V21 = C16_A_V20 + 2 * 1 * 512, C16_A_R0 = C16_A_V21 + 1 * 1 * 512, C16_A_R1 = C16_A_R0 + 1 * 2 * 512,
              C16_A_R2 = C16_A_R1 + 1 * 1 * 512, C16_A_S = C16_A_R2 + 1 * 1 * 512, C16_A_END = C16_A_S + 4 * 5 * 512;
constexpr int C16_B_RD0 = C16_A_END, C16_B_RD1 = C16_B_RD0 + 16, C16_B_B0 = C16_B_RD1 + 64, C16_B_B1 = C16_B_B0 + 64, C16_B_V0 = C16_B_B1 + 32,
              C16_B_V1 = C16_B_V0 + 32, C16_B_V20 = C16_B_V1 + 48, C16_B_V21 = C16_B_V20 + 32, C16_B_R0 = C16_B_V21 + 16, C16_B_R1 = C16_B_R0 + 16,
              C16_B_R2 = C16_B_R1 + 16, C16_SCALAR = C16_B_R2 + 16, C16_TOTAL = C16_SCALAR + 4;

struct Split8h { h16x8 hi, lo; };
__device__ __forceinline__ Split8h split8f(const float (&x)[8], float m1) {
    union { h16x8 v8; h16x2 v2[4]; hh16x2 w2[4]; } hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi.v2[i] = __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]);
        lo.v2[i] = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hi.w2[i][0], m1, x[2 * i]), __builtin_fmaf((float)hi.w2[i][1], m1, x[2 * i + 1]));
    }
    return {hi.v8, lo.v8};
}

// one k-step of 32 for NB output blocks: acc[b] += A[b][step] (x) B, three matrix instructions per block (lo*hi + hi*lo + hi*hi), term-major over
// at most two blocks at a time (16 operand registers in flight)
template <int NB, int NS, int B0, int B1>
__device__ __forceinline__ void c16_step_part(f32x4 (&acc)[NB], const float4* A /* segment + lane */, int step, const Split8h& b) {
    h16x8 ahi[B1 - B0], alo[B1 - B0];
#pragma unroll
    for (int nb = B0; nb < B1; ++nb) {
        ahi[nb - B0] = __builtin_bit_cast(h16x8, A[((nb * NS + step) * 2 + 0) * 64]);
        alo[nb - B0] = __builtin_bit_cast(h16x8, A[((nb * NS + step) * 2 + 1) * 64]);
    }
#pragma unroll
    for (int nb = B0; nb < B1; ++nb) acc[nb] = MFMA16_F16(alo[nb - B0], b.hi, acc[nb]);
#pragma unroll
    for (int nb = B0; nb < B1; ++nb) acc[nb] = MFMA16_F16(ahi[nb - B0], b.lo, acc[nb]);
#pragma unroll
    for (int nb = B0; nb < B1; ++nb) acc[nb] = MFMA16_F16(ahi[nb - B0], b.hi, acc[nb]);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NB, int NS>
__device__ __forceinline__ void c16_step(f32x4 (&acc)[NB], const float4* A, int step, const Split8h& b) {
    if constexpr (NB > 2) {
        c16_step_part<NB, NS, 0, 2>(acc, A, step, b);
        c16_step_part<NB, NS, 2, NB>(acc, A, step, b);
    } else {
        c16_step_part<NB, NS, 0, NB>(acc, A, step, b);
    }
}
template <int NB>
__device__ __forceinline__ void c16_bias(f32x4 (&acc)[NB], const float* bias, int g) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float4 t = *reinterpret_cast<const float4*>(bias + (nb * 4 + g) * 4);
        acc[nb] = f32x4{t.x, t.y, t.z, t.w};
    }
}
__device__ __forceinline__ void celu4(f32x4& y) {
    const f32x2 a = celu2(y[0], y[1]), b = celu2(y[2], y[3]);
    y = f32x4{a[0], a[1], b[0], b[1]};
}
// the 8 operands of a k-step from two consecutive 16-row blocks (rows 4g .. 4g+3 of each), optionally scaled
__device__ __forceinline__ Split8h chain8(const f32x4& b0, const f32x4& b1, float scale, float m1) {
    const float x[8] = {b0[0] * scale, b0[1] * scale, b0[2] * scale, b0[3] * scale, b1[0] * scale, b1[1] * scale, b1[2] * scale, b1[3] * scale};
    return split8f(x, m1);
}
__device__ __forceinline__ Split8h one_block8(const f32x4& b0, float m1) {
    const float x[8] = {b0[0], b0[1], b0[2], b0[3], 0.f, 0.f, 0.f, 0.f};
    return split8f(x, m1);
}

constexpr int C16_THREADS = 512, C16_NVC = 8;       // 2 waves per SIMD; cached views
__global__ __launch_bounds__(C16_THREADS) void k_color_c16(ColorMArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < C16_TOTAL; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(a.blob + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, n16 = lane & 15, g = lane >> 4;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float4* sh_slot = reinterpret_cast<float4*>(lds + C16_TOTAL + wave * 1024) + lane;      // lane-private: 4 x float4 at stride 64
    const long long n = a.n_dev ? (long long)*a.n_dev : a.n;
    const float m1 = opaque_minus_one();
    const float s_abs = fabsf(lds[C16_SCALAR]) * LOG2E;
    const int V = a.V;
    const bool skip_views = !(a.sched & 4);
    const int base_prio = 0;
    const float4* AL = reinterpret_cast<const float4*>(lds) + lane;                            // A segments: float4 index = floats / 4
    unsigned st_a = 0, st_b = 0, st_t = 0, st_full = 0;
    const long long ntiles = (n + 15) / 16;
    for (long long tile = (long long)blockIdx.x * nwave + wave; tile < ntiles; tile += (long long)gridDim.x * nwave) {
        const long long i = tile * 16 + n16;
        const bool live = i < n;
        const long long slot = live ? (a.index ? (long long)a.index[i] : i) : 0;
        const float px = live ? a.pts[3 * slot] : 0.f, py = live ? a.pts[3 * slot + 1] : 0.f, pz = live ? a.pts[3 * slot + 2] : 0.f;
        // ---- pass 0: geometry feature (this lane's 4 channels), validity, query direction --------------------------------------------------
        float bs[40];                               // per-lane operands of the shared rows: geo (4) | mean (16) | var (16) | pad (4)
#pragma unroll
        for (int c = 0; c < 40; ++c) bs[c] = 0.f;
        bool gvalid;
        {
            float msum = 0.f;
            const Axis2 ax = axis_taps_zeros(px, a.D), ay = axis_taps_zeros(py, a.D), az = axis_taps_zeros(pz, a.D);
#pragma unroll
            for (int tap = 0; tap < 8; ++tap) {
                const int ia = (tap >> 2) & 1, ib = (tap >> 1) & 1, ic = tap & 1;
                const float w = ax.w[ia] * ay.w[ib] * az.w[ic];
                if (w != 0.f) {
                    const size_t vox = ((size_t)ax.i[ia] * a.D + ay.i[ib]) * a.D + az.i[ic];
                    msum += w * a.maskvol[vox];
                    const float4 t = *(reinterpret_cast<const float4*>(a.vol_cl + vox * 16) + g);
                    bs[0] = fmaf(t.x, w, bs[0]); bs[1] = fmaf(t.y, w, bs[1]); bs[2] = fmaf(t.z, w, bs[2]); bs[3] = fmaf(t.w, w, bs[3]);
                }
            }
            gvalid = fabsf(px) < 1.f && fabsf(py) < 1.f && fabsf(pz) < 1.f && msum > 0.f;
        }
        float qx, qy, qz;
        if (a.normals) {
            const float nx = a.normals[3 * slot], ny = a.normals[3 * slot + 1], nz = a.normals[3 * slot + 2];
            const float rn = crcp(fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f));
            qx = nx * rn; qy = ny * rn; qz = nz * rn;
        } else {
            const float tx = a.query_cam[0] - px, ty = a.query_cam[1] - py, tz = a.query_cam[2] - pz;
            const float rn = crcp(sqrtf(tx * tx + ty * ty + tz * tz) + 1e-6f);
            qx = tx * rn; qy = ty * rn; qz = tz * rn;
        }
        float emin = INFINITY;
        for (int v = 0; v < V; ++v) {
            const float sx = a.cam_pos[3 * v] - px, sy = a.cam_pos[3 * v + 1] - py, sz = a.cam_pos[3 * v + 2] - pz;
            const float rsn = crcp(sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
            const float dot = qx * (sx * rsn) + qy * (sy * rsn) + qz * (sz * rsn);
            emin = fminf(emin, __builtin_amdgcn_exp2f(s_abs * (dot - 1.f)));
        }
        // ---- pass A ------------------------------------------------------------------------------------------------------------------------------
        Split8h cx[C16_NVC][2];                     // cached base_fc.0 operands of every view (static indices only: registers)
        float wsum = 0.f, nvis = 0.f;
        unsigned active = 0u;
        {
            float mean[16], m2[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) { mean[c] = 0.f; m2[c] = 0.f; }
#pragma unroll 1
            for (int v = 0; v < V; ++v) {
                const ViewGeom vg = view_geom(a, v, px, py, pz, qx, qy, qz, gvalid, s_abs);
                if (skip_views && __builtin_amdgcn_ballot_w64(vg.m != 0.f) == 0ull) continue;
                active |= 1u << v;
                ++st_a;
                // this lane's 16 pixel floats, bilinear, zero padding, scaled domain
                float xf[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) xf[c] = 0.f;
                if (a.sched & 2) set_wave_prio(3);
                {
                    const Taps2D tp = bilinear_taps(vg.gx, vg.gy, a.H, a.W_img);
                    const float4* img = reinterpret_cast<const float4*>(a.cmaps + (size_t)v * a.H * a.W_img * 64) + 4 * g;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (tp.w[k] != 0.f) {
                            const float4* px4 = img + (size_t)tp.idx[k] * 16;
                            const float wk = tp.w[k] * LOG2E;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 t = px4[q];
                                xf[4 * q] = fmaf(t.x, wk, xf[4 * q]); xf[4 * q + 1] = fmaf(t.y, wk, xf[4 * q + 1]);
                                xf[4 * q + 2] = fmaf(t.z, wk, xf[4 * q + 2]); xf[4 * q + 3] = fmaf(t.w, wk, xf[4 * q + 3]);
                            }
                        }
                }
                if (a.sched & 2) set_wave_prio(base_prio);
                // ray_dir_fc: 4 -> 16 -> 59, added to the lane's own pixel floats
                {
                    f32x4 d1[1];
                    c16_bias<1>(d1, lds + C16_B_RD0, g);
                    const float b0[8] = {g == 0 ? vg.rd[0] : 0.f, g == 0 ? vg.rd[1] : 0.f, g == 0 ? vg.rd[2] : 0.f, g == 0 ? vg.rd[3] : 0.f, 0.f, 0.f, 0.f, 0.f};
                    c16_step<1, 1>(d1, AL + C16_A_RD0 / 4, 0, split8f(b0, m1));
                    celu4(d1[0]);
                    f32x4 d2[4];
                    c16_bias<4>(d2, lds + C16_B_RD1, g);
                    c16_step<4, 1>(d2, AL + C16_A_RD1 / 4, 0, one_block8(d1[0], m1));
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        celu4(d2[b]);
                        xf[4 * b] += d2[b][0]; xf[4 * b + 1] += d2[b][1]; xf[4 * b + 2] += d2[b][2]; xf[4 * b + 3] += d2[b][3];
                    }
                }
                const float raw = (vg.e - emin) * vg.m;
                nvis += vg.m;
                wsum += raw;
                const float r0 = raw > 0.f ? raw * crcp(wsum) : 0.f;
                const float rq = raw > 0.f ? fmaf(fmaf(-wsum, r0, raw), crcp(wsum), r0) : 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float d = xf[c] - mean[c];
                    mean[c] = fmaf(rq, d, mean[c]);
                    m2[c] = fmaf(raw * d, xf[c] - mean[c], m2[c]);
                }
                const float x0[8] = {xf[0], xf[1], xf[2], xf[3], xf[4], xf[5], xf[6], xf[7]};
                const float x1[8] = {xf[8], xf[9], xf[10], xf[11], xf[12], xf[13], xf[14], xf[15]};
                const Split8h s0 = split8f(x0, m1), s1 = split8f(x1, m1);
                switch (v) {
                    case 0: cx[0][0] = s0; cx[0][1] = s1; break;
                    case 1: cx[1][0] = s0; cx[1][1] = s1; break;
                    case 2: cx[2][0] = s0; cx[2][1] = s1; break;
                    case 3: cx[3][0] = s0; cx[3][1] = s1; break;
                    case 4: cx[4][0] = s0; cx[4][1] = s1; break;
                    case 5: cx[5][0] = s0; cx[5][1] = s1; break;
                    case 6: cx[6][0] = s0; cx[6][1] = s1; break;
                    default: cx[7][0] = s0; cx[7][1] = s1; break;
                }
            }
            const float rden0 = crcp(wsum + 1e-8f);
            const float S = wsum * rden0, k2 = S * (1.f - S) * (1.f - S);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                bs[4 + c] = S * mean[c];
                bs[20 + c] = fmaf(k2 * mean[c], mean[c], m2[c] * rden0);
            }
        }
        const float rden = crcp(wsum + 1e-8f);
        // ---- view-independent rows of base_fc.0, once per point ---------------------------------------------------------------------------------
        {
            f32x4 sh[4];
            c16_bias<4>(sh, lds + C16_B_B0, g);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float x[8] = {bs[8 * s], bs[8 * s + 1], bs[8 * s + 2], bs[8 * s + 3], bs[8 * s + 4], bs[8 * s + 5], bs[8 * s + 6], bs[8 * s + 7]};
                c16_step<4, 5>(sh, AL + C16_A_S / 4, s, split8f(x, m1));
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) sh_slot[b * 64] = float4{sh[b][0], sh[b][1], sh[b][2], sh[b][3]};
        }
        // ---- pass B -------------------------------------------------------------------------------------------------------------------------------
        float smax = -INFINITY, ssum = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
        const bool skip_b = skip_views && __builtin_amdgcn_ballot_w64(live && nvis == 0.f) == 0ull;
        ++st_t;
        st_full += skip_b ? 0u : 1u;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const bool cached = (active >> v) & 1u;
            if (skip_b && !cached) continue;
            ++st_b;
            const ViewGeom vg = view_geom(a, v, px, py, pz, qx, qy, qz, gvalid, s_abs);
            const float m = vg.m;
            // colours of this view at the point (pixel floats 0..2, before the direction feature): lanes of group 0, re-gathered (16 bytes per tap)
            float rgb0 = 0.f, rgb1 = 0.f, rgb2 = 0.f;
            {
                const Taps2D tp = bilinear_taps(vg.gx, vg.gy, a.H, a.W_img);
                const float4* img = reinterpret_cast<const float4*>(a.cmaps + (size_t)v * a.H * a.W_img * 64);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tp.w[k] != 0.f && g == 0) {
                        const float4 t = img[(size_t)tp.idx[k] * 16];
                        const float wk = tp.w[k] * LOG2E;
                        rgb0 = fmaf(t.x, wk, rgb0); rgb1 = fmaf(t.y, wk, rgb1); rgb2 = fmaf(t.z, wk, rgb2);
                    }
            }
            Split8h s0, s1;
            if (cached) {
                switch (v) {
                    case 0: s0 = cx[0][0]; s1 = cx[0][1]; break;
                    case 1: s0 = cx[1][0]; s1 = cx[1][1]; break;
                    case 2: s0 = cx[2][0]; s1 = cx[2][1]; break;
                    case 3: s0 = cx[3][0]; s1 = cx[3][1]; break;
                    case 4: s0 = cx[4][0]; s1 = cx[4][1]; break;
                    case 5: s0 = cx[5][0]; s1 = cx[5][1]; break;
                    case 6: s0 = cx[6][0]; s1 = cx[6][1]; break;
                    default: s0 = cx[7][0]; s1 = cx[7][1]; break;
                }
            } else {
                // a view that saw none of the tile's points in pass A but must be blended (a point of the tile has NO visible view): m = 0 for every
                // lane, so the network's result is multiplied away (score = -1e9); any finite operands do
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                s0 = split8f(z, m1); s1 = s0;
            }
            const float wgt = (vg.e - emin) * m * rden;
            // ---- base_fc: (shared + per-view floats) -> 64 -> 32
            f32x4 x32[2];
            {
                f32x4 acc[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) { const float4 t = sh_slot[b * 64]; acc[b] = f32x4{t.x, t.y, t.z, t.w}; }
                c16_step<4, 2>(acc, AL + C16_A_B0 / 4, 0, s0);
                c16_step<4, 2>(acc, AL + C16_A_B0 / 4, 1, s1);
#pragma unroll
                for (int b = 0; b < 4; ++b) celu4(acc[b]);
                c16_bias<2>(x32, lds + C16_B_B1, g);
                c16_step<2, 2>(x32, AL + C16_A_B1 / 4, 0, chain8(acc[0], acc[1], 1.f, m1));
                c16_step<2, 2>(x32, AL + C16_A_B1 / 4, 1, chain8(acc[2], acc[3], 1.f, m1));
                celu4(x32[0]); celu4(x32[1]);
            }
            // ---- vis_fc
            float vis;
            {
                f32x4 t1[2];
                c16_bias<2>(t1, lds + C16_B_V0, g);
                c16_step<2, 1>(t1, AL + C16_A_V0 / 4, 0, chain8(x32[0], x32[1], wgt, m1));
                celu4(t1[0]); celu4(t1[1]);
                f32x4 t2[3];
                c16_bias<3>(t2, lds + C16_B_V1, g);
                c16_step<3, 1>(t2, AL + C16_A_V1 / 4, 0, chain8(t1[0], t1[1], 1.f, m1));
                celu4(t2[0]); celu4(t2[1]);
                x32[0] += t2[0]; x32[1] += t2[1];
                vis = csigm(celu(t2[2][0])) * m;
            }
            // ---- vis_fc2
            {
                f32x4 t1[2];
                c16_bias<2>(t1, lds + C16_B_V20, g);
                c16_step<2, 1>(t1, AL + C16_A_V20 / 4, 0, chain8(x32[0], x32[1], vis, m1));
                celu4(t1[0]); celu4(t1[1]);
                f32x4 t2[1];
                c16_bias<1>(t2, lds + C16_B_V21, g);
                c16_step<1, 1>(t2, AL + C16_A_V21 / 4, 0, chain8(t1[0], t1[1], 1.f, m1));
                vis = csigm(t2[0][0]) * m;
            }
            // ---- rgb_fc: [x | vis | ray_diff] (37) -> 16 -> 8 -> 1
            float score;
            {
                f32x4 r16[1];
                c16_bias<1>(r16, lds + C16_B_R0, g);
                c16_step<1, 2>(r16, AL + C16_A_R0 / 4, 0, chain8(x32[0], x32[1], 1.f, m1));
                const float ex[8] = {g == 0 ? vis : 0.f, g == 0 ? vg.rd[0] : 0.f, g == 0 ? vg.rd[1] : 0.f, g == 0 ? vg.rd[2] : 0.f, g == 0 ? vg.rd[3] : 0.f, 0.f, 0.f, 0.f};
                c16_step<1, 2>(r16, AL + C16_A_R0 / 4, 1, split8f(ex, m1));
                celu4(r16[0]);
                f32x4 r8[1];
                c16_bias<1>(r8, lds + C16_B_R1, g);
                c16_step<1, 1>(r8, AL + C16_A_R1 / 4, 0, one_block8(r16[0], m1));
                celu4(r8[0]);
                f32x4 sc[1];
                c16_bias<1>(sc, lds + C16_B_R2, g);
                c16_step<1, 1>(sc, AL + C16_A_R2 / 4, 0, one_block8(r8[0], m1));
                score = sc[0][0];
            }
            if (m == 0.f) score = -1e9f;
            const float nmax = fmaxf(smax, score);
            const float sc_old = __builtin_amdgcn_exp2f(smax - nmax), ex = __builtin_amdgcn_exp2f(score - nmax);
            ssum = fmaf(ssum, sc_old, ex);
            o0 = fmaf(o0, sc_old, ex * rgb0); o1 = fmaf(o1, sc_old, ex * rgb1); o2 = fmaf(o2, sc_old, ex * rgb2);
            smax = nmax;
        }
        if (live && g == 0) {
            const float rs = crcp(ssum) * LN2;
            a.out_rgb[3 * slot] = o0 * rs; a.out_rgb[3 * slot + 1] = o1 * rs; a.out_rgb[3 * slot + 2] = o2 * rs;
            if (a.out_nviews) a.out_nviews[slot] = (uint8_t)(nvis + 0.5f);
        }
    }
    if (a.stats && lane == 0) {
        atomicAdd(a.stats + 0, (unsigned long long)st_a); atomicAdd(a.stats + 1, (unsigned long long)st_b);
        atomicAdd(a.stats + 2, (unsigned long long)st_t); atomicAdd(a.stats + 3, (unsigned long long)st_full);
    }
}

unsigned long long* color_stats_buffer();

int color_c16_launch(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj, const float* cam_pos,
                     int V, int H, int W, const float* pts, const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                     const float* normals, float* out_rgb, uint8_t* out_nviews, void* stream) {
    O2345_REQUIRE(V >= 1 && V <= C16_NVC, "color_points_c16: 1..8 source views (got %d)", V);
    ColorMArgs a{blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews};
    a.sched = color_sched_mode();
    a.stats = color_stats_buffer();
    const int n_cu = cu_count();
    const long long per_block = (long long)(C16_THREADS / 64) * 16;
    const long long want = n_dev ? n_cu : (n + per_block - 1) / per_block;
    const unsigned grid = (unsigned)(want < n_cu ? want : n_cu);
    const size_t lds = (size_t)(C16_TOTAL + (C16_THREADS / 64) * 1024) * sizeof(float);
    O2345_HIP(hipFuncSetAttribute((const void*)k_color_c16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_color_c16, dim3(grid), dim3(C16_THREADS), lds, (hipStream_t)stream, a);
    return check_launch("color_points (16-column kernel)");
}

}  // namespace o2345

extern "C" {

int o2345_color_c16_blob_floats(void) { return o2345::C16_TOTAL; }

/* the 16-column colour kernel (csrc/color_c16.hip), blob from weights.pack_color_c16_blob; V <= 8 */
int o2345_color_points_c16(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj,
                           const float* cam_pos, int V, int H, int W, const float* pts, const int32_t* index, const int32_t* n_dev, long long n,
                           const float* query_cam, const float* normals, float* out_rgb, uint8_t* out_nviews, void* stream) {
    using namespace o2345;
    O2345_REQUIRE(blob && vol_cl && maskvol && cmaps && proj && cam_pos && pts && out_rgb, "color_points_c16: null pointer");
    O2345_REQUIRE((query_cam != nullptr) != (normals != nullptr), "color_points_c16: give exactly one of query_cam / normals");
    if (n <= 0 && !n_dev) return 0;
    return color_c16_launch(blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews, stream);
}

}  // extern "C"
