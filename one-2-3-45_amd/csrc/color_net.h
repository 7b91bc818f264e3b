// Device helpers shared by the two colour-network kernels (csrc/color_mfma.hip: columns = (point, view) pairs;
// csrc/color_pts.hip: columns = points, views looped): blob layout, matrix-step loops in both numerical forms, the scaled-domain
// ELU, DPP group reductions, projection.  See csrc/color_mfma.hip for the layer structure and weights.pack_color_mfma_blob /
// pack_color_x3_blob for the operand order.
#pragma once
#include "common.h"
#include "geom_math.h"

namespace o2345 {

using f32x16 = __attribute__((ext_vector_type(16))) float;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ---- blob layout (floats) -- must match weights.CM_SEGS / CM_BIAS -------------------------------------------------
constexpr int CM_A_RD0 = 0;                         // [1][2][64]
constexpr int CM_A_RD1 = CM_A_RD0 + 1 * 2 * 64;     // [2][8][64]
constexpr int CM_A_B0 = CM_A_RD1 + 2 * 8 * 64;      // [2][32][64]
constexpr int CM_A_B1 = CM_A_B0 + 2 * 32 * 64;      // [1][32][64]
constexpr int CM_A_V0 = CM_A_B1 + 32 * 64;          // [1][16][64]
constexpr int CM_A_V1 = CM_A_V0 + 16 * 64;          // [1][16][64]  (rows 0..31 of vis_fc.2; row 32 is a dot product)
constexpr int CM_A_V20 = CM_A_V1 + 16 * 64;         // [1][16][64]
constexpr int CM_A_R0 = CM_A_V20 + 16 * 64;         // [1][19][64]
constexpr int CM_A_R1 = CM_A_R0 + 19 * 64;          // [1][8][64]
constexpr int CM_BIAS0 = CM_A_R1 + 8 * 64;          // biases / per-lane vectors, [block][half][16] each
constexpr int CM_B_RD0 = CM_BIAS0, CM_B_RD1 = CM_B_RD0 + 32, CM_B_B0 = CM_B_RD1 + 64, CM_B_B1 = CM_B_B0 + 64,
              CM_B_V0 = CM_B_B1 + 32, CM_B_V1 = CM_B_V0 + 32, CM_B_V20 = CM_B_V1 + 32, CM_B_R0 = CM_B_V20 + 32,
              CM_B_R1 = CM_B_R0 + 32, CM_V_V1X = CM_B_R1 + 32, CM_V_V21 = CM_V_V1X + 32, CM_V_R2 = CM_V_V21 + 32;
constexpr int CM_W_S = CM_V_R2 + 32;                // [144][64]
constexpr int CM_S = CM_W_S + 144 * 64;             // [s, bias vis_fc.2[32], bias vis_fc2.2, bias rgb_fc.4]
constexpr int CM_TOTAL = CM_S + 4;
// split-f16 blob: A segments [block][k-step of 16][hi|lo][64 lanes][8 f16 = 4 floats], then the fp32 tail (CM_BIAS0 .. CM_TOTAL)
constexpr int CX_A_RD0 = 0;                         // [1][1]
constexpr int CX_A_RD1 = CX_A_RD0 + 1 * 1 * 512;    // [2][1]
constexpr int CX_A_B0 = CX_A_RD1 + 2 * 1 * 512;     // [2][4]
constexpr int CX_A_B1 = CX_A_B0 + 2 * 4 * 512;      // [1][4]
constexpr int CX_A_V0 = CX_A_B1 + 4 * 512;          // [1][2]
constexpr int CX_A_V1 = CX_A_V0 + 2 * 512;          // [1][2]
constexpr int CX_A_V20 = CX_A_V1 + 2 * 512;         // [1][2]
constexpr int CX_A_R0 = CX_A_V20 + 2 * 512;         // [1][3]
constexpr int CX_A_R1 = CX_A_R0 + 3 * 512;          // [1][1]
constexpr int CX_A_END = CX_A_R1 + 1 * 512;
constexpr int CX_TOTAL = CX_A_END + (CM_TOTAL - CM_BIAS0);
// The view-independent rows of base_fc.0 (geo | mean | var -> 64) once more as a matrix-core A operand, for csrc/color_pts.hip where a
// column is a POINT and the rows are evaluated by MFMA (2 output blocks, 72 per-half operands: 8 geometry channels, 32 means, 32
// variances of the half's pixel floats).  Appended behind the tail so that the prefix read by k_color_mfma is unchanged.
constexpr int CM_A_S = CM_TOTAL;                    // fp32 form [2][72][64]
constexpr int CM_TOTAL2 = CM_A_S + 2 * 72 * 64;
constexpr int CX_A_S = CX_TOTAL;                    // split-f16 form [2][9 k-steps][hi|lo][64][8 f16]
constexpr int CX_TOTAL2 = CX_A_S + 2 * 9 * 512;

struct ColorMArgs {
    const float* blob;
    const float* vol_cl; const float* maskvol; int D;
    const float* cmaps; const float* proj; const float* cam_pos; int V, H, W_img;
    const float* pts; const int* index; const int* n_dev; long long n;
    const float* query_cam; const float* normals;
    float* out_rgb; uint8_t* out_nviews;
    // materialised inputs of GeneralRenderingNetwork.forward (k_color_pts<.., FEATS = true> only): the reference's view-major tensors
    const float* f_geo;       // [P,16]
    const float* f_rgb;       // [V,P,59]  colours (3) | features (56)
    const float* f_rdiff;     // [V,P,4]
    const float* f_mask;      // [V,P]     non-zero = the projection is valid
    unsigned long long* stats;  // optional caller-owned device counters (stats_dev of o2345_color_points_*): [0] += (tile, view) pairs evaluated in pass A, [1] += in pass B,
                              // [2] += tiles, [3] += tiles that evaluated every view in pass B because one of their points has no visible view
    int sched;                // scheduling knobs (O2345_COLOR_SCHED; default 10 = bits 1 + 3, measured on MI355X with tools/ab_sched.py):
                              //   bit 0  static wave priority by SIMD slot (the k-th wave of a SIMD runs at priority k): no gain
                              //   bit 1  priority 3 while a wave issues its pixel gathers: -1.5 %
                              //   bit 2  k_color_pts evaluates EVERY view (no skipping of views that see none of a tile's points): +13 % (the round-2 kernel)
                              //   bit 3  k_color_pts: block-interleaved tile schedule instead of one contiguous eighth of the list per XCD -- with view
                              //          skipping a tile's cost depends on where its rays look: -1 % at 8 views, -20 % at 32 views
};

inline int color_sched_mode() { return knobs().color_sched; }
#if defined(__HIPCC__)
__device__ __forceinline__ void set_wave_prio(int p) {       // s_setprio takes an immediate
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#endif

// ELU is evaluated ~150 times per lane and tile (a quarter of the kernel's VALU instructions), so the whole network runs in a
// log2(e)-scaled domain: every layer that feeds an ELU produces y = log2(e) * x (its weights / bias are pre-scaled on the host,
// weights.pack_color_mfma_blob) and the activation is   ELU_y(y) = log2(e) * ELU(x) = max(y, log2(e) * (min(2^y, 1) - 1)):
// v_exp_f32 with its [0,1] output clamp, one fma, one max -- no multiply by log2(e) in front of the exponential.  The scale
// cancels in the next layer (ln2 * log2e = 1, so hidden-layer weights are unchanged; only biases and the first / last layers
// carry a factor).  e^x - 1 has an ABSOLUTE error of ~1e-7 (one ulp of 1.0), which is what matters downstream.
// Reciprocals are the hardware v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence.
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
__device__ __forceinline__ float celu(float y) {
    const float t = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y), 0.f, 1.f);
    return fmaxf(y, fmaf(t, LOG2E, -LOG2E));
}
// two at a time: the multiply-add is one packed-fp32 instruction for both (consecutive accumulator registers are an aligned pair)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 celu2(float y0, float y1) {
    f32x2 t;
    t[0] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y0), 0.f, 1.f);
    t[1] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y1), 0.f, 1.f);
    const f32x2 e = __builtin_elementwise_fma(t, f32x2{LOG2E, LOG2E}, f32x2{-LOG2E, -LOG2E});
    return f32x2{fmaxf(y0, e[0]), fmaxf(y1, e[1])};
}
__device__ __forceinline__ float crcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float csigm(float z) { return crcp(1.f + __builtin_amdgcn_exp2f(-z)); }      // sigmoid of z / log2(e)

// NB output blocks, N k-steps whose B operands are b[0..N-1]; A operands come from LDS, next step prefetched
template <int NB, int NST, int N>
__device__ __forceinline__ void cm_run(f32x16 (&acc)[NB], const float* A /* + lane */, int step0, const float (&b)[N]) {
    float cur[NB], nxt[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) cur[nb] = A[(nb * NST + step0) * 64];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        if (r + 1 < N) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) nxt[nb] = A[(nb * NST + step0 + r + 1) * 64];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA32(cur[nb], b[r], acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) cur[nb] = nxt[nb];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// split-f16 form of the same step loop: b[] is the per-half operand list of the fp32 form, consumed 8 per MFMA step
struct Split8 { h16x8 hi, lo; };
typedef _Float16 hh16x2 __attribute__((ext_vector_type(2)));
// hi = f16(x) rounded toward zero, lo = f16(x - hi) with the exact difference from one v_fma_mix_f32 (see csrc/sdf_mlp_x3.hip)
__device__ __forceinline__ float opaque_minus_one() {
    float m1 = -1.f;
    asm volatile("" : "+v"(m1));
    return m1;
}
template <int N>
__device__ __forceinline__ Split8 split8(const float (&b)[N], int s0, float m1) {      // s0 compile-time after unrolling
    union { h16x8 v8; h16x2 v2[4]; hh16x2 w2[4]; } hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = (s0 + 2 * i < N) ? b[s0 + 2 * i < N ? s0 + 2 * i : 0] : 0.f;
        const float y = (s0 + 2 * i + 1 < N) ? b[s0 + 2 * i + 1 < N ? s0 + 2 * i + 1 : 0] : 0.f;
        hi.v2[i] = __builtin_amdgcn_cvt_pkrtz(x, y);
        // (the v_fma_mixlo_f16 / v_fma_mixhi_f16 form of csrc/sdf_mlp_x3.hip was measured here too: 40.9 vs 39.6 ms for k_color_pts -- the two-instruction
        // asm block constrains the scheduler more than it saves -- so the colour kernels keep the three-instruction C form; -DO2345_COLOR_SPLIT_MIXLO=1: A/B)
#if defined(O2345_COLOR_SPLIT_MIXLO) && O2345_COLOR_SPLIT_MIXLO
        lo.v2[i] = __builtin_bit_cast(h16x2, split_lo_pair_bits(__builtin_bit_cast(unsigned, hi.v2[i]), x, y));
#else
        lo.v2[i] = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hi.w2[i][0], m1, x), __builtin_fmaf((float)hi.w2[i][1], m1, y));
#endif
    }
    return {hi.v8, lo.v8};
}
template <int NB, int N>
__device__ __forceinline__ void cx_run(f32x16 (&acc)[NB], const float4* A /* segment + lane */, const float (&b)[N], float m1) {
    constexpr int NS = (N + 7) / 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const Split8 sp = split8(b, 8 * s, m1);
        h16x8 ahi[NB], alo[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            ahi[nb] = __builtin_bit_cast(h16x8, A[((nb * NS + s) * 2 + 0) * 64]);
            alo[nb] = __builtin_bit_cast(h16x8, A[((nb * NS + s) * 2 + 1) * 64]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(alo[nb], sp.hi, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(ahi[nb], sp.lo, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(ahi[nb], sp.hi, acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// the same with the accumulators STARTING from `init` (the first matrix instruction of every block reads it as its C operand and writes `acc`: no
// register copies of a value that must stay live, e.g. the view-independent rows shared by all views of a point)
template <int NB, int N>
__device__ __forceinline__ void cx_run_from(f32x16 (&acc)[NB], const f32x16 (&init)[NB], const float4* A, const float (&b)[N], float m1) {
    constexpr int NS = (N + 7) / 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const Split8 sp = split8(b, 8 * s, m1);
        h16x8 ahi[NB], alo[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            ahi[nb] = __builtin_bit_cast(h16x8, A[((nb * NS + s) * 2 + 0) * 64]);
            alo[nb] = __builtin_bit_cast(h16x8, A[((nb * NS + s) * 2 + 1) * 64]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(alo[nb], sp.hi, s == 0 ? init[nb] : acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(ahi[nb], sp.lo, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(ahi[nb], sp.hi, acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// one layer's matrix part in either form
template <bool X3, int NB, int N>
__device__ __forceinline__ void cm_layer(f32x16 (&acc)[NB], const float* lds, int lane, int off32, int offx, const float (&b)[N], float m1) {
    if constexpr (X3) cx_run<NB, N>(acc, reinterpret_cast<const float4*>(lds + offx) + lane, b, m1);
    else cm_run<NB, N, N>(acc, lds + lane + off32, 0, b);
}

// biases are stored [block][half][16 registers]: four 16-byte LDS reads straight into the accumulator tuple
template <int NB>
__device__ __forceinline__ void cm_bias(f32x16 (&acc)[NB], const float* bias, int h) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float4* p = reinterpret_cast<const float4*>(bias + (nb * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = p[q];
            acc[nb][4 * q] = t.x; acc[nb][4 * q + 1] = t.y; acc[nb][4 * q + 2] = t.z; acc[nb][4 * q + 3] = t.w;
        }
    }
}

// Reductions over the G view lanes of a point (G consecutive lanes, G | 32) with DPP lane permutes instead of
// ds_bpermute: xor 1 / xor 2 are quad_perm, then row_half_mirror (i <-> 7-i) and row_mirror (i <-> 15-i) combine quads
// that already hold their own partial result; only the 16 <-> 16 step of G = 32 goes through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define O2345_GROUP_REDUCE(NAME, OP)                                                   \
    template <int G>                                                                   \
    __device__ __forceinline__ float NAME(float v) {                                   \
        v = OP(v, dpp_mov<0xB1>(v));                        /* quad_perm [1,0,3,2] */  \
        v = OP(v, dpp_mov<0x4E>(v));                        /* quad_perm [2,3,0,1] */  \
        if (G >= 8) v = OP(v, dpp_mov<0x141>(v));           /* row_half_mirror     */  \
        if (G >= 16) v = OP(v, dpp_mov<0x140>(v));          /* row_mirror          */  \
        if (G >= 32) v = OP(v, __shfl_xor(v, 16));                                     \
        return v;                                                                      \
    }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
O2345_GROUP_REDUCE(gsum, op_add)
O2345_GROUP_REDUCE(gmin, fminf)
O2345_GROUP_REDUCE(gmax, fmaxf)
#undef O2345_GROUP_REDUCE

__device__ __forceinline__ void cm_project(const float* __restrict__ P, float x, float y, float z, int H, int W, float& gx, float& gy) {
    const float X = P[0] * x + P[1] * y + P[2] * z + P[3];
    const float Y = P[4] * x + P[5] * y + P[6] * z + P[7];
    const float Z = fmaxf(P[8] * x + P[9] * y + P[10] * z + P[11], 1e-3f);
    // a / b as a * rcp(b) with one Newton step on the quotient (q += (a - b q) * r): within 1 ulp of the IEEE quotient (almost
    // always identical) in 3 instructions instead of the ~12 of the division sequence; the reciprocals are shared
    const float rz = crcp(Z), rw = crcp((float)(W - 1)), rh = crcp((float)(H - 1));
    auto div = [](float a_, float b_, float r_) { const float q = a_ * r_; return fmaf(fmaf(-b_, q, a_), r_, q); };
    gx = div(2.f * div(X, Z, rz), (float)(W - 1), rw) - 1.f;
    gy = div(2.f * div(Y, Z, rz), (float)(H - 1), rh) - 1.f;
    if (gx > 1.f || gx < -1.f) gx = 2.f;
    if (gy > 1.f || gy < -1.f) gy = 2.f;
}


#if defined(__HIPCC__)
// geometry of one source view for this lane's point: ray_diff (4), pooling exponent, projection mask and the bilinear taps
struct ViewGeom {
    float rd[4];
    float e;          // 2^(s_abs * (dot - 1))
    float m;          // 1 if the point is valid and projects inside view v
    float gx, gy;
};

__device__ __forceinline__ ViewGeom view_geom(const ColorMArgs& a, int v, float px, float py, float pz, float qx, float qy, float qz,
                                              bool gvalid, float s_abs) {
    ViewGeom g;
    cm_project(a.proj + 12 * v, px, py, pz, a.H, a.W_img, g.gx, g.gy);
    g.m = (gvalid && fabsf(g.gx) < 1.f && fabsf(g.gy) < 1.f) ? 1.f : 0.f;
    const float sx = a.cam_pos[3 * v] - px, sy = a.cam_pos[3 * v + 1] - py, sz = a.cam_pos[3 * v + 2] - pz;
    const float rsn = crcp(sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
    const float ux = sx * rsn, uy = sy * rsn, uz = sz * rsn;
    const float dx = qx - ux, dy = qy - uy, dz = qz - uz;
    const float rdn = crcp(fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f));
    g.rd[0] = dx * rdn; g.rd[1] = dy * rdn; g.rd[2] = dz * rdn;
    g.rd[3] = qx * ux + qy * uy + qz * uz;
    g.e = __builtin_amdgcn_exp2f(s_abs * (g.rd[3] - 1.f));       // s_abs carries log2(e)
    return g;
}

#endif

// csrc/color_pts.hip: the points-as-columns kernel (the product kernel; k_color_mfma of csrc/color_mfma.hip exists only in -DO2345_TILES_KERNEL test builds)
int project_features_launch(const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj, const float* cam_pos, int V, int H, int W,
                            const float* pts, long long n, const float* query_cam, const float* normals, float* geo, float* rgb_feat, float* rdiff, float* mask,
                            void* stream);
int color_feats_launch(int x3, const float* blob, const float* geo, const float* rgb_feat, const float* ray_diff, const float* mask, int V, long long n,
                       float* out_rgb, uint8_t* out_nviews, void* stream);
int color_pts_launch(int x3, const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj,
                     const float* cam_pos, int V, int H, int W, const float* pts, const int32_t* index, const int32_t* n_dev, long long n,
                     const float* query_cam, const float* normals, float* out_rgb, uint8_t* out_nviews, unsigned long long* stats_dev, void* stream);

}  // namespace o2345
