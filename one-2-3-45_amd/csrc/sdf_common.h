// Shared pieces of the SDF-network kernels (csrc/sdf_mlp.hip: exact fp32 MFMA; csrc/sdf_mlp_x3.hip: split-f16 operands): blob geometry, argument block, softplus, ATen-exact linspace, A-operand fetch and the pinned MFMA step loop.
#pragma once
#include "common.h"
#include "geom_math.h"
#include "pe_math.h"

namespace o2345 {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---- blob geometry (must match one-2-3-45_amd/weights.py) --------------------------------------------------------
constexpr int ST0 = 20;          // layer-0 k steps  (40 PE slots = 39 + 1 pad)
constexpr int ST1 = 72;          // layer-1/2 k steps (64 hidden + 8 latent)
constexpr int STB = 64;          // backward k steps (128 upstream neurons)
constexpr int OFF_A0 = 0;                          // [4][ST0][64]
constexpr int OFF_A1 = OFF_A0 + 4 * ST0 * 64;      // [4][ST1][64]
constexpr int OFF_A2 = OFF_A1 + 4 * ST1 * 64;      // [4][ST1][64]
constexpr int OFF_A1T = OFF_A2 + 4 * ST1 * 64;     // [5][STB][64]   d/d(h0 | latent)
constexpr int OFF_A0T = OFF_A1T + 5 * STB * 64;    // [2][STB][64]   d/d(pe)
constexpr int OFF_MISC = OFF_A0T + 2 * STB * 64;   // b0[128] b1[128] b2[128] w2row_h[128] w2row_lat[16] (lane-half order)
constexpr int MISC_B0 = 0, MISC_B1 = 128, MISC_B2 = 256, MISC_W2H = 384, MISC_W2L = 512, MISC_SIZE = 528;
constexpr int BLOB_F32_FLOATS = OFF_MISC + MISC_SIZE;
constexpr int OFFX_MISC = BLOB_F32_FLOATS;         // the split-f16 kernels' MISC block: b0, b1 in the t domain (weights.py SOFTPLUS_SCALE), the rest as above
// reserved section (the bf16 operand copies of the bf16 mode removed in round 3): [block][step][64 lanes][4 floats]; the split-f16 offsets follow it
constexpr int STH1 = 9;                                  // layer-1 k steps of 16 (8 hidden + 1 latent)
constexpr int STHB = 8;                                  // backward k steps of 16 (128 upstream neurons)
constexpr int OFFH_A1 = BLOB_F32_FLOATS;                 // [4][STH1][64][4]
constexpr int OFFH_A1T = OFFH_A1 + 4 * STH1 * 64 * 4;    // [5][STHB][64][4]
constexpr int OFFH_A0T = OFFH_A1T + 5 * STHB * 64 * 4;   // [2][STHB][64][4]
constexpr int BLOB_BF16_END = OFFH_A0T + 2 * STHB * 64 * 4;
// split-f16 ("f16x3") copies for csrc/sdf_mlp_x3.hip: every weight as hi = f16(w), lo = f16(w - hi);
// [block][k-step of 16][hi|lo][64 lanes][8 f16 = 4 floats]
constexpr int STX0 = 3;                                  // layer-0 k steps of 16 (2 x 20 PE slots padded to 2 x 24)
constexpr int OFFX_A0 = BLOB_BF16_END;                   // [4][STX0][2][64][4]
constexpr int OFFX_A1 = OFFX_A0 + 4 * STX0 * 2 * 256;    // [4][STH1][2][64][4]
constexpr int OFFX_A1T = OFFX_A1 + 4 * STH1 * 2 * 256;   // [5][STHB][2][64][4]
constexpr int OFFX_A0T = OFFX_A1T + 5 * STHB * 2 * 256;  // [2][STHB][2][64][4]
constexpr int BLOB_FLOATS = OFFX_A0T + 2 * STHB * 2 * 256;

enum : int { VAR_SDF = 0, VAR_FULL = 1, VAR_GRAD = 2 };

struct SdfArgs {
    const float* blob;        // BLOB_FLOATS floats
    const float* vol_cl;      // [D,D,D,16] channel-last latent volume
    int D;
    const float* pts;         // [P,3] (mode 0) or null (mode 1: x-major grid of side R on linspace(-1,1,R))
    const int* index;         // optional gather/scatter list: point i is pts[index[i]] and results go to slot index[i]
    const int* n_dev;         // optional device-side count overriding n
    long long n;
    int R;
    float sign;               // sdf output multiplier (extract_fields stores u = -sdf)
    float* out_sdf;           // [P]
    float* out_feat;          // [P,128] or null (VAR_FULL)
    float* out_lat;           // [P,16] or null
    float* out_grad;          // [P,3] or null (VAR_GRAD)
    const float* lat_in;      // optional [P,16]: use this latent instead of sampling the volume (get_sdf_volume)
    // lattice mode of k_sdf_mlp_x3 only: layer 0 tabulated per axis (weights.sdf_grid_tables): its pre-activation is separable on the lattice,
    // a0(ix,iy,iz) = b0 + Tx[ix] + Ty[iy] + Tz[iz]; rows of 128 floats in lane order [wave half][accumulator block * 16 + register]
    const float* tab_xy;      // [R*R][128] = b0 + Tx[ix] + Ty[iy]
    const float* tab_z;       // [R][128]
};

// Softplus(beta=100, threshold=20) and its derivative (torch: x if 100x > 20 else log1p(exp(100x))/100; backward
// z/(z+1)).  The network evaluates 256 of these per point and the kernels are VALU-bound, so this is written for the
// hardware exp2/log2 units:  softplus(a) = max(a,0) + log1p(e)/100,  e = exp(-|100a|) in (0,1].
// log1p(e) is taken as ln2*log2(fl(1+e)): the rounding of 1+e is an ABSOLUTE error of <= 6e-8 in the logarithm, i.e.
// <= 6e-10 in the result after the /100 -- far below the fp32 resolution of the O(0.1) pre-activations it is added to
// (a relative-accuracy correction of the tiny tail would cost 6 more instructions per call and buys nothing).
// For 100a > 16.7, fl(1+e) == 1 and the result is exactly a, which also realises torch's threshold branch.
__device__ __forceinline__ float softplus100(float a, float& dsig) {
    const float e = __builtin_amdgcn_exp2f(fabsf(a) * -144.269504088896340736f);     // exp(-|100 a|)
    const float u = 1.f + e;
    const float l2 = __builtin_amdgcn_logf(u);                                       // log2(1 + e)
    dsig = (a >= 0.f ? 1.f : e) * __builtin_amdgcn_rcpf(u);                           // sigmoid(100 a); dead code unless used
    return fmaf(l2, 0.00693147180559945309f, fmaxf(a, 0.f));
}

// Two at a time: the multiply, the 1 + e and the final multiply-add are packed fp32 instructions (v_pk_mul/add/fma_f32 work
// on an even-aligned register pair -- consecutive accumulator registers are one), -|t| rides on v_exp_f32's source modifiers:
// 4.5 instructions per value instead of 6.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 softplus100_pair(f32x2 a) {
    const f32x2 t = a * 144.269504088896340736f;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[0]));
    e[1] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[1]));
    const f32x2 u = e + 1.f;
    f32x2 l, m;
    l[0] = __builtin_amdgcn_logf(u[0]); l[1] = __builtin_amdgcn_logf(u[1]);
    m[0] = fmaxf(a[0], 0.f); m[1] = fmaxf(a[1], 0.f);
    return __builtin_elementwise_fma(l, f32x2{0.00693147180559945309f, 0.00693147180559945309f}, m);
}

// value and derivative for a pair: sigmoid(100 a) = 1/u for a >= 0 and 1 - 1/u for a < 0, i.e. 0.5 + copysign(1/u - 0.5, a)
__device__ __forceinline__ f32x2 softplus100_pair(f32x2 a, f32x2& dsig) {
    const f32x2 t = a * 144.269504088896340736f;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[0]));
    e[1] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[1]));
    const f32x2 u = e + 1.f;
    f32x2 l, m, ru;
    l[0] = __builtin_amdgcn_logf(u[0]); l[1] = __builtin_amdgcn_logf(u[1]);
    ru[0] = __builtin_amdgcn_rcpf(u[0]); ru[1] = __builtin_amdgcn_rcpf(u[1]);
    m[0] = fmaxf(a[0], 0.f); m[1] = fmaxf(a[1], 0.f);
    f32x2 q = ru - 0.5f;
    q[0] = __builtin_copysignf(q[0], a[0]); q[1] = __builtin_copysignf(q[1], a[1]);
    dsig = q + 0.5f;
    return __builtin_elementwise_fma(l, f32x2{0.00693147180559945309f, 0.00693147180559945309f}, m);
}

// ---- the same in the t domain (csrc/sdf_mlp_x3.hip): t = 100 a / ln 2 arrives from the matrix cores (the factor is folded into the packed operands,
// weights.py SOFTPLUS_SCALE) and the activation leaves as s' = softplus(a) * 100 / ln 2 = max(t, 0) + log2(1 + 2^-|t|): no multiply in front, a plain add
// at the end -- 5 vector instructions per value instead of 6.  For t > 24, fl(1 + 2^-t) == 1 and s' == t exactly (torch's threshold branch, 100 a > 20
// <=> t > 28.9).  softplus'(a) = sigmoid(100 a) = sigmoid(t ln 2) = 1 / (1 + 2^-t).
constexpr float SOFTPLUS_INV_SCALE = 0.00693147180559945309f;                           // ln 2 / 100: applied ONCE per point, to the SDF row's hidden sum
__device__ __forceinline__ f32x2 softplus_t_pair(f32x2 t) {
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[0]));
    e[1] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[1]));
    const f32x2 u = e + 1.f;
    f32x2 l, m;
    l[0] = __builtin_amdgcn_logf(u[0]); l[1] = __builtin_amdgcn_logf(u[1]);
    m[0] = fmaxf(t[0], 0.f); m[1] = fmaxf(t[1], 0.f);
    return l + m;
}
__device__ __forceinline__ f32x2 softplus_t_pair(f32x2 t, f32x2& dsig) {
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[0]));
    e[1] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[1]));
    const f32x2 u = e + 1.f;
    f32x2 l, m, ru;
    l[0] = __builtin_amdgcn_logf(u[0]); l[1] = __builtin_amdgcn_logf(u[1]);
    ru[0] = __builtin_amdgcn_rcpf(u[0]); ru[1] = __builtin_amdgcn_rcpf(u[1]);
    m[0] = fmaxf(t[0], 0.f); m[1] = fmaxf(t[1], 0.f);
    f32x2 q = ru - 0.5f;
    q[0] = __builtin_copysignf(q[0], t[0]); q[1] = __builtin_copysignf(q[1], t[1]);
    dsig = q + 0.5f;
    return l + m;
}
__device__ __forceinline__ float softplus_t_d(float t) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-t));
}

// derivative only (the gradient kernels re-evaluate layer 0 just for this): sigmoid(100 a) = 1 / (1 + exp(-100 a))
__device__ __forceinline__ float softplus100_d(float a) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(a * -144.269504088896340736f));
}

// torch.linspace(-1, 1, R)[i] in fp32, bit-exact with ATen's CPU kernel (symmetric evaluation, fused multiply-add)
__device__ __forceinline__ float lin11(int i, int R) {
    const float step = 2.f / (float)(R - 1);
    return (i < R / 2) ? fmaf(step, (float)i, -1.f) : fmaf(-step, (float)(R - 1 - i), 1.f);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// A-operand source: an LDS-resident blob segment (plain indexing, ds_read with immediate offsets) or a segment of the
// global blob read through a BUFFER descriptor (wave-uniform rsrc + scalar offset + lane*4): with flat/global loads
// hipcc materialises one 64-bit per-lane address per load site, hoists ~200 of them out of the tile loop and spills them.
struct ASrc {
    const float* lds;
    __amdgpu_buffer_rsrc_t rsrc;
    int base;                 // float offset of the segment in the blob (global case)
};
template <bool GLOBAL>
__device__ __forceinline__ float a_load(const ASrc& s, int idx, int lane) {
    if constexpr (GLOBAL) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, lane * 4, (s.base + idx) * 4, 0));
    else return s.lds[idx + lane];
}

// One k-step: acc[nb] += A[nb][step] (x) b for all output blocks.  The A operands of the NEXT step are fetched
// before this step's MFMAs and a scheduling barrier pins that order: without it hipcc hoists hundreds of operand
// loads to the top of the (single, fully unrolled) basic block and spills.
template <int NB, int NST, int N, bool GLOBAL>
__device__ __forceinline__ void mma_run(f32x16 (&acc)[NB], const ASrc& A, int blk0, int lane, int step0, const float (&b)[N]) {
    float cur[NB], nxt[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) cur[nb] = a_load<GLOBAL>(A, ((blk0 + nb) * NST + step0) * 64, lane);
#pragma unroll
    for (int r = 0; r < N; ++r) {
        if (r + 1 < N) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) nxt[nb] = a_load<GLOBAL>(A, ((blk0 + nb) * NST + step0 + r + 1) * 64, lane);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA(cur[nb], b[r], acc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) cur[nb] = nxt[nb];
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NB, int NST, bool GLOBAL>
__device__ __forceinline__ void mma_block16(f32x16 (&acc)[NB], const ASrc& A, int lane, int step0, const f32x16& x) {
    float b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = x[r];
    mma_run<NB, NST, 16, GLOBAL>(acc, A, 0, lane, step0, b);
}


}  // namespace o2345
