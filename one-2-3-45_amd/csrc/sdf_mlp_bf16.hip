// SDF network, throughput mode for BASELINE config 2 ("bf16 SDF MLP"): same function, same blob, same lane layout as
// csrc/sdf_mlp.hip, but the wide layers run on v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate, fp32 accumulate).
// Mixed-precision recipe (which measured bf16 everywhere to flip 5e-4 of SDF signs):
//   * layer 0 (positional encoding incl. the raw xyz columns) stays on the exact fp32 MFMA
//   * layer 1 (128+16 -> 128) uses bf16 operands: hidden activations are converted pairwise (v_cvt_pk_bf16_f32) right
//     where they are produced; the MFMA result layout is again the k enumeration of the consumer (8 consecutive
//     registers of one accumulator block per 16-wide k step), so nothing leaves registers
//   * the SDF output row (144 -> 1) is an fp32 dot product; softplus / sigmoid in fp32
//   * gradient variant: both transposed GEMMs on bf16 operands, layer 0 re-run in fp32 for softplus'
// Parity is tolerance-based (tests/test_gpu_parity.py::test_sdf_mlp_bf16): this mode is opt-in, fp32 is the default.
#include "sdf_common.h"

namespace o2345 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
    return r;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int r0) {     // r0 is a compile-time constant after unrolling
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[r0 + i];
    return r;
}

// acc[ob] += A[ob][step] * b for NB output blocks; A: LDS, [NB][NST][64 lanes] float4 (= 8 bf16)
template <int NB, int NST>
__device__ __forceinline__ void mma_h(f32x16 (&acc)[NB], const float4* A, int lane, int step, bf16x8 b) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
        const float4 av = A[(ob * NST + step) * 64 + lane];
        acc[ob] = MFMA_BF16(__builtin_bit_cast(bf16x8, av), b, acc[ob]);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the operand fetches of later steps from being hoisted (register pressure)
}

template <int VARIANT>      // VAR_SDF or VAR_GRAD
__global__ __launch_bounds__(512) void k_sdf_mlp_bf16(SdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int N_A0 = 4 * ST0 * 64, N_A1H = 4 * STH1 * 64 * 4, N_A1TH = 5 * STHB * 64 * 4, N_A0TH = 2 * STHB * 64 * 4;
    constexpr int L_A0 = 0, L_A1H = N_A0, L_A1TH = L_A1H + N_A1H, L_A0TH = L_A1TH + (VARIANT == VAR_GRAD ? N_A1TH : 0),
                  L_MISC = L_A0TH + (VARIANT == VAR_GRAD ? N_A0TH : 0);
    auto stage = [&](int dst, int src, int nfl) {
        for (int i = threadIdx.x * 4; i < nfl; i += blockDim.x * 4)
            *reinterpret_cast<float4*>(lds + dst + i) = *reinterpret_cast<const float4*>(a.blob + src + i);
    };
    stage(L_A0, OFF_A0, N_A0);
    stage(L_A1H, OFFH_A1, N_A1H);
    if (VARIANT == VAR_GRAD) { stage(L_A1TH, OFFH_A1T, N_A1TH); stage(L_A0TH, OFFH_A0T, N_A0TH); }
    for (int i = threadIdx.x; i < MISC_SIZE; i += blockDim.x) lds[L_MISC + i] = a.blob[OFF_MISC + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const ASrc A0{lds + L_A0, __builtin_amdgcn_make_buffer_rsrc((void*)a.blob, 0, BLOB_FLOATS * 4, 0x00020000), OFF_A0};
    const float4* A1H = reinterpret_cast<const float4*>(lds + L_A1H);
    const float4* A1TH = reinterpret_cast<const float4*>(lds + L_A1TH);
    const float4* A0TH = reinterpret_cast<const float4*>(lds + L_A0TH);
    const float* misc = lds + L_MISC;

    const long long n = a.n_dev ? (long long)*a.n_dev : a.n;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const TileSched ts = tile_schedule(n, 32, wave, nwave);
    for (long long tile = ts.first; tile < ts.end; tile += ts.stride) {
        const long long t0 = tile * 32;
        const long long i = t0 + j;
        const bool live = i < n;
        long long slot = live ? (a.index ? (long long)a.index[i] : i) : 0;
        float px, py, pz;
        if (a.pts) {
            px = live ? a.pts[slot * 3 + 0] : 0.f; py = live ? a.pts[slot * 3 + 1] : 0.f; pz = live ? a.pts[slot * 3 + 2] : 0.f;
        } else {
            const int R = a.R;
            const unsigned us = (unsigned)slot, uR = (unsigned)R;          // R^3 < 2^32: 32-bit divisions (the 64-bit ones cost 240 instructions)
            const unsigned uq = us / uR;
            const int iz = (int)(us - uq * uR), ix = (int)(uq / uR), iy = (int)(uq - (uq / uR) * uR);
            px = lin11(ix, R); py = lin11(iy, R); pz = lin11(iz, R);
        }
        // ---- trilinear latent (this half's 8 channels) + Jacobian: identical to the fp32 kernel ---------------------------
        float lat[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) lat[c] = 0.f;
        {
            const Taps3D tp = trilinear_ref_taps(px, py, pz, a.D);
            if (tp.ok && live) {
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dz = 0; dz < 2; ++dz) {
                            const size_t vox = ((size_t)tp.ix[dx] * a.D + tp.iy[dy]) * a.D + tp.iz[dz];
                            const float4* p4 = reinterpret_cast<const float4*>(a.vol_cl + vox * 16 + 8 * h);
                            const float4 v0 = p4[0], v1 = p4[1];
                            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                            const float w = tp.fz[dz] * tp.fy[dy] * tp.fx[dx];
#pragma unroll
                            for (int c = 0; c < 8; ++c) lat[c] = fmaf(v[c], w, lat[c]);
                        }
            }
        }
        // ---- positional encoding (fp32) ------------------------------------------------------------------------------------------
        float pe[20];
        const float p3[3] = {px, py, pz};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int c = 9 * h + t;
            const float f = (float)(1 << (c / 3));
            float s, co;
            sincos_pe(p3[t % 3] * f, s, co);
            pe[t] = s; pe[9 + t] = co;
        }
        pe[18] = h ? pz : px;
        pe[19] = h ? 0.f : py;
        // ---- layer 0: exact fp32 MFMA ------------------------------------------------------------------------------------------------
        f32x16 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = misc[MISC_B0 + (nb * 16 + r) * 2 + h];
        mma_run<4, ST0, 20, false>(acc, A0, 0, lane, 0, pe);
        bf16x8 hb[8];                   // softplus(layer 0) as the 8 bf16 k-step operands of layer 1
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float hv[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 sp = softplus100_pair(f32x2{acc[nb][r], acc[nb][r + 1]});
                hv[r] = sp[0]; hv[r + 1] = sp[1];
            }
            hb[2 * nb] = pack8(hv); hb[2 * nb + 1] = pack8(hv + 8);
        }
        // ---- layer 1: bf16 operands, fp32 accumulate ---------------------------------------------------------------------------------
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = misc[MISC_B1 + (nb * 16 + r) * 2 + h];
#pragma unroll
        for (int s = 0; s < 8; ++s) mma_h<4, STH1>(acc, A1H, lane, s, hb[s]);
        mma_h<4, STH1>(acc, A1H, lane, 8, pack8(lat));
        bf16x8 g1b[8];                  // d sdf / d a1 = w2row * softplus'(a1), already as backward k-step operands
        float y0 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float gv[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 d;
                const f32x2 v = (VARIANT == VAR_GRAD) ? softplus100_pair(f32x2{acc[nb][r], acc[nb][r + 1]}, d)
                                                      : softplus100_pair(f32x2{acc[nb][r], acc[nb][r + 1]});
                const float w2a = misc[MISC_W2H + (nb * 16 + r) * 2 + h], w2b = misc[MISC_W2H + (nb * 16 + r + 1) * 2 + h];
                y0 = fmaf(w2a, v[0], y0); y0 = fmaf(w2b, v[1], y0);
                if (VARIANT == VAR_GRAD) { gv[r] = w2a * d[0]; gv[r + 1] = w2b * d[1]; }
            }
            if (VARIANT == VAR_GRAD) { g1b[2 * nb] = pack8(gv); g1b[2 * nb + 1] = pack8(gv + 8); }
        }
        // ---- SDF output row in fp32 -----------------------------------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 8; ++t) y0 += misc[MISC_W2L + 8 * h + t] * lat[t];
        y0 += __shfl_xor(y0, 32);
        y0 += misc[MISC_B2];
        if (live && h == 0) a.out_sdf[slot] = a.sign * y0;
        // ---- backward: d sdf / d x --------------------------------------------------------------------------------------------------------
        if (VARIANT == VAR_GRAD) {
            f32x16 g[5];
#pragma unroll
            for (int nb = 0; nb < 5; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) g[nb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) mma_h<5, STHB>(g, A1TH, lane, s, g1b[s]);
            f32x16 gp[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) gp[nb][r] = 0.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {                      // softplus'(a0): re-run layer 0 (fp32) block by block
                f32x16 a0r[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) a0r[0][r] = misc[MISC_B0 + (nb * 16 + r) * 2 + h];
                mma_run<1, ST0, 20, false>(a0r, A0, nb, lane, 0, pe);
                float gv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gv[r] = g[nb][r] * softplus100_d(a0r[0][r]);
                mma_h<2, STHB>(gp, A0TH, lane, 2 * nb, pack8(gv));
                mma_h<2, STHB>(gp, A0TH, lane, 2 * nb + 1, pack8(gv + 8));
            }
            float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int c = 9 * h + t;
                const int d = t % 3;
                const float f = (float)(1 << (c / 3));
                const float gs = gp[0][t];
                const float gc = (9 + t < 16) ? gp[0][9 + t] : gp[1][9 + t - 16];
                gx[d] += (gs * pe[9 + t] - gc * pe[t]) * f;
            }
            if (h) gx[2] += gp[1][2]; else { gx[0] += gp[1][2]; gx[1] += gp[1][3]; }
            {   // latent path: re-gather the 8 taps (L2 hits) and contract d sdf / d latent with the trilinear Jacobian on the fly
                float gl[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) gl[t] = g[4][t] + misc[MISC_W2L + 8 * h + t];
                const Taps3D tp = trilinear_ref_taps(px, py, pz, a.D);
                if (tp.ok && live) {
                    const float half_span = (float)(a.D - 1) * 0.5f;
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dz = 0; dz < 2; ++dz) {
                                const size_t vox = ((size_t)tp.ix[dx] * a.D + tp.iy[dy]) * a.D + tp.iz[dz];
                                const float4* p4 = reinterpret_cast<const float4*>(a.vol_cl + vox * 16 + 8 * h);
                                const float4 v0 = p4[0], v1 = p4[1];
                                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                                float dv = 0.f;
#pragma unroll
                                for (int c = 0; c < 8; ++c) dv = fmaf(v[c], gl[c], dv);
                                gx[0] = fmaf((dx ? half_span : -half_span) * tp.fy[dy] * tp.fz[dz], dv, gx[0]);
                                gx[1] = fmaf((dy ? half_span : -half_span) * tp.fx[dx] * tp.fz[dz], dv, gx[1]);
                                gx[2] = fmaf((dz ? half_span : -half_span) * tp.fx[dx] * tp.fy[dy], dv, gx[2]);
                            }
                }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) gx[d] += __shfl_xor(gx[d], 32);
            if (live && h == 0 && a.out_grad) {
                a.out_grad[slot * 3 + 0] = gx[0]; a.out_grad[slot * 3 + 1] = gx[1]; a.out_grad[slot * 3 + 2] = gx[2];
            }
        }
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

// variant: 0 = SDF only, 2 = SDF + analytic gradient (the 128-feature variant exists in fp32 only)
int o2345_sdf_mlp_bf16(int variant, const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index,
                       const int32_t* n_dev, long long n, int grid_R, float sign, float* out_sdf, float* out_grad, void* stream) {
    O2345_REQUIRE(blob && vol_cl && out_sdf, "sdf_mlp_bf16: null pointer");
    O2345_REQUIRE(variant == VAR_SDF || variant == VAR_GRAD, "sdf_mlp_bf16: variant must be 0 or 2 (got %d)", variant);
    O2345_REQUIRE(pts || (grid_R >= 2 && grid_R <= 1600), "sdf_mlp_bf16: need points or a grid resolution in [2, 1600]");
    O2345_REQUIRE(variant != VAR_GRAD || out_grad, "sdf_mlp_bf16: gradient variant needs out_grad");
    if (n <= 0 && !n_dev) return 0;
    SdfArgs a{blob, vol_cl, D, pts, index, n_dev, n, grid_R, sign, out_sdf, nullptr, nullptr, out_grad, nullptr};
    const int n_cu = cu_count();
    const int threads = 512;
    const long long per_block = (threads / 64) * 32;
    long long want = n_dev ? n_cu : (n + per_block - 1) / per_block;
    const unsigned grid = persistent_grid(want, n_cu);
    size_t lds_floats = 4 * ST0 * 64 + 4 * STH1 * 64 * 4 + MISC_SIZE + (variant == VAR_GRAD ? (5 + 2) * STHB * 64 * 4 : 0);
    const size_t lds_bytes = lds_floats * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (variant == VAR_SDF) {
        (void)hipFuncSetAttribute((const void*)k_sdf_mlp_bf16<VAR_SDF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_sdf_mlp_bf16<VAR_SDF>, dim3(grid), dim3(threads), lds_bytes, s, a);
    } else {
        (void)hipFuncSetAttribute((const void*)k_sdf_mlp_bf16<VAR_GRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_sdf_mlp_bf16<VAR_GRAD>, dim3(grid), dim3(threads), lds_bytes, s, a);
    }
    return check_launch("sdf_mlp_bf16");
}

}  // extern "C"
