// Sparse 3x3x3 convolution on the matrix cores (same semantics as k_sparse_conv in csrc/sparse.hip:
// gather-form implicit GEMM over the dense index grid, modes same / stride-2 down / transposed up).
//
//   out[co][row] = sum over (neighbour k, input channel ci) of  W[k][ci][co] * in[ nbr(row, k) ][ci]
//
// D[co][row] on v_mfma_f32_32x32x16_f16 in the split-f16 form of csrc/sdf_mlp_x3.hip (hi*hi + hi*lo + lo*hi, fp32 accumulate,
// fp32-class accuracy): a wave owns 32 output rows (B column = lane & 31); the two wave halves supply 8 input channels each of
// one 16-channel group of one neighbour, so a k step is (neighbour, channel group).  The B operand is gathered straight
// from the neighbour's row (two dwordx4 per lane) and split in registers; the A operand (weights, [27][CIN/16][blocks][hi|lo]
// [64 lanes][8 f16], packed by weights.pack_sparse_conv_x3) streams from L2 through a buffer descriptor -- identical for every
// wave of the grid, 2 KB per block and step, fetched one step ahead.  Neighbours that no row of the wave has are skipped
// (wave-uniform ballot), which removes 7/8 of the steps of the transposed mode.
// The thread-per-row fp32 VALU kernel remains the strict-fp32 path.
#include "common.h"

namespace o2345 {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hh16x2 __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct Lattice3 { int nx, ny, nz; };

template <int NB>
struct AOp { h16x8 hi[NB], lo[NB]; };

template <int NB>
__device__ __forceinline__ AOp<NB> a_fetch_lds(const float* wlds, int step, int lane) {
    AOp<NB> r;
    const float4* A = reinterpret_cast<const float4*>(wlds);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        r.hi[nb] = __builtin_bit_cast(h16x8, A[((step * NB + nb) * 2 + 0) * 64 + lane]);
        r.lo[nb] = __builtin_bit_cast(h16x8, A[((step * NB + nb) * 2 + 1) * 64 + lane]);
    }
    return r;
}
template <int NB>
__device__ __forceinline__ AOp<NB> a_fetch(__amdgpu_buffer_rsrc_t rs, int step, int lane) {
    AOp<NB> r;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int base = ((step * NB + nb) * 2) * 1024;          // bytes: [step][block][hi|lo][64 lanes][16 B]
        r.hi[nb] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, base, 0));
        r.lo[nb] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, base + 1024, 0));
    }
    return r;
}

// LDSW: the layer's operand blob (27 * CIN/16 * blocks * 2 KB) fits in LDS -- persistent 1024-thread workgroups (one per CU, four
// waves per SIMD) stage it once and loop over the tiles; otherwise (the 64-channel layers, which only exist at the coarse levels)
// 256-thread workgroups stream it from L2.
template <int CIN, int COUT, int MODE, bool LDSW>
__global__ __launch_bounds__(LDSW ? 1024 : 256) void k_sparse_conv_x3(const float* __restrict__ in, const int* __restrict__ out_coords,
                                                                      int n_out, int ts_out, const int* __restrict__ in_grid, Lattice3 lin,
                                                                      const float* __restrict__ wblob, float* __restrict__ out) {
    constexpr int NU = CIN / 16, NB = (COUT + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    if (LDSW) {
        for (int i = threadIdx.x * 4; i < 27 * NU * NB * 512; i += blockDim.x * 4)
            *reinterpret_cast<float4*>(wlds + i) = *reinterpret_cast<const float4*>(wblob + i);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int nwave = blockDim.x >> 6, ntiles = (n_out + 31) / 32;
    float m1 = -1.f;
    asm volatile("" : "+v"(m1));                                     // keeps fma(hi, -1, x) a v_fma_mix_f32 (see sdf_mlp_x3.hip)
  // XCD-contiguous schedule (common.h): block b runs on XCD b % 8 and every XCD has its own L2; one contiguous eighth of the row list per XCD
  // (the list is x-major: an eighth = a slab of x-planes) instead of tiles dealt round-robin over the blocks (-20 % on the stride-2 layer of the
  // finest level; the finest same-resolution layer does not move: it is bound by the L1's access rate)
  const TileSched tsch = tile_schedule(n_out, 32, threadIdx.x >> 6, nwave);
  for (long long tile_ll = tsch.first; tile_ll < tsch.end; tile_ll += tsch.stride) {
    const int tile = (int)tile_ll;
    const int q = tile * 32 + j;
    const bool live = q < n_out;
    int cx = 0, cy = 0, cz = 0;
    if (live) {
        const int4 c4 = reinterpret_cast<const int4*>(out_coords)[q];
        cx = c4.x / ts_out; cy = c4.y / ts_out; cz = c4.z / ts_out;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 27 * NU * NB * 2 * 1024, 0x00020000);
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    // all 27 neighbour rows first (independent index-grid loads in flight together), then the gather / MFMA steps
    int nbr[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
        int nx, ny, nz;
        bool ok = live;
        if (MODE == 0) { nx = cx + ox; ny = cy + oy; nz = cz + oz; }
        else if (MODE == 1) { nx = 2 * cx + ox; ny = 2 * cy + oy; nz = 2 * cz + oz; }
        else {
            nx = cx - ox; ny = cy - oy; nz = cz - oz;
            ok = ok && !((nx | ny | nz) & 1);
            nx >>= 1; ny >>= 1; nz >>= 1;
        }
        ok = ok && nx >= 0 && ny >= 0 && nz >= 0 && nx < lin.nx && ny < lin.ny && nz < lin.nz;
        nbr[k] = ok ? in_grid[((size_t)nx * lin.ny + ny) * lin.nz + nz] : -1;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int r = nbr[k];
        if (__ballot(r >= 0) == 0ull) continue;                       // no row of this wave has neighbour k
        const float4* src = reinterpret_cast<const float4*>(in + (size_t)(r >= 0 ? r : 0) * CIN) + 2 * h;
        AOp<NB> cur = LDSW ? a_fetch_lds<NB>(wlds, k * NU, lane) : a_fetch<NB>(rs, k * NU, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            AOp<NB> nxt;
            if (u + 1 < NU) nxt = LDSW ? a_fetch_lds<NB>(wlds, k * NU + u + 1, lane) : a_fetch<NB>(rs, k * NU + u + 1, lane);
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (r >= 0) { v0 = src[4 * u]; v1 = src[4 * u + 1]; }
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            union { h16x8 v8; h16x2 v2[4]; hh16x2 w2[4]; } bh, bl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bh.v2[i] = __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]);
                bl.v2[i] = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)bh.w2[i][0], m1, x[2 * i]),
                                                      __builtin_fmaf((float)bh.w2[i][1], m1, x[2 * i + 1]));
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(cur.lo[nb], bh.v8, acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(cur.hi[nb], bl.v8, acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_F16(cur.hi[nb], bh.v8, acc[nb]);
            if (u + 1 < NU) cur = nxt;
        }
    }
    if (!live) continue;
    float* dst = out + (size_t)q * COUT;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * nb + 8 * g + 4 * h;                  // registers 4g..4g+3 hold outputs co..co+3
            if (co < COUT) *reinterpret_cast<float4*>(dst + co) = make_float4(acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]);
        }
  }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_sparse_conv_x3_blob_floats(int cin, int cout) { return 27 * (cin / 16) * ((cout + 31) / 32) * 2 * 256; }

#define O2345_CONVX_LAUNCH(CI, CO, MD, LW)                                                                             \
    {                                                                                                                   \
        if (LW) (void)hipFuncSetAttribute((const void*)k_sparse_conv_x3<CI, CO, MD, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_sparse_conv_x3<CI, CO, MD, LW>), LW ? pgrid : grid, dim3(LW ? 1024 : 256), LW ? lds : 0, s, in, out_coords, n_out, ts_out, in_grid, lin, wblob, out); \
    }
#define O2345_CONVX_CASE(CI, CO)                                                                                        \
    if (cin == CI && cout == CO) {                                                                                      \
        constexpr size_t lds = (size_t)27 * (CI / 16) * ((CO + 31) / 32) * 2048;                                        \
        constexpr bool LW = lds <= 120 * 1024;                                                                          \
        if (mode == 0) O2345_CONVX_LAUNCH(CI, CO, 0, LW)                                                                \
        else if (mode == 1) O2345_CONVX_LAUNCH(CI, CO, 1, LW)                                                           \
        else O2345_CONVX_LAUNCH(CI, CO, 2, LW)                                                                          \
        return check_launch("sparse_conv3d_x3");                                                                        \
    }

// Same contract as o2345_sparse_conv3d; wblob = the layer's kernel packed by weights.pack_sparse_conv_x3
// (o2345_sparse_conv_x3_blob_floats(cin, cout) floats).
int o2345_sparse_conv3d_x3(int mode, const float* in, int cin, const int32_t* in_grid, int gx, int gy, int gz,
                           const int32_t* out_coords, int n_out, int ts_out, const float* wblob, int cout, float* out, void* stream) {
    O2345_REQUIRE(in && in_grid && out_coords && wblob && out, "sparse_conv3d_x3: null pointer");
    O2345_REQUIRE(mode >= 0 && mode <= 2, "sparse_conv3d_x3: bad mode %d", mode);
    if (n_out == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Lattice3 lin{gx, gy, gz};
    dim3 grid(cdiv(n_out, 128));
    const int n_cu = cu_count();
    const unsigned want = cdiv(n_out, 32 * 16);                       // 16 tiles (waves) per persistent workgroup round
    dim3 pgrid(want < (unsigned)n_cu ? want : (unsigned)n_cu);
    O2345_CONVX_CASE(32, 16) O2345_CONVX_CASE(16, 16) O2345_CONVX_CASE(16, 32) O2345_CONVX_CASE(32, 32)
    O2345_CONVX_CASE(32, 64) O2345_CONVX_CASE(64, 64) O2345_CONVX_CASE(64, 32) O2345_CONVX_CASE(48, 16)
    O2345_REQUIRE(false, "sparse_conv3d_x3: unsupported channels %d -> %d", cin, cout);
    return -1;
}

}  // extern "C"
