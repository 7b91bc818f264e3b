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

// ---- brick form of the finest same-resolution layer (32 -> 16 channels, MODE 0) ---------------------------------------------------------------
// The gather form above fetches every neighbour row once per (output row, offset): 27 x per input row, 4 GB of 16-byte-per-lane gathers for
// 150 MB of rows at 128^3 (0.48 ms, bound by the L1's access rate; profiles/r02_pmc_sparse_conv.json).  Here a workgroup owns a 4 x 4 x 16 brick of
// lattice sites: it stages the brick's 6 x 6 x 18 halo of input rows in LDS ONCE -- already split into the f16 hi | lo halves the matrix cores
// consume, so the operand split is also done once per input row instead of 27 times -- and every output site of the brick takes its 27 neighbours
// from LDS.  D[cout 16][site 16] on v_mfma_f32_16x16x32_f16: a wave owns a z-run of 16 sites (column n = lane & 15), one k step = one neighbour
// offset x all 32 input channels (lane group g = lane >> 4 supplies channels 8g .. 8g+7: one conflict-free ds_read_b128 per half), three matrix
// instructions per offset.  The weights are the same blob as above, re-indexed while they are staged (a 16 x 32 A fragment of lane (m, g) is the
// 32 x 16 fragment of lane m + 32 (g & 1) of channel group g >> 1).
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA16_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
constexpr int BRX = 4, BRY = 4, BRZ = 16, HX = BRX + 2, HY = BRY + 2, HZ = BRZ + 2, HALO = HX * HY * HZ;     // 648 sites
constexpr int SITE_F = 36;                    // floats per staged site: 16 (hi, 32 f16) + 16 (lo) + 4 pad -> 144-byte stride, conflict-free b128 reads
constexpr int BRICK_THREADS = 512;
constexpr int BRICK_W_F = 27 * 2 * 64 * 4;   // weights in LDS: [27][hi|lo][64 lanes][8 f16]
constexpr size_t BRICK_LDS = (size_t)(BRICK_W_F + HALO * SITE_F) * 4;

__global__ __launch_bounds__(BRICK_THREADS) void k_sparse_conv_brick_32_16(const float* __restrict__ in, const int* __restrict__ grid, Lattice3 lin,
                                                                            const float* __restrict__ wblob, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* hl = lds + BRICK_W_F;
    {   // weights: dst[(k * 2 + half) * 64 + lane (m, g)] = src[((k * 2 + (g >> 1)) * 2 + half) * 64 + m + 32 * (g & 1)]   (float4 units)
        const float4* src = reinterpret_cast<const float4*>(wblob);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = threadIdx.x; i < 27 * 2 * 64; i += blockDim.x) {
            const int l = i & 63, half = (i >> 6) & 1, k = i >> 7;
            const int m = l & 15, g = l >> 4;
            dst[i] = (m < 16) ? src[((k * 2 + (g >> 1)) * 2 + half) * 64 + m + 32 * (g & 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float m1 = -1.f;
    asm volatile("" : "+v"(m1));
    const int lane = threadIdx.x & 63, n16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int nbx = (lin.nx + BRX - 1) / BRX, nby = (lin.ny + BRY - 1) / BRY, nbz = (lin.nz + BRZ - 1) / BRZ;
    const int nbricks = nbx * nby * nbz;
    const float4* WL = reinterpret_cast<const float4*>(wl) + lane;
    for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
        const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
        const int x0 = bx * BRX, y0 = by * BRY, z0 = bz * BRZ;
        // any occupied output site in the brick?
        int occ = 0;
        if (threadIdx.x < BRX * BRY * BRZ) {
            const int t = threadIdx.x, z = z0 + t % BRZ, y = y0 + (t / BRZ) % BRY, x = x0 + t / (BRZ * BRY);
            occ = (x < lin.nx && y < lin.ny && z < lin.nz) ? (grid[((size_t)x * lin.ny + y) * lin.nz + z] >= 0) : 0;
        }
        if (!__syncthreads_or(occ)) continue;        // (also the barrier that protects the halo buffer of the previous brick)
        // ---- stage the halo: item = (site, 8-channel chunk); hi | lo halves, zeros for empty / outside sites.  Three batched phases so that the
        // dependent loads (index grid -> row) of a thread's items are in flight together instead of one item at a time
        constexpr int NIT = (HALO * 4 + BRICK_THREADS - 1) / BRICK_THREADS;
        int rr[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = threadIdx.x + i * BRICK_THREADS;
            const int site = it >> 2;
            const int hz = site % HZ, hy = (site / HZ) % HY, hx = site / (HZ * HY);
            const int x = x0 + hx - 1, y = y0 + hy - 1, z = z0 + hz - 1;
            rr[i] = (it < HALO * 4 && x >= 0 && y >= 0 && z >= 0 && x < lin.nx && y < lin.ny && z < lin.nz) ? grid[((size_t)x * lin.ny + y) * lin.nz + z] : -1;
        }
        float4 va[NIT], vb[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c8 = (threadIdx.x + i * BRICK_THREADS) & 3;
            va[i] = make_float4(0.f, 0.f, 0.f, 0.f); vb[i] = va[i];
            if (rr[i] >= 0) {
                const float4* src = reinterpret_cast<const float4*>(in + (size_t)rr[i] * 32) + 2 * c8;
                va[i] = src[0]; vb[i] = src[1];
            }
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = threadIdx.x + i * BRICK_THREADS;
            if (it >= HALO * 4) break;
            const int site = it >> 2, c8 = it & 3;
            union { h16x8 v8; h16x2 v2[4]; hh16x2 w2[4]; float4 f4; } bh, bl;
            const float xv[8] = {va[i].x, va[i].y, va[i].z, va[i].w, vb[i].x, vb[i].y, vb[i].z, vb[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh.v2[j] = __builtin_amdgcn_cvt_pkrtz(xv[2 * j], xv[2 * j + 1]);
                bl.v2[j] = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)bh.w2[j][0], m1, xv[2 * j]), __builtin_fmaf((float)bh.w2[j][1], m1, xv[2 * j + 1]));
            }
            float4* d = reinterpret_cast<float4*>(hl + site * SITE_F);
            d[c8] = bh.f4;
            d[4 + c8] = bl.f4;
        }
        __syncthreads();
        // ---- compute: 16 z-runs, two per wave ---------------------------------------------------------------------------------------------
        for (int run = wave; run < BRX * BRY; run += BRICK_THREADS / 64) {
            const int rx = run / BRY, ry = run % BRY;
            const int x = x0 + rx, y = y0 + ry, z = z0 + n16;
            const int q = (x < lin.nx && y < lin.ny && z < lin.nz) ? grid[((size_t)x * lin.ny + y) * lin.nz + z] : -1;
            if (__ballot(q >= 0) == 0ull) continue;
            f32x4v acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;      // three independent accumulation chains (the three partial products)
            // halo coordinates of this lane's site: (rx + 1, ry + 1, n16 + 1); neighbour k adds (ox, oy, oz)
            const float4* base = reinterpret_cast<const float4*>(hl + (((rx + 1) * HY + (ry + 1)) * HZ + (n16 + 1)) * SITE_F) + g;
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
                const float4* nb = base + ((ox * HY + oy) * HZ + oz) * (SITE_F / 4);
                const h16x8 bhi = __builtin_bit_cast(h16x8, nb[0]), blo = __builtin_bit_cast(h16x8, nb[4]);
                const h16x8 ahi = __builtin_bit_cast(h16x8, WL[(k * 2 + 0) * 64]), alo = __builtin_bit_cast(h16x8, WL[(k * 2 + 1) * 64]);
                acc1 = MFMA16_F16(alo, bhi, acc1);
                acc2 = MFMA16_F16(ahi, blo, acc2);
                acc = MFMA16_F16(ahi, bhi, acc);
            }
            acc += acc1 + acc2;
            if (q >= 0) *reinterpret_cast<float4*>(out + (size_t)q * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_sparse_conv_x3_blob_floats(int cin, int cout) { return 27 * (cin / 16) * ((cout + 31) / 32) * 2 * 256; }

#define O2345_CONVX_LAUNCH(CI, CO, MD, LW)                                                                             \
    {                                                                                                                   \
        if (LW) O2345_ENSURE_LDS((k_sparse_conv_x3<CI, CO, MD, LW>), lds);                                                \
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
// (o2345_sparse_conv_x3_blob_floats(cin, cout) floats).  identity_rows: the caller guarantees output row q = input row q (out_coords IS the list in_grid
// was built from) -- the precondition of the brick form, which writes out[in_grid[site]] and never reads out_coords (include/o2345.h).
int o2345_sparse_conv3d_x3(int mode, const float* in, int cin, const int32_t* in_grid, int gx, int gy, int gz,
                           const int32_t* out_coords, int n_out, int ts_out, const float* wblob, int cout, int identity_rows, float* out, void* stream) {
    O2345_REQUIRE(in && in_grid && out_coords && wblob && out, "sparse_conv3d_x3: null pointer");
    O2345_REQUIRE(mode >= 0 && mode <= 2, "sparse_conv3d_x3: bad mode %d", mode);
    if (n_out == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Lattice3 lin{gx, gy, gz};
    dim3 grid(cdiv(n_out, 128));
    const int n_cu = cu_count();
    const unsigned want = cdiv(n_out, 32 * 16);                       // 16 tiles (waves) per persistent workgroup round
    dim3 pgrid(want < (unsigned)n_cu ? want : (unsigned)n_cu);
    {
        // finest same-resolution layer: LDS-tiled brick form, ONLY under the caller's identity-row guarantee (O2345_SPARSE_BRICK=0 keeps the gather form: A/B runs)
        if (mode == 0 && cin == 32 && cout == 16 && identity_rows && knobs().sparse_brick) {
            const int nbricks = ((gx + BRX - 1) / BRX) * ((gy + BRY - 1) / BRY) * ((gz + BRZ - 1) / BRZ);
            O2345_ENSURE_LDS(k_sparse_conv_brick_32_16, BRICK_LDS);
            hipLaunchKernelGGL(k_sparse_conv_brick_32_16, dim3(nbricks < n_cu ? nbricks : n_cu), dim3(BRICK_THREADS), BRICK_LDS, s, in, in_grid, lin, wblob, out);
            return check_launch("sparse_conv3d_x3 (brick form)");
        }
    }
    O2345_CONVX_CASE(32, 16) O2345_CONVX_CASE(16, 16) O2345_CONVX_CASE(16, 32) O2345_CONVX_CASE(32, 32)
    O2345_CONVX_CASE(32, 64) O2345_CONVX_CASE(64, 64) O2345_CONVX_CASE(64, 32) O2345_CONVX_CASE(48, 16)
    O2345_REQUIRE(false, "sparse_conv3d_x3: unsupported channels %d -> %d", cin, cout);
    return -1;
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_sparse_mfma() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_sparse_conv_brick_32_16));
}
}  // namespace o2345
