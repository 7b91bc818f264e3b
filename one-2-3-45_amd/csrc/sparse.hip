// Sparse cost-regularisation CNN engine (tsparse/modules.py:94-124,259-304 on torchsparse v1.4.0
// semantics, restated in oracle/recon.py).  Replaces torchsparse's hash table + per-offset gather-GEMM-scatter.
//
// MI355X design: every active coordinate lives on a dense lattice (<= 257^3 cells per level), so the "hash table"
// is a dense int32 index grid per level (row id or -1; 8 MB at 128^3, L2/MALL resident) and the kernel map is
// implicit: a conv is a gather-form implicit GEMM  out[q] = sum_k in[ nbr(q, k) ] * W[k]  with
//   mode 0 (stride 1)        nbr = cell(q) + o_k                      on the same level
//   mode 1 (stride 2, down)  nbr = 2*cell(q) + o_k                    on the finer level
//   mode 2 (transposed, up)  nbr = (cell(q) - o_k)/2 if all even      on the coarser level
// o_k in {-1,0,1}^3, k = (oz+1)*9 + (oy+1)*3 + (ox+1)  (x fastest, torchsparse odd-kernel order).
// Rows of a level are numbered in x-major lattice order, which is the order torch.unique gives torchsparse.
// BatchNorm uses batch statistics (the reference never calls .eval()): fp64 column sums in a deterministic
// two-stage reduction, then a fused normalise + ReLU (+ skip add) pass.
#include "common.h"

namespace o2345 {

struct Lattice {
    int nx, ny, nz;   // cells per axis of this level's index grid
};

// ---- coarse-level construction (spdownsample, kernel 3 stride 2) --------------------------------------------------
// fine cell c marks coarse cells (c+o)/2 for o in {-1,0,1} with (c+o) even, subject to (c+o) >= cmin (cell units)
__global__ __launch_bounds__(256) void k_coord_min(const int* __restrict__ coords /*[N,4]*/, int n, int ts, int* __restrict__ cmin /*[3]*/) {
    // grid-stride + wave + block reduction: three atomics per BLOCK (one per wave serialised ~18k atomics on one line: 0.23 ms)
    __shared__ int sm[4][3];
    int x = 1 << 30, y = 1 << 30, z = 1 << 30;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        x = min(x, c.x / ts); y = min(y, c.y / ts); z = min(z, c.z / ts);
    }
    for (int off = 32; off; off >>= 1) {
        x = min(x, __shfl_xor(x, off)); y = min(y, __shfl_xor(y, off)); z = min(z, __shfl_xor(z, off));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w][0] = x; sm[w][1] = y; sm[w][2] = z; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int m = min(min(sm[0][threadIdx.x], sm[1][threadIdx.x]), min(sm[2][threadIdx.x], sm[3][threadIdx.x]));
        atomicMin(cmin + threadIdx.x, m);
    }
}

__global__ void k_mark_coarse(const int* __restrict__ coords, int n, int ts, const int* __restrict__ cmin, Lattice lc,
                              uint8_t* __restrict__ flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c4 = reinterpret_cast<const int4*>(coords)[i];
    const int c[3] = {c4.x / ts, c4.y / ts, c4.z / ts};
    int cand[3][2], nc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        nc[a] = 0;
        if ((c[a] & 1) == 0) {
            if (c[a] >= cmin[a]) cand[a][nc[a]++] = c[a] >> 1;
        } else {
            if (c[a] - 1 >= cmin[a]) cand[a][nc[a]++] = (c[a] - 1) >> 1;
            cand[a][nc[a]++] = (c[a] + 1) >> 1;
        }
    }
    for (int ix = 0; ix < nc[0]; ++ix)
        for (int iy = 0; iy < nc[1]; ++iy)
            for (int iz = 0; iz < nc[2]; ++iz)
                flag[((size_t)cand[0][ix] * lc.ny + cand[1][iy]) * lc.nz + cand[2][iz]] = 1;
}

constexpr int IDX_BLOCK = 256;

__global__ __launch_bounds__(IDX_BLOCK) void k_flag_count(const uint8_t* __restrict__ flag, long long ncell,
                                                          int* __restrict__ block_tot) {
    __shared__ int wtot[IDX_BLOCK / 64];
    const long long v = (long long)blockIdx.x * IDX_BLOCK + threadIdx.x;
    int tot;
    (void)block_prefix<IDX_BLOCK / 64>(v < ncell && flag[v], wtot, tot);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_scan_small2(int* __restrict__ a, int n, int* __restrict__ total) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) {
        int v = a[i];
        a[i] = run;
        run += v;
    }
    if (t == 1023) *total = part[1023];
}

__global__ __launch_bounds__(IDX_BLOCK) void k_flag_assign(const uint8_t* __restrict__ flag, Lattice l, int ts,
                                                           const int* __restrict__ block_base,
                                                           int* __restrict__ row_of_cell, int* __restrict__ coords) {
    __shared__ int wtot[IDX_BLOCK / 64];
    const long long ncell = (long long)l.nx * l.ny * l.nz;
    const long long v = (long long)blockIdx.x * IDX_BLOCK + threadIdx.x;
    const bool keep = v < ncell && flag[v];
    int tot;
    const int p = block_prefix<IDX_BLOCK / 64>(keep, wtot, tot);
    if (v < ncell) {
        const int row = keep ? block_base[blockIdx.x] + p : -1;
        row_of_cell[v] = row;
        if (keep) {
            int z = (int)(v % l.nz), y = (int)((v / l.nz) % l.ny), x = (int)(v / ((long long)l.nz * l.ny));
            reinterpret_cast<int4*>(coords)[row] = make_int4(x * ts, y * ts, z * ts, 0);
        }
    }
}

// ---- gather-form sparse convolution -------------------------------------------------------------------------------
// One thread per output row, all COUT accumulators in registers; the 27 x CIN x COUT weights are wave-uniform
// (scalar loads); a neighbour row is CIN contiguous floats (dwordx4 gathers, L1/L2 hits for lattice neighbours).
template <int CIN, int COUT, int MODE>
__global__ __launch_bounds__(256) void k_sparse_conv(const float* __restrict__ in, const int* __restrict__ out_coords,
                                                     int n_out, int ts_out, const int* __restrict__ in_grid, Lattice lin,
                                                     const float* __restrict__ Wk /*[27,CIN,COUT]*/,
                                                     float* __restrict__ out /*[n_out,COUT]*/) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_out) return;
    const int4 c4 = reinterpret_cast<const int4*>(out_coords)[q];
    const int cx = c4.x / ts_out, cy = c4.y / ts_out, cz = c4.z / ts_out;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int k = 0; k < 27; ++k) {
        const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
        int nx, ny, nz;
        bool ok = true;
        if (MODE == 0) { nx = cx + ox; ny = cy + oy; nz = cz + oz; }
        else if (MODE == 1) { nx = 2 * cx + ox; ny = 2 * cy + oy; nz = 2 * cz + oz; }
        else {
            nx = cx - ox; ny = cy - oy; nz = cz - oz;
            ok = !((nx | ny | nz) & 1);
            nx >>= 1; ny >>= 1; nz >>= 1;
        }
        ok = ok && nx >= 0 && ny >= 0 && nz >= 0 && nx < lin.nx && ny < lin.ny && nz < lin.nz;
        int r = -1;
        if (ok) r = in_grid[((size_t)nx * lin.ny + ny) * lin.nz + nz];
        if (r < 0) continue;
        const float4* src = reinterpret_cast<const float4*>(in + (size_t)r * CIN);
        const float* w = Wk + (size_t)k * CIN * COUT;
#pragma unroll
        for (int i4 = 0; i4 < CIN / 4; ++i4) {
            const float4 v = src[i4];
            const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(xv[u], w[(i4 * 4 + u) * COUT + o], acc[o]);
        }
    }
    float4* dst = reinterpret_cast<float4*>(out + (size_t)q * COUT);
#pragma unroll
    for (int o4 = 0; o4 < COUT / 4; ++o4) dst[o4] = make_float4(acc[4 * o4], acc[4 * o4 + 1], acc[4 * o4 + 2], acc[4 * o4 + 3]);
}

// ---- batch-statistics BatchNorm + ReLU (+ skip) ---------------------------------------------------------------------
// stage 1: per-block fp64 partial sums of x and x^2 per channel; stage 2: one block reduces the partials in a fixed
// order -> scale/shift; stage 3: y = relu(x*scale + shift) [+ skip].  Deterministic (no float atomics).
template <int C>
__global__ __launch_bounds__(256) void k_col_partial(const float* __restrict__ x, int n, double* __restrict__ part /*[nb,2,C]*/) {
    __shared__ double sm[256 / C][2][C];
    const int c = threadIdx.x % C, lane_row = threadIdx.x / C;
    constexpr int RPB = 256 / C;             // rows per pass
    double s = 0.0, s2 = 0.0;
    for (long long r = (long long)blockIdx.x * RPB + lane_row; r < n; r += (long long)gridDim.x * RPB) {
        const double v = (double)x[r * C + c];
        s += v; s2 += v * v;
    }
    sm[lane_row][0][c] = s; sm[lane_row][1][c] = s2;
    __syncthreads();
    if (lane_row == 0) {
        for (int i = 1; i < RPB; ++i) { s += sm[i][0][c]; s2 += sm[i][1][c]; }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = s2;
    }
}

// one block per channel: thread t sums every 256-th partial in index order, then a fixed tree over the 256 sub-sums (wave shuffles, 4 wave
// totals in LDS): deterministic, and ~4 us instead of the 13-19 us of a single block walking all partials of all channels
template <int C>
__global__ __launch_bounds__(256) void k_col_finish(const double* __restrict__ part, int nblocks, int n, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, int abs_gamma,
                                                    float* __restrict__ scale_shift /*[2,C]*/, float* __restrict__ mean_var /*[2,C] or null*/) {
    __shared__ double sm[2][4];
    const int c = blockIdx.x;
    double s = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { s += part[((size_t)b * 2 + 0) * C + c]; s2 += part[((size_t)b * 2 + 1) * C + c]; }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s; sm[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    s2 = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
    const double mean = s / n;
    double var = s2 / n - mean * mean;          // biased batch variance
    if (var < 0.0) var = 0.0;
    float g = gamma[c];
    if (abs_gamma) g = fabsf(g) + eps;          // inplace_abn convention
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    scale_shift[c] = g * inv;
    scale_shift[C + c] = beta[c] - (float)mean * g * inv;
    if (mean_var) { mean_var[c] = (float)mean; mean_var[C + c] = (float)var; }
}

// x: [n, C] rows (channel-last).  slope = 0 -> ReLU, 0.01 -> leaky ReLU (InPlaceABN).  skip may be null.
template <int C>
__global__ __launch_bounds__(256) void k_bn_act(const float* __restrict__ x, long long n_elems,
                                                const float* __restrict__ scale_shift, float slope,
                                                const float* __restrict__ skip, float* __restrict__ y) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n_elems) return;
    const int c = (int)(i % C);
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float t = r[u] * scale_shift[c + u] + scale_shift[C + c + u];
        t = t >= 0.f ? t : t * slope;
        r[u] = t;
    }
    if (skip) {
        const float4 s = *reinterpret_cast<const float4*>(skip + i);
        r[0] += s.x; r[1] += s.y; r[2] += s.z; r[3] += s.w;
    }
    *reinterpret_cast<float4*>(y + i) = make_float4(r[0], r[1], r[2], r[3]);
}

// channel-first variant for InPlaceABN on [V,C,H,W] feature maps: stats over (V,H,W) per channel
__global__ __launch_bounds__(256) void k_nchw_partial(const float* __restrict__ x, int V, int C, long long HW,
                                                      double* __restrict__ part /*[C, nb_per_c, 2]*/, int nb_per_c) {
    __shared__ double sm[2][4];
    const int c = blockIdx.y, b = blockIdx.x;
    double s = 0.0, s2 = 0.0;
    const long long tot = (long long)V * HW;
    for (long long i = (long long)b * 256 + threadIdx.x; i < tot; i += (long long)nb_per_c * 256) {
        const long long v = i / HW, p = i % HW;
        const double t = (double)x[(v * C + c) * HW + p];
        s += t; s2 += t * t;
    }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s; sm[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)c * nb_per_c + b) * 2 + 0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        part[((size_t)c * nb_per_c + b) * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    }
}

__global__ void k_nchw_finish(const double* __restrict__ part, int nb_per_c, long long count, int C,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int abs_gamma,
                              float* __restrict__ scale_shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, s2 = 0.0;
    for (int b = 0; b < nb_per_c; ++b) { s += part[((size_t)c * nb_per_c + b) * 2]; s2 += part[((size_t)c * nb_per_c + b) * 2 + 1]; }
    const double mean = s / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float g = gamma[c];
    if (abs_gamma) g = fabsf(g) + eps;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    scale_shift[c] = g * inv;
    scale_shift[C + c] = beta[c] - (float)mean * g * inv;
}

// fused normalise + leaky ReLU + NCHW -> NHWC re-layout (the cost-volume gather wants channel-last maps)
template <int C>
__global__ __launch_bounds__(256) void k_abn_apply_nhwc(const float* __restrict__ x /*[V,C,HW]*/,
                                                        const float* __restrict__ scale_shift, float slope, int HW,
                                                        float* __restrict__ y_nchw /*or null*/, float* __restrict__ y_nhwc /*or null*/) {
    __shared__ float tile[C][65];
    const int v = blockIdx.y, p0 = blockIdx.x * 64;
    const float* src = x + (size_t)v * C * HW;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int c = i / 64, p = i % 64;
        float t = 0.f;
        if (p0 + p < HW) {
            t = src[(size_t)c * HW + p0 + p] * scale_shift[c] + scale_shift[C + c];
            t = t >= 0.f ? t : t * slope;
            if (y_nchw) y_nchw[((size_t)v * C + c) * HW + p0 + p] = t;
        }
        tile[c][p] = t;
    }
    if (!y_nhwc) return;
    __syncthreads();
    float* dst = y_nhwc + (size_t)v * HW * C;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int p = i / C, c = i % C;
        if (p0 + p < HW) dst[(size_t)(p0 + p) * C + c] = tile[c][p];
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

// Build the next-coarser level (stride 2, kernel 3) from the coordinates of the current one.
//   coords_fine [n_fine,4] int32 (x,y,z,b), multiples of ts;  coarse lattice = (nxc,nyc,nzc) cells of size 2*ts.
//   Outputs: row_of_cell [nxc*nyc*nzc], coords_coarse [capacity,4], n_coarse (device scalar).
//   workspace: flags (ncell bytes, 16-aligned) + block totals + 4 ints.
size_t o2345_sparse_downsample_workspace_bytes(int nxc, int nyc, int nzc) {
    long long ncell = (long long)nxc * nyc * nzc;
    return (size_t)((ncell + 15) / 16 * 16) + (size_t)(cdiv(ncell, IDX_BLOCK) + 16) * sizeof(int);
}

int o2345_sparse_downsample(const int32_t* coords_fine, int n_fine, int ts, int nxc, int nyc, int nzc,
                            int32_t* row_of_cell, int32_t* coords_coarse, int32_t* n_coarse_dev, void* workspace,
                            size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(coords_fine && row_of_cell && coords_coarse && n_coarse_dev && workspace, "sparse_downsample: null pointer");
    O2345_REQUIRE(workspace_bytes >= o2345_sparse_downsample_workspace_bytes(nxc, nyc, nzc), "sparse_downsample: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long long ncell = (long long)nxc * nyc * nzc;
    uint8_t* flag = (uint8_t*)workspace;
    int* block_tot = (int*)((char*)workspace + (ncell + 15) / 16 * 16);
    const unsigned nb = cdiv(ncell, IDX_BLOCK);
    int* cmin = block_tot + nb + 1;
    Lattice lc{nxc, nyc, nzc};
    O2345_HIP(hipMemsetAsync(flag, 0, ncell, s));
    O2345_HIP(hipMemsetAsync(cmin, 0x3f, 3 * sizeof(int), s));
    if (n_fine > 0) {
        hipLaunchKernelGGL(k_coord_min, dim3(cdiv(n_fine, 256) < 512 ? cdiv(n_fine, 256) : 512), dim3(256), 0, s, coords_fine, n_fine, ts, cmin);
        hipLaunchKernelGGL(k_mark_coarse, dim3(cdiv(n_fine, 256)), dim3(256), 0, s, coords_fine, n_fine, ts, cmin, lc, flag);
    }
    hipLaunchKernelGGL(k_flag_count, dim3(nb), dim3(IDX_BLOCK), 0, s, flag, ncell, block_tot);
    hipLaunchKernelGGL(k_scan_small2, dim3(1), dim3(1024), 0, s, block_tot, (int)nb, n_coarse_dev);
    hipLaunchKernelGGL(k_flag_assign, dim3(nb), dim3(IDX_BLOCK), 0, s, flag, lc, 2 * ts, block_tot, row_of_cell, coords_coarse);
    return check_launch("sparse_downsample");
}

#define O2345_CONV_CASE(CI, CO)                                                                                        \
    if (cin == CI && cout == CO) {                                                                                     \
        if (mode == 0) hipLaunchKernelGGL((k_sparse_conv<CI, CO, 0>), grid, dim3(256), 0, s, in, out_coords, n_out, ts_out, in_grid, lin, kernel, out); \
        else if (mode == 1) hipLaunchKernelGGL((k_sparse_conv<CI, CO, 1>), grid, dim3(256), 0, s, in, out_coords, n_out, ts_out, in_grid, lin, kernel, out); \
        else hipLaunchKernelGGL((k_sparse_conv<CI, CO, 2>), grid, dim3(256), 0, s, in, out_coords, n_out, ts_out, in_grid, lin, kernel, out); \
        return check_launch("sparse_conv3d");                                                                          \
    }

// mode 0: stride 1 (in level == out level), 1: stride-2 down (in = finer level), 2: transposed stride-2 up (in = coarser)
int o2345_sparse_conv3d(int mode, const float* in, int cin, const int32_t* in_grid, int gx, int gy, int gz,
                        const int32_t* out_coords, int n_out, int ts_out, const float* kernel, int cout, float* out,
                        void* stream) {
    O2345_REQUIRE(in && in_grid && out_coords && kernel && out, "sparse_conv3d: null pointer");
    O2345_REQUIRE(mode >= 0 && mode <= 2, "sparse_conv3d: bad mode %d", mode);
    if (n_out == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Lattice lin{gx, gy, gz};
    dim3 grid(cdiv(n_out, 256));
    O2345_CONV_CASE(32, 16) O2345_CONV_CASE(16, 16) O2345_CONV_CASE(16, 32) O2345_CONV_CASE(32, 32)
    O2345_CONV_CASE(32, 64) O2345_CONV_CASE(64, 64) O2345_CONV_CASE(64, 32) O2345_CONV_CASE(48, 16)
    O2345_REQUIRE(false, "sparse_conv3d: unsupported channels %d -> %d", cin, cout);
    return -1;
}

// y = act(batchnorm_batchstats(x)) [+ skip] on rows [n, C]; workspace >= o2345_bn_workspace_bytes(C)
size_t o2345_bn_workspace_bytes(int C) { return (size_t)1024 * 2 * C * sizeof(double) + 4 * C * sizeof(float); }

int o2345_bn_act_rows(const float* x, int n, int C, const float* gamma, const float* beta, float eps, float slope,
                      int abs_gamma, const float* skip, float* y, float* mean_var_out, void* workspace,
                      size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(x && gamma && beta && y && workspace, "bn_act_rows: null pointer");
    O2345_REQUIRE(C == 16 || C == 32 || C == 64, "bn_act_rows: C must be 16/32/64 (got %d)", C);
    O2345_REQUIRE(workspace_bytes >= o2345_bn_workspace_bytes(C), "bn_act_rows: workspace too small");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* ss = (float*)(part + (size_t)1024 * 2 * C);
    const int rpb = 256 / C;
    int nb = (int)cdiv(n, rpb * 8);
    if (nb > 1024) nb = 1024;
    const long long ne = (long long)n * C;
#define O2345_BN_CASE(CC)                                                                                             \
    if (C == CC) {                                                                                                    \
        hipLaunchKernelGGL(k_col_partial<CC>, dim3(nb), dim3(256), 0, s, x, n, part);                                 \
        hipLaunchKernelGGL(k_col_finish<CC>, dim3(CC), dim3(256), 0, s, part, nb, n, gamma, beta, eps, abs_gamma, ss, mean_var_out); \
        hipLaunchKernelGGL(k_bn_act<CC>, dim3(cdiv(ne, 1024)), dim3(256), 0, s, x, ne, ss, slope, skip, y);          \
    }
    O2345_BN_CASE(16) O2345_BN_CASE(32) O2345_BN_CASE(64)
    return check_launch("bn_act_rows");
}

// InPlaceABN forward (training-mode statistics, leaky ReLU) on NCHW maps; writes NCHW and/or NHWC outputs.
size_t o2345_abn_workspace_bytes(int C) { return (size_t)C * 64 * 2 * sizeof(double) + 2 * C * sizeof(float); }

int o2345_abn_nchw(const float* x, int V, int C, int H, int W, const float* gamma, const float* beta, float eps,
                   float slope, int abs_gamma, float* y_nchw, float* y_nhwc, void* workspace, size_t workspace_bytes,
                   void* stream) {
    O2345_REQUIRE(x && gamma && beta && workspace, "abn_nchw: null pointer");
    O2345_REQUIRE(C == 8 || C == 16 || C == 32, "abn_nchw: C must be 8, 16 or 32 (got %d)", C);
    O2345_REQUIRE(workspace_bytes >= o2345_abn_workspace_bytes(C), "abn_nchw: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nbc = 64;
    const long long HW = (long long)H * W;
    double* part = (double*)workspace;
    float* ss = (float*)(part + (size_t)C * nbc * 2);
    hipLaunchKernelGGL(k_nchw_partial, dim3(nbc, C), dim3(256), 0, s, x, V, C, HW, part, nbc);
    hipLaunchKernelGGL(k_nchw_finish, dim3(1), dim3(64), 0, s, part, nbc, (long long)V * HW, C, gamma, beta, eps, abs_gamma, ss);
    dim3 grid(cdiv(HW, 64), V);
    if (C == 32) hipLaunchKernelGGL(k_abn_apply_nhwc<32>, grid, dim3(256), 0, s, x, ss, slope, (int)HW, y_nchw, y_nhwc);
    else if (C == 16) hipLaunchKernelGGL(k_abn_apply_nhwc<16>, grid, dim3(256), 0, s, x, ss, slope, (int)HW, y_nchw, y_nhwc);
    else hipLaunchKernelGGL(k_abn_apply_nhwc<8>, grid, dim3(256), 0, s, x, ss, slope, (int)HW, y_nchw, y_nhwc);
    return check_launch("abn_nchw");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_sparse() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_coord_min));
}
}  // namespace o2345
