// Cost-volume construction for gfx950.
//
// Reference path: sparse_sdf_network.py:286-400 = generate_grid -> back_project_sparse_type(only_mask) ->
// keep voxels seen by > min_views -> back_project_sparse_type -> aggregate_multiview_features -> (CNN) ->
// sparse_to_dense_volume.  The reference materialises [N,V,C] and makes three passes over it; here:
//
//   costvol_index  : per voxel, project into the V views (matrices in SGPRs, no memory traffic except the matrices),
//                    count the views that see it, then an order-preserving compaction (ballot/popcount prefix per
//                    wave, block totals, one small scan) -> row_of_voxel[D^3], coords[N,4], N.
//   costvol_gather : per kept voxel, 4 lanes x 4 channels: each bilinear tap of the channel-last feature map
//                    [V,H,W,16] is one 64-byte segment read by a lane quad (dwordx4 per lane); running sum /
//                    sum-of-squares in registers over ALL views, write var|mean as two dwordx4.
//   scatter_dense  : rows -> dense volume in both layouts (channel-last for our samplers, channel-first for the
//                    reference API) + float mask, one pass, no memset.
#include "common.h"
#include "geom_math.h"
#include "costvol_math.h"

namespace o2345 {

constexpr int IDX_BLOCK = 256;

// ---- pass 1a: visible-view count per voxel + per-block number of kept voxels -----------------------------------
__global__ __launch_bounds__(IDX_BLOCK) void k_vis_count(const float* __restrict__ proj, int V, int H, int W, VolGeom g,
                                                         int min_views, uint8_t* __restrict__ cnt,
                                                         int* __restrict__ block_tot) {
    __shared__ int wtot[IDX_BLOCK / 64];
    const long long nvox = (long long)g.dx * g.dy * g.dz;
    const long long v = (long long)blockIdx.x * IDX_BLOCK + threadIdx.x;
    int c = 0;
    if (v < nvox) {
        int x, y, z;
        voxel_xyz(v, g, x, y, z);
        c = visible_views(proj, V, H, W, g, x, y, z);
        cnt[v] = (uint8_t)c;
    }
    int tot;
    (void)block_prefix<IDX_BLOCK / 64>(v < nvox && c > min_views, wtot, tot);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

// ---- generic: exclusive scan of n ints by ONE 1024-thread block (n <= a few 100k block totals) -----------------
__global__ __launch_bounds__(1024) void k_scan_small(int* __restrict__ a, int n, int* __restrict__ total) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // Hillis-Steele inclusive scan over the 1024 partials
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

// ---- pass 1b: assign rows in voxel (x-major) order ---------------------------------------------------------------
__global__ __launch_bounds__(IDX_BLOCK) void k_vis_assign(const uint8_t* __restrict__ cnt, VolGeom g, int min_views,
                                                          const int* __restrict__ block_base,
                                                          int* __restrict__ row_of_voxel, int* __restrict__ coords) {
    __shared__ int wtot[IDX_BLOCK / 64];
    const long long nvox = (long long)g.dx * g.dy * g.dz;
    const long long v = (long long)blockIdx.x * IDX_BLOCK + threadIdx.x;
    const bool keep = v < nvox && (int)cnt[v] > min_views;
    int tot;
    const int p = block_prefix<IDX_BLOCK / 64>(keep, wtot, tot);
    if (v < nvox) {
        const int row = keep ? block_base[blockIdx.x] + p : -1;
        row_of_voxel[v] = row;
        if (keep) {
            int x, y, z;
            voxel_xyz(v, g, x, y, z);
            reinterpret_cast<int4*>(coords)[row] = make_int4(x, y, z, 0);   // (x,y,z,batch) as SparseTensor wants
        }
    }
}

// ---- pass 2: gather + variance/mean aggregation -------------------------------------------------------------------
// One lane quad per kept voxel; lane q owns channels 4q..4q+3, so one bilinear tap of the channel-last map is one
// 64-byte segment per quad.  The projection + tap set-up (~90 VALU ops incl. two IEEE divisions that decide the frustum
// test exactly) is NOT repeated by the four lanes: lane q prepares view 4i+q and the quad exchanges the four
// (index, weight) tap sets with DPP quad broadcasts -- the kernel is VALU / L2-gather co-limited, see DESIGN.md.
template <int K>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, false); }
template <int K>
__device__ __forceinline__ float quad_bcast_f(float v) { return __builtin_bit_cast(float, quad_bcast_i<K>(__builtin_bit_cast(int, v))); }

__device__ __forceinline__ void tap4_accumulate(const float4* __restrict__ base, const int (&idx)[4], const float (&w)[4],
                                                float4& s1, float4& s2) {
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (w[k] != 0.f) {              // fully-outside taps: no load (zero padding)
#ifdef O2345_COSTVOL_TAPMASK      // timing experiment only (wrong results): every tap falls into a few L1-resident lines -> what is left is issue + VALU
            const float4 a = base[(size_t)(idx[k] & O2345_COSTVOL_TAPMASK) * 4];
#else
            const float4 a = base[(size_t)idx[k] * 4];
#endif
            f.x += a.x * w[k]; f.y += a.y * w[k]; f.z += a.z * w[k]; f.w += a.w * w[k];
        }
    }
    s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
    s2.x += f.x * f.x; s2.y += f.y * f.y; s2.z += f.z * f.z; s2.w += f.w * f.w;
}

__device__ __forceinline__ void costvol_row_quad16(const float* __restrict__ feats, const float* __restrict__ proj, int V, int H, int W,
                                                   const VolGeom& g, const uint8_t* __restrict__ cnt, const int* __restrict__ coords,
                                                   int row, int q, float* __restrict__ out) {
    const int4 c = reinterpret_cast<const int4*>(coords)[row];
    const float wx = (float)c.x * g.voxel_size + g.ox, wy = (float)c.y * g.voxel_size + g.oy, wz = (float)c.z * g.voxel_size + g.oz;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const size_t plane = (size_t)H * W;
    for (int vb = 0; vb < V; vb += 4) {
        int idx[4] = {0, 0, 0, 0};
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (vb + q < V) {
            float gx, gy;
            bool ok;
            project_voxel(proj + 16 * (vb + q), wx, wy, wz, H, W, gx, gy, ok);
            const Taps2D tp = bilinear_taps(gx, gy, H, W);
#pragma unroll
            for (int k = 0; k < 4; ++k) { idx[k] = tp.idx[k]; w[k] = tp.w[k]; }
        }
#ifndef O2345_COSTVOL_BATCH
#define O2345_COSTVOL_BATCH 0      // measured (round 3): 0.166 ms batched vs 0.139 ms one view at a time -- see the comment below and profiles/NOTES.md
#endif
#if O2345_COSTVOL_BATCH
        // A/B form: the taps of the round's four views requested TOGETHER (up to 16 independent 16-byte loads per lane in flight), accumulated afterwards in
        // the same order (bit-identical sums).  If the kernel were bound by the latency of its dependent gathers this would win; it LOSES (130 registers,
        // 3 instead of 8 waves per SIMD): the kernel is bound by what the L2 -> L1 path delivers for 64-byte taps, and more waves beat more loads per wave.
        float4 tv[4][4];
        float wv[4][4];
#define O2345_QUAD_LOAD(K)                                                                                             \
        {                                                                                                              \
            const float4* base = reinterpret_cast<const float4*>(feats + (size_t)(vb + K < V ? vb + K : 0) * plane * 16) + q;   \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                         \
                const int ik = quad_bcast_i<K>(idx[k]);                                                                \
                wv[K][k] = (vb + K < V) ? quad_bcast_f<K>(w[k]) : 0.f;                                                 \
                tv[K][k] = make_float4(0.f, 0.f, 0.f, 0.f);                                                            \
                if (wv[K][k] != 0.f) tv[K][k] = base[(size_t)ik * 4];                                                  \
            }                                                                                                          \
        }
        O2345_QUAD_LOAD(0) O2345_QUAD_LOAD(1) O2345_QUAD_LOAD(2) O2345_QUAD_LOAD(3)
#undef O2345_QUAD_LOAD
#pragma unroll
        for (int K = 0; K < 4; ++K)
            if (vb + K < V) {
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (wv[K][k] != 0.f) {
                        const float4 a = tv[K][k];
                        f.x += a.x * wv[K][k]; f.y += a.y * wv[K][k]; f.z += a.z * wv[K][k]; f.w += a.w * wv[K][k];
                    }
                s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
                s2.x += f.x * f.x; s2.y += f.y * f.y; s2.z += f.z * f.z; s2.w += f.w * f.w;
            }
#else
#define O2345_QUAD_VIEW(K)                                                                                             \
        if (vb + K < V) {                                                                                              \
            int ik[4]; float wk[4];                                                                                    \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) { ik[k] = quad_bcast_i<K>(idx[k]); wk[k] = quad_bcast_f<K>(w[k]); } \
            tap4_accumulate(reinterpret_cast<const float4*>(feats + (size_t)(vb + K) * plane * 16) + q, ik, wk, s1, s2); \
        }
        O2345_QUAD_VIEW(0) O2345_QUAD_VIEW(1) O2345_QUAD_VIEW(2) O2345_QUAD_VIEW(3)
#undef O2345_QUAD_VIEW
#endif
    }
    const long long v = ((long long)c.x * g.dy + c.y) * g.dz + c.z;
    const float ic = 1.f / ((float)cnt[v] + 1e-5f);           // sparse_sdf_network.py:242
    float4 mean = make_float4(s1.x * ic, s1.y * ic, s1.z * ic, s1.w * ic);
    float4 var = make_float4(s2.x * ic - mean.x * mean.x, s2.y * ic - mean.y * mean.y, s2.z * ic - mean.z * mean.z, s2.w * ic - mean.w * mean.w);
    float4* o = reinterpret_cast<float4*>(out + (size_t)row * 32);
    o[q] = var;
    o[4 + q] = mean;
}

template <int C>
__global__ __launch_bounds__(256) void k_costvol_gather(const float* __restrict__ feats /*[V,H,W,C]*/,
                                                        const float* __restrict__ proj, int V, int H, int W, VolGeom g,
                                                        const uint8_t* __restrict__ cnt, const int* __restrict__ coords,
                                                        int n_rows, float* __restrict__ out /*[N,2C]*/) {
    constexpr int Q = C / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    int row = (int)(t / Q);
    const int q = (int)(t % Q);
    if (C == 16) {
        // whole quads stay together (DPP needs all four lanes); a tail quad re-does the last row (same values written)
        if (row >= n_rows) row = n_rows - 1;
        costvol_row_quad16(feats, proj, V, H, W, g, cnt, coords, row, q, out);
    } else {
        if (row >= n_rows) return;
        costvol_row<C>(feats, proj, V, H, W, g, cnt, coords, row, q, out);
    }
}

// ---- lod > 0: explicit voxel lists (children of the pruned lod-0 voxels, arbitrary order) ---------------------------
__global__ __launch_bounds__(256) void k_vis_count_list(const float* __restrict__ proj, int V, int H, int W, VolGeom g,
                                                        const int* __restrict__ coords, int n, uint8_t* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    cnt[i] = (uint8_t)visible_views(proj, V, H, W, g, c.x, c.y, c.z);
}

template <int C>
__global__ __launch_bounds__(256) void k_costvol_gather_list(const float* __restrict__ feats, const float* __restrict__ proj, int V,
                                                             int H, int W, VolGeom g, const uint8_t* __restrict__ cnt_row,
                                                             const int* __restrict__ coords, int n_rows, float* __restrict__ out) {
    constexpr int Q = C / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(t / Q), q = (int)(t % Q);
    if (row >= n_rows) return;
    costvol_row<C>(feats, proj, V, H, W, g, cnt_row, coords, row, q, out, true);
}

// dense index grid of an arbitrary coordinate list: grid[cell] = row (the "hash table" of the sparse-conv engine)
__global__ __launch_bounds__(256) void k_index_grid(const int* __restrict__ coords, int n, int ts, int nx, int ny, int nz,
                                                    int* __restrict__ grid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int x = c.x / ts, y = c.y / ts, z = c.z / ts;
    if (x >= 0 && y >= 0 && z >= 0 && x < nx && y < ny && z < nz) grid[((size_t)x * ny + y) * nz + z] = i;
}

// get_valid_sparse_coords_by_sdf (sparse_neus_renderer.py:838-848): |sdf| < thr, dilated by a (2r+1)^3 box, AND mask
__global__ __launch_bounds__(256) void k_prune_dilate(const float* __restrict__ sdf, const float* __restrict__ mask, int D, float thr,
                                                      int r, uint8_t* __restrict__ out) {
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= (long long)D * D * D) return;
    bool hit = false;
    if (mask[v] > 0.f) {
        const int z = (int)(v % D), y = (int)((v / D) % D), x = (int)(v / ((long long)D * D));
        for (int dx = -r; dx <= r && !hit; ++dx)
            for (int dy = -r; dy <= r && !hit; ++dy)
                for (int dz = -r; dz <= r; ++dz) {
                    const int a = x + dx, b = y + dy, c = z + dz;
                    if (a >= 0 && b >= 0 && c >= 0 && a < D && b < D && c < D && fabsf(sdf[((size_t)a * D + b) * D + c]) < thr) { hit = true; break; }
                }
    }
    out[v] = hit ? 1 : 0;
}

// ---- NCHW -> NHWC re-layout of the (compressed) feature maps: [V,C,H,W] -> [V,H,W,C] ---------------------------
// 64-pixel x C tile through LDS so that both the read and the write are coalesced.
template <int C>
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int HW) {
    __shared__ float tile[C][65];
    const int v = blockIdx.y;
    const int p0 = blockIdx.x * 64;
    const float* src = in + (size_t)v * C * HW;
    float* dst = out + (size_t)v * HW * C;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        int c = i / 64, p = i % 64;
        tile[c][p] = (p0 + p < HW) ? src[(size_t)c * HW + p0 + p] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        int p = i / C, c = i % C;
        if (p0 + p < HW) dst[(size_t)(p0 + p) * C + c] = tile[c][p];
    }
}

// ---- a7: rows -> dense volumes ------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_scatter_dense(const float* __restrict__ rows /*[N,C]*/,
                                                       const int* __restrict__ row_of_voxel, long long nvox,
                                                       float* __restrict__ dense_cl /*[D^3,C] or null*/,
                                                       float* __restrict__ dense_cf /*[C,D^3] or null*/,
                                                       float* __restrict__ mask /*[D^3] or null*/) {
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const int r = row_of_voxel[v];
    float4 f[C / 4];
#pragma unroll
    for (int i = 0; i < C / 4; ++i)
        f[i] = r >= 0 ? reinterpret_cast<const float4*>(rows + (size_t)r * C)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (mask) mask[v] = r >= 0 ? 1.f : 0.f;
    if (dense_cl) {
#pragma unroll
        for (int i = 0; i < C / 4; ++i) reinterpret_cast<float4*>(dense_cl + (size_t)v * C)[i] = f[i];
    }
    if (dense_cf) {
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            dense_cf[(size_t)(4 * i + 0) * nvox + v] = f[i].x;
            dense_cf[(size_t)(4 * i + 1) * nvox + v] = f[i].y;
            dense_cf[(size_t)(4 * i + 2) * nvox + v] = f[i].z;
            dense_cf[(size_t)(4 * i + 3) * nvox + v] = f[i].w;
        }
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

size_t o2345_costvol_workspace_bytes(int dx, int dy, int dz) {
    long long nvox = (long long)dx * dy * dz;
    return (size_t)(cdiv(nvox, IDX_BLOCK) + 16) * sizeof(int);
}

int o2345_costvol_index(const float* proj, int V, int H, int W, int dx, int dy, int dz, float voxel_size,
                        const float* origin_host, int min_views, uint8_t* cnt, int32_t* row_of_voxel, int32_t* coords,
                        int32_t* n_rows_dev, void* workspace, size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(proj && cnt && row_of_voxel && coords && n_rows_dev && workspace && origin_host, "costvol_index: null pointer");
    O2345_REQUIRE(V > 0 && V <= 255 && H > 1 && W > 1 && dx > 0 && dy > 0 && dz > 0, "costvol_index: bad sizes");
    O2345_REQUIRE(workspace_bytes >= o2345_costvol_workspace_bytes(dx, dy, dz), "costvol_index: workspace too small");
    VolGeom g{dx, dy, dz, voxel_size, origin_host[0], origin_host[1], origin_host[2]};
    const long long nvox = (long long)dx * dy * dz;
    const unsigned nb = cdiv(nvox, IDX_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    int* block_tot = (int*)workspace;
    hipLaunchKernelGGL(k_vis_count, dim3(nb), dim3(IDX_BLOCK), 0, s, proj, V, H, W, g, min_views, cnt, block_tot);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, block_tot, (int)nb, n_rows_dev);
    hipLaunchKernelGGL(k_vis_assign, dim3(nb), dim3(IDX_BLOCK), 0, s, cnt, g, min_views, block_tot, row_of_voxel, coords);
    return check_launch("costvol_index");
}

int o2345_costvol_gather(const float* feats_nhwc, const float* proj, int V, int H, int W, int C, int dx, int dy, int dz,
                         float voxel_size, const float* origin_host, const uint8_t* cnt, const int32_t* coords,
                         int n_rows, float* out_rows, void* stream) {
    O2345_REQUIRE(feats_nhwc && proj && cnt && coords && out_rows && origin_host, "costvol_gather: null pointer");
    O2345_REQUIRE(C == 16 || C == 8, "costvol_gather: C must be 8 or 16 (got %d)", C);
    if (n_rows == 0) return 0;
    VolGeom g{dx, dy, dz, voxel_size, origin_host[0], origin_host[1], origin_host[2]};
    hipStream_t s = (hipStream_t)stream;
    if (C == 16)
        hipLaunchKernelGGL(k_costvol_gather<16>, dim3(cdiv((long long)n_rows * 4, 256)), dim3(256), 0, s, feats_nhwc, proj,
                           V, H, W, g, cnt, coords, n_rows, out_rows);
    else
        hipLaunchKernelGGL(k_costvol_gather<8>, dim3(cdiv((long long)n_rows * 2, 256)), dim3(256), 0, s, feats_nhwc, proj,
                           V, H, W, g, cnt, coords, n_rows, out_rows);
    return check_launch("costvol_gather");
}

int o2345_visible_count_list(const float* proj, int V, int H, int W, float voxel_size, const float* origin_host,
                             const int32_t* coords, int n, uint8_t* cnt, void* stream) {
    O2345_REQUIRE(proj && coords && cnt && origin_host, "visible_count_list: null pointer");
    if (n == 0) return 0;
    VolGeom g{0, 0, 0, voxel_size, origin_host[0], origin_host[1], origin_host[2]};
    hipLaunchKernelGGL(k_vis_count_list, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, proj, V, H, W, g, coords, n, cnt);
    return check_launch("visible_count_list");
}

int o2345_costvol_gather_list(const float* feats_nhwc, const float* proj, int V, int H, int W, int C, float voxel_size,
                              const float* origin_host, const uint8_t* cnt_row, const int32_t* coords, int n_rows,
                              float* out_rows, void* stream) {
    O2345_REQUIRE(feats_nhwc && proj && cnt_row && coords && out_rows && origin_host, "costvol_gather_list: null pointer");
    O2345_REQUIRE(C == 16 || C == 8, "costvol_gather_list: C must be 8 or 16 (got %d)", C);
    if (n_rows == 0) return 0;
    VolGeom g{0, 0, 0, voxel_size, origin_host[0], origin_host[1], origin_host[2]};
    hipStream_t s = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL(k_costvol_gather_list<16>, dim3(cdiv((long long)n_rows * 4, 256)), dim3(256), 0, s, feats_nhwc, proj, V, H, W, g, cnt_row, coords, n_rows, out_rows);
    else hipLaunchKernelGGL(k_costvol_gather_list<8>, dim3(cdiv((long long)n_rows * 2, 256)), dim3(256), 0, s, feats_nhwc, proj, V, H, W, g, cnt_row, coords, n_rows, out_rows);
    return check_launch("costvol_gather_list");
}

int o2345_build_index_grid(const int32_t* coords, int n, int ts, int nx, int ny, int nz, int32_t* grid, void* stream) {
    O2345_REQUIRE(grid && (coords || n == 0), "build_index_grid: null pointer");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(grid, 0xff, (size_t)nx * ny * nz * sizeof(int), s);
    if (n > 0) hipLaunchKernelGGL(k_index_grid, dim3(cdiv(n, 256)), dim3(256), 0, s, coords, n, ts, nx, ny, nz, grid);
    return check_launch("build_index_grid");
}

int o2345_prune_dilate(const float* sdf, const float* mask, int D, float threshold, int radius, uint8_t* out, void* stream) {
    O2345_REQUIRE(sdf && mask && out && D > 0 && radius >= 0, "prune_dilate: bad arguments");
    hipLaunchKernelGGL(k_prune_dilate, dim3(cdiv((long long)D * D * D, 256)), dim3(256), 0, (hipStream_t)stream, sdf, mask, D, threshold, radius, out);
    return check_launch("prune_dilate");
}

int o2345_nchw_to_nhwc(const float* in, float* out, int V, int C, int H, int W, void* stream) {
    O2345_REQUIRE(in && out, "nchw_to_nhwc: null pointer");
    O2345_REQUIRE(C == 16 || C == 8 || C == 64, "nchw_to_nhwc: C must be 8, 16 or 64 (got %d)", C);
    dim3 grid(cdiv((long long)H * W, 64), V);
    hipStream_t s = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL(k_nchw_to_nhwc<16>, grid, dim3(256), 0, s, in, out, H * W);
    else if (C == 8) hipLaunchKernelGGL(k_nchw_to_nhwc<8>, grid, dim3(256), 0, s, in, out, H * W);
    else hipLaunchKernelGGL(k_nchw_to_nhwc<64>, grid, dim3(256), 0, s, in, out, H * W);
    return check_launch("nchw_to_nhwc");
}

int o2345_scatter_dense(const float* rows, const int32_t* row_of_voxel, int C, long long nvox, float* dense_cl,
                        float* dense_cf, float* mask, void* stream) {
    O2345_REQUIRE(rows && row_of_voxel, "scatter_dense: null pointer");
    O2345_REQUIRE(C == 16 || C == 8, "scatter_dense: C must be 8 or 16 (got %d)", C);
    hipStream_t s = (hipStream_t)stream;
    if (C == 16)
        hipLaunchKernelGGL(k_scatter_dense<16>, dim3(cdiv(nvox, 256)), dim3(256), 0, s, rows, row_of_voxel, nvox, dense_cl, dense_cf, mask);
    else
        hipLaunchKernelGGL(k_scatter_dense<8>, dim3(cdiv(nvox, 256)), dim3(256), 0, s, rows, row_of_voxel, nvox, dense_cl, dense_cf, mask);
    return check_launch("scatter_dense");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_costvol() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_scan_small));
}
}  // namespace o2345
