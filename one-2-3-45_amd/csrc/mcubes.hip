// Marching cubes on the device (call site sparse_neus_renderer.py:932-936, PyMCubes contract:
// vertices float64 [Nv,3] in index coordinates, shared per crossing grid edge, triangles [Nt,3]).
//
// Same conventions as the oracle (oracle/mc.c): corner inside when u <= iso; the vertex on the edge between grid
// point g and g - e_axis is "slot axis" of g; vertices are numbered by owning grid point (x-major) then slot x,y,z
// (= PyMCubes' creation order for surfaces that do not touch the volume boundary); triangles are emitted cell by
// cell in traversal order, table order inside a cell.  Deterministic: counts -> exclusive scans -> emit, no atomics.
// Everything stays on the device; u (67 MB at 256^3) is read three times, HBM-bound.
#include "common.h"
#include <math.h>
#include "mc_tables.h"

namespace o2345 {

__constant__ int8_t MC_TRI[256][16] = O2345_MC_TRI_TABLE_INIT;
__constant__ int8_t MC_FAR[12][3] = {{1,0,0},{1,1,0},{1,1,0},{0,1,0},{1,0,1},{1,1,1},{1,1,1},{0,1,1},{0,0,1},{1,0,1},{1,1,1},{0,1,1}};
__constant__ int8_t MC_AXIS[12] = {0,1,0,1,0,1,0,1,2,2,2,2};

constexpr int MC_ITEMS = 8;                 // grid points per thread
constexpr int MC_TILE = 256 * MC_ITEMS;     // per block

struct McGrid {
    int n0, n1, n2;
    long long s0, s1, n;
    float iso;        // classification threshold: the largest float <= iso_d, so that  u <= iso  <=>  (double)u <= iso_d  for every float u
    double iso_d;     // PyMCubes takes a DOUBLE isovalue: interpolation uses it unrounded
};

// host: McGrid for a double isovalue
static McGrid mc_grid(int n0, int n1, int n2, double iso) {
    float f = (float)iso;
    if ((double)f > iso) f = nextafterf(f, -INFINITY);
    return McGrid{n0, n1, n2, (long long)n1 * n2, (long long)n2, (long long)n0 * n1 * n2, f, iso};
}

// grid point p -> (x, y, z); 32-bit divisions whenever the volume has fewer than 2^32 points (a 64-bit division costs ~200 instructions)
__device__ __forceinline__ void mc_decompose(const McGrid& g, long long p, int& x, int& y, int& z) {
    if (g.n <= 0xFFFFFFFFll) {
        const unsigned up = (unsigned)p, q = up / (unsigned)g.n2;
        z = (int)(up - q * (unsigned)g.n2);
        x = (int)(q / (unsigned)g.n1);
        y = (int)(q - (unsigned)x * (unsigned)g.n1);
    } else {
        z = (int)(p % g.n2); y = (int)((p / g.n2) % g.n1); x = (int)(p / g.s0);
    }
}

__device__ __forceinline__ void mc_point_counts(const float* __restrict__ u, const McGrid& g, long long p, int x, int y, int z, int& nv, int& nt) {
    const bool in0 = u[p] <= g.iso;
    nv = 0;
    if (x > 0) nv += (in0 != (u[p - g.s0] <= g.iso));
    if (y > 0) nv += (in0 != (u[p - g.s1] <= g.iso));
    if (z > 0) nv += (in0 != (u[p - 1] <= g.iso));
    nt = 0;
    if (x + 1 < g.n0 && y + 1 < g.n1 && z + 1 < g.n2) {
        unsigned ci = in0 ? 1u : 0u;
        ci |= (u[p + g.s0] <= g.iso) << 1;
        ci |= (u[p + g.s0 + g.s1] <= g.iso) << 2;
        ci |= (u[p + g.s1] <= g.iso) << 3;
        ci |= (u[p + 1] <= g.iso) << 4;
        ci |= (u[p + g.s0 + 1] <= g.iso) << 5;
        ci |= (u[p + g.s0 + g.s1 + 1] <= g.iso) << 6;
        ci |= (u[p + g.s1 + 1] <= g.iso) << 7;
        while (nt < 5 && MC_TRI[ci][3 * nt] >= 0) ++nt;
        nt |= (int)ci << 8;
    }
}

// block-wide exclusive scan of one int per thread (256 threads)
__device__ __forceinline__ int block_scan_excl(int v, int* lds /*[5]*/, int& total) {
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if ((threadIdx.x & 63) >= off) inc += t;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) lds[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int t = lds[i]; if (i < w) base += t; tot += t; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_mc_count(const float* __restrict__ u, McGrid g, uint8_t* __restrict__ vcnt,
                                                  uint16_t* __restrict__ tcase, int* __restrict__ vblock, int* __restrict__ tblock) {
    __shared__ int lds[5];
    // item i of thread t is grid point tile + i * 256 + t: the lanes of a wave read consecutive floats (the counts only need the tile's totals,
    // so the order inside the tile is free here; k_mc_offsets scans the stored per-point counts in grid order)
    const long long p0 = (long long)blockIdx.x * MC_TILE + threadIdx.x;
    int sv = 0, st = 0;
#pragma unroll
    for (int i = 0; i < MC_ITEMS; ++i) {
        const long long p = p0 + i * 256;
        if (p < g.n) {
            int x, y, z, nv, nt;
            mc_decompose(g, p, x, y, z);
            mc_point_counts(u, g, p, x, y, z, nv, nt);
            vcnt[p] = (uint8_t)nv;
            tcase[p] = (uint16_t)nt;          // low byte: triangle count, high byte: cube index
            sv += nv; st += (nt & 0xff);
        }
    }
    int tv, tt;
    (void)block_scan_excl(sv, lds, tv);
    (void)block_scan_excl(st, lds, tt);
    if (threadIdx.x == 0) { vblock[blockIdx.x] = tv; tblock[blockIdx.x] = tt; }
}

__global__ __launch_bounds__(1024) void k_scan_small3(int* __restrict__ a, int n, long long* __restrict__ total) {
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
    for (int i = lo; i < hi; ++i) { int v = a[i]; a[i] = run; run += v; }
    if (t == 1023) *total = (long long)part[1023];
}

__global__ __launch_bounds__(256) void k_mc_offsets(McGrid g, const uint8_t* __restrict__ vcnt, const uint16_t* __restrict__ tcase,
                                                    const int* __restrict__ vblock, const int* __restrict__ tblock,
                                                    int* __restrict__ vbase, int* __restrict__ tbase) {
    __shared__ int lds[5];
    const long long p0 = (long long)blockIdx.x * MC_TILE + (long long)threadIdx.x * MC_ITEMS;
    int cv[MC_ITEMS], ct[MC_ITEMS], sv = 0, st = 0;
#pragma unroll
    for (int i = 0; i < MC_ITEMS; ++i) {
        const long long p = p0 + i;
        cv[i] = p < g.n ? vcnt[p] : 0;
        ct[i] = p < g.n ? (tcase[p] & 0xff) : 0;
        sv += cv[i]; st += ct[i];
    }
    int tv, tt;
    int bv = block_scan_excl(sv, lds, tv) + vblock[blockIdx.x];
    int bt = block_scan_excl(st, lds, tt) + tblock[blockIdx.x];
#pragma unroll
    for (int i = 0; i < MC_ITEMS; ++i) {
        const long long p = p0 + i;
        if (p < g.n) { vbase[p] = bv; tbase[p] = bt; }
        bv += cv[i]; bt += ct[i];
    }
}

__global__ __launch_bounds__(256) void k_mc_verts(const float* __restrict__ u, McGrid g, const uint8_t* __restrict__ vcnt,
                                                  const int* __restrict__ vbase, double* __restrict__ verts) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= g.n || vcnt[p] == 0) return;
    int cx, cy, cz;
    mc_decompose(g, p, cx, cy, cz);
    const int c[3] = {cx, cy, cz};
    const long long back[3] = {g.s0, g.s1, 1};
    const double f1 = (double)u[p], iso = g.iso_d;
    long long id = vbase[p];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (c[a] == 0) continue;
        const double f2 = (double)u[p - back[a]];
        if ((f1 <= iso) == (f2 <= iso)) continue;
        double v[3] = {(double)c[0], (double)c[1], (double)c[2]};
        v[a] = (double)c[a] + (iso - f1) * ((double)(c[a] - 1) - (double)c[a]) / (f2 - f1);
        verts[3 * id] = v[0]; verts[3 * id + 1] = v[1]; verts[3 * id + 2] = v[2];
        ++id;
    }
}

__device__ __forceinline__ int mc_vertex_id(const float* __restrict__ u, const McGrid& g, const int* __restrict__ vbase,
                                            long long q, int x, int y, int axis) {
    // rank of slot `axis` among the present slots of grid point q = (x, y, .)
    const bool in0 = u[q] <= g.iso;
    int id = vbase[q];
    if (axis > 0 && x > 0) id += (in0 != (u[q - g.s0] <= g.iso));
    if (axis > 1 && y > 0) id += (in0 != (u[q - g.s1] <= g.iso));
    return id;
}

template <typename IDX>
__global__ __launch_bounds__(256) void k_mc_tris(const float* __restrict__ u, McGrid g, const uint16_t* __restrict__ tcase,
                                                 const int* __restrict__ vbase, const int* __restrict__ tbase, IDX* __restrict__ tris) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= g.n) return;
    const int tc = tcase[p];
    const int nt = tc & 0xff, ci = tc >> 8;
    if (!nt) return;
    long long t0 = tbase[p];
    int x, y, z;
    mc_decompose(g, p, x, y, z);
    for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = MC_TRI[ci][3 * t + q];
            const long long gq = p + MC_FAR[e][0] * g.s0 + MC_FAR[e][1] * g.s1 + MC_FAR[e][2];
            tris[3 * (t0 + t) + q] = (IDX)mc_vertex_id(u, g, vbase, gq, x + MC_FAR[e][0], y + MC_FAR[e][1], MC_AXIS[e]);
        }
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

size_t o2345_mc_workspace_bytes(int n0, int n1, int n2) {
    const size_t n = (size_t)n0 * n1 * n2;
    const size_t nb = (n + MC_TILE - 1) / MC_TILE;
    return ((n + 15) / 16 * 16) + ((2 * n + 15) / 16 * 16) + 2 * n * sizeof(int) + (2 * nb + 8) * sizeof(int) + 64;
}

static void mc_carve(void* ws, size_t n, uint8_t*& vcnt, uint16_t*& tcase, int*& vbase, int*& tbase, int*& vblock, int*& tblock, long long*& totals) {
    char* p = (char*)ws;
    vcnt = (uint8_t*)p; p += (n + 15) / 16 * 16;
    tcase = (uint16_t*)p; p += (2 * n + 15) / 16 * 16;
    vbase = (int*)p; p += n * sizeof(int);
    tbase = (int*)p; p += n * sizeof(int);
    const size_t nb = (n + MC_TILE - 1) / MC_TILE;
    totals = (long long*)p; p += 32;
    vblock = (int*)p; p += (nb + 4) * sizeof(int);
    tblock = (int*)p;
}

// Pass 1 of the two-call protocol: classifies, scans, and returns the vertex / triangle counts on the HOST
// (synchronises the stream once -- the caller must allocate the outputs).
int o2345_marching_cubes_count(const float* u, int n0, int n1, int n2, double iso, void* workspace, size_t workspace_bytes,
                               long long* nv_host, long long* nt_host, void* stream) {
    O2345_REQUIRE(u && workspace && nv_host && nt_host, "marching_cubes_count: null pointer");
    O2345_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "marching_cubes_count: grid must be at least 2^3");
    O2345_REQUIRE(workspace_bytes >= o2345_mc_workspace_bytes(n0, n1, n2), "marching_cubes_count: workspace too small");
    const size_t n = (size_t)n0 * n1 * n2;
    O2345_REQUIRE(n < (1ull << 31), "marching_cubes_count: grid too large");
    uint8_t* vcnt; uint16_t* tcase; int *vbase, *tbase, *vblock, *tblock; long long* totals;
    mc_carve(workspace, n, vcnt, tcase, vbase, tbase, vblock, tblock, totals);
    const McGrid g = mc_grid(n0, n1, n2, iso);
    const unsigned nb = (unsigned)((n + MC_TILE - 1) / MC_TILE);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mc_count, dim3(nb), dim3(256), 0, s, u, g, vcnt, tcase, vblock, tblock);
    hipLaunchKernelGGL(k_scan_small3, dim3(1), dim3(1024), 0, s, vblock, (int)nb, totals);
    hipLaunchKernelGGL(k_scan_small3, dim3(1), dim3(1024), 0, s, tblock, (int)nb, totals + 1);
    hipLaunchKernelGGL(k_mc_offsets, dim3(nb), dim3(256), 0, s, g, vcnt, tcase, vblock, tblock, vbase, tbase);
    int rc = check_launch("marching_cubes_count");
    if (rc) return rc;
    long long h[2];
    hipError_t e = hipMemcpyAsync(h, totals, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    O2345_REQUIRE(e == hipSuccess, "marching_cubes_count: %s", hipGetErrorString(e));
    *nv_host = h[0]; *nt_host = h[1];
    return 0;
}

// Pass 2: emit.  verts float64 [nv,3]; tris int32 or int64 [nt,3] (index_bytes = 4 or 8).  Same workspace as pass 1.
int o2345_marching_cubes_emit(const float* u, int n0, int n1, int n2, double iso, void* workspace, double* verts, void* tris,
                              int index_bytes, void* stream) {
    O2345_REQUIRE(u && workspace, "marching_cubes_emit: null pointer");
    O2345_REQUIRE(index_bytes == 4 || index_bytes == 8, "marching_cubes_emit: index_bytes must be 4 or 8");
    const size_t n = (size_t)n0 * n1 * n2;
    uint8_t* vcnt; uint16_t* tcase; int *vbase, *tbase, *vblock, *tblock; long long* totals;
    mc_carve(workspace, n, vcnt, tcase, vbase, tbase, vblock, tblock, totals);
    const McGrid g = mc_grid(n0, n1, n2, iso);
    hipStream_t s = (hipStream_t)stream;
    if (verts) hipLaunchKernelGGL(k_mc_verts, dim3(cdiv(n, 256)), dim3(256), 0, s, u, g, vcnt, vbase, verts);
    if (tris) {
        if (index_bytes == 4) hipLaunchKernelGGL(k_mc_tris<int>, dim3(cdiv(n, 256)), dim3(256), 0, s, u, g, tcase, vbase, tbase, (int*)tris);
        else hipLaunchKernelGGL(k_mc_tris<long long>, dim3(cdiv(n, 256)), dim3(256), 0, s, u, g, tcase, vbase, tbase, (long long*)tris);
    }
    return check_launch("marching_cubes_emit");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_mcubes() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_scan_small3));
}
}  // namespace o2345
