// Occupied-point list grouped by view-visibility signature (round 3).
//
// The colour kernel (csrc/color_pts.hip) evaluates a (32-point tile, view) pair unless NO point of the tile projects into the view.  With the list in the
// order k_ray_finalize emits it (wave-major, sample-major) 80.7 % of the pairs of BASELINE config 2 are evaluated although only 72.2 % of the (point, view)
// pairs are visible: tiles straddle the borders of the source views' frusta.  The results of the network kernels do not depend on the order of the list
// (every entry is a slot, outputs are scattered by slot; MFMA columns are independent -- bit-identical, tests/test_gpu_parity.py), so the list is sorted,
// STABLY (neighbours in space stay neighbours: the gathers keep their cache lines), by the V-bit signature "view v sees the point": every tile then holds
// points with one signature and the evaluated pairs drop to the visible ones.  Measured: colour kernel 40.0 -> 36.1 ms at 8 views, 44.7 -> 37.1 ms at 32.
//
// LSD radix sort, one 8-bit digit per 8 views, each pass a stable counting sort:
//   k_sort_hist    : block = 4096 consecutive entries; (pass 0: signature from the point's projection -> keys[]) ; per-block digit histogram -> H[digit][block]
//   k_sort_scan_bin: one block per digit: exclusive scan of its row of H, row total -> T[digit]
//   k_sort_scan_tot: exclusive scan of T (256 values)
//   k_sort_scatter : wave w of a block owns entries [w*1024, (w+1)*1024) in order; per-wave digit counts give each wave its start per digit; inside a wave the
//                    64 entries of a step are ranked by ballot among equal digits -- stable by construction, no atomics on global memory.
// The number of entries is read from device memory (the render call never synchronises with the host); blocks past it do nothing.
#include "common.h"

namespace o2345 {

constexpr int SORT_BLOCK = 4096, SORT_THREADS = 256, SORT_PER_WAVE = 1024;

// 1 bit per view: the point projects strictly inside the image.  (X/Z in (0, W-1)  <=>  0 < X < (W-1) Z for Z > 0; the colour kernel's own test differs from
// this one only for points within rounding of a border, which costs nothing but a slightly less pure tile.)
__device__ __forceinline__ unsigned vis_signature(const float* __restrict__ proj, int V, int H, int W, float x, float y, float z) {
    unsigned key = 0;
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    for (int v = 0; v < V; ++v) {
        const float* P = proj + 12 * v;
        const float X = P[0] * x + P[1] * y + P[2] * z + P[3];
        const float Y = P[4] * x + P[5] * y + P[6] * z + P[7];
        const float Z = fmaxf(P[8] * x + P[9] * y + P[10] * z + P[11], 1e-3f);
        if (X > 0.f && X < wm * Z && Y > 0.f && Y < hm * Z) key |= 1u << v;
    }
    return key;
}

template <bool MAKE_KEYS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const int* __restrict__ list, const int* __restrict__ count, const float* __restrict__ pts,
                                                            const float* __restrict__ proj, int V, int H, int W, unsigned* __restrict__ keys, int shift,
                                                            int nblk, int* __restrict__ Hm) {
    __shared__ int hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int n = *count;
    const long long base = (long long)blockIdx.x * SORT_BLOCK;
    for (int j = 0; j < SORT_BLOCK / SORT_THREADS; ++j) {
        const long long i = base + j * SORT_THREADS + threadIdx.x;
        if (i < n) {
            unsigned key;
            if (MAKE_KEYS) {
                const size_t slot = (size_t)list[i];
                key = vis_signature(proj, V, H, W, pts[3 * slot], pts[3 * slot + 1], pts[3 * slot + 2]);
                keys[i] = key;
            } else {
                key = keys[i];
            }
            atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
    }
    __syncthreads();
    Hm[(size_t)threadIdx.x * nblk + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan of row `blockIdx.x` of H (nblk values) in place; the row total goes to T[blockIdx.x]
__global__ __launch_bounds__(256) void k_sort_scan_bin(int* __restrict__ Hm, int nblk, int* __restrict__ T) {
    __shared__ int part[256];
    int* row = Hm + (size_t)blockIdx.x * nblk;
    const int per = (nblk + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nblk);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += row[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
        T[blockIdx.x] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int b = b0; b < b1; ++b) { const int v = row[b]; row[b] = run; run += v; }
}

__global__ void k_sort_scan_tot(int* __restrict__ T) {       // one thread: 256 values (T[256..511] = exclusive scan of T[0..255])
    int run = 0;
    for (int d = 0; d < 256; ++d) { T[256 + d] = run; run += T[d]; }
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const int* __restrict__ list, const unsigned* __restrict__ keys, const int* __restrict__ count,
                                                               int shift, int nblk, const int* __restrict__ Hm, const int* __restrict__ T,
                                                               int* __restrict__ out_list, unsigned* __restrict__ out_keys) {
    __shared__ int cur[SORT_THREADS / 64][256];
    const int n = *count;
    const long long base = (long long)blockIdx.x * SORT_BLOCK;
    if (base >= n) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = lane; d < 256; d += 64) cur[wave][d] = 0;
    __syncthreads();
    const long long wbase = base + (long long)wave * SORT_PER_WAVE;
    // (a) this wave's digit counts
    for (int j = 0; j < SORT_PER_WAVE / 64; ++j) {
        const long long i = wbase + j * 64 + lane;
        if (i < n) atomicAdd(&cur[wave][(keys[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    // (b) counts -> start offsets: global start of the digit + this block's share + the waves in front of this one
    if (wave == 0) {
        for (int d = lane; d < 256; d += 64) {
            int run = T[256 + d] + Hm[(size_t)d * nblk + blockIdx.x];
            for (int w = 0; w < SORT_THREADS / 64; ++w) { const int c = cur[w][d]; cur[w][d] = run; run += c; }
        }
    }
    __syncthreads();
    // (c) stable placement, 64 entries at a time in list order
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int j = 0; j < SORT_PER_WAVE / 64; ++j) {
        const long long i = wbase + j * 64 + lane;
        const bool valid = i < n;
        const unsigned key = valid ? keys[i] : 0u;
        const int val = valid ? list[i] : 0;
        const int d = (int)((key >> shift) & 255u);
        unsigned long long remaining = __ballot(valid);
        int dst = 0;
        while (remaining) {                                   // wave-uniform loop over the distinct digits of this step (usually one or two)
            const int leader = __ffsll((long long)remaining) - 1;
            const int dl = __shfl(d, leader);
            const unsigned long long same = __ballot(valid && d == dl);
            int start = 0;
            if (lane == leader) start = atomicAdd(&cur[wave][dl], __popcll(same));
            start = __shfl(start, leader);
            if (valid && d == dl) dst = start + __popcll(same & lt);
            remaining &= ~same;
        }
        if (valid) {
            out_list[dst] = val;
            if (out_keys) out_keys[dst] = key;
        }
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

size_t o2345_list_sort_workspace_bytes(long long n_max, int V) {
    const size_t nblk = cdiv(n_max, SORT_BLOCK);
    const int passes = (V + 7) / 8;
    // keys A (+ keys B and a second list for more than one pass) + H[256][nblk] + T[512]
    return ((size_t)n_max * (passes > 1 ? 3 : 1) + 256 * nblk + 512 + 64) * 4;
}

int o2345_list_sort_by_visibility(const float* pts, const int32_t* list, const int32_t* count_dev, long long n_max, const float* proj, int V, int H, int W,
                                  int32_t* list_out, uint32_t* keys_out, void* workspace, size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(pts && list && count_dev && proj && list_out && workspace, "list_sort_by_visibility: null pointer");
    O2345_REQUIRE(V >= 1 && V <= 32 && n_max > 0 && n_max < 2147483647LL, "list_sort_by_visibility: 1..32 views, n_max < 2^31 (got V = %d)", V);
    O2345_REQUIRE(workspace_bytes >= o2345_list_sort_workspace_bytes(n_max, V), "list_sort_by_visibility: workspace too small");
    O2345_REQUIRE(list_out != list, "list_sort_by_visibility: in-place sorting is not supported");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)cdiv(n_max, SORT_BLOCK);
    const int passes = (V + 7) / 8;
    unsigned* keysA = (unsigned*)workspace;
    unsigned* keysB = passes > 1 ? keysA + n_max : nullptr;
    int* tmp = passes > 1 ? (int*)(keysB + n_max) : nullptr;
    int* Hm = (int*)(keysA + (size_t)n_max * (passes > 1 ? 3 : 1));
    int* T = Hm + (size_t)256 * nblk;
    const int* src_list = list;
    unsigned* src_keys = keysA;
    for (int p = 0; p < passes; ++p) {
        const bool last = p == passes - 1;
        int* dst_list = ((passes - 1 - p) % 2 == 0) ? list_out : tmp;      // the last pass writes list_out
        unsigned* dst_keys = last ? (unsigned*)keys_out : (src_keys == keysA ? keysB : keysA);
        if (p == 0)
            hipLaunchKernelGGL(k_sort_hist<true>, dim3(nblk), dim3(SORT_THREADS), 0, s, src_list, count_dev, pts, proj, V, H, W, src_keys, 0, nblk, Hm);
        else
            hipLaunchKernelGGL(k_sort_hist<false>, dim3(nblk), dim3(SORT_THREADS), 0, s, src_list, count_dev, pts, proj, V, H, W, src_keys, 8 * p, nblk, Hm);
        hipLaunchKernelGGL(k_sort_scan_bin, dim3(256), dim3(256), 0, s, Hm, nblk, T);
        hipLaunchKernelGGL(k_sort_scan_tot, dim3(1), dim3(1), 0, s, T);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(SORT_THREADS), 0, s, src_list, src_keys, count_dev, 8 * p, nblk, Hm, T, dst_list, dst_keys);
        src_list = dst_list;
        src_keys = dst_keys;
    }
    return check_launch("list_sort_by_visibility");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_list_sort() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_sort_scan_bin));
}
}  // namespace o2345
