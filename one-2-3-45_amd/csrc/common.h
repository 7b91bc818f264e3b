// Shared helpers for the o2345 HIP library (gfx950 only).
#pragma once
#include "../../include/o2345.h"      // every translation unit sees the public prototypes: an entry point that drifts from the header does not compile
#include <atomic>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define O2345_HD __host__ __device__ __forceinline__

namespace o2345 {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

#define O2345_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            o2345::set_error(__VA_ARGS__);  \
            return -1;                      \
        }                                   \
    } while (0)

// runtime calls on the way to a launch (memset, attribute queries): failure -> error string + status, like O2345_REQUIRE
#define O2345_HIP(call)                                                               \
    do {                                                                              \
        const hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                       \
            o2345::set_error("%s: %s", #call, hipGetErrorString(e_));                 \
            return -2;                                                                \
        }                                                                             \
    } while (0)

inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// Compute units of the CURRENT device (the one the caller's stream belongs to), cached per device ordinal: persistent kernels
// size their grids from it, and one process may drive several GPUs.
inline int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// wave64 helpers --------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return __lane_id(); }

// exclusive prefix of a 1-bit predicate inside the wave + wave total (ballot + popcount; no LDS)
__device__ __forceinline__ int wave_prefix(bool pred, int& total) {
    unsigned long long m = __ballot(pred);
    total = __popcll(m);
    return __popcll(m & ((1ull << lane_id()) - 1ull));
}

// exclusive prefix of a 1-bit predicate over a 256/512/1024-thread block; returns block total via ref
template <int NWAVES>
__device__ __forceinline__ int block_prefix(bool pred, int* lds_wave_tot /*[NWAVES+1]*/, int& block_total) {
    int wtot;
    int p = wave_prefix(pred, wtot);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) lds_wave_tot[w] = wtot;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) {
        int t = lds_wave_tot[i];
        if (i < w) base += t;
        tot += t;
    }
    block_total = tot;
    __syncthreads();
    return base + p;
}


// Debug / A-B knobs of the library, read from the environment ONCE (first use; function-local static: thread-safe, and no getenv on a launch path --
// getenv is not safe against a concurrent setenv, and the library is used from one host thread per stream / device):
//   O2345_LIST_SORT=0     render call: keep the occupied-point list in emission order (no grouping by view-visibility signature)
//   O2345_COLOR_SCHED=n   scheduling bits of the colour kernel (color_net.h), default 10
//   O2345_SPARSE_BRICK=0  finest sparse convolution in the gather form instead of the LDS-tiled brick form
//   O2345_FLAT_SCHED=1    persistent network kernels: flat block-interleaved tile schedule (odd grid) instead of one eighth of the list per XCD
//   O2345_COLOR_KERNEL=tiles  (only in a -DO2345_TILES_KERNEL test build) the (point, view)-column colour kernel instead of k_color_pts
//   O2345_RAY_STREAM_MIN=n  render call: ray batches of at least n rays run the streaming sampler kernels (one lane per ray, lists read from global
//                         memory at full occupancy), smaller ones the sixteen-lanes-per-ray kernels (default 4096; 0 = always streaming, a huge value = never)
struct Knobs {
    bool list_sort, sparse_brick, flat_sched, color_tiles;
    int color_sched;
    long long ray_stream_min;
};
inline const Knobs& knobs() {
    static const Knobs k = [] {
        auto is = [](const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; };
        const char* cs = getenv("O2345_COLOR_SCHED");
        const char* rs = getenv("O2345_RAY_STREAM_MIN");
        return Knobs{!is("O2345_LIST_SORT", '0'), !is("O2345_SPARSE_BRICK", '0'), is("O2345_FLAT_SCHED", '1'), is("O2345_COLOR_KERNEL", 't'), cs ? atoi(cs) : 10,
                     rs ? atoll(rs) : 4096};
    }();
    return k;
}

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, raised at most once per (call site, device, size): call sites keep one `MaxLds` object
// (function-local static).  Replaces a hipFuncSetAttribute on every launch.
struct MaxLds {
    std::atomic<int> cur[64];
    MaxLds() { for (auto& c : cur) c.store(0, std::memory_order_relaxed); }
    hipError_t ensure(const void* func, size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if ((int)bytes <= cur[dev].load(std::memory_order_acquire)) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) cur[dev].store((int)bytes, std::memory_order_release);
        return e;
    }
};
#define O2345_ENSURE_LDS(kernel, bytes)                                                   \
    do {                                                                                  \
        static o2345::MaxLds max_lds_;                                                    \
        O2345_HIP(max_lds_.ensure((const void*)(kernel), (size_t)(bytes)));               \
    } while (0)

// host side: grid of a persistent network kernel.  O2345_FLAT_SCHED=1 (debug / A-B knob) makes the grid odd, which selects the
// flat block-interleaved schedule in tile_schedule() below.
inline unsigned persistent_grid(long long want, int n_cu) {
    unsigned g = (unsigned)(want < n_cu ? want : n_cu);
    if (knobs().flat_sched && g > 8 && (g & 7) == 0) --g;
    return g;
}

#if defined(__HIPCC__)
// Residual halves of the split-f16 operand form: lo = f16(x - hi) for a PAIR of values whose hi halves are packed in `hi` -- v_fma_mixlo_f16 /
// v_fma_mixhi_f16 take the f16 half as an operand, subtract exactly in fp32 and write the rounded f16 straight into the low / high half of the
// result: 2 instructions per pair instead of 2 x v_fma_mix_f32 + v_cvt_pkrtz (hipcc does not select them from C code; checked value by value against
// the C form on hardware, tools/ubench/mixlo_check.hip: identical up to the rounding of the residual, nearest instead of toward zero).  Used by the SDF
// kernels (csrc/sdf_mlp_x3.hip): k_sdf_grad_x3 drops from 2766 to 2650 vector instructions and, more importantly, below the 256-register line -- its
// 224 bytes per lane of scratch disappear: 11.9 -> 9.9 ms on 29.5 M points.  O2345_SPLIT_MIXLO=0 builds the three-instruction form (A/B).
#ifndef O2345_SPLIT_MIXLO
#define O2345_SPLIT_MIXLO 1
#endif
__device__ __forceinline__ unsigned split_lo_pair_bits(unsigned hi_bits, float a, float b) {
    unsigned lo;
    // `volatile` matters: as a "pure" asm the pair gave wrong colours in k_color_mfma (G < 32) on hardware while s_nop-padded and volatile builds of the same
    // source were correct -- LLVM moves / merges side-effect-free asm in ways the EXEC-ignoring matrix instructions that consume the result do not survive
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(lo) : "v"(hi_bits), "v"(a), "v"(b));
    return lo;
}

// Tile schedule of the persistent network kernels.  Workgroups are dispatched round-robin over the 8 XCDs (block b runs on XCD
// b % 8) and every XCD has its own 4 MB L2: handing consecutive tiles to consecutive blocks makes each XCD stream the whole
// working set (source-view maps, latent volume) through its L2.  Instead every XCD gets one contiguous eighth of the tile list
// (neighbouring rays / samples share map pixels and voxels), interleaved over its own blocks and waves.
struct TileSched { long long first, end, stride; };
__device__ __forceinline__ TileSched tile_schedule(long long n_units, int units_per_tile, int wave, int nwave) {
    const long long ntiles = (n_units + units_per_tile - 1) / units_per_tile;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
        const long long chunk = (ntiles + 7) / 8, lo = xcd * chunk, hi = lo + chunk;
        return {lo + (long long)slot * nwave + wave, hi < ntiles ? hi : ntiles, (long long)per_xcd * nwave};
    }
    return {(long long)blockIdx.x * nwave + wave, ntiles, (long long)gridDim.x * nwave};
}
#endif

}  // namespace o2345
