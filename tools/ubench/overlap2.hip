// (1) in-wave interleave: per MFMA (independent accumulators, NACC of them) K independent VALU fmas placed right after it.
// (2) cross-wave: even wave-quads run MFMA only, odd wave-quads VALU only, on the same SIMDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int NACC>
__global__ void k_inwave(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {};
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(i * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>   // 0: all waves MFMA; 1: all waves VALU; 2: wave-quads alternate (even MFMA-only, odd VALU-only)
__global__ void k_cross(float* out, int iters) {
    f32x16 acc = {};
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(i * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const bool mf = MODE == 0 || (MODE == 2 && (((threadIdx.x >> 8) & 1) == 0));
    if (mf) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 128; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10); hipDeviceSynchronize();
    hipEventRecord(e0); launch(20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / 20000;     // cycles per iteration at nominal clock
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
#define IW(K, NACC, W) timeit([&](int it) { k_inwave<K, NACC><<<256, 256 * W>>>(d, it); })
    printf("in-wave, 1 wave/SIMD, 8 MFMA/iter, NACC=2: K=0 %.0f  K=2 %.0f  K=4 %.0f  K=6 %.0f  K=8 %.0f  K=12 %.0f  K=16 %.0f\n",
           IW(0, 2, 1), IW(2, 2, 1), IW(4, 2, 1), IW(6, 2, 1), IW(8, 2, 1), IW(12, 2, 1), IW(16, 2, 1));
    printf("in-wave, 1 wave/SIMD, NACC=1 (dependent): K=0 %.0f  K=4 %.0f  K=8 %.0f  K=12 %.0f  K=16 %.0f\n",
           IW(0, 1, 1), IW(4, 1, 1), IW(8, 1, 1), IW(12, 1, 1), IW(16, 1, 1));
    printf("in-wave, 2 waves/SIMD, NACC=1: K=0 %.0f  K=4 %.0f  K=8 %.0f  K=12 %.0f  K=16 %.0f\n",
           IW(0, 1, 2), IW(4, 1, 2), IW(8, 1, 2), IW(12, 1, 2), IW(16, 1, 2));
    printf("in-wave, 3 waves/SIMD, NACC=1: K=0 %.0f  K=4 %.0f  K=8 %.0f  K=12 %.0f  K=16 %.0f\n",
           IW(0, 1, 3), IW(4, 1, 3), IW(8, 1, 3), IW(12, 1, 3), IW(16, 1, 3));
#define CR(M, W) timeit([&](int it) { k_cross<M><<<256, 256 * W>>>(d, it); })
    printf("cross-wave, 2 waves/SIMD: all-MFMA(8/iter) %.0f  all-VALU(128/iter) %.0f  one-of-each %.0f\n", CR(0, 2), CR(1, 2), CR(2, 2));
    printf("cross-wave, 4 waves/SIMD: all-MFMA %.0f  all-VALU %.0f  two-of-each %.0f\n", CR(0, 4), CR(1, 4), CR(2, 4));
    return 0;
}
