// Micro-benchmark: do MFMA and VALU work of DIFFERENT waves on one SIMD overlap on gfx950?
// Each wave alternates a block of NV dependent-free VALU fmas with a block of NM v_mfma_f32_32x32x16_f16 on ONE accumulator
// (dependent chain, like one layer of the network kernels).  Launch: 256 blocks x (64*W*4) threads -> W waves per SIMD.
// Prints cycles per iteration for (NV,0), (0,NM), (NV,NM): sum model vs overlap model.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NM>
__global__ void k(float* out, int iters) {
    f32x16 acc = {};
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(i * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    // de-synchronise the waves of a SIMD: waves are placed on SIMDs round-robin, so wave ids w, w+4, w+8 share a SIMD; every other
    // one of them starts with an extra half iteration of VALU work
    if (NM > 0 && NV > 0 && ((threadIdx.x >> 8) & 1)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
        // make the MFMA block depend on the VALU block (like split -> MFMA) and vice versa (MFMA -> activation)
        if (NM > 0) a[0] = (_Float16)v[0];
#pragma unroll
        for (int j = 0; j < NM; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (NM > 0) v[1] += acc[0];
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NM>
float run(int waves_per_simd, int iters) {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 64 * 4 * waves_per_simd;
    k<NV, NM><<<256, threads>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, NM><<<256, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}

int main() {
    const int iters = 20000;
    for (int w = 1; w <= 4; ++w) {
        const float tv = run<64, 0>(w, iters), tm = run<0, 8>(w, iters), tb = run<64, 8>(w, iters);
        const float tv2 = run<128, 0>(w, iters), tb2 = run<128, 8>(w, iters), tm2 = run<0, 16>(w, iters), tb3 = run<64, 16>(w, iters);
        // cycles per iteration per SIMD at 2.4 GHz (nominal)
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / iters; };
        printf("waves/SIMD %d: VALU64 %.0f  MFMA8 %.0f  both %.0f | VALU128 %.0f both(128,8) %.0f | MFMA16 %.0f both(64,16) %.0f   [cycles/iter/SIMD]\n",
               w, cyc(tv), cyc(tm), cyc(tb), cyc(tv2), cyc(tb2), cyc(tm2), cyc(tb3));
    }
    return 0;
}
