// Do per-lane gathers (texture-addresser-bound) of one wave overlap with the VALU work of the other waves on its SIMD / CU?
// Each wave alternates: NG x (8 scattered dwordx4 loads of 128 B per lane, consumed) and NV independent v_fma.  3 waves / SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NG, int NV>
__global__ __launch_bounds__(768) void k(const float4* __restrict__ src, unsigned n_pix, int iters, float* out) {
    unsigned s = (blockIdx.x * 768 + threadIdx.x) * 2654435761u + 12345u;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    float acc = 0.f;
    // de-synchronise the waves: every wave starts with a different amount of VALU work
    for (int j = 0; j < (int)(threadIdx.x >> 6) * 37; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            s = s * 1664525u + 1013904223u;
            const unsigned wave_base = (s >> 8) % n_pix;
            const unsigned pix = (__builtin_amdgcn_readfirstlane(wave_base) + (threadIdx.x & 63) * 3u) % n_pix;
            const float4* p = src + (size_t)pix * 16;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float4 t = p[q]; acc += t.x + t.y + t.z + t.w; }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
    }
    for (int i = 0; i < 8; ++i) acc += v[i];
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}

template <int NG, int NV>
float run(const float4* src, unsigned n_pix, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NG, NV><<<256, 768>>>(src, n_pix, 10, out); hipDeviceSynchronize();
    const int iters = 3000;
    hipEventRecord(e0); k<NG, NV><<<256, 768>>>(src, n_pix, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / iters;      // nominal cycles per iteration (all 3 waves of a SIMD)
}

int main() {
    const size_t bytes = 16ull << 20;
    float4* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
    float* out; hipMalloc(&out, 256 * 768 * sizeof(float));
    const unsigned n_pix = bytes / 256;
    printf("3 waves/SIMD, cycles per iteration:  4 gathers only %.0f | 1024 VALU only %.0f | both %.0f   ||  4 gathers %.0f | 2048 VALU %.0f | both %.0f\n",
           run<4, 0>(src, n_pix, out), run<0, 1024>(src, n_pix, out), run<4, 1024>(src, n_pix, out),
           run<4, 0>(src, n_pix, out), run<0, 2048>(src, n_pix, out), run<4, 2048>(src, n_pix, out));
    return 0;
}
