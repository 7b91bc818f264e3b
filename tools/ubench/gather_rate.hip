// Gather throughput of the vector memory path on gfx950: every lane reads 128 contiguous bytes (8 x dwordx4, like one bilinear tap of a
// half pixel in the colour kernels / 2 x dwordx4 like a cost-volume tap) at a per-lane pseudo-random pixel of a working set of W bytes.
// Reports bytes / clk / CU (nominal 2.4 GHz) for W in the L1, L2, MALL and HBM regimes.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NLOAD>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, unsigned n_pix, int iters, float* out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        // neighbouring lanes read neighbouring pixels (like neighbouring rays): base from the wave, small per-lane offset
        const unsigned wave_base = (s >> 8) % n_pix;
        const unsigned pix = (__builtin_amdgcn_readfirstlane(wave_base) + (threadIdx.x & 63) * 3u) % n_pix;
        const float4* p = src + (size_t)pix * 16 + ((threadIdx.x >> 6) & 1) * 8;     // 256-byte pixels, this wave's half
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) { const float4 t = p[q]; acc += t.x + t.y + t.z + t.w; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t max_bytes = 1ull << 30;
    float4* src; hipMalloc(&src, max_bytes); hipMemset(src, 0, max_bytes);
    float* out; hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {16ull << 10, 1ull << 20, 16ull << 20, 128ull << 20, 1ull << 30};
    const char* names[] = {"16 KB (L1)", "1 MB (L2)", "16 MB (L2, all XCDs)", "128 MB (MALL)", "1 GB (HBM)"};
    for (int nl = 8; nl >= 2; nl -= 6)
        for (int i = 0; i < 5; ++i) {
            const unsigned n_pix = (unsigned)(sizes[i] / 256);
            const int iters = 2000, blocks = 2048;                                  // 8 waves per SIMD worth of blocks in flight
            auto launch = [&](int it) { if (nl == 8) k<8><<<blocks, 256>>>(src, n_pix, it, out); else k<2><<<blocks, 256>>>(src, n_pix, it, out); };
            launch(50); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)blocks * 256 * iters * nl * 16;
            printf("%d x dwordx4 per lane, working set %-22s: %7.1f GB/s = %5.1f B/clk/CU\n", nl, names[i], bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.4e9 / 256);
        }
    return 0;
}
