// Does the vector L1 charge a wave load by the number of distinct cache lines it touches?  64 lanes x dwordx4, L2-resident working set:
//   mode 0: every lane its own 256-B pixel (64 lines)   mode 1: lane octets share a 128-B line (8 lines, 1 KB... 8 x 128 B)
//   mode 2: fully contiguous 1 KB (8 lines)              mode 3: mode 1 through global_load_lds (LDS-DMA) + one ds_read_b128 per lane
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, unsigned n_pix, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) float4 stage[4][64];
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned wb = __builtin_amdgcn_readfirstlane((s >> 8) % n_pix);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4* p;
            if (MODE == 0) p = src + (size_t)((wb + lane * 3u + q * 211u) % n_pix) * 16 + (q & 7);
            else if (MODE == 1 || MODE == 3) p = src + (size_t)((wb + (lane >> 3) * 3u + q * 211u) % n_pix) * 16 + (lane & 7);
            else p = src + (size_t)((wb + q * 211u) % (n_pix - 4)) * 16 + lane;
            if (MODE == 3) {
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p, (void __attribute__((address_space(3)))*)&stage[wave][0], 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const float4 t = stage[wave][lane ^ 5];
                acc += t.x + t.y + t.z + t.w;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                const float4 t = *p;
                acc += t.x + t.y + t.z + t.w;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const float4* src, unsigned n_pix, float* out, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 2048;
    k<MODE><<<blocks, 256>>>(src, n_pix, 20, out); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(src, n_pix, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * 8 * 16;
    printf("%-58s %7.1f GB/s = %5.1f B/clk/CU\n", name, bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
    const size_t bytes = 16ull << 20;
    float4* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
    float* out; hipMalloc(&out, 2048 * 256 * sizeof(float));
    const unsigned n_pix = bytes / 256;
    run<0>(src, n_pix, out, "64 lanes -> 64 different lines (16 B each)");
    run<1>(src, n_pix, out, "lane octets share a 128-B line (8 lines per instruction)");
    run<2>(src, n_pix, out, "1 KB contiguous (8 lines per instruction)");
    run<3>(src, n_pix, out, "octets, through global_load_lds + ds_read (serialised)");
    return 0;
}
