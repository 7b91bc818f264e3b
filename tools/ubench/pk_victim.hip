// Minimal victims for the packed-FP32 / co-resident-MFMA corruption (profiles/NOTES.md).  Run idle -> reference; run next to k_aggr_mfma (poison.hip) of
// another stream -> compare bitwise.  Built twice: as is, and with -Xclang -target-feature -Xclang -packed-fp32-ops (no v_pk_* instructions).
//   k_pk_victim      : pure ALU -- a dependent chain of v_pk_mul/add/fma_f32 and v_floor_f32 reads of the halves (pixel arithmetic of the gather)
//   k_pk_load_victim : global_load_dwordx4 -> s_waitcnt vmcnt(0) -> v_pk_mul_f32 / v_pk_add_f32 on the loaded registers (the tap accumulation of the gather)
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk_victim(int iters, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f2 x = {(float)(t & 1023) * 0.37f + 0.11f, (float)(t >> 3 & 1023) * 0.53f + 0.29f};
    const f2 half = {0.5f, 0.5f}, size = {255.f, 255.f}, one = {1.f, 1.f};
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        f2 p = (x + one) * half;
        p = p * size;
        const float fx = floorf(p.x), fy = floorf(p.y);
        const f2 fl = {fx, fy};
        const f2 fr = p - fl;
        acc += fr.x * fr.y + fx * 1e-3f;
        x = fr * (f2){3.7f, 2.9f} + (f2){0.13f + (float)i * 1e-4f, 0.07f};
    }
    out[t] = acc;
}
__global__ __launch_bounds__(256) void k_pk_load_victim(const float4* __restrict__ table, int mask, int iters, float4* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    unsigned h = (unsigned)t * 2654435761u;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (int i = 0; i < iters; ++i) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            const float w = (float)(h >> 24) * (1.f / 256.f);
            if (w != 0.f) {                                     // the same divergent "tap inside the image" shape as the gather
                const float4 a = table[(h >> 8) & mask];
                f.x += a.x * w; f.y += a.y * w; f.z += a.z * w; f.w += a.w * w;
            }
        }
        s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
        s2.x += f.x * f.x; s2.y += f.y * f.y; s2.z += f.z * f.z; s2.w += f.w * f.w;
    }
    out[2 * t] = s1;
    out[2 * t + 1] = s2;
}
extern "C" int pk_victim_launch(int iters, int blocks, float* out, void* stream) {
    hipLaunchKernelGGL(k_pk_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
    return (int)hipGetLastError();
}
extern "C" int pk_load_victim_launch(const float* table, int mask, int iters, int blocks, float* out, void* stream) {
    hipLaunchKernelGGL(k_pk_load_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)table, mask, iters, (float4*)out);
    return (int)hipGetLastError();
}
