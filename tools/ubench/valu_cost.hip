// Issue cost (cycles per wave-instruction, 3 waves/SIMD, 8 independent chains per wave) of the VALU instructions the network kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hh16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, int iters) {
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i * 0.1f; p[i] = f32x2{v[i], v[i] + 1.f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int c = j & 7;
            if (OP == 0) v[c] = __builtin_fmaf(v[c], 1.0001f, 0.5f);
            if (OP == 1) p[c] = __builtin_elementwise_fma(p[c], f32x2{1.0001f, 1.0002f}, f32x2{0.5f, 0.25f});
            if (OP == 2) v[c] = __builtin_amdgcn_exp2f(v[c]) * 0.0f + v[c];                      // exp + fma
            if (OP == 3) v[c] = __builtin_fmaxf(v[c], 0.25f * (float)j);
            if (OP == 4) v[c] = v[c] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[c]), 0xB1, 0xf, 0xf, true));
            if (OP == 5) { h16x2 h = __builtin_amdgcn_cvt_pkrtz(v[c], v[(c + 1) & 7]); v[c] += (float)__builtin_bit_cast(hh16x2, h)[0]; }   // cvt_pkrtz + fma_mix
            if (OP == 6) v[c] = __builtin_amdgcn_rcpf(v[c]) * 0.0f + v[c];
            if (OP == 7) p[c] = p[c] * f32x2{1.0001f, 0.9999f};
            if (OP == 8) v[c] = __builtin_amdgcn_fmed3f(v[c], 0.f, 1.f) + 0.5f;
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
float run() {
    float* d; hipMalloc(&d, 256 * 768 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 768>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<256, 768>>>(d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d);
    return ms * 1e-3 * 2.4e9 / 20000 / (3 * 64);      // nominal cycles per wave-instruction-slot (64 op groups per iteration, 3 waves/SIMD)
}

int main() {
    printf("per 'op group' (nominal cycles at 2.4 GHz, 3 waves/SIMD):\n");
    printf(" v_fma_f32            %.2f\n v_pk_fma_f32         %.2f\n v_exp_f32 + v_fma    %.2f\n v_max_f32            %.2f\n v_add_f32_dpp        %.2f\n"
           " cvt_pkrtz + fma_mix  %.2f\n v_rcp_f32 + v_fma    %.2f\n v_pk_mul_f32         %.2f\n v_med3 + v_add       %.2f\n",
           run<0>(), run<1>(), run<2>(), run<3>(), run<4>(), run<5>(), run<6>(), run<7>(), run<8>());
    return 0;
}
