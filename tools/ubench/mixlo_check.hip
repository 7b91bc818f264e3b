// checks split_lo_pair_bits (v_fma_mixlo_f16 / v_fma_mixhi_f16) against the C form for random and edge inputs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include <string.h>
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hh16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned split_lo_pair_bits(unsigned hi_bits, float a, float b) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi_bits), "v"(a), "v"(b));
    return lo;
}
__global__ void k(const float* x, int n, unsigned* hi_o, unsigned* lo_asm, unsigned* lo_c, float m1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    union { h16x2 v; hh16x2 w; unsigned u; } hi, lo;
    hi.v = __builtin_amdgcn_cvt_pkrtz(a, b);
    lo.v = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hi.w[0], m1, a), __builtin_fmaf((float)hi.w[1], m1, b));
    hi_o[i] = hi.u; lo_c[i] = lo.u; lo_asm[i] = split_lo_pair_bits(hi.u, a, b);
}
static float h2f(unsigned short h) { _Float16 f; memcpy(&f, &h, 2); return (float)f; }
int main() {
    const int n = 1 << 20;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float u = (float)rand() / RAND_MAX * 2 - 1;
        int e = rand() % 40 - 30;
        x[i] = ldexpf(u, e);
    }
    x[0] = 0.f; x[1] = -0.f; x[2] = 1.f; x[3] = -1.f; x[4] = 65504.f; x[5] = 1e-7f; x[6] = 3.0001f; x[7] = -2.9999f;
    float* dx; unsigned *dh, *da, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&da, n * 2); hipMalloc(&dc, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, n, dh, da, dc, -1.f);
    std::vector<unsigned> h(n / 2), a(n / 2), c(n / 2);
    hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 2, hipMemcpyDeviceToHost);
    double worst_asm = 0, worst_c = 0; int diff = 0, shown = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int s = 0; s < 2; ++s) {
            float v = x[2 * i + s];
            float hi = h2f((h[i] >> (16 * s)) & 0xffff), la = h2f((a[i] >> (16 * s)) & 0xffff), lc = h2f((c[i] >> (16 * s)) & 0xffff);
            double ea = fabs((double)v - hi - la), ec = fabs((double)v - hi - lc);
            double scale = fabs(v) > 1e-30 ? fabs(v) : 1;
            if (ea / scale > worst_asm) worst_asm = ea / scale;
            if (ec / scale > worst_c) worst_c = ec / scale;
            if (la != lc) { ++diff; if (shown < 6) { printf("x=%.9g hi=%.9g lo_asm=%.9g lo_c=%.9g\n", v, hi, la, lc); ++shown; } }
        }
    printf("values %d, lo differs in %d, worst relative |x - hi - lo|: asm %.3e, C %.3e\n", n, diff, worst_asm, worst_c);
    return 0;
}
