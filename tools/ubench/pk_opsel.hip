// Standalone reproducer of the packed-FP32 operand-swizzle fault found on MI355X in round 3 (profiles/NOTES.md, "co-resident MFMA").
// ISA-level bisect of the cost-volume gather (tools/hazard/) left exactly two instructions that make it fail, both of the form
//     v_pk_add_f32 vD[0:1], vA[0:1], vB[0:1] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]        (D.lo = A.lo - B.hi ; D.hi = A.hi - B.lo)
// i.e. packed FP32 with CROSS-HALF source selection.  Each thread below executes such instructions in a loop on varying operands and compares every result with the
// same arithmetic done by two plain v_sub_f32; mismatches are counted per lane.  Run alone: 0.  Run next to k_aggr_mfma (poison.hip) of another stream: ?
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int FORM>
__device__ __forceinline__ f2 pk_op(f2 a, f2 b) {
    f2 d;
    if (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));   // the gather's form
    if (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));                               // swizzle, no negation
    if (FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));                                  // negation, no swizzle
    if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));                               // multiply with swizzle
    if (FORM == 4) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));                                            // broadcast of B.lo (the common compiler form)
    if (FORM == 5) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));                                                            // plain
    if (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b));                        // fma, B swizzled
    if (FORM == 7) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));                                               // D.lo = A.hi, D.hi = B.lo
    if (FORM == 8) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));                               // swizzle on src0 instead of src1
    return d;
}
// the reference halves as single plain instructions (inline asm, so that the compiler cannot re-pack them)
__device__ __forceinline__ float s_sub(float x, float y) { float r; asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float s_add(float x, float y) { float r; asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float s_mul(float x, float y) { float r; asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
template <int FORM>
__device__ __forceinline__ f2 ref_op(f2 a, f2 b) {
    if (FORM == 0) return f2{s_sub(a.x, b.y), s_sub(a.y, b.x)};
    if (FORM == 1) return f2{s_add(a.x, b.y), s_add(a.y, b.x)};
    if (FORM == 2) return f2{s_sub(a.x, b.x), s_sub(a.y, b.y)};
    if (FORM == 3) return f2{s_mul(a.x, b.y), s_mul(a.y, b.x)};
    if (FORM == 4) return f2{s_add(a.x, b.x), s_add(a.y, b.x)};
    if (FORM == 6) { float lo, hi; asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(lo) : "v"(a.x), "v"(b.y)); asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(hi) : "v"(a.y), "v"(b.x)); return f2{lo, hi}; }
    if (FORM == 7) return f2{a.y, b.x};
    if (FORM == 8) return f2{s_add(a.y, b.x), s_add(a.x, b.y)};
    return f2{s_add(a.x, b.x), s_add(a.y, b.y)};
}
template <int FORM>
__global__ __launch_bounds__(256) void k_pk_opsel(int iters, unsigned* __restrict__ lane_errs /*[64]*/, float* __restrict__ sample /*[8]*/) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f2 a = {(float)(t & 255) + 0.37f, (float)((t >> 8) & 255) * 1.5f + 0.11f};
    f2 b = {(float)((t * 7) & 127), (float)((t * 13) & 63) + 100.f};
    unsigned errs = 0;
    for (int i = 0; i < iters; ++i) {
        const f2 d = pk_op<FORM>(a, b);
        const f2 e = ref_op<FORM>(a, b);
        if (__builtin_bit_cast(unsigned, d.x) != __builtin_bit_cast(unsigned, e.x) || __builtin_bit_cast(unsigned, d.y) != __builtin_bit_cast(unsigned, e.y)) {
            if (errs == 0 && atomicAdd(&lane_errs[64], 1u) == 0) {              // first failure of the launch: keep the operands and both results
                sample[0] = a.x; sample[1] = a.y; sample[2] = b.x; sample[3] = b.y; sample[4] = d.x; sample[5] = d.y; sample[6] = e.x; sample[7] = e.y;
            }
            ++errs;
        }
        a = f2{s_add(s_mul(a.y, 0.5f), 1.25f), s_add(s_mul(a.x, 0.5f), 0.75f)};   // plain VALU between the packed instructions, operands keep changing
        b = f2{s_add(b.y, 1.f), s_add(b.x, 3.f)};
    }
    if (errs) atomicAdd(&lane_errs[threadIdx.x & 63], errs);
}
extern "C" int pk_opsel_launch(int form, int iters, int blocks, unsigned* lane_errs, float* sample, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (form) {
        case 0: hipLaunchKernelGGL(k_pk_opsel<0>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 1: hipLaunchKernelGGL(k_pk_opsel<1>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 2: hipLaunchKernelGGL(k_pk_opsel<2>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 3: hipLaunchKernelGGL(k_pk_opsel<3>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 4: hipLaunchKernelGGL(k_pk_opsel<4>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 6: hipLaunchKernelGGL(k_pk_opsel<6>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 7: hipLaunchKernelGGL(k_pk_opsel<7>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        case 8: hipLaunchKernelGGL(k_pk_opsel<8>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
        default: hipLaunchKernelGGL(k_pk_opsel<5>, dim3(blocks), dim3(256), 0, s, iters, lane_errs, sample); break;
    }
    return (int)hipGetLastError();
}
