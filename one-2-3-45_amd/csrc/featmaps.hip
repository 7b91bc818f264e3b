// Feature-pyramid glue of FeatureNet as two fused HIP kernels (models/featurenet.py:68-91,
// models/trainer_generic.py:1104-1125).  The convolutions are csrc/convnet.hip; what is fused here is everything between them:
//
//   k_fpn_level     f_out = lateral 1x1 conv (C_in -> 32, + bias) of the finer map + bilinear x2 up-sampling (align_corners = True) of the
//                   coarser map                                   (featurenet.py:73-76, 85-86: `_upsample_add(feat, lat(conv))`)
//   k_pyramid_pack  fused pyramid [x4 up-sampled f2 (32) | x2 up-sampled smooth1 (16) | smooth0 (8)] written ONCE, both as the channel-first
//                   tensor the reference API / the compress convolution expect and as the channel-last colour map [V,H,W,64] = rgb | 56
//                   features | pad that the colour kernels gather from             (trainer_generic.py:1117-1123 + the former k_pack_cmaps)
//
// Bilinear weights follow ATen's upsample_bilinear2d (align_corners: src = dst * (in - 1) / (out - 1) in fp32, i1 = i0 + (i0 < in - 1),
// value = l0y (l0x a + l1x b) + l1y (l0x c + l1x d)).
#include "common.h"

namespace o2345 {

struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp up_coord(int dst, int n_in, int n_out) {
    const float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    const float src = scale * (float)dst;
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ plane, int w_in, const Lerp& y, const Lerp& x) {
    const float a = plane[y.i0 * w_in + x.i0], b = plane[y.i0 * w_in + x.i1], c = plane[y.i1 * w_in + x.i0], d = plane[y.i1 * w_in + x.i1];
    return y.l0 * (x.l0 * a + x.l1 * b) + y.l1 * (x.l0 * c + x.l1 * d);
}

// grid (ceil(H*W / 256), V); one thread per output pixel, all 32 channels; weights [32][CIN] + bias [32] in LDS
template <int CIN>
__global__ __launch_bounds__(256) void k_fpn_level(const float* __restrict__ fine /*[V,CIN,H,W]*/, const float* __restrict__ coarse /*[V,32,H/2,W/2]*/,
                                                    const float* __restrict__ w /*[32,CIN]*/, const float* __restrict__ bias /*[32]*/, int H, int W,
                                                    float* __restrict__ out /*[V,32,H,W]*/, const float* __restrict__ fine_ss /*[2*CIN] or null*/, float slope) {
    __shared__ float sw[32 * CIN + 32];
    for (int i = threadIdx.x; i < 32 * CIN + 32; i += 256) sw[i] = i < 32 * CIN ? w[i] : bias[i - 32 * CIN];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (p >= H * W) return;
    const int y = p / W, x = p % W, hc = H / 2, wc = W / 2;
    float in[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        float t = fine[((size_t)v * CIN + c) * H * W + p];
        if (fine_ss) {                                      // `fine` is a raw convolution output: its InPlaceABN is applied here (csrc/convnet.hip)
            t = t * fine_ss[c] + fine_ss[CIN + c];
            t = t >= 0.f ? t : t * slope;
        }
        in[c] = t;
    }
    const Lerp ly = up_coord(y, hc, H), lx = up_coord(x, wc, W);
    const float* cbase = coarse + (size_t)v * 32 * hc * wc;
    for (int o = 0; o < 32; ++o) {
        float acc = sw[32 * CIN + o];
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc = fmaf(sw[o * CIN + c], in[c], acc);
        out[((size_t)v * 32 + o) * H * W + p] = bilerp(cbase + (size_t)o * hc * wc, wc, ly, lx) + acc;       // F.interpolate(x) + y
    }
}

// grid (ceil(H*W / 256), V); one thread per pixel.  fmaps_nchw may be null.  The channel-last pixel (256 bytes) is not written by its own thread
// (64 lanes x 16 bytes at a 256-byte stride = 64 cache lines per store instruction) but through a per-wave LDS transpose in two halves of 32
// channels: a store instruction then covers 8 pixels x 128 contiguous bytes (8 full lines).
__global__ __launch_bounds__(256) void k_pyramid_pack(const float* __restrict__ f2 /*[V,32,H/4,W/4]*/, const float* __restrict__ s1 /*[V,16,H/2,W/2]*/,
                                                       const float* __restrict__ s0 /*[V,8,H,W]*/, const float* __restrict__ rgb /*[V,3,H,W]*/, int H, int W,
                                                       float* __restrict__ fmaps_nchw /*[V,56,H,W]*/, float* __restrict__ cmaps /*[V,H,W,64]*/) {
    __shared__ float tr[4][64][33];
    const int p = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y, HW = H * W;
    const bool live = p < HW;
    const int pc = live ? p : HW - 1;
    const int y = pc / W, x = pc % W;
    float px[64];
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = rgb[((size_t)v * 3 + c) * HW + pc];
    {
        const int h4 = H / 4, w4 = W / 4;
        const Lerp ly = up_coord(y, h4, H), lx = up_coord(x, w4, W);
        const float* b = f2 + (size_t)v * 32 * h4 * w4;
#pragma unroll
        for (int c = 0; c < 32; ++c) px[3 + c] = bilerp(b + (size_t)c * h4 * w4, w4, ly, lx);
    }
    {
        const int h2 = H / 2, w2 = W / 2;
        const Lerp ly = up_coord(y, h2, H), lx = up_coord(x, w2, W);
        const float* b = s1 + (size_t)v * 16 * h2 * w2;
#pragma unroll
        for (int c = 0; c < 16; ++c) px[35 + c] = bilerp(b + (size_t)c * h2 * w2, w2, ly, lx);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) px[51 + c] = s0[((size_t)v * 8 + c) * HW + pc];
#pragma unroll
    for (int c = 59; c < 64; ++c) px[c] = 0.f;
    if (fmaps_nchw && live) {
#pragma unroll
        for (int c = 0; c < 56; ++c) fmaps_nchw[((size_t)v * 56 + c) * HW + p] = px[3 + c];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p0 = blockIdx.x * 256 + wave * 64;                       // first pixel of this wave
    float* dst = cmaps + ((size_t)v * HW + p0) * 64;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int c = 0; c < 32; ++c) tr[wave][lane][c] = px[32 * half + c];
        // one wave writes and reads its own tile: no workgroup barrier needed (the LDS operations of a wave complete in order); the wave barriers
        // only keep the compiler from moving the cross-lane reads above the writes / the next half's writes above the reads
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pp = q * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            const float4 t = make_float4(tr[wave][pp][c4], tr[wave][pp][c4 + 1], tr[wave][pp][c4 + 2], tr[wave][pp][c4 + 3]);
            if (p0 + pp < HW) *reinterpret_cast<float4*>(dst + (size_t)pp * 64 + 32 * half + c4) = t;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_fpn_level_act(const float* fine, const float* fine_scale_shift, float slope, int c_in, const float* coarse, const float* weight,
                        const float* bias, int V, int H, int W, float* out, void* stream) {
    O2345_REQUIRE(fine && coarse && weight && bias && out, "fpn_level: null pointer");
    O2345_REQUIRE(H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 && V >= 1, "fpn_level: even map sizes required (got %d x %d)", H, W);
    const dim3 grid(cdiv((long long)H * W, 256), V);
    hipStream_t s = (hipStream_t)stream;
    if (c_in == 8) hipLaunchKernelGGL(k_fpn_level<8>, grid, dim3(256), 0, s, fine, coarse, weight, bias, H, W, out, fine_scale_shift, slope);
    else if (c_in == 16) hipLaunchKernelGGL(k_fpn_level<16>, grid, dim3(256), 0, s, fine, coarse, weight, bias, H, W, out, fine_scale_shift, slope);
    else O2345_REQUIRE(false, "fpn_level: FeatureNet's lateral layers have 8 or 16 input channels (got %d)", c_in);
    return check_launch("fpn_level");
}

int o2345_fpn_level(const float* fine, int c_in, const float* coarse, const float* weight, const float* bias, int V, int H, int W, float* out,
                    void* stream) {
    return o2345_fpn_level_act(fine, nullptr, 0.f, c_in, coarse, weight, bias, V, H, W, out, stream);
}

int o2345_pyramid_pack(const float* f2, const float* s1, const float* s0, const float* rgb, int V, int H, int W, float* fmaps_nchw, float* cmaps_nhwc64,
                       void* stream) {
    O2345_REQUIRE(f2 && s1 && s0 && rgb && cmaps_nhwc64, "pyramid_pack: null pointer");
    O2345_REQUIRE(H % 4 == 0 && W % 4 == 0 && H >= 4 && W >= 4 && V >= 1, "pyramid_pack: map sizes must be multiples of 4 (got %d x %d)", H, W);
    hipLaunchKernelGGL(k_pyramid_pack, dim3(cdiv((long long)H * W, 256), V), dim3(256), 0, (hipStream_t)stream, f2, s1, s0, rgb, H, W, fmaps_nchw, cmaps_nhwc64);
    return check_launch("pyramid_pack");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_featmaps() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_pyramid_pack));
}
}  // namespace o2345
