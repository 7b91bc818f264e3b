// 2-D convolutions of FeatureNet and of the compress layer as direct HIP kernels (models/featurenet.py:12-91 `ConvBnReLU`,
// `FeatureNet`; models/sparse_sdf_network.py:171-173 `compress_layer`).  nn.Conv2d + InPlaceABN pairs become ONE pass per layer:
//
//   * the convolution reads the RAW output of the previous convolution and applies that layer's batch-norm + leaky ReLU while it stages
//     the input tile in LDS (y = max(t, slope t), t = x scale + shift -- the same arithmetic as the stand-alone ABN kernel of sparse.hip);
//   * it writes its own raw output once and, for layers followed by InPlaceABN, the per-channel sum / sum of squares of its tile
//     (fp32 inside a wave, doubles across waves and blocks, fixed order: deterministic);  k_conv_stats_finish turns them into the
//     (scale, shift) pair the NEXT kernel applies on load.  The activated tensor is never materialised unless a caller asks for it.
//
// The channel counts are tiny (3..56 -> 8..32), so this is fp32 VALU work with the weights coming through the SCALAR cache
// ([cin][ky][kx][cout] packing: the cout weights of one tap are one s_load_dwordx8/x16, the FMAs take them as SGPR operands); a thread owns one
// output pixel and all (or a block's share of the) output channels.  8 views x 256^2: 17.8 GFLOP for all 16 convolutions.
#include "common.h"

namespace o2345 {

struct ConvArgs {
    const float* in;          // [V,CIN,Hi,Wi] raw producer output (or images)
    const float* in_ss;       // [2*CIN] scale | shift of the producer's ABN, or null (input used as is)
    float slope;
    const float* w;           // packed [CIN][K][K][COUT]
    const float* bias;        // [COUT] or null
    int Hi, Wi, Ho, Wo;
    float* out;               // [V,COUT,Ho,Wo]
    double* part;             // [COUT][nblk][2] per-block sum / sum of squares, or null
    int nbx, nblk;            // tiles per row, tiles per view * views (set by the launcher)
    const float* gamma; const float* beta; float eps; int abs_gamma; float* out_ss;        // batch-norm parameters of this layer (statistics pass)
    // input addressing: element (v, c, y, x) = in[v * view_stride + c * chan_stride + (y * Wi + x) * pix_stride]  (NCHW: Hi*Wi, 1; channel-last
    // maps such as the [V,H,W,64] colour map with its features at channel 3: 1, 64, `in` already advanced by the channel offset)
    long long view_stride, chan_stride; int pix_stride;
};

constexpr int CV_TX = 32, CV_TH = 8;          // threads of a block: 32 x 8; a thread owns PX horizontally adjacent output pixels

// CPB: output channels per block (blockIdx.z selects the group); CC: input channels per LDS stage; PX: output pixels per thread.
// Inside a stage a thread first pulls the input patch of one channel row out of LDS into registers and then runs the FMAs of that row with the
// weights as SGPR operands: PX * CPB FMAs per scalar-loaded weight vector, and no LDS instruction between the scalar loads of a row (LDS and
// scalar loads share one counter, interleaving them serialises every tap on the scalar-cache latency).
template <int CIN, int COUT, int CPB, int K, int STRIDE, int CC, int PX>
__global__ __launch_bounds__(256) void k_conv2d(ConvArgs a) {
    constexpr int TW = CV_TX * PX;                                                         // output pixels per tile row
    constexpr int PAD = K / 2, IW = (TW - 1) * STRIDE + K, IH = (CV_TH - 1) * STRIDE + K;
    constexpr int IWP = (IW + 3) / 4 * 4 + (PX == 1 ? 1 : 0);                              // PX > 1: rows stay 16-byte aligned for vector reads
    constexpr int NX = (PX - 1) * STRIDE + K;                                              // input values of a patch row
    static_assert(CIN % CC == 0 && COUT % CPB == 0, "channel grouping");
    __shared__ __attribute__((aligned(16))) float tile[CC][IH][IWP];
    __shared__ double red[4][CPB][2];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int bx = blockIdx.x % a.nbx, by = blockIdx.x / a.nbx, v = blockIdx.y, cg = blockIdx.z;
    const int ox = bx * TW + tx * PX, oy = by * CV_TH + ty;
    const int gx0 = bx * TW * STRIDE - PAD, gy0 = by * CV_TH * STRIDE - PAD;
    float acc[PX][CPB];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int co = 0; co < CPB; ++co) acc[p][co] = a.bias ? a.bias[cg * CPB + co] : 0.f;
    const float* src = a.in + (size_t)v * a.view_stride;
    const float* __restrict__ w = a.w + cg * CPB;
#pragma unroll 1
    for (int c0 = 0; c0 < CIN; c0 += CC) {
        __syncthreads();
        for (int i = threadIdx.x; i < CC * IH * IW; i += 256) {
            const int c = i / (IH * IW), r = i % (IH * IW), iy = r / IW, ix = r % IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            float t = 0.f;                                                     // zero padding applies to the ACTIVATED input
            if (gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) {
                t = src[(size_t)(c0 + c) * a.chan_stride + ((size_t)gy * a.Wi + gx) * a.pix_stride];
                if (a.in_ss) {
                    t = t * a.in_ss[c0 + c] + a.in_ss[CIN + c0 + c];
                    t = t >= 0.f ? t : t * a.slope;
                }
            }
            tile[c][iy][ix] = t;
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float x[NX];
                const float* row = &tile[c][ty * STRIDE + ky][tx * PX * STRIDE];
                if constexpr (PX > 1 && (PX * STRIDE) % 4 != 0) {            // 8-byte aligned patch rows
#pragma unroll
                    for (int q = 0; q + 2 <= NX; q += 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(row + q);
                        x[q] = t2.x; x[q + 1] = t2.y;
                    }
                    if constexpr (NX % 2) x[NX - 1] = row[NX - 1];
                } else if constexpr (PX > 1) {
#pragma unroll
                    for (int q = 0; q + 4 <= NX; q += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(row + q);
                        x[q] = t4.x; x[q + 1] = t4.y; x[q + 2] = t4.z; x[q + 3] = t4.w;
                    }
#pragma unroll
                    for (int q = NX / 4 * 4; q < NX; ++q) x[q] = row[q];
                } else {
#pragma unroll
                    for (int q = 0; q < NX; ++q) x[q] = row[q];
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* __restrict__ wt = w + (size_t)(((c0 + c) * K + ky) * K + kx) * COUT;
#pragma unroll
                    for (int co = 0; co < CPB; ++co) {
                        const float wv = wt[co];
#pragma unroll
                        for (int p = 0; p < PX; ++p) acc[p][co] = fmaf(wv, x[p * STRIDE + kx], acc[p][co]);
                    }
                }
            }
    }
    float* dst = a.out + (((size_t)v * COUT + cg * CPB) * a.Ho + oy) * a.Wo + ox;
    if (oy < a.Ho) {
#pragma unroll
        for (int p = 0; p < PX; ++p)
            if (ox + p < a.Wo) {
#pragma unroll
                for (int co = 0; co < CPB; ++co) dst[(size_t)co * a.Ho * a.Wo + p] = acc[p][co];
            }
    }
    if (!a.part) return;
    // per-channel sum and sum of squares of this tile: wave tree in fp32, doubles from there on
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int co = 0; co < CPB; ++co) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int p = 0; p < PX; ++p)
            if (oy < a.Ho && ox + p < a.Wo) { s += acc[p][co]; q = fmaf(acc[p][co], acc[p][co], q); }
#pragma unroll
        for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
        if (lane == 0) { red[wave][co][0] = (double)s; red[wave][co][1] = (double)q; }
    }
    __syncthreads();
    if (threadIdx.x < CPB * 2) {
        const int co = threadIdx.x >> 1, k = threadIdx.x & 1;
        const double t = (red[0][co][k] + red[1][co][k]) + (red[2][co][k] + red[3][co][k]);
        a.part[((size_t)(cg * CPB + co) * a.nblk + (size_t)v * gridDim.x + blockIdx.x) * 2 + k] = t;
    }
}

// one block per channel: batch statistics -> (scale, shift) of InPlaceABN (|gamma| + eps convention selectable, as in sparse.hip)
__global__ __launch_bounds__(256) void k_conv_stats_finish(const double* __restrict__ part, int nblk, double count, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int abs_gamma, float* __restrict__ scale_shift) {
    __shared__ double sm[2][4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) { s += part[((size_t)c * nblk + b) * 2]; q += part[((size_t)c * nblk + b) * 2 + 1]; }
    for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s; sm[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
        q = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        float g = gamma[c];
        if (abs_gamma) g = fabsf(g) + eps;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        scale_shift[c] = g * inv;
        scale_shift[C + c] = beta[c] - (float)mean * g * inv;
    }
}

// nn.Conv2d weight [COUT][CIN][K][K] -> [CIN][K][K][COUT]
__global__ void k_conv_pack(const float* __restrict__ w, int cout, int cin, int kk, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cout * cin * kk) return;
    const int co = i % cout, r = i / cout, t = r % kk, ci = r / kk;
    out[i] = w[((size_t)co * cin + ci) * kk + t];
}

// stand-alone application of a (scale, shift) pair + leaky ReLU: NCHW and / or channel-last output (the compress layer's feature maps)
template <int C>
__global__ __launch_bounds__(256) void k_ss_apply(const float* __restrict__ x /*[V,C,HW]*/, const float* __restrict__ ss, float slope, int HW,
                                                  float* __restrict__ y_nchw, float* __restrict__ y_nhwc) {
    __shared__ float tile[C][65];
    const int v = blockIdx.y, p0 = blockIdx.x * 64;
    const float* src = x + (size_t)v * C * HW;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int c = i / 64, p = i % 64;
        float t = 0.f;
        if (p0 + p < HW) {
            t = src[(size_t)c * HW + p0 + p] * ss[c] + ss[C + c];
            t = t >= 0.f ? t : t * slope;
            if (y_nchw) y_nchw[((size_t)v * C + c) * HW + p0 + p] = t;
        }
        tile[c][p] = t;
    }
    if (!y_nhwc) return;
    __syncthreads();
    float* dst = y_nhwc + (size_t)v * HW * C;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int p = i / C, c = i % C;
        if (p0 + p < HW) dst[(size_t)(p0 + p) * C + c] = tile[c][p];
    }
}

// ---- matrix-core form (default numerical mode, config.py "f16x3") ---------------------------------------------------------------------------
// The same convolution as an implicit GEMM  D[co][pixel] = sum over (tap, ci) W[co][ci][tap] * act(in)[ci][pixel + tap]  on
// v_mfma_f32_32x32x16_f16 in the split-f16 form of csrc/sparse_mfma.hip (hi*hi + hi*lo + lo*hi, fp32 accumulate, fp32-class accuracy).
// A wave owns ROWS row segments of 32 output pixels (B column = lane & 31) and all <= 32 output channels; a k step is (tap, 16-channel group):
// the two wave halves supply 8 input channels each.  A workgroup (4 waves, 32 x 4*ROWS output pixels) stages its whole input tile once, all
// channels: activation (the producer's ABN, on load) and the f16 split happen ONCE per staged value, the halves go to LDS as 16-byte items
// [hi|lo][channel octet][pixel] so that a B operand is one conflict-free ds_read_b128; one barrier, then the wave runs its K*K*CINP/16 steps.
// The A operands (weights, [tap][group][hi|lo][64 lanes][8 f16], packed by k_conv_pack_x3) stream from L2 through a buffer descriptor a few
// steps ahead -- 2 KB per step, identical for every wave of the grid.
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hh16x2 __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct AOpX { h16x8 hi, lo; };
__device__ __forceinline__ AOpX conv_a_fetch(__amdgpu_buffer_rsrc_t rs, int step, int lane) {
    AOpX r;
    r.hi = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, step * 2048, 0));
    r.lo = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, step * 2048 + 1024, 0));
    return r;
}

template <int CINP, int K, int STRIDE, int ROWS>
__global__ __launch_bounds__(256) void k_conv2d_x3(ConvArgs a, int cin, int cout) {
    constexpr int NU = CINP / 16, TH = 4 * ROWS, PAD = K / 2, IW = 31 * STRIDE + K, IH = (TH - 1) * STRIDE + K, NPIX = IH * IW;
    constexpr int NLD = (NPIX + 255) / 256;                                      // tile pixels per thread; a thread stages ALL channels of its pixels
    __shared__ float4 plane[2][2 * NU][NPIX];                                     // [hi | lo][channel octet][pixel]: a B operand is one 16-byte item
    static_assert(sizeof(plane) >= 4 * 32 * 2 * sizeof(double), "the statistics scratch reuses the tile memory");
    double (*red)[32][2] = reinterpret_cast<double (*)[32][2]>(&plane[0][0][0]);  // [wave][channel][sum | sum of squares], after the MFMAs
    __shared__ float ssl[2][CINP];                                                // the producer's (scale | shift): read per staged value, kept in LDS
    if (a.in_ss && threadIdx.x < 2 * CINP) {
        const int k = threadIdx.x / CINP, c = threadIdx.x % CINP;
        ssl[k][c] = c < cin ? a.in_ss[k * cin + c] : 0.f;
    }
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % a.nbx, by = blockIdx.x / a.nbx, v = blockIdx.y;
    const int ox = bx * 32 + j, oy0 = by * TH + wave * ROWS;
    const int gx0 = bx * 32 * STRIDE - PAD, gy0 = by * TH * STRIDE - PAD;
    const float* src = a.in + (size_t)v * a.view_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, K * K * NU * 2048, 0x00020000);
    float m1 = -1.f;
    asm volatile("" : "+v"(m1));                                                 // keeps fma(hi, -1, x) a v_fma_mix_f32 (see sdf_mlp_x3.hip)
    // ---- stage the whole input tile, all channel groups at once: ONE global round trip per tile (staging group by group exposed one HBM latency
    //      per group and two barriers -- the matrix pipe was busy 23 % of the time); loads are unconditional (clamped pixel / channel), what lies
    //      outside the image or beyond cin is zeroed after the activation (zero padding applies to the ACTIVATED input)
    {
        unsigned pix_src[NLD];                                                    // element offsets from `src` (a view has < 2^32 elements)
        bool pix_in[NLD];
#pragma unroll
        for (int jj = 0; jj < NLD; ++jj) {
            const int r = min((int)threadIdx.x + 256 * jj, NPIX - 1), iy = r / IW, ix = r % IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            pix_in[jj] = gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
            pix_src[jj] = (unsigned)(min(max(gy, 0), a.Hi - 1) * a.Wi + min(max(gx, 0), a.Wi - 1)) * (unsigned)a.pix_stride;
        }
        float pre[NLD][CINP];
#pragma unroll
        for (int jj = 0; jj < NLD; ++jj)
#pragma unroll
            for (int c = 0; c < CINP; ++c) pre[jj][c] = (src + (size_t)min(c, cin - 1) * a.chan_stride)[pix_src[jj]];     // wave-uniform channel plane
        __syncthreads();                                                         // ssl is in place
#pragma unroll
        for (int jj = 0; jj < NLD; ++jj) {
            const int r = threadIdx.x + 256 * jj;
            if (r < NPIX) {
#pragma unroll
                for (int oct = 0; oct < 2 * NU; ++oct) {
                    float x[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int c = 8 * oct + t;
                        float val = pre[jj][c];
                        if (a.in_ss) {
                            val = val * ssl[0][c] + ssl[1][c];
                            val = fmaxf(val, val * a.slope);                     // leaky ReLU, 0 <= slope < 1
                        }
                        x[t] = (pix_in[jj] && c < cin) ? val : 0.f;
                    }
                    union { h16x8 v8; h16x2 v2[4]; hh16x2 w2[4]; float4 f4; } bh, bl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bh.v2[q] = __builtin_amdgcn_cvt_pkrtz(x[2 * q], x[2 * q + 1]);
                        bl.v2[q] = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)bh.w2[q][0], m1, x[2 * q]), __builtin_fmaf((float)bh.w2[q][1], m1, x[2 * q + 1]));
                    }
                    plane[0][oct][r] = bh.f4;
                    plane[1][oct][r] = bl.f4;
                }
            }
        }
    }
    // ---- k steps (tap, channel group) in the order of the operand records; the A operands stream from L2 PD steps ahead (nothing else is in
    //      flight on the vector-memory counter, so a wait for one record waits for nothing else)
    constexpr int NS = K * K * NU, PD = NS < 4 ? NS : 4;
    AOpX abuf[PD];
#pragma unroll
    for (int st = 0; st < PD; ++st) abuf[st] = conv_a_fetch(rs, st, lane);
    f32x16 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
    __syncthreads();                                                             // the tile is staged
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const AOpX A = abuf[st % PD];
        if (st + PD < NS) abuf[st % PD] = conv_a_fetch(rs, st + PD, lane);
        const int tap = st / NU, u = st % NU, ky = tap / K, kx = tap % K;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int pix = ((wave * ROWS + r) * STRIDE + ky) * IW + j * STRIDE + kx;
            const h16x8 bh = __builtin_bit_cast(h16x8, plane[0][2 * u + h][pix]), bl = __builtin_bit_cast(h16x8, plane[1][2 * u + h][pix]);
            acc[r] = MFMA_F16(A.lo, bh, acc[r]);
            acc[r] = MFMA_F16(A.hi, bl, acc[r]);
            acc[r] = MFMA_F16(A.hi, bh, acc[r]);
        }
    }
    // register 4g + i of a lane holds output channel 8g + 4h + i of pixel column j
    float s[16], q2[16], bv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { s[q] = 0.f; q2[q] = 0.f; bv[q] = 0.f; }
    if (a.bias) {                                                                // all 16 loads in flight together (clamped index: no branch per value)
#pragma unroll
        for (int q = 0; q < 16; ++q) bv[q] = a.bias[min(8 * (q >> 2) + 4 * h + (q & 3), cout - 1)];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = oy0 + r;
        const bool live = oy < a.Ho && ox < a.Wo;
        float* dst = a.out + ((size_t)v * cout * a.Ho + (live ? oy : 0)) * a.Wo + (live ? ox : 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = 8 * (q >> 2) + 4 * h + (q & 3);
            const float val = acc[r][q] + bv[q];
            if (live && co < cout) {
                dst[(size_t)co * a.Ho * a.Wo] = val;
                s[q] += val; q2[q] = fmaf(val, val, q2[q]);
            }
        }
    }
    if (!a.part) return;
    __syncthreads();                                                             // every wave is done with the tile: its memory becomes `red`
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int off = 16; off; off >>= 1) { s[q] += __shfl_xor(s[q], off); q2[q] += __shfl_xor(q2[q], off); }
        if (j == 0) { const int co = 8 * (q >> 2) + 4 * h + (q & 3); red[wave][co][0] = (double)s[q]; red[wave][co][1] = (double)q2[q]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * cout) {
        const int co = threadIdx.x >> 1, k = threadIdx.x & 1;
        const double t = (red[0][co][k] + red[1][co][k]) + (red[2][co][k] + red[3][co][k]);
        a.part[((size_t)co * a.nblk + (size_t)v * gridDim.x + blockIdx.x) * 2 + k] = t;
    }
}

// The same kernel with the input tile staged one 16-channel group at a time (two barriers and one global round trip per group, the next group's
// raw values in flight during the current group's MFMAs): for 64 input channels the whole tile (87 KB with two rows per wave) would leave one
// workgroup per CU, and one row per wave costs more in halo loads than the single round trip saves (measured on the compress layer: 181 us with
// the whole tile staged at once and one row per wave, 110 us staged by groups).
template <int CINP, int K, int STRIDE, int ROWS, bool CL>
__global__ __launch_bounds__(256) void k_conv2d_x3_staged(ConvArgs a, int cin, int cout) {
    constexpr int NU = CINP / 16, TH = 4 * ROWS, PAD = K / 2, IW = 31 * STRIDE + K, IH = (TH - 1) * STRIDE + K, NPIX = IH * IW;
    // staging work of a thread per 16-channel group.  Channel-first source: NLD tile pixels, all 16 channels of each (consecutive lanes = consecutive
    // pixels of one channel plane: coalesced).  Channel-LAST source (CL, chan_stride = 1): one channel (lane & 15) of NLD tile pixels, 16 pixels per
    // pass of the workgroup -- consecutive lanes read consecutive channels, a wave load covers 4 pixels x 64 contiguous bytes instead of 64 pixels
    // 256 bytes apart (measured on the compress layer reading the [V,H,W,64] colour map: 168 -> 118 us per call at 8 views, 615 -> 412 us at 32)
    constexpr int NLD = CL ? (NPIX + 15) / 16 : (NPIX + 255) / 256;
    __shared__ float4 plane[2][2][NPIX];                                          // [hi | lo][channel octet][pixel]: a B operand is one 16-byte item
    __shared__ double red[4][32][2];
    __shared__ float ssl[2][CINP];                                                // the producer's (scale | shift): read per staged value, kept in LDS
    if (a.in_ss && threadIdx.x < 2 * CINP) {
        const int k = threadIdx.x / CINP, c = threadIdx.x % CINP;
        ssl[k][c] = c < cin ? a.in_ss[k * cin + c] : 0.f;
    }
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int bx = blockIdx.x % a.nbx, by = blockIdx.x / a.nbx, v = blockIdx.y;
    const int ox = bx * 32 + j, oy0 = by * TH + wave * ROWS;
    const int gx0 = bx * 32 * STRIDE - PAD, gy0 = by * TH * STRIDE - PAD;
    const float* src = a.in + (size_t)v * a.view_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, K * K * NU * 2048, 0x00020000);
    float m1 = -1.f;
    asm volatile("" : "+v"(m1));                                                 // keeps fma(hi, -1, x) a v_fma_mix_f32 (see sdf_mlp_x3.hip)
    f32x16 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
    // the pixels this thread stages: clamped source address (loads are unconditional), in-image flag (zero padding applies to the ACTIVATED input)
    unsigned pix_src[NLD];                                                        // element offsets from `src` (a view has < 2^32 elements)
    bool pix_in[NLD];
    const int my_c = threadIdx.x & 15;                                            // CL: this thread's channel inside a group
#pragma unroll
    for (int jj = 0; jj < NLD; ++jj) {
        const int r = min(CL ? (int)(threadIdx.x >> 4) + 16 * jj : (int)threadIdx.x + 256 * jj, NPIX - 1), iy = r / IW, ix = r % IW;
        const int gy = gy0 + iy, gx = gx0 + ix;
        pix_in[jj] = gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
        pix_src[jj] = (unsigned)(min(max(gy, 0), a.Hi - 1) * a.Wi + min(max(gx, 0), a.Wi - 1)) * (unsigned)a.pix_stride;
    }
    float pre[NLD][CL ? 1 : 16];
    auto fetch = [&](int u) {
#pragma unroll
        for (int jj = 0; jj < NLD; ++jj) {
            if constexpr (CL) pre[jj][0] = src[pix_src[jj] + (unsigned)min(16 * u + my_c, cin - 1)];
            else {
#pragma unroll
                for (int t = 0; t < 16; ++t) pre[jj][t] = (src + (size_t)min(16 * u + t, cin - 1) * a.chan_stride)[pix_src[jj]];   // wave-uniform channel plane
            }
        }
    };
    fetch(0);
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
        __syncthreads();                                                         // the previous group's MFMAs have read their operands
        if constexpr (CL) {
            const int c = 16 * u + my_c;
            const float sc = a.in_ss ? ssl[0][c] : 1.f, sh = a.in_ss ? ssl[1][c] : 0.f;
            _Float16* const ph = reinterpret_cast<_Float16*>(&plane[0][my_c >> 3][0]) + (my_c & 7);
            _Float16* const pl = reinterpret_cast<_Float16*>(&plane[1][my_c >> 3][0]) + (my_c & 7);
#pragma unroll
            for (int jj = 0; jj < NLD; ++jj) {
                const int r = (int)(threadIdx.x >> 4) + 16 * jj;
                if (r < NPIX) {
                    float val = pre[jj][0];
                    if (a.in_ss) {
                        val = val * sc + sh;
                        val = fmaxf(val, val * a.slope);
                    }
                    val = (pix_in[jj] && c < cin) ? val : 0.f;
                    union { h16x2 v2; hh16x2 w2; } bh, bl;
                    bh.v2 = __builtin_amdgcn_cvt_pkrtz(val, 0.f);
                    bl.v2 = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)bh.w2[0], m1, val), 0.f);
                    ph[8 * r] = bh.w2[0];
                    pl[8 * r] = bl.w2[0];
                }
            }
        } else {
#pragma unroll
        for (int jj = 0; jj < NLD; ++jj) {
            const int r = threadIdx.x + 256 * jj;
            if (r < NPIX) {
#pragma unroll
                for (int oct = 0; oct < 2; ++oct) {
                    float x[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int c = 16 * u + 8 * oct + t;
                        float val = pre[jj][8 * oct + t];
                        if (a.in_ss) {
                            val = val * ssl[0][c] + ssl[1][c];
                            val = fmaxf(val, val * a.slope);                     // leaky ReLU, 0 <= slope < 1
                        }
                        x[t] = (pix_in[jj] && c < cin) ? val : 0.f;
                    }
                    union { h16x8 v8; h16x2 v2[4]; hh16x2 w2[4]; float4 f4; } bh, bl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bh.v2[q] = __builtin_amdgcn_cvt_pkrtz(x[2 * q], x[2 * q + 1]);
                        bl.v2[q] = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)bh.w2[q][0], m1, x[2 * q]), __builtin_fmaf((float)bh.w2[q][1], m1, x[2 * q + 1]));
                    }
                    plane[0][oct][r] = bh.f4;
                    plane[1][oct][r] = bl.f4;
                }
            }
        }
        }
        __syncthreads();
        // A operands of the first taps, THEN the next group's raw values, then the MFMAs: the vector-memory counter retires in order, so a wait for
        // an A operand issued after the prefetch would wait for the whole prefetch
        constexpr int KK = K * K, NA0 = KK <= 9 ? KK : 13;
        AOpX aop[NA0];
#pragma unroll
        for (int tap = 0; tap < NA0; ++tap) aop[tap] = conv_a_fetch(rs, tap * NU + u, lane);
        if (u + 1 < NU) fetch(u + 1);                                            // in flight during this group's MFMAs
        auto taps = [&](int tap, const AOpX& A) {
            const int ky = tap / K, kx = tap % K;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int pix = ((wave * ROWS + r) * STRIDE + ky) * IW + j * STRIDE + kx;
                const h16x8 bh = __builtin_bit_cast(h16x8, plane[0][h][pix]), bl = __builtin_bit_cast(h16x8, plane[1][h][pix]);
                acc[r] = MFMA_F16(A.lo, bh, acc[r]);
                acc[r] = MFMA_F16(A.hi, bl, acc[r]);
                acc[r] = MFMA_F16(A.hi, bh, acc[r]);
            }
        };
#pragma unroll
        for (int tap = 0; tap < NA0; ++tap) taps(tap, aop[tap]);
        if constexpr (KK > NA0) {
            AOpX bop[KK - NA0];
#pragma unroll
            for (int tap = NA0; tap < KK; ++tap) bop[tap - NA0] = conv_a_fetch(rs, tap * NU + u, lane);
#pragma unroll
            for (int tap = NA0; tap < KK; ++tap) taps(tap, bop[tap - NA0]);
        }
    }
    // register 4g + i of a lane holds output channel 8g + 4h + i of pixel column j
    float s[16], q2[16], bv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { s[q] = 0.f; q2[q] = 0.f; bv[q] = 0.f; }
    if (a.bias) {                                                                // all 16 loads in flight together (clamped index: no branch per value)
#pragma unroll
        for (int q = 0; q < 16; ++q) bv[q] = a.bias[min(8 * (q >> 2) + 4 * h + (q & 3), cout - 1)];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = oy0 + r;
        const bool live = oy < a.Ho && ox < a.Wo;
        float* dst = a.out + ((size_t)v * cout * a.Ho + (live ? oy : 0)) * a.Wo + (live ? ox : 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = 8 * (q >> 2) + 4 * h + (q & 3);
            const float val = acc[r][q] + bv[q];
            if (live && co < cout) {
                dst[(size_t)co * a.Ho * a.Wo] = val;
                s[q] += val; q2[q] = fmaf(val, val, q2[q]);
            }
        }
    }
    if (!a.part) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int off = 16; off; off >>= 1) { s[q] += __shfl_xor(s[q], off); q2[q] += __shfl_xor(q2[q], off); }
        if (j == 0) { const int co = 8 * (q >> 2) + 4 * h + (q & 3); red[wave][co][0] = (double)s[q]; red[wave][co][1] = (double)q2[q]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * cout) {
        const int co = threadIdx.x >> 1, k = threadIdx.x & 1;
        const double t = (red[0][co][k] + red[1][co][k]) + (red[2][co][k] + red[3][co][k]);
        a.part[((size_t)co * a.nblk + (size_t)v * gridDim.x + blockIdx.x) * 2 + k] = t;
    }
}

// nn.Conv2d weight [cout][cin][K][K] -> A operands [tap][group][hi|lo][64 lanes][8 f16]: lane (i = lane & 31, h = lane >> 5) of step (tap, u) holds
// W[i][16u + 8h + t][tap], t = 0..7 (zeros beyond cout / cin); hi = f16(w), lo = f16(w - hi), round to nearest (weights.f16_split)
__global__ void k_conv_pack_x3(const float* __restrict__ w, int cout, int cin, int kk, int nu, _Float16* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;                                 // one (step, lane, t)
    if (i >= kk * nu * 64 * 8) return;
    const int t = i & 7, lane = (i >> 3) & 63, step = i >> 9, u = step % nu, tap = step / nu;
    const int co = lane & 31, c = 16 * u + 8 * (lane >> 5) + t;
    const float x = (co < cout && c < cin) ? w[((size_t)co * cin + c) * kk + tap] : 0.f;
    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
    out[(size_t)step * 1024 + lane * 8 + t] = hi;
    out[(size_t)step * 1024 + 512 + lane * 8 + t] = lo;
}

template <int CINP, int K, int STRIDE, int ROWS, bool STAGED = false>
static void launch_conv_x3(ConvArgs a, int V, int cin, int cout, hipStream_t s) {
    a.nbx = (int)cdiv(a.Wo, 32);
    const int nby = (int)cdiv(a.Ho, 4 * ROWS);
    a.nblk = a.nbx * nby * V;                                                       // <= cdiv(Wo, 32) * cdiv(Ho, 4) * V (workspace bound)
    if constexpr (STAGED) {
        if (a.chan_stride == 1) hipLaunchKernelGGL((k_conv2d_x3_staged<CINP, K, STRIDE, ROWS, true>), dim3(a.nbx * nby, V), dim3(256), 0, s, a, cin, cout);
        else hipLaunchKernelGGL((k_conv2d_x3_staged<CINP, K, STRIDE, ROWS, false>), dim3(a.nbx * nby, V), dim3(256), 0, s, a, cin, cout);
    }
    else hipLaunchKernelGGL((k_conv2d_x3<CINP, K, STRIDE, ROWS>), dim3(a.nbx * nby, V), dim3(256), 0, s, a, cin, cout);
    if (a.part)
        hipLaunchKernelGGL(k_conv_stats_finish, dim3(cout), dim3(256), 0, s, a.part, a.nblk, (double)V * a.Ho * a.Wo, cout, a.gamma, a.beta, a.eps, a.abs_gamma, a.out_ss);
}

template <int CIN, int COUT, int CPB, int K, int STRIDE, int CC, int PX>
static void launch_conv(ConvArgs a, int V, hipStream_t s) {
    a.nbx = (int)cdiv(a.Wo, CV_TX * PX);
    a.nblk = a.nbx * (int)cdiv(a.Ho, CV_TH) * V;
    hipLaunchKernelGGL((k_conv2d<CIN, COUT, CPB, K, STRIDE, CC, PX>), dim3(a.nbx * cdiv(a.Ho, CV_TH), V, COUT / CPB), dim3(256), 0, s, a);
    if (a.part)
        hipLaunchKernelGGL(k_conv_stats_finish, dim3(COUT), dim3(256), 0, s, a.part, a.nblk, (double)V * a.Ho * a.Wo, COUT, a.gamma, a.beta, a.eps, a.abs_gamma, a.out_ss);
}

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_conv2d_pack_weights(const float* w_oihw, int cout, int cin, int k, float* packed, void* stream) {
    O2345_REQUIRE(w_oihw && packed && cout > 0 && cin > 0 && k > 0, "conv2d_pack_weights: bad arguments");
    hipLaunchKernelGGL(k_conv_pack, dim3(cdiv((long long)cout * cin * k * k, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, k * k, packed);
    return check_launch("conv2d_pack_weights");
}

size_t o2345_conv2d_workspace_bytes(int V, int cout, int Ho, int Wo) {
    // one (sum, sum of squares) pair of doubles per channel and block; the smallest tile of any kernel here is 32 x 4 output pixels
    // (k_conv2d_x3 with one row per wave, the stride-2 layers), so this bounds the block count of every variant
    return (size_t)cout * cdiv(Wo, 32) * cdiv(Ho, 4) * V * 2 * sizeof(double);
}

// out = conv2d(act(in), w) (+ bias); act = leaky(in * scale + shift) when in_scale_shift is given.  padding = k / 2.
// When gamma / beta / out_scale_shift are given, the batch statistics of `out` over (V, Ho, Wo) are reduced and out_scale_shift [2*cout] receives
// the InPlaceABN (scale, shift) of this layer -- to be applied by the consumer (o2345_conv2d / o2345_fpn_level_act / o2345_scale_shift_act).
static void set_input_layout(ConvArgs& a, int cin, int in_pixel_stride, int in_channel_offset) {
    if (in_pixel_stride <= 0) { a.view_stride = (long long)cin * a.Hi * a.Wi; a.chan_stride = (long long)a.Hi * a.Wi; a.pix_stride = 1; }
    else { a.view_stride = (long long)a.Hi * a.Wi * in_pixel_stride; a.chan_stride = 1; a.pix_stride = in_pixel_stride; a.in += in_channel_offset; }
}

int o2345_conv2d(const float* in, int V, int cin, int Hi, int Wi, int in_pixel_stride, int in_channel_offset, const float* in_scale_shift, float slope,
                 const float* w_packed, const float* bias, int cout, int k, int stride, float* out, const float* gamma, const float* beta, float eps,
                 int abs_gamma, float* out_scale_shift, void* workspace, size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(in && w_packed && out && V >= 1 && Hi >= 1 && Wi >= 1, "conv2d: bad arguments");
    O2345_REQUIRE(in_pixel_stride <= 0 || (in_channel_offset >= 0 && in_channel_offset + cin <= in_pixel_stride), "conv2d: channel-last input: offset + cin must fit the pixel stride");
    O2345_REQUIRE(stride == 1 || stride == 2, "conv2d: stride 1 or 2 (got %d)", stride);
    const int pad = k / 2;
    const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
    const bool stats = out_scale_shift != nullptr;
    if (stats) {
        O2345_REQUIRE(gamma && beta && workspace, "conv2d: batch statistics need gamma, beta and a workspace");
        O2345_REQUIRE(workspace_bytes >= o2345_conv2d_workspace_bytes(V, cout, Ho, Wo), "conv2d: workspace too small");
    }
    ConvArgs a{in, in_scale_shift, slope, w_packed, bias, Hi, Wi, Ho, Wo, out, stats ? (double*)workspace : nullptr, 0, 0,
               gamma, beta, eps, abs_gamma, out_scale_shift, 0, 0, 0};
    set_input_layout(a, cin, in_pixel_stride, in_channel_offset);
    hipStream_t s = (hipStream_t)stream;
    // pixels per thread / output channels per block by map size, so that a launch covers the chip (1024 SIMDs) a few times over:
    // 4 pixels x <= 16 channels on large maps (most FMAs per scalar weight load), 1 pixel x 8 channels on small ones
    const long long pix = (long long)V * Ho * Wo;
    const int tier = pix >= 400000 ? 2 : pix >= 100000 ? 1 : 0;
    bool done = false;
#define O2345_CONV(CI, CO, KK, ST, CC, PX4, B4, B2, B1)                                   \
    if (!done && cin == CI && cout == CO && k == KK && stride == ST) {                    \
        done = true;                                                                      \
        if (tier == 2) launch_conv<CI, CO, B4, KK, ST, CC, PX4>(a, V, s);                 \
        else if (tier == 1) launch_conv<CI, CO, B2, KK, ST, CC, 2>(a, V, s);              \
        else launch_conv<CI, CO, B1, KK, ST, CC, 1>(a, V, s);                             \
    }
    // (PX4: pixels per thread on the largest maps -- 2 for the 5x5 stride-2 layers, whose 4-pixel input tile would not fit 64 KB of LDS)
    O2345_CONV(3, 8, 3, 1, 3, 4, 8, 8, 8)
    O2345_CONV(8, 8, 3, 1, 8, 4, 8, 8, 8)
    O2345_CONV(8, 16, 5, 2, 4, 2, 16, 16, 8)
    O2345_CONV(16, 16, 3, 1, 8, 4, 16, 16, 8)
    O2345_CONV(16, 32, 5, 2, 4, 2, 16, 16, 8)
    O2345_CONV(32, 32, 3, 1, 8, 4, 16, 16, 8)
    O2345_CONV(32, 32, 1, 1, 8, 4, 16, 16, 8)
    O2345_CONV(32, 16, 3, 1, 8, 4, 16, 16, 8)
    O2345_CONV(32, 8, 3, 1, 8, 4, 8, 8, 8)
    O2345_CONV(56, 16, 3, 1, 8, 4, 16, 16, 8)
    O2345_CONV(56, 8, 3, 1, 8, 4, 8, 8, 8)               // a compress layer with d_pyramid_feature_compress = 8
#undef O2345_CONV
    O2345_REQUIRE(done, "conv2d: no kernel for %d -> %d channels, %dx%d, stride %d (FeatureNet / compress-layer shapes only)", cin, cout, k, k, stride);
    return check_launch("conv2d");
}

size_t o2345_conv2d_x3_weight_floats(int cin, int k) { return (size_t)k * k * ((cin + 15) / 16) * 512; }

int o2345_conv2d_pack_weights_x3(const float* w_oihw, int cout, int cin, int k, float* packed, void* stream) {
    O2345_REQUIRE(w_oihw && packed && cout > 0 && cout <= 32 && cin > 0 && k > 0, "conv2d_pack_weights_x3: bad arguments (cout <= 32)");
    const int nu = (cin + 15) / 16;
    hipLaunchKernelGGL(k_conv_pack_x3, dim3(cdiv((long long)k * k * nu * 512, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, k * k, nu, (_Float16*)packed);
    return check_launch("conv2d_pack_weights_x3");
}

// the same function as o2345_conv2d on the matrix cores (split-f16, fp32 accumulate); w_packed_x3 from o2345_conv2d_pack_weights_x3
int o2345_conv2d_x3(const float* in, int V, int cin, int Hi, int Wi, int in_pixel_stride, int in_channel_offset, const float* in_scale_shift, float slope,
                    const float* w_packed_x3, const float* bias, int cout, int k, int stride, float* out, const float* gamma, const float* beta, float eps,
                    int abs_gamma, float* out_scale_shift, void* workspace, size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(in && w_packed_x3 && out && V >= 1 && Hi >= 1 && Wi >= 1, "conv2d_x3: bad arguments");
    O2345_REQUIRE(in_pixel_stride <= 0 || (in_channel_offset >= 0 && in_channel_offset + cin <= in_pixel_stride), "conv2d_x3: channel-last input: offset + cin must fit the pixel stride");
    O2345_REQUIRE(cout >= 1 && cout <= 32 && cin >= 1 && cin <= 64, "conv2d_x3: at most 64 input and 32 output channels (got %d -> %d)", cin, cout);
    O2345_REQUIRE((long long)Hi * Wi * (in_pixel_stride > 0 ? in_pixel_stride : 1) < (1ll << 31), "conv2d_x3: a view must have fewer than 2^31 elements per channel plane");
    const int pad = k / 2;
    const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
    const bool stats = out_scale_shift != nullptr;
    if (stats) {
        O2345_REQUIRE(gamma && beta && workspace, "conv2d_x3: batch statistics need gamma, beta and a workspace");
        O2345_REQUIRE(workspace_bytes >= o2345_conv2d_workspace_bytes(V, cout, Ho, Wo), "conv2d_x3: workspace too small");
    }
    ConvArgs a{in, in_scale_shift, slope, w_packed_x3, bias, Hi, Wi, Ho, Wo, out, stats ? (double*)workspace : nullptr, 0, 0,
               gamma, beta, eps, abs_gamma, out_scale_shift, 0, 0, 0};
    set_input_layout(a, cin, in_pixel_stride, in_channel_offset);
    hipStream_t s = (hipStream_t)stream;
    const int cinp = (cin + 15) / 16 * 16;
    bool done = true;
    // two rows of 32 pixels per wave: four rows (276 registers, one wave per SIMD) measured 138 vs 112 us on the compress layer -- the stages are
    // bound by the latency of their loads, occupancy matters more than operand reuse
    if (k == 3 && stride == 1 && cinp == 16) launch_conv_x3<16, 3, 1, 2>(a, V, cin, cout, s);
    else if (k == 3 && stride == 1 && cinp == 32) launch_conv_x3<32, 3, 1, 2>(a, V, cin, cout, s);
    else if (k == 3 && stride == 1 && cinp == 64) launch_conv_x3<64, 3, 1, 2, true>(a, V, cin, cout, s);        // staged by channel groups (see above)
    else if (k == 5 && stride == 2 && cinp == 16) launch_conv_x3<16, 5, 2, 1>(a, V, cin, cout, s);
    else if (k == 1 && stride == 1 && cinp == 32) launch_conv_x3<32, 1, 1, 2>(a, V, cin, cout, s);
    else done = false;
    O2345_REQUIRE(done, "conv2d_x3: no kernel for %d -> %d channels, %dx%d, stride %d (FeatureNet / compress-layer shapes only)", cin, cout, k, k, stride);
    return check_launch("conv2d_x3");
}

int o2345_scale_shift_act(const float* x, int V, int C, int H, int W, const float* scale_shift, float slope, float* y_nchw, float* y_nhwc, void* stream) {
    O2345_REQUIRE(x && scale_shift && (y_nchw || y_nhwc), "scale_shift_act: null pointer");
    O2345_REQUIRE(C == 8 || C == 16 || C == 32, "scale_shift_act: C must be 8, 16 or 32 (got %d)", C);
    const long long HW = (long long)H * W;
    const dim3 grid(cdiv(HW, 64), V);
    hipStream_t s = (hipStream_t)stream;
    if (C == 32) hipLaunchKernelGGL(k_ss_apply<32>, grid, dim3(256), 0, s, x, scale_shift, slope, (int)HW, y_nchw, y_nhwc);
    else if (C == 16) hipLaunchKernelGGL(k_ss_apply<16>, grid, dim3(256), 0, s, x, scale_shift, slope, (int)HW, y_nchw, y_nhwc);
    else hipLaunchKernelGGL(k_ss_apply<8>, grid, dim3(256), 0, s, x, scale_shift, slope, (int)HW, y_nchw, y_nhwc);
    return check_launch("scale_shift_act");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_convnet() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_conv_stats_finish));
}
}  // namespace o2345
