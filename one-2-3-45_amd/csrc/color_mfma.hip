// Image-based colour blending on the matrix cores: the same function as csrc/color.hip
// (Projector.compute / compute_view_independent + GeneralRenderingNetwork.forward, models/projector.py:96-425,
// models/rendering_network.py:75-129) with every per-(point,view) linear layer on v_mfma_f32_32x32x2_f32.
//
// A wave owns 32 columns = 32/G points x G source views (G = pow2 >= V, V <= 32).  Column j = lane & 31; the two wave
// halves h = lane >> 5 hold the SAME column and supply the two k rows of each MFMA step.  As in csrc/sdf_mlp.hip the
// MFMA result layout (neuron 32b + (r&3) + 8(r>>2) + 4h in register r of block b) is used as the k enumeration of the
// next layer, so activations stay in registers through the whole network; the weights are pre-permuted on the host
// (weights.pack_color_mfma_blob, checked lane-by-lane by tests/test_weights_packing.py).
//   * the 64 floats of a pixel of the channel-last map [V,H,W,64] (rgb | 56 features | pad) are split 32|32 between the
//     halves: each lane gathers 8 dwordx4 per bilinear tap and owns those channels for the whole kernel
//   * reductions over views (min, weighted mean / variance, softmax) are xor-shuffles inside the G-lane group
//   * the view-independent rows of base_fc (geo | mean | var: 134 of 193 inputs) are evaluated once per point: the 2G
//     lanes of a point split the 64 outputs, exchange them through LDS, and they enter the MFMA accumulators as bias
//   * all weight blobs (58 KB of A operands + 37 KB shared rows) are staged in LDS once per persistent workgroup
//
// Two instantiations share everything but the matrix step (template flag X3):
//   X3 = false  v_mfma_f32_32x32x2_f32, the exact fp32 chain (189 MFMAs of 64 cycles per tile)
//   X3 = true   split-f16 operands as in csrc/sdf_mlp_x3.hip: x = hi + lo (two f16 halves, 22 bits), products accumulated
//               in fp32 as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (75 MFMAs of 32 cycles per tile).  The per-half
//               k enumeration is the same (register r of the fp32 form = slot 8s+t of step s), so the x3 blob is a regrouping
//               of the fp32 one (weights.pack_color_x3_blob).
#include "color_net.h"

namespace o2345 {

// k_color_mfma is a TEST-ONLY build variant since round 4 (-DO2345_TILES_KERNEL, build.build_variant("tiles", ["-DO2345_TILES_KERNEL"])): it
// loses against k_color_pts at every view count (45.7 - 47.7 vs 36.1 - 37.0 ms at 8 views, 52.3 vs 37.1 ms at 32) and the product library does not carry it.
#ifdef O2345_TILES_KERNEL
template <int G, bool X3>
__global__ __launch_bounds__(768) void k_color_mfma(ColorMArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PPT = 32 / G;                 // points per wave tile
    constexpr int OPV = 64 / G;                 // shared-part outputs per view lane (per half)
    constexpr int PST = 2 * 64 + 4;             // per-point stride of the exchange buffer: +4 floats so that the points of a tile map to
                                                // different LDS banks (the 32/G points read the same offsets 512 B apart otherwise)
    constexpr int SB = PPT * PST;               // floats of the per-wave exchange buffer
    constexpr int TOTAL = X3 ? CX_TOTAL : CM_TOTAL;        // floats of the staged blob
    constexpr int TAIL = X3 ? CX_A_END - CM_BIAS0 : 0;      // shift of the fp32 tail (biases, shared rows, scalars)
    for (int i = threadIdx.x * 4; i < TOTAL; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(a.blob + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, ptl = j / G, v = j % G;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float* sbuf = lds + TOTAL + wave * SB;
    const long long n = a.n_dev ? (long long)*a.n_dev : a.n;
    const float m1 = X3 ? opaque_minus_one() : -1.f;
    const float s_abs = fabsf(lds[TAIL + CM_S]) * LOG2E;
    const int base_prio = (a.sched & 1) ? (wave >> 2) : 0;          // waves w, w + 4, w + 8 share a SIMD (cyclic SIMD assignment)
    if (a.sched & 1) set_wave_prio(base_prio);
    const TileSched ts = tile_schedule(n, PPT, wave, nwave);
    for (long long tile = ts.first; tile < ts.end; tile += ts.stride) {
        const long long t0 = tile * PPT;
        const long long i = t0 + ptl;
        const bool live = i < n;
        const long long slot = live ? (a.index ? (long long)a.index[i] : i) : 0;
        const float px = live ? a.pts[3 * slot] : 0.f, py = live ? a.pts[3 * slot + 1] : 0.f, pz = live ? a.pts[3 * slot + 2] : 0.f;
        const bool view_ok = v < a.V;
        const int vv = view_ok ? v : 0;
        // ---- geometry feature (all 16 channels; needed by the shared rows) and validity ---------------------------------
        // the 8 trilinear taps are split over the view lanes of the point (G/.. lanes each load ONE 64-byte voxel), then
        // summed over the group: 4 loads per lane instead of 32 identical ones in every lane
        float geo[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) geo[c] = 0.f;
        float msum = 0.f;
        {
            const Axis2 ax = axis_taps_zeros(px, a.D), ay = axis_taps_zeros(py, a.D), az = axis_taps_zeros(pz, a.D);
            constexpr int TPL = G >= 8 ? 1 : 8 / G;        // taps per lane
#pragma unroll
            for (int tt = 0; tt < TPL; ++tt) {
                const int tap = v * TPL + tt;               // 0..7 for v < 8/TPL; lanes beyond carry no tap
                const int ia = (tap >> 2) & 1, ib = (tap >> 1) & 1, ic = tap & 1;
                const float w = (tap < 8) ? (ia ? ax.w[1] : ax.w[0]) * (ib ? ay.w[1] : ay.w[0]) * (ic ? az.w[1] : az.w[0]) : 0.f;
                if (w != 0.f) {
                    const size_t vox = ((size_t)(ia ? ax.i[1] : ax.i[0]) * a.D + (ib ? ay.i[1] : ay.i[0])) * a.D + (ic ? az.i[1] : az.i[0]);
                    msum += w * a.maskvol[vox];
                    const float4* p4 = reinterpret_cast<const float4*>(a.vol_cl + vox * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t = p4[q];
                        geo[4 * q] = fmaf(t.x, w, geo[4 * q]); geo[4 * q + 1] = fmaf(t.y, w, geo[4 * q + 1]);
                        geo[4 * q + 2] = fmaf(t.z, w, geo[4 * q + 2]); geo[4 * q + 3] = fmaf(t.w, w, geo[4 * q + 3]);
                    }
                }
            }
            msum = gsum<G>(msum);
#pragma unroll
            for (int c = 0; c < 16; ++c) geo[c] = gsum<G>(geo[c]);
        }
        const bool gvalid = fabsf(px) < 1.f && fabsf(py) < 1.f && fabsf(pz) < 1.f && msum > 0.f;
        // geometry part of the view-independent rows right away (half 0 owns it): OPV partial sums stay live instead of 16 channels
        float sacc[OPV];
#pragma unroll
        for (int o = 0; o < OPV; ++o) sacc[o] = 0.f;
        if (h == 0) {
            const float* WG = lds + TAIL + CM_W_S + v * OPV;
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int o = 0; o < OPV; ++o) sacc[o] = fmaf(geo[c], WG[c * 64 + o], sacc[o]);
        }
        // ---- projection into this lane's view; this half's 32 pixel floats ----------------------------------------------------
        float gx, gy;
        cm_project(a.proj + 12 * vv, px, py, pz, a.H, a.W_img, gx, gy);
        const float m = (view_ok && gvalid && fabsf(gx) < 1.f && fabsf(gy) < 1.f) ? 1.f : 0.f;
        float rf[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) rf[c] = 0.f;
        if (a.sched & 2) set_wave_prio(3);
        {
            const Taps2D tp = bilinear_taps(gx, gy, a.H, a.W_img);
            const float4* img = reinterpret_cast<const float4*>(a.cmaps + (size_t)vv * a.H * a.W_img * 64) + 8 * h;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tp.w[k] != 0.f) {
                    const float4* px4 = img + (size_t)tp.idx[k] * 16;
                    const float wk = tp.w[k] * LOG2E;                       // pixel floats enter the network in the scaled domain
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 t = px4[q];
                        rf[4 * q] = fmaf(t.x, wk, rf[4 * q]); rf[4 * q + 1] = fmaf(t.y, wk, rf[4 * q + 1]);
                        rf[4 * q + 2] = fmaf(t.z, wk, rf[4 * q + 2]); rf[4 * q + 3] = fmaf(t.w, wk, rf[4 * q + 3]);
                    }
                }
        }
        if (a.sched & 2) set_wave_prio(base_prio);
        const float rgb0 = rf[0], rgb1 = rf[1], rgb2 = rf[2];      // log2(e) * colours (meaningful in half 0), before the direction feature
        // ---- ray direction difference ------------------------------------------------------------------------------------------
        float rd[4];
        {
            float qx, qy, qz;
            if (a.normals) {
                const float nx = a.normals[3 * slot], ny = a.normals[3 * slot + 1], nz = a.normals[3 * slot + 2];
                const float rn = crcp(fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f));
                qx = nx * rn; qy = ny * rn; qz = nz * rn;
            } else {
                const float tx = a.query_cam[0] - px, ty = a.query_cam[1] - py, tz = a.query_cam[2] - pz;
                const float rn = crcp(sqrtf(tx * tx + ty * ty + tz * tz) + 1e-6f);
                qx = tx * rn; qy = ty * rn; qz = tz * rn;
            }
            const float sx = a.cam_pos[3 * vv] - px, sy = a.cam_pos[3 * vv + 1] - py, sz = a.cam_pos[3 * vv + 2] - pz;
            const float rsn = crcp(sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
            const float ux = sx * rsn, uy = sy * rsn, uz = sz * rsn;
            const float dx = qx - ux, dy = qy - uy, dz = qz - uz;
            const float rdn = crcp(fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f));
            rd[0] = dx * rdn; rd[1] = dy * rdn; rd[2] = dz * rdn;
            rd[3] = qx * ux + qy * uy + qz * uz;
        }
        // ---- ray_dir_fc: 4 -> 16 -> 59, added to the sampled features -----------------------------------------------------------
        {
            f32x16 acc1[1];
            cm_bias<1>(acc1, lds + TAIL + CM_B_RD0, h);
            const float b0[2] = {h ? rd[1] : rd[0], h ? rd[3] : rd[2]};
            cm_layer<X3, 1, 2>(acc1, lds, lane, CM_A_RD0, CX_A_RD0, b0, m1);
            float d16[8];
#pragma unroll
            for (int r = 0; r < 8; r += 2) { const f32x2 e2 = celu2(acc1[0][r], acc1[0][r + 1]); d16[r] = e2[0]; d16[r + 1] = e2[1]; }
            f32x16 acc2[2];
            cm_bias<2>(acc2, lds + TAIL + CM_B_RD1, h);
            cm_layer<X3, 2, 8>(acc2, lds, lane, CM_A_RD1, CX_A_RD1, d16, m1);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(acc2[b][r], acc2[b][r + 1]); rf[16 * b + r] += e2[0]; rf[16 * b + r + 1] += e2[1]; }
        }
        // ---- pooling weights over views ----------------------------------------------------------------------------------------------
        const float e = __builtin_amdgcn_exp2f(s_abs * (rd[3] - 1.f));       // s_abs carries log2(e)
        const float emin = gmin<G>(view_ok ? e : INFINITY);
        float wgt = (e - emin) * m;
        wgt = wgt * crcp(gsum<G>(wgt) + 1e-8f);
        // ---- view-independent rows: this lane's OPV outputs over its half's channels -------------------------------------------------
        {
            const float* WS = lds + TAIL + CM_W_S + v * OPV;
            const float* WM = WS + (16 + 32 * h) * 64;
            const float* WV = WS + (80 + 32 * h) * 64;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float mean = gsum<G>(rf[c] * wgt);
                const float dd = rf[c] - mean;
                const float var = gsum<G>(wgt * dd * dd);
#pragma unroll
                for (int o = 0; o < OPV; ++o) sacc[o] = fmaf(var, WV[c * 64 + o], fmaf(mean, WM[c * 64 + o], sacc[o]));
            }
            float* sb = sbuf + ptl * PST + h * 64 + v * OPV;
#pragma unroll
            for (int o = 0; o < OPV; ++o) sb[o] = sacc[o];
            __builtin_amdgcn_wave_barrier();
        }
        // ---- base_fc: (shared + 59 per-view features) -> 64 -> 32 -------------------------------------------------------------------
        f32x16 x32[1];
        {
            f32x16 acc[2];
            cm_bias<2>(acc, lds + TAIL + CM_B_B0, h);
            const float* s0 = sbuf + ptl * PST;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nidx = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
                    acc[b][r] += s0[nidx] + s0[64 + nidx];
                }
            __builtin_amdgcn_wave_barrier();
            cm_layer<X3, 2, 32>(acc, lds, lane, CM_A_B0, CX_A_B0, rf, m1);
            float hb[32];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(acc[b][r], acc[b][r + 1]); hb[16 * b + r] = e2[0]; hb[16 * b + r + 1] = e2[1]; }
            cm_bias<1>(x32, lds + TAIL + CM_B_B1, h);
            cm_layer<X3, 1, 32>(x32, lds, lane, CM_A_B1, CX_A_B1, hb, m1);
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(x32[0][r], x32[0][r + 1]); x32[0][r] = e2[0]; x32[0][r + 1] = e2[1]; }
        }
        // ---- vis_fc --------------------------------------------------------------------------------------------------------------------------
        float vis;
        {
            float bin[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bin[r] = x32[0][r] * wgt;
            f32x16 t1[1];
            cm_bias<1>(t1, lds + TAIL + CM_B_V0, h);
            cm_layer<X3, 1, 16>(t1, lds, lane, CM_A_V0, CX_A_V0, bin, m1);
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(t1[0][r], t1[0][r + 1]); bin[r] = e2[0]; bin[r + 1] = e2[1]; }
            f32x16 t2[1];
            cm_bias<1>(t2, lds + TAIL + CM_B_V1, h);
            cm_layer<X3, 1, 16>(t2, lds, lane, CM_A_V1, CX_A_V1, bin, m1);
            float vr = 0.f;                                           // output 32 of vis_fc.2: dot product over both halves
#pragma unroll
            for (int r = 0; r < 16; ++r) vr = fmaf(bin[r], lds[TAIL + CM_V_V1X + h * 16 + r], vr);
            vr += __shfl_xor(vr, 32);
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(t2[0][r], t2[0][r + 1]); x32[0][r] += e2[0]; x32[0][r + 1] += e2[1]; }
            vis = csigm(celu(vr + lds[TAIL + CM_S + 1])) * m;
        }
        // ---- vis_fc2 ------------------------------------------------------------------------------------------------------------------------
        {
            float bin[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bin[r] = x32[0][r] * vis;
            f32x16 t1[1];
            cm_bias<1>(t1, lds + TAIL + CM_B_V20, h);
            cm_layer<X3, 1, 16>(t1, lds, lane, CM_A_V20, CX_A_V20, bin, m1);
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(t1[0][r], t1[0][r + 1]); bin[r] = e2[0]; bin[r + 1] = e2[1]; }
            float vr = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) vr = fmaf(bin[r], lds[TAIL + CM_V_V21 + h * 16 + r], vr);
            vr += __shfl_xor(vr, 32);
            vis = csigm(vr + lds[TAIL + CM_S + 2]) * m;
        }
        // ---- rgb_fc: [x | vis | ray_diff] (37) -> 16 -> 8 -> 1 ----------------------------------------------------------------------------
        float score;
        {
            float bin[19];
#pragma unroll
            for (int r = 0; r < 16; ++r) bin[r] = x32[0][r];
            bin[16] = h ? rd[0] : vis; bin[17] = h ? rd[2] : rd[1]; bin[18] = h ? 0.f : rd[3];
            f32x16 t1[1];
            cm_bias<1>(t1, lds + TAIL + CM_B_R0, h);
            cm_layer<X3, 1, 19>(t1, lds, lane, CM_A_R0, CX_A_R0, bin, m1);
            float r16[8];
#pragma unroll
            for (int r = 0; r < 8; r += 2) { const f32x2 e2 = celu2(t1[0][r], t1[0][r + 1]); r16[r] = e2[0]; r16[r + 1] = e2[1]; }
            f32x16 t2[1];
            cm_bias<1>(t2, lds + TAIL + CM_B_R1, h);
            cm_layer<X3, 1, 8>(t2, lds, lane, CM_A_R1, CX_A_R1, r16, m1);
            float r8[4];
#pragma unroll
            for (int r = 0; r < 4; r += 2) { const f32x2 e2 = celu2(t2[0][r], t2[0][r + 1]); r8[r] = e2[0]; r8[r + 1] = e2[1]; }
            float sr = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) sr = fmaf(r8[r], lds[TAIL + CM_V_R2 + h * 16 + r], sr);
            score = sr + __shfl_xor(sr, 32) + lds[TAIL + CM_S + 3];
        }
        // ---- masked softmax over views, blended colour ----------------------------------------------------------------------------------
        if (m == 0.f) score = -1e9f;
        if (!view_ok) score = -INFINITY;
        const float smax = gmax<G>(score);
        const float ex = view_ok ? __builtin_amdgcn_exp2f(score - smax) : 0.f;      // scores are in the scaled domain
        const float bw = ex * crcp(gsum<G>(ex));
        const float bwn = bw * LN2;                                 // undo the scale of the colours
        const float c0 = gsum<G>(rgb0 * bwn), c1 = gsum<G>(rgb1 * bwn), c2 = gsum<G>(rgb2 * bwn);
        const float nv = gsum<G>(m);
        if (live && v == 0 && h == 0) {
            a.out_rgb[3 * slot] = c0; a.out_rgb[3 * slot + 1] = c1; a.out_rgb[3 * slot + 2] = c2;
            if (a.out_nviews) a.out_nviews[slot] = (uint8_t)(nv + 0.5f);
        }
    }
}

#endif  // O2345_TILES_KERNEL

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_color_mfma_blob_floats(void) { return CM_TOTAL2; }
int o2345_color_x3_blob_floats(void) { return CX_TOTAL2; }

static int color_mfma_launch(bool x3, const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                             const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                             const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                             const float* normals, float* out_rgb, uint8_t* out_nviews, unsigned long long* stats_dev, void* stream) {
    O2345_REQUIRE(blob && vol_cl && maskvol && cmaps && proj && cam_pos && pts && out_rgb, "color_points: null pointer");
    O2345_REQUIRE((query_cam != nullptr) != (normals != nullptr), "color_points: give exactly one of query_cam / normals");
    O2345_REQUIRE(V >= 1 && V <= 255, "color_points: V must be in [1,255] (valid-view counts are stored as uint8; got %d)", V);
    if (n <= 0 && !n_dev) return 0;
#ifdef O2345_TILES_KERNEL
    // test-only variant: O2345_COLOR_KERNEL=tiles selects k_color_mfma (columns = (point, view) pairs, view count padded to a power of two <= 32, every
    // pair evaluated) -- the A/B partner of k_color_pts in tests/test_gpu_parity.py::test_color_points and tools/ab_color.py
    if (knobs().color_tiles && V <= 32) {
        ColorMArgs a{blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews};
        a.sched = color_sched_mode();
        int G = 4;
        while (G < V) G <<= 1;
        const int n_cu = cu_count();
        const int threads = 768, ppt = 32 / G;
        const long long per_block = (long long)(threads / 64) * ppt;
        long long want = n_dev ? n_cu : (n + per_block - 1) / per_block;
        const unsigned grid = persistent_grid(want, n_cu);
        const size_t lds = (size_t)((x3 ? CX_TOTAL : CM_TOTAL) + (threads / 64) * ppt * (2 * 64 + 4)) * sizeof(float);
        hipStream_t s = (hipStream_t)stream;
#define O2345_CM_CASE(GG, XX)                                                                                              \
    if (G == GG && x3 == XX) {                                                                                             \
        O2345_ENSURE_LDS((k_color_mfma<GG, XX>), lds);                                                                     \
        hipLaunchKernelGGL((k_color_mfma<GG, XX>), dim3(grid), dim3(threads), lds, s, a);                                  \
    }
        O2345_CM_CASE(4, false) O2345_CM_CASE(8, false) O2345_CM_CASE(16, false) O2345_CM_CASE(32, false)
        O2345_CM_CASE(4, true) O2345_CM_CASE(8, true) O2345_CM_CASE(16, true) O2345_CM_CASE(32, true)
#undef O2345_CM_CASE
        return check_launch("color_points (tiles kernel)");
    }
#endif
    return color_pts_launch(x3 ? 1 : 0, blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals,
                            out_rgb, out_nviews, stats_dev, stream);
}

int o2345_color_points_mfma(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                            const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                            const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                            const float* normals, float* out_rgb, uint8_t* out_nviews, unsigned long long* stats_dev, void* stream) {
    return color_mfma_launch(false, blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews, stats_dev, stream);
}

// Projector.compute (query_cam) / compute_view_independent (normals) materialised: geometry_feat [P,16], rgb_feat [V,P,59], ray_diff [V,P,4],
// mask [V,P] (1 / 0) in the reference's view-major layout (models/projector.py:96-425)
int o2345_project_features(const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj, const float* cam_pos, int V, int H,
                           int W, const float* pts, long long P, const float* query_cam, const float* normals, float* geometry_feat, float* rgb_feat,
                           float* ray_diff, float* mask, void* stream) {
    O2345_REQUIRE(vol_cl && maskvol && cmaps && proj && cam_pos && pts && geometry_feat && rgb_feat && ray_diff && mask, "project_features: null pointer");
    O2345_REQUIRE((query_cam != nullptr) != (normals != nullptr), "project_features: give exactly one of query_cam / normals");
    O2345_REQUIRE(V >= 1 && V <= 255 && P >= 0 && P * V < (1ll << 33), "project_features: bad sizes (V in [1,255])");
    if (P == 0) return 0;
    return project_features_launch(vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, P, query_cam, normals, geometry_feat, rgb_feat, ray_diff, mask, stream);
}

// GeneralRenderingNetwork.forward(geometry_feat, rgb_feat, ray_diff, mask) on materialised tensors in the reference's layout (view-major):
// geometry_feat [P,16], rgb_feat [V,P,59], ray_diff [V,P,4], mask [V,P] (non-zero = valid) -> rgb [P,3], number of valid views [P].
// x3 = 1: blob from weights.pack_color_x3_blob (split-f16 form), 0: weights.pack_color_mfma_blob (fp32 MFMA).
int o2345_color_from_features(const float* blob, int x3, const float* geometry_feat, const float* rgb_feat, const float* ray_diff, const float* mask,
                              int V, long long P, float* out_rgb, uint8_t* out_nviews, void* stream) {
    O2345_REQUIRE(blob && geometry_feat && rgb_feat && ray_diff && mask && out_rgb, "color_from_features: null pointer");
    O2345_REQUIRE(V >= 1 && V <= 255 && P >= 0, "color_from_features: bad sizes (V in [1,255]: valid-view counts are stored as uint8)");
    if (P == 0) return 0;
    return color_feats_launch(x3, blob, geometry_feat, rgb_feat, ray_diff, mask, V, P, out_rgb, out_nviews, stream);
}

// split-f16 form (blob from weights.pack_color_x3_blob, o2345_color_x3_blob_floats() floats)
int o2345_color_points_x3(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                          const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                          const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                          const float* normals, float* out_rgb, uint8_t* out_nviews, unsigned long long* stats_dev, void* stream) {
    return color_mfma_launch(true, blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews, stats_dev, stream);
}

}  // extern "C"
