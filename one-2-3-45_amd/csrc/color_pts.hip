// Image-based colour blending, "points as columns" form: the same function as csrc/color_mfma.hip
// (Projector.compute / compute_view_independent + GeneralRenderingNetwork.forward, models/projector.py:96-425,
// models/rendering_network.py:75-129) with a different work decomposition.
//
// csrc/color_mfma.hip makes a (point, view) pair a matrix column, so everything that couples the views of a point -- pooling
// weights, weighted mean / variance of 64 pixel floats, the view-independent rows of base_fc.0, the final softmax -- crosses
// lanes: 236 DPP adds (6.3 cycles each on gfx950), 320 packed FMAs on 128 KB of LDS weight rows and an LDS exchange per 4 points.
// On this chip a SIMD issues EITHER vector OR matrix work (profiles/r02_ubench_issue_model.md), so that vector work is not hidden.
//
// Here a wave owns 32 POINTS (column j = lane & 31, the two halves h = lane >> 5 own 32 of the 64 pixel floats each, as before)
// and walks the V source views in three passes, keeping every per-point quantity in registers of the point's two lanes:
//   pass 0   geometry feature (8 channels per half), validity, query direction, min over views of the pooling exponent;
//   pass A   per view: bilinear gather of the half pixel, ray_dir_fc (4 -> 16 -> 59, matrix cores), weighted running mean and
//            M2 of the 32 pixel floats (Welford update with the un-normalised pooling weights: no cross-lane traffic, no cancellation);
//   shared   view-independent rows of base_fc.0 (geo | mean | var -> 64) ONCE per point on the matrix cores (N = 32 points): the
//            result (incl. bias) is the accumulator input of every view's base_fc.0;
//   pass B   per view: gather + ray_dir_fc again (recomputed: 59 floats x V per point do not fit anywhere), base_fc, vis_fc,
//            vis_fc2, rgb_fc exactly as in color_mfma.hip, and an online softmax over the views (running max / sum / rgb).
// View-uniform data (projection rows, camera centres) is read through scalar loads.  V is a run-time loop bound: no power-of-two
// padding of the view count, any V >= 1.  LDS holds the operand blobs only (<= 100 KB).
#include "color_net.h"

namespace o2345 {

// this half's 32 pixel floats of view v at (g.gx, g.gy), bilinear, ATen zero padding, in the log2(e)-scaled domain.
// (Measured alternatives, all slower on MI355X: branch-free taps 50-59 ms; taps of view v + 1 requested during the network of view v
// -- one tap, 32 registers, at a time -- 48.4 ms; lane octets fetching whole 128-byte half pixels through global_load_lds into a per-wave
// LDS staging area (8 lines per instruction instead of up to 64) 48.9 ms; this form 44.5-45.2 ms.  See DESIGN.md section 8.)
__device__ __forceinline__ void gather_now(const ColorMArgs& a, int h, int v, const ViewGeom& g, float (&rf)[32]) {
#pragma unroll
    for (int c = 0; c < 32; ++c) rf[c] = 0.f;
    const Taps2D tp = bilinear_taps(g.gx, g.gy, a.H, a.W_img);
    const float4* img = reinterpret_cast<const float4*>(a.cmaps + (size_t)v * a.H * a.W_img * 64) + 8 * h;
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

// FEATS form (GeneralRenderingNetwork.forward on materialised tensors, rendering_network.py:75-129): the same per-view quantities read from the
// reference's view-major inputs instead of being derived from the point
__device__ __forceinline__ ViewGeom feat_geom(const ColorMArgs& a, int v, long long slot, float s_abs) {
    ViewGeom g;
    const float4 rd = *reinterpret_cast<const float4*>(a.f_rdiff + ((size_t)v * a.n + slot) * 4);
    g.rd[0] = rd.x; g.rd[1] = rd.y; g.rd[2] = rd.z; g.rd[3] = rd.w;
    g.e = __builtin_amdgcn_exp2f(s_abs * (rd.w - 1.f));
    g.m = a.f_mask[(size_t)v * a.n + slot] != 0.f ? 1.f : 0.f;
    g.gx = g.gy = 0.f;
    return g;
}
__device__ __forceinline__ void load_feats(const ColorMArgs& a, int h, int v, long long slot, float (&rf)[32]) {
    const float* src = a.f_rgb + ((size_t)v * a.n + slot) * 59 + 32 * h;
#pragma unroll
    for (int c = 0; c < 32; ++c) rf[c] = (h == 0 || c < 27) ? src[(h == 0 || c < 27) ? c : 0] * LOG2E : 0.f;
}

// ray_dir_fc (4 -> 16 -> 59) of view geometry g in two steps: layer 1 -> d16, layer 2 added to this half's 32 gathered pixel floats
template <bool X3>
__device__ __forceinline__ void direction_layer1(const float* lds, int tail, int lane, int h, const ViewGeom& g, float m1, float (&d16)[8]) {
    f32x16 acc1[1];
    cm_bias<1>(acc1, lds + tail + CM_B_RD0, h);
    const float b0[2] = {h ? g.rd[1] : g.rd[0], h ? g.rd[3] : g.rd[2]};
    cm_layer<X3, 1, 2>(acc1, lds, lane, CM_A_RD0, CX_A_RD0, b0, m1);
#pragma unroll
    for (int r = 0; r < 8; r += 2) { const f32x2 e2 = celu2(acc1[0][r], acc1[0][r + 1]); d16[r] = e2[0]; d16[r + 1] = e2[1]; }
}
template <bool X3>
__device__ __forceinline__ void direction_layer2(const float* lds, int tail, int lane, int h, const float (&d16)[8], float m1, float (&rf)[32]) {
    f32x16 acc2[2];
    cm_bias<2>(acc2, lds + tail + CM_B_RD1, h);
    cm_layer<X3, 2, 8>(acc2, lds, lane, CM_A_RD1, CX_A_RD1, d16, m1);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const f32x2 e2 = celu2(acc2[b][r], acc2[b][r + 1]); rf[16 * b + r] += e2[0]; rf[16 * b + r + 1] += e2[1]; }
}

// 512-thread workgroups (2 waves per SIMD, <= 256 VGPRs: no spills; measured 45.2 ms vs 46.0 ms with 768 threads / 168 VGPRs / 140 B of spills)
constexpr int CP_THREADS = 512;
template <bool X3, bool FEATS>
__global__ __launch_bounds__(CP_THREADS) void k_color_pts(ColorMArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // staged: [A segments | biases and per-lane vectors] [scalars (4)] [A_S]  -- the VALU weight rows W_S of color_mfma.hip are skipped
    constexpr int HEAD = X3 ? (CX_A_END + CM_W_S - CM_BIAS0) : CM_W_S;       // floats before W_S in the blob
    constexpr int SRC_S = X3 ? (CX_TOTAL - 4) : CM_S;                       // scalars in the blob
    constexpr int NAS = X3 ? 2 * 9 * 512 : 2 * 72 * 64;
    constexpr int TAIL = X3 ? CX_A_END - CM_BIAS0 : 0;                      // shift of the bias block, as in color_mfma.hip
    constexpr int L_S = HEAD, L_AS = HEAD + 4;                              // LDS offsets of the scalars and of A_S
    for (int i = threadIdx.x * 4; i < HEAD; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(a.blob + i);
    for (int i = threadIdx.x * 4; i < 4 + NAS; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(lds + L_S + i) = *reinterpret_cast<const float4*>(a.blob + SRC_S + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const long long n = a.n_dev ? (long long)*a.n_dev : a.n;
    const float m1 = X3 ? opaque_minus_one() : -1.f;
    const float s_abs = fabsf(lds[L_S]) * LOG2E;
    const int V = a.V;
    const bool skip_views = V <= 64 && !(a.sched & 4);              // bit 2 of O2345_COLOR_SCHED: evaluate every view (A/B runs)
    const int base_prio = (a.sched & 1) ? (wave >> 2) : 0;          // waves w and w + 4 share a SIMD (cyclic SIMD assignment)
    if (a.sched & 1) set_wave_prio(base_prio);
    // bit 3 of O2345_COLOR_SCHED: block-interleaved tiles instead of one contiguous eighth of the list per XCD (A/B: with view skipping the
    // cost of a tile depends on where its rays look, and a contiguous eighth of the image is not an eighth of the work)
    const TileSched ts = (a.sched & 8) ? TileSched{(long long)blockIdx.x * nwave + wave, (n + 31) / 32, (long long)gridDim.x * nwave}
                                       : tile_schedule(n, 32, wave, nwave);
    unsigned st_a = 0, st_b = 0, st_t = 0, st_full = 0;          // wave-uniform work counters (a.stats)
    for (long long tile = ts.first; tile < ts.end; tile += ts.stride) {
        const long long i = tile * 32 + j;
        const bool live = i < n;
        const long long slot = live ? (a.index ? (long long)a.index[i] : i) : 0;
        const float px = (live && !FEATS) ? a.pts[3 * slot] : 0.f, py = (live && !FEATS) ? a.pts[3 * slot + 1] : 0.f,
                    pz = (live && !FEATS) ? a.pts[3 * slot + 2] : 0.f;
        // ---- pass 0: geometry feature (this half's 8 channels), validity, query direction ---------------------------------------
        float bs[72];                               // per-half operands of the shared rows: geo (8) | mean (32) | var (32)
        bool gvalid;
        if constexpr (FEATS) {
            const float4* g4 = reinterpret_cast<const float4*>(a.f_geo + (size_t)slot * 16) + 2 * h;
            const float4 t0 = g4[0], t1 = g4[1];
            bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
            gvalid = true;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) bs[c] = 0.f;
            float msum = 0.f;
            const Axis2 ax = axis_taps_zeros(px, a.D), ay = axis_taps_zeros(py, a.D), az = axis_taps_zeros(pz, a.D);
#pragma unroll
            for (int tap = 0; tap < 8; ++tap) {
                const int ia = (tap >> 2) & 1, ib = (tap >> 1) & 1, ic = tap & 1;
                const float w = ax.w[ia] * ay.w[ib] * az.w[ic];
                if (w != 0.f) {
                    const size_t vox = ((size_t)ax.i[ia] * a.D + ay.i[ib]) * a.D + az.i[ic];
                    msum += w * a.maskvol[vox];
                    const float4* p4 = reinterpret_cast<const float4*>(a.vol_cl + vox * 16) + 2 * h;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4 t = p4[q];
                        bs[4 * q] = fmaf(t.x, w, bs[4 * q]); bs[4 * q + 1] = fmaf(t.y, w, bs[4 * q + 1]);
                        bs[4 * q + 2] = fmaf(t.z, w, bs[4 * q + 2]); bs[4 * q + 3] = fmaf(t.w, w, bs[4 * q + 3]);
                    }
                }
            }
            gvalid = fabsf(px) < 1.f && fabsf(py) < 1.f && fabsf(pz) < 1.f && msum > 0.f;
        }
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if constexpr (FEATS) {
        } else if (a.normals) {
            const float nx = a.normals[3 * slot], ny = a.normals[3 * slot + 1], nz = a.normals[3 * slot + 2];
            const float rn = crcp(fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f));
            qx = nx * rn; qy = ny * rn; qz = nz * rn;
        } else {
            const float tx = a.query_cam[0] - px, ty = a.query_cam[1] - py, tz = a.query_cam[2] - pz;
            const float rn = crcp(sqrtf(tx * tx + ty * ty + tz * tz) + 1e-6f);
            qx = tx * rn; qy = ty * rn; qz = tz * rn;
        }
        // min over ALL views of the pooling exponent (rendering_network.py:94: exp(...).min over the view axis, mask or not)
        float emin = INFINITY;
        for (int v = 0; v < V; ++v) {
            if constexpr (FEATS) {
                emin = fminf(emin, __builtin_amdgcn_exp2f(s_abs * (a.f_rdiff[((size_t)v * a.n + slot) * 4 + 3] - 1.f)));
                continue;
            }
            const float sx = a.cam_pos[3 * v] - px, sy = a.cam_pos[3 * v + 1] - py, sz = a.cam_pos[3 * v + 2] - pz;
            const float rsn = crcp(sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
            const float dot = qx * (sx * rsn) + qy * (sy * rsn) + qz * (sz * rsn);
            emin = fminf(emin, __builtin_amdgcn_exp2f(s_abs * (dot - 1.f)));
        }
        // ---- pass A: weighted mean / variance over the views of this half's 32 pixel floats ----------------------------------------
        // Welford update with the un-normalised weights raw_v = (e_v - emin) m_v: mean_w = sum(raw x)/sum(raw), M2 = sum raw (x - mean_w)^2
        float wsum = 0.f, nvis = 0.f;
        unsigned long long active = 0ull;            // wave-uniform: views that see at least one point of the tile
        {
            float mean[32], m2[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) { mean[c] = 0.f; m2[c] = 0.f; }
#pragma unroll 1
            for (int v = 0; v < V; ++v) {
                const ViewGeom g = FEATS ? feat_geom(a, v, slot, s_abs) : view_geom(a, v, px, py, pz, qx, qy, qz, gvalid, s_abs);
                // A view that sees NONE of the tile's 32 points (wave-uniform test) contributes exactly nothing to this pass: raw = 0 leaves wsum,
                // nvis, mean and M2 bit-unchanged.  Points of a tile are neighbours (32 adjacent rays at one sample index), so visibility is
                // coherent: at BASELINE config 2 a point is seen by 4.8 of the 8 views on average and 37 % of the (tile, view) pairs are skipped.
                if (skip_views && __builtin_amdgcn_ballot_w64(g.m != 0.f) == 0ull) continue;
                active |= 1ull << (v & 63);
                ++st_a;
                float rf[32];
                if constexpr (FEATS) load_feats(a, h, v, slot, rf);
                else {
                    if (a.sched & 2) set_wave_prio(3);
                    gather_now(a, h, v, g, rf);
                    if (a.sched & 2) set_wave_prio(base_prio);
                }
                {
                    float d16[8];
                    direction_layer1<X3>(lds, TAIL, lane, h, g, m1, d16);
                    direction_layer2<X3>(lds, TAIL, lane, h, d16, m1, rf);
                }
                const float raw = (g.e - emin) * g.m;
                nvis += g.m;
                wsum += raw;
                const float r0 = raw > 0.f ? raw * crcp(wsum) : 0.f;
                // one Newton step on the quotient: raw / wsum to <= 1 ulp
                const float rq = raw > 0.f ? fmaf(fmaf(-wsum, r0, raw), crcp(wsum), r0) : 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    const float d = rf[c] - mean[c];
                    mean[c] = fmaf(rq, d, mean[c]);
                    m2[c] = fmaf(raw * d, rf[c] - mean[c], m2[c]);
                }
            }
            // the reference normalises the weights by (sum + 1e-8): w_v = raw_v / (wsum + 1e-8), S = sum w_v <= 1
            //   mean_ref = sum w x = S mean_w,   var_ref = sum w (x - mean_ref)^2 = M2 / (wsum + 1e-8) + S (1 - S)^2 mean_w^2
            const float rden = crcp(wsum + 1e-8f);
            const float S = wsum * rden, k2 = S * (1.f - S) * (1.f - S);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                bs[8 + c] = S * mean[c];
                bs[40 + c] = fmaf(k2 * mean[c], mean[c], m2[c] * rden);
            }
        }
        const float rden = crcp(wsum + 1e-8f);
        // ---- view-independent rows of base_fc.0, once per point: sh = bias + W_shared [geo | mean | var] -------------------------
        f32x16 sh[2];
        cm_bias<2>(sh, lds + TAIL + CM_B_B0, h);
        if constexpr (X3) cx_run<2, 72>(sh, reinterpret_cast<const float4*>(lds + L_AS) + lane, bs, m1);
        else cm_run<2, 72, 72>(sh, lds + L_AS + lane, 0, bs);
        // ---- pass B: per view network, online softmax over the views ------------------------------------------------------------------
        float smax = -INFINITY, ssum = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
        // A masked view enters the softmax with score -1e9: its blending weight is exp2(-1e9 - max) = 0 EXACTLY as soon as the point has one
        // visible view, whatever the order.  Only a point with NO visible view blends the masked views (uniformly): if the tile holds such a
        // point, every view is evaluated as before; otherwise the views that see none of the tile's points are skipped -- bit-identical results.
        const bool skip_b = skip_views && __builtin_amdgcn_ballot_w64(live && nvis == 0.f) == 0ull;
        ++st_t;
        st_full += skip_b ? 0u : 1u;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            if (skip_b && !((active >> (v & 63)) & 1ull)) continue;
            ++st_b;
            const ViewGeom g = FEATS ? feat_geom(a, v, slot, s_abs) : view_geom(a, v, px, py, pz, qx, qy, qz, gvalid, s_abs);
            const float m = g.m;
            float rf[32];
            if constexpr (FEATS) load_feats(a, h, v, slot, rf);
            else {
                if (a.sched & 2) set_wave_prio(3);
                gather_now(a, h, v, g, rf);
                if (a.sched & 2) set_wave_prio(base_prio);
            }
            const float rgb0 = rf[0], rgb1 = rf[1], rgb2 = rf[2];   // log2(e) * colours (meaningful in half 0), before the direction feature
            {
                float d16[8];
                direction_layer1<X3>(lds, TAIL, lane, h, g, m1, d16);
                direction_layer2<X3>(lds, TAIL, lane, h, d16, m1, rf);
            }
            const float wgt = (g.e - emin) * m * rden;
            // ---- base_fc: (shared + 59 per-view features) -> 64 -> 32
            f32x16 x32[1];
            {
                f32x16 acc[2];
                if constexpr (X3) cx_run_from<2, 32>(acc, sh, reinterpret_cast<const float4*>(lds + CX_A_B0) + lane, rf, m1);
                else { acc[0] = sh[0]; acc[1] = sh[1]; cm_layer<X3, 2, 32>(acc, lds, lane, CM_A_B0, CX_A_B0, rf, m1); }
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
            // ---- vis_fc
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
                vis = csigm(celu(vr + lds[L_S + 1])) * m;
            }
            // ---- vis_fc2
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
                vis = csigm(vr + lds[L_S + 2]) * m;
            }
            // ---- rgb_fc: [x | vis | ray_diff] (37) -> 16 -> 8 -> 1
            float score;
            {
                float bin[19];
#pragma unroll
                for (int r = 0; r < 16; ++r) bin[r] = x32[0][r];
                bin[16] = h ? g.rd[0] : vis; bin[17] = h ? g.rd[2] : g.rd[1]; bin[18] = h ? 0.f : g.rd[3];
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
                score = sr + __shfl_xor(sr, 32) + lds[L_S + 3];
            }
            // ---- masked softmax over the views, running form (scores are in the scaled domain: base 2)
            if (m == 0.f) score = -1e9f;
            const float nmax = fmaxf(smax, score);
            const float sc_old = __builtin_amdgcn_exp2f(smax - nmax), ex = __builtin_amdgcn_exp2f(score - nmax);
            ssum = fmaf(ssum, sc_old, ex);
            o0 = fmaf(o0, sc_old, ex * rgb0); o1 = fmaf(o1, sc_old, ex * rgb1); o2 = fmaf(o2, sc_old, ex * rgb2);
            smax = nmax;
        }
        if (live && h == 0) {
            const float rs = crcp(ssum) * LN2;                       // undo the scale of the colours
            a.out_rgb[3 * slot] = o0 * rs; a.out_rgb[3 * slot + 1] = o1 * rs; a.out_rgb[3 * slot + 2] = o2 * rs;
            if (a.out_nviews) a.out_nviews[slot] = (uint8_t)(nvis + 0.5f);
        }
    }
    if (a.stats && lane == 0) {
        atomicAdd(a.stats + 0, (unsigned long long)st_a); atomicAdd(a.stats + 1, (unsigned long long)st_b);
        atomicAdd(a.stats + 2, (unsigned long long)st_t); atomicAdd(a.stats + 3, (unsigned long long)st_full);
    }
}

}  // namespace o2345

namespace o2345 {

// launcher shared by o2345_color_points_mfma / o2345_color_points_x3 (csrc/color_mfma.hip decides which kernel runs)
int color_pts_launch(int x3, const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj,
                           const float* cam_pos, int V, int H, int W, const float* pts, const int32_t* index, const int32_t* n_dev,
                           long long n, const float* query_cam, const float* normals, float* out_rgb, uint8_t* out_nviews,
                           unsigned long long* stats_dev /* optional, caller-owned work counters [4] */, void* stream) {
    ColorMArgs a{blob, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, index, n_dev, n, query_cam, normals, out_rgb, out_nviews};
    a.sched = color_sched_mode();
    a.stats = stats_dev;
    const int n_cu = cu_count();
    const int threads = CP_THREADS;
    const long long per_block = (long long)(threads / 64) * 32;
    long long want = n_dev ? n_cu : (n + per_block - 1) / per_block;
    const unsigned grid = persistent_grid(want, n_cu);
    const size_t lds = (size_t)(x3 ? (CX_A_END + CM_W_S - CM_BIAS0) + 4 + 2 * 9 * 512 : CM_W_S + 4 + 2 * 72 * 64) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (x3) {
        O2345_ENSURE_LDS((k_color_pts<true, false>), lds);
        hipLaunchKernelGGL((k_color_pts<true, false>), dim3(grid), dim3(threads), lds, s, a);
    } else {
        O2345_ENSURE_LDS((k_color_pts<false, false>), lds);
        hipLaunchKernelGGL((k_color_pts<false, false>), dim3(grid), dim3(threads), lds, s, a);
    }
    return check_launch("color_points (points-as-columns kernel)");
}

// Projector.compute / compute_view_independent MATERIALISED (models/projector.py:96-425): the four tensors the reference's own
// GeneralRenderingNetwork.forward takes, in its layout.  One wave per (point, view), lane = channel of the 64-float pixel.  Only for callers that
// want the tensors (a foreign rendering network); the fused kernels above never store them.
__global__ __launch_bounds__(256) void k_project_features(ColorMArgs a, float* __restrict__ geo /*[P,16]*/, float* __restrict__ rgb_feat /*[V,P,59]*/,
                                                          float* __restrict__ rdiff /*[V,P,4]*/, float* __restrict__ mask /*[V,P]*/) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.n * a.V) return;
    const long long p = w / a.V;
    const int v = (int)(w - p * a.V), c = threadIdx.x & 63;
    const float px = a.pts[3 * p], py = a.pts[3 * p + 1], pz = a.pts[3 * p + 2];
    float msum = 0.f, gch = 0.f;
    {
        const Axis2 ax = axis_taps_zeros(px, a.D), ay = axis_taps_zeros(py, a.D), az = axis_taps_zeros(pz, a.D);
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
            const int ia = (tap >> 2) & 1, ib = (tap >> 1) & 1, ic = tap & 1;
            const float wt = ax.w[ia] * ay.w[ib] * az.w[ic];
            if (wt != 0.f) {
                const size_t vox = ((size_t)ax.i[ia] * a.D + ay.i[ib]) * a.D + az.i[ic];
                msum += wt * a.maskvol[vox];
                if (c < 16) gch = fmaf(a.vol_cl[vox * 16 + c], wt, gch);
            }
        }
    }
    const bool gvalid = fabsf(px) < 1.f && fabsf(py) < 1.f && fabsf(pz) < 1.f && msum > 0.f;
    float qx, qy, qz;
    if (a.normals) {
        const float nx = a.normals[3 * p], ny = a.normals[3 * p + 1], nz = a.normals[3 * p + 2];
        const float rn = crcp(fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f));
        qx = nx * rn; qy = ny * rn; qz = nz * rn;
    } else {
        const float tx = a.query_cam[0] - px, ty = a.query_cam[1] - py, tz = a.query_cam[2] - pz;
        const float rn = crcp(sqrtf(tx * tx + ty * ty + tz * tz) + 1e-6f);
        qx = tx * rn; qy = ty * rn; qz = tz * rn;
    }
    const ViewGeom g = view_geom(a, v, px, py, pz, qx, qy, qz, gvalid, 0.f);
    float val = 0.f;
    {
        const Taps2D tp = bilinear_taps(g.gx, g.gy, a.H, a.W_img);
        const float* img = a.cmaps + (size_t)v * a.H * a.W_img * 64 + c;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tp.w[k] != 0.f) val = fmaf(img[(size_t)tp.idx[k] * 64], tp.w[k], val);
    }
    const size_t vp = (size_t)v * a.n + p;
    if (c < 59) rgb_feat[vp * 59 + c] = val;
    if (c < 4) rdiff[vp * 4 + c] = g.rd[c];
    if (c == 0) mask[vp] = g.m;
    if (v == 0 && c < 16) geo[(size_t)p * 16 + c] = gch;
}

int project_features_launch(const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj, const float* cam_pos, int V, int H, int W,
                            const float* pts, long long n, const float* query_cam, const float* normals, float* geo, float* rgb_feat, float* rdiff, float* mask,
                            void* stream) {
    ColorMArgs a{nullptr, vol_cl, maskvol, D, cmaps, proj, cam_pos, V, H, W, pts, nullptr, nullptr, n, query_cam, normals, nullptr, nullptr};
    hipLaunchKernelGGL(k_project_features, dim3(cdiv(n * V, 4)), dim3(256), 0, (hipStream_t)stream, a, geo, rgb_feat, rdiff, mask);
    return check_launch("project_features");
}

// GeneralRenderingNetwork.forward on the reference's materialised tensors (the drop-in form; the fused Projector path above is the fast one)
int color_feats_launch(int x3, const float* blob, const float* geo, const float* rgb_feat, const float* ray_diff, const float* mask, int V, long long n,
                       float* out_rgb, uint8_t* out_nviews, void* stream) {
    ColorMArgs a{};
    a.blob = blob; a.V = V; a.n = n; a.out_rgb = out_rgb; a.out_nviews = out_nviews;
    a.f_geo = geo; a.f_rgb = rgb_feat; a.f_rdiff = ray_diff; a.f_mask = mask;
    const int n_cu = cu_count();
    const long long per_block = (long long)(CP_THREADS / 64) * 32;
    const unsigned grid = persistent_grid((n + per_block - 1) / per_block, n_cu);
    const size_t lds = (size_t)(x3 ? (CX_A_END + CM_W_S - CM_BIAS0) + 4 + 2 * 9 * 512 : CM_W_S + 4 + 2 * 72 * 64) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (x3) {
        O2345_ENSURE_LDS((k_color_pts<true, true>), lds);
        hipLaunchKernelGGL((k_color_pts<true, true>), dim3(grid), dim3(CP_THREADS), lds, s, a);
    } else {
        O2345_ENSURE_LDS((k_color_pts<false, true>), lds);
        hipLaunchKernelGGL((k_color_pts<false, true>), dim3(grid), dim3(CP_THREADS), lds, s, a);
    }
    return check_launch("color_from_features");
}

}  // namespace o2345

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_color_pts() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_project_features));
}
}  // namespace o2345
