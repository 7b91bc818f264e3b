// Per-ray NeuS sampling / compositing math shared by the HIP kernels and the host-side check build.
// All per-ray arrays are SAMPLE-MAJOR: a[s * R + r] (lane = ray -> every access is coalesced across the wave).
// Reference: reconstruction/models/sparse_neus_renderer.py (up_sample :73-115, cat_z_vals :117-151, render_core
// :171-455, render :457-635) and models/render_utils.py (sample_pdf :8-51).
#pragma once
#include "geom_math.h"

namespace o2345 {

// torch.linspace(start, end, steps)[i] for fp32, bit-exact with ATen's CPU kernel (RangeFactories.cpp: symmetric
// evaluation, step*i + start contracted to ONE fused multiply-add by its vectorised path)
O2345_HD float linspace_at(float start, float end, int steps, int i) {
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}

O2345_HD float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct RayGeom {
    const float* rays_o;   // [R,3]
    const float* rays_d;   // [R,3]
    int R;
};

O2345_HD void ray_point(const RayGeom& g, int r, float z, float& x, float& y, float& zz) {
    x = g.rays_o[3 * r + 0] + g.rays_d[3 * r + 0] * z;
    y = g.rays_o[3 * r + 1] + g.rays_d[3 * r + 1] * z;
    zz = g.rays_o[3 * r + 2] + g.rays_d[3 * r + 2] * z;
}

O2345_HD float mask_at(const float* __restrict__ maskvol, int D, float x, float y, float z) {
    const int v = nearest_voxel(x, y, z, D);
    return v < 0 ? 0.f : maskvol[v];
}

// ---- per-ray list accessors ----------------------------------------------------------------------------------------------------------------
// The sampler math below is written ONCE, against an accessor: rows of ONE ray's sorted lists (depth z, SDF, point-inside-mask flag), a scratch
// row for the section weights, and a sink for the new depths.  Two accessors exist: GlobalRay (the sample-major global arrays, a[s * R + r]: the
// host-check build and the definition of the semantics) and LdsRay (csrc/render.hip: the lists of 64 rays staged in LDS, a[s * 64 + lane]: every
// access of the serial per-ray chains costs an LDS round trip instead of a dependent trip to L2 / HBM).
struct GlobalRay {
    RayGeom g; int r; size_t R;
    const float* z_; const float* sdf_; float* w_; float* out_;
    const float* maskvol; int D;
    O2345_HD float z(int s) const { return z_[(size_t)s * R + r]; }
    O2345_HD float sdf(int s) const { return sdf_[(size_t)s * R + r]; }
    O2345_HD float msk(int s, float zs) const {            // occupancy of the sample point (nearest voxel of the mask volume)
        float x, y, w;
        ray_point(g, r, zs, x, y, w);
        return mask_at(maskvol, D, x, y, w);
    }
    O2345_HD void set_w(int s, float v) { w_[(size_t)s * R + r] = v; }
    O2345_HD float w(int s) const { return w_[(size_t)s * R + r]; }
    O2345_HD void out(int t, float v) { out_[(size_t)t * R + r] = v; }
};

// One section [s, s + 1] of up_sample (:89-107): the opacity alpha_s from the two samples' depth / SDF / occupancy and the slope of the PREVIOUS
// section (prev_dot; 0 for the first).  dot_raw_out = this section's slope (the next section's prev_dot).  Independent of the running transmittance.
O2345_HD float upsample_section_alpha(float z0, float s0, float m0, float z1, float s1, float m1, float prev_dot, float inv_s, float& dot_raw_out) {
    const float pm = m0 * m1;
    const float mid = (s0 + s1) * 0.5f;
    const float dot_raw = (s1 - s0) / (z1 - z0 + 1e-5f);
    float dot = fminf(prev_dot, dot_raw);
    dot = fminf(fmaxf(dot, -10.f), 0.f) * pm;
    dot_raw_out = dot_raw;
    const float dist = z1 - z0;
    const float pe = mid - dot * dist * 0.5f, ne = mid + dot * dist * 0.5f;
    const float pc = sigmoidf_(pe * inv_s), nc = sigmoidf_(ne * inv_s);
    return pm * ((pc - nc + 1e-5f) / (pc + 1e-5f));
}

// up_sample + sample_pdf(det=True): from S sorted samples (z, sdf) of one ray produce n_imp new z values (a.out(t, z_new), t < n_imp).
// STREAMING form: both passes touch the lists at statically known, ascending rows, a block of CB rows is requested before any of it is used -- with
// the global-memory accessor every block is ONE round trip with 3 x CB loads in flight (round 3 paid a dependent trip per sample), and no pass indexes
// a list at a data-dependent position.  The arithmetic, operation by operation and in the same order, is the reference's:
//   pass 1 (:84-107)  section weights w_s = alpha_s * T_s + 1e-5 with the running transmittance T, and their sum;
//   pass 2 (render_utils.py:24-50)  cdf_k = cdf_{k-1} + w_{k-1} / sum; for the ascending u_t the search index k only moves forward, so the inverse
//           CDF is a single walk over k that emits every u_t whose interval closes at k (searchsorted(right = True): first k with cdf_k > u_t, or S).
template <class A>
O2345_HD void upsample_core(A& a, int S, float inv_s, int n_imp) {
    constexpr int CB = 8;
    float z0 = a.z(0), s0 = a.sdf(0);
    float m0 = a.msk(0, z0);
    float prev_dot = 0.f, T = 1.f, wsum = 0.f;
    for (int sb = 0; sb + 1 < S; sb += CB) {
        float zb[CB], sv[CB], mb[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int i = sb + 1 + k < S ? sb + 1 + k : S - 1;          // the tail re-reads the last row (never used)
            zb[k] = a.z(i); sv[k] = a.sdf(i);
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) mb[k] = a.msk(sb + 1 + k < S ? sb + 1 + k : S - 1, zb[k]);
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (sb + 1 + k < S) {
                const float z1 = zb[k], s1 = sv[k], m1 = mb[k];
                float dot_raw;
                const float alpha = upsample_section_alpha(z0, s0, m0, z1, s1, m1, prev_dot, inv_s, dot_raw);
                prev_dot = dot_raw;
                const float w = alpha * T + 1e-5f;            // sample_pdf: weights + 1e-5
                T = T * (1.f - alpha + 1e-7f);
                a.set_w(sb + k, w);
                wsum += w;
                z0 = z1; s0 = s1; m0 = m1;
            }
        }
    }
    // inverse CDF.  State after advancing to k: c_lo = cdf[k-1], c_hi = cdf[k] (cdf[0] = 0; k = S: nothing is added).  u_0 > 0 = cdf[0], so every
    // sample is emitted at some k >= 1 with below = k - 1, above = min(k, S - 1).
    int t = 0;
    float c_lo = 0.f, c_hi = 0.f;
    float u = linspace_at(0.5f / (float)n_imp, 1.f - 0.5f / (float)n_imp, n_imp, 0);
    float zlo = a.z(0);                                       // z[k - 1]
    for (int kb = 1; kb <= S && t < n_imp; kb += CB) {
        float wb[CB], zb[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int k = kb + j;
            wb[j] = a.w(k - 1 < S - 1 ? k - 1 : S - 2 >= 0 ? S - 2 : 0);      // w[k - 1], defined for k < S
            zb[j] = a.z(k < S ? k : S - 1);                                    // z[above]
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int k = kb + j;
            if (k <= S) {
                c_lo = c_hi;
                if (k < S) c_hi = c_hi + wb[j] / wsum;
                const float zhi = zb[j];
                while (t < n_imp && (k == S || c_hi > u)) {
                    const float cb = c_lo;
                    const float ca = (k == S) ? cb : c_hi;                     // above == below only when k == S
                    float den = ca - cb;
                    if (den < 1e-5f) den = 1.f;
                    const float tt = (u - cb) / den;
                    a.out(t, zlo + tt * (zhi - zlo));
                    ++t;
                    u = linspace_at(0.5f / (float)n_imp, 1.f - 0.5f / (float)n_imp, n_imp, t < n_imp ? t : n_imp - 1);
                }
                zlo = zhi;
            }
        }
    }
}

// wbuf: scratch [>= S-1][R].  Outputs new_z[t*R + r], t < n_imp.
O2345_HD void upsample_ray(const RayGeom& g, int r, const float* __restrict__ z, const float* __restrict__ sdf, int S,
                           float inv_s, const float* __restrict__ maskvol, int D, float* __restrict__ wbuf,
                           int n_imp, float* __restrict__ new_z) {
    GlobalRay a{g, r, (size_t)g.R, z, sdf, wbuf, new_z, maskvol, D};
    upsample_core(a, S, inv_s, n_imp);
}

// cat_z_vals: merge n_new samples into the sorted list of S samples (capacity S + n_new): a stable merge in which, among equal depths, existing
// samples stay in front of new ones.  One DESCENDING sweep over the existing list at static rows (blocks of CB rows requested together), with a pointer c
// into the SORTED new block: c = #{new j : z_new[j] < z[i]} only moves down as i goes down, so
//   existing sample i moves to row i + c (in place: the target row >= i has been vacated already);
//   every new sample the pointer passes while at i (z_new[c-1] >= z[i]) belongs right behind sample i: row i + c;
//   the new samples left when the sweep ends go to rows 0 .. c - 1.
// S + n_new steps of one compare each (a rank merge -- #new < z[i] by comparing against all n_new -- was measured first: 11 k instructions per ray).
// Accessors: A = the list (z / sdf / tag read, put write); N = the new block (z / sdf / tag read, set write): any order on entry, sorted here by insertion.
template <class A, class N>
O2345_HD void merge_core(A& a, int S, N& nb, int n_new) {
    bool sorted = true;
    for (int j = 0; j + 1 < n_new; ++j) sorted = sorted && !(nb.z(j) > nb.z(j + 1));
    if (!sorted) {                                            // (ascending in practice: the inverse CDF of ascending u)
        for (int i = 1; i < n_new; ++i) {
            const float kz = nb.z(i), ks = nb.sdf(i);
            const unsigned kt = nb.tag(i);
            int b = i - 1;
            while (b >= 0 && nb.z(b) > kz) { nb.set(b + 1, nb.z(b), nb.sdf(b), nb.tag(b)); --b; }
            nb.set(b + 1, kz, ks, kt);
        }
    }
    constexpr int CB = 8;
    int c = n_new;
    float zc = n_new ? nb.z(n_new - 1) : 0.f;                 // z_new[c - 1]
    for (int ib = S - 1; ib >= 0 && c > 0; ib -= CB) {        // c == 0: the rest of the list stays where it is
        float zb[CB], sb[CB];
        unsigned tb[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int i = ib - k >= 0 ? ib - k : 0;
            zb[k] = a.z(i); sb[k] = a.sdf(i); tb[k] = a.tag(i);
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int i = ib - k;
            if (i >= 0 && c > 0) {
                const float zi = zb[k];
                while (c > 0 && !(zc < zi)) {                 // new sample c - 1 is not in front of sample i: it goes right behind it
                    a.put(i + c, zc, nb.sdf(c - 1), nb.tag(c - 1));
                    --c;
                    if (c > 0) zc = nb.z(c - 1);
                }
                if (c > 0) a.put(i + c, zi, sb[k], tb[k]);
            }
        }
    }
    for (int j = c - 1; j >= 0; --j) a.put(j, nb.z(j), nb.sdf(j), nb.tag(j));      // new samples in front of the whole list
}

// the new block as plain arrays (host-check build, generic device fallback)
struct ArrayBlock {
    float* z_; float* s_; unsigned* t_;
    O2345_HD float z(int j) const { return z_[j]; }
    O2345_HD float sdf(int j) const { return s_[j]; }
    O2345_HD unsigned tag(int j) const { return t_[j]; }
    O2345_HD void set(int j, float zv, float sv, unsigned tv) { z_[j] = zv; s_[j] = sv; t_[j] = tv; }
};

// The same merge as a RANK merge for a new block of EXACTLY N samples held in registers (device path, N = n_importance / 4 = 16): existing sample i moves to
// i + #{new j : z_new[j] < z[i]}, new sample j to j + #{i : z[i] <= z_new[j]} -- 2 N compares per existing sample, but branch-free, every index into the block
// a compile-time constant (no scratch, no LDS), the block sorted by an odd-even transposition network (skipped when already ascending).  On the MI355X this
// beats the pointer walk above inside the streaming kernel (262,144 rays: 365 vs 426 us per merge + up-sample round; the walk's data-dependent inner loop
// diverges and its LDS-resident block halves the occupancy), so the walk serves the host-check build and odd block sizes, the rank form the product path.
template <class A, int N>
O2345_HD void merge_core_fixed(A& a, int S, float (&nz)[N], float (&ns)[N], unsigned (&nt)[N], bool block_is_sorted) {
    if (!block_is_sorted) {
#pragma unroll
        for (int round = 0; round < N; ++round) {
#pragma unroll
            for (int i = round & 1; i + 1 < N; i += 2) {
                const bool sw = nz[i] > nz[i + 1];                     // strict: equal depths keep their order
                const float z0 = sw ? nz[i + 1] : nz[i], z1 = sw ? nz[i] : nz[i + 1];
                const float s0 = sw ? ns[i + 1] : ns[i], s1 = sw ? ns[i] : ns[i + 1];
                const unsigned t0 = sw ? nt[i + 1] : nt[i], t1 = sw ? nt[i] : nt[i + 1];
                nz[i] = z0; nz[i + 1] = z1; ns[i] = s0; ns[i + 1] = s1; nt[i] = t0; nt[i + 1] = t1;
            }
        }
    }
    int cnt[N];
#pragma unroll
    for (int j = 0; j < N; ++j) cnt[j] = 0;
    constexpr int CB = 8;                       // rows are requested CB at a time, descending (writes of a block go to rows >= its lowest row: never to a row not yet read)
    for (int ib = S - 1; ib >= 0; ib -= CB) {
        float zb[CB], sb[CB];
        unsigned tb[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int i = ib - k >= 0 ? ib - k : 0;
            zb[k] = a.z(i); sb[k] = a.sdf(i); tb[k] = a.tag(i);
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int i = ib - k;
            if (i >= 0) {
                const float zi = zb[k];
                int c = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) { c += nz[j] < zi ? 1 : 0; cnt[j] += zi <= nz[j] ? 1 : 0; }
                if (c) a.put(i + c, zi, sb[k], tb[k]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) a.put(j + cnt[j], nz[j], ns[j], nt[j]);
}

// cat_z_vals on the global sample-major arrays: merge n_new samples (new_z/new_sdf [n_new][R]) into the sorted list (z/sdf [S][R]) in place
// (capacity S + n_new).  Equal keys keep existing samples first.  (The new block is also left sorted in new_z / new_sdf, as before.)
struct GlobalMerge {
    float* z_; float* sdf_; size_t R; int r;
    O2345_HD float z(int i) const { return z_[(size_t)i * R + r]; }
    O2345_HD float sdf(int i) const { return sdf_[(size_t)i * R + r]; }
    O2345_HD unsigned tag(int) const { return 0u; }
    O2345_HD void put(int i, float zv, float sv, unsigned) { z_[(size_t)i * R + r] = zv; sdf_[(size_t)i * R + r] = sv; }
};
O2345_HD void merge_ray(int r, int R, float* __restrict__ z, float* __restrict__ sdf, int S, float* __restrict__ new_z,
                        float* __restrict__ new_sdf, int n_new) {
    constexpr int NMAX = 64;
    float nz[NMAX], ns[NMAX];
    unsigned nt[NMAX];
    for (int base = 0; base < n_new; base += NMAX) {                 // blocks of at most NMAX new samples (the renderer uses 16)
        const int nb = n_new - base < NMAX ? n_new - base : NMAX;
        for (int j = 0; j < nb; ++j) { nz[j] = new_z[(size_t)(base + j) * R + r]; ns[j] = new_sdf[(size_t)(base + j) * R + r]; nt[j] = 0u; }
        GlobalMerge a{z, sdf, (size_t)R, r};
        ArrayBlock blk{nz, ns, nt};
        merge_core(a, S + base, blk, nb);
        for (int j = 0; j < nb; ++j) { new_z[(size_t)(base + j) * R + r] = nz[j]; new_sdf[(size_t)(base + j) * R + r] = ns[j]; }
    }
}

// render_core compositing for one ray (sparse_neus_renderer.py:340-429), general rendering, alpha_type 'div'.
struct CompositeOut {
    float* color;         // [R,3]
    float* depth;         // [R]
    float* weights;       // [S][R]
    float* cdf;           // [S][R]  (prev_cdf)
    float* weights_sum;   // [R]
    float* weights_max;   // [R]
    float* depth_var;     // [R]
    float* alpha_sum;     // [R]
    float* grad_err;      // [R,2]  (sum pm*(|g|-1)^2, sum pm)
    uint8_t* color_mask;  // [R]
};

// One sample of render_core's compositing (:340-372): opacity alpha and the sigmoid pc (the returned "cdf") from the sample's SDF, gradient, section
// length and occupancy; independent of the running transmittance.
O2345_HD float composite_sample_alpha(float dx, float dy, float dz, float gx, float gy, float gz, float m, float dist, float sv, float inv_s,
                                      float alpha_inter_ratio, float& pc_out) {
    const float tdot = dx * gx + dy * gy + dz * gz;
    float icos = -(fmaxf(-tdot * 0.5f + 0.5f, 0.f) * (1.f - alpha_inter_ratio) + fmaxf(-tdot, 0.f) * alpha_inter_ratio);
    icos = icos * m;
    const float half = fminf(fmaxf(icos, -10.f), 10.f) * dist * 0.5f;
    const float pc = sigmoidf_((sv - half) * inv_s), nc = sigmoidf_((sv + half) * inv_s);
    float alpha = (pc - nc + 1e-5f) / (pc + 1e-5f);
    pc_out = pc;
    return fminf(fmaxf(alpha, 0.f), 1.f) * m;
}

O2345_HD void composite_ray(const RayGeom& g, int r, int S, const float* __restrict__ mid_z,
                            const float* __restrict__ dists, const float* __restrict__ pm, const float* __restrict__ sdf,
                            const float* __restrict__ grad /*[S*R,3]*/, const float* __restrict__ rgb /*[S*R,3]*/,
                            const uint8_t* __restrict__ nviews /*[S*R]*/, float inv_s, float alpha_inter_ratio,
                            float background, const CompositeOut& o) {
    const int R = g.R;
    const float dx = g.rays_d[3 * r], dy = g.rays_d[3 * r + 1], dz = g.rays_d[3 * r + 2];
    float T = 1.f, wsum = 0.f, wmax = 0.f, asum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f, ge = 0.f, gm = 0.f;
    int n_seen = 0;
    // The per-sample inputs do not depend on the running transmittance: a block of CB samples is LOADED first (11 values each, all loads of the
    // block in flight together), then the serial chain runs over registers -- the arithmetic and its order are those of a plain loop over s.
    constexpr int CB = 8;
    float* __restrict__ wout = o.weights;
    float* __restrict__ cout_ = o.cdf;
    for (int sb = 0; sb < S; sb += CB) {
        float bm[CB], bd[CB], bs[CB], bz[CB], bg[CB][3], bc[CB][3];
        int bn[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int s = sb + k < S ? sb + k : S - 1;                 // the tail re-reads the last sample (never used)
            const size_t p = (size_t)s * R + r;
            bm[k] = pm[p]; bd[k] = dists[p]; bs[k] = sdf[p]; bz[k] = mid_z[p]; bn[k] = nviews[p];
            bg[k][0] = grad[3 * p]; bg[k][1] = grad[3 * p + 1]; bg[k][2] = grad[3 * p + 2];
            bc[k][0] = rgb[3 * p]; bc[k][1] = rgb[3 * p + 1]; bc[k][2] = rgb[3 * p + 2];
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (sb + k < S) {
                const size_t p = (size_t)(sb + k) * R + r;
                const float m = bm[k];
                const float gx = bg[k][0], gy = bg[k][1], gz = bg[k][2];
                float pc;
                const float alpha = composite_sample_alpha(dx, dy, dz, gx, gy, gz, m, bd[k], bs[k], inv_s, alpha_inter_ratio, pc);
                const float w = alpha * T;
                T = T * (1.f - alpha + 1e-7f);
                wout[p] = w;
                cout_[p] = pc;
                wsum += w; wmax = fmaxf(wmax, w); asum += alpha;
                c0 += bc[k][0] * w; c1 += bc[k][1] * w; c2 += bc[k][2] * w;
                dep += bz[k] * w;
                const float gn = sqrtf(gx * gx + gy * gy + gz * gz) - 1.f;
                ge += m * (gn * gn); gm += m;
                n_seen += bn[k] >= 2 ? 1 : 0;
            }
        }
    }
    const float bg = background * (1.f - wsum);
    o.color[3 * r] = c0 + bg; o.color[3 * r + 1] = c1 + bg; o.color[3 * r + 2] = c2 + bg;
    o.depth[r] = dep;
    o.weights_sum[r] = wsum; o.weights_max[r] = wmax; o.alpha_sum[r] = asum;
    o.grad_err[2 * r] = ge; o.grad_err[2 * r + 1] = gm;
    o.color_mask[r] = n_seen > 8 ? 1 : 0;
    float dv = 0.f;
    for (int sb = 0; sb < S; sb += CB) {
        float bz[CB], bw[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int s = sb + k < S ? sb + k : S - 1;
            const size_t p = (size_t)s * R + r;
            bz[k] = mid_z[p]; bw[k] = wout[p];
        }
#pragma unroll
        for (int k = 0; k < CB; ++k)
            if (sb + k < S) {
                const float d = bz[k] - dep;
                dv += d * d * bw[k];
            }
    }
    o.depth_var[r] = dv;
}

}  // namespace o2345
