// Hierarchical ray sampling + NeuS compositing: SparseNeuSRenderer.render
// (models/sparse_neus_renderer.py:457-635) with up_sample (:73-115), cat_z_vals (:117-151), render_core (:171-455).
//
// The reference runs ~40 small kernels per up-sampling round per 512-ray chunk.  Here one lane owns one ray; all
// per-ray lists are sample-major ([S][R]) so every access of the wave is a coalesced row segment; the SDF network
// is called on flat point lists (occupied points compacted with a ballot/popcount prefix into an index list that
// the MFMA kernel consumes with a device-side count -- no host synchronisation anywhere in a render call).
#include "common.h"
#include "render_math.h"

namespace o2345 {

// coarse samples: z = near + (far-near) * linspace(0,1,S)  and their points, point index p = s*R + r
// t_rand (optional): the reference's stratified jitter (sparse_neus_renderer.py:506-515).  The reference draws
// t_rand = torch.rand(z_vals.shape) on the HOST ([R][S], ray-major) and sets z = lower + (upper - lower) * t_rand with
// lower/upper the midpoints to the neighbouring coarse samples; the caller hands the same tensor over, so the path is
// bit-reproducible under torch.manual_seed.
__global__ __launch_bounds__(256) void k_ray_coarse(RayGeom g, float near, float far, int S, const float* __restrict__ t_rand,
                                                    float* __restrict__ z, float* __restrict__ pts) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long long)S * g.R) return;
    const int s = (int)(p / g.R), r = (int)(p % g.R);
    float zz = near + (far - near) * linspace_at(0.f, 1.f, S, s);
    if (t_rand) {
        const float zp = near + (far - near) * linspace_at(0.f, 1.f, S, s > 0 ? s - 1 : 0);
        const float zn = near + (far - near) * linspace_at(0.f, 1.f, S, s < S - 1 ? s + 1 : S - 1);
        const float lower = s > 0 ? 0.5f * (zz + zp) : zz;            // mids = .5 * (z[1:] + z[:-1])
        const float upper = s < S - 1 ? 0.5f * (zn + zz) : zz;
        zz = lower + (upper - lower) * t_rand[(long long)r * S + s];
    }
    z[p] = zz;
    float x, y, w;
    ray_point(g, r, zz, x, y, w);
    pts[3 * p] = x; pts[3 * p + 1] = y; pts[3 * p + 2] = w;
}

// Occupied points are appended to a global list (order inside the list is irrelevant for the results: they are scattered back
// by slot).  One lane owns one ray and walks its samples, so a wave first collects the validity of all (sample, ray) pairs it
// owns as bit masks, reserves its whole range with ONE atomic, and then writes its slots sample by sample (ballot + popcount
// prefix): no block barrier and 1/S of the atomics of a per-sample reservation.
struct ValidBits { unsigned w[8]; };                 // up to 256 samples per ray
__device__ __forceinline__ void append_wave(const ValidBits& bits, int n_samples, int my_count, int R, int r,
                                            int* __restrict__ list, int* __restrict__ count) {
    int total = my_count;
#pragma unroll
    for (int off = 32; off; off >>= 1) total += __shfl_xor(total, off);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && total) base = atomicAdd(count, total);
    base = __shfl(base, 0);
    if (!total) return;
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
    for (int s = 0; s < n_samples; ++s) {
        const bool valid = (bits.w[s >> 5] >> (s & 31)) & 1u;
        const unsigned long long m = __ballot(valid);
        if (valid) list[base + __popcll(m & lt)] = s * R + r;
        base += __popcll(m);
    }
}

__global__ __launch_bounds__(256) void k_ray_upsample(RayGeom g, const float* __restrict__ z, const float* __restrict__ sdf, int S,
                                                      float inv_s, const float* __restrict__ maskvol, int D,
                                                      float* __restrict__ wbuf, int n_imp, float* __restrict__ new_z,
                                                      float* __restrict__ new_pts, float* __restrict__ new_sdf,
                                                      int* __restrict__ list, int* __restrict__ count) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool live = r < g.R;
    if (live) upsample_ray(g, r, z, sdf, S, inv_s, maskvol, D, wbuf, n_imp, new_z);
    ValidBits bits{};
    int cnt = 0;
    for (int t = 0; t < n_imp; ++t) {
        const int slot = t * g.R + r;
        if (live) {
            float x, y, w;
            ray_point(g, r, new_z[slot], x, y, w);
            new_pts[3 * (size_t)slot] = x; new_pts[3 * (size_t)slot + 1] = y; new_pts[3 * (size_t)slot + 2] = w;
            new_sdf[slot] = 100.f;                               // cat_z_vals default outside the mask (:135)
            if (mask_at(maskvol, D, x, y, w) > 0.f) { bits.w[t >> 5] |= 1u << (t & 31); ++cnt; }
        }
    }
    append_wave(bits, n_imp, cnt, g.R, r, list, count);
}

// cat_z_vals quirk (:137): the SDF of the new points is evaluated only if MORE THAN ONE of them is inside the mask
__global__ void k_quirk_min2(int* count) { if (*count <= 1) *count = 0; }
// render_core quirk (:222-223): with no valid point at all, the first 100 points of the chunk (ray 0, samples 0..99
// in the reference's ray-major order) are evaluated anyway
__global__ void k_quirk_first100(int* count, int* list, int R, int S) {
    if (*count >= 1) return;
    const int t = threadIdx.x;                     // launched with 128 threads
    if (t < 100 && t < S) list[t] = t * R;         // slot of (ray 0, sample t)
    if (t == 0) *count = (100 < S ? 100 : S);
}

__global__ __launch_bounds__(256) void k_ray_merge(int R, float* z, float* sdf, int S, float* new_z, float* new_sdf, int n_new) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) merge_ray(r, R, z, sdf, S, new_z, new_sdf, n_new);
}

// render_core head (:204-218): section lengths, mid points, occupancy of the mid points, defaults, valid list
__global__ __launch_bounds__(256) void k_ray_finalize(RayGeom g, const float* __restrict__ z, int S, float sample_dist,
                                                      const float* __restrict__ maskvol, int D, float* __restrict__ mid_z,
                                                      float* __restrict__ dists, float* __restrict__ pts,
                                                      float* __restrict__ pm, float* __restrict__ sdf, float* __restrict__ grad,
                                                      float* __restrict__ rgb, int* __restrict__ list, int* __restrict__ count,
                                                      int defaults_everywhere) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool live = r < g.R;
    ValidBits bits{};
    int cnt = 0;
    for (int s = 0; s < S; ++s) {
        const size_t p = (size_t)s * g.R + r;
        if (live) {
            const float z0 = z[p];
            const float d = (s + 1 < S) ? z[p + g.R] - z0 : sample_dist;
            const float mz = z0 + d * 0.5f;
            float x, y, w;
            ray_point(g, r, mz, x, y, w);
            const float m = mask_at(maskvol, D, x, y, w);
            dists[p] = d; mid_z[p] = mz; pm[p] = m;
            pts[3 * p] = x; pts[3 * p + 1] = y; pts[3 * p + 2] = w;
            if (m > 0.f) { bits.w[s >> 5] |= 1u << (s & 31); ++cnt; }
            if (!(m > 0.f) || defaults_everywhere) {
                // the reference's defaults (:231: sdf = 100, gradients = colours = 0).  Inside o2345_render_rays occupied points are ALWAYS overwritten by the
                // network kernels that consume the list (every list entry is evaluated), so only unoccupied points need them there: 28 bytes less per
                // occupied point.  The public stage entry initialises every slot (a caller may evaluate only part of the list).
                sdf[p] = 100.f;
                grad[3 * p] = 0.f; grad[3 * p + 1] = 0.f; grad[3 * p + 2] = 0.f;
                rgb[3 * p] = 0.f; rgb[3 * p + 1] = 0.f; rgb[3 * p + 2] = 0.f;
            }
        }
    }
    append_wave(bits, S, cnt, g.R, r, list, count);
}

__global__ __launch_bounds__(256) void k_ray_composite(RayGeom g, int S, const float* __restrict__ mid_z, const float* __restrict__ dists,
                                                       const float* __restrict__ pm, const float* __restrict__ sdf,
                                                       const float* __restrict__ grad, const float* __restrict__ rgb,
                                                       const uint8_t* __restrict__ nviews, float inv_s, float air, float bg,
                                                       CompositeOut o) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < g.R) composite_ray(g, r, S, mid_z, dists, pm, sdf, grad, rgb, nviews, inv_s, air, bg, o);
}

}  // namespace o2345

using namespace o2345;

extern "C" {

int o2345_sdf_mlp(int variant, const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index,
                  const int32_t* n_dev, long long n, int grid_R, float sign, float* out_sdf, float* out_feat,
                  float* out_lat, float* out_grad, void* stream);
int o2345_sdf_mlp_x3(const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index, const int32_t* n_dev,
                     long long n, int grid_R, float sign, float* out_sdf, void* stream);
int o2345_sdf_grad_x3(const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index, const int32_t* n_dev,
                      long long n, int grid_R, float sign, float* out_sdf, float* out_grad, void* stream);
int o2345_color_points_x3(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                          const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                          const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                          const float* normals, float* out_rgb, uint8_t* out_nviews, void* stream);
int o2345_color_points(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                       const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                       const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                       const float* normals, float* out_rgb, uint8_t* out_nviews, void* stream);
int o2345_color_points_mfma(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                            const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                            const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                            const float* normals, float* out_rgb, uint8_t* out_nviews, void* stream);
int o2345_view_count_unlisted(const float* pts, long long n, const float* skip_if_positive, const float* maskvol, int D, const float* proj, int V,
                              int H, int W, uint8_t* out, void* stream);

// ---- stage entry points (used by the parity tests; the orchestrator below calls the same kernels) -----------------
int o2345_ray_coarse_jitter(const float* rays_o, const float* rays_d, int R, float near, float far, int S, const float* t_rand,
                            float* z, float* pts, void* stream) {
    O2345_REQUIRE(rays_o && rays_d && z && pts && R > 0 && S > 1, "ray_coarse: bad arguments");
    RayGeom g{rays_o, rays_d, R};
    hipLaunchKernelGGL(k_ray_coarse, dim3(cdiv((long long)R * S, 256)), dim3(256), 0, (hipStream_t)stream, g, near, far, S, t_rand, z, pts);
    return check_launch("ray_coarse");
}

int o2345_ray_coarse(const float* rays_o, const float* rays_d, int R, float near, float far, int S, float* z, float* pts, void* stream) {
    return o2345_ray_coarse_jitter(rays_o, rays_d, R, near, far, S, nullptr, z, pts, stream);
}

int o2345_ray_upsample(const float* rays_o, const float* rays_d, int R, const float* z, const float* sdf, int S, float inv_s,
                       const float* maskvol, int D, float* wbuf, int n_imp, float* new_z, float* new_pts, float* new_sdf,
                       int32_t* list, int32_t* count_dev, void* stream) {
    O2345_REQUIRE(rays_o && rays_d && z && sdf && maskvol && wbuf && new_z && new_pts && new_sdf && list && count_dev, "ray_upsample: null pointer");
    O2345_REQUIRE(n_imp >= 1 && n_imp <= 256, "ray_upsample: 1..256 new samples per call (got %d)", n_imp);
    RayGeom g{rays_o, rays_d, R};
    hipStream_t s = (hipStream_t)stream;
    O2345_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_ray_upsample, dim3(cdiv(R, 256)), dim3(256), 0, s, g, z, sdf, S, inv_s, maskvol, D, wbuf, n_imp, new_z, new_pts, new_sdf, list, count_dev);
    hipLaunchKernelGGL(k_quirk_min2, dim3(1), dim3(1), 0, s, count_dev);
    return check_launch("ray_upsample");
}

int o2345_ray_merge(int R, float* z, float* sdf, int S, float* new_z, float* new_sdf, int n_new, void* stream) {
    O2345_REQUIRE(z && sdf && new_z && new_sdf, "ray_merge: null pointer");
    hipLaunchKernelGGL(k_ray_merge, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, R, z, sdf, S, new_z, new_sdf, n_new);
    return check_launch("ray_merge");
}

static int ray_finalize_launch(const float* rays_o, const float* rays_d, int R, const float* z, int S, float sample_dist,
                               const float* maskvol, int D, float* mid_z, float* dists, float* pts, float* pm, float* sdf,
                               float* grad, float* rgb, int32_t* list, int32_t* count_dev, void* stream, int defaults_everywhere) {
    O2345_REQUIRE(rays_o && rays_d && z && maskvol && mid_z && dists && pts && pm && sdf && grad && rgb && list && count_dev, "ray_finalize: null pointer");
    O2345_REQUIRE(S >= 1 && S <= 256, "ray_finalize: at most 256 samples per ray (got %d)", S);
    RayGeom g{rays_o, rays_d, R};
    hipStream_t s = (hipStream_t)stream;
    O2345_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_ray_finalize, dim3(cdiv(R, 256)), dim3(256), 0, s, g, z, S, sample_dist, maskvol, D, mid_z, dists, pts, pm, sdf, grad, rgb, list, count_dev,
                       defaults_everywhere);
    return check_launch("ray_finalize");
}

int o2345_ray_finalize(const float* rays_o, const float* rays_d, int R, const float* z, int S, float sample_dist,
                       const float* maskvol, int D, float* mid_z, float* dists, float* pts, float* pm, float* sdf,
                       float* grad, float* rgb, int32_t* list, int32_t* count_dev, void* stream) {
    return ray_finalize_launch(rays_o, rays_d, R, z, S, sample_dist, maskvol, D, mid_z, dists, pts, pm, sdf, grad, rgb, list, count_dev, stream, 1);
}

int o2345_ray_composite(const float* rays_o, const float* rays_d, int R, int S, const float* mid_z, const float* dists,
                        const float* pm, const float* sdf, const float* grad, const float* rgb, const uint8_t* nviews,
                        float inv_s, float alpha_inter_ratio, float background, float* color, float* depth, float* weights,
                        float* cdf, float* weights_sum, float* weights_max, float* depth_var, float* alpha_sum,
                        float* grad_err, uint8_t* color_mask, void* stream) {
    O2345_REQUIRE(rays_o && rays_d && mid_z && dists && pm && sdf && grad && rgb && nviews && color && depth && weights && cdf &&
                  weights_sum && weights_max && depth_var && alpha_sum && grad_err && color_mask, "ray_composite: null pointer");
    RayGeom g{rays_o, rays_d, R};
    CompositeOut o{color, depth, weights, cdf, weights_sum, weights_max, depth_var, alpha_sum, grad_err, color_mask};
    hipLaunchKernelGGL(k_ray_composite, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, g, S, mid_z, dists, pm, sdf, grad, rgb, nviews, inv_s, alpha_inter_ratio, background, o);
    return check_launch("ray_composite");
}

// ---- the whole render() call -----------------------------------------------------------------------------------------
// Workspace layout (floats unless noted), S = n_samples + n_importance, R rays:
//   z[S*R] sdf[S*R] wbuf[S*R] new_z[NI*R] new_sdf[NI*R] pts[3*S*R] list[S*R ints] count[16 ints]
//   ... sorted list[S*R ints] + the workspace of o2345_list_sort_by_visibility for up to 32 views (csrc/list_sort.hip)
size_t o2345_list_sort_workspace_bytes(long long n_max, int V);
int o2345_list_sort_by_visibility(const float* pts, const int32_t* list, const int32_t* count_dev, long long n_max, const float* proj, int V, int H, int W,
                                  int32_t* list_out, uint32_t* keys_out, void* workspace, size_t workspace_bytes, void* stream);
static size_t render_core_workspace_bytes(int R, int n_samples, int n_importance) {
    const size_t S = (size_t)n_samples + n_importance, NI = (size_t)(n_importance / 4 > 0 ? n_importance / 4 : 1);
    return ((S * 3 + NI * 2 + 3 * S + S) * (size_t)R + 64) * 4;
}
size_t o2345_render_workspace_bytes(int R, int n_samples, int n_importance) {
    const size_t S = (size_t)n_samples + n_importance;
    return render_core_workspace_bytes(R, n_samples, n_importance) + S * (size_t)R * 4 + o2345_list_sort_workspace_bytes((long long)(S * (size_t)R), 32);
}

struct O2345RenderIO {
    // scene
    const float* sdf_blob; const float* color_blob; const float* vol_cl; const float* maskvol; int D;
    const float* cmaps; const float* proj; const float* cam_pos; int V, H, W;
    // rays
    const float* rays_o; const float* rays_d; int R; float near, far; int n_samples, n_importance;
    float inv_s, alpha_inter_ratio, background; const float* query_cam;
    // outputs: per-sample arrays are sample-major [S][R] (+[,3])
    float* mid_z; float* dists; float* pm; float* sdf; float* grad; float* rgb; uint8_t* nviews;
    float* color; float* depth; float* weights; float* cdf; float* weights_sum; float* weights_max; float* depth_var;
    float* alpha_sum; float* grad_err; uint8_t* color_mask; float* z_vals;
    const float* color_mfma_blob;
    int sdf_bf16;               // SDF network mode: 0 fp32 MFMA, 2 split-f16 (sdf_mlp_x3.hip).  (1 was the bf16 mode removed in round 3; the field keeps its name: ABI)
    const float* color_x3_blob; // optional: split-f16 colour kernel
    const float* t_rand;        // optional [R][n_samples]: stratified jitter of the coarse samples (perturb > 0)
};

int o2345_render_rays(const O2345RenderIO* io, void* workspace, size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(io && workspace, "render_rays: null pointer");
    const int R = io->R, NS = io->n_samples, NIMP = io->n_importance;
    O2345_REQUIRE(NIMP % 4 == 0 && NIMP > 0 && NS > 1, "render_rays: n_importance must be a positive multiple of 4");
    O2345_REQUIRE(workspace_bytes >= o2345_render_workspace_bytes(R, NS, NIMP), "render_rays: workspace too small");
    O2345_REQUIRE(R > 0 && ((long long)NS + NIMP) * (long long)R < 2147483647LL, "render_rays: R * (n_samples + n_importance) must stay below 2^31 "
                  "(sample slots are 32-bit); split the ray batch (got R = %d)", R);
    const size_t S = (size_t)NS + NIMP, NI = NIMP / 4, RR = R;
    float* z = (float*)workspace;
    float* sdf = z + S * RR;
    float* wbuf = sdf + S * RR;
    float* new_z = wbuf + S * RR;
    float* new_sdf = new_z + NI * RR;
    float* pts = new_sdf + NI * RR;
    int* list = (int*)(pts + 3 * S * RR);
    int* count = list + S * RR;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    O2345_REQUIRE(io->sdf_bf16 == 0 || io->sdf_bf16 == 2, "render_rays: SDF mode %d (0 = fp32, 2 = split-f16; the bf16 mode was removed)", io->sdf_bf16);
    auto sdf_eval = [&](int variant, const float* p, const int* idx, const int* cnt, long long n, float* out, float* grad) {
        if (io->sdf_bf16 == 2 && variant == 0) return o2345_sdf_mlp_x3(io->sdf_blob, io->vol_cl, io->D, p, idx, cnt, n, 0, 1.f, out, stream);
        if (io->sdf_bf16 == 2 && variant == 2) return o2345_sdf_grad_x3(io->sdf_blob, io->vol_cl, io->D, p, idx, cnt, n, 0, 1.f, out, grad, stream);
        return o2345_sdf_mlp(variant, io->sdf_blob, io->vol_cl, io->D, p, idx, cnt, n, 0, 1.f, out, nullptr, nullptr, grad, stream);
    };
    if ((rc = o2345_ray_coarse_jitter(io->rays_o, io->rays_d, R, io->near, io->far, NS, io->t_rand, z, pts, stream))) return rc;
    // coarse SDF on ALL points (not masked, :525-528)
    if ((rc = sdf_eval(0, pts, nullptr, nullptr, (long long)NS * R, sdf, nullptr))) return rc;
    int cur = NS;
    for (int i = 0; i < 4; ++i) {
        if ((rc = o2345_ray_upsample(io->rays_o, io->rays_d, R, z, sdf, cur, 64.f * (float)(1 << i), io->maskvol, io->D, wbuf, (int)NI, new_z, pts, new_sdf, list, count, stream))) return rc;
        if ((rc = sdf_eval(0, pts, list, count, 0, new_sdf, nullptr))) return rc;
        if ((rc = o2345_ray_merge(R, z, sdf, cur, new_z, new_sdf, (int)NI, stream))) return rc;
        cur += (int)NI;
    }
    const float sample_dist = (io->far - io->near) / (float)NS;
    float* fpts = pts;   // reuse
    if ((rc = ray_finalize_launch(io->rays_o, io->rays_d, R, z, (int)S, sample_dist, io->maskvol, io->D, io->mid_z, io->dists, fpts, io->pm, io->sdf, io->grad, io->rgb, list, count, stream, 0))) return rc;
    hipLaunchKernelGGL(k_quirk_first100, dim3(1), dim3(128), 0, s, count, list, R, (int)S);
    // the list grouped by view-visibility signature (stable): the colour kernel then skips every (tile, view) pair in which no point sees the view instead of
    // 3/4 of them -- 40.0 -> 36.1 ms at 8 views, bit-identical results (csrc/list_sort.hip).  O2345_LIST_SORT=0: the emission order (A/B knob)
    {
        const char* e = getenv("O2345_LIST_SORT");
        if (!(e && e[0] == '0') && io->V <= 32) {
            int* slist = (int*)((char*)workspace + render_core_workspace_bytes(R, NS, NIMP));
            void* sort_ws = (void*)(slist + S * RR);
            if ((rc = o2345_list_sort_by_visibility(fpts, list, count, (long long)(S * RR), io->proj, io->V, io->H, io->W, slist, nullptr, sort_ws,
                                                    o2345_list_sort_workspace_bytes((long long)(S * RR), 32), stream))) return rc;
            list = slist;
        }
    }
    if ((rc = sdf_eval(2, fpts, list, count, 0, io->sdf, io->grad))) return rc;
    // valid-view counts (feed the per-ray colour mask): the colour kernels write them for the points they evaluate (the occupied ones, 88 % at
    // BASELINE config 2); this pass covers the rest (four IEEE divisions per view make it VALU-bound: 0.48 ms over all points)
    if ((rc = o2345_view_count_unlisted(fpts, (long long)S * R, io->pm, io->maskvol, io->D, io->proj, io->V, io->H, io->W, io->nviews, stream))) return rc;
    if (io->color_x3_blob)
        rc = o2345_color_points_x3(io->color_x3_blob, io->vol_cl, io->maskvol, io->D, io->cmaps, io->proj, io->cam_pos, io->V, io->H, io->W, fpts, list, count, 0, io->query_cam, nullptr, io->rgb, io->nviews, stream);
    else if (io->color_mfma_blob)
        rc = o2345_color_points_mfma(io->color_mfma_blob, io->vol_cl, io->maskvol, io->D, io->cmaps, io->proj, io->cam_pos, io->V, io->H, io->W, fpts, list, count, 0, io->query_cam, nullptr, io->rgb, io->nviews, stream);
    else
        rc = o2345_color_points(io->color_blob, io->vol_cl, io->maskvol, io->D, io->cmaps, io->proj, io->cam_pos, io->V, io->H, io->W, fpts, list, count, 0, io->query_cam, nullptr, io->rgb, io->nviews, stream);
    if (rc) return rc;
    if ((rc = o2345_ray_composite(io->rays_o, io->rays_d, R, (int)S, io->mid_z, io->dists, io->pm, io->sdf, io->grad, io->rgb, io->nviews, io->inv_s, io->alpha_inter_ratio, io->background,
                                  io->color, io->depth, io->weights, io->cdf, io->weights_sum, io->weights_max, io->depth_var, io->alpha_sum, io->grad_err, io->color_mask, stream))) return rc;
    if (io->z_vals) O2345_HIP(hipMemcpyAsync(io->z_vals, z, S * RR * sizeof(float), hipMemcpyDeviceToDevice, s));
    return check_launch("render_rays");
}

}  // extern "C"
