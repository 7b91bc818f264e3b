// Error reporting + version for the o2345 C-ABI (see include/o2345.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

namespace o2345 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace o2345

#include <stddef.h>

namespace o2345 {
int preload_costvol();
int preload_sparse();
int preload_sparse_mfma();
int preload_sdf_mlp();
int preload_sdf_mlp_x3();
int preload_render();
int preload_list_sort();
int preload_color_maps();
int preload_color_pts();
int preload_mcubes();
int preload_mesh_pack();
int preload_featmaps();
int preload_convnet();
}

extern "C" {
const char* o2345_last_error(void) { return o2345::g_err; }
int o2345_version(void) { return 210; }     // 2.1: segment_rays + weight_cull in O2345RenderIO; 2.0: see include/o2345.h (1.5: o2345_list_sort_by_visibility; 1.4: color stats, bf16 entry removed; 1.3: o2345_conv2d family; 1.2: t_rand)

int o2345_preload(void) {
    int e, bad = 0;
    if ((e = o2345::preload_costvol())) bad = e;
    if ((e = o2345::preload_sparse())) bad = e;
    if ((e = o2345::preload_sparse_mfma())) bad = e;
    if ((e = o2345::preload_sdf_mlp())) bad = e;
    if ((e = o2345::preload_sdf_mlp_x3())) bad = e;
    if ((e = o2345::preload_render())) bad = e;
    if ((e = o2345::preload_list_sort())) bad = e;
    if ((e = o2345::preload_color_maps())) bad = e;
    if ((e = o2345::preload_color_pts())) bad = e;
    if ((e = o2345::preload_mcubes())) bad = e;
    if ((e = o2345::preload_mesh_pack())) bad = e;
    if ((e = o2345::preload_featmaps())) bad = e;
    if ((e = o2345::preload_convnet())) bad = e;
    O2345_REQUIRE(bad == 0, "preload: hipFuncGetAttributes failed (%s)", hipGetErrorString((hipError_t)bad));
    return 0;
}

const char* o2345_knobs(void) {
    static const struct Text { char s[160]; Text() { const o2345::Knobs& k = o2345::knobs();
        snprintf(s, sizeof s, "list_sort=%d sparse_brick=%d flat_sched=%d color_tiles=%d color_sched=%d ray_stream_min=%lld", k.list_sort, k.sparse_brick, k.flat_sched,
                 k.color_tiles, k.color_sched, k.ray_stream_min); } } t;
    return t.s;
}

// layout of O2345RenderIO as compiled into this library, in declaration order (include/o2345.h); the ctypes binding compares its own struct with it
size_t o2345_render_io_size(void) { return sizeof(O2345RenderIO); }
int o2345_render_io_layout(size_t* offsets_host, int n) {
#define F(name) offsetof(O2345RenderIO, name)
    static const size_t off[] = {
        F(sdf_blob), F(color_x3_blob), F(color_mfma_blob), F(vol_cl), F(maskvol), F(cmaps), F(proj), F(cam_pos), F(D), F(V), F(H), F(W),
        F(rays_o), F(rays_d), F(near_ray), F(far_ray), F(query_cam), F(t_rand), F(R), F(n_samples), F(n_importance), F(sdf_mode), F(segment_rays),
        F(near), F(far), F(sample_dist), F(inv_s), F(alpha_inter_ratio), F(background), F(weight_cull),
        F(mid_z), F(dists), F(pm), F(sdf), F(grad), F(rgb), F(nviews), F(color), F(depth), F(weights), F(cdf), F(weights_sum), F(weights_max),
        F(depth_var), F(alpha_sum), F(grad_err), F(color_mask), F(z_vals), F(scalars), F(color_stats)};
#undef F
    const int nf = (int)(sizeof off / sizeof off[0]);
    for (int i = 0; i < n && i < nf && offsets_host; ++i) offsets_host[i] = off[i];
    return nf;
}
}
