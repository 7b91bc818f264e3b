/* o2345.h -- C ABI of libo2345_hip.so: the MI355X (gfx950) reconstruction back end for One-2-3-45.
 *
 * Drop-in boundary: the reference has no native code of its own; its hot path reaches native kernels
 * only through three un-vendored pip packages (torchsparse v1.4.0, inplace_abn, PyMCubes) and ATen.  This header is
 * what a binding for that path links against; the Python shims in one-2-3-45_amd/ (ctypes) expose
 * the reference's own Python operator surface on top of it.  Every entry cites the reference interface it replaces
 * (paths relative to /root/reference/reconstruction).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.  All pointers are DEVICE pointers unless the name
 *     ends in _host.  The caller owns every buffer; the library never allocates, frees or retains them.
 *   - every function returns 0 on success, a negative code on error; o2345_last_error() (thread-local) explains.
 *   - `stream` is a hipStream_t (NULL = default stream).  Functions are asynchronous unless stated otherwise and
 *     re-entrant (one Python thread per device under nn.DataParallel is fine).
 *   - all arithmetic is fp32 (fp64 for batch-norm statistics and marching-cubes vertices), voxel / row ids int32.
 *   - variable-size outputs: capacity buffer + device-side count (cost volume, sparse levels) or the two-call
 *     count -> emit protocol (marching cubes).
 *   - no process-global state: nothing is retained between calls except, per library instance, the O2345_* debug knobs (read once, o2345_knobs())
 *     and which kernels had their LDS limit raised on which device.  One host thread per stream / device; several streams may share a GPU.
 *
 * Accuracy contract (HIP vs the reference's arithmetic as restated by oracle/, which is pinned to reference-generated golden vectors)
 *   - exact: kept-voxel set, visible-view counts, sparse level coordinates, nearest-mask lookups, valid-view counts and colour masks, list contents,
 *     marching-cubes triangles for an identical scalar field; results do not depend on batch composition (chunked == one call, bit for bit) or on
 *     which of the two sampler kernel forms runs.
 *   - per stage, max abs error relative to max(1, |reference|): cost-volume rows 2e-5 (1.3e-4 at 128^3: var = E[x^2] - mean^2 cancels), sparse CNN and
 *     dense volume 3.5e-5, SDF 2e-6, SDF gradient 1e-5, new sample depths 2e-4 and 5e-3 of their bin, compositing on identical sample lists colour 3e-5,
 *     depth / weights / depth variance 2e-5.  Both numerical modes (O2345RenderIO.sdf_mode 2 = split-f16 on the matrix cores, 0 = fp32 MFMA) meet the same bounds.
 *   - end to end through render(): the reference's hierarchical sampler amplifies fp32-class SDF differences (sigmoid slopes 64 ... 512, inverse CDF over
 *     nearly empty bins), so the bound is distributional: sample lists within one coarse section, rays with coinciding lists as tight as the stage bound,
 *     and at BASELINE config 2 colour error max <= 6e-2 (measured 4.2e-2), q99 <= 8e-3 (3.1e-3), <= 6 % of the rays above 1e-3 (2.2 %) -- within 4x what
 *     the reference itself does under 1e-6 SDF noise.  Mesh: a lattice node may change sign only where |u| <= 9e-7 (0 - 1 node of 262,144), sizes equal, IoU >= 0.999.
 *   - unpinned (no source in the reference tree): inplace_abn's |gamma| + eps, torchsparse v1.4.0's kernel maps, PyMCubes' vertex numbering.
 */
#ifndef O2345_H
#define O2345_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* o2345_last_error(void);
/* 200 = ABI 2.0 (round 4): O2345RenderIO re-laid out (sdf_mode replaces the fossil name sdf_bf16; color_blob / the VALU colour kernel removed;
 * per-ray near / far; caller-owned colour work counters; per-call scalars) and layout-checked (o2345_render_io_*), o2345_camera_terms,
 * `identity_rows` on o2345_sparse_conv3d_x3, no process-global state left in the library (the colour work counters are a caller-owned buffer). */
/* 210 = ABI 2.1 (round 5): O2345RenderIO gains `segment_rays` (the reference's two per-CALL rules applied per segment of a fused call: a whole image
 * behind the trainer's unchanged 512-ray chunk loop) and `weight_cull` (tolerance-bounded colour work removal); o2345_ray_composite is unchanged. */
int o2345_version(void);
/* Layout self-description of O2345RenderIO as THIS library was compiled (sizeof, and offsetof of every field in declaration order): a binding
 * asserts its own struct against it at load time (one-2-3-45_amd/_lib.py does) -- a field added on one side only cannot corrupt calls silently.
 * render_io_layout writes min(n, number of fields) offsets and returns the number of fields. */
size_t o2345_render_io_size(void);
int o2345_render_io_layout(size_t* offsets_host, int n);
/* The debug / A-B knobs this library instance runs with, e.g. "list_sort=1 sparse_brick=1 flat_sched=0 color_tiles=0 color_sched=10".  They are read from
 * the environment (O2345_LIST_SORT, O2345_SPARSE_BRICK, O2345_FLAT_SCHED, O2345_COLOR_KERNEL, O2345_COLOR_SCHED) ONCE, at the first call that needs
 * them -- this call included --, never on a launch path. */
const char* o2345_knobs(void);

/* ---- cost volume -------------------------------------------------------------------------------------------------
 * replaces: ops/generate_grids.py:4 generate_grid, ops/back_project.py:5 back_project_sparse_type (both calls),
 * models/sparse_sdf_network.py:221 aggregate_multiview_features, the frustum filter at :330-334.
 * proj: [V,4,4] row-major affine matrices (sample['affine_mats']); voxel (x,y,z) -> world = xyz*voxel_size + origin. */
size_t o2345_costvol_workspace_bytes(int dx, int dy, int dz);
/* visible-view count per voxel, order-preserving compaction of the voxels seen by > min_views views:
 * cnt [dx*dy*dz] u8, row_of_voxel [dx*dy*dz] (row id or -1), coords [capacity,4] int32 (x,y,z,batch=0), *n_rows_dev. */
int o2345_costvol_index(const float* proj, int V, int H, int W, int dx, int dy, int dz, float voxel_size,
                        const float* origin_host, int min_views, uint8_t* cnt, int32_t* row_of_voxel, int32_t* coords,
                        int32_t* n_rows_dev, void* workspace, size_t workspace_bytes, void* stream);
/* feats_nhwc [V,H,W,C] (C = 8|16) -> out_rows [n_rows, 2C] = cat(variance, mean) over the V views. */
int o2345_costvol_gather(const float* feats_nhwc, const float* proj, int V, int H, int W, int C, int dx, int dy, int dz,
                         float voxel_size, const float* origin_host, const uint8_t* cnt, const int32_t* coords,
                         int n_rows, float* out_rows, void* stream);
int o2345_nchw_to_nhwc(const float* in, float* out, int V, int C, int H, int W, void* stream);

/* ---- feature pyramid glue of FeatureNet (replaces the ATen kernels between its convolutions; ABI 1.2) ----------------------------------------
 * fpn_level: out [V,32,H,W] = 1x1 conv (weight [32,c_in], bias [32]) of fine [V,c_in,H,W] + bilinear x2 up-sampling (align_corners = True) of
 *   coarse [V,32,H/2,W/2]   (models/featurenet.py:73-76, 85-86: _upsample_add(feat, lat(conv))); c_in = 8 or 16.
 * pyramid_pack: the fused pyramid of models/trainer_generic.py:1117-1123 = [x4 up-sampled f2 (32) | x2 up-sampled s1 (16) | s0 (8)], written once:
 *   fmaps_nchw [V,56,H,W] (optional, may be NULL) and cmaps_nhwc64 [V,H,W,64] = rgb (3) | those 56 features | 5 zeros (the map the colour kernels
 *   gather from; replaces o2345_pack_color_maps on this path). */
int o2345_fpn_level(const float* fine, int c_in, const float* coarse, const float* weight, const float* bias, int V, int H, int W,
                    float* out, void* stream);
int o2345_pyramid_pack(const float* f2, const float* s1, const float* s0, const float* rgb, int V, int H, int W, float* fmaps_nchw,
                       float* cmaps_nhwc64, void* stream);
/* ---- the convolutions of FeatureNet and of the compress layer (ABI 1.3; replace nn.Conv2d + InPlaceABN pairs: models/featurenet.py:12-22
 * ConvBnReLU, :40-66 FeatureNet, models/sparse_sdf_network.py:171-173 compress_layer) ---------------------------------------------------------
 * conv2d: out [V,cout,Ho,Wo] = conv2d(act(in), w) (+ bias), padding k / 2, stride 1 or 2; w_packed = [cin][k][k][cout] (conv2d_pack_weights of an
 *   nn.Conv2d weight [cout,cin,k,k]).  in_scale_shift [2*cin] (may be NULL): the InPlaceABN of the PRODUCER of `in` is applied while `in` is read,
 *   act(x) = leaky_relu(x * scale + shift, slope) -- the activated tensor is never stored.  out_scale_shift [2*cout] (may be NULL): batch
 *   statistics of `out` over (V,Ho,Wo) -> this layer's own (scale, shift) = ((|gamma| + eps) / sqrt(var + eps), beta - mean * scale) for the
 *   consumer to apply (needs gamma, beta and a workspace of conv2d_workspace_bytes).  Shapes (fp32 form): the (cin, cout, k, stride) combinations of FeatureNet
 *   and the compress layer (16 or 8 outputs); anything else is an error.  The matrix-core form takes any cin <= 64, cout <= 32 with those (k, stride).
 * fpn_level_act: fpn_level whose `fine` input is a raw convolution output with its (scale, shift) applied on load.
 * scale_shift_act: y = leaky_relu(x * scale + shift) for x [V,C,H,W], written as NCHW and / or channel-last NHWC (C = 8, 16, 32). */
int o2345_conv2d_pack_weights(const float* w_oihw, int cout, int cin, int k, float* packed, void* stream);
size_t o2345_conv2d_workspace_bytes(int V, int cout, int Ho, int Wo);
int o2345_conv2d(const float* in, int V, int cin, int Hi, int Wi, int in_pixel_stride, int in_channel_offset, const float* in_scale_shift, float slope,
                 const float* w_packed, const float* bias, int cout, int k, int stride, float* out, const float* gamma, const float* beta, float eps,
                 int abs_gamma, float* out_scale_shift, void* workspace, size_t workspace_bytes, void* stream);
/* in_pixel_stride = 0: `in` is [V,cin,Hi,Wi] (channel-first).  in_pixel_stride > 0: `in` is a channel-last map [V,Hi,Wi,in_pixel_stride] and the
 * convolution reads its channels in_channel_offset .. + cin (the compress layer reads the 56 features of the [V,H,W,64] colour map at offset 3, so the
 * channel-first fused pyramid is never written on that path).
 * the same convolution on the matrix cores (split-f16 operands, fp32 accumulate: the default numerical mode); cin <= 64, cout <= 32;
 * w_packed_x3 [conv2d_x3_weight_floats] from conv2d_pack_weights_x3 */
size_t o2345_conv2d_x3_weight_floats(int cin, int k);
int o2345_conv2d_pack_weights_x3(const float* w_oihw, int cout, int cin, int k, float* packed, void* stream);
int o2345_conv2d_x3(const float* in, int V, int cin, int Hi, int Wi, int in_pixel_stride, int in_channel_offset, const float* in_scale_shift, float slope,
                    const float* w_packed_x3, const float* bias, int cout, int k, int stride, float* out, const float* gamma, const float* beta, float eps,
                    int abs_gamma, float* out_scale_shift, void* workspace, size_t workspace_bytes, void* stream);
int o2345_fpn_level_act(const float* fine, const float* fine_scale_shift, float slope, int c_in, const float* coarse, const float* weight,
                        const float* bias, int V, int H, int W, float* out, void* stream);
int o2345_scale_shift_act(const float* x, int V, int C, int H, int W, const float* scale_shift, float slope, float* y_nchw, float* y_nhwc,
                          void* stream);
/* lod > 0 (sparse_sdf_network.py:335-357): the same two passes for an explicit voxel list coords [n,4] (x,y,z,b) in arbitrary
 * order; cnt / cnt_row are per LIST ROW.  build_index_grid makes the dense row lookup of such a list (cells of size ts). */
int o2345_visible_count_list(const float* proj, int V, int H, int W, float voxel_size, const float* origin_host,
                             const int32_t* coords, int n, uint8_t* cnt, void* stream);
int o2345_costvol_gather_list(const float* feats_nhwc, const float* proj, int V, int H, int W, int C, float voxel_size,
                              const float* origin_host, const uint8_t* cnt_row, const int32_t* coords, int n_rows,
                              float* out_rows, void* stream);
int o2345_build_index_grid(const int32_t* coords, int n, int ts, int nx, int ny, int nz, int32_t* grid, void* stream);
/* replaces the prune() of get_valid_sparse_coords_by_sdf (sparse_neus_renderer.py:838-848): out[v] = mask[v] > 0 and any
 * |sdf| < threshold within the (2*radius+1)^3 box around v (the reference's avg_pool3d(7) > 0 dilation: radius 3). */
int o2345_prune_dilate(const float* sdf, const float* mask, int D, float threshold, int radius, uint8_t* out, void* stream);
/* replaces: tsparse/torchsparse_utils.py:125 sparse_to_dense_channel + sparse_sdf_network.py:252 sparse_to_dense_volume.
 * rows [N,C] -> dense_cl [D^3,C] (channel-last, what the samplers here read), dense_cf [C,D^3] (the reference's
 * [1,C,X,Y,Z]) and mask [D^3]; any output may be NULL. */
int o2345_scatter_dense(const float* rows, const int32_t* row_of_voxel, int C, long long nvox, float* dense_cl,
                        float* dense_cf, float* mask, void* stream);

/* ---- sparse cost-regularisation CNN (replaces torchsparse v1.4.0: SparseTensor kernel maps, spnn.Conv3d,
 * spnn.BatchNorm, spnn.ReLU as used by tsparse/modules.py:94-124,259-304) -------------------------------------------- */
size_t o2345_sparse_downsample_workspace_bytes(int nxc, int nyc, int nzc);
/* spdownsample(stride 2, kernel 3): coarse level (cell size 2*ts) from the fine coordinates. */
int o2345_sparse_downsample(const int32_t* coords_fine, int n_fine, int ts, int nxc, int nyc, int nzc,
                            int32_t* row_of_cell, int32_t* coords_coarse, int32_t* n_coarse_dev, void* workspace,
                            size_t workspace_bytes, void* stream);
/* spnn.Conv3d(kernel 3, no bias).  mode 0: stride 1; 1: stride 2 (in = finer level); 2: transposed stride 2
 * (in = coarser level).  in_grid: the index grid (gx,gy,gz cells) of the INPUT level; kernel [27,cin,cout]. */
int o2345_sparse_conv3d(int mode, const float* in, int cin, const int32_t* in_grid, int gx, int gy, int gz,
                        const int32_t* out_coords, int n_out, int ts_out, const float* kernel, int cout, float* out,
                        void* stream);
/* the same convolution on the f16 matrix cores in split precision (fp32-class accuracy, see o2345_sdf_mlp_x3);
 * wblob: the kernel packed by weights.pack_sparse_conv_x3, o2345_sparse_conv_x3_blob_floats(cin, cout) floats.
 * identity_rows != 0 (mode 0 only): the CALLER GUARANTEES that out_coords is the coordinate list in_grid was built from, in the same order
 * (output row q = input row q, n_out = number of input rows) -- what a stride-1 spnn.Conv3d always produces.  Only then may the LDS-tiled brick
 * kernel run (32 -> 16 channels: it writes out[in_grid[site]] and never reads out_coords); with identity_rows = 0 every shape takes the gather
 * form, which honours any subset / order of out_coords. */
int o2345_sparse_conv_x3_blob_floats(int cin, int cout);
int o2345_sparse_conv3d_x3(int mode, const float* in, int cin, const int32_t* in_grid, int gx, int gy, int gz,
                           const int32_t* out_coords, int n_out, int ts_out, const float* wblob, int cout, int identity_rows, float* out,
                           void* stream);
size_t o2345_bn_workspace_bytes(int C);
/* spnn.BatchNorm in training mode (batch statistics; the reference never calls .eval()) + activation (+ skip):
 * y = act(bn(x)) [+ skip]; slope 0 = ReLU.  mean_var_out [2,C] optional. */
int o2345_bn_act_rows(const float* x, int n, int C, const float* gamma, const float* beta, float eps, float slope,
                      int abs_gamma, const float* skip, float* y, float* mean_var_out, void* workspace,
                      size_t workspace_bytes, void* stream);
/* replaces inplace_abn.InPlaceABN forward (featurenet.py:12-22, sparse_sdf_network.py:171-173): batch-stat BN +
 * leaky ReLU on [V,C,H,W], C in {8, 16, 32}; writes NCHW and/or channel-last NHWC. */
size_t o2345_abn_workspace_bytes(int C);
int o2345_abn_nchw(const float* x, int V, int C, int H, int W, const float* gamma, const float* beta, float eps,
                   float slope, int abs_gamma, float* y_nchw, float* y_nhwc, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ---- SDF network (replaces ops/grid_sampler.py:64 grid_sample_3d + models/embedder.py:63 Embedding +
 * sparse_sdf_network.py:35 LatentSDFLayer, :402 sdf(), :476 gradient(), and the per-chunk loop of
 * sparse_neus_renderer.py:881 extract_fields) ------------------------------------------------------------------------
 * blob: o2345_sdf_blob_floats() floats packed by weights.pack_sdf_blob.  vol_cl: [D,D,D,16] channel-last.
 * variant 0: SDF only; 1: all 128 outputs; 2: SDF + d sdf/d x.  Points: pts [P,3], or pts = NULL and grid_R = R for
 * the x-major lattice linspace(-1,1,R)^3.  index / n_dev: optional list of point slots and device-side count.
 * out_sdf[slot] = sign * sdf. */
int o2345_sdf_blob_floats(void);
int o2345_sdf_mlp(int variant, const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index,
                  const int32_t* n_dev, long long n, int grid_R, float sign, float* out_sdf, float* out_feat,
                  float* out_lat, float* out_grad, void* stream);
/* same, with lat_in [P,16]: use the given latent per point instead of sampling the volume (replaces
 * sparse_sdf_network.py:441 get_sdf_volume: SDF at voxel centres with the voxel's own latent); variants 0/1, explicit points. */
int o2345_sdf_mlp_ex(int variant, const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index,
                     const int32_t* n_dev, long long n, int grid_R, float sign, const float* lat_in, float* out_sdf,
                     float* out_feat, float* out_lat, float* out_grad, void* stream);
/* fp32-class accuracy on the f16 matrix cores: every operand split into two f16 halves (hi + lo, 22 bits) and each product
 * accumulated in fp32 as hi*hi + hi*lo + lo*hi -- three v_mfma_f32_32x32x16_f16 instead of eight fp32 MFMAs per 16 k.
 * Same function as o2345_sdf_mlp variant 0 within ~1e-6 (tests/test_gpu_parity.py::test_sdf_mlp_x3). */
int o2345_sdf_mlp_x3(const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index, const int32_t* n_dev,
                     long long n, int grid_R, float sign, float* out_sdf, void* stream);
/* SDF + analytic gradient, wide layers (144->128 and its transpose) split-f16, the 40-wide layer 0 on the exact fp32 MFMA */
int o2345_sdf_grad_x3(const float* blob, const float* vol_cl, int D, const float* pts, const int32_t* index, const int32_t* n_dev,
                      long long n, int grid_R, float sign, float* out_sdf, float* out_grad, void* stream);
/* extract_fields on the lattice with layer 0 of the SDF network TABULATED (ABI 1.3): on the x-major lattice linspace(-1,1,R)^3 the first layer's
 * pre-activation is separable, b0 + W0 . PE(x,y,z) = Txy[ix,iy] + Tz[iz].  tab_axes [3][R][128]: per-axis tables in the kernels' lane order
 * (weights.sdf_grid_tables); sdf_grid_tables adds x + y + bias into tab_xy [R*R][128]; sdf_grid_x3 evaluates out_sdf [R^3] = sign * sdf with
 * tab_xy and tab_z = tab_axes + 2*R*128.  Same network as o2345_sdf_mlp_x3 with pts = NULL, grid_R = R, minus 18 sincos and 36 matrix steps per tile. */
int o2345_sdf_grid_tables(const float* tab_axes, const float* bias_lane_order, int grid_R, float* tab_xy, void* stream);
int o2345_sdf_grid_x3(const float* blob, const float* vol_cl, int D, int grid_R, float sign, const float* tab_xy, const float* tab_z,
                      float* out_sdf, void* stream);

/* ---- ray rendering (replaces models/sparse_neus_renderer.py:457 render and everything it calls) ------------------
 * Per-sample arrays are sample-major [S][R]. */
int o2345_ray_coarse(const float* rays_o, const float* rays_d, int R, float near, float far, int S, float* z, float* pts,
                     void* stream);
/* the same with the reference's stratified jitter (models/sparse_neus_renderer.py:506-515): t_rand [R][S] (ray-major, what
 * torch.rand(z_vals.shape) returns) or NULL */
int o2345_ray_coarse_jitter(const float* rays_o, const float* rays_d, int R, float near, float far, int S,
                            const float* t_rand, float* z, float* pts, void* stream);
/* the same with per-ray near / far [R] (the reference broadcasts near + (far - near) * linspace per ray, :486-490) */
int o2345_ray_coarse_per_ray(const float* rays_o, const float* rays_d, int R, const float* near_ray, const float* far_ray, int S,
                             const float* t_rand, float* z, float* pts, void* stream);
/* up_sample + sample_pdf of one round (:73-115, render_utils.py:8-51): z / sdf [S][R] sorted per ray -> new_z [n_imp][R], their points, new_sdf = 100
 * (the cat_z_vals default), and the list of new points inside the mask (slots t * R + r) with its device-side count.  wbuf: scratch [S][R] floats for
 * the streaming kernel, or NULL: the LDS-staged kernel then runs whatever R is (the two give bit-identical results; csrc/render.hip). */
int o2345_ray_upsample(const float* rays_o, const float* rays_d, int R, const float* z, const float* sdf, int S,
                       float inv_s, const float* maskvol, int D, float* wbuf, int n_imp, float* new_z, float* new_pts,
                       float* new_sdf, int32_t* list, int32_t* count_dev, void* stream);
int o2345_ray_merge(int R, float* z, float* sdf, int S, float* new_z, float* new_sdf, int n_new, void* stream);
/* finalize: mid points, section lengths, occupancy of the mid points and the list of occupied slots; every slot of sdf / grad / rgb receives the
 * reference's defaults (sdf = 100, grad = rgb = 0; models/sparse_neus_renderer.py:231), occupied or not -- a caller may evaluate any part of the list.
 * (o2345_render_rays, which evaluates every list entry, uses an internal variant that skips the occupied slots.) */
int o2345_ray_finalize(const float* rays_o, const float* rays_d, int R, const float* z, int S, float sample_dist,
                       const float* maskvol, int D, float* mid_z, float* dists, float* pts, float* pm, float* sdf,
                       float* grad, float* rgb, int32_t* list, int32_t* count_dev, void* stream);
int o2345_ray_composite(const float* rays_o, const float* rays_d, int R, int S, const float* mid_z, const float* dists,
                        const float* pm, const float* sdf, const float* grad, const float* rgb, const uint8_t* nviews,
                        float inv_s, float alpha_inter_ratio, float background, float* color, float* depth,
                        float* weights, float* cdf, float* weights_sum, float* weights_max, float* depth_var,
                        float* alpha_sum, float* grad_err, uint8_t* color_mask, void* stream);

/* One render() call (models/sparse_neus_renderer.py:457-635).  The struct is declared HERE only: csrc/ includes this header, the ctypes binding
 * generates its Structure from this text and checks it against o2345_render_io_size / o2345_render_io_layout of the loaded library. */
typedef struct O2345RenderIO {
    /* scene */
    const float* sdf_blob;          /* weights.pack_sdf_blob */
    const float* color_x3_blob;     /* split-f16 colour network (weights.pack_color_x3_blob); takes precedence */
    const float* color_mfma_blob;   /* fp32 matrix-core colour network (weights.pack_color_mfma_blob); one of the two is required */
    const float* vol_cl;            /* [D,D,D,16] */
    const float* maskvol;           /* [D^3] */
    const float* cmaps;             /* [V,H,W,64] */
    const float* proj;              /* [V,3,4] */
    const float* cam_pos;           /* [V,3] */
    int D, V, H, W;
    /* rays */
    const float* rays_o; const float* rays_d;
    const float* near_ray;          /* optional [R]: per-ray near / far (the reference's [N_rays,1] form, :486-490); NULL: the scalars below */
    const float* far_ray;
    const float* query_cam;         /* [3] */
    const float* t_rand;            /* optional [R][n_samples]: the reference's perturb > 0 jitter, drawn by the caller */
    int R, n_samples, n_importance;
    int sdf_mode;                   /* 0: exact fp32 MFMA SDF kernels; 2: split-f16 ("f16x3", fp32-class accuracy).  (ABI 1.x called this sdf_bf16.) */
    int segment_rays;               /* 0: the call is ONE render() call of the reference.  > 0 (a multiple of 64): the R rays are consecutive render() calls of
                                     * segment_rays rays each (the last one may be shorter) evaluated together -- the reference's two per-call rules (cat_z_vals' "more
                                     * than one new point inside the mask", :137; render_core's "first 100 points when nothing is occupied", :222-223) are applied
                                     * PER SEGMENT, t_rand holds the segments' draws back to back, and `scalars` becomes [ceil(R / segment_rays)][4] */
    float near, far;                /* used when near_ray == NULL */
    float sample_dist;              /* length of the last section (:484: ((far - near) / n_samples).mean()); <= 0: (far - near) / n_samples */
    float inv_s, alpha_inter_ratio, background;
    float weight_cull;              /* 0: the colour network is evaluated on every occupied sample, like the reference.  > 0: only on occupied samples whose
                                     * compositing weight w = alpha * T (the value `weights` returns, computed first from the SDF / gradient pass) is >= weight_cull;
                                     * the others keep rgb = 0.  Every colour component lies in [0, 1] (a softmax blend of source pixels), so a ray's colour moves
                                     * by at most S * weight_cull: 128 * 2^-24 = 7.6e-6 at the default 2^-24, below the 3e-5 stage tolerance.  depth, weights,
                                     * gradients and the colour mask (valid-view counts of culled points come from the counting kernel) are unaffected.  Only samples
                                     * BEHIND a surface qualify: in free space the reference's +1e-5 in alpha keeps w ~ 1e-5 (:349-375), those are never dropped. */
    /* outputs: per-sample arrays are sample-major [S][R] (+[,3]) */
    float* mid_z; float* dists; float* pm; float* sdf; float* grad; float* rgb; uint8_t* nviews;
    float* color; float* depth; float* weights; float* cdf; float* weights_sum; float* weights_max; float* depth_var;
    float* alpha_sum; float* grad_err; uint8_t* color_mask;
    float* z_vals;                  /* optional [S][R] */
    float* scalars;                 /* optional [4]: alpha_sum.mean(), alpha_sum.sum() / (R S), sum grad_err[.,0] / (sum grad_err[.,1] + 1e-5), number of
                                     * evaluated list entries (fixed-order fp64 reduction, deterministic): the scalar entries of render()'s returned
                                     * dict (alpha_sum, alpha_mean, gradient_error_fine; :586-633) without a pass over the per-ray arrays */
    unsigned long long* color_stats;/* optional [4] device counters the colour kernel ADDS to (diagnostics, caller-owned and caller-zeroed): (32-point tile,
                                     * view) pairs evaluated in the pooling pass / the network pass, tiles, tiles that evaluated every view */
} O2345RenderIO;
/* The occupied-point list of a render call grouped, stably, by view-visibility signature (bit v = the point projects inside view v): every 32-point tile of the
 * colour kernels then holds points that see the same views, and the kernels skip the views nobody sees.  The network kernels scatter by slot, so their
 * results do not depend on the order (no reference counterpart: the reference evaluates every (point, view) pair, models/projector.py:96-228).
 * list [<= n_max] slots, *count_dev entries (device-side count); list_out != list; keys_out (optional) receives the signatures in output order. */
size_t o2345_list_sort_workspace_bytes(long long n_max, int V);
int o2345_list_sort_by_visibility(const float* pts, const int32_t* list, const int32_t* count_dev, long long n_max, const float* proj, int V, int H, int W,
                                  int32_t* list_out, uint32_t* keys_out, void* workspace, size_t workspace_bytes, void* stream);
/* V: the scene's view count -- the list-sort buffers are part of the workspace only when the call will sort its list (V <= 32 and at least 2^20 sample slots) */
size_t o2345_render_workspace_bytes(int R, int n_samples, int n_importance, int V);
int o2345_render_rays(const O2345RenderIO* io, void* workspace, size_t workspace_bytes, void* stream);

/* proj [V,3,4] = intrinsics [V,3,3] @ w2cs[:, :3, :] (models/render_utils.py:106) and cam_pos [V,3] = inverse(w2cs)[:, :3, 3] (models/projector.py:60-70)
 * in ONE launch, no BLAS / solver library behind it (the first torch.matmul + torch.inverse of a process cost 180 ms of library initialisation
 * inside the reference's own timing bracket).  Products are fp32 FMA chains in k order (what rocBLAS / ATen evaluate for k = 3); the inverse is a
 * general 4x4 inverse by cofactors in fp64, rounded once to fp32 (singular w2c: cam_pos = inf / nan like torch.inverse's). */
int o2345_camera_terms(const float* intrinsics, const float* w2cs, int V, float* proj_out, float* cam_pos_out, void* stream);

/* ---- colour blending (replaces models/projector.py:96 Projector.compute / :231 compute_view_independent +
 * models/rendering_network.py:75 GeneralRenderingNetwork.forward) ---------------------------------------------------
 * cmaps [V,H,W,64] = rgb(3) | features(56) | pad; proj [V,3,4] = K @ w2c[:3]; cam_pos [V,3]. */
int o2345_pack_color_maps(const float* feat_nchw, const float* color_nchw, int V, int H, int W, float* out_nhwc64,
                          void* stream);
int o2345_view_count(const float* pts, long long n, const float* maskvol, int D, const float* proj, int V, int H, int W,
                     uint8_t* out, void* stream);
/* the same count only where skip_if_positive[i] <= 0 (NULL: everywhere): the colour kernels write out_nviews for the points they evaluate */
int o2345_view_count_unlisted(const float* pts, long long n, const float* skip_if_positive, const float* maskvol, int D, const float* proj,
                              int V, int H, int W, uint8_t* out, void* stream);
/* Projector + GeneralRenderingNetwork fused, every linear layer on fp32 MFMA (k_color_pts, csrc/color_pts.hip: a wave owns 32 points and walks the
 * views; any V in [1,255]); blob from weights.pack_color_mfma_blob.  pts [P,3]; index / n_dev: optional list of point slots + device-side count;
 * exactly one of query_cam [3] (Projector.compute) / normals [P,3] (compute_view_independent).  out_rgb [P,3], out_nviews [P] (optional).
 * stats_dev (optional, [4], caller-owned and caller-zeroed): work counters the launch ADDS to -- (32-point tile, view) pairs evaluated in the
 * pooling pass / the network pass, tiles, tiles that evaluated every view.  (ABI 1.x kept these in a process-global buffer; the pure-VALU
 * kernel o2345_color_points of ABI 1.x is gone: 206 ms against 37.) */
int o2345_color_mfma_blob_floats(void);
int o2345_color_points_mfma(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                            const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                            const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                            const float* normals, float* out_rgb, uint8_t* out_nviews, unsigned long long* stats_dev, void* stream);
/* same kernel, split-f16 matrix steps (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate; fp32-class
 * accuracy, see o2345_sdf_mlp_x3); blob from weights.pack_color_x3_blob */
int o2345_color_x3_blob_floats(void);
int o2345_color_points_x3(const float* blob, const float* vol_cl, const float* maskvol, int D, const float* cmaps,
                          const float* proj, const float* cam_pos, int V, int H, int W, const float* pts,
                          const int32_t* index, const int32_t* n_dev, long long n, const float* query_cam,
                          const float* normals, float* out_rgb, uint8_t* out_nviews, unsigned long long* stats_dev, void* stream);
/* GeneralRenderingNetwork.forward on MATERIALISED inputs in the reference's own (view-major) layout (models/rendering_network.py:75-129):
 * geometry_feat [P,16], rgb_feat [V,P,59] (colours | features), ray_diff [V,P,4], mask [V,P] (non-zero = valid) -> rgb [P,3] and the number of
 * valid views [P] (optional).  blob: the x3 (x3 = 1) or fp32-MFMA (x3 = 0) packing of the network.  The fused Projector path above is the fast
 * one; this entry makes the network a drop-in on its own (any Projector). */
/* Projector.compute (query_cam) / compute_view_independent (normals) MATERIALISED (models/projector.py:96-425): the four tensors of
 * GeneralRenderingNetwork.forward in the reference's layout -- geometry_feat [P,16], rgb_feat [V,P,59], ray_diff [V,P,4], mask [V,P] (1 / 0).
 * For callers that want the tensors themselves (a foreign rendering network); the fused o2345_color_points_* never store them. */
int o2345_project_features(const float* vol_cl, const float* maskvol, int D, const float* cmaps, const float* proj, const float* cam_pos, int V,
                           int H, int W, const float* pts, long long P, const float* query_cam, const float* normals, float* geometry_feat,
                           float* rgb_feat, float* ray_diff, float* mask, void* stream);
int o2345_color_from_features(const float* blob, int x3, const float* geometry_feat, const float* rgb_feat, const float* ray_diff,
                              const float* mask, int V, long long P, float* out_rgb, uint8_t* out_nviews, void* stream);

/* ---- marching cubes (replaces mcubes.marching_cubes, call site models/sparse_neus_renderer.py:932) -----------------
 * u [n0,n1,n2] float32 on the device; iso is a DOUBLE like PyMCubes' isovalue (ABI 1.2).  count() synchronises the stream and returns the sizes on the host;
 * emit() writes verts float64 [nv,3] (index coordinates) and tris int32/int64 [nt,3]. */
size_t o2345_mc_workspace_bytes(int n0, int n1, int n2);
int o2345_marching_cubes_count(const float* u, int n0, int n1, int n2, double iso, void* workspace,
                               size_t workspace_bytes, long long* nv_host, long long* nt_host, void* stream);
int o2345_marching_cubes_emit(const float* u, int n0, int n1, int n2, double iso, void* workspace, double* verts,
                              void* tris, int index_bytes, void* stream);

/* ---- mesh serialisation (replaces the numpy / trimesh tail of validate_mesh and validate_colored_mesh,
 * models/trainer_generic.py:1287-1303, 1365-1382: index -> world frame, scale_mat, trans_mat, uint8 colours, PLY records) -----
 * verts_idx: device fp64 [n,3] index coordinates on an R^3 grid (o2345_marching_cubes_emit); bound_min/max [3], scale_mat and
 * trans_mat (4x4 row-major fp32, may be NULL) are HOST arrays; rgb device fp32 [n,3] or NULL.
 * vertex_records: n x (3 x float32 [+ rgba uint8]) = 16 (12 without colours) bytes each; face_records: m x (uint8 3, 3 x int32). */
int o2345_mesh_pack_vertices(const double* verts_idx, long long n, int grid_R, const float* bound_min, const float* bound_max,
                             const float* scale_mat, const float* trans_mat, const float* rgb, uint8_t* vertex_records, void* stream);
int o2345_mesh_pack_faces(const void* tris, int index_bytes, long long m, uint8_t* face_records, void* stream);
/* extract_geometry's index -> world step on the device, in place and in fp64 (sparse_neus_renderer.py:936: vertices / (R - 1) * (bound_max - bound_min)
 * + bound_min; the same IEEE expression numpy evaluates): verts [n,3] device fp64; extent = bound_max - bound_min (the reference subtracts the fp32 bounds,
 * THEN promotes) and offset = bound_min as HOST fp64 [3]. */
int o2345_mc_verts_to_world(double* verts, long long n, int grid_R, const double* extent_host, const double* offset_host, void* stream);
/* The PLY records of a mesh that is already on the HOST (trimesh.Trimesh(vertices, faces, vertex_colors).export(), trainer_generic.py:1302-1303,
 * 1377-1382): vertices fp64 [n,3], colors uint8 [n,3|4] or NULL, faces int64 [m,3], all host pointers -> vertex_records n x (16 | 12) bytes,
 * face_records m x 13 bytes.  Plain host code (up to four threads), no device work, no stream. */
int o2345_ply_records_host(const double* vertices, long long n, const uint8_t* colors, int color_channels, const long long* faces, long long m,
                           uint8_t* vertex_records, uint8_t* face_records);
/* Load every code object of the library on the current device now (the HIP runtime loads a translation unit's kernels at its first launch: 5 - 60 ms
 * each for the 15 units of this library, which otherwise land inside the first calls of a fresh process).  Launches nothing that touches user memory. */
int o2345_preload(void);

#ifdef __cplusplus
}
#endif
#endif
