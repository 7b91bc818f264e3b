// Hierarchical ray sampling + NeuS compositing: SparseNeuSRenderer.render
// (models/sparse_neus_renderer.py:457-635) with up_sample (:73-115), cat_z_vals (:117-151), render_core (:171-455).
//
// The reference runs ~40 small kernels per up-sampling round per 512-ray chunk.  Here one lane owns one ray; all
// per-ray lists are sample-major ([S][R]) so every access of the wave is a coalesced row segment; the SDF network
// is called on flat point lists (occupied points compacted with a ballot/popcount prefix into an index list that
// the MFMA kernel consumes with a device-side count -- no host synchronisation anywhere in a render call).
#include "common.h"                    // pulls in include/o2345.h: O2345RenderIO is declared THERE only (layout-checked by the binding at load time)
#include "render_math.h"

namespace o2345 {

// coarse samples: z = near + (far-near) * linspace(0,1,S)  and their points, point index p = s*R + r
// t_rand (optional): the reference's stratified jitter (sparse_neus_renderer.py:506-515).  The reference draws
// t_rand = torch.rand(z_vals.shape) on the HOST ([R][S], ray-major) and sets z = lower + (upper - lower) * t_rand with
// lower/upper the midpoints to the neighbouring coarse samples; the caller hands the same tensor over, so the path is
// bit-reproducible under torch.manual_seed.  msk (optional, [S][R] bytes): 1 where the sample point lies in an occupied voxel of the mask volume
// (what up_sample asks of every sample, :84-88) -- kept next to z / sdf from here on, so that no later kernel has to gather it again.
__global__ __launch_bounds__(256) void k_ray_coarse(RayGeom g, float near, float far, const float* __restrict__ near_ray,
                                                    const float* __restrict__ far_ray, int S, const float* __restrict__ t_rand,
                                                    float* __restrict__ z, float* __restrict__ pts, const float* __restrict__ maskvol, int D,
                                                    uint8_t* __restrict__ msk, int* __restrict__ zero16) {
    // o2345_render_rays: the first kernel of the call also clears the call's 16 device-side counters (they are first touched by a later launch)
    if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long long)S * g.R) return;
    const int s = (int)(p / g.R), r = (int)(p % g.R);
    if (near_ray) { near = near_ray[r]; far = far_ray[r]; }      // the reference's [N_rays, 1] near / far (:486-490): the same expression per ray
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
    if (msk) msk[p] = mask_at(maskvol, D, x, y, w) > 0.f ? 1 : 0;
}

// Occupied points are appended to a global list (order inside the list is irrelevant for the results: they are scattered back
// by slot).  One lane owns one ray and walks its samples, so a wave first collects the validity of all (sample, ray) pairs it
// owns as bit masks, reserves its whole range with ONE atomic, and then writes its slots sample by sample (ballot + popcount
// prefix): no block barrier and 1/S of the atomics of a per-sample reservation.
struct ValidBits { unsigned w[8]; };                 // up to 256 samples per ray
__device__ __forceinline__ void append_wave(const ValidBits& bits, int n_samples, int my_count, int R, int r,
                                            int* __restrict__ list, int* __restrict__ count, int* __restrict__ seg_cnt = nullptr, int seg_rays = 0) {
    int total = my_count;
#pragma unroll
    for (int off = 32; off; off >>= 1) total += __shfl_xor(total, off);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && total) {
        base = atomicAdd(count, total);
        if (seg_cnt) atomicAdd(seg_cnt + r / seg_rays, total);     // a wave's 64 consecutive rays lie in ONE segment (seg_rays is a multiple of 64); lane 0's ray is live when total > 0
    }
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

// cat_z_vals quirk (:137): the SDF of the new points is evaluated only if MORE THAN ONE of them is inside the mask
// (o2345_render_rays applies it inside the round kernel: round_epilogue; this launch follows the stage entry o2345_ray_upsample)
__global__ void k_quirk_min2(int* count) { if (*count <= 1) *count = 0; }

// ---- one round of the hierarchical sampler (up_sample + sample_pdf, preceded by the cat_z_vals of the previous round; or that merge + render_core's head)
// Round 3 ran three kernels per round (up-sample, merge, and the SDF network between them) that walked a ray's samples straight from the sample-major
// global lists, one lane per ray: every step of the serial chains (transmittance, CDF walk, back-to-front merge) was a DEPENDENT trip to L2 / HBM, 60 - 110
// of them per ray and launch -- 2.9 ms per 262,144 rays, and 81 / 40 / 95 us per launch for one 512-ray chunk of the reference's val loop.  Round 4:
//   * cat_z_vals of the PREVIOUS round (its 16 new samples, now with their SDF values) is fused into the next kernel as a rank merge, and the last
//     merge into render_core's head (mid points, section lengths, occupancy, defaults, occupied-point list): four launches and two passes over the lists less;
//   * the occupancy flag of every sample point travels with the list (one byte per sample, written where the point is produced) instead of being
//     gathered from the mask volume again in every round;
//   * two forms of the same round, selected by the batch size (knobs().ray_stream_min), sharing render_math.h and bit-identical to each other:
//     k_ray_stream (large batches: one lane per ray, static access pattern, blocks of 8 rows in flight, full occupancy -> HBM-bound) and
//     k_ray_group (small batches: sixteen lanes per ray, only the two scans serial).
constexpr int NFIX = 16;                             // new samples per round held in registers (n_importance / 4 of the released configuration)
constexpr int RM_UPSAMPLE = 0, RM_FINALIZE = 1, RM_MERGE_ONLY = 2;

struct RoundArgs {
    RayGeom g;
    float* z; float* sdf; uint8_t* msk;              // the lists [rows][R] (msk may be null: occupancy is then gathered while staging)
    int S;                                           // samples per ray in the lists (before the merge)
    const float* new_z; const float* new_sdf; const uint8_t* new_msk; int n_new;      // previous round's samples to merge first (n_new = 0: none)
    const float* maskvol; int D;
    // upsample
    float inv_s; int n_imp;
    float* out_z; float* out_pts; float* out_sdf; uint8_t* out_msk; int* list; int* count;
    int* done;                                       // optional zeroed counter: the round applies the reference's per-call quirk itself (round_epilogue)
    // finalize
    float sample_dist;
    float* mid_z; float* dists; float* pts; float* pm; float* o_sdf; float* grad; float* rgb; int defaults_everywhere;
    int merge_all_lists;                             // finalize + merge: also write the merged sdf / occupancy lists back (nobody reads them after the last round)
    // O2345RenderIO.segment_rays: the rays are consecutive render() calls of seg_rays rays evaluated together; the two per-call rules per segment
    int seg_rays, n_seg;                             // 0: the launch is one call
    int* seg_cnt;                                    // [n_seg] zeroed: list entries this round appends, per segment
    const int* seg_prev;                             // [n_seg] the PREVIOUS up-sampling round's counts: its new samples keep sdf = 100 where <= 1 (cat_z_vals :137)
};
// cat_z_vals' rule in segment mode: the SDF kernel has evaluated every listed new sample (at most ONE too many per segment); the merge that consumes
// them reads 100 instead wherever the sample's segment had at most one new sample inside the mask -- the value the reference leaves there.
__device__ __forceinline__ bool seg_keeps_default(const RoundArgs& a, int r) { return a.seg_prev && a.seg_prev[r / a.seg_rays] <= 1; }

// The reference's two per-CALL rules on the emitted list, applied by the round kernel itself instead of a one-thread launch each (six launches of the
// 22 of a 512-ray chunk): the workgroup that finishes LAST (a.done counts finished workgroups) sees the final count and
//   up-sampling round  cat_z_vals (:137): the SDF of the new samples is evaluated only if MORE THAN ONE of them is inside the mask -> count <= 1 becomes 0;
//   finalize           render_core (:222-223): with no occupied point at all the first 100 points of the chunk (ray 0, samples 0..99, in the
//                      reference's ray-major order) are evaluated anyway.
// Without a.done (the stage entry points) the launches k_quirk_min2 / nothing follow as before.
// Ordering without fences: `count` and `done` are only ever touched by agent-scope atomic read-modify-writes, which are performed at the device's
// coherent point; a workgroup's appends have RETURNED (their result is consumed) before its barrier, its increment of `done` is issued after the barrier,
// and the last workgroup reads `count` (an atomic again) after its own increment has returned -- so every append precedes that read.  An agent-scope
// fence (__threadfence) here is an L2 write-back of everything the kernel has written on this multi-XCD part: measured + 31 / 40 / 72 us per launch
// at 262,144 rays (profiles/NOTES.md), for an ordering the atomics already have.
template <int MODE>
__device__ __forceinline__ void round_epilogue(const RoundArgs& a, int S) {
    if (!a.done || MODE == RM_MERGE_ONLY) return;    // kernel argument: uniform over the launch
    if (a.seg_rays && MODE == RM_UPSAMPLE) return;   // segment mode: the merge that consumes the samples applies the rule (seg_keeps_default)
    __shared__ int last;
    __syncthreads();                                 // every append of this workgroup has returned its base
    if (threadIdx.x == 0) last = atomicAdd(a.done, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    // (the reference sets pts_mask_bool[:100] on the RAY-MAJOR flattened [N_rays * S] mask: point t is (ray t / S, sample t % S) -- with S < 100 the rule
    //  runs on into the next rays; slots here are sample-major, s * R + r)
    if (a.seg_rays) {                                // render_core's rule per segment: a segment without any occupied point evaluates its first 100 points
        for (int k = threadIdx.x; k < a.n_seg; k += blockDim.x) {
            if (atomicAdd(a.seg_cnt + k, 0) >= 1) continue;
            const int r0 = k * a.seg_rays;
            const int rays = (a.g.R - r0) < a.seg_rays ? (a.g.R - r0) : a.seg_rays;
            const long long avail = (long long)rays * S;
            const int n = avail < 100 ? (int)avail : 100;
            const int base = atomicAdd(a.count, n);
            for (int t = 0; t < n; ++t) a.list[base + t] = (t % S) * a.g.R + r0 + t / S;
            atomicExch(a.seg_cnt + k, n);             // the segment's "evaluated points" (scalars[3])
        }
        return;
    }
    const int c = atomicAdd(a.count, 0);
    if (MODE == RM_UPSAMPLE) {
        if (threadIdx.x == 0 && c <= 1) atomicExch(a.count, 0);
    } else if (c < 1) {
        const long long avail = (long long)a.g.R * S;
        const int n = avail < 100 ? (int)avail : 100;
        for (int t = threadIdx.x; t < n; t += blockDim.x) a.list[t] = (t % S) * a.g.R + t / S;     // slot of (ray t / S, sample t % S)
        if (threadIdx.x == 0) atomicExch(a.count, n);
    }
}

// ---- the same round as a STREAMING kernel: one lane per ray, the lists read from global memory ------------------------------------------------
// Nothing is staged (an LDS-staged form -- 64 rays x 128 rows x 9 bytes per wave -- was measured this round: two waves per CU, 0.82 ms per round at
// 262,144 rays against 0.36 here): render_math.h's passes touch the lists at static, ascending rows in blocks of 8, so every block is one trip with 24
// loads in flight, eight waves per SIMD hide those trips, and a round is bound by its HBM traffic.  The section weights go through a global scratch
// row set (wbuf); in-place rank merge on the global lists (writes of a block of rows never reach a row not yet read).
struct StreamRay {                                   // accessor of upsample_core on the global lists
    const float* z_; const float* sdf_; const uint8_t* m_; float* w_; float* out_;
    size_t R; int r; RayGeom g; const float* maskvol; int D;
    __device__ __forceinline__ float z(int s) const { return z_[(size_t)s * R + r]; }
    __device__ __forceinline__ float sdf(int s) const { return sdf_[(size_t)s * R + r]; }
    __device__ __forceinline__ float msk(int s, float zs) const {
        if (m_) return (float)m_[(size_t)s * R + r];
        float x, y, w;
        ray_point(g, r, zs, x, y, w);
        return mask_at(maskvol, D, x, y, w) > 0.f ? 1.f : 0.f;
    }
    __device__ __forceinline__ void set_w(int s, float v) { w_[(size_t)s * R + r] = v; }
    __device__ __forceinline__ float w(int s) const { return w_[(size_t)s * R + r]; }
    __device__ __forceinline__ void out(int t, float v) { out_[(size_t)t * R + r] = v; }
};
struct GlobalMergeTag {
    float* z_; float* sdf_; uint8_t* m_; size_t R; int r;
    __device__ __forceinline__ float z(int i) const { return z_[(size_t)i * R + r]; }
    __device__ __forceinline__ float sdf(int i) const { return sdf_[(size_t)i * R + r]; }
    __device__ __forceinline__ unsigned tag(int i) const { return m_ ? m_[(size_t)i * R + r] : 0u; }
    __device__ __forceinline__ void put(int i, float zv, float sv, unsigned t) {
        z_[(size_t)i * R + r] = zv; sdf_[(size_t)i * R + r] = sv;
        if (m_) m_[(size_t)i * R + r] = (uint8_t)t;
    }
};
struct GlobalMergeZ {                               // the LAST cat_z_vals of a render call: only the depths are read again (render_core re-evaluates the SDF at the mid points)
    float* z_; size_t R; int r;
    __device__ __forceinline__ float z(int i) const { return z_[(size_t)i * R + r]; }
    __device__ __forceinline__ float sdf(int) const { return 0.f; }
    __device__ __forceinline__ unsigned tag(int) const { return 0u; }
    __device__ __forceinline__ void put(int i, float zv, float, unsigned) { z_[(size_t)i * R + r] = zv; }
};

template <int MODE, bool MERGE>
__global__ __launch_bounds__(256) void k_ray_stream(RoundArgs a, float* __restrict__ wbuf) {
    const int R = a.g.R;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool live = r < R;
    const int rr = live ? r : R - 1;
    int S = a.S;
    if (MERGE) {
        float nz[NFIX], ns[NFIX];
        unsigned nt[NFIX];
        const bool z_only = MODE == RM_FINALIZE && !a.merge_all_lists;
#pragma unroll
        for (int j = 0; j < NFIX; ++j) {
            nz[j] = a.new_z[(size_t)j * R + rr];
            ns[j] = z_only ? 0.f : a.new_sdf[(size_t)j * R + rr];
            nt[j] = (a.new_msk && !z_only) ? a.new_msk[(size_t)j * R + rr] : 0u;
        }
        if (!z_only && seg_keeps_default(a, rr)) {
#pragma unroll
            for (int j = 0; j < NFIX; ++j) ns[j] = 100.f;
        }
        if (MODE == RM_UPSAMPLE && a.msk && !a.new_msk) {
#pragma unroll
            for (int j = 0; j < NFIX; ++j) {
                float x, y, w;
                ray_point(a.g, rr, nz[j], x, y, w);
                nt[j] = mask_at(a.maskvol, a.D, x, y, w) > 0.f ? 1u : 0u;
            }
        }
        bool sorted = true;
#pragma unroll
        for (int j = 0; j + 1 < NFIX; ++j) sorted = sorted && !(nz[j] > nz[j + 1]);
        if (live) {                                 // in-place on the global lists: lanes past the last ray must not write
            if (z_only) {
                GlobalMergeZ m{a.z, (size_t)R, r};
                merge_core_fixed<GlobalMergeZ, NFIX>(m, S, nz, ns, nt, sorted);
            } else {
                GlobalMergeTag m{a.z, a.sdf, a.msk, (size_t)R, r};
                merge_core_fixed<GlobalMergeTag, NFIX>(m, S, nz, ns, nt, sorted);
            }
        }
        S += NFIX;
    }
    if (MODE == RM_UPSAMPLE) {
        if (live) {
            StreamRay acc{a.z, a.sdf, a.msk, wbuf, a.out_z, (size_t)R, r, a.g, a.maskvol, a.D};
            upsample_core(acc, S, a.inv_s, a.n_imp);
        }
        ValidBits bits{};
        int cnt = 0;
        for (int t0 = 0; t0 < a.n_imp; t0 += 8) {
            float m8[8], z8[8], p8[8][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) z8[k] = a.out_z[(size_t)(t0 + k < a.n_imp ? t0 + k : a.n_imp - 1) * R + rr];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ray_point(a.g, rr, z8[k], p8[k][0], p8[k][1], p8[k][2]);
                m8[k] = mask_at(a.maskvol, a.D, p8[k][0], p8[k][1], p8[k][2]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = t0 + k;
                if (live && t < a.n_imp) {
                    const size_t slot = (size_t)t * R + r;
                    a.out_pts[3 * slot] = p8[k][0]; a.out_pts[3 * slot + 1] = p8[k][1]; a.out_pts[3 * slot + 2] = p8[k][2];
                    a.out_sdf[slot] = 100.f;                             // cat_z_vals default outside the mask (:135)
                    const bool in = m8[k] > 0.f;
                    if (a.out_msk) a.out_msk[slot] = in ? 1 : 0;
                    if (in) { bits.w[t >> 5] |= 1u << (t & 31); ++cnt; }
                }
            }
        }
        append_wave(bits, a.n_imp, cnt, R, r, a.list, a.count, a.seg_cnt, a.seg_rays);
    } else if (MODE == RM_FINALIZE) {
        ValidBits bits{};
        int cnt = 0;
        float znext = a.z[rr];
        for (int s0 = 0; s0 < S; s0 += 8) {
            float zc8[9], m8[8], d8[8], z8[8], p8[8][3];
            zc8[0] = znext;
#pragma unroll
            for (int k = 1; k <= 8; ++k) zc8[k] = a.z[(size_t)(s0 + k < S ? s0 + k : S - 1) * R + rr];
            znext = zc8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = s0 + k;
                const float d = (s + 1 < S) ? zc8[k + 1] - zc8[k] : a.sample_dist;
                const float mz = zc8[k] + d * 0.5f;
                ray_point(a.g, rr, mz, p8[k][0], p8[k][1], p8[k][2]);
                m8[k] = mask_at(a.maskvol, a.D, p8[k][0], p8[k][1], p8[k][2]);
                d8[k] = d; z8[k] = mz;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int s = s0 + k;
                if (live && s < S) {
                    const size_t p = (size_t)s * R + r;
                    const float m = m8[k];
                    a.dists[p] = d8[k]; a.mid_z[p] = z8[k]; a.pm[p] = m;
                    a.pts[3 * p] = p8[k][0]; a.pts[3 * p + 1] = p8[k][1]; a.pts[3 * p + 2] = p8[k][2];
                    if (m > 0.f) { bits.w[s >> 5] |= 1u << (s & 31); ++cnt; }
                    if (!(m > 0.f) || a.defaults_everywhere) {
                        // the reference's defaults (:231: sdf = 100, gradients = colours = 0).  Inside o2345_render_rays occupied points are ALWAYS overwritten by the
                        // network kernels that consume the list (every list entry is evaluated), so only unoccupied points need them there: 28 bytes less per
                        // occupied point.  The public stage entry initialises every slot (a caller may evaluate only part of the list).
                        a.o_sdf[p] = 100.f;
                        a.grad[3 * p] = 0.f; a.grad[3 * p + 1] = 0.f; a.grad[3 * p + 2] = 0.f;
                        a.rgb[3 * p] = 0.f; a.rgb[3 * p + 1] = 0.f; a.rgb[3 * p + 2] = 0.f;
                    }
                }
            }
        }
        append_wave(bits, S, cnt, R, r, a.list, a.count, a.seg_cnt, a.seg_rays);
    }
    round_epilogue<MODE>(a, S);
}

// ---- the same round for SMALL batches: sixteen lanes per ray ---------------------------------------------------------------------------------------
// A 512-ray chunk of the reference's val loop (trainer_generic.py:415-416) is 8 waves with one lane per ray: every kernel of the round is one wave's
// ~20 k dependent instructions long whatever the GPU could do in parallel (50 - 130 us per launch with either form above, x 6 launches x 128 chunks per
// image).  Only TWO things in a round are inherently serial per ray: the running transmittance / weight sum (pass 1) and the running CDF (pass 2) --
// two scans of ~110 two-operation steps whose order is the reference's order and is kept.  Everything else is element-wise per section / per sample and
// is spread over the 16 lanes of the ray's group (4 rays per wave, the lists of the 4 rays in LDS, ray-contiguous so that lane l reads row l + 16 i):
//   merge   rank merge: each existing sample finds #new < it (16 compares against the new block), each new sample its rank in the block and #old <= it
//           (binary search); out of place, in LDS;
//   pass 1  alpha_s of every section in parallel (upsample_section_alpha; a section recomputes its predecessor's slope) -> lane 0 runs the T / w / sum scan;
//   pass 2  pdf = w / sum in parallel -> lane 0 accumulates the CDF in order -> every new sample t does its own searchsorted (binary search) and
//           interpolation: identical to the forward walk, which stops at the first k with cdf_k > u_t as well;
//   points, occupancy and the list of the new samples in parallel (one ballot per wave).
// Same arithmetic per element, same order in the two scans => bit-identical to the other two forms (tests/test_gpu_parity.py).
constexpr int GL = 16, GR = 4;                       // lanes per ray, rays per wave
constexpr int GROUP_ARRAYS = 5;
__host__ __device__ constexpr size_t group_lds_bytes(int rows) { return (size_t)GROUP_ARRAYS * GR * (rows + 1) * sizeof(float); }

template <int MODE, bool MERGE>
__global__ __launch_bounds__(64) void k_ray_group(RoundArgs a) {
    extern __shared__ float lds[];
    const int R = a.g.R, lane = threadIdx.x, sub = lane >> 4, l = lane & 15;
    const int r = blockIdx.x * GR + sub;
    const bool live = r < R;
    const int rr = live ? r : R - 1;
    int S = a.S;
    const int cap = S + (MERGE ? NFIX : 0) + 1;
    float* zM = lds + (size_t)(0 * GR + sub) * cap;   // the (merged) lists of this ray
    float* sM = lds + (size_t)(1 * GR + sub) * cap;
    float* mM = lds + (size_t)(2 * GR + sub) * cap;
    float* xA = lds + (size_t)(3 * GR + sub) * cap;   // new block (merge) -> alpha -> w -> pdf
    float* cA = lds + (size_t)(4 * GR + sub) * cap;   // old depths (merge) -> cdf
    const bool need_mask = MODE == RM_UPSAMPLE;
    auto mask_of = [&](float zs) {
        float x, y, w;
        ray_point(a.g, rr, zs, x, y, w);
        return mask_at(a.maskvol, a.D, x, y, w) > 0.f ? 1.f : 0.f;
    };
    if (!MERGE) {
        for (int i = l; i < S; i += GL) {
            const float zi = a.z[(size_t)i * R + rr];
            zM[i] = zi;
            sM[i] = a.sdf[(size_t)i * R + rr];
            if (need_mask) mM[i] = a.msk ? (float)a.msk[(size_t)i * R + rr] : mask_of(zi);
        }
        __syncthreads();
    } else {
        // new block: lane j holds new sample j
        const float nzj = a.new_z[(size_t)l * R + rr];
        const float nsj = (!(MODE == RM_FINALIZE && !a.merge_all_lists) && seg_keeps_default(a, rr)) ? 100.f : a.new_sdf[(size_t)l * R + rr];
        float ntj = 0.f;
        if (need_mask || a.msk) ntj = a.new_msk ? (float)a.new_msk[(size_t)l * R + rr] : (need_mask ? mask_of(nzj) : 0.f);
        xA[l] = nzj;
        for (int i = l; i < S; i += GL) cA[i] = a.z[(size_t)i * R + rr];
        __syncthreads();
        float nzv[NFIX];
#pragma unroll
        for (int k = 0; k < NFIX; ++k) nzv[k] = xA[k];
        int rank = 0;
#pragma unroll
        for (int k = 0; k < NFIX; ++k) rank += (nzv[k] < nzj || (nzv[k] == nzj && k < l)) ? 1 : 0;
        int lo = 0, hi = S;                                     // #old <= nzj (the old list is sorted)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cA[mid] <= nzj) lo = mid + 1; else hi = mid;
        }
        zM[rank + lo] = nzj; sM[rank + lo] = nsj; mM[rank + lo] = ntj;
        for (int i = l; i < S; i += GL) {
            const float zi = cA[i];
            int c = 0;
#pragma unroll
            for (int k = 0; k < NFIX; ++k) c += nzv[k] < zi ? 1 : 0;
            zM[i + c] = zi;
            sM[i + c] = a.sdf[(size_t)i * R + rr];
            if (need_mask || a.msk) mM[i + c] = a.msk ? (float)a.msk[(size_t)i * R + rr] : mask_of(zi);
        }
        __syncthreads();
        S += NFIX;
        if (live) {
            const bool z_only = MODE == RM_FINALIZE && !a.merge_all_lists;
            for (int i = l; i < S; i += GL) {
                a.z[(size_t)i * R + r] = zM[i];
                if (!z_only) {
                    a.sdf[(size_t)i * R + r] = sM[i];
                    if (a.msk) a.msk[(size_t)i * R + r] = (uint8_t)mM[i];
                }
            }
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (MODE == RM_UPSAMPLE) {
        // pass 1: section opacities in parallel, then the transmittance / weight scan in the reference's order
        for (int sct = l; sct + 1 < S; sct += GL) {
            const float z0 = zM[sct], s0 = sM[sct], z1 = zM[sct + 1], s1 = sM[sct + 1];
            const float prev_dot = sct > 0 ? (s0 - sM[sct - 1]) / (z0 - zM[sct - 1] + 1e-5f) : 0.f;      // the previous section's dot_raw, recomputed
            float dot_raw;
            xA[sct] = upsample_section_alpha(z0, s0, mM[sct], z1, s1, mM[sct + 1], prev_dot, a.inv_s, dot_raw);
        }
        __syncthreads();
        float wsum = 0.f;
        if (l == 0) {
            float T = 1.f;
            for (int sct = 0; sct + 1 < S; ++sct) {
                const float alpha = xA[sct];
                const float w = alpha * T + 1e-5f;
                T = T * (1.f - alpha + 1e-7f);
                xA[sct] = w;
                wsum += w;
            }
        }
        wsum = __shfl(wsum, sub * GL);
        __syncthreads();
        // pass 2: pdf in parallel, CDF in order, then every new sample searches for itself
        for (int k = l; k + 1 < S; k += GL) xA[k] = xA[k] / wsum;
        __syncthreads();
        if (l == 0) {
            float c = 0.f;
            cA[0] = 0.f;
            for (int k = 1; k < S; ++k) { c = c + xA[k - 1]; cA[k] = c; }
        }
        __syncthreads();
        for (int t0 = 0; t0 < a.n_imp; t0 += GL) {
            const int t = t0 + l;
            const bool act = t < a.n_imp;
            float zn = 0.f, x = 0.f, y = 0.f, w = 0.f;
            bool in = false;
            if (act) {
                const float u = linspace_at(0.5f / (float)a.n_imp, 1.f - 0.5f / (float)a.n_imp, a.n_imp, t);
                int lo = 0, hi = S;                                 // searchsorted(right = True): first k with cdf[k] > u (k >= 1 since cdf[0] = 0 < u), else S
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cA[mid] > u) hi = mid; else lo = mid + 1;
                }
                const int k = lo;
                const float cb = cA[k - 1];
                const float ca = (k == S) ? cb : cA[k];
                float den = ca - cb;
                if (den < 1e-5f) den = 1.f;
                const float tt = (u - cb) / den;
                const float zlo = zM[k - 1], zhi = zM[k < S ? k : S - 1];
                zn = zlo + tt * (zhi - zlo);
                ray_point(a.g, rr, zn, x, y, w);
                in = mask_at(a.maskvol, a.D, x, y, w) > 0.f;
                if (live) {
                    const size_t slot = (size_t)t * R + r;
                    a.out_z[slot] = zn;
                    a.out_pts[3 * slot] = x; a.out_pts[3 * slot + 1] = y; a.out_pts[3 * slot + 2] = w;
                    a.out_sdf[slot] = 100.f;                         // cat_z_vals default outside the mask (:135)
                    if (a.out_msk) a.out_msk[slot] = in ? 1 : 0;
                }
            }
            const bool valid = act && live && in;
            const unsigned long long bm = __ballot(valid);
            if (bm) {
                int base = 0;
                if (lane == 0) {
                    base = atomicAdd(a.count, __popcll(bm));
                    if (a.seg_cnt) atomicAdd(a.seg_cnt + (blockIdx.x * GR) / a.seg_rays, __popcll(bm));     // the wave's four rays lie in one segment
                }
                base = __shfl(base, 0);
                if (valid) a.list[base + __popcll(bm & lt)] = t * R + r;
            }
        }
    } else if (MODE == RM_FINALIZE) {
        for (int s0 = 0; s0 < S; s0 += GL) {
            const int smp = s0 + l;
            const bool act = smp < S && live;
            bool occ = false;
            if (act) {
                const float zc = zM[smp];
                const float d = (smp + 1 < S) ? zM[smp + 1] - zc : a.sample_dist;
                const float mz = zc + d * 0.5f;
                float x, y, w;
                ray_point(a.g, r, mz, x, y, w);
                const float m = mask_at(a.maskvol, a.D, x, y, w);
                const size_t p = (size_t)smp * R + r;
                a.dists[p] = d; a.mid_z[p] = mz; a.pm[p] = m;
                a.pts[3 * p] = x; a.pts[3 * p + 1] = y; a.pts[3 * p + 2] = w;
                occ = m > 0.f;
                if (!occ || a.defaults_everywhere) {                 // see k_ray_stream
                    a.o_sdf[p] = 100.f;
                    a.grad[3 * p] = 0.f; a.grad[3 * p + 1] = 0.f; a.grad[3 * p + 2] = 0.f;
                    a.rgb[3 * p] = 0.f; a.rgb[3 * p + 1] = 0.f; a.rgb[3 * p + 2] = 0.f;
                }
            }
            const unsigned long long bm = __ballot(occ);
            if (bm) {
                int base = 0;
                if (lane == 0) {
                    base = atomicAdd(a.count, __popcll(bm));
                    if (a.seg_cnt) atomicAdd(a.seg_cnt + (blockIdx.x * GR) / a.seg_rays, __popcll(bm));
                }
                base = __shfl(base, 0);
                if (occ) a.list[base + __popcll(bm & lt)] = smp * R + r;
            }
        }
    }
    round_epilogue<MODE>(a, S);
}

// render_core's compositing for small batches, sixteen lanes per ray: per-sample opacities and products in parallel, ONE lane runs the ordered
// accumulation (transmittance, weight / colour / depth sums, then the depth variance) out of LDS, the per-sample outputs are written in parallel.
constexpr int COMP_ARRAYS = 9;
__global__ __launch_bounds__(64) void k_ray_composite_group(RayGeom g, int S, const float* __restrict__ mid_z, const float* __restrict__ dists,
                                                            const float* __restrict__ pm, const float* __restrict__ sdf,
                                                            const float* __restrict__ grad, const float* __restrict__ rgb,
                                                            const uint8_t* __restrict__ nviews, float inv_s, float air, float bg, CompositeOut o) {
    extern __shared__ float lds[];
    const int R = g.R, lane = threadIdx.x, sub = lane >> 4, l = lane & 15;
    const int r = blockIdx.x * GR + sub;
    const bool live = r < R;
    const int rr = live ? r : R - 1;
    float* L[COMP_ARRAYS];
#pragma unroll
    for (int k = 0; k < COMP_ARRAYS; ++k) L[k] = lds + (size_t)(k * GR + sub) * S;
    float *aL = L[0], *c0L = L[1], *c1L = L[2], *c2L = L[3], *zL = L[4], *gL = L[5], *mL = L[6], *nL = L[7], *wL = L[8];
    const float dx = g.rays_d[3 * rr], dy = g.rays_d[3 * rr + 1], dz = g.rays_d[3 * rr + 2];
    for (int smp = l; smp < S; smp += GL) {
        const size_t p = (size_t)smp * R + rr;
        const float m = pm[p];
        const float gx = grad[3 * p], gy = grad[3 * p + 1], gz = grad[3 * p + 2];
        float pc;
        aL[smp] = composite_sample_alpha(dx, dy, dz, gx, gy, gz, m, dists[p], sdf[p], inv_s, air, pc);
        if (live) o.cdf[p] = pc;
        c0L[smp] = rgb[3 * p]; c1L[smp] = rgb[3 * p + 1]; c2L[smp] = rgb[3 * p + 2];
        zL[smp] = mid_z[p];
        const float gn = sqrtf(gx * gx + gy * gy + gz * gz) - 1.f;
        gL[smp] = m * (gn * gn);
        mL[smp] = m;
        nL[smp] = nviews[p] >= 2 ? 1.f : 0.f;
    }
    __syncthreads();
    if (l == 0 && live) {
        float T = 1.f, wsum = 0.f, wmax = 0.f, asum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f, ge = 0.f, gm = 0.f;
        int n_seen = 0;
        for (int smp = 0; smp < S; ++smp) {
            const float alpha = aL[smp];
            const float w = alpha * T;
            T = T * (1.f - alpha + 1e-7f);
            wL[smp] = w;
            wsum += w; wmax = fmaxf(wmax, w); asum += alpha;
            c0 += c0L[smp] * w; c1 += c1L[smp] * w; c2 += c2L[smp] * w;
            dep += zL[smp] * w;
            ge += gL[smp]; gm += mL[smp];
            n_seen += nL[smp] > 0.f ? 1 : 0;
        }
        const float bgc = bg * (1.f - wsum);
        o.color[3 * r] = c0 + bgc; o.color[3 * r + 1] = c1 + bgc; o.color[3 * r + 2] = c2 + bgc;
        o.depth[r] = dep;
        o.weights_sum[r] = wsum; o.weights_max[r] = wmax; o.alpha_sum[r] = asum;
        o.grad_err[2 * r] = ge; o.grad_err[2 * r + 1] = gm;
        o.color_mask[r] = n_seen > 8 ? 1 : 0;
        float dv = 0.f;
        for (int smp = 0; smp < S; ++smp) {
            const float d = zL[smp] - dep;
            dv += d * d * wL[smp];
        }
        o.depth_var[r] = dv;
    }
    __syncthreads();
    if (live)
        for (int smp = l; smp < S; smp += GL) o.weights[(size_t)smp * R + r] = wL[smp];
}

// cat_z_vals for a block size other than NFIX: the plain per-ray merge on the global lists (render_math.h merge_ray), occupancy bytes along
__global__ __launch_bounds__(64) void k_ray_merge_any(int R, float* z, float* sdf, uint8_t* msk, int S, const float* new_z, const float* new_sdf,
                                                      const uint8_t* new_msk, int n_new, const int* seg_prev, int seg_rays) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const bool keep_default = seg_prev && seg_prev[r / seg_rays] <= 1;         // cat_z_vals' rule per segment (seg_keeps_default)
    constexpr int NMAX = 32;
    float nz[NMAX], ns[NMAX];
    unsigned nt[NMAX];
    for (int base = 0; base < n_new; base += NMAX) {
        const int nb = n_new - base < NMAX ? n_new - base : NMAX;
        for (int j = 0; j < nb; ++j) {
            nz[j] = new_z[(size_t)(base + j) * R + r]; ns[j] = keep_default ? 100.f : new_sdf[(size_t)(base + j) * R + r];
            nt[j] = new_msk ? new_msk[(size_t)(base + j) * R + r] : 0u;
        }
        GlobalMergeTag a{z, sdf, msk, (size_t)R, r};
        ArrayBlock blk{nz, ns, nt};
        merge_core(a, S + base, blk, nb);
    }
}

__global__ __launch_bounds__(64) void k_ray_composite(RayGeom g, int S, const float* __restrict__ mid_z, const float* __restrict__ dists,
                                                       const float* __restrict__ pm, const float* __restrict__ sdf,
                                                       const float* __restrict__ grad, const float* __restrict__ rgb,
                                                       const uint8_t* __restrict__ nviews, float inv_s, float air, float bg,
                                                       CompositeOut o) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r < g.R) composite_ray(g, r, S, mid_z, dists, pm, sdf, grad, rgb, nviews, inv_s, air, bg, o);
}

// ---- tolerance-bounded colour work removal (O2345RenderIO.weight_cull) ------------------------------------------------------------------------
// After the SDF + gradient pass every compositing weight w = alpha * T is known -- it does not depend on the colours.  This pass runs composite_ray's
// transmittance chain (the SAME functions in the same order: the w it thresholds is bit-identical to the `weights` the composite kernel returns) and
// emits the list of the occupied samples with w >= thr: only those go through the visibility sort and the colour network.  keep[p] = 1 there, 0 on
// every other slot (unoccupied or culled: the counting kernel supplies their valid-view counts, so the per-ray colour mask stays exact); culled occupied
// samples get the colour 0 the composite kernel will multiply by their w < thr.
__global__ __launch_bounds__(256) void k_ray_cull(RayGeom g, int S, const float* __restrict__ dists, const float* __restrict__ pm, const float* __restrict__ sdf,
                                                  const float* __restrict__ grad, float inv_s, float air, float thr, float* __restrict__ keep,
                                                  float* __restrict__ rgb, int* __restrict__ list, int* __restrict__ count) {
    const int R = g.R;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool live = r < R;
    const int rr = live ? r : R - 1;
    const float dx = g.rays_d[3 * rr], dy = g.rays_d[3 * rr + 1], dz = g.rays_d[3 * rr + 2];
    ValidBits bits{};
    int cnt = 0;
    float T = 1.f;
    constexpr int CB = 8;
    for (int sb = 0; sb < S; sb += CB) {
        float bm[CB], bd[CB], bs[CB], bg[CB][3];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int s = sb + k < S ? sb + k : S - 1;
            const size_t p = (size_t)s * R + rr;
            bm[k] = pm[p]; bd[k] = dists[p]; bs[k] = sdf[p];
            bg[k][0] = grad[3 * p]; bg[k][1] = grad[3 * p + 1]; bg[k][2] = grad[3 * p + 2];
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (sb + k < S) {
                const int s = sb + k;
                const size_t p = (size_t)s * R + rr;
                float pc;
                const float alpha = composite_sample_alpha(dx, dy, dz, bg[k][0], bg[k][1], bg[k][2], bm[k], bd[k], bs[k], inv_s, air, pc);
                const float w = alpha * T;
                T = T * (1.f - alpha + 1e-7f);
                const bool occ = bm[k] > 0.f;
                const bool kp = live && occ && w >= thr;
                if (live) {
                    keep[p] = kp ? 1.f : 0.f;
                    if (occ && !kp) { rgb[3 * p] = 0.f; rgb[3 * p + 1] = 0.f; rgb[3 * p + 2] = 0.f; }
                }
                if (kp) { bits.w[s >> 5] |= 1u << (s & 31); ++cnt; }
            }
        }
    }
    append_wave(bits, S, cnt, R, rr, list, count);
}

// The same for SMALL batches, sixteen lanes per ray (cf. k_ray_composite_group): the per-sample opacities in parallel, ONE lane runs the transmittance scan in
// the reference's order, the threshold test and the list append in parallel again.  A 512-ray chunk: 62 us with one lane per ray (128 dependent steps of two
// exponentials and a division each), ~20 us here.  Same functions, same order in the scan: the same w, the same list (as a set), the same flags.
__global__ __launch_bounds__(64) void k_ray_cull_group(RayGeom g, int S, const float* __restrict__ dists, const float* __restrict__ pm, const float* __restrict__ sdf,
                                                       const float* __restrict__ grad, float inv_s, float air, float thr, float* __restrict__ keep,
                                                       float* __restrict__ rgb, int* __restrict__ list, int* __restrict__ count) {
    extern __shared__ float lds[];
    const int R = g.R, lane = threadIdx.x, sub = lane >> 4, l = lane & 15;
    const int r = blockIdx.x * GR + sub;
    const bool live = r < R;
    const int rr = live ? r : R - 1;
    float* aL = lds + (size_t)(0 * GR + sub) * S;
    float* mL = lds + (size_t)(1 * GR + sub) * S;
    float* wL = lds + (size_t)(2 * GR + sub) * S;
    const float dx = g.rays_d[3 * rr], dy = g.rays_d[3 * rr + 1], dz = g.rays_d[3 * rr + 2];
    for (int smp = l; smp < S; smp += GL) {
        const size_t p = (size_t)smp * R + rr;
        const float m = pm[p];
        float pc;
        aL[smp] = composite_sample_alpha(dx, dy, dz, grad[3 * p], grad[3 * p + 1], grad[3 * p + 2], m, dists[p], sdf[p], inv_s, air, pc);
        mL[smp] = m;
    }
    __syncthreads();
    if (l == 0) {
        float T = 1.f;
        for (int smp = 0; smp < S; ++smp) {
            const float alpha = aL[smp];
            wL[smp] = alpha * T;
            T = T * (1.f - alpha + 1e-7f);
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int s0 = 0; s0 < S; s0 += GL) {
        const int smp = s0 + l;
        const bool act = smp < S && live;
        const bool occ = act && mL[smp] > 0.f;
        const bool kp = occ && wL[smp] >= thr;
        if (act) {
            const size_t p = (size_t)smp * R + r;
            keep[p] = kp ? 1.f : 0.f;
            if (occ && !kp) { rgb[3 * p] = 0.f; rgb[3 * p + 1] = 0.f; rgb[3 * p + 2] = 0.f; }
        }
        const unsigned long long bm = __ballot(kp);
        if (bm) {
            int base = 0;
            if (lane == 0) base = atomicAdd(count, __popcll(bm));
            base = __shfl(base, 0);
            if (kp) list[base + __popcll(bm & lt)] = smp * R + r;
        }
    }
}

// per-call scalars of render()'s returned dict (:586-633): sums over the rays in a FIXED order (thread t adds rays t, t + 1024, ... in fp64, then a
// fixed LDS tree): deterministic, one workgroup, no atomics.  out[0] = alpha_sum.mean(), out[1] = alpha_sum.sum() / (R S) ("alpha_mean"),
// out[2] = sum grad_err[.,0] / (sum grad_err[.,1] + 1e-5) ("gradient_error_fine"), out[3] = number of list entries the network kernels evaluated.
// Segment mode (seg_rays > 0): workgroup k reduces the rays of segment k in the order a call on that segment alone would -> out[k][0..3].
__global__ __launch_bounds__(1024) void k_ray_scalars(int R_all, int S, const float* __restrict__ alpha_sum, const float* __restrict__ grad_err,
                                                      const int* __restrict__ count, float* __restrict__ out, int seg_rays) {
    __shared__ double red[3][1024];
    const int r0 = seg_rays ? (int)blockIdx.x * seg_rays : 0;
    const int R = seg_rays ? (R_all - r0 < seg_rays ? R_all - r0 : seg_rays) : R_all;
    alpha_sum += r0; grad_err += 2 * (size_t)r0; count += seg_rays ? blockIdx.x : 0; out += 4 * (size_t)blockIdx.x;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int r = threadIdx.x; r < R; r += 1024) { a += (double)alpha_sum[r]; b += (double)grad_err[2 * r]; c += (double)grad_err[2 * r + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
    __syncthreads();
    for (int off = 512; off; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; red[2][threadIdx.x] += red[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(red[0][0] / (double)R);
        out[1] = (float)(red[0][0] / ((double)R * (double)S));
        out[2] = (float)(red[1][0] / (red[2][0] + 1e-5));
        out[3] = (float)*count;
    }
}

}  // namespace o2345

using namespace o2345;

extern "C" {

// ---- stage entry points (used by the parity tests; the orchestrator below calls the same kernels) -----------------
static int ray_coarse_launch(const float* rays_o, const float* rays_d, int R, float near, float far, const float* near_ray, const float* far_ray, int S,
                             const float* t_rand, float* z, float* pts, const float* maskvol, int D, uint8_t* msk, void* stream, int* zero16 = nullptr) {
    O2345_REQUIRE(rays_o && rays_d && z && pts && R > 0 && S > 1, "ray_coarse: bad arguments");
    O2345_REQUIRE((near_ray != nullptr) == (far_ray != nullptr), "ray_coarse: per-ray near and far come together");
    RayGeom g{rays_o, rays_d, R};
    hipLaunchKernelGGL(k_ray_coarse, dim3(cdiv((long long)R * S, 256)), dim3(256), 0, (hipStream_t)stream, g, near, far, near_ray, far_ray, S, t_rand, z, pts,
                       maskvol, D, msk, zero16);
    return check_launch("ray_coarse");
}

int o2345_ray_coarse_jitter(const float* rays_o, const float* rays_d, int R, float near, float far, int S, const float* t_rand,
                            float* z, float* pts, void* stream) {
    return ray_coarse_launch(rays_o, rays_d, R, near, far, nullptr, nullptr, S, t_rand, z, pts, nullptr, 0, nullptr, stream);
}

int o2345_ray_coarse_per_ray(const float* rays_o, const float* rays_d, int R, const float* near_ray, const float* far_ray, int S,
                             const float* t_rand, float* z, float* pts, void* stream) {
    O2345_REQUIRE(near_ray && far_ray, "ray_coarse_per_ray: null pointer");
    return ray_coarse_launch(rays_o, rays_d, R, 0.f, 0.f, near_ray, far_ray, S, t_rand, z, pts, nullptr, 0, nullptr, stream);
}

int o2345_ray_coarse(const float* rays_o, const float* rays_d, int R, float near, float far, int S, float* z, float* pts, void* stream) {
    return o2345_ray_coarse_jitter(rays_o, rays_d, R, near, far, S, nullptr, z, pts, stream);
}

// One round launch.  A merge of exactly NFIX samples runs fused; any other block size is merged on the global lists first (k_ray_merge_any).
static int ray_round_launch(int mode, RoundArgs a, float* wbuf /* [rows][R] scratch of the streaming up-sample kernel, or null */, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int R = a.g.R;
    if (a.n_new > 0 && a.n_new != NFIX) {
        hipLaunchKernelGGL(k_ray_merge_any, dim3(cdiv(R, 64)), dim3(64), 0, s, R, a.z, a.sdf, a.msk, a.S, a.new_z, a.new_sdf, a.new_msk, a.n_new, a.seg_prev, a.seg_rays);
        a.seg_prev = nullptr;                    // applied
        a.S += a.n_new;
        a.n_new = 0;
        if (mode == RM_MERGE_ONLY) return check_launch("ray_merge");
    }
    const bool merge = a.n_new > 0;
    // large batches: streaming kernels (one lane per ray, full occupancy); small ones: sixteen lanes per ray (csrc/common.h knobs(): O2345_RAY_STREAM_MIN)
    const bool streaming = (long long)R >= knobs().ray_stream_min && (mode != RM_UPSAMPLE || wbuf != nullptr);
    if (streaming) {
        const dim3 grid(cdiv(R, 256)), block(256);
#define O2345_STREAM(M, MG) hipLaunchKernelGGL((k_ray_stream<M, MG>), grid, block, 0, s, a, wbuf);
        if (mode == RM_UPSAMPLE) { if (merge) O2345_STREAM(RM_UPSAMPLE, true) else O2345_STREAM(RM_UPSAMPLE, false) }
        else if (mode == RM_FINALIZE) { if (merge) O2345_STREAM(RM_FINALIZE, true) else O2345_STREAM(RM_FINALIZE, false) }
        else if (merge) O2345_STREAM(RM_MERGE_ONLY, true)
#undef O2345_STREAM
        return check_launch("ray_round (streaming)");
    }
    {
        const int rows = a.S + (merge ? NFIX : 0);
        const size_t lds = group_lds_bytes(rows);
        O2345_REQUIRE(lds <= 64 * 1024, "ray kernels: %d list rows per ray (at most 600)", rows);
        const dim3 grid(cdiv(R, GR)), block(64);
#define O2345_GROUP(M, MG) hipLaunchKernelGGL((k_ray_group<M, MG>), grid, block, lds, s, a);
        if (mode == RM_UPSAMPLE) { if (merge) O2345_GROUP(RM_UPSAMPLE, true) else O2345_GROUP(RM_UPSAMPLE, false) }
        else if (mode == RM_FINALIZE) { if (merge) O2345_GROUP(RM_FINALIZE, true) else O2345_GROUP(RM_FINALIZE, false) }
        else if (merge) O2345_GROUP(RM_MERGE_ONLY, true)
#undef O2345_GROUP
    }
    return check_launch("ray_round");
}

// ---- stage entry points (used by the parity tests; the orchestrator below launches the same kernel with the merge fused in) -----------------
int o2345_ray_upsample(const float* rays_o, const float* rays_d, int R, const float* z, const float* sdf, int S, float inv_s,
                       const float* maskvol, int D, float* wbuf, int n_imp, float* new_z, float* new_pts, float* new_sdf,
                       int32_t* list, int32_t* count_dev, void* stream) {
    O2345_REQUIRE(rays_o && rays_d && z && sdf && maskvol && new_z && new_pts && new_sdf && list && count_dev, "ray_upsample: null pointer");
    O2345_REQUIRE(n_imp >= 1 && n_imp <= 256 && S >= 2 && R > 0, "ray_upsample: 1..256 new samples per call, at least 2 samples per ray (got %d, %d)", n_imp, S);
    hipStream_t s = (hipStream_t)stream;
    O2345_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), s));
    RoundArgs a{};
    a.g = RayGeom{rays_o, rays_d, R};
    a.z = const_cast<float*>(z); a.sdf = const_cast<float*>(sdf); a.S = S;          // read-only without a merge
    a.maskvol = maskvol; a.D = D; a.inv_s = inv_s; a.n_imp = n_imp;
    a.out_z = new_z; a.out_pts = new_pts; a.out_sdf = new_sdf; a.list = list; a.count = count_dev;
    int rc = ray_round_launch(RM_UPSAMPLE, a, wbuf, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_quirk_min2, dim3(1), dim3(1), 0, s, count_dev);
    return check_launch("ray_upsample");
}

int o2345_ray_merge(int R, float* z, float* sdf, int S, float* new_z, float* new_sdf, int n_new, void* stream) {
    O2345_REQUIRE(z && sdf && new_z && new_sdf && R > 0 && S >= 1 && n_new >= 1, "ray_merge: bad arguments");
    RoundArgs a{};
    a.g = RayGeom{nullptr, nullptr, R};
    a.z = z; a.sdf = sdf; a.S = S; a.new_z = new_z; a.new_sdf = new_sdf; a.n_new = n_new;
    return ray_round_launch(RM_MERGE_ONLY, a, nullptr, stream);
}

int o2345_ray_finalize(const float* rays_o, const float* rays_d, int R, const float* z, int S, float sample_dist,
                       const float* maskvol, int D, float* mid_z, float* dists, float* pts, float* pm, float* sdf,
                       float* grad, float* rgb, int32_t* list, int32_t* count_dev, void* stream) {
    O2345_REQUIRE(rays_o && rays_d && z && maskvol && mid_z && dists && pts && pm && sdf && grad && rgb && list && count_dev, "ray_finalize: null pointer");
    O2345_REQUIRE(S >= 1 && S <= 256 && R > 0, "ray_finalize: at most 256 samples per ray (got %d)", S);
    O2345_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), (hipStream_t)stream));
    RoundArgs a{};
    a.g = RayGeom{rays_o, rays_d, R};
    a.z = const_cast<float*>(z); a.sdf = const_cast<float*>(z); a.S = S;            // the SDF list is not used by the finalize step (staged, never read)
    a.maskvol = maskvol; a.D = D; a.sample_dist = sample_dist;
    a.mid_z = mid_z; a.dists = dists; a.pts = pts; a.pm = pm; a.o_sdf = sdf; a.grad = grad; a.rgb = rgb; a.defaults_everywhere = 1;
    a.list = list; a.count = count_dev;
    return ray_round_launch(RM_FINALIZE, a, nullptr, stream);
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
    if ((long long)R >= knobs().ray_stream_min || (size_t)COMP_ARRAYS * GR * S * sizeof(float) > 64 * 1024)
        hipLaunchKernelGGL(k_ray_composite, dim3(cdiv(R, 64)), dim3(64), 0, (hipStream_t)stream, g, S, mid_z, dists, pm, sdf, grad, rgb, nviews, inv_s, alpha_inter_ratio, background, o);
    else      // small batches: sixteen lanes per ray
        hipLaunchKernelGGL(k_ray_composite_group, dim3(cdiv(R, GR)), dim3(64), (size_t)COMP_ARRAYS * GR * S * sizeof(float), (hipStream_t)stream, g, S, mid_z, dists, pm, sdf,
                           grad, rgb, nviews, inv_s, alpha_inter_ratio, background, o);
    return check_launch("ray_composite");
}

// ---- the whole render() call -----------------------------------------------------------------------------------------
// Workspace layout (floats unless noted), S = n_samples + n_importance, NI = n_importance / 4, R rays:
//   z[S*R] sdf[S*R] new_z[NI*R] new_sdf[NI*R] pts[3*S*R] list[S*R ints] count[64 ints] msk[S*R bytes] new_msk[NI*R bytes] wbuf[S*R] (streaming kernels only)
//   culled list[S*R ints] (weight_cull > 0; its keep flags reuse sdf[], dead after the last merge)
//   segment counters[5][R/64 + 1 ints] (segment_rays > 0)
//   ... and, when the list is sorted: sorted list[S*R ints] + the workspace of o2345_list_sort_by_visibility (csrc/list_sort.hip)
static size_t render_cull_list_offset(int R, int n_samples, int n_importance) {
    const size_t S = (size_t)n_samples + n_importance, NI = (size_t)(n_importance / 4 > 0 ? n_importance / 4 : 1);
    size_t bytes = ((S * 2 + NI * 2 + 3 * S + S) * (size_t)R + 64) * 4 + ((S + NI) * (size_t)R + 3) / 4 * 4;
    if ((long long)R >= knobs().ray_stream_min) bytes += S * (size_t)R * 4;
    return (bytes + 255) / 256 * 256;
}
static size_t render_seg_counters_offset(int R, int n_samples, int n_importance) {
    const size_t S = (size_t)n_samples + n_importance;
    const size_t bytes = render_cull_list_offset(R, n_samples, n_importance) + S * (size_t)R * 4;      // + the culled list (weight_cull > 0)
    return (bytes + 255) / 256 * 256;
}
static size_t render_core_workspace_bytes(int R, int n_samples, int n_importance) {
    const size_t bytes = render_seg_counters_offset(R, n_samples, n_importance) + 5 * ((size_t)R / 64 + 1) * 4;   // + per-segment counters of the five rounds (segment_rays > 0)
    return (bytes + 255) / 256 * 256;
}
// The occupied-point list is grouped by view-visibility signature only where that pays: the sort is 4 launches per 8 views and costs 0.1 - 0.25 ms
// whatever the list length, the colour kernel gains ~10 % of its time.  Below 2^20 sample slots (a 512-ray chunk of the reference's val loop has
// 65,536: colour kernel 0.5 ms) the emission order is kept.  O2345_LIST_SORT=0 disables the sort everywhere (A/B knob).
static bool render_sorts_list(int R, int n_samples, int n_importance, int V) {
    return knobs().list_sort && V <= 32 && ((long long)n_samples + n_importance) * (long long)R >= (1ll << 20);
}
size_t o2345_render_workspace_bytes(int R, int n_samples, int n_importance, int V) {
    const size_t S = (size_t)n_samples + n_importance;
    size_t b = render_core_workspace_bytes(R, n_samples, n_importance);
    if (render_sorts_list(R, n_samples, n_importance, V)) b += S * (size_t)R * 4 + o2345_list_sort_workspace_bytes((long long)(S * (size_t)R), V);
    return b;
}

int o2345_render_rays(const O2345RenderIO* io, void* workspace, size_t workspace_bytes, void* stream) {
    O2345_REQUIRE(io && workspace, "render_rays: null pointer");
    const int R = io->R, NS = io->n_samples, NIMP = io->n_importance;
    O2345_REQUIRE(NIMP % 4 == 0 && NIMP > 0 && NS > 1, "render_rays: n_importance must be a positive multiple of 4");
    O2345_REQUIRE(R > 0 && ((long long)NS + NIMP) * (long long)R < 2147483647LL, "render_rays: R * (n_samples + n_importance) must stay below 2^31 "
                  "(sample slots are 32-bit); split the ray batch (got R = %d)", R);
    O2345_REQUIRE(workspace_bytes >= o2345_render_workspace_bytes(R, NS, NIMP, io->V), "render_rays: workspace too small");
    O2345_REQUIRE(io->color_x3_blob || io->color_mfma_blob, "render_rays: a colour network blob is required (color_x3_blob or color_mfma_blob)");
    O2345_REQUIRE((io->near_ray != nullptr) == (io->far_ray != nullptr), "render_rays: per-ray near and far come together");
    const size_t S = (size_t)NS + NIMP, NI = NIMP / 4, RR = R;
    O2345_REQUIRE(S <= 256, "render_rays: at most 256 samples per ray (got %d)", (int)S);
    float* z = (float*)workspace;
    float* sdf = z + S * RR;
    float* new_z = sdf + S * RR;
    float* new_sdf = new_z + NI * RR;
    float* pts = new_sdf + NI * RR;
    int* list = (int*)(pts + 3 * S * RR);
    int* count = list + S * RR;                  // [0..3]: new points of the four up-sampling rounds, [4]: occupied mid-points
    uint8_t* msk = (uint8_t*)(count + 64);       // occupancy of every sample point, carried with the lists
    uint8_t* new_msk = msk + S * RR;
    float* wbuf = (long long)R >= knobs().ray_stream_min ? (float*)(msk + ((S + NI) * RR + 3) / 4 * 4) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    O2345_REQUIRE(io->sdf_mode == 0 || io->sdf_mode == 2, "render_rays: SDF mode %d (0 = fp32, 2 = split-f16; the bf16 mode 1 was removed)", io->sdf_mode);
    // segment mode: the rays are consecutive render() calls of G rays each, evaluated together, the reference's two per-call rules applied per segment
    const int G = io->segment_rays;
    O2345_REQUIRE(G >= 0 && G % 64 == 0, "render_rays: segment_rays must be 0 or a multiple of 64 (got %d)", G);
    O2345_REQUIRE(G == 0 || !io->near_ray, "render_rays: segment_rays needs one near / far pair (sample_dist is a per-call mean for per-ray near / far)");
    const int n_seg = G ? (R + G - 1) / G : 0;
    int* segc = (int*)((char*)workspace + render_seg_counters_offset(R, NS, NIMP));      // [5][n_seg]
    if (G) O2345_HIP(hipMemsetAsync(segc, 0, (size_t)5 * n_seg * sizeof(int), s));
    auto sdf_eval = [&](int variant, const float* p, const int* idx, const int* cnt, long long n, float* out, float* grad) {
        if (io->sdf_mode == 2 && variant == 0) return o2345_sdf_mlp_x3(io->sdf_blob, io->vol_cl, io->D, p, idx, cnt, n, 0, 1.f, out, stream);
        if (io->sdf_mode == 2 && variant == 2) return o2345_sdf_grad_x3(io->sdf_blob, io->vol_cl, io->D, p, idx, cnt, n, 0, 1.f, out, grad, stream);
        return o2345_sdf_mlp(variant, io->sdf_blob, io->vol_cl, io->D, p, idx, cnt, n, 0, 1.f, out, nullptr, nullptr, grad, stream);
    };
    // count[0..15]: every device-side counter of the call, cleared by the first kernel ([8..12]: finished workgroups of the five rounds, round_epilogue)
    if ((rc = ray_coarse_launch(io->rays_o, io->rays_d, R, io->near, io->far, io->near_ray, io->far_ray, NS, io->t_rand, z, pts, io->maskvol, io->D, msk, stream, count))) return rc;
    // coarse SDF on ALL points (not masked, :525-528)
    if ((rc = sdf_eval(0, pts, nullptr, nullptr, (long long)NS * R, sdf, nullptr))) return rc;
    // four up-sampling rounds (:531-547); round i > 0 first merges round i - 1's samples (cat_z_vals) inside the same kernel
    RoundArgs ra{};
    ra.g = RayGeom{io->rays_o, io->rays_d, R};
    ra.z = z; ra.sdf = sdf; ra.msk = msk; ra.maskvol = io->maskvol; ra.D = io->D;
    ra.new_z = new_z; ra.new_sdf = new_sdf; ra.new_msk = new_msk;
    ra.n_imp = (int)NI; ra.out_z = new_z; ra.out_pts = pts; ra.out_sdf = new_sdf; ra.out_msk = new_msk; ra.list = list;
    int cur = NS;
    for (int i = 0; i < 4; ++i) {
        ra.S = cur; ra.n_new = i ? (int)NI : 0; ra.inv_s = 64.f * (float)(1 << i); ra.count = count + i; ra.done = count + 8 + i;
        ra.seg_rays = G; ra.n_seg = n_seg; ra.seg_cnt = G ? segc + (size_t)i * n_seg : nullptr; ra.seg_prev = (G && i) ? segc + (size_t)(i - 1) * n_seg : nullptr;
        if ((rc = ray_round_launch(RM_UPSAMPLE, ra, wbuf, stream))) return rc;             // incl. cat_z_vals' "more than one point" rule (round_epilogue)
        cur += ra.n_new;
        if ((rc = sdf_eval(0, pts, list, count + i, 0, new_sdf, nullptr))) return rc;
    }
    count += 4;
    // :484 -- the caller passes the mean over the rays for per-ray near / far; the scalar form is (far - near) / n_samples
    const float sample_dist = io->sample_dist > 0.f ? io->sample_dist : (io->far - io->near) / (float)NS;
    float* fpts = pts;   // reuse
    // the last cat_z_vals fused with render_core's head
    ra.S = cur; ra.n_new = (int)NI; ra.count = count; ra.done = count + 8; ra.sample_dist = sample_dist;        // count was advanced by 4: slot 12
    ra.seg_cnt = G ? segc + (size_t)4 * n_seg : nullptr; ra.seg_prev = G ? segc + (size_t)3 * n_seg : nullptr;
    ra.mid_z = io->mid_z; ra.dists = io->dists; ra.pts = fpts; ra.pm = io->pm; ra.o_sdf = io->sdf; ra.grad = io->grad; ra.rgb = io->rgb; ra.defaults_everywhere = 0;
    if ((rc = ray_round_launch(RM_FINALIZE, ra, nullptr, stream))) return rc;               // incl. render_core's "first 100 points" rule (round_epilogue)
    const bool cull = io->weight_cull > 0.f;
    const bool sorts = render_sorts_list(R, NS, NIMP, io->V);
    int* slist = (int*)((char*)workspace + render_core_workspace_bytes(R, NS, NIMP));
    void* sort_ws = (void*)(slist + S * RR);
    // the list grouped by view-visibility signature (stable): the colour kernel then skips every (tile, view) pair in which no point sees the view instead of
    // 3/4 of them -- 40.0 -> 36.1 ms at 8 views, bit-identical results (csrc/list_sort.hip); only for lists long enough to pay for it (render_sorts_list).
    // The SDF-gradient kernel does not care about the order (10.17 vs 10.15 ms): with weight culling it runs on the emission-order list and only the
    // (shorter) culled list is sorted.
    auto sort_list = [&](const int* in, const int* n_dev) {
        return o2345_list_sort_by_visibility(fpts, in, n_dev, (long long)(S * RR), io->proj, io->V, io->H, io->W, slist, nullptr, sort_ws,
                                             o2345_list_sort_workspace_bytes((long long)(S * RR), io->V), stream);
    };
    const int* clist = list;                     // what the colour network evaluates
    const int* ccount = count;
    const float* counted_elsewhere = io->pm;     // slots whose valid-view count the colour kernel writes itself
    if (!cull && sorts) {
        if ((rc = sort_list(list, count))) return rc;
        list = slist; clist = slist;
    }
    if ((rc = sdf_eval(2, fpts, list, count, 0, io->sdf, io->grad))) return rc;
    if (cull) {
        int* list2 = (int*)((char*)workspace + render_cull_list_offset(R, NS, NIMP));
        float* keep = sdf;                       // the workspace's SDF list is dead after the last merge
        if ((long long)R >= knobs().ray_stream_min || (size_t)3 * GR * S * sizeof(float) > 64 * 1024)
            hipLaunchKernelGGL(k_ray_cull, dim3(cdiv(R, 256)), dim3(256), 0, s, ra.g, (int)S, io->dists, io->pm, io->sdf, io->grad, io->inv_s, io->alpha_inter_ratio,
                               io->weight_cull, keep, io->rgb, list2, count + 1);
        else      // small batches: sixteen lanes per ray
            hipLaunchKernelGGL(k_ray_cull_group, dim3(cdiv(R, GR)), dim3(64), (size_t)3 * GR * S * sizeof(float), s, ra.g, (int)S, io->dists, io->pm, io->sdf, io->grad,
                               io->inv_s, io->alpha_inter_ratio, io->weight_cull, keep, io->rgb, list2, count + 1);
        if ((rc = check_launch("ray_cull"))) return rc;
        clist = list2; ccount = count + 1; counted_elsewhere = keep;
        if (sorts) {
            if ((rc = sort_list(list2, count + 1))) return rc;
            clist = slist;
        }
    }
    // valid-view counts (feed the per-ray colour mask): the colour kernels write them for the points they evaluate (the occupied ones, 88 % at
    // BASELINE config 2); this pass covers the rest (four IEEE divisions per view make it VALU-bound: 0.48 ms over all points)
    if ((rc = o2345_view_count_unlisted(fpts, (long long)S * R, counted_elsewhere, io->maskvol, io->D, io->proj, io->V, io->H, io->W, io->nviews, stream))) return rc;
    if (io->color_x3_blob)
        rc = o2345_color_points_x3(io->color_x3_blob, io->vol_cl, io->maskvol, io->D, io->cmaps, io->proj, io->cam_pos, io->V, io->H, io->W, fpts, clist, ccount, 0, io->query_cam, nullptr, io->rgb, io->nviews, io->color_stats, stream);
    else
        rc = o2345_color_points_mfma(io->color_mfma_blob, io->vol_cl, io->maskvol, io->D, io->cmaps, io->proj, io->cam_pos, io->V, io->H, io->W, fpts, clist, ccount, 0, io->query_cam, nullptr, io->rgb, io->nviews, io->color_stats, stream);
    if (rc) return rc;
    if ((rc = o2345_ray_composite(io->rays_o, io->rays_d, R, (int)S, io->mid_z, io->dists, io->pm, io->sdf, io->grad, io->rgb, io->nviews, io->inv_s, io->alpha_inter_ratio, io->background,
                                  io->color, io->depth, io->weights, io->cdf, io->weights_sum, io->weights_max, io->depth_var, io->alpha_sum, io->grad_err, io->color_mask, stream))) return rc;
    if (io->scalars) hipLaunchKernelGGL(k_ray_scalars, dim3(G ? n_seg : 1), dim3(1024), 0, s, R, (int)S, io->alpha_sum, io->grad_err, G ? segc + (size_t)4 * n_seg : count,
                                        io->scalars, G);
    if (io->z_vals) O2345_HIP(hipMemcpyAsync(io->z_vals, z, S * RR * sizeof(float), hipMemcpyDeviceToDevice, s));
    return check_launch("render_rays");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_render() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_quirk_min2));
}
}  // namespace o2345
