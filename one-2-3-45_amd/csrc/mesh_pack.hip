// Mesh serialisation: the tail of validate_mesh / validate_colored_mesh (models/trainer_generic.py:1287-1303,
// 1365-1382): index-space marching-cubes vertices -> world frame (bounds, scale_mat, trans_mat) -> uint8 colours -> the
// binary little-endian PLY records trimesh would write (vertex: 3 x float32 [+ 4 x uint8 rgba]; face: uint8 count = 3 +
// 3 x int32).  The reference does this in numpy on the host after a D2H copy of float64 vertices; here both record arrays
// are produced on the device, so export is one D2H copy of 16 B/vertex + 13 B/face followed by a single file write.
// Arithmetic follows the reference's numpy expressions in fp64 (verts / (R-1) * (bmax - bmin) + bmin; * s + t; trans @ [v,1])
// and rounds to float32 only in the record, as trimesh's PLY exporter does.
#include "common.h"
#include <string.h>
#include <thread>
#include <vector>

namespace o2345 {

// verts[i][d] = verts[i][d] / div * ext[d] + off[d] in fp64, in place: extract_geometry's index -> world step (sparse_neus_renderer.py:936), the same
// expression numpy evaluates on the host (IEEE fp64 division, multiplication, addition: bit-identical), 8 MB less to touch on the host per mesh
__global__ __launch_bounds__(256) void k_verts_to_world(double* __restrict__ v, long long n3, double div, double e0, double e1, double e2, double o0,
                                                        double o1, double o2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const int d = (int)(i % 3);
    const double e = d == 0 ? e0 : d == 1 ? e1 : e2, o = d == 0 ? o0 : d == 1 ? o1 : o2;
    v[i] = v[i] / div * e + o;
}

struct MeshXform {
    double inv_rm1;            // 1 / (R - 1)   (the reference divides; kept as a division below)
    int R;
    double bmin[3], bext[3];   // bound_min, bound_max - bound_min
    int has_scale; double s, t[3];          // scale_mat[0,0], scale_mat[:3,3]
    int has_trans; double T[12];            // rows 0..2 of trans_mat (4x4, row-major)
};

__global__ __launch_bounds__(256) void k_pack_vertices(const double* __restrict__ vidx /*[n,3] index coords*/, long long n, MeshXform x,
                                                       const float* __restrict__ rgb /*[n,3] or null*/, uint8_t* __restrict__ rec,
                                                       int stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        v[d] = vidx[3 * i + d] / (double)(x.R - 1) * x.bext[d] + x.bmin[d];        // sparse_neus_renderer.py:936
        if (x.has_scale) v[d] = v[d] * x.s + x.t[d];
    }
    if (x.has_trans) {
        double w[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) w[r] = ((x.T[4 * r] * v[0] + x.T[4 * r + 1] * v[1]) + x.T[4 * r + 2] * v[2]) + x.T[4 * r + 3];
        v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
    }
    uint8_t* o = rec + i * stride;
    const float f[3] = {(float)v[0], (float)v[1], (float)v[2]};
    __builtin_memcpy(o, f, 12);
    if (rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[12 + c] = (uint8_t)(int)(rgb[3 * i + c] * 255.f);    // np.array(color * 255, dtype=uint8): truncation
        o[15] = 255;
    }
}

template <typename IDX>
__global__ __launch_bounds__(256) void k_pack_faces(const IDX* __restrict__ tris /*[m,3]*/, long long m, uint8_t* __restrict__ rec) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    uint8_t* o = rec + i * 13;
    o[0] = 3;
    const int32_t f[3] = {(int32_t)tris[3 * i], (int32_t)tris[3 * i + 1], (int32_t)tris[3 * i + 2]};
    __builtin_memcpy(o + 1, f, 12);
}

}  // namespace o2345

using namespace o2345;

extern "C" {

// verts_idx: device fp64 [n,3] (marching-cubes index coordinates of an R^3 grid); bounds / scale / trans are HOST arrays
// (bound_min[3], bound_max[3]; scale_mat 4x4 row-major or NULL; trans_mat 4x4 row-major or NULL, both fp32 like the reference's).
// rgb: device fp32 [n,3] in [0,1] or NULL.  vertex_records: device, n * (rgb ? 16 : 12) bytes.
int o2345_mesh_pack_vertices(const double* verts_idx, long long n, int grid_R, const float* bound_min, const float* bound_max,
                             const float* scale_mat, const float* trans_mat, const float* rgb, uint8_t* vertex_records, void* stream) {
    O2345_REQUIRE(bound_min && bound_max && grid_R >= 2, "mesh_pack_vertices: bad bounds / resolution");
    if (n <= 0) return 0;
    O2345_REQUIRE(verts_idx && vertex_records, "mesh_pack_vertices: null pointer");
    MeshXform x{};
    x.R = grid_R;
    for (int d = 0; d < 3; ++d) { x.bmin[d] = (double)bound_min[d]; x.bext[d] = (double)(bound_max[d] - bound_min[d]); }   // fp32 subtraction, as torch does
    x.has_scale = scale_mat != nullptr;
    if (scale_mat) { x.s = (double)scale_mat[0]; x.t[0] = (double)scale_mat[3]; x.t[1] = (double)scale_mat[7]; x.t[2] = (double)scale_mat[11]; }
    x.has_trans = trans_mat != nullptr;
    if (trans_mat) for (int k = 0; k < 12; ++k) x.T[k] = (double)trans_mat[k];
    hipLaunchKernelGGL(k_pack_vertices, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, verts_idx, n, x, rgb, vertex_records, rgb ? 16 : 12);
    return check_launch("mesh_pack_vertices");
}

int o2345_mc_verts_to_world(double* verts, long long n, int grid_R, const double* extent_host, const double* offset_host, void* stream) {
    O2345_REQUIRE(extent_host && offset_host && grid_R >= 2, "mc_verts_to_world: bad extent / offset / resolution");
    if (n <= 0) return 0;
    O2345_REQUIRE(verts, "mc_verts_to_world: null pointer");
    const double* e = extent_host; const double* o = offset_host;
    hipLaunchKernelGGL(k_verts_to_world, dim3(cdiv(3 * n, 256)), dim3(256), 0, (hipStream_t)stream, verts, 3 * n, (double)grid_R - 1.0, e[0], e[1], e[2], o[0], o[1], o[2]);
    return check_launch("mc_verts_to_world");
}

// HOST arrays in, HOST records out: the same two record layouts for a mesh that is already on the host (what trimesh.Trimesh(vertices, faces,
// vertex_colors).export() receives from the reference's trainer).  vertices fp64 [n,3] -> float32, colours uint8 [n,3|4] (or NULL) -> rgba with
// alpha 255 when 3 channels are given; faces int64 [m,3] -> int32.  Up to four threads; no device work.
int o2345_ply_records_host(const double* vertices, long long n, const uint8_t* colors, int color_channels, const long long* faces, long long m,
                           uint8_t* vertex_records, uint8_t* face_records) {
    O2345_REQUIRE((n == 0 || (vertices && vertex_records)) && (m == 0 || (faces && face_records)), "ply_records_host: null pointer");
    O2345_REQUIRE(!colors || color_channels == 3 || color_channels == 4, "ply_records_host: 3 or 4 colour channels");
    const int vs = colors ? 16 : 12;
    auto vert = [=](long long i0, long long i1) {
        for (long long i = i0; i < i1; ++i) {
            uint8_t* o = vertex_records + i * vs;
            const float f[3] = {(float)vertices[3 * i], (float)vertices[3 * i + 1], (float)vertices[3 * i + 2]};
            memcpy(o, f, 12);
            if (colors) {
                const uint8_t* c = colors + i * color_channels;
                o[12] = c[0]; o[13] = c[1]; o[14] = c[2]; o[15] = color_channels == 4 ? c[3] : 255;
            }
        }
    };
    auto face = [=](long long i0, long long i1) {
        for (long long i = i0; i < i1; ++i) {
            uint8_t* o = face_records + i * 13;
            o[0] = 3;
            const int32_t f[3] = {(int32_t)faces[3 * i], (int32_t)faces[3 * i + 1], (int32_t)faces[3 * i + 2]};
            memcpy(o + 1, f, 12);
        }
    };
    const int nt = (n + m) > (1 << 16) ? 4 : 1;
    if (nt == 1) { vert(0, n); face(0, m); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=] { vert(n * t / nt, n * (t + 1) / nt); face(m * t / nt, m * (t + 1) / nt); });
    for (auto& t : th) t.join();
    return 0;
}

// tris: device [m,3] int64 (index_bytes 8) or int32 (4); face_records: device, m * 13 bytes
int o2345_mesh_pack_faces(const void* tris, int index_bytes, long long m, uint8_t* face_records, void* stream) {
    O2345_REQUIRE(index_bytes == 4 || index_bytes == 8, "mesh_pack_faces: index_bytes must be 4 or 8");
    if (m <= 0) return 0;
    O2345_REQUIRE(tris && face_records, "mesh_pack_faces: null pointer");
    if (index_bytes == 8) hipLaunchKernelGGL(k_pack_faces<long long>, dim3(cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)tris, m, face_records);
    else hipLaunchKernelGGL(k_pack_faces<int32_t>, dim3(cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, (const int32_t*)tris, m, face_records);
    return check_launch("mesh_pack_faces");
}

}  // extern "C"

// o2345_preload (csrc/api.cpp): querying one kernel makes the HIP runtime load this translation unit's code object on the current device
namespace o2345 {
int preload_mesh_pack() {
    hipFuncAttributes at;
    return (int)hipFuncGetAttributes(&at, (const void*)(k_verts_to_world));
}
}  // namespace o2345
