// seal.hip — Seal-3D's bounding-box proxy mapper on the device (SealNeRF/seal_utils.py:132-153 map_mask, :237-279
// map_to_origin, :630-685 the two-ray Moller-Trumbore inside test): for every sample point of the edited scene decide whether
// it lies in the edited region and, if so, carry it (and its view direction) back to the source space.
// The reference does this with boolean masks, gathers, [2m x F] einsum temporaries and two host synchronisations
// (`mask.any()`) inside every teacher render; here it is one pass, one lane per point, no temporaries, no sync:
//   mask_i = all(p != 0) && any_b( lo_b < p < hi_b )  &&  hit(p, +d) && hit(p, -d)      (d = the reference's fixed test axis)
//   p'     = (T^-1 [p;1] - c) * (1/s) + c ,  dir' = R^-1 dir        for masked points; optionally the source box is emptied
//   (points inside `empty_bound` are sent to `map_source`) before that, as in the reference.
#include "s3d_common.hpp"
#include <algorithm>

namespace s3d {
namespace {

constexpr uint32_t kSealMaxTris = 32;   // a bbox tool has 12 (one box) or 24 (boundType 'both')
constexpr uint32_t kSealMaxBounds = 2;
struct SealMap {
    float tri[kSealMaxTris][3][3];
    float lo[kSealMaxBounds][3], hi[kSealMaxBounds][3];
    float T[3][4];       // inverse transform (rows of the 4x4)
    float R[3][3];       // inverse rotation
    float inv_scale[3], center[3];
    float empty_lo[3], empty_hi[3], source[3];
    uint32_t n_tris, n_bounds, has_source;
};

// do the rays (o, d) AND (o, -d) each hit a triangle?  seal_utils.py:630-665, expression by expression, for both rays inside one
// walk over the triangles: the kernel is a chain of scalar loads (the triangles live in the kernel arguments) and ~40
// dependent flops per triangle and ray — one walk instead of two, four triangles' loads in flight (19.7 -> 16 us per teacher
// sample batch)
__device__ __forceinline__ bool hit_both(const SealMap& m, float ox, float oy, float oz, float dx, float dy, float dz) {
    bool hit_p = false, hit_n = false;
    auto test = [&](float dx_, float dy_, float dz_, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z, float nx, float ny,
                    float nz, float ax, float ay, float az) {
        const float invdet = 1.0f / -((dx_ * nx + dy_ * ny + dz_ * nz) + 1e-8f);
        const float cx = ay * dz_ - az * dy_, cy = az * dx_ - ax * dz_, cz = ax * dy_ - ay * dx_;  // cross(A0, d)
        const float u = (cx * e2x + cy * e2y + cz * e2z) * invdet;
        const float v = -(cx * e1x + cy * e1y + cz * e1z) * invdet;
        const float t = (ax * nx + ay * ny + az * nz) * invdet;
        return (t >= 0.0f) && (u >= 0.0f) && (v >= 0.0f) && ((u + v) <= 1.0f);
    };
#pragma unroll 4
    for (uint32_t f = 0; f < m.n_tris; f++) {
        const float* v0 = m.tri[f][0];
        const float e1x = m.tri[f][1][0] - v0[0], e1y = m.tri[f][1][1] - v0[1], e1z = m.tri[f][1][2] - v0[2];
        const float e2x = m.tri[f][2][0] - v0[0], e2y = m.tri[f][2][1] - v0[1], e2z = m.tri[f][2][2] - v0[2];
        const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
        const float ax = ox - v0[0], ay = oy - v0[1], az = oz - v0[2];
        hit_p |= test(dx, dy, dz, e1x, e1y, e1z, e2x, e2y, e2z, nx, ny, nz, ax, ay, az);
        hit_n |= test(-dx, -dy, -dz, e1x, e1y, e1z, e2x, e2y, e2z, nx, ny, nz, ax, ay, az);
    }
    return hit_p && hit_n;
}

__global__ void __launch_bounds__(256) k_seal_map(const float* __restrict__ points, const float* __restrict__ dirs, uint32_t M,
                                                  SealMap m, float* __restrict__ out_p, float* __restrict__ out_d,
                                                  uint8_t* __restrict__ mask, const int32_t* __restrict__ n_valid) {
    // (grid-stride: a padded batch of N x max_steps rows with 2e5 of them filled would otherwise dispatch 16,000 workgroups
    //  that leave at once — ~13 of this kernel's 19 us in the teacher's proxy render)
    const uint32_t Mv = valid_rows(M, n_valid);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < Mv; i += gridDim.x * 256) {
    const float px = points[(size_t)i * 3], py = points[(size_t)i * 3 + 1], pz = points[(size_t)i * 3 + 2];
    bool in = false;
    for (uint32_t b = 0; b < m.n_bounds; b++)
        in |= (m.hi[b][0] > px) && (px > m.lo[b][0]) && (m.hi[b][1] > py) && (py > m.lo[b][1]) && (m.hi[b][2] > pz) && (pz > m.lo[b][2]);
    in = in && (px != 0.0f) && (py != 0.0f) && (pz != 0.0f);  // `points.all(1)` of the reference (seal_utils.py:141)
    if (in) {
        const float tx = 0.4395064455f, ty = 0.617598629942f, tz = 0.652231566745f;  // seal_utils.py:676-678
        in = hit_both(m, px, py, pz, tx, ty, tz);
    }
    float ox = px, oy = py, oz = pz;
    if (m.has_source && (m.empty_hi[0] > px) && (px > m.empty_lo[0]) && (m.empty_hi[1] > py) && (py > m.empty_lo[1]) &&
        (m.empty_hi[2] > pz) && (pz > m.empty_lo[2])) {
        ox = m.source[0]; oy = m.source[1]; oz = m.source[2];
    }
    if (in) {
        const float mx = m.T[0][0] * px + m.T[0][1] * py + m.T[0][2] * pz + m.T[0][3];
        const float my = m.T[1][0] * px + m.T[1][1] * py + m.T[1][2] * pz + m.T[1][3];
        const float mz = m.T[2][0] * px + m.T[2][1] * py + m.T[2][2] * pz + m.T[2][3];
        ox = (mx - m.center[0]) * m.inv_scale[0] + m.center[0];
        oy = (my - m.center[1]) * m.inv_scale[1] + m.center[1];
        oz = (mz - m.center[2]) * m.inv_scale[2] + m.center[2];
    }
    out_p[(size_t)i * 3] = ox; out_p[(size_t)i * 3 + 1] = oy; out_p[(size_t)i * 3 + 2] = oz;
    if (dirs && out_d) {
        const float dx = dirs[(size_t)i * 3], dy = dirs[(size_t)i * 3 + 1], dz = dirs[(size_t)i * 3 + 2];
        float rx = dx, ry = dy, rz = dz;
        if (in) {
            rx = m.R[0][0] * dx + m.R[0][1] * dy + m.R[0][2] * dz;
            ry = m.R[1][0] * dx + m.R[1][1] * dy + m.R[1][2] * dz;
            rz = m.R[2][0] * dx + m.R[2][1] * dy + m.R[2][2] * dz;
        }
        out_d[(size_t)i * 3] = rx; out_d[(size_t)i * 3 + 1] = ry; out_d[(size_t)i * 3 + 2] = rz;
    }
    mask[i] = in ? 1 : 0;
    }
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_seal_bbox_map(const float* points, const float* dirs, uint32_t M, const float* triangles, uint32_t n_tris,
                                 const float* bounds, uint32_t n_bounds, const float* inv_transform, const float* inv_rotation,
                                 const float* inv_scale, const float* center, const float* empty_bound, const float* map_source,
                                 float* out_points, float* out_dirs, uint8_t* mask, const int32_t* n_valid,
                                 s3d_stream_t stream) {
    if (M == 0) return S3D_OK;
    S3D_REQUIRE(points && triangles && bounds && inv_transform && inv_rotation && inv_scale && center && out_points && mask,
                "seal_bbox_map: null pointer");
    S3D_REQUIRE(n_tris >= 1 && n_tris <= kSealMaxTris, "seal_bbox_map: 1..%u triangles (bbox tool), got %u", kSealMaxTris, n_tris);
    S3D_REQUIRE(n_bounds >= 1 && n_bounds <= kSealMaxBounds, "seal_bbox_map: 1..%u bounds, got %u", kSealMaxBounds, n_bounds);
    S3D_REQUIRE((dirs == nullptr) == (out_dirs == nullptr), "seal_bbox_map: dirs and out_dirs go together");
    S3D_REQUIRE((empty_bound == nullptr) == (map_source == nullptr), "seal_bbox_map: empty_bound and map_source go together");
    SealMap m;
    memset(&m, 0, sizeof(m));
    memcpy(m.tri, triangles, sizeof(float) * 9 * n_tris);           // HOST arrays: the mapper is a handful of constants
    for (uint32_t b = 0; b < n_bounds; b++)
        for (int d = 0; d < 3; d++) { m.lo[b][d] = bounds[(b * 2 + 0) * 3 + d]; m.hi[b][d] = bounds[(b * 2 + 1) * 3 + d]; }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 4; c++) m.T[r][c] = inv_transform[r * 4 + c];
        for (int c = 0; c < 3; c++) m.R[r][c] = inv_rotation[r * 3 + c];
        m.inv_scale[r] = inv_scale[r];
        m.center[r] = center[r];
    }
    if (empty_bound) {
        for (int d = 0; d < 3; d++) { m.empty_lo[d] = empty_bound[d]; m.empty_hi[d] = empty_bound[3 + d]; m.source[d] = map_source[d]; }
        m.has_source = 1;
    }
    m.n_tris = n_tris; m.n_bounds = n_bounds;
    hipLaunchKernelGGL(k_seal_map, dim3(std::min<uint32_t>(div_up<uint32_t>(M, 256), 2048u)), dim3(256), 0, as_stream(stream), points, dirs, M, m, out_points,
                       out_dirs, mask, n_valid);
    return check_launch("seal_bbox_map");
}
