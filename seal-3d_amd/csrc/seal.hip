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


// ---- colour edit of the bbox tool (seal_utils.py:48-58 map_color, :739-769 modify_hsv / modify_rgb, color_utils.py:33-66) ----
// The renderers re-colour only the samples the proxy moved: `rgbs[mask] = map_color(.., rgbs[mask])` (SealNeRF/renderer.py:316,
// 396-399) — in torch a boolean gather (host sync), ~40 elementwise launches with masked scatters, and a scatter back.  Here:
// one pass for the `hsv` offsets alone; with an `rgb` target two passes, because each sample keeps its brightness offset from
// the MEAN brightness of the moved samples of the batch (a batch statistic): k_seal_color_stats sums the (hsv-shifted) value
// channel of the masked rows — as 64-bit fixed point 2^-32, so the sum does not depend on the order — then k_seal_color_apply.
struct SealColor {
    float hsv[3], target_hs[2], target_v, light;
    uint32_t has_hsv, has_rgb;
};
__device__ __forceinline__ float floor_mod(float a, float m) { return a - m * floorf(a / m); }  // torch's `%` (remainder)
__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
    // color_utils.py:33-46: hue from the FIRST maximal channel (torch.max's tie rule), grey -> 0
    const float cmax = fmaxf(r, fmaxf(g, b)), cmin = fminf(r, fminf(g, b)), delta = cmax - cmin;
    const int idx = (r >= g && r >= b) ? 0 : (g >= b ? 1 : 2);
    if (delta == 0.0f) h = 0.0f;
    else if (idx == 0) h = floor_mod((g - b) / delta, 6.0f);
    else if (idx == 1) h = (b - r) / delta + 2.0f;
    else h = (r - g) / delta + 4.0f;
    h = h / 6.0f;
    s = cmax == 0.0f ? 0.0f : delta / cmax;
    v = cmax;
}
__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
    // color_utils.py:49-66: sextant = (h * 6) converted to uint8 (truncation, modulo 256), modulo 6
    const float c = v * s;
    const float x = c * (-fabsf(floor_mod(h * 6.0f, 2.0f) - 1.0f) + 1.0f);
    const float m = v - c;
    const uint32_t k = ((uint32_t)(int)truncf(fminf(fmaxf(h * 6.0f, -2.0e9f), 2.0e9f)) & 0xffu) % 6u;
    r = k == 0 || k == 5 ? c : (k == 1 || k == 4 ? x : 0.0f);
    g = k == 1 || k == 2 ? c : (k == 0 || k == 3 ? x : 0.0f);
    b = k == 3 || k == 4 ? c : (k == 2 || k == 5 ? x : 0.0f);
    r += m; g += m; b += m;
}
__global__ void k_seal_zero2(unsigned long long* w) { if (threadIdx.x < 2) w[threadIdx.x] = 0ull; }
template <typename T> __device__ __forceinline__ float color_ld(const T* p) { return (float)*p; }
template <typename T>
__device__ __forceinline__ void shifted_hsv(const T* __restrict__ rgb, size_t i, const SealColor& c, float& h, float& s, float& v) {
    float r = color_ld(rgb + i * 3), g = color_ld(rgb + i * 3 + 1), b = color_ld(rgb + i * 3 + 2);
    rgb2hsv(r, g, b, h, s, v);
    if (c.has_hsv) {
        if (c.has_rgb) {  // the `rgb` step converts the hsv step's RGB result again
            hsv2rgb(h + c.hsv[0], s + c.hsv[1], v + c.hsv[2], r, g, b);
            rgb2hsv(r, g, b, h, s, v);
        } else {
            h += c.hsv[0]; s += c.hsv[1]; v += c.hsv[2];
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(256) k_seal_color_stats(const T* __restrict__ rgb, const uint8_t* __restrict__ mask, uint32_t M,
                                                          SealColor c, const int32_t* __restrict__ n_valid,
                                                          unsigned long long* __restrict__ stats) {
    const uint32_t Mv = valid_rows(M, n_valid);
    long long sum = 0;
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < Mv; i += gridDim.x * 256) {
        if (!mask[i]) continue;
        float h, s, v;
        shifted_hsv(rgb, i, c, h, s, v);
        sum += (long long)rintf(fminf(fmaxf(v, -1.0e6f), 1.0e6f) * 1048576.0f);  // v * 2^20 (|v| <= 1e6: 2^40 per term, 2^32 terms fit)
        cnt++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicAdd(stats, (unsigned long long)sum);
        atomicAdd(stats + 1, (unsigned long long)cnt);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) k_seal_color_apply(const T* __restrict__ rgb, const uint8_t* __restrict__ mask, uint32_t M,
                                                          SealColor c, const int32_t* __restrict__ n_valid,
                                                          const unsigned long long* __restrict__ stats, T* __restrict__ out) {
    const uint32_t Mv = valid_rows(M, n_valid);
    float mean = 0.0f;
    if (c.has_rgb) {
        const long long sum = (long long)stats[0];
        const unsigned long long cnt = stats[1];
        mean = cnt ? (float)((double)sum / 1048576.0 / (double)cnt) : 0.0f;
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < Mv; i += gridDim.x * 256) {
        if (!mask[i]) {
            if (out != rgb) { out[(size_t)i * 3] = rgb[(size_t)i * 3]; out[(size_t)i * 3 + 1] = rgb[(size_t)i * 3 + 1]; out[(size_t)i * 3 + 2] = rgb[(size_t)i * 3 + 2]; }
            continue;
        }
        float h, s, v, r, g, b;
        shifted_hsv(rgb, i, c, h, s, v);
        if (c.has_rgb) {
            const float val = fminf(1.0f, fmaxf(0.0f, c.target_v + (v - mean) + c.light));
            hsv2rgb(c.target_hs[0], c.target_hs[1], val, r, g, b);
        } else {
            hsv2rgb(h, s, v, r, g, b);
        }
        out[(size_t)i * 3] = (T)r; out[(size_t)i * 3 + 1] = (T)g; out[(size_t)i * 3 + 2] = (T)b;
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

S3D_EXPORT int s3d_seal_map_color(const void* rgbs, const uint8_t* mask, uint32_t M, int dtype, const float* hsv, const float* rgb_target,
                                  float light_offset, void* out, void* stats, const int32_t* n_valid, s3d_stream_t stream) {
    if (M == 0) return S3D_OK;
    S3D_REQUIRE(rgbs && mask && out, "seal_map_color: null pointer");
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "seal_map_color: dtype must be f32 or f16");
    S3D_REQUIRE(hsv || rgb_target, "seal_map_color: neither an hsv offset nor an rgb target");
    S3D_REQUIRE(!rgb_target || stats, "seal_map_color: the rgb edit needs the 16-byte statistics buffer");
    SealColor c;
    memset(&c, 0, sizeof(c));
    if (hsv) { c.has_hsv = 1; for (int k = 0; k < 3; k++) c.hsv[k] = hsv[k]; }
    if (rgb_target) {
        // hue / saturation / value of the target colour (host: rgb2hsv of one triple, the device function's arithmetic)
        const float r = rgb_target[0], g = rgb_target[1], b = rgb_target[2];
        const float cmax = std::max(r, std::max(g, b)), cmin = std::min(r, std::min(g, b)), delta = cmax - cmin;
        float h;
        if (delta == 0.0f) h = 0.0f;
        else if (r >= g && r >= b) { const float a = (g - b) / delta; h = a - 6.0f * floorf(a / 6.0f); }
        else if (g >= b) h = (b - r) / delta + 2.0f;
        else h = (r - g) / delta + 4.0f;
        c.has_rgb = 1;
        c.target_hs[0] = h / 6.0f;
        c.target_hs[1] = cmax == 0.0f ? 0.0f : delta / cmax;
        c.target_v = cmax;
        c.light = light_offset;
    }
    hipStream_t st = as_stream(stream);
    const dim3 grid(std::min<uint32_t>(div_up<uint32_t>(M, 256), 2048u)), block(256);
    auto* sw = reinterpret_cast<unsigned long long*>(stats);
    if (c.has_rgb) {
        hipLaunchKernelGGL(k_seal_zero2, dim3(1), dim3(64), 0, st, sw);  // (a kernel, not a memset node: see tensorf.hip's bins)
        if (dtype == S3D_F32) hipLaunchKernelGGL(k_seal_color_stats<float>, grid, block, 0, st, (const float*)rgbs, mask, M, c, n_valid, sw);
        else hipLaunchKernelGGL(k_seal_color_stats<_Float16>, grid, block, 0, st, (const _Float16*)rgbs, mask, M, c, n_valid, sw);
    }
    if (dtype == S3D_F32) hipLaunchKernelGGL(k_seal_color_apply<float>, grid, block, 0, st, (const float*)rgbs, mask, M, c, n_valid, sw, (float*)out);
    else hipLaunchKernelGGL(k_seal_color_apply<_Float16>, grid, block, 0, st, (const _Float16*)rgbs, mask, M, c, n_valid, sw, (_Float16*)out);
    return check_launch("seal_map_color");
}
