// tensorf.hip — TensoRF vector-matrix features (tensoRF/network.py:112-153 of the reference: get_sigma_feat / get_color_feat).
//
// The reference samples, per point, three plane factors [1,R,H,W] and three line factors [1,R,D,1] with twelve
// F.grid_sample calls (bilinear, zeros padding, align_corners=True; the lines as "fake 2-D" images of width 1), stacks
// and concatenates the [R,N] results, multiplies and (for the density) sums them.  Here one lane does all of it for
// one point: out[N] = sum_i sum_r plane_i[r](u_i, v_i) * line_i[r](w_i)  (reduce = 1), or the products themselves as
// [sum_i R_i, N] (reduce = 0: the operand the reference transposes into basis_mat).  Same interpolation arithmetic as
// torch's grid sampler: index = ((c + 1) / 2) * (size - 1), corner weights as products of the distances to the opposite
// corner, corners accumulated in the order nw, ne, sw, se, out-of-range corners skipped.
#include "s3d_common.hpp"

namespace s3d {
namespace {

struct VmFactors {
    const float* plane[3];
    const float* line[3];
    uint32_t rank[3];
    uint32_t W[3], H[3], Dn[3];   // plane i is [rank, H, W] (W <-> coordinate mat_ids[i][0], H <-> mat_ids[i][1]); line i [rank, Dn]
    uint32_t cu[3], cv[3], cw[3]; // coordinate index of u (-> W), v (-> H), w (-> line)
    uint32_t row0[3];             // first output row of component i (reduce = 0)
};

__device__ __forceinline__ float unnormalize(float c, uint32_t size) { return ((c + 1.0f) / 2.0f) * (float)(size - 1); }

template <bool REDUCE>
__global__ void __launch_bounds__(256) k_vm_features(const float* __restrict__ x, uint32_t N, VmFactors f, float* __restrict__ out) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float p[3] = {x[(size_t)n * 3], x[(size_t)n * 3 + 1], x[(size_t)n * 3 + 2]};
    float total = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) {
        const int W = (int)f.W[i], H = (int)f.H[i], Dn = (int)f.Dn[i];
        const float ix = unnormalize(p[f.cu[i]], f.W[i]), iy = unnormalize(p[f.cv[i]], f.H[i]), iz = unnormalize(p[f.cw[i]], f.Dn[i]);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        // corner weights (grid_sampler: nw = (ix_se - ix)(iy_se - iy), ne = (ix - ix_sw)(iy_sw - iy), ...)
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
        const float lz1 = iz - fz, lz0 = (fz + 1.0f) - iz;
        // (non-finite coordinates: every corner is out of range, like the within-bounds tests of the reference kernel)
        const bool okx = fabsf(ix) < 1e9f, oky = fabsf(iy) < 1e9f, okz = fabsf(iz) < 1e9f;
        const int x0 = okx ? (int)fx : -2, y0 = oky ? (int)fy : -2, z0 = okz ? (int)fz : -2;
        const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W;
        const bool by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
        const bool bz0 = z0 >= 0 && z0 < Dn, bz1 = z0 + 1 >= 0 && z0 + 1 < Dn;
        const float* P = f.plane[i];
        const float* Lq = f.line[i];
        const size_t plane_stride = (size_t)H * W;
        const int o_nw = y0 * W + x0;
        float comp = 0.0f;
        for (uint32_t r = 0; r < f.rank[i]; r++) {
            const float* pr = P + r * plane_stride;
            const float* lr = Lq + (size_t)r * Dn;
            // gathers first, then the fixed-order accumulation
            const float v_nw = (bx0 && by0) ? pr[o_nw] : 0.0f, v_ne = (bx1 && by0) ? pr[o_nw + 1] : 0.0f;
            const float v_sw = (bx0 && by1) ? pr[o_nw + W] : 0.0f, v_se = (bx1 && by1) ? pr[o_nw + W + 1] : 0.0f;
            const float l0 = bz0 ? lr[z0] : 0.0f, l1 = bz1 ? lr[z0 + 1] : 0.0f;
            float m = 0.0f;
            if (bx0 && by0) m += v_nw * nw;
            if (bx1 && by0) m += v_ne * ne;
            if (bx0 && by1) m += v_sw * sw;
            if (bx1 && by1) m += v_se * se;
            float l = 0.0f;
            if (bz0) l += l0 * lz0;
            if (bz1) l += l1 * lz1;
            const float prod = m * l;
            if (REDUCE) comp += prod;
            else out[(size_t)(f.row0[i] + r) * N + n] = prod;
        }
        total += comp;
    }
    if (REDUCE) out[n] = total;
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_vm_features_forward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                                       const uint32_t* rank, const uint32_t* resolution, int reduce, float* out,
                                       s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && planes && lines && rank && resolution && out, "vm_features_forward: null pointer");
    static const uint32_t mat_ids[3][2] = {{0, 1}, {0, 2}, {1, 2}};  // tensoRF/network.py:37-38
    static const uint32_t vec_ids[3] = {2, 1, 0};
    VmFactors f;
    uint32_t row = 0;
    for (uint32_t i = 0; i < 3; i++) {
        S3D_REQUIRE(planes[i] && lines[i] && rank[i] > 0 && resolution[i] > 0, "vm_features_forward: empty factor %u", i);
        f.plane[i] = planes[i];
        f.line[i] = lines[i];
        f.rank[i] = rank[i];
        f.cu[i] = mat_ids[i][0];
        f.cv[i] = mat_ids[i][1];
        f.cw[i] = vec_ids[i];
        f.W[i] = resolution[mat_ids[i][0]];
        f.H[i] = resolution[mat_ids[i][1]];
        f.Dn[i] = resolution[vec_ids[i]];
        f.row0[i] = row;
        row += rank[i];
    }
    S3D_REQUIRE((uint64_t)row * N < (1ull << 32) * 4, "vm_features_forward: output too large");
    const dim3 grid(div_up<uint32_t>(N, 256)), block(256);
    if (reduce) hipLaunchKernelGGL((k_vm_features<true>), grid, block, 0, as_stream(stream), x, N, f, out);
    else hipLaunchKernelGGL((k_vm_features<false>), grid, block, 0, as_stream(stream), x, N, f, out);
    return check_launch("vm_features_forward");
}
