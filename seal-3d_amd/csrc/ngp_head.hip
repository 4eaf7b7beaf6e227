// ngp_head.hip — the elementwise glue between the two fused MLPs of the NGP network (nerf/network_ff.py:55-96 of the
// reference: `sigma = trunc_exp(h[..., 0]); geo = h[..., 1:]; d = SH(d); rgb = sigmoid(color_net(cat[d, geo, 0]))`).
// In torch that is ~15 kernel launches per direction (slice, exp, SH, zeros, cat with type promotion, half cast, sigmoid,
// float casts, and their backward nodes), each a full pass over [M, 16..32] activations.  Here it is two streaming kernels
// per direction with the same arithmetic and the same rounding points:
//   mid  forward : h [B,16] f16, dirs [B,3] f32  ->  sigma [B] f32 = exp(float(h0))            (activation.py:8-11)
//                                                    cin [B,32] f16 = [half(SH_4(d)) | h1..h15 | 0]
//   mid  backward: d_cin [B,32] f16, d_sigma [B] f32, h -> d_h [B,16] f16 = [half(d_sigma * exp(clamp(h0,-15,15))) | d_cin[16..30]]
//   rgb  forward : out [B,16] f16 -> rgb [B,3] f32 = float(half(sigmoid(float(out[:3]))))       (torch.sigmoid on fp16)
//   rgb  backward: d_rgb [B,3] f32, rgb -> d_out [B,16] f16 = [half(float(half(d_rgb)) * y(1-y)) | 0 x 13]
// One lane = one point; every lane reads/writes whole 16-byte pieces of its rows.
#include "s3d_common.hpp"
#include "sh_eval.hpp"

namespace s3d {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) k_ngp_mid_forward(const _Float16* __restrict__ h, const float* __restrict__ dirs,
                                                         uint32_t B, ShNorm K, float* __restrict__ sigma,
                                                         _Float16* __restrict__ cin, const int32_t* __restrict__ n_valid) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= valid_rows(B, n_valid)) return;
    const h8 h0 = *reinterpret_cast<const h8*>(h + (size_t)b * 16), h1 = *reinterpret_cast<const h8*>(h + (size_t)b * 16 + 8);
    const float x = dirs[(size_t)b * 3], y = dirs[(size_t)b * 3 + 1], z = dirs[(size_t)b * 3 + 2];
    float o[16], j0[1], j1[1], j2[1];
    sh_eval<4, false>(x, y, z, K, o, j0, j1, j2);
    sigma[b] = expf((float)h0[0]);
    h8 c0, c1, c2, c3;
#pragma unroll
    for (int i = 0; i < 8; i++) { c0[i] = (_Float16)o[i]; c1[i] = (_Float16)o[8 + i]; }
#pragma unroll
    for (int i = 0; i < 7; i++) { c2[i] = h0[i + 1]; c3[i] = h1[i + 1]; }
    c2[7] = h1[0];
    c3[7] = (_Float16)0.0f;
    h8* dst = reinterpret_cast<h8*>(cin + (size_t)b * 32);
    dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3;
}

__global__ void __launch_bounds__(256) k_ngp_mid_backward(const _Float16* __restrict__ d_cin, const float* __restrict__ d_sigma,
                                                          const _Float16* __restrict__ h, uint32_t B,
                                                          _Float16* __restrict__ d_h, const int32_t* __restrict__ n_valid) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= valid_rows(B, n_valid)) return;
    const h8 g2 = *reinterpret_cast<const h8*>(d_cin + (size_t)b * 32 + 16), g3 = *reinterpret_cast<const h8*>(d_cin + (size_t)b * 32 + 24);
    const float h0 = (float)h[(size_t)b * 16];
    const float gs = d_sigma ? d_sigma[b] * expf(fminf(15.0f, fmaxf(-15.0f, h0))) : 0.0f;  // activation.py:13-16
    h8 o0, o1;
    o0[0] = (_Float16)gs;
#pragma unroll
    for (int i = 0; i < 7; i++) { o0[i + 1] = g2[i]; o1[i + 1] = g3[i]; }
    o1[0] = g2[7];
    h8* dst = reinterpret_cast<h8*>(d_h + (size_t)b * 16);
    dst[0] = o0; dst[1] = o1;
}

// Two-encoder network of Seal-3D (nerf/network.py:99-128): the colour net's input is `cat[SH_4(d), geo_feat, encoder_color(x)]`
// (16 + 15 + 32 = 63 columns, padded to 64 with a zero so that the MFMA MLP reads whole 16-byte pieces).  The second
// encoder's features arrive — and their gradient leaves — in the grid kernels' own level-major layout [16][B][2]: no
// permute copy on either side.
//   mid2 forward : h [B,16], dirs [B,3] f32, enc [16][B][2] -> sigma [B] f32, cin [B,64] = [half(SH) | h1..h15 | enc | 0]
//   mid2 backward: d_cin [B,64], d_sigma, h -> d_h [B,16] (as mid backward), d_enc [16][B][2] = d_cin[31..62]
__global__ void __launch_bounds__(256) k_ngp_mid2_forward(const _Float16* __restrict__ h, const float* __restrict__ dirs,
                                                          const _Float16* __restrict__ enc, uint32_t B, ShNorm K,
                                                          float* __restrict__ sigma, _Float16* __restrict__ cin,
                                                          const int32_t* __restrict__ n_valid) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= valid_rows(B, n_valid)) return;
    const h8 h0 = *reinterpret_cast<const h8*>(h + (size_t)b * 16), h1 = *reinterpret_cast<const h8*>(h + (size_t)b * 16 + 8);
    const float x = dirs[(size_t)b * 3], y = dirs[(size_t)b * 3 + 1], z = dirs[(size_t)b * 3 + 2];
    float o[16], j0[1], j1[1], j2[1];
    sh_eval<4, false>(x, y, z, K, o, j0, j1, j2);
    sigma[b] = expf((float)h0[0]);
    _Float16 row[64];
#pragma unroll
    for (int i = 0; i < 16; i++) row[i] = (_Float16)o[i];
#pragma unroll
    for (int i = 0; i < 7; i++) { row[16 + i] = h0[i + 1]; row[24 + i] = h1[i + 1]; }
    row[23] = h1[0];
#pragma unroll
    for (int l = 0; l < 16; l++) {  // one 4-byte load per level, consecutive lanes on consecutive addresses
        const __half2 e = *reinterpret_cast<const __half2*>(enc + ((size_t)l * B + b) * 2);
        row[31 + 2 * l] = (_Float16)__low2half(e);
        row[32 + 2 * l] = (_Float16)__high2half(e);
    }
    row[63] = (_Float16)0.0f;
    h8* dst = reinterpret_cast<h8*>(cin + (size_t)b * 64);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        h8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = row[8 * k + i];
        dst[k] = v;
    }
}

__global__ void __launch_bounds__(256) k_ngp_mid2_backward(const _Float16* __restrict__ d_cin, const float* __restrict__ d_sigma,
                                                           const _Float16* __restrict__ h, uint32_t h_stride, uint32_t B, _Float16* __restrict__ d_h,
                                                           _Float16* __restrict__ d_enc, const int32_t* __restrict__ n_valid) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= valid_rows(B, n_valid)) return;
    _Float16 row[48];  // columns 16..63
    const h8* src = reinterpret_cast<const h8*>(d_cin + (size_t)b * 64 + 16);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const h8 v = src[k];
#pragma unroll
        for (int i = 0; i < 8; i++) row[8 * k + i] = v[i];
    }
    const float h0 = (float)h[(size_t)b * h_stride];
    const float gs = d_sigma ? d_sigma[b] * expf(fminf(15.0f, fmaxf(-15.0f, h0))) : 0.0f;  // activation.py:13-16
    h8 o0, o1;
    o0[0] = (_Float16)gs;
#pragma unroll
    for (int i = 0; i < 7; i++) { o0[i + 1] = row[i]; o1[i + 1] = row[8 + i]; }
    o1[0] = row[7];
    h8* dst = reinterpret_cast<h8*>(d_h + (size_t)b * 16);
    dst[0] = o0; dst[1] = o1;
    if (d_enc) {
#pragma unroll
        for (int l = 0; l < 16; l++)
            *reinterpret_cast<__half2*>(d_enc + ((size_t)l * B + b) * 2) = __halves2half2((__half)row[15 + 2 * l], (__half)row[16 + 2 * l]);
    }
}

__global__ void __launch_bounds__(256) k_ngp_rgb_forward(const _Float16* __restrict__ out, uint32_t B, float* __restrict__ rgb,
                                                         const int32_t* __restrict__ n_valid) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= valid_rows(B, n_valid)) return;
    const _Float16* o = out + (size_t)b * 16;
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[(size_t)b * 3 + c] = (float)(_Float16)(1.0f / (1.0f + expf(-(float)o[c])));
}

__global__ void __launch_bounds__(256) k_ngp_rgb_backward(const float* __restrict__ d_rgb, const float* __restrict__ rgb,
                                                          uint32_t B, _Float16* __restrict__ d_out,
                                                          const int32_t* __restrict__ n_valid) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= valid_rows(B, n_valid)) return;
    h8 o0, o1;
#pragma unroll
    for (int i = 0; i < 8; i++) { o0[i] = (_Float16)0.0f; o1[i] = (_Float16)0.0f; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float y = rgb[(size_t)b * 3 + c];
        o0[c] = (_Float16)((float)(_Float16)d_rgb[(size_t)b * 3 + c] * (y * (1.0f - y)));
    }
    h8* dst = reinterpret_cast<h8*>(d_out + (size_t)b * 16);
    dst[0] = o0; dst[1] = o1;
}

// ---- background compositing + MSE loss of one ray batch (nerf/renderer.py:316 `image + (1 - weights_sum) * bg_color`,
// nerf/utils.py:484 MSE) as one kernel per direction: ~14 tiny torch launches otherwise.  One workgroup, fixed-order tree
// reduction (deterministic); N is a ray batch (4,096), not a sample batch.
__global__ void __launch_bounds__(1024) k_bg_mse_forward(const float* __restrict__ image, const float* __restrict__ ws,
                                                         const float* __restrict__ gt, float bg0, float bg1, float bg2, uint32_t N,
                                                         float* __restrict__ loss, const float* __restrict__ grad_loss,
                                                         float* __restrict__ grad_image, float* __restrict__ grad_ws,
                                                         const float* __restrict__ depth, const float* __restrict__ gt_depth,
                                                         float depth_weight) {
    __shared__ float part[16], dpart[16];
    const float bg[3] = {bg0, bg1, bg2};
    // grad_loss given: the gradient the backward kernel would write for that upstream gradient is written here as well
    // (same formula, same order: bit-identical) — under loss scaling the upstream gradient of the loss is known in advance
    const float k = grad_loss ? *grad_loss * (2.0f / (3.0f * (float)N)) : 0.0f;
    float acc = 0.0f;
    for (uint32_t n = threadIdx.x; n < N; n += 1024) {
        const float w = 1.0f - ws[n];
        float gw = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = (image[(size_t)n * 3 + c] + w * bg[c]) - gt[(size_t)n * 3 + c];
            acc += d * d;
            if (grad_loss) {
                const float gd = k * d;
                grad_image[(size_t)n * 3 + c] = gd;
                gw -= gd * bg[c];
            }
        }
        if (grad_loss) grad_ws[n] = gw;
    }
    // Seal's depth term (nerf/utils.py:486-489: `loss += L1Loss(nan_to_num(depth), gt_depth)`): a VALUE only — the reference's
    // composite backward does not propagate the depth gradient (raymarching.py:274 "grad_depth is not used now")
    float dacc = 0.0f;
    if (depth) {
        for (uint32_t n = threadIdx.x; n < N; n += 1024) {
            float dv = depth[n];
            dv = dv != dv ? 0.0f : fminf(fmaxf(dv, -3.402823466e38f), 3.402823466e38f);  // torch.nan_to_num(nan=0.)
            dacc += fabsf(dv - gt_depth[n]);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { acc += __shfl_xor(acc, d, 64); dacc += __shfl_xor(dacc, d, 64); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; dpart[threadIdx.x >> 6] = dacc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f, td = 0.0f;
        for (int w = 0; w < 16; w++) { t += part[w]; td += dpart[w]; }
        t = t / (3.0f * (float)N);
        if (depth) t = t + depth_weight * (td / (float)N);
        *loss = t;
    }
}

// Targets of a teacher-rendered ray batch (SealNeRF/trainer.py:506-586): rgb = nan_to_num(image + (1 - weights_sum) * bg),
// depth = nan_to_num(depth) — nerf/renderer.py:316 + the two nan_to_num(nan=0.) of proxy_truth — written straight into the
// caller's (static) target buffers: one launch for a rsub, a mul, an add, two nan_to_num and two copies.
__device__ __forceinline__ float nan_to_num0(float v) { return v != v ? 0.0f : fminf(fmaxf(v, -3.402823466e38f), 3.402823466e38f); }
__global__ void __launch_bounds__(256) k_bg_targets(const float* __restrict__ image, const float* __restrict__ ws,
                                                    const float* __restrict__ depth, float bg0, float bg1, float bg2, uint32_t N,
                                                    float* __restrict__ out_rgb, float* __restrict__ out_depth) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float bg[3] = {bg0, bg1, bg2};
    const float w = 1.0f - ws[n];
#pragma unroll
    for (int c = 0; c < 3; c++) out_rgb[(size_t)n * 3 + c] = nan_to_num0(image[(size_t)n * 3 + c] + w * bg[c]);
    if (out_depth) out_depth[n] = nan_to_num0(depth[n]);
}

// Local-pretraining loss of Seal-3D on one point chunk (SealNeRF/trainer.py:455-469): L1Loss(sigma) + L1Loss(colour), both
// means, with the shard normalisation of parallel/dist.py (`n_total` = points of the whole chunk):
//   loss = sum |sigma - gt_sigma| / n_total + sum |color - gt_color| / (3 n_total)
// and, for a known upstream gradient (the loss scale), the gradients sign(.) * g / n_total resp. / (3 n_total) in the same
// launch.  Partial sums per workgroup in a fixed order, the last workgroup to arrive (ticket) adds them up in index order:
// the value does not depend on the arrival order.  `partial` = gridDim.x + 1 floats, `ticket` one zeroed word the kernel
// leaves zeroed.
constexpr uint32_t kL1Blocks = 256;
__global__ void __launch_bounds__(256) k_l1_pair(const float* __restrict__ sigma, const float* __restrict__ color,
                                                 const float* __restrict__ gt_sigma, const float* __restrict__ gt_color, uint32_t n,
                                                 uint32_t n_rows, float inv_total, float* __restrict__ loss, const float* __restrict__ grad_loss,
                                                 float* __restrict__ g_sigma, float* __restrict__ g_color, float* __restrict__ partial,
                                                 uint32_t* __restrict__ ticket) {
    __shared__ float part[4];
    __shared__ uint32_t is_last;
    const float gs = grad_loss ? *grad_loss * inv_total : 0.0f, gc = gs * (1.0f / 3.0f);
    float a = 0.0f, b = 0.0f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_rows; i += gridDim.x * 256) {
        if (i >= n) {  // padding rows of the prediction (whole 128-row tiles of the fused network path): no term, zero gradient
            if (grad_loss) {
                g_sigma[i] = 0.0f;
                g_color[(size_t)i * 3] = 0.0f; g_color[(size_t)i * 3 + 1] = 0.0f; g_color[(size_t)i * 3 + 2] = 0.0f;
            }
            continue;
        }
        const float ds = sigma[i] - gt_sigma[i];
        a += fabsf(ds);
        if (grad_loss) g_sigma[i] = ds > 0.0f ? gs : (ds < 0.0f ? -gs : 0.0f);  // torch: sign(0) = 0
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float dc = color[(size_t)i * 3 + c] - gt_color[(size_t)i * 3 + c];
            b += fabsf(dc);
            if (grad_loss) g_color[(size_t)i * 3 + c] = dc > 0.0f ? gc : (dc < 0.0f ? -gc : 0.0f);
        }
    }
    float t = a * inv_total + b * (inv_total * (1.0f / 3.0f));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float s = 0.0f;
    for (uint32_t k = threadIdx.x; k < gridDim.x; k += 256) s += __hip_atomic_load(partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        *loss = (part[0] + part[1]) + (part[2] + part[3]);
        *ticket = 0u;
    }
}

__global__ void __launch_bounds__(256) k_bg_mse_backward(const float* __restrict__ image, const float* __restrict__ ws,
                                                         const float* __restrict__ gt, float bg0, float bg1, float bg2, uint32_t N,
                                                         const float* __restrict__ grad_loss, float* __restrict__ grad_image,
                                                         float* __restrict__ grad_ws) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float bg[3] = {bg0, bg1, bg2};
    const float k = *grad_loss * (2.0f / (3.0f * (float)N));
    const float w = 1.0f - ws[n];
    float gw = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = k * ((image[(size_t)n * 3 + c] + w * bg[c]) - gt[(size_t)n * 3 + c]);
        grad_image[(size_t)n * 3 + c] = d;
        gw -= d * bg[c];
    }
    grad_ws[n] = gw;
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_bg_mse_forward(const float* image, const float* weights_sum, const float* gt, const float* bg_rgb, uint32_t N,
                                  float* loss, const float* grad_loss, float* grad_image, float* grad_weights_sum,
                                  const float* depth, const float* gt_depth, float depth_weight, s3d_stream_t stream) {
    S3D_REQUIRE(image && weights_sum && gt && bg_rgb && loss && N > 0, "bg_mse_forward: null pointer / empty batch");
    S3D_REQUIRE(!grad_loss || (grad_image && grad_weights_sum), "bg_mse_forward: grad_loss needs grad_image and grad_weights_sum");
    S3D_REQUIRE(!depth == !gt_depth, "bg_mse_forward: depth and gt_depth come together");
    hipLaunchKernelGGL(k_bg_mse_forward, dim3(1), dim3(1024), 0, as_stream(stream), image, weights_sum, gt, bg_rgb[0], bg_rgb[1],
                       bg_rgb[2], N, loss, grad_loss, grad_image, grad_weights_sum, depth, gt_depth, depth_weight);
    return check_launch("bg_mse_forward");
}

S3D_EXPORT int s3d_bg_targets(const float* image, const float* weights_sum, const float* depth, const float* bg_rgb, uint32_t N,
                              float* out_rgb, float* out_depth, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(image && weights_sum && bg_rgb && out_rgb && (!out_depth || depth), "bg_targets: null pointer");
    hipLaunchKernelGGL(k_bg_targets, dim3(div_up<uint32_t>(N, 256)), dim3(256), 0, as_stream(stream), image, weights_sum, depth,
                       bg_rgb[0], bg_rgb[1], bg_rgb[2], N, out_rgb, out_depth);
    return check_launch("bg_targets");
}

S3D_EXPORT size_t s3d_l1_pair_workspace_size(void) { return (size_t)(kL1Blocks + 2) * sizeof(float); }

S3D_EXPORT int s3d_l1_pair_loss(const float* sigma, const float* color, const float* gt_sigma, const float* gt_color, uint32_t n,
                                uint32_t n_rows, uint32_t n_total, float* loss, const float* grad_loss, float* grad_sigma, float* grad_color,
                                void* workspace, s3d_stream_t stream) {
    S3D_REQUIRE(sigma && color && gt_sigma && gt_color && loss && workspace && n > 0 && n_total >= n && n_rows >= n,
                "l1_pair_loss: null pointer / empty batch / n_total < n / n_rows < n");
    S3D_REQUIRE(!grad_loss || (grad_sigma && grad_color), "l1_pair_loss: grad_loss needs grad_sigma and grad_color");
    const uint32_t blocks = std::min<uint32_t>(kL1Blocks, div_up<uint32_t>(n_rows, 256));
    float* partial = (float*)workspace;
    uint32_t* ticket = (uint32_t*)(partial + kL1Blocks);  // zeroed once by the caller, left zeroed by every call
    hipLaunchKernelGGL(k_l1_pair, dim3(blocks), dim3(256), 0, as_stream(stream), sigma, color, gt_sigma, gt_color, n,
                       n_rows, 1.0f / (float)n_total, loss, grad_loss, grad_sigma, grad_color, partial, ticket);
    return check_launch("l1_pair_loss");
}

S3D_EXPORT int s3d_bg_mse_backward(const float* image, const float* weights_sum, const float* gt, const float* bg_rgb, uint32_t N,
                                   const float* grad_loss, float* grad_image, float* grad_weights_sum, s3d_stream_t stream) {
    S3D_REQUIRE(image && weights_sum && gt && bg_rgb && grad_loss && grad_image && grad_weights_sum && N > 0,
                "bg_mse_backward: null pointer / empty batch");
    hipLaunchKernelGGL(k_bg_mse_backward, dim3(div_up<uint32_t>(N, 256)), dim3(256), 0, as_stream(stream), image, weights_sum, gt,
                       bg_rgb[0], bg_rgb[1], bg_rgb[2], N, grad_loss, grad_image, grad_weights_sum);
    return check_launch("bg_mse_backward");
}

S3D_EXPORT int s3d_ngp_mid_forward(const uint16_t* h, const float* dirs, uint32_t B, float* sigma, uint16_t* color_in,
                                   const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(h && dirs && sigma && color_in, "ngp_mid_forward: null pointer");
    ShNorm K;
    host_sh_norm(4, K);
    hipLaunchKernelGGL(k_ngp_mid_forward, dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, as_stream(stream), (const _Float16*)h, dirs, B,
                       K, sigma, (_Float16*)color_in, n_valid);
    return check_launch("ngp_mid_forward");
}

S3D_EXPORT int s3d_ngp_mid_backward(const uint16_t* grad_color_in, const float* grad_sigma, const uint16_t* h, uint32_t B,
                                    uint16_t* grad_h, const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad_color_in && h && grad_h, "ngp_mid_backward: null pointer");
    hipLaunchKernelGGL(k_ngp_mid_backward, dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, as_stream(stream),
                       (const _Float16*)grad_color_in, grad_sigma, (const _Float16*)h, B, (_Float16*)grad_h, n_valid);
    return check_launch("ngp_mid_backward");
}

S3D_EXPORT int s3d_ngp_mid2_forward(const uint16_t* h, const float* dirs, const uint16_t* enc_color, uint32_t B, float* sigma,
                                    uint16_t* color_in, const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(h && dirs && enc_color && sigma && color_in, "ngp_mid2_forward: null pointer");
    ShNorm K;
    host_sh_norm(4, K);
    hipLaunchKernelGGL(k_ngp_mid2_forward, dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, as_stream(stream), (const _Float16*)h, dirs,
                       (const _Float16*)enc_color, B, K, sigma, (_Float16*)color_in, n_valid);
    return check_launch("ngp_mid2_forward");
}

S3D_EXPORT int s3d_ngp_mid2_backward(const uint16_t* grad_color_in, const float* grad_sigma, const uint16_t* h, uint32_t h_stride,
                                     uint32_t B, uint16_t* grad_h, uint16_t* grad_enc_color, const int32_t* n_valid,
                                     s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad_color_in && h && grad_h, "ngp_mid2_backward: null pointer");
    S3D_REQUIRE(h_stride == 16 || h_stride == 1, "ngp_mid2_backward: h_stride is 16 (rows of the density network's output) or 1 (its first column alone)");
    hipLaunchKernelGGL(k_ngp_mid2_backward, dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, as_stream(stream),
                       (const _Float16*)grad_color_in, grad_sigma, (const _Float16*)h, h_stride, B, (_Float16*)grad_h,
                       (_Float16*)grad_enc_color, n_valid);
    return check_launch("ngp_mid2_backward");
}

S3D_EXPORT int s3d_ngp_rgb_forward(const uint16_t* out, uint32_t B, float* rgb, const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(out && rgb, "ngp_rgb_forward: null pointer");
    hipLaunchKernelGGL(k_ngp_rgb_forward, dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, as_stream(stream), (const _Float16*)out, B, rgb, n_valid);
    return check_launch("ngp_rgb_forward");
}

S3D_EXPORT int s3d_ngp_rgb_backward(const float* grad_rgb, const float* rgb, uint32_t B, uint16_t* grad_out,
                                    const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad_rgb && rgb && grad_out, "ngp_rgb_backward: null pointer");
    hipLaunchKernelGGL(k_ngp_rgb_backward, dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, as_stream(stream), grad_rgb, rgb, B,
                       (_Float16*)grad_out, n_valid);
    return check_launch("ngp_rgb_backward");
}
