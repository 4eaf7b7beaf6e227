// ffmlp_generic.hip — the fully fused MLP's LAYER-BY-LAYER path for the network shapes the register-resident MFMA kernels of
// ffmlp.hip do not cover: hidden_dim 16 / 256, hidden_dim 128 beyond its LDS budget, input_dim > 64 at widths 32 / 64
// (ffmlp.cu:40-44 dispatches hidden_dim in {16, 32, 64, 128, 256}).
//
// Hand-written MFMA kernels (v_mfma_f32_32x32x16_f16, the lane mapping of ffmlp.hip), no BLAS library:
//   k_layer<TRANSW>   one launch per layer and direction.  A wave owns 64 batch rows (two 32-row B operands per weight
//                     fragment) and walks the output features in blocks of 32; operands come straight from global memory —
//                     the weights of a layer (<= 128 KiB) stay L2-resident, the activations stream once.  Epilogue fused:
//                       forward   Y = act(half(X W^T))                 (the fused kernels' rounding points)
//                       backward  G_in = half(G W) * act'(stored output)   (TRANSW: the A fragments read W transposed)
//   k_wgrad_tile      dW = G^T X with the batch as the MFMA K dimension: one wave per (32 x 32 tile of dW, batch segment),
//                     fp32 partial planes in the caller's workspace; k_wgrad_finish adds the segments in a FIXED order, rounds
//                     to fp16 (optionally accumulating) and raises GradScaler's flag on a non-finite result.  Deterministic.
// Same dense math Y = act(X W^T) per layer that defines ffmlp's numerics (fp16 storage, fp32 accumulation).  This is the
// drop-in path for completeness of the boundary, not the hot path: the networks of the BASELINE configs (hidden 64 / 128,
// in <= 160) never come here.  Buffers: forward_buffer / backward_buffer are plain row-major [n, B, W]; inference uses
// inference_buffer [2, B, W] as ping-pong scratch.
#include "s3d_common.hpp"
#include <algorithm>

namespace s3d {
namespace {

enum { G_ACT_RELU = 0, G_ACT_EXP = 1, G_ACT_SINE = 2, G_ACT_SIGMOID = 3, G_ACT_SQUAREPLUS = 4, G_ACT_SOFTPLUS = 5, G_ACT_NONE = 6 };
constexpr float kActScale = 10.0f;  // utils.h:424-589 (K_ACT)

typedef _Float16 g_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 g_half4 __attribute__((ext_vector_type(4)));
typedef float g_float16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float g_act_fwd(uint32_t a, float x) {
    switch (a) {
        case G_ACT_RELU: return x > 0.0f ? x : 0.0f;
        case G_ACT_EXP: return expf(x);
        case G_ACT_SINE: return sinf(x);
        case G_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case G_ACT_SQUAREPLUS: { const float y = x * kActScale; return 0.5f * (y + sqrtf(y * y + 4)) / kActScale; }
        case G_ACT_SOFTPLUS: return logf(expf(x * kActScale) + 1.0f) / kActScale;
        default: return x;
    }
}
__device__ __forceinline__ float g_act_bwd(uint32_t a, float g, float fwd) {
    switch (a) {
        case G_ACT_RELU: return fwd > 0.0f ? g : 0.0f;
        case G_ACT_EXP: return g * fwd;
        case G_ACT_SIGMOID: return g * (fwd * (1.0f - fwd));
        case G_ACT_SQUAREPLUS: { const float y = fwd * kActScale; return g * (y * y / (y * y + 1)); }
        case G_ACT_SOFTPLUS: return g * (1.0f - expf(-fwd * kActScale));
        default: return g;
    }
}

// K index of element j of lane-half h inside one 16-wide k-step (the operand layout of v_mfma_f32_32x32x16_f16 as ffmlp.hip uses it)
__device__ __forceinline__ uint32_t g_kperm(uint32_t h, uint32_t j) { return (j & 3u) + 8u * (j >> 2) + 4u * h; }
__device__ __forceinline__ g_half8 g_zero8() {
    g_half8 z;
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = (_Float16)0.0f;
    return z;
}
// fragment of k-step s from a row of a row-major matrix (two 8-byte loads: k = 4h..4h+3 and 8+4h..8+4h+3)
__device__ __forceinline__ g_half8 g_frag_row(const _Float16* row, uint32_t s, uint32_t h) {
    const g_half4 lo = *reinterpret_cast<const g_half4*>(row + 16 * s + 4 * h);
    const g_half4 hi = *reinterpret_cast<const g_half4*>(row + 16 * s + 8 + 4 * h);
    g_half8 b;
#pragma unroll
    for (int i = 0; i < 4; i++) { b[i] = lo[i]; b[4 + i] = hi[i]; }
    return b;
}
// fragment whose k runs DOWN a column of a row-major matrix: element j = m[(k0 + kperm(h, j)) * ld + col]
__device__ __forceinline__ g_half8 g_frag_col(const _Float16* m, size_t ld, size_t k0, uint32_t col, uint32_t h) {
    g_half8 b;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) b[j] = m[(k0 + g_kperm(h, j)) * ld + col];
    return b;
}

// One layer, one direction.  In [B, K] row-major, out [B, N] row-major (K, N multiples of 16).
//   TRANSW = false: out = epi(In Wm^T), Wm [N, K] row-major (rows >= n_live of Wm count as zero: the padded output layer)
//   TRANSW = true : out = epi(In Wm),   Wm [K, N] row-major (the data gradient through a layer whose matrix is [K_out = K, N_in = N])
// epi: forward (bwd_ref == nullptr): act(half(.)); backward: act'(bwd_ref) applied to half(.), bwd_ref [B, N] = the layer's stored output
template <bool TRANSW>
__global__ void __launch_bounds__(256) k_layer(const _Float16* __restrict__ In, const _Float16* __restrict__ Wm, _Float16* __restrict__ Out,
                                               uint32_t B, uint32_t K, uint32_t N, uint32_t n_live, uint32_t act,
                                               const _Float16* __restrict__ bwd_ref) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t units = B / 64;  // (B is a multiple of 128)
    for (uint32_t u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
        const size_t r0 = (size_t)u * 64 + n, r1 = r0 + 32;
        for (uint32_t nb = 0; nb * 32 < N; nb++) {
            g_float16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; i++) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
            const uint32_t feat = nb * 32 + n;  // the output feature whose weights this lane supplies
            for (uint32_t s = 0; s < K / 16; s++) {
                g_half8 a;
                if (feat < n_live && feat < N) a = TRANSW ? g_frag_col(Wm, N, (size_t)16 * s, feat, h) : g_frag_row(Wm + (size_t)feat * K, s, h);
                else a = g_zero8();
                const g_half8 b0 = g_frag_row(In + r0 * K, s, h), b1 = g_frag_row(In + r1 * K, s, h);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc1, 0, 0, 0);
            }
            // lane (n, h) holds, for batch rows r0 / r1, the output features nb*32 + 8q + 4h + e (q < 4, e < 4) in acc[4q + e]
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t f0 = nb * 32 + 8 * q + 4 * h;
                if (f0 >= N) continue;
#pragma unroll
                for (uint32_t t = 0; t < 2; t++) {
                    const size_t row = t ? r1 : r0;
                    g_half4 v;
                    g_half4 ref;
                    if (bwd_ref) ref = *reinterpret_cast<const g_half4*>(bwd_ref + row * N + f0);
#pragma unroll
                    for (uint32_t e = 0; e < 4; e++) {
                        const float pre = (float)(_Float16)(t ? acc1[4 * q + e] : acc0[4 * q + e]);
                        v[e] = (_Float16)(bwd_ref ? g_act_bwd(act, pre, (float)ref[e]) : g_act_fwd(act, pre));
                    }
                    *reinterpret_cast<g_half4*>(Out + row * N + f0) = v;
                }
            }
        }
    }
}

// dW[No, Ki] (+)= G[B, No]^T X[B, Ki]: wave (tile, segment) -> fp32 partial plane[segment][No, Ki]
__global__ void __launch_bounds__(64) k_wgrad_tile(const _Float16* __restrict__ G, uint32_t ldg, const _Float16* __restrict__ X, uint32_t ldx,
                                                   uint32_t B, uint32_t No, uint32_t Ki, uint32_t n_live, uint32_t seg_rows,
                                                   float* __restrict__ planes, size_t plane_stride) {
    const uint32_t lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    const uint32_t tiles_k = (Ki + 31) / 32;
    const uint32_t ob = blockIdx.x / tiles_k, kb = blockIdx.x % tiles_k;
    const uint32_t seg = blockIdx.y;
    const size_t p_begin = (size_t)seg * seg_rows;
    const size_t p_end = p_begin + seg_rows < B ? p_begin + seg_rows : B;
    g_float16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    const uint32_t orow = ob * 32 + n, kcol = kb * 32 + n;
    for (size_t p0 = p_begin; p0 < p_end; p0 += 16) {  // (segments are multiples of 16 rows)
        const g_half8 a = (orow < n_live && orow < No) ? g_frag_col(G, ldg, p0, orow, h) : g_zero8();
        const g_half8 b = kcol < Ki ? g_frag_col(X, ldx, p0, kcol, h) : g_zero8();
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    float* dst = planes + (size_t)seg * plane_stride;
    if (kcol < Ki) {
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) {
            const uint32_t o = ob * 32 + (r & 3u) + 8u * (r >> 2) + 4u * h;
            if (o < No) dst[(size_t)o * Ki + kcol] = acc[r];
        }
    }
}
// fixed-order sum of the segments -> fp16 gradient (optionally accumulated), GradScaler's non-finite flag
__global__ void __launch_bounds__(256) k_wgrad_finish(const float* __restrict__ planes, size_t plane_stride, uint32_t nseg, size_t n,
                                                      _Float16* __restrict__ gw, int accumulate, float* __restrict__ found_inf) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = planes[i];
        for (uint32_t s = 1; s < nseg; s++) v += planes[(size_t)s * plane_stride + i];
        if (accumulate) v += (float)gw[i];
        const _Float16 o = (_Float16)v;
        gw[i] = o;
        bad |= !(fabsf((float)o) <= 65504.0f);
    }
    if (found_inf && bad) *found_inf = 1.0f;  // (benign race: everyone writes 1)
}

inline uint32_t layer_grid(uint32_t B) { return std::min<uint32_t>(div_up<uint32_t>(B / 64, 4), 2048u); }

}  // namespace

// Shapes of the register-resident MFMA kernels (csrc/ffmlp.hip).  W = 128: forward / data-gradient / weight-gradient kernels of
// the stored-activation path (its three weight matrices' fragments need 136 KB of the CU's 160 KB of LDS at most); the
// re-computing fused backward stays at W <= 64 (a 128 x 128 fp32 weight-gradient accumulator is 256 registers per lane).
bool ffmlp_native_shape(uint32_t in_dim, uint32_t W, uint32_t n_layers) {
    if (W == 32 || W == 64) return in_dim <= 64;
    if (W != 128 || in_dim > 160 || n_layers < 2) return false;  // (128 < in_dim <= 160: the first weight gradient in two column blocks)
    const uint32_t NH = n_layers - 1;
    const uint32_t fwd_frags = 4 * (in_dim / 16) + NH * 32 + 8, bwd_frags = 4 + NH * 32 + ((in_dim + 31) / 32) * 8;
    return fwd_frags <= 144 && bwd_frags <= 144;  // 1 KiB each
}

// bytes of fp32 scratch the layer-by-layer backward wants at least (one partial plane of every weight matrix)
size_t ffmlp_generic_min_workspace(uint32_t in_dim, uint32_t W, uint32_t n_layers) {
    return ((size_t)W * in_dim + (size_t)(n_layers - 1) * W * W + (size_t)16 * W) * sizeof(float);
}

// forward: X [B, in] -> out [B, 16]; `acts` = forward_buffer [n, B, W] (training) or inference_buffer [2, B, W]
int ffmlp_generic_forward(const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t n_layers,
                          uint32_t act, uint32_t out_act, _Float16* acts, bool training, _Float16* out, hipStream_t st) {
    const size_t bw = (size_t)B * W;
    const _Float16* in = X;
    uint32_t k = in_dim;
    const _Float16* w = Wt;
    const dim3 grid(layer_grid(B)), block(256);
    for (uint32_t l = 0; l < n_layers; l++) {
        _Float16* y = acts + (training ? (size_t)l : (size_t)(l & 1u)) * bw;
        hipLaunchKernelGGL(k_layer<false>, grid, block, 0, st, in, w, y, B, k, W, W, act, (const _Float16*)nullptr);
        w += (size_t)W * k;
        in = y;
        k = W;
    }
    hipLaunchKernelGGL(k_layer<false>, grid, block, 0, st, in, w, out, B, W, 16u, 16u, out_act, (const _Float16*)nullptr);
    return check_launch("ffmlp_forward (layer-by-layer path)");
}

// backward: grad [B, 16] w.r.t. the (linear) output; fwd = forward_buffer [n, B, W]; bwd = backward_buffer [n, B, W] scratch;
// workspace: fp32 partial planes of the weight gradients (ffmlp_generic_min_workspace() bytes at least; more = more batch segments)
int ffmlp_generic_backward(const _Float16* grad, const _Float16* X, const _Float16* Wt, const _Float16* fwd, uint32_t B, uint32_t in_dim,
                           uint32_t W, uint32_t n_layers, uint32_t act, _Float16* bwd, _Float16* grad_inputs, _Float16* grad_weights,
                           bool accumulate, float* found_inf, float* workspace, size_t workspace_bytes, hipStream_t st) {
    const size_t bw = (size_t)B * W;
    size_t woff[16];
    S3D_REQUIRE(n_layers + 1 <= 16, "ffmlp (layer-by-layer path): too many layers");
    woff[0] = 0;
    woff[1] = (size_t)W * in_dim;
    for (uint32_t l = 2; l <= n_layers; l++) woff[l] = woff[l - 1] + (size_t)W * W;
    const size_t total = woff[n_layers] + (size_t)16 * W;
    S3D_REQUIRE(workspace && workspace_bytes >= total * sizeof(float), "ffmlp_backward (layer-by-layer path): workspace too small");
    // batch segments of the weight gradient: as many planes as the workspace holds, at most 64, each a multiple of 16 rows
    uint32_t nseg = (uint32_t)std::min<size_t>(workspace_bytes / (total * sizeof(float)), 64);
    uint32_t seg_rows = (div_up<uint32_t>(B, nseg) + 15u) & ~15u;
    nseg = div_up<uint32_t>(B, seg_rows);
    const dim3 grid(layer_grid(B)), block(256);
    auto wgrad = [&](const _Float16* G, uint32_t No, uint32_t n_live, const _Float16* Xin, uint32_t Ki, size_t off) {
        hipLaunchKernelGGL(k_wgrad_tile, dim3(((No + 31) / 32) * ((Ki + 31) / 32), nseg), dim3(64), 0, st, G, No, Xin, Ki, B, No, Ki, n_live,
                           seg_rows, workspace + off, total);
    };
    // output layer: dW_last[16, W] = grad^T fwd_{n-1};  G_{n-1}[B, W] = (grad W_last) * act'(fwd_{n-1})
    const _Float16* a_last = fwd + (size_t)(n_layers - 1) * bw;
    wgrad(grad, 16u, 16u, a_last, W, woff[n_layers]);
    _Float16* g = bwd + (size_t)(n_layers - 1) * bw;
    hipLaunchKernelGGL(k_layer<true>, grid, block, 0, st, grad, Wt + woff[n_layers], g, B, 16u, W, W, act, a_last);
    for (uint32_t l = n_layers - 1; l >= 1; l--) {  // hidden matrix l: input fwd_{l-1}, output gradient g = bwd_l
        const _Float16* a_in = fwd + (size_t)(l - 1) * bw;
        wgrad(g, W, W, a_in, W, woff[l]);
        _Float16* gp = bwd + (size_t)(l - 1) * bw;
        hipLaunchKernelGGL(k_layer<true>, grid, block, 0, st, (const _Float16*)g, Wt + woff[l], gp, B, W, W, W, act, a_in);
        g = gp;
    }
    wgrad(g, W, W, X, in_dim, 0);  // first matrix: input X
    if (grad_inputs)
        hipLaunchKernelGGL(k_layer<true>, grid, block, 0, st, (const _Float16*)g, Wt, grad_inputs, B, W, in_dim, in_dim, (uint32_t)G_ACT_NONE,
                           (const _Float16*)nullptr);
    hipLaunchKernelGGL(k_wgrad_finish, dim3(stream_grid(total, 256)), dim3(256), 0, st, (const float*)workspace, total, nseg, total, grad_weights,
                       accumulate ? 1 : 0, found_inf);
    return check_launch("ffmlp_backward (layer-by-layer path)");
}

}  // namespace s3d
