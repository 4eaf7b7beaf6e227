// ffmlp_generic.hip — the fully fused MLP for the network shapes the register-resident MFMA kernels of ffmlp.hip do not
// cover: hidden_dim 16 / 128 / 256 and input_dim > 64 (ffmlp.cu:40-44 dispatches hidden_dim in {16, 32, 64, 128, 256}).
//
// Layer by layer, one library GEMM per matrix (rocBLAS, fp16 storage, fp32 accumulation — the same dense math
// Y = act(X W^T) that defines ffmlp's numerics) and one elementwise launch per layer for the activation / its gradient.
// This is the drop-in path for completeness of the boundary, not the hot path: the networks of the BASELINE configs
// (hidden 64, in <= 64) never come here.  Buffers: forward_buffer / backward_buffer are plain row-major [n, B, W];
// inference uses inference_buffer [2, B, W] as ping-pong scratch.
#include "s3d_common.hpp"
#include <rocblas/rocblas.h>
#include <mutex>

namespace s3d {
namespace {

enum { G_ACT_RELU = 0, G_ACT_EXP = 1, G_ACT_SINE = 2, G_ACT_SIGMOID = 3, G_ACT_SQUAREPLUS = 4, G_ACT_SOFTPLUS = 5, G_ACT_NONE = 6 };
constexpr float kActScale = 10.0f;  // utils.h:424-589 (K_ACT)

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

// in place: y = act(y)  (y already rounded to fp16 by the GEMM, as the fused kernels round the pre-activation)
__global__ void __launch_bounds__(256) k_act_forward(_Float16* __restrict__ y, size_t n, uint32_t act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (_Float16)g_act_fwd(act, (float)y[i]);
}
// in place: g = g * act'(.) expressed through the layer's stored output
__global__ void __launch_bounds__(256) k_act_backward(_Float16* __restrict__ g, const _Float16* __restrict__ fwd, size_t n, uint32_t act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        g[i] = (_Float16)g_act_bwd(act, (float)g[i], (float)fwd[i]);
}
__global__ void __launch_bounds__(256) k_half_nonfinite(const _Float16* __restrict__ p, size_t n, float* __restrict__ found_inf) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= !(fabsf((float)p[i]) <= 3.402823466e38f);
    if (bad) *found_inf = 1.0f;
}

rocblas_handle handle_for_current_device() {
    static std::mutex mu;
    static rocblas_handle handles[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!handles[dev] && rocblas_create_handle(&handles[dev]) != rocblas_status_success) handles[dev] = nullptr;
    return handles[dev];
}

// row-major C[M, N] (+)= op(A) op(B) through rocBLAS' column-major interface (a row-major matrix is its transpose in
// column-major): see the three call sites for the operand order
int gemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const _Float16* a, int lda, const _Float16* b,
         int ldb, _Float16* c, int ldc, float beta) {
    const float alpha = 1.0f;
    const rocblas_status st = rocblas_gemm_ex(h, ta, tb, m, n, k, &alpha, a, rocblas_datatype_f16_r, lda, b, rocblas_datatype_f16_r, ldb,
                                              &beta, c, rocblas_datatype_f16_r, ldc, c, rocblas_datatype_f16_r, ldc,
                                              rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    if (st != rocblas_status_success) {
        set_error("ffmlp (generic path): rocblas_gemm_ex failed with status %d", (int)st);
        return S3D_ERR_HIP;
    }
    return S3D_OK;
}

inline uint32_t ew_grid(size_t n) { return stream_grid(n, 256); }

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

// forward: X [B, in] -> out [B, 16]; `acts` = forward_buffer [n, B, W] (training) or inference_buffer [2, B, W]
int ffmlp_generic_forward(const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t n_layers,
                          uint32_t act, uint32_t out_act, _Float16* acts, bool training, _Float16* out, hipStream_t st) {
    rocblas_handle h = handle_for_current_device();
    S3D_REQUIRE(h, "ffmlp (generic path): no rocBLAS handle");
    if (rocblas_set_stream(h, st) != rocblas_status_success) { set_error("ffmlp (generic path): rocblas_set_stream failed"); return S3D_ERR_HIP; }
    const size_t bw = (size_t)B * W;
    const _Float16* in = X;
    uint32_t k = in_dim;
    const _Float16* w = Wt;
    for (uint32_t l = 0; l < n_layers; l++) {
        _Float16* y = acts + (training ? (size_t)l : (size_t)(l & 1u)) * bw;
        // Y[B, W] = in[B, k] W_l[W, k]^T : column-major Y^T[W, B] = W_l^T-as-stored (k x W, ld k) transposed x in^T (k x B, ld k)
        if (int rc = gemm(h, rocblas_operation_transpose, rocblas_operation_none, (int)W, (int)B, (int)k, w, (int)k, in, (int)k, y, (int)W, 0.0f)) return rc;
        hipLaunchKernelGGL(k_act_forward, dim3(ew_grid(bw)), dim3(256), 0, st, y, bw, act);
        w += (size_t)W * k;
        in = y;
        k = W;
    }
    if (int rc = gemm(h, rocblas_operation_transpose, rocblas_operation_none, 16, (int)B, (int)W, w, (int)W, in, (int)W, out, 16, 0.0f)) return rc;
    if (out_act != G_ACT_NONE) hipLaunchKernelGGL(k_act_forward, dim3(ew_grid((size_t)B * 16)), dim3(256), 0, st, out, (size_t)B * 16, out_act);
    return check_launch("ffmlp_forward (generic path)");
}

// backward: grad [B, 16] w.r.t. the (linear) output; fwd = forward_buffer [n, B, W]; bwd = backward_buffer [n, B, W] scratch
int ffmlp_generic_backward(const _Float16* grad, const _Float16* X, const _Float16* Wt, const _Float16* fwd, uint32_t B, uint32_t in_dim,
                           uint32_t W, uint32_t n_layers, uint32_t act, _Float16* bwd, _Float16* grad_inputs, _Float16* grad_weights,
                           bool accumulate, float* found_inf, hipStream_t st) {
    rocblas_handle h = handle_for_current_device();
    S3D_REQUIRE(h, "ffmlp (generic path): no rocBLAS handle");
    if (rocblas_set_stream(h, st) != rocblas_status_success) { set_error("ffmlp (generic path): rocblas_set_stream failed"); return S3D_ERR_HIP; }
    const size_t bw = (size_t)B * W;
    const float beta = accumulate ? 1.0f : 0.0f;
    // offsets of the layers' matrices
    size_t woff[16];
    S3D_REQUIRE(n_layers + 1 <= 16, "ffmlp (generic path): too many layers");
    woff[0] = 0;
    woff[1] = (size_t)W * in_dim;
    for (uint32_t l = 2; l <= n_layers; l++) woff[l] = woff[l - 1] + (size_t)W * W;
    const size_t total = woff[n_layers] + (size_t)16 * W;
    // output layer: dW_last[16, W] = grad^T fwd_{n-1};  G_{n-1}[B, W] = (grad W_last) * act'(fwd_{n-1})
    const _Float16* a_last = fwd + (size_t)(n_layers - 1) * bw;
    // row-major dW[N, K] = G^T X : column-major dW^T (K x N, ld K) = X^T-as-stored (K x B, ld K) x G-as-stored^T (B x N)
    if (int rc = gemm(h, rocblas_operation_none, rocblas_operation_transpose, (int)W, 16, (int)B, a_last, (int)W, grad, 16,
                      grad_weights + woff[n_layers], (int)W, beta)) return rc;
    _Float16* g = bwd + (size_t)(n_layers - 1) * bw;
    // row-major Gin[B, K] = G[B, N] W[N, K] : column-major Gin^T (K x B, ld K) = W-as-stored (K x N, ld K) x G-as-stored (N x B, ld N)
    if (int rc = gemm(h, rocblas_operation_none, rocblas_operation_none, (int)W, (int)B, 16, Wt + woff[n_layers], (int)W, grad, 16, g, (int)W, 0.0f)) return rc;
    hipLaunchKernelGGL(k_act_backward, dim3(ew_grid(bw)), dim3(256), 0, st, g, a_last, bw, act);
    for (uint32_t l = n_layers - 1; l >= 1; l--) {  // hidden matrix l: input fwd_{l-1}, output gradient g = bwd_l
        const _Float16* a_in = fwd + (size_t)(l - 1) * bw;
        if (int rc = gemm(h, rocblas_operation_none, rocblas_operation_transpose, (int)W, (int)W, (int)B, a_in, (int)W, g, (int)W,
                          grad_weights + woff[l], (int)W, beta)) return rc;
        _Float16* gp = bwd + (size_t)(l - 1) * bw;
        if (int rc = gemm(h, rocblas_operation_none, rocblas_operation_none, (int)W, (int)B, (int)W, Wt + woff[l], (int)W, g, (int)W, gp, (int)W, 0.0f)) return rc;
        hipLaunchKernelGGL(k_act_backward, dim3(ew_grid(bw)), dim3(256), 0, st, gp, a_in, bw, act);
        g = gp;
    }
    // first matrix: input X
    if (int rc = gemm(h, rocblas_operation_none, rocblas_operation_transpose, (int)in_dim, (int)W, (int)B, X, (int)in_dim, g, (int)W,
                      grad_weights, (int)in_dim, beta)) return rc;
    if (grad_inputs)
        if (int rc = gemm(h, rocblas_operation_none, rocblas_operation_none, (int)in_dim, (int)B, (int)W, Wt, (int)in_dim, g, (int)W, grad_inputs,
                          (int)in_dim, 0.0f)) return rc;
    if (found_inf) hipLaunchKernelGGL(k_half_nonfinite, dim3(ew_grid(total)), dim3(256), 0, st, grad_weights, total, found_inf);
    return check_launch("ffmlp_backward (generic path)");
}

}  // namespace s3d
