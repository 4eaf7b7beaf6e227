// optim.hip — the parameter update of the training step for gfx950: Adam straight from the fp16 gradients.
//
// The reference trains with torch.optim.Adam + torch.cuda.amp.GradScaler (nerf/utils.py:356-361,
// main_SealNeRF.py:283-288: betas (0.9, 0.99), eps 1e-15).  Around a 12.2 M-entry hash table that costs, per step:
// table fp32->fp16 cast for the forward (73 MB), fp16->fp32 gradient cast (73 MB), gradient accumulate (147 MB),
// non-finite check + unscale (98 MB) and the Adam pass itself (343 MB) — ~0.7 GB of HBM traffic for an update
// whose inputs are a 24.5 MB fp16 gradient.  Here the gradient is consumed where the backward kernel left it:
//   s3d_grads_nonfinite  one read of the gradient, raises the found_inf flag (GradScaler semantics)
//   s3d_adam_step        p, m, v (fp32) <- Adam(g / grad_scale), skipped as a whole when found_inf is set; optionally
//                        writes the fp16 copy of p that the next forward (autocast) reads instead of re-casting
// Update rule = torch's fused Adam functor (no amsgrad, no weight decay, maximize off), fp32 math:
//   m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// with the step count t kept on the device (graph-capturable) and advanced only by steps that are not skipped.
#include "s3d_common.hpp"

namespace s3d {
namespace {

template <typename G> __device__ __forceinline__ float grad_to_f(G g);
template <> __device__ __forceinline__ float grad_to_f<float>(float g) { return g; }
template <> __device__ __forceinline__ float grad_to_f<__half>(__half g) { return __half2float(g); }

template <typename G>
__global__ void __launch_bounds__(256) k_grads_nonfinite(const G* __restrict__ g, size_t n, float* __restrict__ found_inf) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = grad_to_f<G>(g[i]);
        bad |= !(fabsf(v) <= 3.402823466e38f);  // inf or NaN
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) *found_inf = 1.0f;  // benign race: everyone writes 1
}

template <typename G>
__global__ void __launch_bounds__(256) k_adam_step(float* __restrict__ p, const G* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, __half* __restrict__ p_half, size_t n, float lr,
                                                   float beta1, float beta2, float eps, const float* __restrict__ step,
                                                   const float* __restrict__ grad_scale, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.0f) return;  // the whole step is skipped (GradScaler.step)
    const float t = *step + 1.0f;                 // this update's step number; k_adam_advance stores it afterwards
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = grad_to_f<G>(g[i]) * inv_scale;
        float mi = m[i], vi = v[i], pi = p[i];
        mi = mi + (1.0f - beta1) * (gi - mi);
        vi = beta2 * vi + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (p_half) p_half[i] = __float2half(pi);
    }
}

__global__ void k_adam_advance(float* __restrict__ step, const float* __restrict__ found_inf) {
    if (!(found_inf && *found_inf != 0.0f)) *step += 1.0f;
}

// torch.amp.GradScaler.update (aten::_amp_update_scale_): back off on overflow, grow after `interval` clean steps; then
// clear the flag for the next step (saves the separate fill launch)
__global__ void k_scaler_update(float* __restrict__ scale, int32_t* __restrict__ growth_tracker, float* __restrict__ found_inf,
                                float growth, float backoff, int32_t interval) {
    if (*found_inf != 0.0f) {
        *scale = *scale * backoff;
        *growth_tracker = 0;
    } else {
        const int32_t ok = *growth_tracker + 1;
        if (ok == interval) {
            const float grown = *scale * growth;
            if (grown <= 3.402823466e38f) *scale = grown;  // (torch keeps the scale when growing would overflow)
            *growth_tracker = 0;
        } else {
            *growth_tracker = ok;
        }
    }
    *found_inf = 0.0f;
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_grads_nonfinite(const void* grad, size_t n, int dtype, float* found_inf, s3d_stream_t stream) {
    if (n == 0) return S3D_OK;
    S3D_REQUIRE(grad && found_inf, "grads_nonfinite: null pointer");
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grads_nonfinite: dtype must be f32 or f16");
    const uint32_t grid = stream_grid(n / 8 + 1, 256);
    if (dtype == S3D_F16)
        hipLaunchKernelGGL(k_grads_nonfinite<__half>, dim3(grid), dim3(256), 0, as_stream(stream), (const __half*)grad, n, found_inf);
    else
        hipLaunchKernelGGL(k_grads_nonfinite<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)grad, n, found_inf);
    return check_launch("grads_nonfinite");
}

S3D_EXPORT int s3d_adam_step(float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq,
                             uint16_t* param_half, size_t n, float lr, float beta1, float beta2, float eps,
                             const float* step, const float* grad_scale, const float* found_inf, s3d_stream_t stream) {
    if (n == 0) return S3D_OK;
    S3D_REQUIRE(param && grad && exp_avg && exp_avg_sq && step, "adam_step: null pointer");
    S3D_REQUIRE(grad_dtype == S3D_F32 || grad_dtype == S3D_F16, "adam_step: grad dtype must be f32 or f16");
    const uint32_t grid = stream_grid(n / 4 + 1, 256);
    if (grad_dtype == S3D_F16)
        hipLaunchKernelGGL(k_adam_step<__half>, dim3(grid), dim3(256), 0, as_stream(stream), param, (const __half*)grad, exp_avg,
                           exp_avg_sq, (__half*)param_half, n, lr, beta1, beta2, eps, step, grad_scale, found_inf);
    else
        hipLaunchKernelGGL(k_adam_step<float>, dim3(grid), dim3(256), 0, as_stream(stream), param, (const float*)grad, exp_avg,
                           exp_avg_sq, (__half*)param_half, n, lr, beta1, beta2, eps, step, grad_scale, found_inf);
    return check_launch("adam_step");
}

S3D_EXPORT int s3d_adam_advance(float* step, const float* found_inf, s3d_stream_t stream) {
    S3D_REQUIRE(step, "adam_advance: null pointer");
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, as_stream(stream), step, found_inf);
    return check_launch("adam_advance");
}

S3D_EXPORT int s3d_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, float growth_factor,
                                 float backoff_factor, int32_t growth_interval, s3d_stream_t stream) {
    S3D_REQUIRE(scale && growth_tracker && found_inf, "scaler_update: null pointer");
    hipLaunchKernelGGL(k_scaler_update, dim3(1), dim3(1), 0, as_stream(stream), scale, growth_tracker, found_inf, growth_factor,
                       backoff_factor, growth_interval);
    return check_launch("scaler_update");
}
