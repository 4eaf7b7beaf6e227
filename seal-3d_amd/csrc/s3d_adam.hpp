// s3d_adam.hpp — the Adam update arithmetic shared by csrc/optim.hip (the optimizer's own launches) and csrc/gridencoder.hip
// (the update applied inside the grid backward's accumulate kernel): one definition, so both routes produce the same bits.
// Update rule = torch's fused Adam functor (no amsgrad, no weight decay, maximize off), fp32 math:
//   m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#pragma once
#include "s3d_common.hpp"

namespace s3d {

struct AdamCoef {
    float beta1, beta2, eps, inv_scale, step_size, bc2_sqrt;
    float l1;  // gradient of an L1 penalty l1 * sum|p| formed here (s3d_adam_tensor.l1): + l1 * sign(p), sign(0) = 0 like torch.sign
};
__device__ __forceinline__ void adam_update(const AdamCoef& c, float g, float& m, float& v, float& p) {
    float gi = g * c.inv_scale;
    if (c.l1 != 0.0f) gi = gi + (p > 0.0f ? c.l1 : (p < 0.0f ? -c.l1 : 0.0f));
    m = m + (1.0f - c.beta1) * (gi - m);
    v = c.beta2 * v + (1.0f - c.beta2) * gi * gi;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = p - c.step_size * (m / denom);
}
// the coefficients of this step from the device-side scalars (step count BEFORE the update; loss scale; schedule factor)
__device__ __forceinline__ AdamCoef adam_coef(float lr, float beta1, float beta2, float eps, float l1, const float* step,
                                              const float* grad_scale, const float* lr_scale) {
    const float t = *step + 1.0f;  // this update's step number; the advance kernel stores it afterwards
    AdamCoef c;
    c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.l1 = l1;
    c.inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    c.step_size = (lr_scale ? lr * *lr_scale : lr) / bc1;  // (lr_scale: the schedule's factor, read at run time by a replayed graph)
    c.bc2_sqrt = sqrtf(bc2);
    return c;
}

}  // namespace s3d
