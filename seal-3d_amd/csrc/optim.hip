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
#include "s3d_adam.hpp"

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

template <typename G> struct GradVec4;
template <> struct GradVec4<float> {
    static __device__ __forceinline__ void load(const float* g, size_t i, float (&o)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(g + i);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
    static __device__ __forceinline__ void zero(float* g, size_t i) { *reinterpret_cast<float4*>(g + i) = make_float4(0, 0, 0, 0); }
};
template <> struct GradVec4<__half> {
    static __device__ __forceinline__ void load(const __half* g, size_t i, float (&o)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(g + i);
        const __half2 a = *reinterpret_cast<const __half2*>(&t.x), b = *reinterpret_cast<const __half2*>(&t.y);
        o[0] = __low2float(a); o[1] = __high2float(a); o[2] = __low2float(b); o[3] = __high2float(b);
    }
    static __device__ __forceinline__ void zero(__half* g, size_t i) { *reinterpret_cast<uint2*>(g + i) = make_uint2(0u, 0u); }
};

// `consume`: the gradient is cleared behind the read (the producers of the next step ACCUMULATE into it: saves the
// optimizer's separate zero fill); `skip`: overflow step, nothing is updated but a consumed gradient is still cleared
template <typename G>
__device__ __forceinline__ void clear_range(G* __restrict__ g, size_t n, bool vec, size_t tid, size_t nthreads) {
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t q = tid; q < n4; q += nthreads) GradVec4<G>::zero(g, q * 4);
    for (size_t i = n4 * 4 + tid; i < n; i += nthreads) g[i] = G(0.0f);
}

template <typename G>
__device__ __forceinline__ void adam_range(const AdamCoef& c, float* __restrict__ p, G* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, __half* __restrict__ p_half, size_t n, bool vec, size_t tid,
                                           size_t nthreads, bool consume) {
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t q = tid; q < n4; q += nthreads) {
        const size_t i = q * 4;
        float gi[4];
        GradVec4<G>::load(g, i, gi);
        if (consume) GradVec4<G>::zero(g, i);
        // the 24 B per element of fp32 state stream through once per step: non-temporal, so that they do not push the fp16
        // table copy (read by the next forward) and the gradient buffer out of the L2 / Infinity Cache
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v mv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m + i));
        const f4v vv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v + i));
        const f4v pv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + i));
        float4 mi = make_float4(mv.x, mv.y, mv.z, mv.w), vi = make_float4(vv.x, vv.y, vv.z, vv.w),
               pi = make_float4(pv.x, pv.y, pv.z, pv.w);
        adam_update(c, gi[0], mi.x, vi.x, pi.x);
        adam_update(c, gi[1], mi.y, vi.y, pi.y);
        adam_update(c, gi[2], mi.z, vi.z, pi.z);
        adam_update(c, gi[3], mi.w, vi.w, pi.w);
        __builtin_nontemporal_store(f4v{mi.x, mi.y, mi.z, mi.w}, reinterpret_cast<f4v*>(m + i));
        __builtin_nontemporal_store(f4v{vi.x, vi.y, vi.z, vi.w}, reinterpret_cast<f4v*>(v + i));
        __builtin_nontemporal_store(f4v{pi.x, pi.y, pi.z, pi.w}, reinterpret_cast<f4v*>(p + i));
        if (p_half) {
            const __half2 a = __floats2half2_rn(pi.x, pi.y), b = __floats2half2_rn(pi.z, pi.w);
            uint2 o;
            o.x = *reinterpret_cast<const uint32_t*>(&a);
            o.y = *reinterpret_cast<const uint32_t*>(&b);
            *reinterpret_cast<uint2*>(p_half + i) = o;
        }
    }
    for (size_t i = n4 * 4 + tid; i < n; i += nthreads) {
        float mi = m[i], vi = v[i], pi = p[i];
        adam_update(c, grad_to_f<G>(g[i]), mi, vi, pi);
        if (consume) g[i] = G(0.0f);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (p_half) p_half[i] = __float2half(pi);
    }
}

// A parameter whose fp16 gradient and fp16 copy live inside a PACKED weight buffer (the nn.Linear weights of the two-encoder
// Seal network inside the fused MLP kernels' [out, in_padded] layout): element i = (row, col) of the [rows, cols] parameter
// sits at row * stride + col of `g` and `p_half`; the fp32 state stays contiguous.  ~10 K elements per tensor: scalar accesses.
template <typename G>
__device__ __forceinline__ void adam_range_packed(const AdamCoef& c, float* __restrict__ p, G* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, __half* __restrict__ p_half, size_t n, uint32_t cols,
                                                  uint32_t stride, size_t tid, size_t nthreads, bool consume) {
    for (size_t i = tid; i < n; i += nthreads) {
        const size_t j = (i / cols) * stride + i % cols;
        float mi = m[i], vi = v[i], pi = p[i];
        adam_update(c, grad_to_f<G>(g[j]), mi, vi, pi);
        if (consume) g[j] = G(0.0f);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (p_half) p_half[j] = __float2half(pi);
    }
}

// Four consecutive elements per lane and trip (16-byte accesses of the fp32 state, 8-byte of the fp16 gradient / copy): the
// update streams 28 B per element and is HBM-bound; `vec` is false for a tensor whose pointers are not 16-byte aligned.
template <typename G>
__global__ void __launch_bounds__(256) k_adam_step(float* __restrict__ p, const G* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, __half* __restrict__ p_half, size_t n, float lr,
                                                   float beta1, float beta2, float eps, const float* __restrict__ step,
                                                   const float* __restrict__ grad_scale, const float* __restrict__ found_inf,
                                                   const float* __restrict__ lr_scale, uint32_t vec) {
    if (found_inf && *found_inf != 0.0f) return;  // the whole step is skipped (GradScaler.step)
    const AdamCoef c = adam_coef(lr, beta1, beta2, eps, 0.0f, step, grad_scale, lr_scale);
    adam_range<G>(c, p, const_cast<G*>(g), m, v, p_half, n, vec != 0, (size_t)blockIdx.x * 256 + threadIdx.x,
                  (size_t)gridDim.x * 256, false);
}

// Every tensor of the optimizer in ONE launch (the hot path updates a 12 M-element hash table and two ~10 K-element MLPs: the
// small ones cost a launch each otherwise).  Blocks [first_block[i], first_block[i+1]) stride over tensor i.
constexpr int kAdamMaxTensors = 16;
struct AdamBatch {
    float* p[kAdamMaxTensors];
    void* g[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    __half* h[kAdamMaxTensors];
    size_t n[kAdamMaxTensors];
    float lr[kAdamMaxTensors], beta1[kAdamMaxTensors], beta2[kAdamMaxTensors], eps[kAdamMaxTensors], l1[kAdamMaxTensors];
    uint32_t first_block[kAdamMaxTensors + 1];
    uint32_t cols[kAdamMaxTensors], stride[kAdamMaxTensors];  // stride != 0: packed layout of g / h (adam_range_packed)
    uint8_t half_grad[kAdamMaxTensors], vec[kAdamMaxTensors], consume[kAdamMaxTensors];
    int32_t count;
};

__global__ void __launch_bounds__(256) k_adam_step_multi(AdamBatch b, const float* __restrict__ step,
                                                         const float* __restrict__ grad_scale,
                                                         const float* __restrict__ found_inf,
                                                         const float* __restrict__ lr_scale) {
    int i = 0;
    while (i + 1 < b.count && blockIdx.x >= b.first_block[i + 1]) i++;
    const size_t tid = (size_t)(blockIdx.x - b.first_block[i]) * 256 + threadIdx.x;
    const size_t nthreads = (size_t)(b.first_block[i + 1] - b.first_block[i]) * 256;
    if (found_inf && *found_inf != 0.0f) {  // skipped step: nothing is updated
        if (b.consume[i] && b.stride[i]) {
            for (size_t k = tid; k < b.n[i]; k += nthreads) {
                const size_t j = (k / b.cols[i]) * b.stride[i] + k % b.cols[i];
                if (b.half_grad[i]) ((__half*)b.g[i])[j] = __half(0.0f);
                else ((float*)b.g[i])[j] = 0.0f;
            }
        } else if (b.consume[i]) {
            if (b.half_grad[i]) clear_range<__half>((__half*)b.g[i], b.n[i], b.vec[i] != 0, tid, nthreads);
            else clear_range<float>((float*)b.g[i], b.n[i], b.vec[i] != 0, tid, nthreads);
        }
        return;
    }
    const AdamCoef c = adam_coef(b.lr[i], b.beta1[i], b.beta2[i], b.eps[i], b.l1[i], step, grad_scale, lr_scale);
    if (b.stride[i]) {
        if (b.half_grad[i])
            adam_range_packed<__half>(c, b.p[i], (__half*)b.g[i], b.m[i], b.v[i], b.h[i], b.n[i], b.cols[i], b.stride[i], tid, nthreads, b.consume[i] != 0);
        else
            adam_range_packed<float>(c, b.p[i], (float*)b.g[i], b.m[i], b.v[i], b.h[i], b.n[i], b.cols[i], b.stride[i], tid, nthreads, b.consume[i] != 0);
    } else if (b.half_grad[i])
        adam_range<__half>(c, b.p[i], (__half*)b.g[i], b.m[i], b.v[i], b.h[i], b.n[i], b.vec[i] != 0, tid, nthreads, b.consume[i] != 0);
    else
        adam_range<float>(c, b.p[i], (float*)b.g[i], b.m[i], b.v[i], b.h[i], b.n[i], b.vec[i] != 0, tid, nthreads, b.consume[i] != 0);
}

__global__ void k_adam_advance(float* __restrict__ step, const float* __restrict__ found_inf) {
    if (!(found_inf && *found_inf != 0.0f)) *step += 1.0f;
}

// torch.amp.GradScaler.update (aten::_amp_update_scale_): back off on overflow, grow after `interval` clean steps; then
// clear the flag for the next step (saves the separate fill launch)
__device__ __forceinline__ void scaler_update(float* __restrict__ scale, int32_t* __restrict__ growth_tracker,
                                              float* __restrict__ found_inf, float growth, float backoff, int32_t interval,
                                              float* __restrict__ adam_step) {
    if (adam_step && *found_inf == 0.0f) *adam_step += 1.0f;  // (k_adam_advance folded in: one launch less per step)
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
__global__ void k_scaler_update(float* __restrict__ scale, int32_t* __restrict__ growth_tracker, float* __restrict__ found_inf,
                                float growth, float backoff, int32_t interval, float* __restrict__ adam_step) {
    scaler_update(scale, growth_tracker, found_inf, growth, backoff, interval, adam_step);
}

// End of a graph-replayed training step: file the step's loss and the marcher's {samples, rays} counter in their 16-slot rings
// (nerf/renderer.py keeps the counters of the last 16 steps for `mean_count`), clear the counter for the next replay and
// advance the slot — what the host otherwise does with two copies and a fill per step.
__device__ __forceinline__ void step_ring_push(const float* __restrict__ loss, int32_t* __restrict__ counter,
                                               float* __restrict__ loss_ring, int32_t* __restrict__ counter_ring,
                                               int32_t* __restrict__ cursor, int32_t ring, int32_t loss_slots) {
    int32_t c = *cursor;
    if (c < 0 || c >= ring) c = 0;
    // the loss history may be longer than the counter ring: slot = running step number % loss_slots (a tensor handed to the
    // caller for step k stays valid until step k + loss_slots); loss_slots <= 0: the counter ring's slot
    if (loss && loss_ring) loss_ring[loss_slots > 0 ? (int32_t)((uint32_t)cursor[1] % (uint32_t)loss_slots) : c] = *loss;
    counter_ring[2 * c] = counter[0];
    counter_ring[2 * c + 1] = counter[1];
    counter[0] = 0;
    counter[1] = 0;
    cursor[0] = (c + 1) % ring;
    cursor[1] += 1;
}
__global__ void k_step_ring_push(const float* __restrict__ loss, int32_t* __restrict__ counter, float* __restrict__ loss_ring,
                                 int32_t* __restrict__ counter_ring, int32_t* __restrict__ cursor, int32_t ring, int32_t loss_slots) {
    step_ring_push(loss, counter, loss_ring, counter_ring, cursor, ring, loss_slots);
}
// both single-thread epilogues of a step in one launch
__global__ void k_step_epilogue(float* __restrict__ scale, int32_t* __restrict__ growth_tracker, float* __restrict__ found_inf,
                                float growth, float backoff, int32_t interval, float* __restrict__ adam_step,
                                const float* __restrict__ loss, int32_t* __restrict__ counter, float* __restrict__ loss_ring,
                                int32_t* __restrict__ counter_ring, int32_t* __restrict__ cursor, int32_t ring, int32_t loss_slots) {
    scaler_update(scale, growth_tracker, found_inf, growth, backoff, interval, adam_step);
    step_ring_push(loss, counter, loss_ring, counter_ring, cursor, ring, loss_slots);
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
                             const float* step, const float* grad_scale, const float* found_inf, const float* lr_scale,
                             s3d_stream_t stream) {
    if (n == 0) return S3D_OK;
    S3D_REQUIRE(param && grad && exp_avg && exp_avg_sq && step, "adam_step: null pointer");
    S3D_REQUIRE(grad_dtype == S3D_F32 || grad_dtype == S3D_F16, "adam_step: grad dtype must be f32 or f16");
    const uint32_t grid = stream_grid(n / 8 + 1, 256);
    const uintptr_t bits = (uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | ((uintptr_t)grad << (grad_dtype == S3D_F16 ? 1 : 0)) |
                           ((uintptr_t)param_half << 1);
    const uint32_t vec = (bits & 15) == 0;
    if (grad_dtype == S3D_F16)
        hipLaunchKernelGGL(k_adam_step<__half>, dim3(grid), dim3(256), 0, as_stream(stream), param, (const __half*)grad, exp_avg,
                           exp_avg_sq, (__half*)param_half, n, lr, beta1, beta2, eps, step, grad_scale, found_inf, lr_scale, vec);
    else
        hipLaunchKernelGGL(k_adam_step<float>, dim3(grid), dim3(256), 0, as_stream(stream), param, (const float*)grad, exp_avg,
                           exp_avg_sq, (__half*)param_half, n, lr, beta1, beta2, eps, step, grad_scale, found_inf, lr_scale, vec);
    return check_launch("adam_step");
}

S3D_EXPORT int s3d_adam_step_multi(const s3d_adam_tensor* tensors, int32_t n_tensors, const float* step, const float* grad_scale,
                                   const float* found_inf, const float* lr_scale, int consume_grads, s3d_stream_t stream) {
    S3D_REQUIRE(n_tensors >= 0 && (tensors || n_tensors == 0) && step, "adam_step_multi: null pointer");
    for (int32_t base = 0; base < n_tensors; base += kAdamMaxTensors) {
        AdamBatch b;
        memset(&b, 0, sizeof(b));
        uint32_t blocks = 0;
        for (int32_t k = base; k < n_tensors && k < base + kAdamMaxTensors; k++) {  // (exactly this batch's index range)
            const s3d_adam_tensor& t = tensors[k];
            if (t.n == 0) continue;
            S3D_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, "adam_step_multi: null pointer in tensor %d", k);
            S3D_REQUIRE(t.grad_dtype == S3D_F32 || t.grad_dtype == S3D_F16, "adam_step_multi: grad dtype must be f32 or f16");
            const int i = b.count++;
            b.p[i] = t.param; b.g[i] = const_cast<void*>(t.grad); b.m[i] = t.exp_avg; b.v[i] = t.exp_avg_sq; b.h[i] = (__half*)t.param_half;
            b.n[i] = t.n; b.lr[i] = t.lr; b.beta1[i] = t.beta1; b.beta2[i] = t.beta2; b.eps[i] = t.eps; b.l1[i] = t.l1;
            b.half_grad[i] = t.grad_dtype == S3D_F16;
            b.consume[i] = (consume_grads || t.consume) ? 1 : 0;
            S3D_REQUIRE((t.pack_stride == 0) || (t.pack_cols > 0 && t.pack_cols <= t.pack_stride && t.n % t.pack_cols == 0),
                        "adam_step_multi: tensor %d: packed layout needs 0 < pack_cols <= pack_stride and whole rows", k);
            b.cols[i] = t.pack_stride ? t.pack_cols : 1u;
            b.stride[i] = t.pack_stride;
            const uintptr_t bits = (uintptr_t)t.param | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq |
                                   ((uintptr_t)t.grad << (t.grad_dtype == S3D_F16 ? 1 : 0)) | ((uintptr_t)t.param_half << 1);
            b.vec[i] = (bits & 15) == 0;
            b.first_block[i] = blocks;
            blocks += stream_grid(t.n / 8 + 1, 256);
            b.first_block[i + 1] = blocks;
        }
        if (b.count == 0) continue;
        hipLaunchKernelGGL(k_adam_step_multi, dim3(blocks), dim3(256), 0, as_stream(stream), b, step, grad_scale, found_inf, lr_scale);
    }
    return check_launch("adam_step_multi");
}

S3D_EXPORT int s3d_adam_advance(float* step, const float* found_inf, s3d_stream_t stream) {
    S3D_REQUIRE(step, "adam_advance: null pointer");
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, as_stream(stream), step, found_inf);
    return check_launch("adam_advance");
}

S3D_EXPORT int s3d_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, float growth_factor,
                                 float backoff_factor, int32_t growth_interval, float* adam_step, s3d_stream_t stream) {
    S3D_REQUIRE(scale && growth_tracker && found_inf, "scaler_update: null pointer");
    hipLaunchKernelGGL(k_scaler_update, dim3(1), dim3(1), 0, as_stream(stream), scale, growth_tracker, found_inf, growth_factor,
                       backoff_factor, growth_interval, adam_step);
    return check_launch("scaler_update");
}

S3D_EXPORT int s3d_step_ring_push(const float* loss, int32_t* counter, float* loss_ring, int32_t* counter_ring, int32_t* cursor,
                                  int32_t ring, int32_t loss_slots, s3d_stream_t stream) {
    S3D_REQUIRE(counter && counter_ring && cursor && ring > 0, "step_ring_push: null pointer or empty ring");
    hipLaunchKernelGGL(k_step_ring_push, dim3(1), dim3(1), 0, as_stream(stream), loss, counter, loss_ring, counter_ring, cursor, ring, loss_slots);
    return check_launch("step_ring_push");
}

S3D_EXPORT int s3d_step_epilogue(float* scale, int32_t* growth_tracker, float* found_inf, float growth_factor,
                                 float backoff_factor, int32_t growth_interval, float* adam_step, const float* loss,
                                 int32_t* counter, float* loss_ring, int32_t* counter_ring, int32_t* cursor, int32_t ring,
                                 int32_t loss_slots, s3d_stream_t stream) {
    S3D_REQUIRE(scale && growth_tracker && found_inf, "step_epilogue: null pointer (scaler)");
    S3D_REQUIRE(counter && counter_ring && cursor && ring > 0, "step_epilogue: null pointer or empty ring");
    hipLaunchKernelGGL(k_step_epilogue, dim3(1), dim3(1), 0, as_stream(stream), scale, growth_tracker, found_inf, growth_factor,
                       backoff_factor, growth_interval, adam_step, loss, counter, loss_ring, counter_ring, cursor, ring, loss_slots);
    return check_launch("step_epilogue");
}
