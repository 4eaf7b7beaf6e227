// encoders.hip — spherical-harmonics and frequency encodings for gfx950.
// Replaces shencoder/src/shencoder.cu (:27-382) and freqencoder/src/freqencoder.cu (:30-94).
//
// SH: the reference hard-codes the degree<=8 real SH basis and Jacobian as expanded polynomials of the
// family  Y_l^{±m} = K_l^m * T_l^m(z) * {Re,Im}(x+iy)^m  with T_l^m = d^m P_l/dz^m  (slot l*l+l±m).  Here
// the same family is evaluated through its recurrences, fully unrolled per degree, in fp32; K is a
// 36-entry host table passed by value.  FP parity (not bit parity): tests compare against vectors
// obtained from the reference's own expressions with a 2e-5 relative tolerance.
// One lane = one point; the deg^2 outputs of a lane are contiguous (float4 stores).
#include "s3d_common.hpp"
#include "sh_eval.hpp"
#include <math.h>

namespace s3d {
namespace {

template <uint32_t DEG, bool JAC>
__global__ void __launch_bounds__(256) k_sh_forward(const float* __restrict__ inputs, float* __restrict__ outputs,
                                                    uint32_t B, uint32_t D, ShNorm K, float* __restrict__ dy_dx) {
    constexpr uint32_t C2 = DEG * DEG;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
    float o[C2];
    float jx[JAC ? C2 : 1], jy[JAC ? C2 : 1], jz[JAC ? C2 : 1];
    sh_eval<DEG, JAC>(x, y, z, K, o, jx, jy, jz);
    float* out = outputs + (size_t)b * C2;
#pragma unroll
    for (uint32_t i = 0; i < C2; i++) out[i] = o[i];
    if (JAC) {
        float* j = dy_dx + (size_t)b * 3 * C2;
#pragma unroll
        for (uint32_t i = 0; i < C2; i++) { j[i] = jx[i]; j[C2 + i] = jy[i]; j[2 * C2 + i] = jz[i]; }
    }
}

__global__ void k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                              const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* g = grad + (size_t)b * C2;
    const float* j = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) acc = __builtin_fmaf(g[ch], j[ch], acc);
    grad_inputs[t] = acc;
}

// freqencoder.cu:30-58 — one lane per output element (coalesced stores)
__global__ void k_freq_forward(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C,
                               float* __restrict__ outputs) {
    const float half_pi = 3.141592653589793f / 2;
    const uint64_t total = (uint64_t)B * C;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
        const float* x = inputs + (size_t)b * D;
        float v;
        if (c < D) v = x[c];
        else {
            const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
            v = sinf(ldexpf(x[d], (int)freq) + (float)(col % 2) * half_pi);
        }
        outputs[t] = v;
    }
}

// freqencoder.cu:63-94
__global__ void k_freq_backward(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B,
                                uint32_t D, uint32_t deg, uint32_t C, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* g = grad + (size_t)b * C;
    const float* o = outputs + (size_t)b * C;
    float result = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; f++) {
        result = __builtin_fmaf(ldexpf(1.0f, (int)f), __builtin_fmaf(g[d], o[D + d], -(g[D + d] * o[d])), result);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = result;
}

// Two frequency encodings side by side in ONE fp16 row (TensoRF's colour MLP input, tensoRF/network.py:48-51, 160-166:
// cat([freq(feat), freq(dirs)]) under fp16 autocast): out[b] = [freq_1(a[b]) | freq_2(d[b]) | 0 ...] with the values of
// k_freq_forward (fp32 arithmetic on float(a), rounded to binary16 once — what the autocast Linear's cast makes of the fp32
// encodings).  One lane per PAIR of output columns (one 4-byte store).
// One lane per INPUT element: it loads x once and writes the identity column and the 2 deg sines of its feature (columns D apart;
// neighbouring lanes are neighbouring features: 2-byte stores side by side).  [One lane per output pair — two divisions and two
// branches per value around the same sinf — took 61 us for 1.05e5 rows; the sines themselves are ~35 us of vector issue.]
__global__ void __launch_bounds__(256) k_freq_pack_forward(const _Float16* __restrict__ a, const float* __restrict__ d, uint32_t B,
                                                            uint32_t D1, uint32_t deg1, uint32_t D2, uint32_t deg2, uint32_t ld,
                                                            uint32_t lanes_per_row, _Float16* __restrict__ out,
                                                            const int32_t* __restrict__ n_valid) {
    const float half_pi = 3.141592653589793f / 2;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b = t / lanes_per_row, e = t - b * lanes_per_row;
    if (b >= valid_rows(B, n_valid)) return;
    const uint32_t C1 = D1 + 2 * D1 * deg1, C2 = D2 + 2 * D2 * deg2;
    _Float16* row = out + (size_t)b * ld;
    if (e < D1 + D2) {
        const bool first = e < D1;
        const uint32_t dd = first ? e : e - D1, D = first ? D1 : D2, deg = first ? deg1 : deg2;
        const float x = first ? (float)a[(size_t)b * D1 + dd] : d[(size_t)b * D2 + dd];
        _Float16* o = row + (first ? 0u : C1);
        o[dd] = (_Float16)x;
        for (uint32_t f = 0; f < deg; f++) {
            const float arg = ldexpf(x, (int)f);
            o[D + (2 * f) * D + dd] = (_Float16)sinf(arg + 0.0f * half_pi);
            o[D + (2 * f + 1) * D + dd] = (_Float16)sinf(arg + 1.0f * half_pi);
        }
    } else {  // the spare lanes of a row clear the padding columns
        const uint32_t spare = lanes_per_row - (D1 + D2);
        for (uint32_t c = C1 + C2 + (e - (D1 + D2)); c < ld; c += spare) row[c] = (_Float16)0.0f;
    }
}
// gradient w.r.t. the FIRST input from the packed row's fp16 gradient (k_freq_backward's expression with sin / cos re-computed
// from the input instead of read from the stored fp32 outputs: the same values), written as binary16 rows of `ldg` columns with
// the columns behind D1 zero (the layout s3d_vm_color_backward reads: four 16-byte words per point)
__global__ void __launch_bounds__(256) k_freq_pack_backward(const _Float16* __restrict__ grad, const _Float16* __restrict__ a, uint32_t B,
                                                             uint32_t D1, uint32_t deg1, uint32_t ld, uint32_t ldg,
                                                             _Float16* __restrict__ grad_a, const int32_t* __restrict__ n_valid) {
    const float half_pi = 3.141592653589793f / 2;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= valid_rows(B, n_valid) * ldg) return;
    const uint32_t b = t / ldg, dd = t - b * ldg;
    float result = 0.0f;
    if (dd < D1) {
        const _Float16* g = grad + (size_t)b * ld;
        const float x = (float)a[(size_t)b * D1 + dd];
        result = (float)g[dd];
        g += D1;
        for (uint32_t f = 0; f < deg1; f++) {
            const float arg = ldexpf(x, (int)f);
            const float sv = sinf(arg), cv = sinf(arg + half_pi);
            result = __builtin_fmaf(ldexpf(1.0f, (int)f), __builtin_fmaf((float)g[dd], cv, -((float)g[D1 + dd] * sv)), result);
            g += 2 * D1;
        }
    }
    grad_a[t] = (_Float16)result;
}

template <uint32_t DEG>
int launch_sh(const float* inputs, float* outputs, uint32_t B, uint32_t D, const ShNorm& K, float* dy_dx, hipStream_t st) {
    const dim3 grid(div_up<uint32_t>(B, 256)), block(256);
    if (dy_dx) hipLaunchKernelGGL((k_sh_forward<DEG, true>), grid, block, 0, st, inputs, outputs, B, D, K, dy_dx);
    else hipLaunchKernelGGL((k_sh_forward<DEG, false>), grid, block, 0, st, inputs, outputs, B, D, K, dy_dx);
    return check_launch("sh_encode_forward");
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree,
                                     float* dy_dx, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && outputs, "sh_encode_forward: null pointer");
    S3D_REQUIRE(D == 3, "SH encoder only support input dim == 3");
    S3D_REQUIRE(degree >= 1 && degree <= kMaxDeg, "SH encoder only supports degree in [1, 8]");
    ShNorm K;
    host_sh_norm(degree, K);
    hipStream_t st = as_stream(stream);
    switch (degree) {
        case 1: return launch_sh<1>(inputs, outputs, B, D, K, dy_dx, st);
        case 2: return launch_sh<2>(inputs, outputs, B, D, K, dy_dx, st);
        case 3: return launch_sh<3>(inputs, outputs, B, D, K, dy_dx, st);
        case 4: return launch_sh<4>(inputs, outputs, B, D, K, dy_dx, st);
        case 5: return launch_sh<5>(inputs, outputs, B, D, K, dy_dx, st);
        case 6: return launch_sh<6>(inputs, outputs, B, D, K, dy_dx, st);
        case 7: return launch_sh<7>(inputs, outputs, B, D, K, dy_dx, st);
        default: return launch_sh<8>(inputs, outputs, B, D, K, dy_dx, st);
    }
}

S3D_EXPORT int s3d_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                                      const float* dy_dx, float* grad_inputs, s3d_stream_t stream) {
    (void)inputs;
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: null pointer");
    hipLaunchKernelGGL(k_sh_backward, dim3(div_up<uint32_t>(B * D, 256)), dim3(256), 0, as_stream(stream), grad, B, D,
                       degree * degree, dy_dx, grad_inputs);
    return check_launch("sh_encode_backward");
}

S3D_EXPORT int s3d_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                       float* outputs, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && outputs, "freq_encode_forward: null pointer");
    S3D_REQUIRE(C == D + 2 * D * deg, "freq_encode_forward: C must equal D + 2*D*deg");
    hipLaunchKernelGGL(k_freq_forward, dim3(stream_grid((uint64_t)B * C, 256)), dim3(256), 0, as_stream(stream), inputs,
                       B, D, C, outputs);
    return check_launch("freq_encode_forward");
}

S3D_EXPORT int s3d_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg,
                                        uint32_t C, float* grad_inputs, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && outputs && grad_inputs, "freq_encode_backward: null pointer");
    S3D_REQUIRE(C == D + 2 * D * deg, "freq_encode_backward: C must equal D + 2*D*deg");
    hipLaunchKernelGGL(k_freq_backward, dim3(div_up<uint32_t>(B * D, 256)), dim3(256), 0, as_stream(stream), grad,
                       outputs, B, D, deg, C, grad_inputs);
    return check_launch("freq_encode_backward");
}

S3D_EXPORT int s3d_freq_encode_pack_forward(const uint16_t* a, const float* d, uint32_t B, uint32_t D1, uint32_t deg1, uint32_t D2,
                                            uint32_t deg2, uint32_t ld, uint16_t* out, const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(a && d && out, "freq_encode_pack_forward: null pointer");
    const uint32_t C1 = D1 + 2 * D1 * deg1, C2 = D2 + 2 * D2 * deg2;
    S3D_REQUIRE(D1 >= 1 && D2 >= 1 && ld % 2 == 0 && ld >= C1 + C2, "freq_encode_pack_forward: ld must be even and >= %u", C1 + C2);
    uint32_t lanes = 1;
    while (lanes < D1 + D2 + 1) lanes <<= 1;  // (a power of two >= D1 + D2 + 1: at least one spare lane per row for the padding)
    S3D_REQUIRE((uint64_t)B * lanes < (1ull << 32), "freq_encode_pack_forward: too many rows");
    hipLaunchKernelGGL(k_freq_pack_forward, dim3(div_up<uint32_t>(B * lanes, 256)), dim3(256), 0, as_stream(stream),
                       (const _Float16*)a, d, B, D1, deg1, D2, deg2, ld, lanes, (_Float16*)out, n_valid);
    return check_launch("freq_encode_pack_forward");
}

S3D_EXPORT int s3d_freq_encode_pack_backward(const uint16_t* grad, const uint16_t* a, uint32_t B, uint32_t D1, uint32_t deg1,
                                             uint32_t ld, uint32_t ldg, uint16_t* grad_a, const int32_t* n_valid, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && a && grad_a, "freq_encode_pack_backward: null pointer");
    S3D_REQUIRE(D1 >= 1 && ld >= D1 + 2 * D1 * deg1 && ldg >= D1 && (uint64_t)B * ldg < (1ull << 32), "freq_encode_pack_backward: bad shape");
    hipLaunchKernelGGL(k_freq_pack_backward, dim3(div_up<uint32_t>(B * ldg, 256)), dim3(256), 0, as_stream(stream),
                       (const _Float16*)grad, (const _Float16*)a, B, D1, deg1, ld, ldg, (_Float16*)grad_a, n_valid);
    return check_launch("freq_encode_pack_backward");
}
