// ffmlp.hip — fully fused fp16 MLP on MFMA for gfx950.
// Replaces ffmlp/src/ffmlp.cu (+ the CUTLASS split-K weight-gradient GEMMs) of the reference.
//
// Design (MI355X-first, not a WMMA port)
//  * v_mfma_f32_32x32x16_f16, fp32 accumulation.  Every layer is computed TRANSPOSED:
//        H_out^T [features x batch] = W [features x k] * H_in^T [k x batch]
//    so the weights are the A operand (M = output features) and the activations the B operand
//    (N = 32 batch points per wave).  The C/D fragment of one layer (lane = batch point, registers =
//    features) is, after cvt to half, *exactly* the B fragment of the next layer when the K index of a
//    16-wide k-step is enumerated as  k(h,j) = (j&3) + 8*(j>>2) + 4*h   (h = lane>>5, j = 0..7): the
//    weights are staged into LDS once per workgroup in that permuted order, and the whole network runs
//    register-to-register — no LDS round trip, no shuffles between layers.
//  * Weights live in LDS as ready-made A fragments (1 KiB each, read with one ds_read_b128 per lane).
//  * forward_buffer / backward_buffer keep the reference's [n, B, W] extent but are stored in fragment
//    order (32-point tiles; per tile [W/8][32 points][8 halfs]) so every store/load is one 8-byte access
//    per lane covering 512 contiguous bytes per wave.  They are private scratch between
//    ffmlp_forward and ffmlp_backward, as in the reference.
//  * Backward = one dgrad launch (transposed weights as A fragments, same register chaining) + one
//    wgrad launch covering all layers (blockIdx.y = layer; batch is the MFMA K dimension, operands
//    transposed through LDS) + one reduce/convert launch.  No atomics, deterministic.
// Numerics: fp16 storage between layers, fp32 accumulate (the reference accumulates in fp16 inside
// WMMA; parity target is the dense math of testing/test_ffmlp.py's torch twin, fp16 tolerance).
#include "s3d_common.hpp"

namespace s3d {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

enum { ACT_RELU = 0, ACT_EXP = 1, ACT_SINE = 2, ACT_SIGMOID = 3, ACT_SQUAREPLUS = 4, ACT_SOFTPLUS = 5, ACT_NONE = 6 };
constexpr float kAct = 10.0f;

__device__ __forceinline__ float act_fwd(uint32_t a, float x) {
    switch (a) {
        case ACT_RELU: return x > 0.0f ? x : 0.0f;
        case ACT_EXP: return expf(x);
        case ACT_SINE: return sinf(x);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case ACT_SQUAREPLUS: { float y = x * kAct; return 0.5f * (y + sqrtf(y * y + 4)) / kAct; }
        case ACT_SOFTPLUS: return logf(expf(x * kAct) + 1.0f) / kAct;
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd(uint32_t a, float g, float fwd) {
    switch (a) {
        case ACT_RELU: return fwd > 0.0f ? g : 0.0f;
        case ACT_EXP: return g * fwd;
        case ACT_SIGMOID: return g * (fwd * (1.0f - fwd));
        case ACT_SQUAREPLUS: { float y = fwd * kAct; return g * (y * y / (y * y + 1)); }
        case ACT_SOFTPLUS: return g * (1.0f - expf(-fwd * kAct));
        default: return g;
    }
}

// K-permutation of one 16-wide k-step: element j of lane-half h
__device__ __forceinline__ uint32_t kperm(uint32_t h, uint32_t j) { return (j & 3u) + 8u * (j >> 2) + 4u * h; }

__device__ __forceinline__ float16v mfma(half8 a, half8 b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float16v zero16() {
    float16v z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}

// 8-byte (4 halfs) global access
__device__ __forceinline__ half4 ld4(const _Float16* p) { return *reinterpret_cast<const half4*>(p); }
__device__ __forceinline__ void st4(_Float16* p, half4 v) { *reinterpret_cast<half4*>(p) = v; }

// B fragment of k-step s from a row-major [rows, width] matrix: row = this lane's batch point
__device__ __forceinline__ half8 load_bfrag_rowmajor(const _Float16* row, uint32_t s, uint32_t h) {
    const half4 lo = ld4(row + 16 * s + 4 * h);
    const half4 hi = ld4(row + 16 * s + 8 + 4 * h);
    half8 b;
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
    b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
    return b;
}

// offset (in halfs) inside one 32-point tile of a fragment-ordered ("native") buffer
__device__ __forceinline__ uint32_t native_off(uint32_t mblk, uint32_t q, uint32_t n, uint32_t h) {
    return ((mblk * 4 + q) * 32 + n) * 8 + h * 4;
}

// ------------------------------------------------------------------------------------ forward
// LDS fragment directory: layer 0: MB*KS0 frags | hidden k: MB*KS frags each | last: KS frags
template <int W, bool TRAIN>
__global__ void __launch_bounds__(256) k_ffmlp_forward(const _Float16* __restrict__ X, const _Float16* __restrict__ Wt,
                                                       uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                                                       uint32_t act, uint32_t out_act, _Float16* __restrict__ fwd,
                                                       _Float16* __restrict__ out) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8* frags = reinterpret_cast<half8*>(smem_raw);

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t KS0 = in_dim / 16, NH = n_layers - 1;
    const uint32_t nf0 = MB * KS0, nfh = NH * MB * KS, total = nf0 + nfh + KS;
    const _Float16* w_hid = Wt + (size_t)W * in_dim;
    const _Float16* w_last = w_hid + (size_t)NH * W * W;

    // stage all weights as A fragments (row = output feature, k permuted)
    for (uint32_t f = wave; f < total; f += 4) {
        half8 v;
        if (f < nf0) {
            const uint32_t mblk = f / KS0, s = f % KS0;
            const _Float16* r = Wt + (size_t)(mblk * 32 + n) * in_dim + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = r[kperm(h, j)];
        } else if (f < nf0 + nfh) {
            const uint32_t g = f - nf0, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            const _Float16* r = w_hid + (size_t)k * W * W + (size_t)(mblk * 32 + n) * W + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = r[kperm(h, j)];
        } else {
            const uint32_t s = f - nf0 - nfh;
            const _Float16* r = w_last + (size_t)n * W + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = (n < out_dim) ? r[kperm(h, j)] : (_Float16)0.0f;
        }
        frags[f * 64 + lane] = v;
    }
    __syncthreads();

    const uint32_t ntiles = B / 32;
    for (uint32_t tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const size_t row = (size_t)tile * 32 + n;
        float16v acc[MB];
        half8 bf[KS];
#pragma unroll
        for (uint32_t m = 0; m < MB; m++) acc[m] = zero16();
        const _Float16* xrow = X + row * in_dim;
        for (uint32_t s = 0; s < KS0; s++) {
            const half8 b = load_bfrag_rowmajor(xrow, s, h);
#pragma unroll
            for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(frags[(m * KS0 + s) * 64 + lane], b, acc[m]);
        }
        for (uint32_t layer = 0;; layer++) {
            // activation, fp16 rounding, re-use as next B operand
#pragma unroll
            for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                for (uint32_t r = 0; r < 16; r++) {
                    const float pre = (float)(_Float16)acc[m][r];
                    bf[2 * m + (r >> 3)][r & 7] = (_Float16)act_fwd(act, pre);
                }
            if (TRAIN) {
                _Float16* dst = fwd + (size_t)layer * B * W + (size_t)tile * 32 * W;
#pragma unroll
                for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) {
                        half4 v;
#pragma unroll
                        for (uint32_t e = 0; e < 4; e++) v[e] = bf[2 * m + (q >> 1)][(q & 1) * 4 + e];
                        st4(dst + native_off(m, q, n, h), v);
                    }
            }
            if (layer == NH) break;
            const half8* a = frags + (size_t)(nf0 + layer * MB * KS) * 64;
#pragma unroll
            for (uint32_t m = 0; m < MB; m++) {
                acc[m] = zero16();
#pragma unroll
                for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(a[(m * KS + s) * 64 + lane], bf[s], acc[m]);
            }
        }
        // output layer (rows >= out_dim of the A fragment are zero)
        float16v o = zero16();
        const half8* a = frags + (size_t)(nf0 + nfh) * 64;
#pragma unroll
        for (uint32_t s = 0; s < KS; s++) o = mfma(a[s * 64 + lane], bf[s], o);
        _Float16* orow = out + row * 16;
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            half4 v;
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) v[e] = (_Float16)act_fwd(out_act, (float)(_Float16)o[4 * q + e]);
            st4(orow + 8 * q + 4 * h, v);
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dgrad
// LDS directory: last^T: MB frags (one k-step, K = 16 outputs) | hidden^T k: MB*KS each | first^T: IMB*KS
template <int W>
__global__ void __launch_bounds__(256) k_ffmlp_dgrad(const _Float16* __restrict__ grad, const _Float16* __restrict__ Wt,
                                                     const _Float16* __restrict__ fwd, uint32_t B, uint32_t in_dim,
                                                     uint32_t out_dim, uint32_t n_layers, uint32_t act,
                                                     _Float16* __restrict__ bwd, _Float16* __restrict__ grad_inputs) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8* frags = reinterpret_cast<half8*>(smem_raw);

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t NH = n_layers - 1;
    const uint32_t IMB = (in_dim + 31) / 32;
    const uint32_t nfl = MB, nfh = NH * MB * KS, nf0 = grad_inputs ? IMB * KS : 0, total = nfl + nfh + nf0;
    const _Float16* w_hid = Wt + (size_t)W * in_dim;
    const _Float16* w_last = w_hid + (size_t)NH * W * W;

    for (uint32_t f = wave; f < total; f += 4) {
        half8 v;
        if (f < nfl) {  // A[i = hidden feature][k = output]  = W_last[k][i]
            const uint32_t i = f * 32 + n;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t k = kperm(h, j);
                v[j] = (k < out_dim) ? w_last[(size_t)k * W + i] : (_Float16)0.0f;
            }
        } else if (f < nfl + nfh) {  // A[i = in feature][k = out feature] = W_k[k][i]
            const uint32_t g = f - nfl, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            const _Float16* wk = w_hid + (size_t)k * W * W;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = wk[(size_t)(16 * s + kperm(h, j)) * W + mblk * 32 + n];
        } else {  // A[i = network input][k = first hidden feature] = W_0[k][i]
            const uint32_t g = f - nfl - nfh, mblk = g / KS, s = g % KS;
            const uint32_t i = mblk * 32 + n;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
                v[j] = (i < in_dim) ? Wt[(size_t)(16 * s + kperm(h, j)) * in_dim + i] : (_Float16)0.0f;
        }
        frags[f * 64 + lane] = v;
    }
    __syncthreads();

    const uint32_t ntiles = B / 32;
    for (uint32_t tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const size_t row = (size_t)tile * 32 + n;
        float16v acc[MB];
        half8 bf[KS];
        // through the output layer: K = 16 outputs = one k-step
        {
            const half8 g = load_bfrag_rowmajor(grad + row * 16, 0, h);
#pragma unroll
            for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(frags[m * 64 + lane], g, zero16());
        }
        for (uint32_t k = 0;; k++) {
            // activation transfer with the stored post-activation values of hidden layer (NH - k)
            const _Float16* f = fwd + (size_t)(NH - k) * B * W + (size_t)tile * 32 * W;
            _Float16* dst = bwd + (size_t)k * B * W + (size_t)tile * 32 * W;
#pragma unroll
            for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const half4 fv = ld4(f + native_off(m, q, n, h));
                    half4 v;
#pragma unroll
                    for (uint32_t e = 0; e < 4; e++) {
                        const float g = (float)(_Float16)acc[m][4 * q + e];
                        v[e] = (_Float16)act_bwd(act, g, (float)fv[e]);
                        bf[2 * m + (q >> 1)][(q & 1) * 4 + e] = v[e];
                    }
                    st4(dst + native_off(m, q, n, h), v);
                }
            if (k == NH) break;
            const half8* a = frags + (size_t)(nfl + (NH - 1 - k) * MB * KS) * 64;
#pragma unroll
            for (uint32_t m = 0; m < MB; m++) {
                acc[m] = zero16();
#pragma unroll
                for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(a[(m * KS + s) * 64 + lane], bf[s], acc[m]);
            }
        }
        if (grad_inputs) {
            const half8* a = frags + (size_t)(nfl + nfh) * 64;
            for (uint32_t m = 0; m < IMB; m++) {
                float16v gi = zero16();
#pragma unroll
                for (uint32_t s = 0; s < KS; s++) gi = mfma(a[(m * KS + s) * 64 + lane], bf[s], gi);
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t feat = m * 32 + 8 * q + 4 * h;
                    if (feat < in_dim) {
                        half4 v;
#pragma unroll
                        for (uint32_t e = 0; e < 4; e++) v[e] = (_Float16)gi[4 * q + e];
                        st4(grad_inputs + row * in_dim + feat, v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward: wgrad
// dW[o][i] = sum_b G[b][o] * X[b][i]; batch is the MFMA K dimension.  One (G, X) pair per layer.
constexpr uint32_t kMaxMlpLayers = 8;
struct WgradLayer {
    const _Float16* G;  // gradient w.r.t. the layer's pre-activation output
    const _Float16* X;  // the layer's input
    uint32_t g_native, x_native;  // fragment-ordered (W wide) or row-major
    uint32_t Fo, Fi;              // real feature counts
    uint32_t w_off;               // offset of this layer's matrix in the flat weight vector
};
struct WgradPlan {
    WgradLayer layer[kMaxMlpLayers];
    uint32_t n;
};

__device__ __forceinline__ uint32_t tile_elem(uint32_t native, uint32_t F, uint32_t b, uint32_t f) {
    if (native) return (((f >> 5) * 4 + ((f >> 3) & 3)) * 32 + b) * 8 + ((f >> 2) & 1) * 4 + (f & 3);
    return b * F + f;
}

constexpr uint32_t kWgradPad = 64;  // partial matrices are stored [64][64] fp32 regardless of W

template <int W>
__global__ void __launch_bounds__(256) k_ffmlp_wgrad(WgradPlan plan, uint32_t B, float* __restrict__ partial) {
    constexpr uint32_t MAXB = (W + 31) / 32;  // 32-blocks per side
    __shared__ __attribute__((aligned(16))) _Float16 tiles[4][2][32 * W];
    __shared__ float red[kWgradPad * kWgradPad];

    const WgradLayer L = plan.layer[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t fl = lane & 31, h = lane >> 5;
    const uint32_t gF = L.g_native ? W : L.Fo, xF = L.x_native ? W : L.Fi;  // storage widths
    const uint32_t MBo = (L.Fo + 31) / 32, NBi = (L.Fi + 31) / 32;

    float16v acc[MAXB][MAXB];
#pragma unroll
    for (uint32_t a = 0; a < MAXB; a++)
#pragma unroll
        for (uint32_t b = 0; b < MAXB; b++) acc[a][b] = zero16();

    _Float16* tg = tiles[wave][0];
    _Float16* tx = tiles[wave][1];
    const uint32_t ntiles = B / 32;  // B % 128 == 0: the four waves of a block always have a tile together
    for (uint32_t base = blockIdx.x * 4; base < ntiles; base += gridDim.x * 4) {
        const uint32_t tile = base + wave;
        // linear 16-byte copies of the two 32-row tiles into LDS
        {
            const uint4* sg = reinterpret_cast<const uint4*>(L.G + (size_t)tile * 32 * gF);
            const uint4* sx = reinterpret_cast<const uint4*>(L.X + (size_t)tile * 32 * xF);
            for (uint32_t i = lane; i < 32 * gF / 8; i += 64) reinterpret_cast<uint4*>(tg)[i] = sg[i];
            for (uint32_t i = lane; i < 32 * xF / 8; i += 64) reinterpret_cast<uint4*>(tx)[i] = sx[i];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t s = 0; s < 2; s++) {
            half8 af[MAXB], bfr[MAXB];
#pragma unroll
            for (uint32_t m = 0; m < MAXB; m++) {
                const uint32_t o = m * 32 + fl;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++)
                    af[m][j] = (m < MBo && o < L.Fo) ? tg[tile_elem(L.g_native, gF, 16 * s + 8 * h + j, o)] : (_Float16)0.0f;
                const uint32_t i = m * 32 + fl;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++)
                    bfr[m][j] = (m < NBi && i < L.Fi) ? tx[tile_elem(L.x_native, xF, 16 * s + 8 * h + j, i)] : (_Float16)0.0f;
            }
#pragma unroll
            for (uint32_t mo = 0; mo < MAXB; mo++)
#pragma unroll
                for (uint32_t ni = 0; ni < MAXB; ni++)
                    if (mo < MBo && ni < NBi) acc[mo][ni] = mfma(af[mo], bfr[ni], acc[mo][ni]);
        }
        __syncthreads();
    }
    // reduce the four waves through LDS, then one coalesced partial per block
    for (uint32_t i = threadIdx.x; i < kWgradPad * kWgradPad; i += 256) red[i] = 0.0f;
    __syncthreads();
    for (uint32_t w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (uint32_t mo = 0; mo < MAXB; mo++)
#pragma unroll
                for (uint32_t ni = 0; ni < MAXB; ni++)
#pragma unroll
                    for (uint32_t r = 0; r < 16; r++) {
                        const uint32_t o = mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, i = ni * 32 + fl;
                        red[o * kWgradPad + i] += acc[mo][ni][r];
                    }
        }
        __syncthreads();
    }
    float* dst = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kWgradPad * kWgradPad;
    for (uint32_t i = threadIdx.x; i < kWgradPad * kWgradPad; i += 256) dst[i] = red[i];
}

__global__ void k_ffmlp_wgrad_reduce(WgradPlan plan, uint32_t nblk, const float* __restrict__ partial,
                                     _Float16* __restrict__ grad_weights) {
    const WgradLayer L = plan.layer[blockIdx.y];
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.Fo * L.Fi) return;
    const uint32_t o = e / L.Fi, i = e - o * L.Fi;
    const float* p = partial + (size_t)blockIdx.y * nblk * kWgradPad * kWgradPad + o * kWgradPad + i;
    // independent partial sums keep several loads in flight (a single dependent chain was latency-bound: 38 us)
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    uint32_t b = 0;
    for (; b + 4 <= nblk; b += 4) {
        s0 += p[(size_t)(b + 0) * kWgradPad * kWgradPad];
        s1 += p[(size_t)(b + 1) * kWgradPad * kWgradPad];
        s2 += p[(size_t)(b + 2) * kWgradPad * kWgradPad];
        s3 += p[(size_t)(b + 3) * kWgradPad * kWgradPad];
    }
    for (; b < nblk; b++) s0 += p[(size_t)b * kWgradPad * kWgradPad];
    grad_weights[L.w_off + e] = (_Float16)((s0 + s1) + (s2 + s3));
}

constexpr uint32_t kWgradBlocks = 256;

int check_shape(uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t W, uint32_t n_layers) {
    S3D_REQUIRE(W == 32 || W == 64, "ffmlp: hidden_dim %u not supported by the MFMA path (32 or 64)", W);
    S3D_REQUIRE(in_dim > 0 && in_dim % 16 == 0 && in_dim <= 64, "ffmlp: input_dim must be 16*m, m in 1..4 (got %u)", in_dim);
    S3D_REQUIRE(out_dim >= 1 && out_dim <= 16, "FFMLP current only supports output dim <= 16, but got %u", out_dim);
    S3D_REQUIRE(n_layers >= 2 && n_layers + 1 <= kMaxMlpLayers, "ffmlp: num_layers must be in [2, %u]", kMaxMlpLayers - 1);
    S3D_REQUIRE(B % 128 == 0, "ffmlp: batch size must be a multiple of 128 (got %u)", B);
    return S3D_OK;
}

template <int W>
int launch_forward(const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                   uint32_t act, uint32_t out_act, _Float16* fwd, _Float16* out, hipStream_t st) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    const uint32_t nfr = MB * (in_dim / 16) + (n_layers - 1) * MB * KS + KS;
    const size_t smem = (size_t)nfr * 64 * sizeof(half8);
    const uint32_t ntiles = B / 32;
    uint32_t grid = div_up<uint32_t>(ntiles, 4);
    if (grid > 1024) grid = 1024;
    if (fwd) hipLaunchKernelGGL((k_ffmlp_forward<W, true>), dim3(grid), dim3(256), smem, st, X, Wt, B, in_dim, out_dim, n_layers, act, out_act, fwd, out);
    else hipLaunchKernelGGL((k_ffmlp_forward<W, false>), dim3(grid), dim3(256), smem, st, X, Wt, B, in_dim, out_dim, n_layers, act, out_act, fwd, out);
    return check_launch("ffmlp_forward");
}

template <int W>
int launch_backward(const _Float16* grad, const _Float16* X, const _Float16* Wt, const _Float16* fwd, uint32_t B,
                    uint32_t in_dim, uint32_t out_dim, uint32_t n_layers, uint32_t act, _Float16* bwd,
                    _Float16* grad_inputs, _Float16* grad_weights, float* partial, hipStream_t st) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    const uint32_t NH = n_layers - 1;
    const uint32_t nfr = MB + NH * MB * KS + (grad_inputs ? ((in_dim + 31) / 32) * KS : 0);
    const size_t smem = (size_t)nfr * 64 * sizeof(half8);
    const uint32_t ntiles = B / 32;
    uint32_t grid = div_up<uint32_t>(ntiles, 4);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL((k_ffmlp_dgrad<W>), dim3(grid), dim3(256), smem, st, grad, Wt, fwd, B, in_dim, out_dim, n_layers,
                       act, bwd, grad_inputs);

    // weight gradients, all layers in one launch
    WgradPlan plan;
    memset(&plan, 0, sizeof(plan));
    plan.n = n_layers + 1;
    const size_t BW = (size_t)B * W;
    // layer 0: G = dPre[0] = bwd[NH], X = inputs
    plan.layer[0] = WgradLayer{bwd + NH * BW, X, 1u, 0u, (uint32_t)W, in_dim, 0u};
    for (uint32_t m = 0; m < NH; m++)  // hidden matrix m: fwd[m] -> layer m+1, G = bwd[NH-1-m]
        plan.layer[1 + m] = WgradLayer{bwd + (size_t)(NH - 1 - m) * BW, fwd + (size_t)m * BW, 1u, 1u, (uint32_t)W, (uint32_t)W,
                                       (uint32_t)(W * in_dim + m * W * W)};
    plan.layer[n_layers] = WgradLayer{grad, fwd + (size_t)NH * BW, 0u, 1u, 16u, (uint32_t)W,
                                      (uint32_t)(W * in_dim + NH * W * W)};
    // the last layer's matrix is [out_pad=16, W]; rows >= out_dim receive the (zero) gradient of the padding
    uint32_t nblk = div_up<uint32_t>(ntiles, 4);
    if (nblk > kWgradBlocks) nblk = kWgradBlocks;
    hipLaunchKernelGGL((k_ffmlp_wgrad<W>), dim3(nblk, plan.n), dim3(256), 0, st, plan, B, partial);
    hipLaunchKernelGGL(k_ffmlp_wgrad_reduce, dim3(div_up<uint32_t>(W * W, 256), plan.n), dim3(256), 0, st, plan, nblk,
                       (const float*)partial, grad_weights);
    return check_launch("ffmlp_backward");
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                                 uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                 uint32_t output_activation, uint16_t* forward_buffer, uint16_t* outputs,
                                 s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && weights && outputs, "ffmlp_forward: null pointer");
    if (int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers)) return rc;
    S3D_REQUIRE(output_dim == 16, "ffmlp_forward: the output must be padded to 16 columns (ffmlp.py:117)");
    const _Float16* X = (const _Float16*)inputs; const _Float16* Wt = (const _Float16*)weights;
    _Float16* fb = (_Float16*)forward_buffer; _Float16* o = (_Float16*)outputs;
    if (hidden_dim == 64) return launch_forward<64>(X, Wt, B, input_dim, output_dim, num_layers, activation, output_activation, fb, o, as_stream(stream));
    return launch_forward<32>(X, Wt, B, input_dim, output_dim, num_layers, activation, output_activation, fb, o, as_stream(stream));
}

S3D_EXPORT int s3d_ffmlp_inference(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                                   uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                   uint32_t output_activation, uint16_t* inference_buffer, uint16_t* outputs,
                                   s3d_stream_t stream) {
    (void)inference_buffer;
    return s3d_ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                             output_activation, nullptr, outputs, stream);
}

S3D_EXPORT size_t s3d_ffmlp_backward_workspace_size(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                                    uint32_t num_layers) {
    (void)input_dim; (void)output_dim; (void)hidden_dim;
    return (size_t)(num_layers + 1) * kWgradBlocks * kWgradPad * kWgradPad * sizeof(float);
}

S3D_EXPORT int s3d_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                                  const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                  uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                  uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer,
                                  uint16_t* grad_inputs, uint16_t* grad_weights, void* workspace, size_t workspace_bytes,
                                  s3d_stream_t stream) {
    (void)output_activation;
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && inputs && weights && forward_buffer && backward_buffer && grad_weights,
                "ffmlp_backward: null pointer");
    if (int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers)) return rc;
    S3D_REQUIRE(output_dim == 16, "ffmlp_backward: the output must be padded to 16 columns (ffmlp.py:117)");
    S3D_REQUIRE(activation != ACT_SINE, "ffmlp_backward: sine needs pre-activations, unsupported (utils.h:546-550)");
    S3D_REQUIRE(!calc_grad_inputs || grad_inputs, "ffmlp_backward: grad_inputs requested but null");
    S3D_REQUIRE(workspace && workspace_bytes >= s3d_ffmlp_backward_workspace_size(input_dim, output_dim, hidden_dim, num_layers),
                "ffmlp_backward: workspace too small");
    _Float16* gi = calc_grad_inputs ? (_Float16*)grad_inputs : nullptr;
    if (hidden_dim == 64)
        return launch_backward<64>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights,
                                   (const _Float16*)forward_buffer, B, input_dim, output_dim, num_layers, activation,
                                   (_Float16*)backward_buffer, gi, (_Float16*)grad_weights, (float*)workspace, as_stream(stream));
    return launch_backward<32>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights,
                               (const _Float16*)forward_buffer, B, input_dim, output_dim, num_layers, activation,
                               (_Float16*)backward_buffer, gi, (_Float16*)grad_weights, (float*)workspace, as_stream(stream));
}

S3D_EXPORT int s3d_ffmlp_allocate_splitk(size_t n) { (void)n; return S3D_OK; }
S3D_EXPORT int s3d_ffmlp_free_splitk(void) { return S3D_OK; }
