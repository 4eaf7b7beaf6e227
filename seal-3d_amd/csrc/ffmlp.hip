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
#include "sh_eval.hpp"
#include <stdlib.h>

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

// Activation selected at COMPILE time when ACT >= 0 (the ReLU instantiations), at run time otherwise.  The kernels apply it
// inside fully unrolled 32-element loops: with a run-time switch every element carried all seven branches (expf, sinf, logf
// inlined) and the kernels grew to ~30k instructions — far beyond the instruction cache.
template <int ACT> __device__ __forceinline__ float act_fwd_t(uint32_t a, float x) {
    if constexpr (ACT >= 0) return act_fwd((uint32_t)ACT, x);
    else return act_fwd(a, x);
}
template <int ACT> __device__ __forceinline__ float act_bwd_t(uint32_t a, float g, float fwd) {
    if constexpr (ACT >= 0) return act_bwd((uint32_t)ACT, g, fwd);
    else return act_bwd(a, g, fwd);
}

// ReLU on PACKED halves (the hot networks): max(x, 0) is one v_pk_max_f16 per two elements, and the backward gate
// "g if a != 0 else 0" three packed integer ops (a = relu(.) is +0 exactly where the gate is closed) — instead of a
// float round trip, compare and select per element.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2p __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2p cvt2(float a, float b) {
    float2v f;
    f[0] = a; f[1] = b;
    return __builtin_convertvector(f, half2p);  // round to nearest even, like (_Float16)a
}
__device__ __forceinline__ half2p relu2(half2p x) {  // v_pk_max_f16 (NaN -> 0 like `x > 0 ? x : 0`)
    uint32_t xb, r;
    __builtin_memcpy(&xb, &x, 4);
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(xb), "v"(0u));
    half2p o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}
__device__ __forceinline__ half2p gate2(half2p g, half2p a) {  // g where a > 0, else +0   (a = relu(.) >= 0 or -0)
    uint32_t ab, gb, m;
    __builtin_memcpy(&ab, &a, 4);
    __builtin_memcpy(&gb, &g, 4);
    ab &= 0x7fff7fffu;                                                          // -0 counts as closed
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(ab), "v"(0x00010001u));        // 0 / 1 per half
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(0xffffffffu));      // 0 / 0xffff per half
    gb &= m;
    half2p r;
    __builtin_memcpy(&r, &gb, 4);
    return r;
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

__device__ __forceinline__ half8 zero_half8() {
    half8 z;
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = (_Float16)0.0f;
    return z;
}

// B fragment of k-step s from a row-major [rows, width] matrix: row = this lane's batch point
__device__ __forceinline__ half8 load_bfrag_rowmajor(const _Float16* row, uint32_t s, uint32_t h) {
    const half4 lo = ld4(row + 16 * s + 4 * h);
    const half4 hi = ld4(row + 16 * s + 8 + 4 * h);
    half8 b;
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
    b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
    return b;
}

// Network input / input gradient in the grid encoder's own layout: level-major [L][B][2] (feature f = 2*level + channel,
// in_dim = 2L).  Reading it here (and writing the input gradient in it) removes the [L,B,C] <-> [B,L*C] permute copies that
// gridencoder/grid.py:57,71 of the reference makes in each direction.  Lanes of a half-wave still touch 128 contiguous bytes.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 load_bfrag_levelmajor(const _Float16* X, uint32_t B, size_t row, uint32_t s, uint32_t h) {
    const uint32_t l0 = 8 * s + 2 * h;
    const half2v a = *reinterpret_cast<const half2v*>(X + ((size_t)(l0 + 0) * B + row) * 2);
    const half2v b = *reinterpret_cast<const half2v*>(X + ((size_t)(l0 + 1) * B + row) * 2);
    const half2v c = *reinterpret_cast<const half2v*>(X + ((size_t)(l0 + 4) * B + row) * 2);
    const half2v d = *reinterpret_cast<const half2v*>(X + ((size_t)(l0 + 5) * B + row) * 2);
    half8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}
__device__ __forceinline__ half8 load_bfrag_input(const _Float16* X, uint32_t layout, uint32_t B, uint32_t in_dim, size_t row,
                                                  uint32_t s, uint32_t h) {
    return layout ? load_bfrag_levelmajor(X, B, row, s, h) : load_bfrag_rowmajor(X + row * in_dim, s, h);
}
// 4 consecutive features [feat, feat+4) of one point's input gradient
__device__ __forceinline__ void store_grad_input(_Float16* G, uint32_t layout, uint32_t B, uint32_t in_dim, size_t row,
                                                 uint32_t feat, half4 v) {
    if (layout) {
        half2v a, b;
        a[0] = v[0]; a[1] = v[1]; b[0] = v[2]; b[1] = v[3];
        *reinterpret_cast<half2v*>(G + ((size_t)(feat / 2) * B + row) * 2) = a;
        *reinterpret_cast<half2v*>(G + ((size_t)(feat / 2 + 1) * B + row) * 2) = b;
    } else {
        st4(G + row * in_dim + feat, v);
    }
}

// offset (in halfs) inside one 32-point tile of a fragment-ordered ("native") buffer
__device__ __forceinline__ uint32_t native_off(uint32_t mblk, uint32_t q, uint32_t n, uint32_t h) {
    return ((mblk * 4 + q) * 32 + n) * 8 + h * 4;
}

// NGP "mid" head of the density network (nerf/network_ff.py:55-96: sigma = trunc_exp(h[:, 0]); colour-net input =
// [half(SH_4(d)) | h[:, 1:] | 0]) folded into the MLP kernels (seal3d_hip.h: mid_* arguments of s3d_ffmlp_forward/backward).
// Same arithmetic and rounding points as k_ngp_mid_forward / k_ngp_mid_backward (ngp_head.hip).
struct MidFwd {
    const float* dirs;   // [B, 3]
    float* sigma;        // [B] out
    _Float16* cin;       // [B, 32] out; nullptr = head off
    _Float16* h0;        // [B] out: the pre-activation of sigma (its gradient needs exp(clamp(h0)))
    ShNorm K;
};
struct MidBwd {
    const float* g_sigma;   // [B] or nullptr
    const _Float16* g_cin;  // [B, 32]; nullptr = head off
    const _Float16* h0;     // [B]
};
constexpr uint32_t kMidRow = 40;  // halfs per row of the per-wave [32 points][32 columns] staging tile (80 B: 16-byte aligned)

// output-gradient fragment of the density network from the head's gradients: output k = 0 is sigma's pre-activation, 1..15 the
// geometry features = columns 16..30 of the colour-net input
__device__ __forceinline__ half8 mid_grad_fragment(const MidBwd& mb, size_t row, uint32_t h) {
    const half8 c2 = *reinterpret_cast<const half8*>(mb.g_cin + row * 32 + 16);
    const half8 c3 = *reinterpret_cast<const half8*>(mb.g_cin + row * 32 + 24);
    _Float16 col[16];  // col[i] = d cin[16 + i]
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) { col[i] = c2[i]; col[8 + i] = c3[i]; }
    half8 lo, hi;  // the fragments of the two lane halves (k = (j&3) + 8(j>>2) + 4h)
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t k0 = (j & 3u) + 8u * (j >> 2), k1 = k0 + 4u;
        lo[j] = k0 == 0 ? (_Float16)0.0f : col[k0 - 1];
        hi[j] = col[k1 - 1];
    }
    half8 g = h ? hi : lo;
    if (h == 0) {
        const float h0 = (float)mb.h0[row];
        g[0] = (_Float16)(mb.g_sigma ? mb.g_sigma[row] * expf(fminf(15.0f, fmaxf(-15.0f, h0))) : 0.0f);  // activation.py:13-16
    }
    return g;
}

// ------------------------------------------------------------------------------------ forward
// LDS fragment directory: layer 0: MB*KS0 frags | hidden k: MB*KS frags each | last: KS frags
template <int W, bool TRAIN, int ACT, int OACT, int KS0T>  // KS0T: layer-0 k-steps known at compile time (2 / 4: the hot input widths 32 / 64), 0 = run-time
__global__ void __launch_bounds__(256) k_ffmlp_forward(const _Float16* __restrict__ X, const _Float16* __restrict__ Wt,
                                                       uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                                                       uint32_t act, uint32_t out_act, _Float16* __restrict__ fwd,
                                                       _Float16* __restrict__ out, uint32_t in_layout,
                                                       const int32_t* __restrict__ n_valid, float* __restrict__ rgb_head,
                                                       const MidFwd mid) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8* frags = reinterpret_cast<half8*>(smem_raw);

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t KS0 = in_dim / 16, NH = n_layers - 1;
    const uint32_t nf0 = MB * KS0, nfh = NH * MB * KS, total = nf0 + nfh + KS;
    _Float16* Tm = reinterpret_cast<_Float16*>(frags + (size_t)total * 64) + (size_t)wave * 32 * kMidRow;  // (mid head only)
    const _Float16* w_hid = Wt + (size_t)W * in_dim;
    const _Float16* w_last = w_hid + (size_t)NH * W * W;

    // stage all weights as A fragments (row = output feature, k permuted).  Element j of lane-half h is k = (j&3) + 8(j>>2) + 4h:
    // two runs of four consecutive halfs, i.e. two 8-byte loads per fragment (rows are 32-byte aligned: widths are multiples
    // of 16) instead of eight 2-byte ones
#pragma unroll 1
    for (uint32_t f = wave; f < total; f += 4) {
        const _Float16* r;
        bool live = true;
        if (f < nf0) {
            const uint32_t mblk = f / KS0, s = f % KS0;
            r = Wt + (size_t)(mblk * 32 + n) * in_dim + 16 * s;
        } else if (f < nf0 + nfh) {
            const uint32_t g = f - nf0, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            r = w_hid + (size_t)k * W * W + (size_t)(mblk * 32 + n) * W + 16 * s;
        } else {
            const uint32_t s = f - nf0 - nfh;
            live = n < out_dim;
            r = w_last + (size_t)(live ? n : 0) * W + 16 * s;
        }
        frags[f * 64 + lane] = live ? load_bfrag_rowmajor(r, 0, h) : zero_half8();
    }
    __syncthreads();

    const uint32_t ntiles = valid_rows(B, n_valid) / 32;
    // the NEXT tile's inputs (and view directions, for the density head) are requested before this tile's arithmetic: a wave
    // walks 3 (training batch) to 15 (inference-loop batch) tiles, and with three waves per SIMD an exposed load latency per
    // tile was most of a tile's time.  Only for the input widths instantiated with a compile-time step count: with a run-time
    // count every layer-0 MFMA sits in its own branch and the accumulators are shuttled between the register files.
    constexpr uint32_t KP = KS0T > 0 ? KS0T : 1;
    const uint32_t tstride = gridDim.x * 4;
    half8 bin[KP];
    float dir[3] = {0.0f, 0.0f, 0.0f};
    auto request = [&](uint32_t t, half8 (&dst)[KP], float (&d)[3]) {
        const size_t r = (size_t)t * 32 + n;
        if constexpr (KS0T > 0) {
#pragma unroll
            for (uint32_t s = 0; s < KP; s++) dst[s] = load_bfrag_input(X, in_layout, B, in_dim, r, s, h);
        }
        if (mid.cin) { d[0] = mid.dirs[r * 3]; d[1] = mid.dirs[r * 3 + 1]; d[2] = mid.dirs[r * 3 + 2]; }
    };
    uint32_t tile = blockIdx.x * 4 + wave;
    if (tile < ntiles) request(tile, bin, dir);
    for (; tile < ntiles; tile += tstride) {
        const size_t row = (size_t)tile * 32 + n;
        float16v acc[MB];
        half8 bf[KS];
        half8 bnx[KP];
        float dnx[3] = {0.0f, 0.0f, 0.0f};
        if (tile + tstride < ntiles) request(tile + tstride, bnx, dnx);
#pragma unroll
        for (uint32_t m = 0; m < MB; m++) acc[m] = zero16();
        if constexpr (KS0T > 0) {
#pragma unroll
            for (uint32_t s = 0; s < KP; s++) {
#pragma unroll
                for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(frags[(m * KP + s) * 64 + lane], bin[s], acc[m]);
            }
#pragma unroll
            for (uint32_t s = 0; s < KP; s++) bin[s] = bnx[s];
        } else {
            for (uint32_t s = 0; s < KS0; s++) {
                const half8 b = load_bfrag_input(X, in_layout, B, in_dim, row, s, h);
#pragma unroll
                for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(frags[(m * KS0 + s) * 64 + lane], b, acc[m]);
            }
        }
        const float dx = dir[0], dy = dir[1], dz = dir[2];
        dir[0] = dnx[0]; dir[1] = dnx[1]; dir[2] = dnx[2];
        for (uint32_t layer = 0;; layer++) {
            // activation, fp16 rounding, re-use as next B operand
#pragma unroll
            for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                for (uint32_t r = 0; r < 16; r += 2) {
                    if constexpr (ACT == ACT_RELU) {
                        const half2p v = relu2(cvt2(acc[m][r], acc[m][r + 1]));
                        bf[2 * m + (r >> 3)][r & 7] = v[0];
                        bf[2 * m + (r >> 3)][(r & 7) + 1] = v[1];
                    } else {
#pragma unroll
                        for (uint32_t q = r; q < r + 2; q++) {
                            const float pre = (float)(_Float16)acc[m][q];
                            bf[2 * m + (q >> 3)][q & 7] = (_Float16)act_fwd_t<ACT>(act, pre);
                        }
                    }
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
        if (mid.cin) {
            // density head: the lanes' output features go through a per-wave LDS tile so that every colour-net input row
            // leaves as whole 16-byte pieces: [half(SH_4(d)) (16) | outputs 1..15 | 0]
#pragma unroll
            for (uint32_t q = 0; q < 2; q++)
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) {
                    const uint32_t f = 8 * q + 4 * h + e;  // output feature of o[4 q + e]
                    const _Float16 v = (_Float16)act_fwd_t<OACT>(out_act, (float)(_Float16)o[4 * q + e]);
                    if (f == 0) {
                        mid.sigma[row] = expf((float)v);  // activation.py:8-11
                        mid.h0[row] = v;
                    } else {
                        Tm[n * kMidRow + 15 + f] = v;
                    }
                }
            if (h == 1) Tm[n * kMidRow + 31] = (_Float16)0.0f;
            {
                const float x = dx, y = dy, z = dz;
                float sh[16], j0[1], j1[1], j2[1];
                sh_eval<4, false>(x, y, z, mid.K, sh, j0, j1, j2);
                // (two separate half-vectors and a select per element: `h ? sh[8 + i] : sh[i]` is turned into ONE indexed read of
                //  sh[], i.e. sixteen scratch stores and eight scratch loads per lane and tile)
                half8 vlo, vhi;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) { vlo[i] = (_Float16)sh[i]; vhi[i] = (_Float16)sh[8 + i]; }
                uint32_t wl[4], wh[4], wv[4];
                __builtin_memcpy(wl, &vlo, 16);
                __builtin_memcpy(wh, &vhi, 16);
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) wv[i] = h ? wh[i] : wl[i];
                half8 v;
                __builtin_memcpy(&v, wv, 16);
                *reinterpret_cast<half8*>(Tm + n * kMidRow + 8 * h) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t c = 0; c < 2; c++)
                *reinterpret_cast<half8*>(mid.cin + row * 32 + 8 * (2 * h + c)) =
                    *reinterpret_cast<const half8*>(Tm + n * kMidRow + 8 * (2 * h + c));
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (rgb_head) {
            // colour head (network_ff.py:103 `torch.sigmoid(h)` on the fp16 output, then compositing in fp32): outputs 0..2 go
            // out as fp32 sigmoid values, rounded where the fp16 op sequence rounds — the [B, 16] fp16 tensor never exists
            if (h == 0) {
#pragma unroll
                for (uint32_t c = 0; c < 3; c++) {
                    const _Float16 v = (_Float16)act_fwd_t<OACT>(out_act, (float)(_Float16)o[c]);
                    rgb_head[row * 3 + c] = (float)(_Float16)(1.0f / (1.0f + expf(-(float)v)));
                }
            }
            continue;
        }
        _Float16* orow = out + row * 16;
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            half4 v;
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) v[e] = (_Float16)act_fwd_t<OACT>(out_act, (float)(_Float16)o[4 * q + e]);
            st4(orow + 8 * q + 4 * h, v);
        }
    }
}

// ------------------------------------------------------------------------------------ inference: density + colour network in one launch
// The inference loop (nerf/renderer.py:341-367) calls the two networks back to back on the same rows: density net -> head
// (trunc_exp, SH_4(d), the colour-net input row) -> colour net -> sigmoid.  With one launch per network the [B, 32] colour
// input crossed HBM twice per loop iteration and every workgroup staged weights twice.  Here a wave keeps the row tile it has
// just built in its LDS staging tile and feeds the colour network from there: same MFMA sequence per network, same rounding
// points (tests/test_gpu_ffmlp.py::test_ngp_pair_matches_the_two_launches: sigma bit for bit, rgb bit for bit wherever this
// kernel's inlined copy of sh_eval rounds like the other one's).  W = 64, 32-wide inputs, ReLU.
#ifndef S3D_PAIR_WAVES  // waves per workgroup of k_ffmlp_ngp_pair (they share one staged copy of both networks' weights)
#define S3D_PAIR_WAVES 8
#endif
constexpr uint32_t kPairWaves = S3D_PAIR_WAVES;
struct PairNets {
    const _Float16* Ws;   // density network  [64*32 | (nl_s-1)*64*64 | 16*64]
    const _Float16* Wc;   // colour network   [64*IN_C | (nl_c-1)*64*64 | 16*64]
    uint32_t nl_s, nl_c;  // hidden layers of each
    const _Float16* enc_c;  // SEAL: the second encoder's features, level-major [16][B][2]
};
// one network on the fragments `a0` (layer 0: MB * KPX | hidden: NH * MB * KS | last: KS), layer-0 operands in b0
template <uint32_t KPX>
__device__ __forceinline__ float16v pair_network(const half8* a0, uint32_t NH, const half8 (&b0)[KPX], uint32_t lane) {
    constexpr uint32_t MB = 2, KS = 4, nf0 = MB * KPX;
    float16v acc[MB];
    half8 bf[KS];
#pragma unroll
    for (uint32_t m = 0; m < MB; m++) acc[m] = zero16();
#pragma unroll
    for (uint32_t s = 0; s < KPX; s++) {
#pragma unroll
        for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(a0[(m * KPX + s) * 64 + lane], b0[s], acc[m]);
    }
    for (uint32_t layer = 0;; layer++) {
#pragma unroll
        for (uint32_t m = 0; m < MB; m++)
#pragma unroll
            for (uint32_t r = 0; r < 16; r += 2) {
                const half2p v = relu2(cvt2(acc[m][r], acc[m][r + 1]));
                bf[2 * m + (r >> 3)][r & 7] = v[0];
                bf[2 * m + (r >> 3)][(r & 7) + 1] = v[1];
            }
        if (layer == NH) break;
        const half8* a = a0 + (size_t)(nf0 + layer * MB * KS) * 64;
#pragma unroll
        for (uint32_t m = 0; m < MB; m++) {
            acc[m] = zero16();
#pragma unroll
            for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(a[(m * KS + s) * 64 + lane], bf[s], acc[m]);
        }
    }
    float16v o = zero16();
    const half8* a = a0 + (size_t)(nf0 + NH * MB * KS) * 64;
#pragma unroll
    for (uint32_t s = 0; s < KS; s++) o = mfma(a[s * 64 + lane], bf[s], o);
    return o;
}
// SEAL = the two-encoder network Seal-3D trains (nerf/network.py:99-128): the colour-net input row is 64 wide,
// [half(SH_4(d)) | h1..h15 | encoder_color(x) (32) | 0] — k_ngp_mid2_forward's row, built in the wave's tile; the second encoder's
// features are requested with the next tile's inputs.
template <bool SEAL>
// (occupancy, measured with tools/bench_pair.py at 1.6e6 rows: the NGP variant needs 114 registers — two 8-wave workgroups per
//  CU; the Seal variant 148: forced down to 128 it spills 14 and takes 104 - 118 us against 98 with ONE workgroup per CU)
#ifndef S3D_PAIR_WPE
#define S3D_PAIR_WPE 0
#endif
#if S3D_PAIR_WPE
#define S3D_PAIR_OCC __attribute__((amdgpu_waves_per_eu(S3D_PAIR_WPE, S3D_PAIR_WPE)))
#else
#define S3D_PAIR_OCC
#endif
__global__ void __launch_bounds__(kPairWaves * 64) S3D_PAIR_OCC k_ffmlp_ngp_pair(const _Float16* __restrict__ X, const PairNets nets, uint32_t B,
                                                                   uint32_t in_layout, const int32_t* __restrict__ n_valid,
                                                                   float* __restrict__ rgb_head, const MidFwd mid) {
    constexpr uint32_t W = 64, MB = 2, KS = 4, KP = 2, IN = 32;
    constexpr uint32_t KPC = SEAL ? 4 : 2, INC = 16 * KPC;      // colour network: layer-0 k-steps, input width
    constexpr uint32_t kRow = SEAL ? 72 : kMidRow;              // halfs per tile row (16-byte aligned)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8* frags = reinterpret_cast<half8*>(smem_raw);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t NHs = nets.nl_s - 1, NHc = nets.nl_c - 1;
    constexpr uint32_t nf0 = MB * KP, nf0c = MB * KPC;
    const uint32_t tot_s = nf0 + NHs * MB * KS + KS, tot_c = nf0c + NHc * MB * KS + KS;
    _Float16* Tm = reinterpret_cast<_Float16*>(frags + (size_t)(tot_s + tot_c) * 64) + (size_t)wave * 32 * kRow;

    // stage both networks' weights as A fragments (k_ffmlp_forward's directory, the colour network behind the density network)
#pragma unroll 1
    for (uint32_t f0 = wave; f0 < tot_s + tot_c; f0 += kPairWaves) {
        const bool second = f0 >= tot_s;
        const uint32_t f = second ? f0 - tot_s : f0, NH = second ? NHc : NHs;
        const uint32_t in_w = second ? INC : IN, ks0 = second ? KPC : KP, n0 = MB * ks0;
        const _Float16* Wt = second ? nets.Wc : nets.Ws;
        const _Float16* w_hid = Wt + (size_t)W * in_w;
        const _Float16* w_last = w_hid + (size_t)NH * W * W;
        const uint32_t nfh = NH * MB * KS;
        const _Float16* r;
        bool live = true;
        if (f < n0) {
            const uint32_t mblk = f / ks0, s = f % ks0;
            r = Wt + (size_t)(mblk * 32 + n) * in_w + 16 * s;
        } else if (f < n0 + nfh) {
            const uint32_t g = f - n0, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            r = w_hid + (size_t)k * W * W + (size_t)(mblk * 32 + n) * W + 16 * s;
        } else {
            const uint32_t s = f - n0 - nfh;
            live = n < 16;
            r = w_last + (size_t)(live ? n : 0) * W + 16 * s;
        }
        frags[f0 * 64 + lane] = live ? load_bfrag_rowmajor(r, 0, h) : zero_half8();
    }
    __syncthreads();

    const uint32_t ntiles = valid_rows(B, n_valid) / 32;
    const uint32_t tstride = gridDim.x * kPairWaves;
    constexpr uint32_t NE = SEAL ? 8 : 1;  // level pairs of the second encoder per lane (lane half h: levels 8 h .. 8 h + 7)
    half8 bin[KP];
    float dir[3] = {0.0f, 0.0f, 0.0f};
    uint32_t enc[NE];
    auto request = [&](uint32_t t, half8 (&dst)[KP], float (&d)[3], uint32_t (&e)[NE]) {
        const size_t r = (size_t)t * 32 + n;
#pragma unroll
        for (uint32_t s = 0; s < KP; s++) dst[s] = load_bfrag_input(X, in_layout, B, IN, r, s, h);
        d[0] = mid.dirs[r * 3]; d[1] = mid.dirs[r * 3 + 1]; d[2] = mid.dirs[r * 3 + 2];
        if constexpr (SEAL) {
#pragma unroll
            for (uint32_t l = 0; l < NE; l++)
                e[l] = *reinterpret_cast<const uint32_t*>(nets.enc_c + ((size_t)(8 * h + l) * B + r) * 2);
        }
    };
    uint32_t tile = blockIdx.x * kPairWaves + wave;
#pragma unroll
    for (uint32_t l = 0; l < NE; l++) enc[l] = 0u;
    if (tile < ntiles) request(tile, bin, dir, enc);
    for (; tile < ntiles; tile += tstride) {
        const size_t row = (size_t)tile * 32 + n;
        half8 bnx[KP];
        float dnx[3] = {0.0f, 0.0f, 0.0f};
        uint32_t enx[NE];
#pragma unroll
        for (uint32_t l = 0; l < NE; l++) enx[l] = 0u;
        if (tile + tstride < ntiles) request(tile + tstride, bnx, dnx, enx);
        const float16v o = pair_network<KP>(frags, NHs, bin, lane);
#pragma unroll
        for (uint32_t s = 0; s < KP; s++) bin[s] = bnx[s];
        const float dx = dir[0], dy = dir[1], dz = dir[2];
        dir[0] = dnx[0]; dir[1] = dnx[1]; dir[2] = dnx[2];
        // density head (k_ffmlp_forward's): sigma leaves, the colour-net input row stays in the wave's tile
#pragma unroll
        for (uint32_t q = 0; q < 2; q++)
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                const uint32_t f = 8 * q + 4 * h + e;  // output feature of o[4 q + e]
                const _Float16 v = (_Float16)o[4 * q + e];
                if (f == 0) {
                    mid.sigma[row] = expf((float)v);  // activation.py:8-11
                    if (mid.h0) mid.h0[row] = v;
                } else {
                    Tm[n * kRow + 15 + f] = v;
                }
            }
        if constexpr (SEAL) {
            // columns 31 .. 62: the second encoder's 32 features (level l: columns 31 + 2 l, 32 + 2 l), 63: the zero pad
#pragma unroll
            for (uint32_t l = 0; l < NE; l++) {
                const uint32_t c0 = 31 + 2 * (8 * h + l);
                _Float16 lo, hi;
                const uint16_t lo16 = (uint16_t)(enc[l] & 0xFFFFu), hi16 = (uint16_t)(enc[l] >> 16);
                __builtin_memcpy(&lo, &lo16, 2);
                __builtin_memcpy(&hi, &hi16, 2);
                Tm[n * kRow + c0] = lo;
                Tm[n * kRow + c0 + 1] = hi;
            }
            if (h == 1) Tm[n * kRow + 63] = (_Float16)0.0f;
#pragma unroll
            for (uint32_t l = 0; l < NE; l++) enc[l] = enx[l];
        } else {
            if (h == 1) Tm[n * kRow + 31] = (_Float16)0.0f;
        }
        {
            float sh[16], j0[1], j1[1], j2[1];
            sh_eval<4, false>(dx, dy, dz, mid.K, sh, j0, j1, j2);
            half8 vlo, vhi;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) { vlo[i] = (_Float16)sh[i]; vhi[i] = (_Float16)sh[8 + i]; }
            uint32_t wl[4], wh[4], wv[4];
            __builtin_memcpy(wl, &vlo, 16);
            __builtin_memcpy(wh, &vhi, 16);
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) wv[i] = h ? wh[i] : wl[i];
            half8 v;
            __builtin_memcpy(&v, wv, 16);
            *reinterpret_cast<half8*>(Tm + n * kRow + 8 * h) = v;
        }
        __builtin_amdgcn_wave_barrier();
        half8 bc[KPC];
#pragma unroll
        for (uint32_t s = 0; s < KPC; s++) bc[s] = load_bfrag_rowmajor(Tm + n * kRow, s, h);
        if (mid.cin) {  // (optional: the colour-net input rows — a training forward keeps them for the backward)
            constexpr uint32_t PC = INC / 16;  // 16-byte pieces per lane half
#pragma unroll
            for (uint32_t c = 0; c < PC; c++)
                *reinterpret_cast<half8*>(mid.cin + row * INC + 8 * (PC * h + c)) =
                    *reinterpret_cast<const half8*>(Tm + n * kRow + 8 * (PC * h + c));
        }
        __builtin_amdgcn_wave_barrier();  // (the tile is free for the next row block once every lane has its fragments)
        const float16v oc = pair_network<KPC>(frags + (size_t)tot_s * 64, NHc, bc, lane);
        // colour head (network_ff.py:103 `torch.sigmoid(h)` on the fp16 output): outputs 0..2 as fp32 sigmoid values, rounded
        // where the fp16 op sequence rounds
        if (h == 0) {
#pragma unroll
            for (uint32_t c = 0; c < 3; c++) {
                const _Float16 v = (_Float16)oc[c];
                rgb_head[row * 3 + c] = (float)(_Float16)(1.0f / (1.0f + expf(-(float)v)));
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dgrad
// LDS directory: last^T: MB frags (one k-step, K = 16 outputs) | hidden^T k: MB*KS each | first^T: IMB*KS
template <int W, int ACT>
__global__ void __launch_bounds__(256) k_ffmlp_dgrad(const _Float16* __restrict__ grad, const _Float16* __restrict__ Wt,
                                                     const _Float16* __restrict__ fwd, uint32_t B, uint32_t in_dim,
                                                     uint32_t out_dim, uint32_t n_layers, uint32_t act,
                                                     _Float16* __restrict__ bwd, _Float16* __restrict__ grad_inputs,
                                                     const int32_t* __restrict__ n_valid) {
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

    const uint32_t ntiles = valid_rows(B, n_valid) / 32;
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
                        v[e] = (_Float16)act_bwd_t<ACT>(act, g, (float)fv[e]);
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
constexpr uint32_t kWgradPad = 64;  // partial weight-gradient matrices are stored [64][64] fp32 regardless of W
struct WgradLayer {
    const _Float16* G;  // gradient w.r.t. the layer's pre-activation output
    const _Float16* X;  // the layer's input
    uint32_t g_native, x_native;  // fragment-ordered (W wide) or row-major
    uint32_t Fo, Fi;              // real feature counts
    uint32_t w_off;               // offset of this layer's matrix in the flat weight vector
    // a column block of a wider layer (W = 128 with 128 < input_dim <= 160: the first matrix's gradient as two plan entries):
    uint32_t x_ld, x_col0;        // row-major X: row stride and first column of the block (x_ld = 0: the block is all of X)
    uint32_t w_ld;                // row stride of the matrix in the flat weight vector (0: Fi); w_off points at the block's first column
};
struct WgradPlan {
    WgradLayer layer[kMaxMlpLayers];
    uint32_t n;
};



// partial weight-gradient matrices of the stored-activation path: [PAD][PAD] fp32 per (layer, workgroup)
template <int W> struct WgradGeom { static constexpr uint32_t PAD = W > 64 ? 128u : kWgradPad; };
inline uint32_t wgrad_pad(uint32_t W) { return W > 64 ? 128u : kWgradPad; }
inline uint32_t wgrad_blocks(uint32_t W);

// fragment of a row-major [32 points][features] LDS tile with `stride` halfs per row (see transpose_load below: the same
// ds_read_b64_tr_b16 pair, any row stride that keeps the 8-byte segments aligned)
__device__ __forceinline__ half8 tile_fragment(const _Float16* __restrict__ T, uint32_t stride, uint32_t blk, uint32_t s, uint32_t fl,
                                               uint32_t h, uint32_t nfeat) {
    typedef __fp16 f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) f16x4 lds_f16x4;
    const uint32_t i = fl & 15, g1 = fl >> 4;  // lane = 32 h + fl: 16-lane group 2 h + g1
    const _Float16* p = T + (16 * s + 8 * h + i / 4) * stride + 32 * blk + 16 * g1 + 4 * (i % 4);
    const f16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)p);
    const f16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)(p + 4 * stride));
    half8 v;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) { v[j] = (_Float16)a[j]; v[4 + j] = (_Float16)b[j]; }
    if (blk * 32 + fl >= nfeat) {
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) v[j] = (_Float16)0.0f;
    }
    return v;
}

// Both operands of dW = G^T X have the batch as the MFMA K dimension: a 32-row tile of each goes to LDS as a row-major image (row =
// point, W + 8 halfs per row) — a fragment-ordered buffer is re-ordered by the 16-byte chunk on the way in — and every lane reads
// its 8 consecutive points of one feature with two transposing 8-byte loads.  The four waves of a workgroup SHARE the tile pair and
// split the 32 x 32 output blocks between them (block j of the layer goes to wave j % 4): 64 accumulator registers per wave
// instead of 256, so four workgroups fit a CU and their waves cover each other's loads; the next pair is requested into registers
// before the current one is multiplied and parked in the other half of a double buffer (one barrier per tile), and every wave
// writes its own blocks of the workgroup's partial plane (no cross-wave sum).  What is left is traffic: 16 KiB per 32-row tile and
// layer for 8 MFMAs per wave (263 MB per launch for TensoRF's network, 80 - 110 us); a second tile in flight per workgroup changed
// nothing.
// [First version: one tile per wave, linear copies, sixteen 2-byte LDS reads per fragment, 256 accumulator registers: 129.7 us for
//  TensoRF's 160 -> 128 -> 128 -> 3 network at 1.05e5 rows; with transposing loads alone 131 -> 95-130 us, still one wave per SIMD
//  waiting for every tile.]
template <int W>
__global__ void __launch_bounds__(256) k_ffmlp_wgrad(WgradPlan plan, uint32_t B, float* __restrict__ partial,
                                                     const int32_t* __restrict__ n_valid) {
    constexpr uint32_t MAXB = (W + 31) / 32;  // 32-blocks per side
    constexpr uint32_t PAD = WgradGeom<W>::PAD;
    constexpr uint32_t TS = W + 8;            // halfs per tile row (8 pad: the transposing loads' 16-lane groups spread over the banks)
    constexpr uint32_t kTile = 32 * TS;       // halfs per tile image
    constexpr uint32_t NBLK = (MAXB * MAXB + 3) / 4;  // output blocks per wave
    constexpr uint32_t NCH = (32 * W / 8 + 255) / 256;  // 16-byte chunks of a W-wide tile per thread
    __shared__ __attribute__((aligned(16))) _Float16 tiles[2][2][kTile];  // [buffer][G | X]

    const WgradLayer L = plan.layer[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t fl = lane & 31, h = lane >> 5;
    const uint32_t gF = L.g_native ? W : L.Fo, xF = L.x_native ? W : L.Fi;  // storage widths
    const uint32_t MBo = (L.Fo + 31) / 32, NBi = (L.Fi + 31) / 32;
    const uint32_t x_ld = L.x_ld ? L.x_ld : xF;

    float16v acc[NBLK];
#pragma unroll
    for (uint32_t t = 0; t < NBLK; t++) acc[t] = zero16();

    // 16-byte chunk i (= thread + 256 k) of a tile's global image, and its place in the row-major LDS image
    auto request = [&](uint4 (&r)[NCH], const _Float16* src, uint32_t native, uint32_t F, uint32_t ld, uint32_t col0, uint32_t tile) {
        const uint32_t per_row = F / 8;
#pragma unroll
        for (uint32_t k = 0; k < NCH; k++) {
            const uint32_t i = threadIdx.x + 256 * k;
            if (i >= 32 * per_row) continue;
            if (native) {  // [F / 8 feature groups][32 points][8 features]
                r[k] = reinterpret_cast<const uint4*>(src + (size_t)tile * 32 * F)[i];
            } else {       // [32 points][ld features], F of them from column col0
                const uint32_t b = i / per_row, c = i - b * per_row;
                r[k] = *reinterpret_cast<const uint4*>(src + ((size_t)tile * 32 + b) * ld + col0 + 8 * c);
            }
        }
    };
    auto park = [&](_Float16* t, const uint4 (&r)[NCH], uint32_t native, uint32_t F) {
        const uint32_t per_row = F / 8;
#pragma unroll
        for (uint32_t k = 0; k < NCH; k++) {
            const uint32_t i = threadIdx.x + 256 * k;
            if (i >= 32 * per_row) continue;
            const uint32_t at = native ? (i & 31u) * (TS / 8) + (i >> 5) : (i / per_row) * (TS / 8) + i % per_row;
            reinterpret_cast<uint4*>(t)[at] = r[k];
        }
    };
    const uint32_t ntiles = valid_rows(B, n_valid) / 32;
    uint4 rg[NCH], rx[NCH];
    if (blockIdx.x < ntiles) {
        request(rg, L.G, L.g_native, gF, gF, 0u, blockIdx.x);
        request(rx, L.X, L.x_native, xF, x_ld, L.x_col0, blockIdx.x);
    }
    uint32_t buf = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1u) {
        _Float16* tg = tiles[buf][0];
        _Float16* tx = tiles[buf][1];
        park(tg, rg, L.g_native, gF);
        park(tx, rx, L.x_native, xF);
        __syncthreads();  // (also: every wave is done with the other buffer, which the next round parks into)
        if (tile + gridDim.x < ntiles) {
            request(rg, L.G, L.g_native, gF, gF, 0u, tile + gridDim.x);
            request(rx, L.X, L.x_native, xF, x_ld, L.x_col0, tile + gridDim.x);
        }
#pragma unroll
        for (uint32_t s = 0; s < 2; s++) {
#pragma unroll
            for (uint32_t t = 0; t < NBLK; t++) {
                const uint32_t j = wave + 4 * t;  // block j = (mo, ni) of this layer
                if (j < MBo * NBi) {
                    const uint32_t mo = j / NBi, ni = j - mo * NBi;
                    acc[t] = mfma(tile_fragment(tg, TS, mo, s, fl, h, L.Fo), tile_fragment(tx, TS, ni, s, fl, h, L.Fi), acc[t]);
                }
            }
        }
    }
    // every wave's blocks straight into the workgroup's partial plane; the blocks no wave owns (layers narrower than W) are zero
    float* dst = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * PAD * PAD;
#pragma unroll
    for (uint32_t t = 0; t < NBLK; t++) {
        const uint32_t j = wave + 4 * t;
        if (j >= MAXB * MAXB) continue;
        const bool mine = j < MBo * NBi;
        // (unused blocks: position j of the MAXB x MAXB grid that no (mo, ni) of this layer maps to)
        const uint32_t mo = mine ? j / NBi : 0, ni = mine ? j - mo * NBi : 0;
        if (!mine) continue;
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) {
            const uint32_t o = mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, i = ni * 32 + fl;
            dst[o * PAD + i] = acc[t][r];
        }
    }
}

// Eight lanes per matrix element: each sums every 8th workgroup partial with independent accumulators (a single chain over
// 256 partials was latency-bound: ~20 us), then a fixed xor-shuffle tree combines the eight.  Deterministic.
constexpr uint32_t kReduceSplit = 8;
// The same reduction for the weight gradients of SEVERAL networks in one launch (s3d_ffmlp_wgrad_reduce_pair: the backward calls
// of the colour and the density network leave their partial sums in their workspaces, accumulate_grad_weights = 2, and this
// launch finishes both — one launch of the step's 18 gone).  Same per-element order of additions as k_ffmlp_wgrad_reduce.
struct ReduceJob {
    const float* partial;   // [nblk][64][64] planes of this layer
    _Float16* gw;           // the network's grad_weights
    float* found_inf;
    uint32_t Fo, Fi, w_off, nblk, accumulate;
};
struct ReduceJobs {
    ReduceJob job[2 * kMaxMlpLayers];
    uint32_t n;
};
__global__ void k_ffmlp_wgrad_reduce_jobs(ReduceJobs jobs);

__global__ void k_ffmlp_wgrad_reduce(WgradPlan plan, uint32_t nblk, const float* __restrict__ partial,
                                     _Float16* __restrict__ grad_weights, uint32_t accumulate, float* __restrict__ found_inf,
                                     uint32_t pad) {
    const WgradLayer L = plan.layer[blockIdx.y];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = t / kReduceSplit, part = t % kReduceSplit;
    const bool live = e < L.Fo * L.Fi;
    const uint32_t o = live ? e / L.Fi : 0, i = live ? e - o * L.Fi : 0;
    const size_t kPlane = (size_t)pad * pad;
    const float* p = partial + (size_t)blockIdx.y * nblk * kPlane + o * pad + i;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (live) {
        uint32_t b = part;
        for (; b + 3 * kReduceSplit < nblk; b += 4 * kReduceSplit) {
            s0 += p[(size_t)(b + 0 * kReduceSplit) * kPlane];
            s1 += p[(size_t)(b + 1 * kReduceSplit) * kPlane];
            s2 += p[(size_t)(b + 2 * kReduceSplit) * kPlane];
            s3 += p[(size_t)(b + 3 * kReduceSplit) * kPlane];
        }
        for (; b < nblk; b += kReduceSplit) s0 += p[(size_t)b * kPlane];
    }
    float v = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int d = 1; d < (int)kReduceSplit; d <<= 1) v += __shfl_xor(v, d, 64);
    if (live && part == 0) {
        const size_t at = (size_t)L.w_off + (L.w_ld ? (size_t)o * L.w_ld + i : (size_t)e);
        if (accumulate) v += (float)grad_weights[at];  // add to the caller's running gradient (fp16 hand-over buffer)
        const _Float16 h = (_Float16)v;
        grad_weights[at] = h;
        // GradScaler's non-finite check made where the gradient is written (benign race: everyone writes 1)
        if (found_inf && !(fabsf((float)h) <= 65504.0f)) *found_inf = 1.0f;
    }
}

__global__ void k_ffmlp_wgrad_reduce_jobs(ReduceJobs jobs) {
    const ReduceJob L = jobs.job[blockIdx.y];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = t / kReduceSplit, part = t % kReduceSplit;
    const bool live = e < L.Fo * L.Fi;
    const uint32_t o = live ? e / L.Fi : 0, i = live ? e - o * L.Fi : 0;
    const float* p = L.partial + o * kWgradPad + i;
    constexpr size_t kPlane = (size_t)kWgradPad * kWgradPad;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (live) {
        uint32_t b = part;
        for (; b + 3 * kReduceSplit < L.nblk; b += 4 * kReduceSplit) {
            s0 += p[(size_t)(b + 0 * kReduceSplit) * kPlane];
            s1 += p[(size_t)(b + 1 * kReduceSplit) * kPlane];
            s2 += p[(size_t)(b + 2 * kReduceSplit) * kPlane];
            s3 += p[(size_t)(b + 3 * kReduceSplit) * kPlane];
        }
        for (; b < L.nblk; b += kReduceSplit) s0 += p[(size_t)b * kPlane];
    }
    float v = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int d = 1; d < (int)kReduceSplit; d <<= 1) v += __shfl_xor(v, d, 64);
    if (live && part == 0) {
        if (L.accumulate) v += (float)L.gw[L.w_off + e];
        const _Float16 h = (_Float16)v;
        L.gw[L.w_off + e] = h;
        if (L.found_inf && !(fabsf((float)h) <= 65504.0f)) *L.found_inf = 1.0f;
    }
}

// ------------------------------------------------------------------------------------ backward: fused
// One kernel for the whole backward pass, nothing but the network input, the output gradient and the weights read
// from HBM: per 32-point tile a wave (1) RE-COMPUTES the forward activations from the input (36 kFLOP per point is
// nothing next to the 256-384 B per point the stored forward_buffer costs to write and read back), (2) walks the
// transposed network for the data gradient exactly like k_ffmlp_dgrad, and (3) accumulates every layer's weight
// gradient dW[o][i] += sum_p G[p][o] X[p][i] on the spot.  For (3) the batch is the MFMA K dimension, so both
// operands are transposed through a per-wave LDS tile: lanes scatter their (point, feature) values into
// T[feature][point] (rows padded to 40 halfs: the two lane halves hit disjoint banks) and read their fragment back
// as ONE 16-byte row segment (8 consecutive points of one feature).  Weight-gradient accumulators stay in registers
// (MFMA accumulation VGPRs) for all tiles of the wave; at the end the four waves are summed through LDS in a fixed
// order and one fp32 partial per workgroup goes to the same reduce kernel as the two-kernel path.  Deterministic.
// HBM traffic per point: in*2 + 32 B read (+ in*2 B grad_inputs) instead of ~1.2 KB.
constexpr uint32_t kTRow = 72;   // halfs per row of a per-wave tile: 64 features + 8 pad (144 B: 8-byte aligned segments)
constexpr uint32_t kTRows = 32;  // rows = the 32 points of the tile
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// The tile is stored the way the lanes hold it — row = point, 64 features side by side: a lane's fragment is two runs of four
// consecutive features per k-step (k index (j&3) + 8(j>>2) + 4h), i.e. two 8-byte stores — and read back TRANSPOSED by
// gfx950's ds_read_b64_tr_b16: the 16 lanes of a group hand in the sixteen 8-byte segments of a [4 points][16 features] block
// (lane i: point i/4, features 4(i%4)..+3) and lane i receives feature i of the four points (tools/ubench/tr16.hip prints the
// mapping).  Two such reads are a lane's 8 consecutive points of one feature: the A / B operand of the weight-gradient MFMA.
// [Before: transposed tile T[feature][point], 8 two-byte stores per k-step and lane, one 16-byte load per fragment: 216
//  ds_write_b16 per 32-point tile, a quarter of the kernel's LDS instructions and their address arithmetic.]
template <int NS>
__device__ __forceinline__ void transpose_store(_Float16* __restrict__ T, const half8 (&bf)[NS], uint32_t ksteps,
                                                uint32_t n, uint32_t h) {
#pragma unroll
    for (uint32_t s = 0; s < (uint32_t)NS; s++)
        if (s < ksteps) {
            half4 lo, hi;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) { lo[j] = bf[s][j]; hi[j] = bf[s][4 + j]; }
            _Float16* row = T + n * kTRow + 16 * s + 4 * h;
            *reinterpret_cast<half4*>(row) = lo;
            *reinterpret_cast<half4*>(row + 8) = hi;
        }
}
// fragment (A or B operand of the weight-gradient MFMA) for feature block `blk`, k-step (16 points) s
__device__ __forceinline__ half8 transpose_load(const _Float16* __restrict__ T, uint32_t blk, uint32_t s, uint32_t fl,
                                                uint32_t h, uint32_t nfeat) {
    const uint32_t i = fl & 15, g1 = fl >> 4;  // lane = 32 h + fl: 16-lane group 2 h + g1
    const _Float16* p = T + (16 * s + 8 * h + i / 4) * kTRow + 32 * blk + 16 * g1 + 4 * (i % 4);
    typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)p);
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(p + 4 * kTRow));
    half8 v;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) { v[j] = (_Float16)a[j]; v[4 + j] = (_Float16)b[j]; }
    if (blk * 32 + fl >= nfeat) {
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) v[j] = (_Float16)0.0f;
    }
    return v;
}

// Sum one weight-gradient matrix over the four waves of the workgroup and write the workgroup's fp32 partial.
// Each wave stores its accumulator blocks to its OWN [64][64] LDS plane (independent stores, no read-modify-write
// chains), then all 256 threads add the four planes in a fixed order.
template <uint32_t MBLK, uint32_t NBLK, uint32_t THREADS = 256, uint32_t NPL = 4, typename Get>
__device__ __forceinline__ void flush_matrix(float* __restrict__ red, float* __restrict__ partial, uint32_t matrix, uint32_t wave,
                                             uint32_t n, uint32_t h, Get&& get) {
    // (THREADS > 256: the waves behind the fourth hold no accumulators — `wave` >= 4 — and only help with the sum)
    constexpr uint32_t kPlane = kWgradPad * kWgradPad;
    __syncthreads();
    float* mine = red + (size_t)wave * kPlane;
    if (wave < NPL) {
#pragma unroll
        for (uint32_t mo = 0; mo < MBLK; mo++)
#pragma unroll
            for (uint32_t ni = 0; ni < NBLK; ni++) {
                const float16v v = get(mo, ni);
#pragma unroll
                for (uint32_t r = 0; r < 16; r++) mine[(mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kWgradPad + ni * 32 + n] = v[r];
            }
    }
    __syncthreads();
    float* dst = partial + ((size_t)matrix * gridDim.x + blockIdx.x) * kPlane;
    for (uint32_t i = threadIdx.x; i < kPlane; i += THREADS) {
        const uint32_t o = i / kWgradPad, c = i % kWgradPad;
        float v = 0.0f;
        if (o < MBLK * 32 && c < NBLK * 32) {
            v = red[i];
#pragma unroll
            for (uint32_t p = 1; p < NPL; p++) v += red[p * kPlane + i];  // (fixed order: ((p0 + p1) + p2) + ...)
        }
        dst[i] = v;
    }
}

// KS0T: compile-time in_dim / 16 (0 = run-time).  With a run-time count every layer-0 MFMA step sits in its own branch and
// the compiler shuttles the accumulators between VGPRs and AGPRs around each one (~1000 v_accvgpr moves per tile).
template <int W, int NH, int IMB, int ACT, int KS0T>
__global__ void __launch_bounds__(256) k_ffmlp_backward_fused(const _Float16* __restrict__ grad, const _Float16* __restrict__ X,
                                                             const _Float16* __restrict__ Wt, uint32_t B, uint32_t in_dim,
                                                             uint32_t out_dim, uint32_t act, _Float16* __restrict__ grad_inputs,
                                                             float* __restrict__ partial, uint32_t in_layout,
                                                             const int32_t* __restrict__ n_valid,
                                                             const float* __restrict__ d_rgb, const float* __restrict__ rgb_head,
                                                             const MidBwd midb) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t KS0 = KS0T ? (uint32_t)KS0T : in_dim / 16;
    // LDS: forward A frags [layer0: MB*KS0 | hidden: NH*MB*KS] | transposed A frags [last^T: MB | hidden^T: NH*MB*KS |
    //      first^T: IMB*KS (only with grad_inputs)] | per-wave transpose tiles TG, TX | (aliased at the end) red[64*64]
    const uint32_t nf_f0 = MB * KS0, nf_fh = NH * MB * KS;
    const uint32_t nf_bl = MB, nf_bh = NH * MB * KS, nf_b0 = grad_inputs ? IMB * KS : 0;
    const uint32_t nfrag = nf_f0 + nf_fh + nf_bl + nf_bh + nf_b0;
    half8* frags = reinterpret_cast<half8*>(smem_raw);
    half8* ff0 = frags;
    half8* ffh = ff0 + (size_t)nf_f0 * 64;
    half8* fbl = ffh + (size_t)nf_fh * 64;
    half8* fbh = fbl + (size_t)nf_bl * 64;
    half8* fb0 = fbh + (size_t)nf_bh * 64;
    _Float16* tiles = reinterpret_cast<_Float16*>(frags + (size_t)nfrag * 64);
    _Float16* TG = tiles + (size_t)wave * 2 * kTRows * kTRow;
    _Float16* TX = TG + (size_t)kTRows * kTRow;
    const _Float16* w_hid = Wt + (size_t)W * in_dim;
    const _Float16* w_last = w_hid + (size_t)NH * W * W;

    for (uint32_t f = wave; f < nfrag; f += 4) {
        half8 v;
        if (f < nf_f0) {  // forward, layer 0: A[row = hidden feature][k = input]
            const uint32_t mblk = f / KS0, s = f % KS0;
            const _Float16* r = Wt + (size_t)(mblk * 32 + n) * in_dim + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = r[kperm(h, j)];
        } else if (f < nf_f0 + nf_fh) {  // forward, hidden k
            const uint32_t g = f - nf_f0, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            const _Float16* r = w_hid + (size_t)k * W * W + (size_t)(mblk * 32 + n) * W + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = r[kperm(h, j)];
        } else if (f < nf_f0 + nf_fh + nf_bl) {  // last^T: A[i = hidden feature][k = output] = W_last[k][i]
            const uint32_t i = (f - nf_f0 - nf_fh) * 32 + n;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t k = kperm(h, j);
                v[j] = (k < out_dim) ? w_last[(size_t)k * W + i] : (_Float16)0.0f;
            }
        } else if (f < nf_f0 + nf_fh + nf_bl + nf_bh) {  // hidden^T: A[i = in feature][k = out feature] = W_k[k][i]
            const uint32_t g = f - nf_f0 - nf_fh - nf_bl, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            const _Float16* wk = w_hid + (size_t)k * W * W;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = wk[(size_t)(16 * s + kperm(h, j)) * W + mblk * 32 + n];
        } else {  // first^T: A[i = network input][k = first hidden feature] = W_0[k][i]
            const uint32_t g = f - nf_f0 - nf_fh - nf_bl - nf_bh, mblk = g / KS, s = g % KS;
            const uint32_t i = mblk * 32 + n;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
                v[j] = (i < in_dim) ? Wt[(size_t)(16 * s + kperm(h, j)) * in_dim + i] : (_Float16)0.0f;
        }
        frags[f * 64 + lane] = v;
    }
    __syncthreads();

    // weight-gradient accumulators: layer 0 [W x in], hidden [W x W] x NH, last [16 (one block) x W]
    float16v dw0[MB][IMB], dwh[NH > 0 ? NH : 1][MB][MB], dwl[MB];
#pragma unroll
    for (uint32_t a = 0; a < MB; a++) {
#pragma unroll
        for (uint32_t b = 0; b < (uint32_t)IMB; b++) dw0[a][b] = zero16();
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NH; k++)
#pragma unroll
            for (uint32_t b = 0; b < MB; b++) dwh[k][a][b] = zero16();
        dwl[a] = zero16();
    }

    const uint32_t ntiles = valid_rows(B, n_valid) / 32;
    for (uint32_t tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const size_t row = (size_t)tile * 32 + n;
        // ---- inputs of the tile: network input (B fragments) and output gradient (one k-step)
        half8 xf[4];  // in_dim <= 64
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            if (s < KS0) xf[s] = load_bfrag_input(X, in_layout, B, in_dim, row, s, h);
        half8 gf;
        if (d_rgb) {  // colour head: d(out_c) = d(rgb_c) * y (1 - y) formed here from the fp32 gradient of the compositing
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) gf[j] = (_Float16)0.0f;
            if (h == 0) {
#pragma unroll
                for (uint32_t c = 0; c < 3; c++) {  // (kperm(0, c) == c for c < 4)
                    const float y = rgb_head[row * 3 + c];
                    gf[c] = (_Float16)((float)(_Float16)d_rgb[row * 3 + c] * (y * (1.0f - y)));
                }
            }
        } else if (midb.g_cin) {
            gf = mid_grad_fragment(midb, row, h);
        } else {
            gf = load_bfrag_rowmajor(grad + row * 16, 0, h);
        }

        // ---- forward re-computation (same operations and roundings as k_ffmlp_forward)
        half8 a[NH + 1][KS];
        {
            float16v acc[MB];
#pragma unroll
            for (uint32_t m = 0; m < MB; m++) acc[m] = zero16();
#pragma unroll
            for (uint32_t s = 0; s < 4; s++)
                if (s < KS0) {
#pragma unroll
                    for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(ff0[(m * KS0 + s) * 64 + lane], xf[s], acc[m]);
                }
#pragma unroll
            for (uint32_t layer = 0; layer <= (uint32_t)NH; layer++) {
#pragma unroll
                for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                    for (uint32_t r = 0; r < 16; r += 2) {
                        if constexpr (ACT == ACT_RELU) {
                            const half2p v = relu2(cvt2(acc[m][r], acc[m][r + 1]));
                            a[layer][2 * m + (r >> 3)][r & 7] = v[0];
                            a[layer][2 * m + (r >> 3)][(r & 7) + 1] = v[1];
                        } else {
#pragma unroll
                            for (uint32_t q = r; q < r + 2; q++) {
                                const float pre = (float)(_Float16)acc[m][q];
                                a[layer][2 * m + (q >> 3)][q & 7] = (_Float16)act_fwd_t<ACT>(act, pre);
                            }
                        }
                    }
                if (layer < (uint32_t)NH) {
                    const half8* wf = ffh + (size_t)(layer * MB * KS) * 64;
#pragma unroll
                    for (uint32_t m = 0; m < MB; m++) {
                        acc[m] = zero16();
#pragma unroll
                        for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(wf[(m * KS + s) * 64 + lane], a[layer][s], acc[m]);
                    }
                }
            }
        }

        // ---- last layer: dW_last += g^T a_NH ; dA_NH = W_last^T g
        {
            const half8 gtmp[1] = {gf};
            transpose_store<1>(TG, gtmp, 1, n, h);
            transpose_store<(int)KS>(TX, a[NH], KS, n, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t s = 0; s < 2; s++) {
                const half8 af = transpose_load(TG, 0, s, n, h, 16);
#pragma unroll
                for (uint32_t ni = 0; ni < MB; ni++) dwl[ni] = mfma(af, transpose_load(TX, ni, s, n, h, W), dwl[ni]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        float16v acc[MB];
#pragma unroll
        for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(fbl[m * 64 + lane], gf, zero16());

        // ---- hidden layers, top down
        half8 G[KS];
#pragma unroll
        for (int k = NH; k >= 0; k--) {
            // gradient w.r.t. the pre-activation of hidden layer k (activation transfer with the re-computed output)
#pragma unroll
            for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                for (uint32_t r = 0; r < 16; r += 2) {
                    if constexpr (ACT == ACT_RELU) {
                        half2p av;
                        av[0] = a[k][2 * m + (r >> 3)][r & 7]; av[1] = a[k][2 * m + (r >> 3)][(r & 7) + 1];
                        const half2p v = gate2(cvt2(acc[m][r], acc[m][r + 1]), av);
                        G[2 * m + (r >> 3)][r & 7] = v[0];
                        G[2 * m + (r >> 3)][(r & 7) + 1] = v[1];
                    } else {
#pragma unroll
                        for (uint32_t q = r; q < r + 2; q++) {
                            const float g = (float)(_Float16)acc[m][q];
                            G[2 * m + (q >> 3)][q & 7] = (_Float16)act_bwd_t<ACT>(act, g, (float)a[k][2 * m + (q >> 3)][q & 7]);
                        }
                    }
                }
            transpose_store<(int)KS>(TG, G, KS, n, h);
            if (k > 0) {
                transpose_store<(int)KS>(TX, a[k - 1], KS, n, h);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (uint32_t s = 0; s < 2; s++) {
                    half8 bfr[MB];
#pragma unroll
                    for (uint32_t ni = 0; ni < MB; ni++) bfr[ni] = transpose_load(TX, ni, s, n, h, W);
#pragma unroll
                    for (uint32_t mo = 0; mo < MB; mo++) {
                        const half8 af = transpose_load(TG, mo, s, n, h, W);
#pragma unroll
                        for (uint32_t ni = 0; ni < MB; ni++) dwh[k - 1][mo][ni] = mfma(af, bfr[ni], dwh[k - 1][mo][ni]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const half8* wf = fbh + (size_t)((k - 1) * MB * KS) * 64;
#pragma unroll
                for (uint32_t m = 0; m < MB; m++) {
                    acc[m] = zero16();
#pragma unroll
                    for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(wf[(m * KS + s) * 64 + lane], G[s], acc[m]);
                }
            } else {
                transpose_store<4>(TX, xf, KS0, n, h);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (uint32_t s = 0; s < 2; s++) {
                    half8 bfr[IMB];
#pragma unroll
                    for (uint32_t ni = 0; ni < (uint32_t)IMB; ni++) bfr[ni] = transpose_load(TX, ni, s, n, h, in_dim);
#pragma unroll
                    for (uint32_t mo = 0; mo < MB; mo++) {
                        const half8 af = transpose_load(TG, mo, s, n, h, W);
#pragma unroll
                        for (uint32_t ni = 0; ni < (uint32_t)IMB; ni++) dw0[mo][ni] = mfma(af, bfr[ni], dw0[mo][ni]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (grad_inputs) {
#pragma unroll
            for (uint32_t m = 0; m < (uint32_t)IMB; m++) {
                float16v gi = zero16();
#pragma unroll
                for (uint32_t s = 0; s < KS; s++) gi = mfma(fb0[(m * KS + s) * 64 + lane], G[s], gi);
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t feat = m * 32 + 8 * q + 4 * h;
                    if (feat < in_dim) {
                        half4 v;
#pragma unroll
                        for (uint32_t e = 0; e < 4; e++) v[e] = (_Float16)gi[4 * q + e];
                        store_grad_input(grad_inputs, in_layout, B, in_dim, row, feat, v);
                    }
                }
            }
        }
    }

    // ---- sum the four waves in a fixed order through LDS, one [64][64] fp32 partial per (matrix, workgroup)
    float* red = reinterpret_cast<float*>(smem_raw);  // 4 planes x 16 KiB over the (no longer needed) fragments and tiles
    flush_matrix<MB, IMB>(red, partial, 0, wave, n, h, [&](auto mo, auto ni) { return dw0[mo][ni]; });
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)NH; k++)
        flush_matrix<MB, MB>(red, partial, 1 + k, wave, n, h, [&](auto mo, auto ni) { return dwh[k][mo][ni]; });
    flush_matrix<1, MB>(red, partial, NH + 1, wave, n, h, [&](auto mo, auto ni) { (void)mo; return dwl[ni]; });
}

template <typename F, uint32_t... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<uint32_t, I...>) {
    (f(std::integral_constant<uint32_t, I>{}), ...);
}
template <uint32_t N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<uint32_t, N>{}); }

#ifdef S3D_FFMLP_PROF  // shader-clock stamps of one pair of one workgroup of k_ffmlp_backward_duo: tools/prof_duo.py only
#ifndef S3D_FFMLP_PROF_BLOCK
#define S3D_FFMLP_PROF_BLOCK 7
#endif
__device__ unsigned long long s3d_ffmlp_prof[2][1024];
#define DUO_STAMP() do { if (prof_on && prof_k < 1024) s3d_ffmlp_prof[role][prof_k++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DUO_STAMP() do { } while (0)
#endif
// ------------------------------------------------------------------------------------ backward: fused, two roles
// The fused kernel above runs ONE 456-register wave per SIMD: nothing hides the latencies of its MFMA chains, conversions and
// LDS round trips, and it spends ~70 % of a launch in the re-computation + data-gradient chain and ~30 % in the weight-gradient
// MFMAs (measured by switching the halves off).  Here the two halves are different WAVES: workgroup = 4 pairs; the compute
// wave of a pair re-computes the activations and walks the data gradient, storing each layer's (gradient, activation) tile to
// LDS; its partner keeps the weight-gradient accumulators (192 registers) and consumes the tiles one stage behind, through a
// double-buffered slot and one workgroup barrier per stage.  Two waves of <= 256 registers per SIMD: the matrix pipe, the VALU
// and the LDS overlap across the pair.  Same per-wave tile sequence, same MFMA order per accumulator: bit-identical results.
// NC compute waves and NG weight-gradient waves per workgroup (NC a multiple of NG): weight-gradient wave g consumes the tiles
// of the compute waves g, g + NG, ... in that order, one stage behind each.  Deterministic; the sums differ from the (NC = NG)
// arrangement only in the order the tiles of a workgroup enter an accumulator.
// The stage barriers of the two-role kernel order LDS traffic only (tile slots): a __syncthreads() would also drain every wave's
// vector-memory queue — the input-gradient stores of a tile's last stage and the next tile's input loads — at each of the NS + 1
// barriers of a round.  S3D_DUO_FULL_BARRIER: the old behaviour (A/B).
__device__ __forceinline__ void duo_barrier() {
#ifdef S3D_DUO_FULL_BARRIER
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}
template <int W, int NH, int IMB, int ACT, int KS0T, int NC = 4, int NG = 4>
__global__ void __launch_bounds__((NC + NG) * 64) k_ffmlp_backward_duo(const _Float16* __restrict__ grad, const _Float16* __restrict__ X,
                                                            const _Float16* __restrict__ Wt, uint32_t B, uint32_t in_dim,
                                                            uint32_t out_dim, uint32_t act, _Float16* __restrict__ grad_inputs,
                                                            float* __restrict__ partial, uint32_t in_layout,
                                                            const int32_t* __restrict__ n_valid,
                                                            const float* __restrict__ d_rgb, const float* __restrict__ rgb_head,
                                                            const MidBwd midb) {
    constexpr uint32_t MB = W / 32, KS = W / 16, NS = NH + 2;  // NS stages per tile: last | hidden NH..1 | first
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(NC % NG == 0, "every weight-gradient wave serves the same number of compute waves");
    constexpr uint32_t NR = NC / NG;  // compute waves (tile streams) per weight-gradient wave
    const uint32_t role = wave < (uint32_t)NC ? 0u : 1u;  // 0: compute, 1: weight gradient
    const uint32_t pair = role == 0 ? wave : wave - (uint32_t)NC;  // compute wave c / weight-gradient wave g (serves c = g + j NG)
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t KS0 = KS0T ? (uint32_t)KS0T : in_dim / 16;
    const uint32_t nf_f0 = MB * KS0, nf_fh = NH * MB * KS;
    const uint32_t nf_bl = MB, nf_bh = NH * MB * KS, nf_b0 = grad_inputs ? IMB * KS : 0;
#ifdef S3D_FFMLP_PROF
    uint32_t prof_k = 0;
    const bool prof_on = NH == S3D_FFMLP_PROF && blockIdx.x == S3D_FFMLP_PROF_BLOCK && pair == 0 && lane == 0;  // -DS3D_FFMLP_PROF=<hidden matrices of the network to stamp>
#endif
    const uint32_t nfrag = nf_f0 + nf_fh + nf_bl + nf_bh + nf_b0;
    half8* frags = reinterpret_cast<half8*>(smem_raw);
    half8* ff0 = frags;
    half8* ffh = ff0 + (size_t)nf_f0 * 64;
    half8* fbl = ffh + (size_t)nf_fh * 64;
    half8* fbh = fbl + (size_t)nf_bl * 64;
    half8* fb0 = fbh + (size_t)nf_bh * 64;
    _Float16* tiles = reinterpret_cast<_Float16*>(frags + (size_t)nfrag * 64);
    constexpr uint32_t kTile = kTRows * kTRow;
    // per COMPUTE wave: [slot 0: TG | TX][slot 1: TG | TX]
    const _Float16* w_hid = Wt + (size_t)W * in_dim;
    const _Float16* w_last = w_hid + (size_t)NH * W * W;

    for (uint32_t f = wave; f < nfrag; f += NC + NG) {
        half8 v;
        if (f < nf_f0) {  // forward, layer 0: A[row = hidden feature][k = input]
            const uint32_t mblk = f / KS0, s = f % KS0;
            const _Float16* r = Wt + (size_t)(mblk * 32 + n) * in_dim + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = r[kperm(h, j)];
        } else if (f < nf_f0 + nf_fh) {  // forward, hidden k
            const uint32_t g = f - nf_f0, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            const _Float16* r = w_hid + (size_t)k * W * W + (size_t)(mblk * 32 + n) * W + 16 * s;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = r[kperm(h, j)];
        } else if (f < nf_f0 + nf_fh + nf_bl) {  // last^T: A[i = hidden feature][k = output] = W_last[k][i]
            const uint32_t i = (f - nf_f0 - nf_fh) * 32 + n;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t k = kperm(h, j);
                v[j] = (k < out_dim) ? w_last[(size_t)k * W + i] : (_Float16)0.0f;
            }
        } else if (f < nf_f0 + nf_fh + nf_bl + nf_bh) {  // hidden^T: A[i = in feature][k = out feature] = W_k[k][i]
            const uint32_t g = f - nf_f0 - nf_fh - nf_bl, k = g / (MB * KS), mblk = (g / KS) % MB, s = g % KS;
            const _Float16* wk = w_hid + (size_t)k * W * W;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = wk[(size_t)(16 * s + kperm(h, j)) * W + mblk * 32 + n];
        } else {  // first^T: A[i = network input][k = first hidden feature] = W_0[k][i]
            const uint32_t g = f - nf_f0 - nf_fh - nf_bl - nf_bh, mblk = g / KS, s = g % KS;
            const uint32_t i = mblk * 32 + n;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
                v[j] = (i < in_dim) ? Wt[(size_t)(16 * s + kperm(h, j)) * in_dim + i] : (_Float16)0.0f;
        }
        frags[f * 64 + lane] = v;
    }
    __syncthreads();

    // tiles of this pair: blockIdx.x * 4 + pair + i * gridDim.x * 4; every wave of the workgroup runs `nit` rounds of NS
    // stages plus one draining stage, whatever its own tile count (the barriers are workgroup-wide)
    const uint32_t ntiles = valid_rows(B, n_valid) / 32;
    const uint32_t first = blockIdx.x * NC, stride = gridDim.x * NC;
    const uint32_t nit = ntiles > first ? (ntiles - first - 1) / stride + 1 : 0;  // rounds of compute wave 0 (the most)
    auto tile_of = [&](uint32_t c, uint32_t i) { return first + c + i * stride; };
    auto slot_of = [&](uint32_t c, uint32_t i, uint32_t st) { return tiles + (size_t)c * 4 * kTile + (size_t)((i * NS + st) & 1u) * 2 * kTile; };

    if (role == 0) {
        // =============================================================== compute wave
        for (uint32_t i = 0; i < nit; i++) {
            const uint32_t tile = tile_of(pair, i);
            const bool live = tile < ntiles;
            const size_t row = (size_t)(live ? tile : 0) * 32 + n;
            half8 xf[4];
            half8 gf;
            half8 a[NH + 1][KS];
            float16v acc[MB];
            half8 G[KS];
            DUO_STAMP();  // 0: top of the tile
            if (live) {
#pragma unroll
                for (uint32_t s = 0; s < 4; s++)
                    if (s < KS0) xf[s] = load_bfrag_input(X, in_layout, B, in_dim, row, s, h);
                if (d_rgb) {
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) gf[j] = (_Float16)0.0f;
                    if (h == 0) {
#pragma unroll
                        for (uint32_t c = 0; c < 3; c++) {
                            const float y = rgb_head[row * 3 + c];
                            gf[c] = (_Float16)((float)(_Float16)d_rgb[row * 3 + c] * (y * (1.0f - y)));
                        }
                    }
                } else if (midb.g_cin) {
                    gf = mid_grad_fragment(midb, row, h);
                } else {
                    gf = load_bfrag_rowmajor(grad + row * 16, 0, h);
                }
                // forward re-computation (same operations and roundings as k_ffmlp_forward)
#pragma unroll
                for (uint32_t m = 0; m < MB; m++) acc[m] = zero16();
#pragma unroll
                for (uint32_t s = 0; s < 4; s++)
                    if (s < KS0) {
#pragma unroll
                        for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(ff0[(m * KS0 + s) * 64 + lane], xf[s], acc[m]);
                    }
                DUO_STAMP();  // 1: inputs arrived, layer 0 issued
#pragma unroll
                for (uint32_t layer = 0; layer <= (uint32_t)NH; layer++) {
#pragma unroll
                    for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                        for (uint32_t r = 0; r < 16; r += 2) {
                            if constexpr (ACT == ACT_RELU) {
                                const half2p v = relu2(cvt2(acc[m][r], acc[m][r + 1]));
                                a[layer][2 * m + (r >> 3)][r & 7] = v[0];
                                a[layer][2 * m + (r >> 3)][(r & 7) + 1] = v[1];
                            } else {
#pragma unroll
                                for (uint32_t q = r; q < r + 2; q++) {
                                    const float pre = (float)(_Float16)acc[m][q];
                                    a[layer][2 * m + (q >> 3)][q & 7] = (_Float16)act_fwd_t<ACT>(act, pre);
                                }
                            }
                        }
                    if (layer < (uint32_t)NH) {
                        const half8* wf = ffh + (size_t)(layer * MB * KS) * 64;
#pragma unroll
                        for (uint32_t m = 0; m < MB; m++) {
                            acc[m] = zero16();
#pragma unroll
                            for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(wf[(m * KS + s) * 64 + lane], a[layer][s], acc[m]);
                        }
                    }
                }
                DUO_STAMP();  // 2: forward re-computed
                // stage 0: tiles of the last layer, then through it
                _Float16* T = slot_of(pair, i, 0);
                const half8 gtmp[1] = {gf};
                transpose_store<1>(T, gtmp, 1, n, h);
                transpose_store<(int)KS>(T + kTile, a[NH], KS, n, h);
#pragma unroll
                for (uint32_t m = 0; m < MB; m++) acc[m] = mfma(fbl[m * 64 + lane], gf, zero16());
            }
            DUO_STAMP();  // 3: stage 0 produced
            duo_barrier();
            DUO_STAMP();  // 4: past the barrier
#pragma unroll
            for (int k = NH; k >= 0; k--) {
                const uint32_t st = (uint32_t)(NH - k) + 1;
                if (live) {
                    // gradient w.r.t. the pre-activation of hidden layer k (activation transfer with the re-computed output)
#pragma unroll
                    for (uint32_t m = 0; m < MB; m++)
#pragma unroll
                        for (uint32_t r = 0; r < 16; r += 2) {
                            if constexpr (ACT == ACT_RELU) {
                                half2p av;
                                av[0] = a[k][2 * m + (r >> 3)][r & 7]; av[1] = a[k][2 * m + (r >> 3)][(r & 7) + 1];
                                const half2p v = gate2(cvt2(acc[m][r], acc[m][r + 1]), av);
                                G[2 * m + (r >> 3)][r & 7] = v[0];
                                G[2 * m + (r >> 3)][(r & 7) + 1] = v[1];
                            } else {
#pragma unroll
                                for (uint32_t q = r; q < r + 2; q++) {
                                    const float g = (float)(_Float16)acc[m][q];
                                    G[2 * m + (q >> 3)][q & 7] = (_Float16)act_bwd_t<ACT>(act, g, (float)a[k][2 * m + (q >> 3)][q & 7]);
                                }
                            }
                        }
                    _Float16* T = slot_of(pair, i, st);
                    transpose_store<(int)KS>(T, G, KS, n, h);
                    if (k > 0) {
                        transpose_store<(int)KS>(T + kTile, a[k - 1], KS, n, h);
                        const half8* wf = fbh + (size_t)((k - 1) * MB * KS) * 64;
#pragma unroll
                        for (uint32_t m = 0; m < MB; m++) {
                            acc[m] = zero16();
#pragma unroll
                            for (uint32_t s = 0; s < KS; s++) acc[m] = mfma(wf[(m * KS + s) * 64 + lane], G[s], acc[m]);
                        }
                    } else {
                        transpose_store<4>(T + kTile, xf, KS0, n, h);
                        if (grad_inputs) {
#pragma unroll
                            for (uint32_t m = 0; m < (uint32_t)IMB; m++) {
                                float16v gi = zero16();
#pragma unroll
                                for (uint32_t s = 0; s < KS; s++) gi = mfma(fb0[(m * KS + s) * 64 + lane], G[s], gi);
#pragma unroll
                                for (uint32_t q = 0; q < 4; q++) {
                                    const uint32_t feat = m * 32 + 8 * q + 4 * h;
                                    if (feat < in_dim) {
                                        half4 v;
#pragma unroll
                                        for (uint32_t e = 0; e < 4; e++) v[e] = (_Float16)gi[4 * q + e];
                                        store_grad_input(grad_inputs, in_layout, B, in_dim, row, feat, v);
                                    }
                                }
                            }
                        }
                    }
                }
                DUO_STAMP();  // 5 + 2j: stage produced
                duo_barrier();
                DUO_STAMP();  // 6 + 2j: past the barrier
            }
        }
        if (nit) duo_barrier();  // the partner's draining stage
    } else {
        // =============================================================== weight-gradient wave
        float16v dw0[MB][IMB], dwh[NH > 0 ? NH : 1][MB][MB], dwl[MB];
#pragma unroll
        for (uint32_t p = 0; p < MB; p++) {
#pragma unroll
            for (uint32_t q = 0; q < (uint32_t)IMB; q++) dw0[p][q] = zero16();
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)NH; k++)
#pragma unroll
                for (uint32_t q = 0; q < MB; q++) dwh[k][p][q] = zero16();
            dwl[p] = zero16();
        }
        // stage st of round i is consumed while the partner produces the next one: one barrier behind
        auto consume = [&](uint32_t i, auto stc) {
            constexpr uint32_t st = decltype(stc)::value;
#pragma unroll 1
            for (uint32_t j = 0; j < NR; j++) {
            const uint32_t cw = pair + j * (uint32_t)NG;  // (fixed order over the served compute waves: deterministic sums)
            if (tile_of(cw, i) >= ntiles) continue;
            const _Float16* T = slot_of(cw, i, st);
            const _Float16* TXs = T + kTile;
            if constexpr (st == 0) {
#pragma unroll
                for (uint32_t s = 0; s < 2; s++) {
                    const half8 af = transpose_load(T, 0, s, n, h, 16);
#pragma unroll
                    for (uint32_t ni = 0; ni < MB; ni++) dwl[ni] = mfma(af, transpose_load(TXs, ni, s, n, h, W), dwl[ni]);
                }
            } else if constexpr (st <= (uint32_t)NH) {
                constexpr uint32_t k = NH - (st - 1);  // hidden layer whose pre-activation gradient is in T
#pragma unroll
                for (uint32_t s = 0; s < 2; s++) {
                    half8 bfr[MB];
#pragma unroll
                    for (uint32_t ni = 0; ni < MB; ni++) bfr[ni] = transpose_load(TXs, ni, s, n, h, W);
#pragma unroll
                    for (uint32_t mo = 0; mo < MB; mo++) {
                        const half8 af = transpose_load(T, mo, s, n, h, W);
#pragma unroll
                        for (uint32_t ni = 0; ni < MB; ni++) dwh[k - 1][mo][ni] = mfma(af, bfr[ni], dwh[k - 1][mo][ni]);
                    }
                }
            } else {
#pragma unroll
                for (uint32_t s = 0; s < 2; s++) {
                    half8 bfr[IMB];
#pragma unroll
                    for (uint32_t ni = 0; ni < (uint32_t)IMB; ni++) bfr[ni] = transpose_load(TXs, ni, s, n, h, in_dim);
#pragma unroll
                    for (uint32_t mo = 0; mo < MB; mo++) {
                        const half8 af = transpose_load(T, mo, s, n, h, W);
#pragma unroll
                        for (uint32_t ni = 0; ni < (uint32_t)IMB; ni++) dw0[mo][ni] = mfma(af, bfr[ni], dw0[mo][ni]);
                    }
                }
            }
            }
        };
        for (uint32_t i = 0; i < nit; i++) {
            DUO_STAMP();
            if (i > 0) consume(i - 1, std::integral_constant<uint32_t, NS - 1>{});
            DUO_STAMP();
            duo_barrier();
            static_for<NS - 1>([&](auto stc) {
                DUO_STAMP();
                consume(i, stc);
                DUO_STAMP();
                duo_barrier();
            });
        }
        if (nit) {
            consume(nit - 1, std::integral_constant<uint32_t, NS - 1>{});
            duo_barrier();
        }
        // ---- sum the four weight-gradient waves in a fixed order through LDS (all 512 threads add), one partial per matrix
        float* red = reinterpret_cast<float*>(smem_raw);
        flush_matrix<MB, IMB, (NC + NG) * 64, NG>(red, partial, 0, pair, n, h, [&](auto mo, auto ni) { return dw0[mo][ni]; });
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NH; k++)
            flush_matrix<MB, MB, (NC + NG) * 64, NG>(red, partial, 1 + k, pair, n, h, [&](auto mo, auto ni) { return dwh[k][mo][ni]; });
        flush_matrix<1, MB, (NC + NG) * 64, NG>(red, partial, NH + 1, pair, n, h, [&](auto mo, auto ni) { (void)mo; return dwl[ni]; });
        return;
    }
    // compute waves: the same barriers as the flushes above, and their share of the sums
    float* red = reinterpret_cast<float*>(smem_raw);
    const float16v none = zero16();
    flush_matrix<MB, IMB, (NC + NG) * 64, NG>(red, partial, 0, NG + pair, n, h, [&](auto, auto) { return none; });
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)NH; k++) flush_matrix<MB, MB, (NC + NG) * 64, NG>(red, partial, 1 + k, NG + pair, n, h, [&](auto, auto) { return none; });
    flush_matrix<1, MB, (NC + NG) * 64, NG>(red, partial, NH + 1, NG + pair, n, h, [&](auto, auto) { return none; });
}

constexpr uint32_t kWgradBlocks = 256;
inline uint32_t wgrad_blocks(uint32_t W) { return W > 64 ? 128u : kWgradBlocks; }  // (W = 128: 64 KiB partial planes)
}  // namespace
// csrc/ffmlp_generic.hip: the shapes the register-resident kernels do not cover (hidden 16 / 256, input_dim > 64 at widths 32 / 64,
// hidden 128 beyond its LDS budget) run layer by layer on hand-written MFMA kernels
bool ffmlp_native_shape(uint32_t in_dim, uint32_t W, uint32_t n_layers = 2);
size_t ffmlp_generic_min_workspace(uint32_t in_dim, uint32_t W, uint32_t n_layers);
int ffmlp_generic_forward(const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t n_layers, uint32_t act,
                          uint32_t out_act, _Float16* acts, bool training, _Float16* out, hipStream_t st);
int ffmlp_generic_backward(const _Float16* grad, const _Float16* X, const _Float16* Wt, const _Float16* fwd, uint32_t B, uint32_t in_dim,
                           uint32_t W, uint32_t n_layers, uint32_t act, _Float16* bwd, _Float16* grad_inputs, _Float16* grad_weights,
                           bool accumulate, float* found_inf, float* workspace, size_t workspace_bytes, hipStream_t st);
namespace {

// `n_valid` of the entry point being served on this thread (see valid_rows()); the launch helpers below pass it on
static thread_local const int32_t* t_n_valid = nullptr;
static thread_local float* t_found_inf = nullptr;  // s3d_ffmlp_backward(found_inf) of the call being served
static thread_local float* t_rgb_out = nullptr;          // colour head of the forward call being served (fp32 [B, 3] out)
static thread_local const float* t_d_rgb = nullptr;      // ... of the backward call: gradient w.r.t. the head's output
static thread_local const float* t_rgb_in = nullptr;     // ... and the head's output
static thread_local MidFwd t_mid_fwd = {};               // density ("mid") head of the forward call being served
static thread_local MidBwd t_mid_bwd = {};               // ... of the backward call
static thread_local _Float16* t_generic_scratch = nullptr;  // inference_buffer of the s3d_ffmlp_inference call being served
struct RowLimitScope {
    explicit RowLimitScope(const int32_t* p, float* found_inf = nullptr) { t_n_valid = p; t_found_inf = found_inf; }
    ~RowLimitScope() { t_n_valid = nullptr; t_found_inf = nullptr; }
};

int check_shape(uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t W, uint32_t n_layers) {
    S3D_REQUIRE(W == 16 || W == 32 || W == 64 || W == 128 || W == 256,
                "FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got %u", W);  // ffmlp.cu:40-44
    S3D_REQUIRE(in_dim > 0 && in_dim % 16 == 0, "ffmlp: input_dim must be a multiple of 16 (got %u)", in_dim);
    S3D_REQUIRE(out_dim >= 1 && out_dim <= 16, "FFMLP current only supports output dim <= 16, but got %u", out_dim);
    S3D_REQUIRE(n_layers >= 2 && n_layers + 1 <= kMaxMlpLayers, "ffmlp: num_layers must be in [2, %u]", kMaxMlpLayers - 1);
    S3D_REQUIRE(B % 128 == 0, "ffmlp: batch size must be a multiple of 128 (got %u)", B);
    return S3D_OK;
}

template <int W>
int launch_forward(const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t n_layers,
                   uint32_t act, uint32_t out_act, _Float16* fwd, _Float16* out, uint32_t in_layout, hipStream_t st) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    const uint32_t nfr = MB * (in_dim / 16) + (n_layers - 1) * MB * KS + KS;
    const size_t smem = (size_t)nfr * 64 * sizeof(half8) + (t_mid_fwd.cin ? (size_t)4 * 32 * kMidRow * sizeof(_Float16) : 0);
    const uint32_t ntiles = B / 32;
    uint32_t grid = div_up<uint32_t>(ntiles, 4);
    // (measured at 2.6e5 points: 768 workgroups 11.8 / 14.6 us for the 2- / 3-matrix net, 1024: 12.7 / 16.2, 2048: 16.9 / 21.6,
    //  512: 12.2 / 14.7 — every workgroup stages all weights once, three per CU still hide the tile latencies)
    if (grid > 768) grid = 768;
    const bool big = smem > 64 * 1024;  // (W = 128: the dynamic allocation has to be announced; one workgroup per CU)
    if (big && grid > 256) grid = 256;
#define S3D_FWD_K(TRAIN, A, O, K0) do { \
        if (big) S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_forward<W, TRAIN, A, O, K0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_ffmlp_forward<W, TRAIN, A, O, K0>), dim3(grid), dim3(256), smem, st, X, Wt, B, in_dim, out_dim, n_layers, act, out_act, fwd, out, in_layout, t_n_valid, t_rgb_out, t_mid_fwd); } while (0)
#define S3D_FWD(TRAIN, A, O) do { if (in_dim == 32) S3D_FWD_K(TRAIN, A, O, 2); else if (in_dim == 64) S3D_FWD_K(TRAIN, A, O, 4); else S3D_FWD_K(TRAIN, A, O, 0); } while (0)
    const bool fast = act == ACT_RELU && out_act == ACT_NONE;  // the networks of the hot path; anything else: run-time switch
    if (fwd) { if (fast) S3D_FWD(true, ACT_RELU, ACT_NONE); else S3D_FWD(true, -1, -1); }
    else { if (fast) S3D_FWD(false, ACT_RELU, ACT_NONE); else S3D_FWD(false, -1, -1); }
#undef S3D_FWD_K
#undef S3D_FWD
    return check_launch("ffmlp_forward");
}

template <int W>
int launch_backward(const _Float16* grad, const _Float16* X, const _Float16* Wt, const _Float16* fwd, uint32_t B,
                    uint32_t in_dim, uint32_t out_dim, uint32_t n_layers, uint32_t act, _Float16* bwd,
                    _Float16* grad_inputs, _Float16* grad_weights, float* partial, uint32_t accumulate, hipStream_t st) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    const uint32_t NH = n_layers - 1;
    const uint32_t nfr = MB + NH * MB * KS + (grad_inputs ? ((in_dim + 31) / 32) * KS : 0);
    const size_t smem = (size_t)nfr * 64 * sizeof(half8);
    const uint32_t ntiles = B / 32;
    uint32_t grid = div_up<uint32_t>(ntiles, 4);
    if (grid > 1024) grid = 1024;
    if (smem > 64 * 1024) {  // (W = 128: one workgroup per CU holds every matrix's fragments)
        if (grid > 256) grid = 256;
        S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_dgrad<W, ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_dgrad<W, -1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    if (act == ACT_RELU)
        hipLaunchKernelGGL((k_ffmlp_dgrad<W, ACT_RELU>), dim3(grid), dim3(256), smem, st, grad, Wt, fwd, B, in_dim, out_dim, n_layers,
                           act, bwd, grad_inputs, t_n_valid);
    else
        hipLaunchKernelGGL((k_ffmlp_dgrad<W, -1>), dim3(grid), dim3(256), smem, st, grad, Wt, fwd, B, in_dim, out_dim, n_layers,
                           act, bwd, grad_inputs, t_n_valid);

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
    if (in_dim > (uint32_t)W) {
        // the operand tiles hold W columns: the first matrix's gradient as two column blocks, [0, W) and [W, in_dim)
        S3D_REQUIRE(plan.n < kMaxMlpLayers, "ffmlp_backward: too many layers for an input wider than the hidden width");
        plan.layer[0].Fi = (uint32_t)W; plan.layer[0].x_ld = in_dim; plan.layer[0].x_col0 = 0u; plan.layer[0].w_ld = in_dim;
        plan.layer[plan.n] = WgradLayer{bwd + NH * BW, X, 1u, 0u, (uint32_t)W, in_dim - (uint32_t)W, (uint32_t)W, in_dim, (uint32_t)W, in_dim};
        plan.n++;
    }
    uint32_t nblk = div_up<uint32_t>(ntiles, 4);
    if (nblk > wgrad_blocks(W)) nblk = wgrad_blocks(W);
    hipLaunchKernelGGL((k_ffmlp_wgrad<W>), dim3(nblk, plan.n), dim3(256), 0, st, plan, B, partial, t_n_valid);
    const uint32_t widest = (uint32_t)W > in_dim ? (uint32_t)W : in_dim;
    hipLaunchKernelGGL(k_ffmlp_wgrad_reduce, dim3(div_up<uint32_t>(W * widest * kReduceSplit, 256), plan.n), dim3(256), 0, st, plan, nblk,
                       (const float*)partial, grad_weights, accumulate, t_found_inf, wgrad_pad(W));
    return check_launch("ffmlp_backward");
}

// hidden [W x W] matrices the fused kernel keeps weight-gradient accumulators for (512 VGPRs per lane: W = 64 with three
// of them spills ~130 registers; two spill ~20, accepted)
inline uint32_t fused_max_hidden(uint32_t W) { return W == 64 ? 2u : 3u; }

inline bool fused_backward_supported(uint32_t in_dim, uint32_t out_dim, uint32_t W, uint32_t n_layers, uint32_t act) {
    if (!ffmlp_native_shape(in_dim, W)) return false;
    return (W == 32 || W == 64) && in_dim % 16 == 0 && in_dim >= 16 && in_dim <= 64 && out_dim >= 1 && out_dim <= 16 &&
           n_layers >= 2 && n_layers - 1 <= fused_max_hidden(W) && act != ACT_SINE;
}

template <int W, int NH, int IMB, int ACT, int KS0T>
int launch_backward_fused_k(const _Float16* grad, const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim,
                            uint32_t out_dim, uint32_t act, _Float16* grad_inputs, _Float16* grad_weights, float* partial,
                            uint32_t in_layout, uint32_t accumulate, hipStream_t st) {
    constexpr uint32_t MB = W / 32, KS = W / 16;
    // Wave mix of the two-role kernel, (compute waves, weight-gradient waves) per workgroup.  The compute wave of a 32-row tile
    // is 3.5x the work of its weight-gradient partner (profiles/r10_mlp_backward.md: 8.2 k against 2.3 k ticks per round), so one
    // weight-gradient wave serves THREE compute waves: 6 + 2 waves of <= 256 registers, two per SIMD, six tile streams per CU
    // behind the same weights in LDS (18 KiB of tile slots per stream).  Measured at 269,824 rows (profiles/r11_mlp_backward.md):
    // colour network 48.0 -> 45.5 us, density network 37.5 - 38.8 (six pairs) -> 36.9 us; 6 + 3 and 6 + 6 within noise of 6 + 2,
    // seven pairs 57 us (three waves per SIMD no longer fit the registers).  S3D_DUO_MIX / S3D_DUO_MIX_LIGHT = 4x4: round 5's pairs.
    constexpr bool light = W == 64 && NH == 1 && IMB == 1 && KS0T == 2 && ACT == ACT_RELU;
    const uint32_t nfrag = MB * (in_dim / 16) + NH * MB * KS + MB + NH * MB * KS + (grad_inputs ? IMB * KS : 0);
    static const bool duo = [] { const char* e = getenv("S3D_FFMLP_DUO"); return !(e && e[0] == '0'); }();  // A/B switch
    static const int mix = [] {
        const char* e = getenv(light ? "S3D_DUO_MIX_LIGHT" : "S3D_DUO_MIX");
        unsigned c = 0, g = 0;
        if (e && sscanf(e, "%ux%u", &c, &g) == 2) return (int)(c * 16 + g);
        return 6 * 16 + 2;
    }();
    const uint32_t ntiles = B / 32;
    // (four tile owners per workgroup decide the partial count whatever the kernel: s3d_ffmlp_wgrad_reduce_pair derives the
    //  layout of a deferred reduce from B alone; a workgroup of a small batch simply has idle streams)
    uint32_t nblk = div_up<uint32_t>(ntiles, 4);
    if (nblk > kWgradBlocks) nblk = kWgradBlocks;
    auto smem_for = [&](uint32_t owners, uint32_t slots, uint32_t planes) {
        size_t v = (size_t)nfrag * 64 * sizeof(half8) + (size_t)owners * slots * 2 * kTRows * kTRow * sizeof(_Float16);
        if (v < planes * kWgradPad * kWgradPad * sizeof(float)) v = planes * kWgradPad * kWgradPad * sizeof(float);  // epilogue planes
        return v;
    };
    auto run_duo = [&](auto ncc, auto ngc) -> int {
        constexpr int NCv = decltype(ncc)::value, NGv = decltype(ngc)::value;
        const size_t smem = smem_for(NCv, 2, NGv);
        S3D_REQUIRE(smem <= 160 * 1024, "ffmlp_backward: wave mix %dx%d needs %zu bytes of LDS", NCv, NGv, smem);
        static std::atomic<uint64_t> attr_devs{0};
        int dev;
        if (device_needs_setup(attr_devs, &dev)) {
            S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_backward_duo<W, NH, IMB, ACT, KS0T, NCv, NGv>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            device_setup_done(attr_devs, dev);
        }
        hipLaunchKernelGGL((k_ffmlp_backward_duo<W, NH, IMB, ACT, KS0T, NCv, NGv>), dim3(nblk), dim3((NCv + NGv) * 64), smem, st, grad, X, Wt,
                           B, in_dim, out_dim, act, grad_inputs, partial, in_layout, t_n_valid, t_d_rgb, t_rgb_in, t_mid_bwd);
        return S3D_OK;
    };
    using std::integral_constant;
    if (duo) {
        int rc;
        switch (mix) {
            case 4 * 16 + 4: rc = run_duo(integral_constant<int, 4>{}, integral_constant<int, 4>{}); break;  // (round 5's pairs: A/B)
            default: rc = run_duo(integral_constant<int, 6>{}, integral_constant<int, 2>{}); break;
        }
        if (rc) return rc;
    } else {
        static std::atomic<uint64_t> attr_devs{0};
        int dev;
        if (device_needs_setup(attr_devs, &dev)) {
            S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_backward_fused<W, NH, IMB, ACT, KS0T>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            device_setup_done(attr_devs, dev);
        }
        hipLaunchKernelGGL((k_ffmlp_backward_fused<W, NH, IMB, ACT, KS0T>), dim3(nblk), dim3(256), smem_for(4, 1, 4), st, grad, X, Wt, B, in_dim,
                           out_dim, act, grad_inputs, partial, in_layout, t_n_valid, t_d_rgb, t_rgb_in, t_mid_bwd);
    }
    WgradPlan plan;
    memset(&plan, 0, sizeof(plan));
    plan.n = NH + 2;
    plan.layer[0] = WgradLayer{nullptr, nullptr, 0u, 0u, (uint32_t)W, in_dim, 0u};
    for (uint32_t m = 0; m < (uint32_t)NH; m++)
        plan.layer[1 + m] = WgradLayer{nullptr, nullptr, 0u, 0u, (uint32_t)W, (uint32_t)W, (uint32_t)(W * in_dim + m * W * W)};
    plan.layer[NH + 1] = WgradLayer{nullptr, nullptr, 0u, 0u, 16u, (uint32_t)W, (uint32_t)(W * in_dim + NH * W * W)};
    if (accumulate != 2u)  // (2: the caller finishes several networks with one s3d_ffmlp_wgrad_reduce_pair)
        hipLaunchKernelGGL(k_ffmlp_wgrad_reduce, dim3(div_up<uint32_t>(W * W * kReduceSplit, 256), plan.n), dim3(256), 0, st, plan, nblk,
                           (const float*)partial, grad_weights, accumulate, t_found_inf, kWgradPad);
    return check_launch("ffmlp_backward (fused)");
}

template <int W>
int launch_backward_fused(const _Float16* grad, const _Float16* X, const _Float16* Wt, uint32_t B, uint32_t in_dim,
                          uint32_t out_dim, uint32_t n_layers, uint32_t act, _Float16* gi, _Float16* gw, float* partial,
                          uint32_t in_layout, uint32_t accumulate, hipStream_t st) {
    const uint32_t NH = n_layers - 1, IMB = (in_dim + 31) / 32;
#define S3D_FUSED_K(NHV, IMBV, ACTV, KSV) \
    launch_backward_fused_k<W, NHV, IMBV, ACTV, KSV>(grad, X, Wt, B, in_dim, out_dim, act, gi, gw, partial, in_layout, accumulate, st)
    // the hot path's networks (ReLU, in_dim a multiple of 32) get the layer-0 step count at compile time
#define S3D_FUSED(NHV, IMBV)                                                                                  \
    (act == ACT_RELU ? (in_dim == 32u * IMBV ? S3D_FUSED_K(NHV, IMBV, ACT_RELU, 2 * IMBV) : S3D_FUSED_K(NHV, IMBV, ACT_RELU, 0)) \
                     : S3D_FUSED_K(NHV, IMBV, -1, 0))
    if (IMB == 1) {
        if (NH == 1) return S3D_FUSED(1, 1);
        if (NH == 2) return S3D_FUSED(2, 1);
        if constexpr (W == 32) return S3D_FUSED(3, 1);
    } else {
        if (NH == 1) return S3D_FUSED(1, 2);
        if (NH == 2) return S3D_FUSED(2, 2);
        if constexpr (W == 32) return S3D_FUSED(3, 2);
    }
    set_error("ffmlp_backward: fused kernel does not cover %u hidden matrices at width %d", NH, W);
    return S3D_ERR_UNSUPPORTED;
#undef S3D_FUSED
#undef S3D_FUSED_K
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                                 uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                 uint32_t output_activation, uint16_t* forward_buffer, uint16_t* outputs,
                                 int input_layout, const int32_t* n_valid, float* rgb_head, const float* mid_dirs,
                                 float* mid_sigma, uint16_t* mid_color_in, uint16_t* mid_h0, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    const RowLimitScope rows(n_valid);
    struct HeadScope {
        HeadScope(float* p, const float* d, float* s, uint16_t* c, uint16_t* h0) {
            t_rgb_out = p;
            t_mid_fwd = MidFwd{};
            if (c) {
                t_mid_fwd.dirs = d; t_mid_fwd.sigma = s; t_mid_fwd.cin = (_Float16*)c; t_mid_fwd.h0 = (_Float16*)h0;
                host_sh_norm(4, t_mid_fwd.K);
            }
        }
        ~HeadScope() { t_rgb_out = nullptr; t_mid_fwd = MidFwd{}; }
    } head(rgb_head, mid_dirs, mid_sigma, mid_color_in, mid_h0);
    S3D_REQUIRE(!mid_color_in || (mid_dirs && mid_sigma && mid_h0 && !rgb_head && output_dim == 16),
                "ffmlp_forward: the density head needs dirs, sigma, color_in and h0 (and no colour head)");
    S3D_REQUIRE(inputs && weights && (outputs || rgb_head || mid_color_in), "ffmlp_forward: null pointer");
    S3D_REQUIRE(!rgb_head || output_dim >= 3, "ffmlp_forward: the colour head reads outputs 0..2");
    S3D_REQUIRE(input_layout == 0 || input_layout == 1, "ffmlp_forward: input_layout must be 0 (row-major) or 1 (level-major [in/2][B][2])");
    if (int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers)) return rc;
    S3D_REQUIRE(output_dim == 16, "ffmlp_forward: the output must be padded to 16 columns (ffmlp.py:117)");
    const _Float16* X = (const _Float16*)inputs; const _Float16* Wt = (const _Float16*)weights;
    _Float16* fb = (_Float16*)forward_buffer; _Float16* o = (_Float16*)outputs;
    if (!ffmlp_native_shape(input_dim, hidden_dim, num_layers)) {
        S3D_REQUIRE(input_layout == 0 && !n_valid && !rgb_head && !mid_color_in && outputs,
                    "ffmlp_forward: hidden_dim %u / input_dim %u take the layer-by-layer path, which implements the reference's "
                    "interface only (row-major inputs, no heads, no n_valid)", hidden_dim, input_dim);
        S3D_REQUIRE(forward_buffer || t_generic_scratch, "ffmlp_forward: this shape needs forward_buffer [n, B, W] (training) or "
                    "inference_buffer [2, B, W] (inference)");
        return ffmlp_generic_forward(X, Wt, B, input_dim, hidden_dim, num_layers, activation, output_activation,
                                     fb ? fb : t_generic_scratch, fb != nullptr, o, as_stream(stream));
    }
    if (hidden_dim == 64) return launch_forward<64>(X, Wt, B, input_dim, output_dim, num_layers, activation, output_activation, fb, o, (uint32_t)input_layout, as_stream(stream));
    if (hidden_dim == 128) {
        S3D_REQUIRE(!rgb_head && !mid_color_in, "ffmlp_forward: the NGP heads belong to the 64-wide networks");
        return launch_forward<128>(X, Wt, B, input_dim, output_dim, num_layers, activation, output_activation, fb, o, (uint32_t)input_layout, as_stream(stream));
    }
    return launch_forward<32>(X, Wt, B, input_dim, output_dim, num_layers, activation, output_activation, fb, o, (uint32_t)input_layout, as_stream(stream));
}

S3D_EXPORT int s3d_ffmlp_inference(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                                   uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                   uint32_t output_activation, uint16_t* inference_buffer, uint16_t* outputs,
                                   int input_layout, const int32_t* n_valid, float* rgb_head, const float* mid_dirs,
                                   float* mid_sigma, uint16_t* mid_color_in, uint16_t* mid_h0, s3d_stream_t stream) {
    struct Scratch { explicit Scratch(_Float16* p) { t_generic_scratch = p; } ~Scratch() { t_generic_scratch = nullptr; } }
        scratch((_Float16*)inference_buffer);  // (only the layer-by-layer path of the non-native shapes uses it: [2, B, W])
    return s3d_ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                             output_activation, nullptr, outputs, input_layout, n_valid, rgb_head, mid_dirs, mid_sigma,
                             mid_color_in, mid_h0, stream);
}

// CUs of the current device, cached per device ordinal (resident-workgroup counts of the persistent kernels)
static uint32_t cu_count() {
    static std::atomic<uint32_t> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    uint32_t n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (uint32_t)v : 256u;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

S3D_EXPORT int s3d_ffmlp_ngp_pair_inference(const uint16_t* inputs, const uint16_t* weights_sigma, const uint16_t* weights_color,
                                           uint32_t B, uint32_t hidden_dim, uint32_t num_layers_sigma, uint32_t num_layers_color,
                                           int input_layout, const int32_t* n_valid, const float* dirs, float* sigma,
                                           float* rgb, uint16_t* color_in, uint16_t* h0, const uint16_t* enc_color,
                                           s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && weights_sigma && weights_color && dirs && sigma && rgb, "ffmlp_ngp_pair_inference: null pointer");
    S3D_REQUIRE(hidden_dim == 64, "ffmlp_ngp_pair_inference: hidden_dim 64 (got %u)", hidden_dim);
    S3D_REQUIRE(input_layout == 0 || input_layout == 1, "ffmlp_ngp_pair_inference: input_layout must be 0 (row-major) or 1 (level-major [16][B][2])");
    const bool seal = enc_color != nullptr;
    if (int rc = check_shape(B, 32, 16, hidden_dim, num_layers_sigma)) return rc;
    if (int rc = check_shape(B, seal ? 64 : 32, 16, hidden_dim, num_layers_color)) return rc;
    MidFwd mid{};
    mid.dirs = dirs; mid.sigma = sigma; mid.cin = (_Float16*)color_in; mid.h0 = (_Float16*)h0;
    host_sh_norm(4, mid.K);
    PairNets nets{(const _Float16*)weights_sigma, (const _Float16*)weights_color, num_layers_sigma, num_layers_color,
                  (const _Float16*)enc_color};
    const uint32_t nfr = (4 + 4) + (seal ? 8 + 4 : 4 + 4) + (num_layers_sigma - 1 + num_layers_color - 1) * 8;
    const size_t smem = (size_t)nfr * 64 * sizeof(half8) + (size_t)kPairWaves * 32 * (seal ? 72 : kMidRow) * sizeof(_Float16);
    static std::atomic<uint64_t> attr_devs{0};
    int dev;
    if (device_needs_setup(attr_devs, &dev)) {
        const int cap = (int)((8 + 12 + 2 * (kMaxMlpLayers - 2) * 8) * 1024 + kPairWaves * 32 * 72 * 2);
        S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_ngp_pair<false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffmlp_ngp_pair<true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        device_setup_done(attr_devs, dev);
    }
    uint32_t grid = div_up<uint32_t>(B / 32, kPairWaves);
    // persistent workgroups, as many as are resident at once: two per CU (NGP: 60 KB of LDS, 114 registers), one per CU (Seal:
    // 148 registers).  [384 workgroups — three waves per SIMD on paper — left half of the CUs with two workgroups and half with
    // one: 93 us per render iteration of 1.6e6 rows against 79 at 512; Seal variant 118 against 98 at 256]
#ifdef S3D_PAIR_CAP  // (variant builds for A/B runs)
    const uint32_t cap = S3D_PAIR_CAP;
#else
    const uint32_t cap = (seal ? 1u : 2u) * cu_count() * 8u / kPairWaves;
#endif
    if (grid > cap) grid = cap;
    if (seal)
        hipLaunchKernelGGL(k_ffmlp_ngp_pair<true>, dim3(grid), dim3(kPairWaves * 64), smem, as_stream(stream), (const _Float16*)inputs, nets, B,
                           (uint32_t)input_layout, n_valid, rgb, mid);
    else
        hipLaunchKernelGGL(k_ffmlp_ngp_pair<false>, dim3(grid), dim3(kPairWaves * 64), smem, as_stream(stream), (const _Float16*)inputs, nets, B,
                           (uint32_t)input_layout, n_valid, rgb, mid);
    return check_launch("ffmlp_ngp_pair_inference");
}

S3D_EXPORT int s3d_ffmlp_wgrad_reduce_pair(const void* workspace_a, uint32_t B_a, uint32_t input_dim_a, uint32_t hidden_dim_a,
                                          uint32_t num_layers_a, uint16_t* grad_weights_a, int accumulate_a, float* found_inf_a,
                                          const void* workspace_b, uint32_t B_b, uint32_t input_dim_b, uint32_t hidden_dim_b,
                                          uint32_t num_layers_b, uint16_t* grad_weights_b, int accumulate_b, float* found_inf_b,
                                          s3d_stream_t stream) {
    S3D_REQUIRE(workspace_a && grad_weights_a && workspace_b && grad_weights_b, "ffmlp_wgrad_reduce_pair: null pointer");
    S3D_REQUIRE(num_layers_a >= 2 && num_layers_b >= 2 && num_layers_a + num_layers_b + 2 <= 2 * kMaxMlpLayers,
                "ffmlp_wgrad_reduce_pair: too many layers");
    S3D_REQUIRE(input_dim_a <= kWgradPad && input_dim_b <= kWgradPad && hidden_dim_a <= kWgradPad && hidden_dim_b <= kWgradPad,
                "ffmlp_wgrad_reduce_pair: the fused backward's shapes only");
    ReduceJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    uint32_t wmax = 0;
    auto add = [&](const void* ws, uint32_t B, uint32_t in_dim, uint32_t W, uint32_t nl, uint16_t* gw, int acc, float* fi) {
        uint32_t nblk = div_up<uint32_t>(B / 32, 4);  // (launch_backward_fused_k's partial count)
        if (nblk > kWgradBlocks) nblk = kWgradBlocks;
        const uint32_t NH = nl - 1, layers = NH + 2;
        for (uint32_t l = 0; l < layers; l++) {
            ReduceJob& j = jobs.job[jobs.n++];
            j.partial = (const float*)ws + (size_t)l * nblk * kWgradPad * kWgradPad;
            j.gw = (_Float16*)gw; j.found_inf = fi; j.nblk = nblk; j.accumulate = acc ? 1u : 0u;
            j.Fo = l == layers - 1 ? 16u : W;
            j.Fi = l == 0 ? in_dim : W;
            j.w_off = l == 0 ? 0u : W * in_dim + (l - 1) * W * W;
        }
        wmax = W > wmax ? W : wmax;
    };
    add(workspace_a, B_a, input_dim_a, hidden_dim_a, num_layers_a, grad_weights_a, accumulate_a, found_inf_a);
    add(workspace_b, B_b, input_dim_b, hidden_dim_b, num_layers_b, grad_weights_b, accumulate_b, found_inf_b);
    hipLaunchKernelGGL(k_ffmlp_wgrad_reduce_jobs, dim3(div_up<uint32_t>(wmax * kWgradPad * kReduceSplit, 256), jobs.n), dim3(256), 0,
                       as_stream(stream), jobs);
    return check_launch("ffmlp_wgrad_reduce_pair");
}

S3D_EXPORT size_t s3d_ffmlp_backward_workspace_size(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                                    uint32_t num_layers) {
    (void)output_dim;
    const size_t planes = (size_t)num_layers + 1 + (input_dim > hidden_dim && hidden_dim == 128 ? 1 : 0);  // (launch_backward: two column blocks)
    const size_t native = planes * wgrad_blocks(hidden_dim) * wgrad_pad(hidden_dim) * wgrad_pad(hidden_dim) * sizeof(float);
    if (ffmlp_native_shape(input_dim, hidden_dim, num_layers)) return native;
    // layer-by-layer path: up to 16 batch segments' worth of fp32 partial planes of every weight matrix
    return std::max(native, 16 * ffmlp_generic_min_workspace(input_dim, hidden_dim, num_layers));
}

S3D_EXPORT int s3d_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                                  const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                  uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                  uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer,
                                  uint16_t* grad_inputs, uint16_t* grad_weights, void* workspace, size_t workspace_bytes,
                                  int input_layout, int accumulate_grad_weights, const int32_t* n_valid, float* found_inf,
                                  const float* grad_rgb, const float* rgb_head, const float* mid_grad_sigma,
                                  const uint16_t* mid_grad_color_in, const uint16_t* mid_h0, s3d_stream_t stream) {
    (void)output_activation;
    const RowLimitScope rows(n_valid, found_inf);
    struct HeadScope {
        HeadScope(const float* g, const float* y, const float* gs, const uint16_t* gc, const uint16_t* h0) {
            t_d_rgb = g; t_rgb_in = y;
            t_mid_bwd = MidBwd{gs, (const _Float16*)gc, (const _Float16*)h0};
        }
        ~HeadScope() { t_d_rgb = nullptr; t_rgb_in = nullptr; t_mid_bwd = MidBwd{}; }
    } head(grad_rgb, rgb_head, mid_grad_sigma, mid_grad_color_in, mid_h0);
    S3D_REQUIRE(!mid_grad_color_in || (mid_h0 && !forward_buffer && !grad_rgb && output_dim == 16),
                "ffmlp_backward: the density head needs grad_color_in and h0 and is implemented by the fused backward");
    S3D_REQUIRE((grad_rgb == nullptr) == (rgb_head == nullptr), "ffmlp_backward: the colour head needs both grad_rgb and rgb_head");
    S3D_REQUIRE(!grad_rgb || (!forward_buffer && output_dim >= 3), "ffmlp_backward: the colour head is implemented by the fused backward");
    const uint32_t accumulate = accumulate_grad_weights == 2 ? 2u : (accumulate_grad_weights ? 1u : 0u);
    S3D_REQUIRE(accumulate != 2u || (!forward_buffer && ffmlp_native_shape(input_dim, hidden_dim) &&
                                     fused_backward_supported(input_dim, output_dim, hidden_dim, num_layers, activation)),
                "ffmlp_backward: accumulate_grad_weights = 2 (deferred reduce) is implemented by the fused backward");
    S3D_REQUIRE(input_layout == 0 || (input_layout == 1 && !forward_buffer),
                "ffmlp_backward: the level-major input layout is implemented by the fused backward (no forward_buffer)");
    if (B == 0) return S3D_OK;
    S3D_REQUIRE((grad || grad_rgb || mid_grad_color_in) && inputs && weights && grad_weights, "ffmlp_backward: null pointer");
    S3D_REQUIRE((forward_buffer == nullptr) == (backward_buffer == nullptr),
                "ffmlp_backward: pass both forward_buffer and backward_buffer, or neither (fused re-computing backward)");
    if (int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers)) return rc;
    S3D_REQUIRE(output_dim == 16, "ffmlp_backward: the output must be padded to 16 columns (ffmlp.py:117)");
    S3D_REQUIRE(activation != ACT_SINE, "ffmlp_backward: sine needs pre-activations, unsupported (utils.h:546-550)");
    S3D_REQUIRE(!calc_grad_inputs || grad_inputs, "ffmlp_backward: grad_inputs requested but null");
    S3D_REQUIRE(workspace && workspace_bytes >= s3d_ffmlp_backward_workspace_size(input_dim, output_dim, hidden_dim, num_layers),
                "ffmlp_backward: workspace too small");
    _Float16* gi = calc_grad_inputs ? (_Float16*)grad_inputs : nullptr;
    if (!ffmlp_native_shape(input_dim, hidden_dim, num_layers)) {
        S3D_REQUIRE(forward_buffer && backward_buffer && grad && input_layout == 0 && !n_valid && !grad_rgb && !mid_grad_color_in,
                    "ffmlp_backward: hidden_dim %u / input_dim %u take the layer-by-layer path: forward_buffer and backward_buffer "
                    "[n, B, W], row-major inputs, no heads, no n_valid", hidden_dim, input_dim);
        return ffmlp_generic_backward((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights, (const _Float16*)forward_buffer,
                                      B, input_dim, hidden_dim, num_layers, activation, (_Float16*)backward_buffer, gi,
                                      (_Float16*)grad_weights, accumulate != 0, found_inf, (float*)workspace, workspace_bytes, as_stream(stream));
    }
    if (!forward_buffer) {
        S3D_REQUIRE(fused_backward_supported(input_dim, 16, hidden_dim, num_layers, activation),
                    "ffmlp_backward: this network shape needs forward_buffer/backward_buffer (s3d_ffmlp_fused_backward_supported)");
        if (hidden_dim == 64)
            return launch_backward_fused<64>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights, B, input_dim,
                                             output_dim, num_layers, activation, gi, (_Float16*)grad_weights, (float*)workspace,
                                             (uint32_t)input_layout, accumulate, as_stream(stream));
        return launch_backward_fused<32>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights, B, input_dim,
                                         output_dim, num_layers, activation, gi, (_Float16*)grad_weights, (float*)workspace,
                                         (uint32_t)input_layout, accumulate, as_stream(stream));
    }
    if (hidden_dim == 64)
        return launch_backward<64>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights,
                                   (const _Float16*)forward_buffer, B, input_dim, output_dim, num_layers, activation,
                                   (_Float16*)backward_buffer, gi, (_Float16*)grad_weights, (float*)workspace, accumulate, as_stream(stream));
    if (hidden_dim == 128)
        return launch_backward<128>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights,
                                    (const _Float16*)forward_buffer, B, input_dim, output_dim, num_layers, activation,
                                    (_Float16*)backward_buffer, gi, (_Float16*)grad_weights, (float*)workspace, accumulate, as_stream(stream));
    return launch_backward<32>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)weights,
                               (const _Float16*)forward_buffer, B, input_dim, output_dim, num_layers, activation,
                               (_Float16*)backward_buffer, gi, (_Float16*)grad_weights, (float*)workspace, accumulate, as_stream(stream));
}

S3D_EXPORT int s3d_ffmlp_fused_backward_supported(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                                  uint32_t num_layers, uint32_t activation) {
    return fused_backward_supported(input_dim, output_dim, hidden_dim, num_layers, activation) ? 1 : 0;
}

S3D_EXPORT int s3d_ffmlp_allocate_splitk(size_t n) { (void)n; return S3D_OK; }
S3D_EXPORT int s3d_ffmlp_free_splitk(void) { return S3D_OK; }

#ifdef S3D_FFMLP_PROF
S3D_EXPORT int s3d_debug_ffmlp_prof_read(unsigned long long* dst, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return S3D_ERR_HIP;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(s3d_ffmlp_prof), sizeof(unsigned long long) * 2 * 1024) != hipSuccess) return S3D_ERR_HIP;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(s3d_ffmlp_prof)) != hipSuccess) return S3D_ERR_HIP;
        if (hipMemset(p, 0, sizeof(unsigned long long) * 2 * 1024) != hipSuccess) return S3D_ERR_HIP;
    }
    return S3D_OK;
}
#endif
